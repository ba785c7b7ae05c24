#!/bin/bash
# round 3, GPU call 11: dense-metric NUTS on the GEMM with 16-byte sweeps (parity + bench + kernel split)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c11
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_gpu.py tests/test_integrators_samplers_gpu.py tests/test_nuts_free_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -4 $O/tests.log
timeout 900 python tools/bench_nuts_dense.py --steps 5 --warmup 3 > $O/nuts_dense.json 2> $O/nuts_dense.err; cat $O/nuts_dense.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/bench_nuts_dense.py --mode gemm --steps 3 --warmup 3 > /dev/null 2> $O/err.txt
cd $R
F=$(ls $O/kt/*/*kernel_stats.csv | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    print(f"{n:60s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
cp $F $O/nuts_dense_kernel_stats.csv; rm -rf $O/kt
