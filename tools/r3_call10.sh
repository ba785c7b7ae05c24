#!/bin/bash
# round 3, GPU call 10: kernel split of dense-metric NUTS on the GEMM
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c10
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/bench_nuts_dense.py --mode gemm --steps 3 > $O/nuts_dense.json 2> $O/err.txt
cd $R
F=$(ls $O/kt/*/*kernel_stats.csv | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
    print(f"{n:70s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
cp $F $O/nuts_dense_kernel_stats.csv; rm -rf $O/kt
