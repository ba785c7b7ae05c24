#!/bin/bash
# ChEES pooled statistics: parity tests, the fused weights + column-statistics pass against the two
# launches it replaces (BJX_CHEES_UNFUSED=1), and a kernel-level split from rocprofv3.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/chees
(timeout 600 python -m pytest tests/test_chees_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/chees/tests.log
cat gpurun_out/chees/tests.log
for v in fused fused_sb unfused; do
  f=$R/gpurun_out/chees/$v.json
  unset BJX_CHEES_UNFUSED BJX_WCOL_DB
  if [ $v = unfused ]; then export BJX_CHEES_UNFUSED=1; fi
  if [ $v = fused_sb ]; then export BJX_WCOL_DB=0; fi
  python tools/bench_chees.py > $f 2>/dev/null
  python - <<PY
import json
d=json.load(open("$f"))
print("$v", round(d["value"]/1e6,1), round(d["pooled_statistics_ms_per_step"],3), {k:(round(v["avg_us"],1), round(v["GBps"])) for k,v in d["kernels"].items() if "leapfrog" not in k})
PY
done
unset BJX_WCOL_DB; export BJX_CHEES_UNFUSED=1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chees/prof -- python $R/tools/bench_chees.py --num-steps 30 > /dev/null 2>&1
cd $R
f=$(ls -t gpurun_out/chees/prof/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/chees/kernel_stats.csv
head -14 $f | cut -c1-160
