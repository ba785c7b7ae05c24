#!/bin/bash
# ChEES pooled statistics: parity tests, then the fused weights + column-statistics pass in its two
# forms (whole rows per wave = default for 128 < D <= 1024; BJX_CHEES_WCOL=1 = a row spread over the
# threads of a workgroup) against the two launches they replace (BJX_CHEES_UNFUSED=1).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/chees
(timeout 600 python -m pytest tests/test_chees_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/chees/tests.log
cat gpurun_out/chees/tests.log
for rep in 1 2; do
for v in rows cols unfused; do
  f=$R/gpurun_out/chees/$v.json
  unset BJX_CHEES_UNFUSED BJX_CHEES_WCOL
  if [ $v = unfused ]; then export BJX_CHEES_UNFUSED=1; fi
  if [ $v = cols ]; then export BJX_CHEES_WCOL=1; fi
  python tools/bench_chees.py > $f 2>/dev/null
  python - <<PY
import json
d=json.load(open("$f"))
print("$v", round(d["value"]/1e6,1), round(d["pooled_statistics_ms_per_step"],3), {k:(round(v["avg_us"],1), round(v["GBps"])) for k,v in d["kernels"].items() if "leapfrog" not in k})
PY
done; done
