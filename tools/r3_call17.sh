#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c17
rm -rf $O; mkdir -p $O
cd $R
BJX_MULTI_WAVES=2 timeout 300 python tools/nuts_tail_clock_probe.py 400 > $O/probe.json 2> $O/probe.err
python - <<PY
import json
j=json.load(open("$O/probe.json"))
print("alone", round(j["alone"]["M_per_s"],1), round(j["alone"]["s"],3))
print("with load", round(j["with_gemm_load_on_a_side_stream"]["M_per_s"],1), round(j["with_gemm_load_on_a_side_stream"]["s"],3))
for s in j["alone"]["clock_samples"][:30]: print(s)
PY
tail -3 $O/probe.err
