#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c20
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_free_gpu.py tests/test_nuts_free_adapt_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -5 $O/tests.log
timeout 600 python tools/bench_nuts_warmup.py > $O/nuts_warmup.json 2> $O/warm.err
python -c "
import json; j=json.load(open('$O/nuts_warmup.json'))
for k in ('free_running','free_running_engine_resident_target','lockstep'): print(k, round(j[k]['value']/1e6,1), round(j[k]['seconds'],3), j[k].get('identical_to_free_running'))
print(j.get('identical_results'), j.get('speedup'))"
tail -3 $O/warm.err
