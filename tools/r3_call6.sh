#!/bin/bash
# round 3, GPU call 6: NUTS tail with lagged polling, lockstep with long deep chunks, MEADS test tolerance
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c6
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_ghmc_gpu.py tests/test_nuts_free_gpu.py tests/test_nuts_gpu.py tests/test_nuts_free_adapt_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -4 $O/tests.log
for T in 20 100 400; do
  timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/nuts_T${T}.json 2>> $O/nuts.err
done
timeout 300 python tools/bench_nuts.py --use-graph --steps 8 > $O/nuts_lockstep.json 2>> $O/nuts.err
BJX_NUTS_MIN_BUCKET=32 timeout 300 python tools/bench_nuts.py --use-graph --steps 8 > $O/nuts_lockstep_b32.json 2>> $O/nuts.err
for f in $O/nuts_*.json; do echo $f; python -c "import json; j=json.load(open('$f')); print(j['value']/1e6, j.get('ticks'), j.get('tick_period_avg_us'), j.get('ms_per_transition'))"; done
tail -3 $O/nuts.err
