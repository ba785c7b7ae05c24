#!/bin/bash
# round 3, first GPU call: yardsticks (membw2, tickbw), the GPU suite, NUTS baseline + tiered tail A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c1
rm -rf $O; mkdir -p $O
cd $R
./tools/membw2 > $O/membw2.json 2> $O/membw2.err
./tools/tickbw > $O/tickbw.json 2> $O/tickbw.err
(time timeout 1200 python -m pytest tests/ -x -q -m gpu) > $O/gpu_tests.log 2>&1
for T in 20 100 400; do
  BJX_NUTS_TAIL_TIERS=1 timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/nuts_T${T}_tiers1.json 2>> $O/nuts.err
  BJX_NUTS_TAIL_TIERS=0 timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/nuts_T${T}_tiers0.json 2>> $O/nuts.err
done
timeout 300 python tools/bench_nuts.py --use-graph --steps 5 > $O/nuts_lockstep.json 2>> $O/nuts.err
tail -3 $O/gpu_tests.log
for f in $O/nuts_*.json; do echo $f; python -c "import json,sys; j=json.load(open('$f')); print(j['value']/1e6, j.get('ticks'), j.get('tick_period_avg_us'))"; done
