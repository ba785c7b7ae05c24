#!/bin/bash
# round 3, GPU call 3: full suite on the current build, ChEES pipelined colstats A/B, NUTS tail (callable-first sequences)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c3
rm -rf $O; mkdir -p $O
cd $R
(time timeout 1500 python -m pytest tests/ -q -m gpu) > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
bash tools/ab_chees.sh > $O/ab_chees.txt 2>&1
cat $O/ab_chees.txt
for T in 20 100 400; do
  timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/nuts_T${T}.json 2>> $O/nuts.err
done
timeout 300 python tools/bench_nuts.py --use-graph --steps 5 > $O/nuts_lockstep.json 2>> $O/nuts.err
for f in $O/nuts_*.json; do echo $f; python -c "import json; j=json.load(open('$f')); print(j['value']/1e6, j.get('ticks'), j.get('tick_period_avg_us'))"; done
tail -5 $O/nuts.err
