#!/bin/bash
# round 3, GPU call 15: step() through the free-running engine with an engine-resident target
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c15
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_free_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -4 $O/tests.log
timeout 300 python tools/bench_nuts.py --use-graph --steps 8 --warmup 3 > $O/lockstep.json 2>> $O/nuts.err
timeout 300 python tools/bench_nuts.py --steps 8 --warmup 3 --fuse-target > $O/lockstep_fused.json 2>> $O/nuts.err
for f in lockstep lockstep_fused; do python -c "import json; j=json.load(open('$O/$f.json')); print('$f', round(j['value']/1e6,1), round(j['ms_per_transition'],2), round(j['frac_of_52B_roofline'],3))"; done
for T in 20 100 400; do
  timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/nuts_T${T}.json 2>> $O/nuts.err
  python -c "import json; j=json.load(open('$O/nuts_T${T}.json')); print('default fused T=$T', round(j['value']/1e6,1), j.get('ticks'), round(j.get('tick_period_avg_us'),2), round(j['frac_of_52B_roofline'],3))"
done
tail -3 $O/nuts.err
