#!/bin/bash
# round 3, GPU call 14: engine-resident target, every chunk one launch: sweep of the chunk length
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c14
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_free_gpu.py tests/test_nuts_free_adapt_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -4 $O/tests.log
for SE in 16 32 64 128; do
  for T in 20 100 400; do
    BJX_NUTS_SYNC_EVERY=$SE timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/nuts_T${T}_se$SE.json 2>> $O/nuts.err
    python -c "import json; j=json.load(open('$O/nuts_T${T}_se$SE.json')); print('sync_every $SE T=$T', round(j['value']/1e6,1), j.get('ticks'), round(j.get('tick_period_avg_us'),2), round(j['frac_of_52B_roofline'],3))"
  done
done
tail -3 $O/nuts.err
