#!/bin/bash
# round 3, GPU call 12: engine-resident target inside the tick kernels (fuse_target): parity + C3 numbers
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c12
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_free_gpu.py tests/test_nuts_free_adapt_gpu.py tests/test_nuts_large_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -6 $O/tests.log
for T in 20 100 400; do
  timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/nuts_T${T}.json 2>> $O/nuts.err
  timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/nuts_T${T}_fused.json 2>> $O/nuts.err
done
for f in $O/nuts_*.json; do echo $f; python -c "import json; j=json.load(open('$f')); print(round(j['value']/1e6,1), j.get('ticks'), round(j.get('tick_period_avg_us'),2), round(j['frac_of_52B_roofline'],3))"; done
tail -3 $O/nuts.err
