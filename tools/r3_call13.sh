#!/bin/bash
# round 3, GPU call 13: multi-tick launches with an engine-resident target
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c13
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_free_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -6 $O/tests.log
run() { # name, env..., T
  local name=$1; shift
  for T in 20 100 400; do
    env "$@" timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/nuts_T${T}_$name.json 2>> $O/nuts.err
    python -c "import json; j=json.load(open('$O/nuts_T${T}_$name.json')); print('$name T=$T', round(j['value']/1e6,1), j.get('ticks'), round(j.get('tick_period_avg_us'),2), round(j['frac_of_52B_roofline'],3))"
  done
}
run multi BJX_X=1
run k1 BJX_NUTS_MULTI_TICK_ROWS=0
run allfused BJX_NUTS_FUSED_ROWS=1000000
tail -3 $O/nuts.err
