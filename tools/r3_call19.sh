#!/bin/bash
# round 3, GPU call 19: A/B on one box: speculative target evaluation on / off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c19
rm -rf $O; mkdir -p $O
S=/tmp/ab_tree
rm -rf $S; mkdir -p $S; cp -r $R/blackjax_amd $R/include $R/tools $S/
cd $S/blackjax_amd/csrc
touch bjx_nuts.hip
make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DBJX_NO_SPEC" > $O/build.log 2>&1
tail -1 $O/build.log
for rep in 1 2; do
for V in spec nospec; do
  if [ $V = spec ]; then cd $R; else cd $S; fi
  for T in 100 400; do
    timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/${V}_T${T}_$rep.json 2>> $O/nuts.err
    python -c "import json; j=json.load(open('$O/${V}_T${T}_$rep.json')); print('$V rep $rep T=$T', round(j['value']/1e6,1), round(j.get('tick_period_avg_us'),2))"
  done
done
done
cd $R
for T in 100 400; do
  timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/ext_T${T}.json 2>> $O/nuts.err
  python -c "import json; j=json.load(open('$O/ext_T${T}.json')); print('external callable T=$T', round(j['value']/1e6,1), round(j.get('tick_period_avg_us'),2))"
done
timeout 300 python tools/bench_nuts.py --use-graph --steps 5 > $O/lockstep.json 2>> $O/nuts.err
python -c "import json; j=json.load(open('$O/lockstep.json')); print('lockstep', round(j['value']/1e6,1))"
