#!/bin/bash
# round 3, GPU call 28: the transition end as a real call in the multi-tick kernel (more waves per SIMD?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c28
rm -rf $O; mkdir -p $O
S=/tmp/ab_tree
rm -rf $S; mkdir -p $S; cp -r $R/blackjax_amd $R/include $R/tools $R/tests $R/oracle $S/
cd $S/blackjax_amd/csrc; touch bjx_nuts.hip
make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DBJX_MULTI_END_CALL" > $O/build.log 2>&1
tail -1 $O/build.log
cd $S; timeout 600 python -m pytest tests/test_nuts_free_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
for V in base call; do
  if [ $V = base ]; then cd $R; else cd $S; fi
  for W in 2 3 4; do
    for T in 20 400; do
      BJX_MULTI_WAVES=$W timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/${V}_w${W}_T$T.json 2>> $O/nuts.err
      python -c "import json; j=json.load(open('$O/${V}_w${W}_T$T.json')); print('$V waves $W T=$T', round(j['value']/1e6,1), round(j.get('tick_period_avg_us'),2))"
    done
  done
done
