#!/bin/bash
# round 3, GPU call 24: order of the LDS operand reads in the dense core loop (A/B on one box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c24
rm -rf $O; mkdir -p $O
S=/tmp/ab_tree
rm -rf $S; mkdir -p $S; cp -r $R/blackjax_amd $R/include $R/tools $S/
cd $S/blackjax_amd/csrc; touch bjx_dense.hip
make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DBJX_DENSE_READ_SPLIT=1" > $O/build.log 2>&1
tail -1 $O/build.log
for rep in 1 2; do
 for V in base split; do
  if [ $V = base ]; then cd $R; else cd $S; fi
  for A in 0 6; do
   BJX_DENSE_ABLATE=$A timeout 300 python tools/bench_dense.py > $O/${V}_a${A}_$rep.json 2>> $O/dense.err
   python -c "import json; j=json.load(open('$O/${V}_a${A}_$rep.json')); r=j['roofline']; print('$V ablate $A rep $rep', round(r['avg_launch_us'],1), 'us', round(r['frac'],3))"
  done
 done
done
cd $S; timeout 600 python -m pytest $R/tests/test_dense_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
