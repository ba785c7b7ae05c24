#!/bin/bash
# round 3, GPU call 18: where does a multi-tick leaf spend its time (instrumented scratch build)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c18
rm -rf $O; mkdir -p $O
S=/tmp/probe_tree
rm -rf $S; mkdir -p $S; cp -r $R/blackjax_amd $R/include $R/tools $S/
cd $S/blackjax_amd/csrc
touch bjx_nuts.hip
make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DBJX_TICK_PROBE" > $O/build.log 2>&1
tail -2 $O/build.log
cd $S
for N in 4 64 32768; do
  timeout 300 python tools/nuts_tick_probe.py 100 $N > $O/probe_N$N.json 2>> $O/probe.err
  echo "== N=$N"; cat $O/probe_N$N.json
done
tail -3 $O/probe.err
