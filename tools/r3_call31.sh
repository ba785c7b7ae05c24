#!/bin/bash
# round 3, GPU call 31: start-to-start period of a lockstep deep-doubling leaf vs a free-running tail tick
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c31
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_lock -- python $R/tools/bench_nuts.py --use-graph --steps 3 --warmup 4 > $O/lock.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/kt_free -- python $R/tools/bench_nuts.py --free-running --steps 400 --no-tick-timing > $O/free.log 2>&1
cd $R
echo "== lockstep (last 1200 kernels before the last k_nuts_post)"
python tools/trace_window.py $(ls $O/kt_lock/*/*kernel_trace.csv | head -1) k_nuts_post 1200
echo "== free-running tail (last 1200 kernels before the last tick)"
python tools/trace_window.py $(ls $O/kt_free/*/*kernel_trace.csv | head -1) async_tick2 1200
rm -f $O/kt_*/*/*kernel_trace.csv
