set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tail_trace; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/tools/bench_nuts.py --free-running --steps 400 --no-tick-timing > $O/kt.log 2>&1
cd $R
F=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python tools/nuts_trace_tail.py $F 4000 > $O/tail.txt 2>&1
python tools/trace_window.py $F async_tick3 6000 > $O/window.txt 2>&1
rm -rf $O/kt
head -20 $O/tail.txt; head -8 $O/window.txt
