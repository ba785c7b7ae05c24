set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/t100_trace; rm -rf $O; mkdir -p $O
cd $R; python tools/scratch/run_t100.py ${1:-100} 3 > $O/plain.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/tools/scratch/run_t100.py ${1:-100} 2 > $O/kt.log 2>&1
cd $R
F=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python tools/nuts_trace_phases.py $F 250 > $O/phases.txt 2>&1
rm -rf $O/kt
cat $O/plain.log | grep run; grep "^run" $O/kt.log; cat $O/phases.txt | tail -110
