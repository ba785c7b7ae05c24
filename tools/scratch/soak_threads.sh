R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/soak_threads; rm -rf $O; mkdir -p $O
cd $R
fails=0
for i in $(seq 1 ${1:-30}); do
  python -X faulthandler -m pytest tests/test_nuts_run_workspace_gpu.py tests/test_edge_cases_gpu.py -x -q -k "interleaved" -p no:cacheprovider > $O/it$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "iteration $i rc=$rc"; grep -n "Fatal\|Segmentation\|Aborted\|thread 0x\|File \"\|Error\|error" $O/it$i.log | head -50; else rm $O/it$i.log; fi
done
echo "fails: $fails of ${1:-30}"
