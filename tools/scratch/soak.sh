# repeats the GPU suite; keeps the log of every iteration that did not end in "passed"
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/soak; rm -rf $O; mkdir -p $O
cd $R
for i in $(seq 1 ${1:-4}); do
  PYTHONFAULTHANDLER=1 python -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider ${2:-} > $O/it$i.log 2>&1
  rc=$?
  echo "iteration $i rc=$rc: $(tail -1 $O/it$i.log | cut -c1-150)"
  if [ $rc -eq 0 ]; then rm $O/it$i.log; else grep -n "Fatal\|Segmentation\|Aborted\|Current thread\|File \"" $O/it$i.log | head -60 > $O/it$i.summary; fi
done
ls $O
