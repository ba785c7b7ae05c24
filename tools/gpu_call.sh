#!/bin/bash
# One GPU call = a tagged list of commands (replaces the per-call scripts of earlier rounds):
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <tag> "<command 1>" "<command 2>" ...'
# Every command runs from the repo root under `timeout ${BJX_CALL_TIMEOUT:-900}`; its stdout goes to
# gpurun_out/<tag>/<i>.out, stderr to <i>.err, and the last lines of both are echoed so they show in
# gpurun's tail.  A command prefixed with "prof:" runs under `rocprofv3 --kernel-trace --stats` (from /tmp,
# CSV into gpurun_out/<tag>/<i>_prof; the raw kernel trace is deleted, the stats files kept).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
i=0
for cmd in "$@"; do
  i=$((i + 1))
  cd $R
  if [[ "$cmd" == prof:* ]]; then
    c=${cmd#prof:}
    (cd /tmp; export TMPDIR=/tmp; timeout ${BJX_CALL_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv \
       -d $O/${i}_prof -- bash -c "cd $R && $c" > $O/$i.out 2> $O/$i.err)
    rc=$?
    find $O/${i}_prof -name '*kernel_trace.csv' -delete 2>/dev/null
  else
    (timeout ${BJX_CALL_TIMEOUT:-900} bash -c "$cmd" > $O/$i.out 2> $O/$i.err)
    rc=$?
  fi
  echo "== [$i] rc=$rc :: $cmd"
  tail -c 1500 $O/$i.out | tail -8
  [ $rc -ne 0 ] && tail -5 $O/$i.err
done
exit 0
