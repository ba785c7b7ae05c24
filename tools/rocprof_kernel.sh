#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command; prints the stats rows whose kernel name matches a pattern.
# usage: tools/rocprof_kernel.sh <out-dir under gpurun_out> <pattern> <command...>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; P=$2; shift 2
rm -rf $O; mkdir -p $O
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O -- bash -c "cd $R && $*" > $O/cmd.out 2> $O/cmd.err)
find $O -name '*kernel_trace.csv' -delete 2>/dev/null
f=$(find $O -name '*kernel_stats.csv' | head -1)
head -1 $f; grep -E "$P" $f | head -5
