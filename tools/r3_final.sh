#!/bin/bash
# round 3: the judged evidence on ONE box -- C2 profile set, then the whole-round summary
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
bash tools/round_summary.sh > gpurun_out/round_summary.log 2>&1; tail -3 gpurun_out/round_summary.log
bash tools/pmc_nuts2.sh > gpurun_out/pmc_nuts2.log 2>&1; tail -8 gpurun_out/pmc_nuts2.log
