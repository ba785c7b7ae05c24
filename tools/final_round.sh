set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
bash tools/round_summary.sh > gpurun_out/round_summary.log 2>&1; tail -3 gpurun_out/round_summary.log
