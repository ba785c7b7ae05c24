#!/bin/bash
# Evidence for bench.py's user-callable lines (VERDICT r2 "next" #6): for the two PyTorch forms of the
# user log-density -- torch_autograd (lambda q: -0.5 * (q*q*inv_var).sum(-1) through autograd) and
# torch_pair (a plain-torch (logp, grad) pair) -- at the C2 shape:
#   * rocprofv3 --kernel-trace --stats of a run whose ONLY timed region is that mode
#   * FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md), summed over EVERY kernel
#     of the process -> measured bytes per chain-leapfrog element (engine + callable)
# Outputs under gpurun_out/torch_modes/; `python tools/collect_torch_modes.py r03` writes
# profiles/r03/torch_modes_*.{json,csv} and profiles/torch_modes_latest.json (read by bench.py).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/torch_modes
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
COMMON="--steps 2 --warmup 1 --chain-block 16384 --no-cpu-baseline --no-rng-pin --no-launch-timing"
for MODE in torch_autograd torch_pair; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$MODE -- python $R/bench.py --only-mode $MODE $COMMON > $O/kt_$MODE.json 2> $O/kt_$MODE.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$MODE -- python $R/bench.py --only-mode $MODE $COMMON > $O/fetch_$MODE.json 2> $O/fetch_$MODE.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$MODE -- python $R/bench.py --only-mode $MODE $COMMON > $O/write_$MODE.json 2> $O/write_$MODE.err
done
cd $R
python tools/collect_torch_modes.py --summarise-only
