#!/bin/bash
# round 3, GPU call 8: NUTS shared dense metric on the GEMM (parity + bench), torch-modes PMC again
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c8
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_gpu.py tests/test_integrators_samplers_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -12 $O/tests.log
timeout 900 python tools/bench_nuts_dense.py > $O/nuts_dense.json 2> $O/nuts_dense.err; tail -3 $O/nuts_dense.err; cat $O/nuts_dense.json
bash tools/pmc_torch_modes.sh > $O/torch_modes.txt 2>&1; tail -3 $O/torch_modes.txt
