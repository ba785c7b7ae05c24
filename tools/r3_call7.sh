#!/bin/bash
# round 3, GPU call 7: dense GEMM staggered start sweep; MEADS bench; torch-modes PMC
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c7
rm -rf $O; mkdir -p $O
cd $R
for MODE in 0 1; do
for US in 0 3 5 7 9 12 16; do
  BJX_DENSE_STAGGER_MODE=$MODE BJX_DENSE_STAGGER_US=$US python tools/bench_dense.py 2>/dev/null | python -c "
import json,sys
j=json.load(sys.stdin); r=j['roofline']
print('stagger mode $MODE us $US:', round(j['value']/1e6,1), 'M/s; launch', round(r['avg_launch_us'],1), 'us', round(r['achieved'],1), 'TF frac', round(r['frac'],3))"
  if [ $MODE = 1 ] && [ $US = 0 ]; then :; fi
done; done 2>&1 | tee $O/dense_stagger.txt
(timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_frows_dense_gpu.py "tests/test_full_shape_gpu.py::test_c5_full_shape_dense_subset_bit_exact" -q -m gpu -x 2>&1 | tail -3) | tee $O/dense_tests.txt
BJX_DENSE_STAGGER_US=7 timeout 600 python -m pytest tests/test_dense_gpu.py "tests/test_full_shape_gpu.py::test_c5_full_shape_dense_subset_bit_exact" -q -m gpu -x 2>&1 | tail -3 | tee -a $O/dense_tests.txt
bash tools/pmc_torch_modes.sh > $O/torch_modes.txt 2>&1; tail -5 $O/torch_modes.txt
