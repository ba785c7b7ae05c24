#!/bin/bash
# round 3, GPU call 4: general integrators for nuts / mhmc / dhmc / dmhmc, ChEES NT A/B, suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c4
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_integrators_samplers_gpu.py tests/test_integrators.py -q -m gpu -x) > $O/integ_tests.log 2>&1
tail -15 $O/integ_tests.log
(time timeout 1500 python -m pytest tests/ -q -m gpu --deselect tests/test_integrators_samplers_gpu.py) > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
for NT in 0 1; do
  BJX_CHEES_NT=$NT python tools/bench_chees.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('chees NT=$NT', round(d['value']/1e6,1), round(d['pooled_statistics_ms_per_step'],3), {k:(round(v['avg_us'],1), round(v['GBps'])) for k,v in d['kernels'].items() if 'leapfrog' not in k})"
done
