#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c25
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_hmc_traj_gpu.py tests/test_hmc_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -6 $O/tests.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('headline', round(j['value']/1e6,1)); print('resident', j['engine_resident_target_mode'])"
tail -2 $O/bench.err
