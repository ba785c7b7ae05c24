#!/bin/bash
# round 3, GPU call 30: wave-uniform threefry blocks on the scalar unit (NUTS v2 kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c30
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_free_gpu.py tests/test_nuts_free_adapt_gpu.py tests/test_full_shape_gpu.py tests/test_device_target.py -q -m gpu -x) > $O/tests.log 2>&1
tail -4 $O/tests.log
for rep in 1 2; do
  for T in 20 100 400; do
    timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/c_T${T}_$rep.json 2>> $O/nuts.err
    python -c "import json; j=json.load(open('$O/c_T${T}_$rep.json')); print('contract T=$T rep $rep', round(j['value']/1e6,1), round(j.get('tick_period_avg_us'),2))"
    timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/f_T${T}_$rep.json 2>> $O/nuts.err
    python -c "import json; j=json.load(open('$O/f_T${T}_$rep.json')); print('resident T=$T rep $rep', round(j['value']/1e6,1), round(j.get('tick_period_avg_us'),2))"
  done
done
timeout 300 python tools/bench_nuts.py --steps 8 --warmup 3 --fuse-target > $O/step_f.json 2>> $O/nuts.err
python -c "import json; j=json.load(open('$O/step_f.json')); print('resident step', round(j['value']/1e6,1), round(j['ms_per_transition'],2))"
