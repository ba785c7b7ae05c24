#!/bin/bash
# round 3, GPU call 16: register-resident multi-tick loop
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c16
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_nuts_free_gpu.py tests/test_nuts_free_adapt_gpu.py tests/test_full_shape_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -4 $O/tests.log
for W in 2 3; do
  for T in 20 100 400; do
    BJX_MULTI_WAVES=$W timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/nuts_T${T}_w$W.json 2>> $O/nuts.err
    python -c "import json; j=json.load(open('$O/nuts_T${T}_w$W.json')); print('waves $W T=$T', round(j['value']/1e6,1), j.get('ticks'), round(j.get('tick_period_avg_us'),2), round(j['frac_of_52B_roofline'],3))"
  done
done
timeout 300 python tools/bench_nuts.py --steps 8 --warmup 3 --fuse-target > $O/lockstep_fused.json 2>> $O/nuts.err
python -c "import json; j=json.load(open('$O/lockstep_fused.json')); print('step fused', round(j['value']/1e6,1), round(j['ms_per_transition'],2), round(j['frac_of_52B_roofline'],3))"
tail -3 $O/nuts.err
