#!/bin/bash
# round 3, GPU call 5: MEADS fold statistics as HIP, ChEES criterion rows kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c5
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_ghmc_gpu.py tests/test_chees_gpu.py -q -m gpu -x) > $O/tests.log 2>&1
tail -15 $O/tests.log
python tools/bench_ghmc.py > $O/ghmc.json 2> $O/ghmc.err; tail -3 $O/ghmc.err
python -c "import json; j=json.load(open('$O/ghmc.json')); print('ghmc', round(j['value']/1e6,1), j['ms_per_transition'], j['frac_of_8TBps']); print('meads', j['meads'])"
for NT in -1 0 1; do
  if [ $NT = -1 ]; then unset BJX_CHEES_NT; else export BJX_CHEES_NT=$NT; fi
  python tools/bench_chees.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('chees NT=$NT', round(d['value']/1e6,1), round(d['pooled_statistics_ms_per_step'],3), {k:(round(v['avg_us'],1), round(v['GBps'])) for k,v in d['kernels'].items() if 'leapfrog' not in k})"
done
unset BJX_CHEES_NT
BJX_CHEES_CRIT_ROWS=0 python tools/bench_chees.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('chees old criterion', round(d['value']/1e6,1), round(d['pooled_statistics_ms_per_step'],3), {k:(round(v['avg_us'],1), round(v['GBps'])) for k,v in d['kernels'].items() if 'leapfrog' not in k})"
