#!/bin/bash
# round 3, GPU call 9: four-wide normal draws (ILP): momentum kernel timing, GHMC, NUTS A/B, parity
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c9
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_hmc_gpu.py tests/test_ghmc_gpu.py tests/test_nuts_free_gpu.py tests/test_oracle_prng.py tests/test_mhmc_gpu.py "tests/test_full_shape_gpu.py::test_c2_full_shape_all_chains_vs_c_port" -q -m gpu -x) > $O/tests.log 2>&1
tail -4 $O/tests.log
python tools/time_momentum.py 2>&1 | tail -1 | tee $O/momentum.txt
python tools/bench_ghmc.py 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('ghmc', round(j['value']/1e6,1), j['ms_per_transition'], j['frac_of_8TBps'], 'meads ms/step', j['meads']['ms_per_step'])" | tee $O/ghmc.txt
V=blackjax_amd/csrc/build/variants/libbjxhip_n4off.so
for T in 20 100; do
  bash tools/ab_variant.sh $V python tools/bench_nuts.py --free-running --steps $T --no-tick-timing 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('=='): print(line)
    elif line.startswith('{'): j=json.loads(line); print('  T=$T', round(j['value']/1e6,1), 'M/s')"
done | tee $O/nuts_ab.txt
python bench.py --steps 20 --headline-only --no-cpu-baseline --no-rng-pin 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(j['value']/1e6,1), j['ms_per_step'])" | tee $O/c2.txt
