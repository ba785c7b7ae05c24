#!/bin/bash
# round 3, GPU call 2: nontemporal leapfrog A/B, NUTS tail anatomy (kernel trace of T = 400), low-latency tick variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c2
rm -rf $O; mkdir -p $O
cd $R
(time timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_bench_launch.py tests/test_torch_callable_gpu.py tests/test_ghmc_gpu.py tests/test_nuts_free_gpu.py tests/test_hmc_gpu.py -x -q -m gpu) > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log
for NT in 0 1; do
  BJX_LF_NT=$NT python bench.py --steps 20 --chain-block 0 --headline-only --no-cpu-baseline --no-rng-pin > $O/c2_stream_nt$NT.json 2> $O/c2_stream_nt$NT.err
  BJX_LF_NT=$NT python bench.py --steps 20 --chain-block 16384 --headline-only --no-cpu-baseline --no-rng-pin > $O/c2_block_nt$NT.json 2> $O/c2_block_nt$NT.err
done
for f in $O/c2_*.json; do echo $f; python -c "import json; j=json.load(open('$f')); print(round(j['value']/1e6,1), j['ms_per_step'], j['roofline'] and (round(j['roofline']['avg_launch_us'],1), round(j['roofline']['frac'],3)))"; done
for LL in 0 2048; do
  for T in 100 400; do
    BJX_NUTS_LOWLAT_ROWS=$LL timeout 300 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/nuts_T${T}_lowlat$LL.json 2>> $O/nuts.err
  done
done
for f in $O/nuts_*.json; do echo $f; python -c "import json; j=json.load(open('$f')); print(j['value']/1e6, j.get('ticks'), j.get('tick_period_avg_us'))"; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt_nuts400 -- python $R/tools/bench_nuts.py --free-running --steps 400 --no-tick-timing > $O/kt_nuts400.log 2>&1
cd $R
F=$(ls $O/kt_nuts400/*/*kernel_trace.csv | head -1)
python tools/nuts_trace_phases.py $F 5000 > $O/nuts_T400_timeline.txt 2>&1
python tools/nuts_trace_tail.py $F 4000 > $O/nuts_T400_tail.txt 2>&1
rm -rf $O/kt_nuts400
cat $O/nuts_T400_tail.txt; tail -25 $O/nuts_T400_timeline.txt
