#!/bin/bash
# Round 6 closing evidence, one box, one call: full GPU suite, smoke(), the default bench (compact line + full record),
# rocprofv3 kernel stats of the C5 dense launch and of a C3 run.  Outputs: gpurun_out/final6/; copy with
# tools/collect_final6.py into profiles/r06/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final6
rm -rf $O; mkdir -p $O
cd $R
(time timeout 1500 python -m pytest tests/ -q -m gpu) > $O/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
BJX_BENCH_FULL=$O/bench_full.json python bench.py > $O/bench_compact.json 2> $O/bench.err
python tools/time_momentum.py > $O/momentum.json 2>> $O/bench.err
bash tools/rocprof_kernel.sh final6/kt_dense k_dense_gemm_tn8 python bench.py --config c5 --no-cpu-baseline --no-parity --steps 20 > $O/dense_kernel_stats.txt 2>&1
bash tools/rocprof_kernel.sh final6/kt_nuts "k_nuts_async_tick3|k_neal_funnel|k_nuts_spec" python bench.py --config c3 --no-cpu-baseline --no-parity --no-c3-t400 > $O/nuts_kernel_stats.txt 2>&1
tail -3 $O/gpu_tests.log; tail -c 400 $O/bench_compact.json; cat $O/momentum.json; cat $O/dense_kernel_stats.txt | cut -c1-160
