#!/bin/bash
# One-box summary of a round: full GPU test suite, smoke(), every bench (C2 default, C2 at 2 gloo ranks
# on one GPU, C4 shard through bench.py --config c4, C3 NUTS free-running T = 20 / 100 / 400 + lockstep,
# C5 dense, NUTS with a shared dense metric on the GEMM, ChEES at C2, GHMC + MEADS, NUTS warm-up) and the
# rocprofv3 kernel stats of the NUTS and dense runs.
# Outputs land in gpurun_out/round_summary/; tools/collect_summary.py <round> turns them into
# profiles/<round>/summary_final_<round>.json and copies the kernel-stats CSVs.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round_summary
rm -rf $O; mkdir -p $O
cd $R
(time timeout 1500 python -m pytest tests/ -q -m gpu) > $O/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
BJX_BENCH_FULL=$O/bench_c2.json python bench.py > $O/bench_c2_compact.json 2> $O/bench_c2.err
BJX_BENCH_FULL=$O/bench_c2_2ranks_one_gpu.json BJX_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --headline-only --no-cpu-baseline > /dev/null 2> $O/bench_2r.err
BJX_BENCH_FULL=$O/bench_c4.json python bench.py --config c4 > /dev/null 2> $O/bench_c4.err   # default: the 1 000-step warm-up of configs[3]
for T in 20 100 400; do timeout 600 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing > $O/nuts_c3_T$T.json 2>> $O/nuts.err; done
timeout 600 python tools/bench_nuts.py --use-graph --steps 8 --warmup 4 > $O/nuts_c3_lockstep.json 2>> $O/nuts.err   # warm-up 4: the bucket graphs are recorded before the timed steps
# the same C3 runs with the funnel evaluated INSIDE the tick kernels (fuse_target=True: engine-resident target,
# outside the external-callable contract -- separately labelled figures)
for T in 20 100 400; do timeout 600 python tools/bench_nuts.py --free-running --steps $T --no-tick-timing --fuse-target > $O/nuts_c3_fused_T$T.json 2>> $O/nuts.err; done
timeout 600 python tools/bench_nuts.py --steps 8 --warmup 3 --fuse-target > $O/nuts_c3_fused_step.json 2>> $O/nuts.err
python tools/bench_dense.py > $O/dense_c5.json 2> $O/dense.err
timeout 900 python tools/bench_nuts_dense.py --steps 5 --warmup 3 > $O/nuts_dense_shared.json 2> $O/nuts_dense.err
python tools/bench_chees.py > $O/chees_c2.json 2> $O/chees.err
python tools/bench_ghmc.py > $O/ghmc_c2.json 2> $O/ghmc.err
python tools/bench_nuts_warmup.py > $O/nuts_warmup_c3.json 2> $O/nuts_warmup.err
python tools/bench_small.py > $O/hmc_small.json 2> $O/small.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_nuts -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing > $O/kt_nuts.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_nuts_fused -- python $R/tools/bench_nuts.py --free-running --steps 100 --no-tick-timing --fuse-target > $O/kt_nuts_fused.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dense -- python $R/tools/bench_dense.py > $O/kt_dense.log 2>&1
cd $R
python tools/nuts_trace_phases.py $(ls $O/kt_nuts/*/*kernel_trace.csv | head -1) 100 > $O/nuts_c3_timeline.txt 2>&1
rm -f $O/kt_nuts/*/*kernel_trace.csv $O/kt_nuts_fused/*/*kernel_trace.csv $O/kt_dense/*/*kernel_trace.csv
tail -3 $O/gpu_tests.log
