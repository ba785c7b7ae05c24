#!/bin/bash
# Round-5 closing evidence on ONE box (the reduced form of tools/final_round.sh that fits the GPU budget left after the
# speculative-tail work): full GPU suite, smoke(), the default bench line, the step-driver and speculative-tail A/Bs at
# C3, and the rocprofv3 kernel stats of a C3 free-running run (tick / integrate / book / callable durations).
# Outputs: gpurun_out/r5b/ ; copied into profiles/r05/ by hand (names in profiles/README.md).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
rm -rf $O; mkdir -p $O
cd $R
(time timeout 1200 python -m pytest tests/ -q -m gpu -p no:cacheprovider) > $O/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/bench_nuts_spec.py --T 20 100 400 --reps 1 --spec 0 128 > $O/nuts_c3_spec_ab.json 2> $O/spec_ab.err
python tools/step_vs_run1.py > $O/nuts_c3_step_drivers.json 2> $O/step.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_nuts -- python $R/tools/bench_nuts.py --free-running --steps 100 --no-tick-timing > $O/kt_nuts.log 2>&1
cd $R
find $O/kt_nuts -name '*kernel_trace.csv' -delete
tail -3 $O/gpu_tests.log; tail -1 $O/smoke.log; tail -c 600 $O/bench_default.json
