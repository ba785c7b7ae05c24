#!/bin/bash
# Collects the judged evidence for one round on the GPU box: the default bench JSON, then kernel-trace
# stats of the SAME kernel in the two scheduling modes the JSON reports (each profiled by its own
# command so the per-kernel average is not a blend of two launch sizes):
#   stream : --chain-block 0      all chains per launch, every launch streams from HBM  (roofline.frac)
#   cache  : --chain-block <auto> Infinity-Cache blocks                                  (cache_assisted_frac)
# and the two PMC passes of the streaming mode (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, as
# MI355X_MICROARCH.md prescribes).  Outputs land in gpurun_out/round_prof/;
# `python tools/collect_profiles.py <tag> r04` then copies the summaries into profiles/r04/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_prof
rm -rf $OUT
mkdir -p $OUT
cd $R
BJX_BENCH_FULL=$OUT/bench_default_full.json python bench.py > $OUT/bench_default_compact.json 2> $OUT/bench_default.err
cp $OUT/bench_default_full.json $OUT/bench_default.json   # the FULL record (the stdout line is the compact one)
CB=$(python -c "import json,sys; j=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print(j['roofline']['cache_assisted']['chains_per_launch'] if j['roofline'].get('cache_assisted') else j['config']['chain_block'])")
cd /tmp; export TMPDIR=/tmp
COMMON="--no-cpu-baseline --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_stream -- python $R/bench.py --steps 5 --warmup 2 $COMMON --chain-block 0 > $OUT/kt_stream.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cache -- python $R/bench.py --steps 5 --warmup 2 $COMMON --chain-block $CB > $OUT/kt_cache.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $R/bench.py --steps 2 --warmup 1 $COMMON --no-launch-timing --chain-block 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $R/bench.py --steps 2 --warmup 1 $COMMON --no-launch-timing --chain-block 0 > $OUT/write.log 2>&1
cd $R
echo "cache block: $CB"
tail -c 900 $OUT/bench_default.json
