#!/bin/bash
# Collects the judged evidence for one round on the GPU box: the default bench JSON, then -- with the
# scheduling the bench's autotune chose, passed explicitly so every pass profiles the same launches --
# kernel-trace stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, as
# MI355X_MICROARCH.md prescribes).  Outputs land in gpurun_out/round_prof/;
# `python tools/collect_profiles.py <tag>` then copies the summaries into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_prof
rm -rf $OUT
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
CB=$(python -c "import json,sys; print(json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])['config']['chain_block'])")
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-plain-mode --chain-block $CB > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-launch-timing --no-plain-mode --chain-block $CB > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-launch-timing --no-plain-mode --chain-block $CB > $OUT/write.log 2>&1
cd $R
echo "chain_block chosen: $CB"
tail -c 700 $OUT/bench_default.json
