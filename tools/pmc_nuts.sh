#!/bin/bash
# PMC passes (counters only) over the free-running NUTS benchmark; per-kernel averages over the
# first 250 ticks of the timed run (all chains busy) for the two tick kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM_NORM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/g$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_nuts/g*/*/*counter_collection.csv'):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'async_leaf' in k or 'async_boundary' in k:
            name = 'leaf' if 'async_leaf' in k else 'boundary'
            seen[(name, r['Counter_Name'])] += 1
            n = seen[(name, r['Counter_Name'])]
            # dispatches 1..~300 belong to the priming run (2 transitions); take 400..650 of the timed run
            if 400 <= n < 650:
                acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-24s %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
