#!/bin/bash
# SQ counters of the free-running NUTS tick kernels at C3 (busy phase: full-ensemble launches only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts3
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/g$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_nuts3/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        import re
        m = re.search(r'(k_nuts_async_tick2<[^>]*>|k_nuts_async_end_list<[^>]*>|k_neal_funnel\w*(<[^>]*>)?)', k)
        if not m:
            continue
        name = m.group(1)
        if 'end_list' not in name and int(r['Grid_Size']) < 32768 * 64:
            continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print('   %-24s %16.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
rm -rf $OUT/g*/
