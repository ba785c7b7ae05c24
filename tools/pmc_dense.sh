#!/bin/bash
# PMC passes (counters only, no tracing) over the dense benchmark; prints per-kernel averages.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_dense
rm -rf $OUT; mkdir -p $OUT
# BJX_DENSE_TN8=0 selects the 4-wave tile kernel
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python $R/tools/bench_dense.py --steps 2 > $OUT/g$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_dense/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        import re
        m = re.search(r'(k_dense_gemm\w*<[^>]*>)', k)
        if m and '<1, 2>' in m.group(1):  # the leapfrog launch: EPI_DRIFT, two kicks
            acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
