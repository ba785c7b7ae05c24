#!/bin/bash
# Dynamic instruction counts of the multi-tick NUTS kernel (fuse_target=True) over a T = 20 run at C3:
# VALU / SALU / memory instructions per leapfrog and the VALU's share of the busy cycles.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts_insts
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --fuse-target --run-graph off > $OUT/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json, collections
tot = collections.Counter()
for f in glob.glob('gpurun_out/pmc_nuts_insts/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_nuts_async_multi' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
j = None
for ln in open(glob.glob('gpurun_out/pmc_nuts_insts/SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_WAVES.log')[0]):
    if ln.startswith('{'):
        j = json.loads(ln)
leaves = j['value'] * 0 + sum([])  if False else None
print(json.dumps({"counters_summed_over_the_multi_tick_launches": dict(tot), "bench_line_value": j and j['value'],
                  "mean_chain_leapfrogs": j and j['utilisation_per_100_transitions'][0]['mean_chain_leapfrogs']}))
PY
