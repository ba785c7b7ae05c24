#!/bin/bash
# Dynamic instruction counts of the CONTRACT-path tick kernels (external callable) over a T = 20 run at C3, per kernel:
# is the busy-phase leaf kernel bound by VALU issue?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts_insts_contract
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/c -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/kt.log 2>&1
cd $R
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.Counter())
full = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_nuts_insts_contract/c/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = None
        for key in ("async_tick2<1, 0", "async_tick2<1, 2", "async_end_list", "k_neal_funnel"):
            if key in k:
                name = key
        if name is None:
            continue
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
        acc[name]['launches_' + r['Counter_Name']] += 1
        if int(r['Grid_Size']) >= 32768 * 64:
            full[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in acc.items():
    out[k] = {"launches": int(c['launches_SQ_INSTS_VALU']), "VALU": c['SQ_INSTS_VALU'], "SALU": c['SQ_INSTS_SALU'], "waves": c['SQ_WAVES']}
    if full[k]['SQ_INSTS_VALU']:
        v = full[k]['SQ_INSTS_VALU']
        out[k]["full_ensemble_launches"] = {"n": len(v), "VALU_per_launch_mean": sum(v) / len(v), "VALU_per_row": sum(v) / len(v) / 32768}
dur = {}
for f in glob.glob('gpurun_out/pmc_nuts_insts_contract/kt/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        for key in ("async_tick2<1, 0", "async_tick2<1, 2", "async_end_list", "k_neal_funnel"):
            if key in r['Name']:
                dur[key] = {"calls": int(r['Calls']), "total_ms": float(r['TotalDurationNs']) / 1e6, "avg_us": float(r['AverageNs']) / 1e3, "max_us": float(r['MaxNs']) / 1e3}
print(json.dumps({"counters": out, "durations": dur}))
PY
rm -rf $OUT/c $OUT/kt
