#!/bin/bash
# Round 4: the busy-phase leaf kernels of the free-running NUTS ticks (BJX_NUTS_LEAF3 = 68: one chain per wave,
# lean registers; 0: the v2 leaf; 16: four chains per wave; list in BJX_LEAF_VARIANTS), same box, same run (C3, 2 + 20 transitions, external funnel callable, plain launches):
# kernel-trace durations (all launches, and the full-ensemble ones) and, in a separate pass as the guide
# prescribes, SQ instruction counters.  JSON -> stdout; copy into profiles/r04/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts_leaf
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in ${BJX_LEAF_VARIANTS:-132 68 0 16}; do
  export BJX_NUTS_LEAF3=$v
  export BJX_NUTS_FUSED_ROWS=8192
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt$v -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/kt$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/c$v -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/c$v.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/d$v -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/d$v.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json, collections
import os
KEYS = ("async_tick3", "async_tick2<1, 0", "async_tick2<1, 2", "async_end_list", "k_neal_funnel")
VARIANTS = [int(x) for x in os.environ.get("BJX_LEAF_VARIANTS", "132 68 0 16").split()]
def name_of(k):
    for key in KEYS:
        if key in k:
            return key
    return None
res = {}
for v in VARIANTS:
    rows_per_wg = 4 if v == 16 else 1
    out = {}
    dur = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/pmc_nuts_leaf/kt{v}/*/*kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            n = name_of(r['Kernel_Name'])
            if n is None:
                continue
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            full = int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', 0)) >= (32768 * 64 // rows_per_wg if n == "async_tick3" else 32768 * 64)
            dur[n].append((d, full))
    for n, ds in dur.items():
        fl = [d for d, f in ds if f]
        out[n] = {"calls": len(ds), "avg_us": sum(d for d, _ in ds) / len(ds), "total_ms": sum(d for d, _ in ds) / 1e3,
                  "full_ensemble_calls": len(fl), "full_ensemble_avg_us": (sum(fl) / len(fl)) if fl else None,
                  "full_ensemble_first20_avg_us": (sum(fl[:20]) / len(fl[:20])) if fl else None}
    cnt = collections.defaultdict(collections.Counter)
    fullc = collections.defaultdict(lambda: collections.defaultdict(list))
    for tag in ("c", "d"):
        for f in glob.glob(f'gpurun_out/pmc_nuts_leaf/{tag}{v}/*/*counter_collection.csv'):
            for r in csv.DictReader(open(f)):
                n = name_of(r['Kernel_Name'])
                if n is None:
                    continue
                cnt[n][r['Counter_Name']] += float(r['Counter_Value'])
                g = int(r['Grid_Size'])
                if g >= (32768 * 64 // rows_per_wg if n == "async_tick3" else 32768 * 64):
                    fullc[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for n in cnt:
        out.setdefault(n, {})["counters_all_launches"] = dict(cnt[n])
        if fullc[n]:
            out[n]["counters_per_full_ensemble_launch"] = {k: sum(x) / len(x) for k, x in fullc[n].items()}
            if "SQ_INSTS_VALU" in fullc[n]:
                x = fullc[n]["SQ_INSTS_VALU"]
                out[n]["VALU_per_row"] = sum(x) / len(x) / 32768
    res[f"leaf3={v}"] = out
    val = None
    try:
        val = [json.loads(ln)["value"] for ln in open(f'gpurun_out/pmc_nuts_leaf/kt{v}.log') if ln.startswith("{")][-1]
    except Exception:
        pass
    res[f"leaf3={v}"]["run_value_under_kernel_trace"] = val
print(json.dumps(res))
PY
rm -rf $OUT/kt? $OUT/c? $OUT/d?
