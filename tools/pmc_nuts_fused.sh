#!/bin/bash
# Free-running NUTS at C3 with the target evaluated inside the multi-tick kernel (fuse_target=True):
# FETCH_SIZE / WRITE_SIZE (separate passes) of k_nuts_async_multi over a T = 20 run, per launch and per
# chain-tick, against the algorithmic bytes.  Outputs in gpurun_out/pmc_nuts_fused/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts_fused
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --fuse-target --run-graph off > $OUT/$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json
tot = {}
n = {}
first = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f'gpurun_out/pmc_nuts_fused/{cn}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_nuts_async_multi' in r['Kernel_Name'] and r['Counter_Name'] == cn:
                vals.append((int(r['Grid_Size']), float(r['Counter_Value'])))
    tot[cn] = sum(v for _, v in vals)
    n[cn] = len(vals)
    full = [v for g, v in vals if g >= 32768 * 64]
    first[cn] = full
j = json.load(open(glob.glob('gpurun_out/pmc_nuts_fused/FETCH_SIZE.log')[0])) if False else None
# counters are in KB at the L2 memory-side interface; FETCH doubled on gfx950 (MI355X_MICROARCH.md)
fetch_kb, write_kb = 2 * tot['FETCH_SIZE'], tot['WRITE_SIZE']
print(json.dumps({"multi_tick_launches": n['FETCH_SIZE'],
                  "memory_side_traffic_MB_whole_run": {"fetch_x2": fetch_kb / 1024, "write": write_kb / 1024,
                                                       "total": (fetch_kb + write_kb) / 1024},
                  "full_ensemble_launches (32 768 rows x 128 ticks)": {
                      "n": len(first['FETCH_SIZE']),
                      "MB_per_launch": [round((2 * a + b) / 1024, 1) for a, b in zip(first['FETCH_SIZE'], first['WRITE_SIZE'])]}}))
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
