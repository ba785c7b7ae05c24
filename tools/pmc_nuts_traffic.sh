#!/bin/bash
# Round 5: MEASURED memory traffic of the free-running NUTS tick at C3 -- the tick kernel
# (k_nuts_async_tick3<64,1,W,DEFER>: leaf + deferred transition ends) and the funnel callable, full-ensemble launches
# only (32 768 rows), FETCH_SIZE and WRITE_SIZE in SEPARATE counters-only passes (FETCH_SIZE doubled per the guide's
# gfx950 correction), durations from a kernel-trace pass of the same command.
# JSON -> stdout; copy into profiles/r05/nuts_c3_pmc.json (bench.py reads profiles/nuts_traffic_latest.json).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
W="python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off"
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- $W > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $W > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $W > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $W > $OUT/tcc.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, collections, sys
out = sys.argv[1]
N, D = 32768, 256
KEYS = {"async_tick3": "tick", "k_neal_funnel": "callable"}
def name_of(k):
    for key, n in KEYS.items():
        if key in k:
            return n
    return None
def full(r):
    g = int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0)
    return g >= N * 64
res = {}
dur = collections.defaultdict(list)
for f in glob.glob(out + '/kt/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        n = name_of(r['Kernel_Name'])
        if n and full(r):
            dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("fetch", "write", "tcc"):
    for f in glob.glob(out + f'/{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            n = name_of(r['Kernel_Name'])
            if n and full(r):
                cnt[n][r['Counter_Name']].append(float(r['Counter_Value']))
tot_bytes, tot_us = 0.0, 0.0
for n in ("tick", "callable"):
    c = {k: sum(v) / len(v) for k, v in cnt[n].items()}
    us = sum(dur[n]) / len(dur[n]) if dur[n] else None
    hbm = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
    res[n] = {"full_ensemble_launches_timed": len(dur[n]), "launch_us_avg": us,
              "launches_counted": {k: len(v) for k, v in cnt[n].items()},
              "fetch_size_KB_raw": c.get("FETCH_SIZE"), "write_size_KB_raw": c.get("WRITE_SIZE"),
              "measured_bytes_per_launch": hbm, "measured_bytes_per_row": hbm / N,
              "measured_bytes_per_element": hbm / (N * D),
              "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if c.get("TCC_HIT_sum") else None,
              "GBps_of_measured_bytes": hbm / (us * 1e-6) / 1e9 if us else None}
    tot_bytes += hbm
    tot_us += us or 0.0
alg = 52.0 * N * D
print(json.dumps({
    "what": "C3 free-running NUTS, busy phase (every chain live): one tick = tick kernel + funnel callable; counters at the L2 "
            "memory-side interface (Infinity-Cache hits are counted, MI355X_MICROARCH.md HBM section)",
    "chains": N, "dim": D, "kernels": res,
    "tick_plus_callable": {"measured_bytes": tot_bytes, "measured_bytes_per_element": tot_bytes / (N * D),
                           "algorithmic_bytes_SURVEY_52B": alg, "traffic_over_algorithmic": tot_bytes / alg,
                           "us": tot_us, "GBps_measured": tot_bytes / (tot_us * 1e-6) / 1e9 if tot_us else None,
                           "frac_of_8TBps_measured": tot_bytes / (tot_us * 1e-6) / 8e12 if tot_us else None,
                           "frac_of_8TBps_at_52B": alg / (tot_us * 1e-6) / 8e12 if tot_us else None},
    "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B for wide coalesced reads); WRITE_SIZE as reported; "
                  "separate --pmc passes; full-ensemble launches only (grid >= 32768 waves)",
    "source": "tools/pmc_nuts_traffic.sh"}, indent=1))
PY
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/tcc
