#!/bin/bash
# Free-running NUTS at C3: kernel-trace timeline + FETCH_SIZE / WRITE_SIZE (separate passes) of the
# tick kernels over the busy phase.  Outputs in gpurun_out/pmc_nuts2/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_nuts2
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python $R/tools/bench_nuts.py --free-running --steps 20 --no-tick-timing --run-graph off > $OUT/$c.log 2>&1
done
cd $R
python tools/nuts_trace_phases.py $(ls $OUT/kt/*/*kernel_trace.csv | head -1) 100 | sed -n 1,3p
python tools/nuts_trace_phases.py $(ls $OUT/kt/*/*kernel_trace.csv | head -1) 100 | awk '$2+0 > 20'
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f'gpurun_out/pmc_nuts2/{cn}/*/*counter_collection.csv'):
        seen = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            name = None
            for key in ("async_tick2", "async_leaf", "async_boundary", "async_fused", "k_neal_funnel"):
                if key in k:
                    name = key + (k.split("async_tick2")[1][:14] if key == "async_tick2" else "")
            if name is None or r['Counter_Name'] != cn:
                continue
            seen[name] += 1
            if int(r['Grid_Size']) >= 32768 * 64 // 1 * 1 and seen[name] > 50:   # full-ensemble launches only
                acc[name][cn].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    f = sum(d['FETCH_SIZE']) / max(len(d['FETCH_SIZE']), 1)
    w = sum(d['WRITE_SIZE']) / max(len(d['WRITE_SIZE']), 1)
    print(f"{k:40s} launches {len(d['FETCH_SIZE']):5d}  FETCH(x2 corrected) {2*f/1024:8.1f} MB  WRITE {w/1024:8.1f} MB  total {(2*f+w)/1024:8.1f} MB  = {(2*f+w)*1024/32768/1024:6.2f} KB per chain")
PY
rm -rf $OUT/kt/*/*kernel_trace.csv $OUT/FETCH_SIZE $OUT/WRITE_SIZE
