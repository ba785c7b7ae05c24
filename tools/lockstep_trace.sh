set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/lockstep_trace; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/tools/bench_nuts.py --use-graph --steps 6 --warmup 4 > $O/kt.log 2>&1
cd $R
F=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python tools/trace_window.py $F k_nuts_post_res 3000 > $O/window.txt 2>&1
python - "$F" <<'PY' > $O/hist.txt 2>&1
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# start-to-start of consecutive post_res launches, bucketed by grid size
prev = None; per = collections.defaultdict(list)
for r in rows:
    if "k_nuts_post_res" not in r["Kernel_Name"]: continue
    s = int(r["Start_Timestamp"]); g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    if prev is not None: per[g].append((s - prev) / 1e3)
    prev = s
for g in sorted(per):
    v = sorted(per[g]); print(g, len(v), "median %.2f p90 %.2f mean %.2f" % (v[len(v)//2], v[len(v)*9//10], sum(v)/len(v)))
dur = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = "post" if "post_res" in n else ("funnel" if "funnel" in n else None)
    if key: dur[(key, int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(dur):
    v = sorted(dur[k]); print(k, len(v), "dur median %.2f mean %.2f" % (v[len(v)//2], sum(v)/len(v)))
PY
rm -rf $O/kt
cat $O/window.txt | head -12; cat $O/hist.txt | head -40
