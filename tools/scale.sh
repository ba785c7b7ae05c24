#!/bin/bash
# The 1 -> 8 GPU scaling table north_star asks for, in one command on an 8-GPU MI355X node:
#   tools/scale.sh [c2|c3|c4|c5|both|all] [max_gpus]
# Runs `bench.py --gpus N` (which spawns one rank per GPU over RCCL and refuses when fewer GPUs are
# visible) for N = 1, 2, 4, 8, keeps every JSON line under gpurun_out/scale/, and prints absolute
# whole-node throughput, the speed-up over N = 1, and the fraction of the HBM roofline per GPU
# (c2: 28 B per chain-leapfrog element; c4: 32 B -- per-chain inverse mass matrix; c3: 52 B; c5: the fp32
# MFMA roofline), followed by ONE JSON line per (config, N) -- {"config", "n_gpus", "ranks", "value",
# "ms_per_step", "frac_per_gpu", "x_vs_1"} -- also written to gpurun_out/scale/scale_lines.jsonl, so a SCALE
# record can be parsed without scraping the table.
# The driver's own SCALE_rNN.json run is the judged one; this script is the same thing for a human.
set -u
CFG=${1:-both}
MAXG=${2:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/scale
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
run_cfg() {
  local cfg=$1 extra=$2
  for n in 1 2 4 8; do
    [ $n -le $MAXG ] || continue
    python bench.py --config $cfg --gpus $n $extra --headline-only --no-cpu-baseline --no-rng-pin \
      > $O/${cfg}_n$n.json 2> $O/${cfg}_n$n.err || echo "bench.py --config $cfg --gpus $n failed: $(tail -1 $O/${cfg}_n$n.err)"
  done
}
case $CFG in
  c2|c3|c4|c5) run_cfg $CFG "" ;;
  all) for c in c2 c3 c4 c5; do run_cfg $c ""; done ;;
  *) run_cfg c2 ""; run_cfg c4 "" ;;
esac
python - "$O" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = {}
lines = []
for f in sorted(glob.glob(os.path.join(out, "c?_n*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    cfg = os.path.basename(f)[:2]
    rows.setdefault(cfg, []).append(j)
for cfg, js in rows.items():
    js.sort(key=lambda j: j["ranks"])
    base = js[0]["value"] / max(js[0]["n_gpus"], 1)
    def frac(j):
        if cfg == "c2":
            return j.get("end_to_end_frac_of_28B_roofline")
        if cfg == "c4":
            return j.get("end_to_end_frac_of_32B_roofline")
        if cfg == "c3":
            return (j.get("roofline") or {}).get("frac")
        return (j.get("end_to_end_TFLOPs") or float("nan")) / 157.3  # c5: fraction of the fp32 MFMA peak
    print(f"{cfg}: {js[0]['config']['workload']}")
    print(f"  {'GPUs':>4} {'ranks':>5} {'M chain-leapfrog/s':>20} {'x vs 1 GPU':>11} {'eff':>6} {'frac of HBM roofline / GPU':>27} {'ms/step':>9}")
    for j in js:
        n = j["n_gpus"]
        print(f"  {n:>4} {j['ranks']:>5} {j['value'] / 1e6:>20.1f} {j['value'] / base:>11.2f} {j['value'] / base / n:>6.2f} "
              f"{(frac(j) if frac(j) is not None else float('nan')):>27.3f} {j['ms_per_step']:>9.2f}")
    for j in js:
        lines.append({"config": cfg, "n_gpus": j["n_gpus"], "ranks": j["ranks"], "value": j["value"],
                      "ms_per_step": j["ms_per_step"], "frac_per_gpu": frac(j), "x_vs_1": j["value"] / base})
with open(os.path.join(out, "scale_lines.jsonl"), "w") as f:
    for ln in lines:
        print(json.dumps(ln))
        f.write(json.dumps(ln) + "\n")
PY
