#!/usr/bin/env python
"""gpurun_out/final6/ (tools/final_round6.sh) -> profiles/r06/summary_r06.json + bench_final_r06.json + kernel stats."""
import glob
import json
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final6")
DST = os.path.join(ROOT, "profiles", "r06")
os.makedirs(DST, exist_ok=True)
full = json.load(open(os.path.join(SRC, "bench_full.json")))
compact_text = open(os.path.join(SRC, "bench_compact.json")).read().strip().splitlines()[-1]
compact = json.loads(compact_text)
json.dump(full, open(os.path.join(DST, "bench_final_r06.json"), "w"), indent=1)
log = open(os.path.join(SRC, "gpu_tests.log")).read()
m = re.search(r"(\d+) passed(?:, (\d+) skipped)?.* in ([\d.]+)s", log)
rev = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %cs %s"], capture_output=True, text=True).stdout.strip()
summary = {
    "build": rev[:100],
    "gpu_suite": {"passed": int(m.group(1)) if m else None, "skipped": int(m.group(2) or 0) if m else None,
                  "seconds": float(m.group(3)) if m else None, "xpassed_or_xfailed": len(re.findall(r"xpassed|xfailed", log)),
                  "failed": len(re.findall(r"^FAILED", log, re.M))},
    "smoke": open(os.path.join(SRC, "smoke.log")).read().strip().splitlines()[:2],
    "bench_compact_line_bytes": len(compact_text),
    "bench_compact_line": compact,
    "momentum_draw": json.loads(open(os.path.join(SRC, "momentum.json")).read().strip().splitlines()[-1]),
    "dense_c5_rocprofv3": [ln for ln in open(os.path.join(SRC, "dense_kernel_stats.txt")).read().splitlines() if "k_dense_gemm" in ln][:4],
    "nuts_c3_rocprofv3": [ln for ln in open(os.path.join(SRC, "nuts_kernel_stats.txt")).read().splitlines() if "k_nuts" in ln or "funnel" in ln][:6],
}
json.dump(summary, open(os.path.join(DST, "summary_r06.json"), "w"), indent=1)
for sub, name in (("kt_dense", "dense_c5_kernel_stats_final.csv"), ("kt_nuts", "nuts_c3_kernel_stats_final.csv")):
    found = glob.glob(os.path.join(SRC, sub, "**", "*kernel_stats.csv"), recursive=True)
    if found:
        shutil.copy(found[0], os.path.join(DST, name))
print(json.dumps({k: summary[k] for k in ("build", "gpu_suite", "bench_compact_line_bytes", "momentum_draw")}, indent=1))
print("C2", compact["value"] / 1e6, compact["roofline"]["frac"], "C5", compact["c5_dense"]["roofline"]["frac"],
      "C3", compact["c3_nuts"]["value"] / 1e6, compact["c3_nuts"].get("lockstep_step"))
