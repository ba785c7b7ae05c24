#!/usr/bin/env python
"""gpurun_out/round_summary/ (tools/round_summary.sh) -> profiles/<round>/summary_final_<round>.json +
kernel-stats CSVs.  usage: python tools/collect_summary.py r02"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round_summary")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
out_dir = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out_dir, exist_ok=True)


def last_json(name):
    p = os.path.join(SRC, name)
    if not os.path.exists(p):
        return None
    lines = [ln for ln in open(p).read().strip().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


log = open(os.path.join(SRC, "gpu_tests.log")).read()
m = re.search(r"(\d+) passed", log)
f = re.search(r"(\d+) failed", log)
t = re.search(r"real\s+(\S+)", log)
summary = {
    "note": "one MI355X box, one gpurun call (tools/round_summary.sh) on the final build of the round",
    "gpu_tests": {"passed": int(m.group(1)) if m else None, "failed": int(f.group(1)) if f else 0,
                  "wall": t.group(1) if t else None},
    "smoke": open(os.path.join(SRC, "smoke.log")).read().strip().splitlines()[-1],
}
for key, name in [("bench_py_C2", "bench_c2.json"), ("bench_py_C2_two_gloo_ranks_on_one_gpu", "bench_c2_2ranks_one_gpu.json"),
                  ("bench_py_C4_shard", "bench_c4.json"), ("nuts_C3_free_running_T20", "nuts_c3_T20.json"),
                  ("nuts_C3_free_running_T100", "nuts_c3_T100.json"), ("nuts_C3_free_running_T400", "nuts_c3_T400.json"),
                  ("nuts_C3_lockstep_step", "nuts_c3_lockstep.json"),
                  ("nuts_C3_engine_resident_target_free_running_T20", "nuts_c3_fused_T20.json"),
                  ("nuts_C3_engine_resident_target_free_running_T100", "nuts_c3_fused_T100.json"),
                  ("nuts_C3_engine_resident_target_free_running_T400", "nuts_c3_fused_T400.json"),
                  ("nuts_C3_engine_resident_target_step", "nuts_c3_fused_step.json"),
                  ("dense_C5", "dense_c5.json"),
                  ("nuts_shared_dense_metric_gemm_vs_matvec", "nuts_dense_shared.json"),
                  ("chees_C2", "chees_c2.json"), ("ghmc_C2_shape_and_meads", "ghmc_c2.json"),
                  ("nuts_C3_warmup", "nuts_warmup_c3.json"),
                  ("hmc_small_batches", "hmc_small.json")]:
    j = last_json(name)
    if j is not None:
        j.pop("gpu_ms_of_each_step", None)
        summary[key] = j
json.dump(summary, open(os.path.join(out_dir, f"summary_final_{rnd}.json"), "w"), indent=1)
for tag, name in (("nuts", "nuts_c3"), ("nuts_fused", "nuts_c3_engine_resident_target"), ("dense", "dense_c5")):
    found = glob.glob(os.path.join(SRC, f"kt_{tag}", "*", "*kernel_stats.csv"))
    if found:  # gpurun MERGES into the local directory: take the newest run
        shutil.copy(max(found, key=os.path.getmtime), os.path.join(out_dir, f"{name}_kernel_stats_final.csv"))
tl = os.path.join(SRC, "nuts_c3_timeline.txt")
if os.path.exists(tl):
    shutil.copy(tl, os.path.join(out_dir, "nuts_c3_timeline_final.txt"))
print(json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in summary.items()}, indent=1))
