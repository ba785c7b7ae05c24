#!/usr/bin/env python
"""Turn the scratch outputs of tools/profile_round.sh (gpurun_out/round_prof/) into the tracked
evidence under profiles/: the rocprofv3 kernel-stats summary, the PMC traffic of the dominant kernel
(FETCH_SIZE and WRITE_SIZE from SEPARATE --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950) and the default bench line.

usage: python tools/collect_profiles.py <tag> [round]      e.g. v1 r02 -> profiles/r02/bench_c2_*_v1.*
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round_prof")
KERNEL = "k_leapfrog_diag<4,2>"  # replaced by the bench line's roofline.kernel in main()


def is_kernel(name):
    # "k_leapfrog_diag_flat<2>" also matches its nontemporal / plain instantiations <2, true> / <2, false>
    key = KERNEL.replace(" ", "")
    key = key[:-1] if key.endswith(">") else key
    return key in name.replace(" ", "")


def counter_avg(sub, counter):
    """-> (mean counter value per launch of KERNEL, number of launches, grid size of those launches)"""
    f = max(glob.glob(os.path.join(SRC, sub, "*", "*counter_collection.csv")), key=os.path.getmtime)  # newest run
    vals, grids = [], set()
    for r in csv.DictReader(open(f)):
        if is_kernel(r["Kernel_Name"]) and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            grids.add(int(r["Grid_Size"]))
    return sum(vals) / len(vals), len(vals), sorted(grids)


def _build():
    import subprocess
    try:
        rev = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %cs %s"], capture_output=True, text=True).stdout.strip()
        return rev[:100] or "unknown"
    except Exception:
        return "unknown"


BUILD = _build()  # the commit whose library the PMC passes profiled (collect right after the GPU call, before further commits)


def main():
    global KERNEL
    tag = sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
    out_dir = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(out_dir, exist_ok=True)
    bench = json.loads(open(os.path.join(SRC, "bench_default.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(out_dir, f"bench_c2_default_{tag}.json"), "w"), indent=1)
    roof = bench["roofline"]
    KERNEL = roof["kernel"]
    cfg = bench["config"]
    N = cfg["chains_per_gpu"]
    stats = {}
    for mode in ("stream", "cache"):
        found = glob.glob(os.path.join(SRC, f"kt_{mode}", "*", "*kernel_stats.csv"))
        if not found:
            continue
        f = max(found, key=os.path.getmtime)  # newest run
        shutil.copy(f, os.path.join(out_dir, f"bench_c2_kernel_stats_{mode}_{tag}.csv"))
        for r in csv.DictReader(open(f)):
            if is_kernel(r["Name"]):
                stats[mode] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])}
    # PMC passes were taken in the streaming mode: all chains per launch
    fetch_kb, n_f, grid_f = counter_avg("fetch", "FETCH_SIZE")
    write_kb, n_w, grid_w = counter_avg("write", "WRITE_SIZE")
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    alg = 20.0 * cfg["dim"] * N
    pmc = {
        "chains": N, "dim": cfg["dim"], "chains_per_launch": N,
        "kernel": KERNEL, "fetch_size_KB_raw": fetch_kb, "write_size_KB_raw": write_kb,
        "launches": {"fetch_pass": n_f, "write_pass": n_w, "grid_sizes": sorted(set(grid_f + grid_w))},
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for wide coalesced reads, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; separate --pmc passes",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
        "note": "streaming mode (all chains per launch, 1.34 GB per launch >> the 256 MiB Infinity Cache); "
                "FETCH_SIZE/WRITE_SIZE are taken at the L2 memory-side interface",
        "rocprofv3_kernel_trace": stats,
        "bench_hip_events": {"stream_avg_us": roof.get("avg_launch_us"),
                             "cache_avg_us": (roof.get("cache_assisted") or {}).get("avg_launch_us")},
        "source": f"profiles/{rnd}/bench_c2_pmc_{tag}.json",
        "measured_on_build": BUILD,
    }
    json.dump(pmc, open(os.path.join(out_dir, f"bench_c2_pmc_{tag}.json"), "w"), indent=1)
    json.dump(pmc, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
    print("rocprofv3 kernel-trace:", stats)
    print("bench HIP events:", pmc["bench_hip_events"], "value M/s:", bench["value"] / 1e6)
    print("traffic ratio:", pmc["ratio"])


if __name__ == "__main__":
    main()
