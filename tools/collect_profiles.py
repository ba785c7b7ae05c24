#!/usr/bin/env python
"""Turn the scratch outputs of tools/profile_round.sh (gpurun_out/round_prof/) into the tracked
evidence under profiles/: the rocprofv3 kernel-stats summary, the PMC traffic of the dominant kernel
(FETCH_SIZE and WRITE_SIZE from SEPARATE --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950) and the default bench line.

usage: python tools/collect_profiles.py <tag>      e.g. v4  -> profiles/r01/bench_c2_*_v4.*
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
    return KERNEL.replace(" ", "") + "(" in name.replace(" ", "")


def counter_avg(sub, counter):
    """-> (mean counter value per launch of KERNEL, number of launches, grid size of those launches)"""
    f = max(glob.glob(os.path.join(SRC, sub, "*", "*counter_collection.csv")), key=os.path.getmtime)  # newest run
    vals, grids = [], set()
    for r in csv.DictReader(open(f)):
        if is_kernel(r["Kernel_Name"]) and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            grids.add(int(r["Grid_Size"]))
    return sum(vals) / len(vals), len(vals), sorted(grids)


def main():
    global KERNEL
    tag = sys.argv[1]
    out_dir = os.path.join(ROOT, "profiles", "r01")
    os.makedirs(out_dir, exist_ok=True)
    bench = json.loads(open(os.path.join(SRC, "bench_default.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(out_dir, f"bench_c2_default_{tag}.json"), "w"), indent=1)
    if bench.get("roofline"):
        KERNEL = bench["roofline"]["kernel"]
    stats = max(glob.glob(os.path.join(SRC, "kt", "*", "*kernel_stats.csv")), key=os.path.getmtime)  # newest run
    shutil.copy(stats, os.path.join(out_dir, f"bench_c2_kernel_stats_{tag}.csv"))
    fetch_kb, n_f, grid_f = counter_avg("fetch", "FETCH_SIZE")
    write_kb, n_w, grid_w = counter_avg("write", "WRITE_SIZE")
    cfg = bench["config"]
    cpl = bench["roofline"]["chains_per_launch"] if bench.get("roofline") else cfg["chain_block"]
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    alg = 20.0 * cfg["dim"] * cpl
    pmc = {
        "chains": cfg["chains_per_gpu"], "dim": cfg["dim"], "chains_per_launch": cpl,
        "kernel": KERNEL, "fetch_size_KB_raw": fetch_kb, "write_size_KB_raw": write_kb,
        "launches": {"fetch_pass": n_f, "write_pass": n_w, "grid_sizes": sorted(set(grid_f + grid_w))},
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for wide coalesced reads, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; separate --pmc passes",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
        "note": "FETCH_SIZE/WRITE_SIZE are taken at the L2 memory-side interface and include Infinity-Cache "
                "hits (MI355X_MICROARCH.md HBM section): they bound wasted re-reads, they do not separate "
                "Infinity-Cache hits from HBM accesses",
        "source": f"profiles/r01/bench_c2_pmc_{tag}.json",
    }
    json.dump(pmc, open(os.path.join(out_dir, f"bench_c2_pmc_{tag}.json"), "w"), indent=1)
    json.dump(pmc, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
    # kernel-trace average of the dominant kernel, to set beside the bench's HIP-event figure
    for r in csv.DictReader(open(stats)):
        if is_kernel(r["Name"]):
            print("rocprofv3 avg us:", float(r["AverageNs"]) / 1e3, "calls", r["Calls"])
    print("bench avg us:", bench["roofline"]["avg_launch_us"], "value M/s:", bench["value"] / 1e6)
    print("traffic ratio:", pmc["ratio"])


if __name__ == "__main__":
    main()
