#!/usr/bin/env python
"""Summarise tools/pmc_torch_modes.sh: measured HBM-side bytes per chain-leapfrog element of the two
PyTorch forms of the user log-density (engine kernels + every torch kernel), and the kernel-time split.

usage (GPU box, end of pmc_torch_modes.sh):  python tools/collect_torch_modes.py --summarise-only
       (here, after the call merged back):    python tools/collect_torch_modes.py r03
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "torch_modes")
STEPS, WARMUP, L = 2, 1, 50  # the flags of pmc_torch_modes.sh; one priming + one unthrottled-issue transition on top


def short(name):
    """kernel name without return type / anonymous namespace / argument list, at most 90 characters"""
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in n:  # cut at the first "(" outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[:90]


def newest(pattern):
    found = glob.glob(pattern)
    return max(found, key=os.path.getmtime) if found else None


def counter_total(sub, counter):
    f = newest(os.path.join(SRC, sub, "*", "*counter_collection.csv"))
    if f is None:
        return None, {}
    tot, per = 0.0, {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            v = float(r["Counter_Value"])
            tot += v
            k = short(r["Kernel_Name"])
            per[k] = per.get(k, 0.0) + v
    return tot, per


def summarise():
    out = {}
    for mode in ("torch_autograd", "torch_pair"):
        line = None
        try:
            line = json.loads(open(os.path.join(SRC, f"kt_{mode}.json")).read().strip().splitlines()[-1])
        except Exception:
            pass
        if line is None:
            continue
        N, D = line["config"]["chains_per_gpu"], line["config"]["dim"]
        transitions = STEPS + WARMUP + 2
        elems = float(N) * D * L * transitions
        f_kb, f_per = counter_total(f"fetch_{mode}", "FETCH_SIZE")
        w_kb, w_per = counter_total(f"write_{mode}", "WRITE_SIZE")
        entry = {"chains": N, "dim": D, "leapfrogs": L, "transitions_in_process": transitions,
                 "value_under_kernel_trace": line["value"], "ms_per_step_under_kernel_trace": line["ms_per_step"]}
        if f_kb is not None and w_kb is not None:
            total = (2.0 * f_kb + w_kb) * 1024.0
            entry.update({
                "fetch_size_KB_raw_total": f_kb, "write_size_KB_raw_total": w_kb,
                "bytes_per_element_total": total / elems,
                "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests of wide coalesced reads at 64 B, "
                              "MI355X_MICROARCH.md HBM section; torch's vectorized elementwise kernels read 16 B per "
                              "lane like the engine's); WRITE_SIZE as reported; separate --pmc passes; summed over "
                              "EVERY kernel of the process (momentum draw, finish, init included: < 1 % at L = 50); "
                              "chain blocks of 16 384: part of this traffic is served by the Infinity Cache "
                              "(the counters sit at the L2 memory-side interface)",
                "top_fetch_kernels_KB": dict(sorted(f_per.items(), key=lambda kv: -kv[1])[:8]),
                "top_write_kernels_KB": dict(sorted(w_per.items(), key=lambda kv: -kv[1])[:8]),
            })
        ks = newest(os.path.join(SRC, f"kt_{mode}", "*", "*kernel_stats.csv"))
        if ks:
            rows = list(csv.DictReader(open(ks)))[:12]
            entry["kernel_time_split"] = [{"kernel": short(r["Name"]), "calls": int(r["Calls"]),
                                           "avg_us": float(r["AverageNs"]) / 1e3, "percent": float(r["Percentage"])}
                                          for r in rows]
        out[mode] = entry
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise-only":
        if not glob.glob(os.path.join(SRC, "fetch_*", "*", "*counter_collection.csv")):
            sys.exit("no raw counter tables under gpurun_out/torch_modes (they only exist on the GPU box, "
                     "inside tools/pmc_torch_modes.sh): not overwriting summary.json")
        out = summarise()
        json.dump(out, open(os.path.join(SRC, "summary.json"), "w"), indent=1)
        for m, e in out.items():
            print(m, "bytes/element", e.get("bytes_per_element_total"), "M/s under trace", e["value_under_kernel_trace"] / 1e6)
        # keep the merge-back small: drop the raw counter / trace tables
        for d in glob.glob(os.path.join(SRC, "*_torch_*")):
            if os.path.isdir(d):
                for f in glob.glob(os.path.join(d, "*", "*")):
                    if not f.endswith("kernel_stats.csv"):
                        os.remove(f)
        return
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    summ = json.load(open(os.path.join(SRC, "summary.json")))
    for m, e in summ.items():
        e["source"] = f"profiles/{rnd}/torch_modes_pmc.json (tools/pmc_torch_modes.sh)"
        ks = newest(os.path.join(SRC, f"kt_{m}", "*", "*kernel_stats.csv"))
        if ks:
            shutil.copy(ks, os.path.join(dst, f"torch_modes_kernel_stats_{m}.csv"))
    json.dump(summ, open(os.path.join(dst, "torch_modes_pmc.json"), "w"), indent=1)
    json.dump(summ, open(os.path.join(ROOT, "profiles", "torch_modes_latest.json"), "w"), indent=1)
    print("wrote", os.path.join(dst, "torch_modes_pmc.json"))


if __name__ == "__main__":
    main()
