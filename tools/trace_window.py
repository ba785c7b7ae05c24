#!/usr/bin/env python
"""Per-kernel duration / idle gap over the LAST n kernels of a rocprofv3 --kernel-trace csv that precede
the last kernel whose name contains <anchor>.  usage: trace_window.py <kernel_trace.csv> <anchor> [n]"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
last = max(i for i, r in enumerate(rows) if anchor in r["Kernel_Name"])
win = rows[max(0, last - n):last + 1]
short = lambda s: s.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    dur[k].append((e - s) / 1e3)
    if prev_end is not None:
        gap[k].append((s - prev_end) / 1e3)
    prev_end = e
span = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e3
n_anchor = sum(1 for r in win if anchor in r["Kernel_Name"])
print(f"window: {len(win)} kernels, {span:.0f} us, {n_anchor} x '{anchor}' -> {span / max(n_anchor, 1):.2f} us per anchor kernel")
starts = [int(r["Start_Timestamp"]) for r in win if anchor in r["Kernel_Name"]]
per = sorted((b - a) / 1e3 for a, b in zip(starts, starts[1:]))
if per:
    print(f"start-to-start period of the anchor kernel: median {per[len(per) // 2]:.2f} us, p10 {per[len(per) // 10]:.2f}, "
          f"p90 {per[len(per) * 9 // 10]:.2f}, mean {sum(per) / len(per):.2f}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], gap[k] or [0.0]
    print(f"{k:66s} x{len(d):5d} dur median {statistics.median(d):6.2f} mean {sum(d)/len(d):6.2f}  gap median "
          f"{statistics.median(g):6.2f} mean {sum(g)/len(g):6.2f}")
