#!/usr/bin/env python
"""Tail of a free-running NUTS run, from a rocprofv3 --kernel-trace csv (argument: path to
*_kernel_trace.csv): what one tick costs once only a few deep trees are left -- per-kernel duration,
the idle gap in front of each kernel, and the start-to-start period of the tick kernel."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
def is_tick(name):  # the one-launch tick: round-1 k_nuts_async_fused, k_nuts_async_tick2<NI, 2, WAVES>, or (round 4)
    # the lean leaf with deferred transition ends k_nuts_async_tick3<64, 1, W, true>
    if "async_fused" in name or "async_tick3<" in name:
        return True
    return "async_tick2<" in name and name.split("async_tick2<")[1].split(",")[1].strip() == "2"


tick_idx = [i for i, r in enumerate(rows) if is_tick(r["Kernel_Name"])]
if not tick_idx:
    sys.exit("no one-launch ticks in the trace")
n_tail = min(int(sys.argv[2]) if len(sys.argv) > 2 else 400, len(tick_idx) - 1)
first = tick_idx[-n_tail - 1]
tail = rows[first:tick_idx[-1]]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    dur[k].append((e - s) / 1e3)
    if prev_end is not None:
        gap[k].append((s - prev_end) / 1e3)
    prev_end = e
starts = [int(rows[i]["Start_Timestamp"]) for i in tick_idx[-n_tail - 1:]]
period = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
period.sort()
print(f"last {n_tail} ticks: start-to-start period median {period[len(period)//2]:.1f} us, mean {sum(period)/len(period):.1f} us")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], gap[k] or [0.0]
    print(f"{k:62s} x{len(d):5d}  dur mean {sum(d)/len(d):6.2f} us   gap before mean {sum(g)/len(g):6.2f} us")
