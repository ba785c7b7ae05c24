#!/usr/bin/env python
"""Timeline of a free-running NUTS run from a rocprofv3 --kernel-trace csv (argument: path to
*_kernel_trace.csv): ticks in buckets of 100, per bucket the mean duration of each tick kernel and of
the callable, the start-to-start period of a tick and the share of that period the GPU was idle.
Shows where a run spends its time: the busy phase (every chain has a leaf in flight), the
two-kernel -> fused switch, and the tail (a few deep trees, bound by dependent launches)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bucket = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def kind(name):
    if "spec_integrate" in name:  # the speculative tail's stream-A integrator: one launch per leapfrog
        return "leaf"
    if "async_tick3<" in name:  # round 4: the lean leaf (with deferred transition ends: the whole tick)
        return "leaf"
    if "async_tick2<" in name:  # k_nuts_async_tick2<NI, MODE, WAVES> (traces of rounds 2-4 only: removed in round 5): MODE 0 leaf, 1 end, 2 fused
        mode = name.split("async_tick2<")[1].split(",")[1].strip()
        return {"0": "leaf", "1": "end", "2": "fused"}.get(mode, "other")
    for key, k in (("async_leaf", "leaf"), ("async_end2", "end"), ("async_boundary", "end"),
                   ("async_fused", "fused"), ("k_neal_funnel", "callable"), ("k_diag_gaussian", "callable"),
                   ("async_compact", "compact"), ("async_gather", "compact")):
        if key in name:
            return k
    return "other"


ticks = []  # one entry per tick: {"start":, "end":, kinds: {kind: dur}}
cur = None
for r in rows:
    k = kind(r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if k in ("leaf", "fused"):
        if cur:
            ticks.append(cur)
        cur = {"start": s, "busy": 0.0, "k": {}}
    if cur is None:
        continue
    cur["k"][k] = cur["k"].get(k, 0.0) + (e - s) / 1e3
    cur["busy"] += (e - s) / 1e3
if cur:
    ticks.append(cur)
print(f"{len(ticks)} ticks; columns: mean us per tick of each kernel kind, tick period, idle share")
print(f"{'ticks':>13s} {'leaf':>7s} {'end':>7s} {'fused':>7s} {'callable':>8s} {'other':>7s} {'period':>8s} {'idle%':>6s}  total ms")
tot_ms = 0.0
for i in range(0, len(ticks) - 1, bucket):
    blk = ticks[i:i + bucket + 1]
    n = len(blk) - 1
    if n <= 0:
        break
    period = (blk[-1]["start"] - blk[0]["start"]) / 1e3 / n
    mean = lambda key: sum(t["k"].get(key, 0.0) for t in blk[:-1]) / n
    busy = sum(t["busy"] for t in blk[:-1]) / n
    other = mean("other") + mean("compact")
    tot_ms += period * n / 1e3
    print(f"{i:6d}-{i + n:6d} {mean('leaf'):7.1f} {mean('end'):7.1f} {mean('fused'):7.1f} {mean('callable'):8.1f} "
          f"{other:7.1f} {period:8.1f} {100 * max(0.0, 1 - busy / period):6.1f}  {tot_ms:8.1f}")
