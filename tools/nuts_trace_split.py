#!/usr/bin/env python
"""Per-kernel durations of the free-running NUTS ticks in which all chains are busy, from a
rocprofv3 --kernel-trace csv (argument: path to *_kernel_trace.csv)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = {"leaf": "async_leaf", "boundary": "async_boundary", "callable": "k_neal_funnel"}
series = {k: [dur(r) for r in rows if v in r["Kernel_Name"]] for k, v in names.items()}
for k, v in series.items():
    v_sorted = sorted(v, reverse=True)
    top = v_sorted[: max(1, len(v) // 10)]
    print(f"{k:9s} calls {len(v):5d}  mean {sum(v)/len(v):7.1f} us  top-decile mean {sum(top)/len(top):7.1f} us  "
          f"median {v_sorted[len(v)//2]:6.1f} us  total {sum(v)/1e3:7.1f} ms")
