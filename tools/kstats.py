#!/usr/bin/env python
"""Compact view of a rocprofv3 kernel_stats.csv: kernel (template arguments kept, parameter lists dropped), calls,
average / total / min / max duration.  usage: kstats.py <dir or csv> [rows]"""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 18
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
for r in list(csv.DictReader(open(path)))[:rows]:
    m = re.search(r"(k_\w+(<[^>(]*>)?)", r["Name"])
    name = m.group(1) if m else r["Name"][:44]
    print(f"{name:44s} calls {int(r['Calls']):7d}  avg {float(r['AverageNs']) / 1e3:8.1f} us  total "
          f"{float(r['TotalDurationNs']) / 1e6:8.1f} ms  min {float(r['MinNs']) / 1e3:6.1f}  max {float(r['MaxNs']) / 1e3:7.1f}")
