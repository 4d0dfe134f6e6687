#!/usr/bin/env python3
"""List per-dispatch durations (ms) of kernels whose name contains a pattern, in launch order.
usage: trace_list.py DIR PATTERN [PATTERN...]   (DIR holds rocprofv3 --kernel-trace --output-format csv output)"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
for s, e, n in rows:
    if any(p in n for p in sys.argv[2:]):
        print(f"{n:40s} {(e - s) / 1e6:9.3f} ms")
