#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection CSVs: one line per dispatch of the kernels whose name starts with PREFIX
usage: pmc_per_dispatch.py PREFIX DIR [DIR...]"""
import collections
import csv
import glob
import os
import sys


def main():
    prefix = sys.argv[1]
    rows = collections.defaultdict(dict)
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Kernel_Name"].startswith(prefix):
                    rows[(d, int(row["Dispatch_Id"]))][row["Counter_Name"]] = float(row["Counter_Value"])
    for (d, i), c in sorted(rows.items()):
        print(os.path.basename(d.rstrip("/")), i, " ".join("%s=%.0f" % kv for kv in sorted(c.items())))


if __name__ == "__main__":
    main()
