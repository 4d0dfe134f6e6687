#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel and counter.
usage: pmc_summary.py DIR [DIR...]   (searches DIR recursively for *counter_collection.csv)"""
import csv, glob, os, sys, collections

def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row["Dispatch_Id"])
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        print(f"{k}  launches {len(launches[k])}")
        for c, v in sorted(acc[k].items()):
            print(f"    {c:28s} {v:18.0f}   per launch {v / max(len(launches[k]), 1):16.0f}")

if __name__ == "__main__":
    main()
