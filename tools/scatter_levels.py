#!/usr/bin/env python3
"""per-level durations of k_rp_scatter / k_rp_hist from a rocprofv3 --kernel-trace CSV (dispatch order: level 1, 2, 3, 1, ...)
usage: scatter_levels.py kernel_trace.csv [levels]"""
import csv
import sys


def main():
    path, levels = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for kern in ("k_rp_scatter", "k_rp_hist", "k_hash_reads"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith(kern)]
        if not d:
            continue
        n = levels if kern == "k_rp_scatter" else (levels - 1 if kern == "k_rp_hist" else 1)
        for l in range(n):
            x = d[l::n]
            print("%s level %d: %d launches, avg %.3f ms, min %.3f, max %.3f" % (kern, l + 1 + (1 if kern == "k_rp_hist" else 0), len(x), sum(x) / len(x), min(x), max(x)))


if __name__ == "__main__":
    main()
