#!/usr/bin/env python3
"""Condense the rocprofv3 output directories of one gpurun call: per directory, kernel-stats rows of the env / agent kernels and
per-kernel averages of every PMC counter.   python tools/pmc_digest.py gpurun_out/<tag>"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            print("==", os.path.relpath(f, root))
            for r in csv.DictReader(open(f)):
                print("  %-90s calls %6s  avg_ns %10s  min %8s  max %9s  pct %s" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"],
                                                                                  r["MaxNs"], r["Percentage"]))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            print("==", os.path.relpath(f, root))
            acc = {}
            for r in csv.DictReader(open(f)):
                acc.setdefault((r["Kernel_Name"][:80], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
            for (k, c), v in sorted(acc.items()):
                print("  %-80s %-22s n=%5d avg=%.1f" % (k, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
