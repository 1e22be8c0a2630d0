#!/usr/bin/env python3
"""Condenses a tools/prof.sh output directory into a short text summary (kernel stats + HBM traffic)."""
import csv
import glob
import os
import sys

d = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return r[0] if r else None


f = find("trace", "*kernel_stats.csv")
if f:
    print("== kernel stats (%s)" % os.path.relpath(f, d))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8:
            print(", ".join(row))
f = find("trace", "*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        k = r["Kernel_Name"]
        by.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== per-kernel durations from the trace (ns): name, calls, avg, min, max; VGPR/SGPR/LDS of first call")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        r0 = next(r for r in rows if r["Kernel_Name"] == k)
        print("%s, %d, %.0f, %d, %d, vgpr=%s sgpr=%s lds=%s scratch=%s wg=%s grid=%s" % (
            k[:90], len(v), sum(v) / len(v), min(v), max(v), r0.get("VGPR_Count"), r0.get("SGPR_Count"),
            r0.get("LDS_Block_Size"), r0.get("Scratch_Size"), r0.get("Workgroup_Size"), r0.get("Grid_Size")))
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    print("== %s per dispatch (raw counter units = KiB per the rocprofv3 definition)" % name)
    for k, v in acc.items():
        print("%s, n=%d, avg=%.1f KiB (%.3f MB)" % (k[:90], len(v), sum(v) / len(v), sum(v) / len(v) * 1024 / 1e6))
