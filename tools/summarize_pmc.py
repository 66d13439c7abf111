#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter rows per kernel (svc:: kernels only).
usage: summarize_pmc.py <dir with pass*/.../*_counter_collection.csv> [out.csv]"""
import csv, glob, os, sys, collections
src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        name = r.get("Kernel_Name", "")
        if "svc::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = [["kernel", "counter", "launches", "mean_per_launch", "max_per_launch"]]
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        lines.append([k, c, len(v), f"{sum(v)/len(v):.6g}", f"{max(v):.6g}"])
out = sys.argv[2] if len(sys.argv) > 2 else None
f = open(out, "w", newline="") if out else sys.stdout
csv.writer(f).writerows(lines)
