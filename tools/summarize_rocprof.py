#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats run into the summary kept under profiles/.
usage: summarize_rocprof.py <dir with *_kernel_stats.csv> <out.csv> [note]"""
import csv, glob, os, sys
src, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
paths = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)
assert paths, f"no *_kernel_stats.csv under {src}"
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
ours = [r for r in rows if "svc::" in r["Name"]]
other_ns = sum(int(r["TotalDurationNs"]) for r in rows if "svc::" not in r["Name"])
with open(out, "w", newline="") as f:
    if note:
        f.write(f"# {note}\n")
    f.write("# source: rocprofv3 --kernel-trace --stats; durations in ns; non-svc rows (torch data generation, copies) summed in the last line\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in sorted(ours, key=lambda r: -int(r["TotalDurationNs"])):
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    w.writerow(["(everything else: torch generator / copies, outside the timed region)", "", other_ns, "", "", "", ""])
print(open(out).read())
