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

if out:
    # what the summary was collected ON, next to it: the kernels seen and the hash of their sources in this checkout (the run and this
    # script share one snapshot of the repo on the GPU box).  bench.py quotes counter traffic only for a build with the same hashes.
    import datetime, json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scalable_video_codec_amd import pipeline
    f.close()
    meta = {"summary": os.path.basename(out), "kernels": sorted(acc), "source_sha16": pipeline.kernel_source_hashes(),
            "collected_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"), "bench_args": os.environ.get("SVC_PMC_BENCH_ARGS", "")}
    with open(out + ".meta.json", "w") as m:
        json.dump(meta, m, indent=1)
