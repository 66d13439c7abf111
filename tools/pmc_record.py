#!/usr/bin/env python3
"""Folds a counter summary (tools/pmc_passes.sh groups 6 and 7 -> summary.csv + summary.csv.meta.json) into profiles/pmc_traffic.json:
HBM-side bytes per launch of the three big kernels from the L2's memory-side request counters by size, WITH what it was collected
on (kernel names, source hashes) so that bench.py can refuse a figure whose kernels have changed since.

usage: pmc_record.py <config name> <summary.csv> <pairs> [--wire] [--as profiles/<name>_traffic_summary.csv]"""
import argparse
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("summary")
ap.add_argument("pairs", type=int)
ap.add_argument("--wire", action="store_true", help="the run was bench.py --wire: only the record-emitting transform's figure is taken")
ap.add_argument("--as", dest="dest", default=None, help="copy the summary (+ meta) to this path under profiles/ and cite that")
ap.add_argument("--only-dct-as", default=None, metavar="KEY",
                help="record ONLY the transform's figure, under KEY (dct_bytes_per_launch: the two-pass order's kernel; dct_luma_bytes_per_launch: the "
                     "front-of-step kernel that also stores the luma plane) -- for a second collection of the same configuration in the other order")
a = ap.parse_args()

rows = {}
for r in csv.DictReader(open(a.summary)):
    rows.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_launch"])
meta = json.load(open(a.summary + ".meta.json"))


def hbm_bytes(k):
    c = rows[k]
    need = ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_64B_sum")
    if any(n not in c for n in need):
        sys.exit(f"{k}: groups 6 and 7 of tools/pmc_passes.sh are needed ({[n for n in need if n not in c]} missing)")
    return 32 * c[need[0]] + 64 * c[need[1]] + 128 * c[need[2]] + 64 * c[need[3]]


def pick(pred):
    ks = [k for k in rows if pred(k)]
    if len(ks) != 1:
        sys.exit(f"expected one kernel, got {ks}")
    return ks[0]


src = a.summary
if a.dest:
    os.makedirs(os.path.dirname(os.path.join(ROOT, a.dest)), exist_ok=True)
    shutil.copy(a.summary, os.path.join(ROOT, a.dest))
    shutil.copy(a.summary + ".meta.json", os.path.join(ROOT, a.dest + ".meta.json"))
    src = a.dest
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
db = json.load(open(path))
rec = db.setdefault(a.config, {})
on = rec.setdefault("collected_on", {"kernels": {}, "source_sha16": {}})
def pick_dct():
    """The step's big transform launch: not the redo of the foreground tiles (SPEC = 2: the last template argument), the one that moves
    the most bytes (a speculating step has both; so has a run that switched order half-way)."""
    ks = [k for k in rows if "dct_kernel" in k and not k.rstrip(">").rstrip().endswith(", 2")]
    if not ks:
        sys.exit("no dct_kernel in the summary")
    return max(ks, key=hbm_bytes)


dct = pick_dct()
if a.only_dct_as:
    rec[a.only_dct_as] = hbm_bytes(dct)
    rec[a.only_dct_as.replace("_bytes_per_launch", "_note")] = f"{dct} ({src})"
    if on["source_sha16"].get("dct") not in (None, meta["source_sha16"]["dct"]):
        sys.exit("the other figures of this config were collected on other dct sources: re-collect them first")
elif a.wire:
    rec["dct_records_bytes_per_launch"] = hbm_bytes(dct)
    rec["dct_records_note"] = f"bench.py --wire, {dct} ({src})"
    if on["source_sha16"].get("dct") not in (None, meta["source_sha16"]["dct"]):
        sys.exit("the plane-form figure of this config was collected on other dct sources: re-collect it first")
else:
    hb = pick(lambda k: "hbma_" in k)
    lus = [k for k in rows if "luma_pyr1_kernel<true" in k]
    spec = dct.rstrip(">").rstrip().endswith(", true, 1")  # the front-of-step kernel (also stores the luma plane): a figure of its own
    for k in ("dct_bytes_per_launch", "dct_note", "dct_luma_bytes_per_launch", "dct_luma_note"):
        rec.pop(k, None)
    rec.update({"source": src, "pairs": a.pairs, "hbma_bytes_per_launch": hbm_bytes(hb),
                "dct_luma_bytes_per_launch" if spec else "dct_bytes_per_launch": hbm_bytes(dct),
                "hbma_note": f"{hb}; L2 memory-side request counters by size", "dct_luma_note" if spec else "dct_note": dct})
    # a step that reads the BGR clip once runs the BGR luma kernel on the clip's first frame only: its figure is not a clip's
    rec.pop("luma_pyr1_bytes_per_launch", None)
    if lus and not any(", true, 1>" in k for k in rows if "dct_kernel" in k):
        rec["luma_pyr1_bytes_per_launch"] = hbm_bytes(lus[0])
    rec.pop("dct_records_bytes_per_launch", None)  # tied to the dct sources of an older collection
    rec.pop("dct_records_note", None)
    for k in ("round3", "hbma_bytes_per_launch_pair_major_order"):
        rec.pop(k, None)
    on["kernels"] = {"hbma": [hb], "dct": [dct], **({"luma_pyr1": lus[:1]} if lus else {})}
    on["source_sha16"] = dict(meta["source_sha16"])
    on["collected_utc"] = meta["collected_utc"]
    on.pop("note", None)
json.dump(db, open(path, "w"), indent=1)
print(json.dumps(rec, indent=1))
