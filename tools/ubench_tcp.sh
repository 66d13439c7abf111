#!/bin/bash
# Runs tools/_bin/ubench_tcp plain (times) and under rocprofv3 --pmc (tag lookups per wave instruction).
set -u
out=gpurun_out/ubench_tcp; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/_bin/ubench_tcp > $out/times.txt 2>&1
cat $out/times.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_WAVES TCP_GATE_EN1_sum --output-format csv -d $out/pmc -- tools/_bin/ubench_tcp > $out/pmc.log 2>&1
echo "pmc exit $?"
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("gpurun_out/ubench_tcp/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/ubench_tcp/lookups.txt", "w") as f:
    for k in sorted(acc, key=lambda s: int(s.split("<")[1].split(">")[0]) if "<" in s else -1):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        line = f"{k:24s} lookups/inst {c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / max(1, c.get('SQ_INSTS_VMEM_RD', 1)):7.2f}  vmem_rd {c.get('SQ_INSTS_VMEM_RD', 0):.4g} gate_en1 {c.get('TCP_GATE_EN1_sum', 0):.4g}"
        print(line); f.write(line + "\n")
PY
rm -rf $out/pmc
