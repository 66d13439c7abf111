#!/bin/bash
# Kernel timeline (start / end per dispatch) of a few steady-state steps: rocprofv3 --kernel-trace, condensed.
# usage: tools/trace_timeline.sh <out.txt> [bench args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/_tl_raw
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_tl_raw -- python3 bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-hbm-probe "$@" > /dev/null 2> gpurun_out/_tl.err
python3 - "$out" <<'PY'
import csv, glob, sys
rows = []
for p in glob.glob("gpurun_out/_tl_raw/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows = [r for r in rows if "svc::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last ~3 steps
tail = rows[-60:]
t0 = int(tail[0]["Start_Timestamp"])
with open(sys.argv[1], "w") as f:
    f.write("start_us end_us dur_us queue kernel\n")
    for r in tail:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("void svc::", "").replace("svc::", "")
        f.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} q{r.get('Queue_Id','?')} {name}\n")
print(open(sys.argv[1]).read())
PY
rm -rf gpurun_out/_tl_raw
