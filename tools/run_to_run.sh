#!/bin/bash
# round 6: the default bench line five times in a row on ONE box (short form: no CPU leg, no end-to-end, no sustained run): how much of the
# box-to-box spread of `value` is really run-to-run.  usage: tools/run_to_run.sh [n]
cd "$GRAFT_REPO_ROOT"
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
print("  value %.1f k  ms/step %.3f  luma+pyramid %.3f  search %.3f  transform %.3f  | once-through %.3f" % (d["value"]/1e3, d["ms_per_step"], k["luma_pyramid"], k["hbma"], k["dct_quant"], d["first_encode"]["once_through"]["ms_median"]))'
for i in $(seq 1 ${1:-5}); do python3 bench.py --no-cpu-baseline --no-end-to-end --sustain-seconds 0 --no-hbm-probe 2>/dev/null | python3 -c "$pick"; done
