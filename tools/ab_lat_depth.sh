#!/bin/bash
# A/B on one GPU box: iterations (and streams) given to RANSAC + segmentation in the pipelined schedule.
set -u
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, {k: round(v,3) for k,v in d.get('overlapped_ms_per_step',{}).items() if k!='note'})"; }
for args in "--config C5-4k-4L-dct16" "--config C5-4k-4L-dct16 --frames 8" "--config C5-4k-4L-dct16 --frames 16" "" "--frames 38" "--frames 19" "--config C3b-1080p-4L-dct8-quant"; do
  for d in 1 2 3; do echo -n "[$args] depth $d: "; SVC_LAT_DEPTH=$d run $args; done
done
