#!/bin/bash
# A/B on one GPU box: the pipelined schedule with the latency-bound stages in their stand-alone shapes
# (SVC_LAUNCH_BESIDE=0) against the shapes that fit next to the bandwidth kernels (default), and the serial schedule.
set -u
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-hbm-probe "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, {k: round(v,3) for k,v in d.get('overlapped_ms_per_step',{}).items() if k!='note'})"; }
for rep in 1 2; do
for args in "" "--frames 150" "--frames 75" "--frames 38" "--config C3b-1080p-4L-dct8-quant" "--config C2-720p-3L-dct8"; do
  echo -n "[$args] beside : "; SVC_LAUNCH_BESIDE=1 run $args
  echo -n "[$args] alone  : "; SVC_LAUNCH_BESIDE=0 run $args
  echo -n "[$args] serial : "; run $args --schedule serial
done; done
