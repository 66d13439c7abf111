#!/bin/bash
# A/B on the GPU box: pair-major vs XCD-region-major workgroup order of the fused motion search (serial schedule).
set -u
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['config']['workload'][:24], 'hbma ms', round(d['kernel_ms_per_step']['hbma'],4), 'frac', round(d['roofline']['frac'],3), 'step', round(d['ms_per_step'],3))"; }
for o in pair region pair region; do echo "== order $o"; for c in C3-1080p-3L-dct8-quant C5-4k-4L-dct16 C3b-1080p-4L-dct8-quant; do SVC_HBMA_ORDER=$o run --config $c; done; done
SVC_HBMA_ORDER=region run --frames 38
SVC_HBMA_ORDER=pair run --frames 38
