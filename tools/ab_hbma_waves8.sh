#!/bin/bash
# A/B on the GPU box: the fused motion search as built (65-66 VGPRs for <3,2> / <4,1>: 7 waves per SIMD) against the same
# source forced to 64 VGPRs (8 waves, small spill).  Serial schedule, so the HIP-event time is the kernel alone.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe "$@" | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['config']['workload'][:24], 'hbma ms', round(d['kernel_ms_per_step']['hbma'],4), 'frac', round(d['roofline']['frac'],3), 'step', round(d['ms_per_step'],3))"; }
echo "== as built"; run; run --config C5-4k-4L-dct16; run --config C3b-1080p-4L-dct8-quant
touch scalable_video_codec_amd/csrc/hbma_fused.hip
SVC_EXTRA_HIPCC_FLAGS=-DSVC_HBMA_WAVES8 python3 -m scalable_video_codec_amd.build > /dev/null
echo "== forced 8 waves/SIMD"; run; run --config C5-4k-4L-dct16; run --config C3b-1080p-4L-dct8-quant
