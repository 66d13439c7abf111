#!/bin/bash
# A/B on the GPU box: the fused motion search with 64-bit per-lane addresses (the previous build, copied to
# scalable_video_codec_amd/_ab_prev_libsvc_hip.so) against scalar plane base + 32-bit lane offsets (as built).  Serial schedule: the event time is the kernel alone.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe "$@" | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['config']['workload'][:24], 'hbma ms', round(d['kernel_ms_per_step']['hbma'],4), 'frac', round(d['roofline']['frac'],3), 'step', round(d['ms_per_step'],3))"; }
all() { run; run --config C5-4k-4L-dct16; run --config C3b-1080p-4L-dct8-quant; run --config C2-720p-3L-dct8; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/new.so
for rep in 1 2; do
  cp scalable_video_codec_amd/_ab_prev_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== previous build (64-bit lane addresses)"; all
  cp /tmp/new.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built (scalar base + 32-bit offsets)"; all
done
python3 -m pytest tests/test_gpu_hbma.py tests/test_gpu_golden.py tests/test_gpu_hbma_property.py -m gpu -q 2>&1 | tail -3
