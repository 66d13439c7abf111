#!/bin/bash
# Same-box A/B (round 6): the BGR luma + level-1 kernel (luma_pyr1_kernel<true,128,32>: the two-pass order's first stage) with its halo pixels fetched by the
# lanes that hold a row's first / last segment (one 16-byte load each; as built) against a second task loop of three byte loads per pixel (-DSVC_LUMA_HALO_EDGE=0).
# Variant library: tools/build_variant.py luma_halo_loop luma_pyramid.hip -DSVC_LUMA_HALO_EDGE=0
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 --first-encode-reps 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']; print(' ', d['config']['workload'][:24], 'luma+pyramid ms', round(k['luma_pyramid'],4), 'step', round(d['ms_per_step'],3))"; }
all() { run --two-bgr-passes; run --config C3b-1080p-4L-dct8-quant; run --config C5-4k-4L-dct16; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so; trap "cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so" EXIT
for rep in 1 2; do
  cp scalable_video_codec_amd/_ab_luma_halo_loop_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== halo pixels by a second task loop (round 5)"; all
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built (halo pixels by the edge-segment lanes, one 16-byte load each)"; all
done
