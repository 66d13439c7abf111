#!/bin/bash
# Same-box A/B (round 4): the plane-to-plane pyramid pass (luma_pyr1_kernel<false,512,32>: level 1 -> 2 at C3, 10 200 workgroups of
# 1 - 2 us each, 0.39 of the HBM peak) as a fixed grid whose workgroups walk the tiles (-DSVC_PYR_PERSIST=<workgroups>) against
# one workgroup per tile.  Variant libraries scalable_video_codec_amd/_ab_pyr_persist_<N>_libsvc_hip.so.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(' ', d['config']['workload'][:24], 'luma+pyramid ms', round(d['kernel_ms_per_step']['luma_pyramid'],4), 'step', round(d['ms_per_step'],3))"; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so
for rep in 1 2; do
  for v in 1024 2048 4096; do
    cp scalable_video_codec_amd/_ab_pyr_persist_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== persistent grid of $v workgroups"; run; run --config C5-4k-4L-dct16
  done
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built (one workgroup per tile)"; run; run --config C5-4k-4L-dct16
done
cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so
