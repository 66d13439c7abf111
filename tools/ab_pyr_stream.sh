#!/bin/bash
# Same-box A/B (round 4): the plane-to-plane pyramid pass as a double-buffered LDS-DMA stream (pyr_plane_stream_kernel,
# -DSVC_PYR_STREAM=<workgroups>) against one workgroup per tile (luma_pyr1_kernel<false,512,32>) and against the plain persistent grid
# (tools/ab_pyr_persist.sh).  Runs the pyramid parity tests with each variant library in place first.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(' ', d['config']['workload'][:24], 'luma+pyramid ms', round(d['kernel_ms_per_step']['luma_pyramid'],4), 'step', round(d['ms_per_step'],3))"; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so
trap 'cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so' EXIT
for v in ${VARIANTS:-1024 2048}; do
  cp scalable_video_codec_amd/_ab_pyr_stream_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so
  echo "== parity, stream grid $v"; timeout -k 10 300 python3 -m pytest tests/test_gpu_ransac_pyramid.py tests/test_gpu_imageops.py tests/test_gpu_misc_property.py -x -q -k "pyr or luma" 2>&1 | tail -2
done
for rep in 1 2; do
  for v in ${VARIANTS:-1024 2048}; do
    cp scalable_video_codec_amd/_ab_pyr_stream_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== LDS-DMA stream, grid of $v workgroups"; run; run --config C5-4k-4L-dct16
  done
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built"; run; run --config C5-4k-4L-dct16
done
