#!/bin/bash
# Same-box A/B (round 4): luma_pyr1_kernel<true, TW, TH> with every load of a workgroup issued in ONE round (tiles whose (TH + 4) x TW / 16
# segment tasks fit 256 lanes: 128 x 28 -- the new default --, 256 x 12, 128 x 12, 64 x 60) against 128 x 32 (two passes of the segment loop
# plus a separate halo loop: three dependent memory round trips per workgroup; the shape shipped until now).  Parity tests run with every
# variant library in place first.  Variant libraries scalable_video_codec_amd/_ab_luma_<TW>x<TH>_libsvc_hip.so.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(' ', d['config']['workload'][:24], 'luma+pyramid ms', round(d['kernel_ms_per_step']['luma_pyramid'],4), 'step', round(d['ms_per_step'],3))"; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so
trap 'cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so' EXIT
echo "== parity, as built (128 x 28, one round)"; timeout -k 10 300 python3 -m pytest tests/test_gpu_ransac_pyramid.py tests/test_gpu_imageops.py tests/test_gpu_misc_property.py tests/test_gpu_shape_sweep.py -x -q 2>&1 | tail -1
for v in 256x12 128x12 64x60; do
  cp scalable_video_codec_amd/_ab_luma_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so
  echo "== parity, $v"; timeout -k 10 300 python3 -m pytest tests/test_gpu_ransac_pyramid.py tests/test_gpu_misc_property.py -x -q -k "pyr or luma" 2>&1 | tail -1
done
for rep in 1 2; do
  for v in 128x32 256x12 128x12 64x60; do
    cp scalable_video_codec_amd/_ab_luma_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== tile $v"; run; run --config C5-4k-4L-dct16
  done
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built (128 x 28, one round of loads)"; run; run --config C5-4k-4L-dct16
done
