#!/bin/bash
# Same-box A/B (round 4): tile shape of the luma + level-1 kernel (luma_pyr1_kernel<true, TW, TH>).  With 128-pixel tiles a
# level-1 row of a tile is 64 bytes -- half a cache line, the other half written by the neighbouring workgroup at another time;
# 256-pixel tiles write whole lines (the record-emitting transform lost a third of its time to partial-line stores:
# profiles/r04_ab_wire_stretch.txt).  Variant libraries scalable_video_codec_amd/_ab_luma_<TW>x<TH>_libsvc_hip.so.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(' ', d['config']['workload'][:24], 'luma+pyramid ms', round(d['kernel_ms_per_step']['luma_pyramid'],4), 'step', round(d['ms_per_step'],3))"; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so
for rep in 1 2; do
  for v in 256x32 256x16 128x16; do
    cp scalable_video_codec_amd/_ab_luma_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== tile $v"; run; run --config C5-4k-4L-dct16
  done
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built (128x32)"; run; run --config C5-4k-4L-dct16
done
cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so
