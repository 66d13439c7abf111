#!/bin/bash
# TIMING experiment (round 6): the ceiling of hbma_tiled16_kernel for any scheme that only reorders or stages level 0's
# reads -- level 0 searched as if the field were coherent: every lane's window on its block's own rows (l0c1), at its own columns too (l0c2).
# The variants compute WRONG vectors (never shipped): tools/build_variant.py hbma_l0c<N> hbma_tiled.hip -DSVC_HBMA_L0_CEILING=<N>.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']; print(' ', d['config']['workload'][:24], 'hbma ms', round(k['hbma'],4), 'frac', round(d['roofline']['frac'],3), d['roofline']['kernel'][:20])"; }
all() { run --config C3b-1080p-4L-dct8-quant; run --config C5-4k-4L-dct16; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so; trap "cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so" EXIT
for rep in 1 2; do
  for v in hbma_l0c1 hbma_l0c2; do
    cp scalable_video_codec_amd/_ab_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== $v (timing only)"; all
  done
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built"; all
done
cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so
