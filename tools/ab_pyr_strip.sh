#!/bin/bash
# Same-box A/B (round 6): the plane-to-plane pyramid pass as a wave per column strip with no LDS (pyr_strip_kernel, as built; variants with
# 8 / 32 output rows per wave) against round 5's LDS-tiled pass (-DSVC_PYR_STRIP=0).  Serial schedule: the event time is the stage alone.
# Variant libraries: tools/build_variant.py pyr_tiled luma_pyramid.hip -DSVC_PYR_STRIP=0, pyr_strip_ob<N> ... -DSVC_PYR_STRIP_OB=<N>.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']; print(' ', d['config']['workload'][:24], 'luma+pyramid ms', round(k['luma_pyramid'],4), 'transform', round(k.get('dct_quant', k.get('dct', 0)),4), 'step', round(d['ms_per_step'],3))"; }
all() { run --always-speculate; run --config C5-4k-4L-dct16; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so; trap "cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so" EXIT
for rep in 1 2; do
  for v in pyr_tiled pyr_halo_all_lanes; do
    cp scalable_video_codec_amd/_ab_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== $v"; all
  done
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built (pyr_strip_kernel<8>, strips of a band side by side, halo dword under a two-lane mask)"; all
done
cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so
