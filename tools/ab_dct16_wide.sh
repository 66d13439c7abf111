#!/bin/bash
# Same-box A/B (round 6): the 16 x 16 transform with two adjacent columns per lane (dct16_wide_kernel: 512 contiguous bytes per store
# instruction, eight segment columns per wave; as built) against dct_kernel<16> (one float per lane: 256 bytes; -DSVC_DCT16_WIDE=0).
# Variant library: tools/build_variant.py dct16_narrow dct.hip -DSVC_DCT16_WIDE=0.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 --config C5-4k-4L-dct16 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']; print(' ', d['config']['workload'][:24], 'transform ms', round(k['dct_quant'],4), 'frac', round(d['roofline_dct']['frac'],3), 'step', round(d['ms_per_step'],3))"; }
all() { echo -n " serial   "; run --schedule serial; echo -n " pipelined"; run; echo -n " serial, speculative (luma plane too)"; run --schedule serial --always-speculate; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so; trap "cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so" EXIT
for rep in 1 2; do
  cp scalable_video_codec_amd/_ab_dct16_narrow_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== dct_kernel<16> (one float per lane)"; all
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built (dct16_wide_kernel)"; all
done
cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so
