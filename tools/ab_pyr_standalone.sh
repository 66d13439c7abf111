#!/bin/bash
# round 6: the plane-to-plane pyramid pass standalone / behind a big store kernel / behind a big read kernel (tools/ubench_pyr_standalone.py):
# round 5's LDS-tiled pass against the strip kernel (as built: strips of a band side by side, 8 output rows per wave) and its variants;
# nomath / nohalo are TIMING experiments (wrong results): the pass without its arithmetic, without the strip's extra dword per row.
set -eu
cd "$GRAFT_REPO_ROOT"
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so; trap "cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so" EXIT
for v in ${PYR_VARIANTS:-pyr_tiled pyr_bandmajor pyr_ob4 pyr_ob12 pyr_ob16 pyr_nt pyr_nohalo pyr_nomath pyr_nomath_nohalo}; do
  cp scalable_video_codec_amd/_ab_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== $v"; python3 tools/ubench_pyr_standalone.py 2>/dev/null | grep -v "alone again"
done
cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built"; python3 tools/ubench_pyr_standalone.py 2>/dev/null
echo "== as built, 4K 64 frames 4 levels"; python3 tools/ubench_pyr_standalone.py 64 3840 2160 4 2>/dev/null
