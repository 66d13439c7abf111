#!/bin/bash
# round 6: counters of a kernel BEFORE / AFTER a change, same box, same command: the shipped library against an A/B variant
# (tools/build_variant.py).  One counter group per pass (tools/pmc_passes.sh).
#   usage (GPU box): tools/pmc_variants.sh <outdir> "<groups>" <variant or 'asbuilt'> [<variant> ...] -- [bench args]
set -u
out=$1; groups=$2; shift 2
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
[ $# -gt 0 ] && shift
cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt_pmc.so; trap "cp /tmp/asbuilt_pmc.so scalable_video_codec_amd/libsvc_hip.so" EXIT
for v in "${variants[@]}"; do
  if [ "$v" = asbuilt ]; then cp /tmp/asbuilt_pmc.so scalable_video_codec_amd/libsvc_hip.so; else cp scalable_video_codec_amd/_ab_${v}_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; fi
  PMC_GROUPS="$groups" bash tools/pmc_passes.sh "$out/$v" "$@" > "$out/$v.log" 2>&1
  cp "$out/$v/summary.csv" "$out/${v}_summary.csv"
  echo "== $v"; cat "$out/${v}_summary.csv" | grep -i "${PMC_KERNEL_FILTER:-.}" || true
done
