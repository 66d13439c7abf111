#!/bin/bash
# Where the reference's UNCHANGED encoder application (tests/dropin/ref_encoder_*: apps/encoder.cpp + libs/encoder.cpp on
# compat/opencv2) spends its host time on a 1080p clip: SVC_COMPAT_PROFILE=1 makes the adapter print wall time per cv:: call.
# usage: tools/ref_encoder_profile.sh [frames]   (on the GPU box)
set -u
cd "$GRAFT_REPO_ROOT"
n=${1:-33}
d=$(mktemp -d -p /dev/shm 2>/dev/null || mktemp -d)
python3 - "$d" "$n" <<'PY'
import sys, struct, numpy as np
sys.path.insert(0, ".")
from scalable_video_codec_amd import synth
d, n = sys.argv[1], int(sys.argv[2])
clip = synth.SynthClip(1920, 1080, n, seed=0x5C0DEC02)
with open(d + "/clip.svcbgr", "wb") as f:
    f.write(b"SVCBGR1\0" + struct.pack("<4I", 1920, 1080, n, 0))
    for t in range(n):
        f.write(clip.frame_bgr(t).numpy().tobytes())
PY
for exe in ref_encoder_sse2 "ref_encoder_generic --pyr-lvl-count 3"; do
  echo "== $exe, $n frames of 1080p, stdout to /dev/null"
  t0=$(date +%s.%N)
  SVC_COMPAT_PROFILE=1 tests/dropin/$exe --verbose 0 "$d/clip.svcbgr" > /dev/null
  t1=$(date +%s.%N)
  python3 -c "print('  wall %.2f s for $n frames (process start, GPU initialisation and the first frame included)' % ($t1 - $t0))"
done
rm -rf "$d"
