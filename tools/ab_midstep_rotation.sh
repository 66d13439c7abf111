#!/bin/bash
# round 6: does test_the_policy_may_flip_at_every_chunk SEE the bug it was written for?  The driver built with
# -DSVC_CLIP_ALLOW_MIDSTEP_ROTATION=1 (the first speculation of a shard may fall on a later chunk of a step: the behaviour before the fix;
# scalable_video_codec_amd/_ab_midstep_libsvc_motion.so, see the commit message for the two build lines) against the driver as built.
cd "$GRAFT_REPO_ROOT"
L=scalable_video_codec_amd/libsvc_motion.so
cp $L /tmp/asbuilt_motion.so
trap "cp /tmp/asbuilt_motion.so $L" EXIT
cp scalable_video_codec_amd/_ab_midstep_libsvc_motion.so $L
echo "== mid-step rotation allowed (before the fix)"
python3 -m pytest tests/test_gpu_clip.py -q -k "flip_at_every_chunk" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -16
cp /tmp/asbuilt_motion.so $L
echo "== as built"
python3 -m pytest tests/test_gpu_clip.py -q -k "flip_at_every_chunk or mixed or chunked" 2>&1 | tail -n 2
