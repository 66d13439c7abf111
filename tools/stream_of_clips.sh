#!/bin/bash
# round 6: first_encode.stream_of_clips (K different clips, each encoded once where it is: svc_clip_step_frames) beside the steady state of the
# resident clip, the once-through step and the others, per configuration; two repetitions.
cd "$GRAFT_REPO_ROOT"
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); fe=d["first_encode"]; sc=fe["stream_of_clips"]
print("  %-16s steady %.3f ms = %.1f k | stream of clips %.3f ms = %.1f k (%d of %d chunk launches one-pass, share %.4f) | once-through %.3f with prior %.3f voided %.3f" % (d["config"]["workload"][:16], d["ms_per_step"], d["value"]/1e3, sc["ms_per_clip"], sc["value"]/1e3, sc["chunk_launches_speculated"], sc["chunk_launches"], sc["foreground_share_last"], fe["once_through"]["ms_median"], fe["with_prior"]["ms_median"], fe["policy_voided_steps"]["ms_per_step"]))'
for r in 1 2; do for c in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do python3 bench.py --config $c --no-cpu-baseline --no-end-to-end --sustain-seconds 0 --no-hbm-probe 2>/dev/null | python3 -c "$pick"; done; python3 bench.py --wire --no-cpu-baseline --no-end-to-end --sustain-seconds 0 --no-hbm-probe 2>/dev/null | python3 -c "$pick"; done
