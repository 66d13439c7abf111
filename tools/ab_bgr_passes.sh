#!/bin/bash
# Same-box A/B (round 5): a step that reads the BGR clip ONCE (the transform at the front of the step also leaves the luma plane; planes + quant:
# by speculation, foreground tiles redone; --wire: type words patched) against the two-pass order, per configuration, default policy / never /
# always.  profiles/r05_ab_speculative_quant.txt and r05_ab_wire_one_pass.txt are runs of this.
# usage (GPU box): tools/ab_bgr_passes.sh [repetitions]
set -u
cd "$GRAFT_REPO_ROOT"
reps=${1:-2}
row() { python3 bench.py --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['workload'][:4], '$*', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, round(d.get('overlapped_ms_per_step',{}).get('type_patch',0),4), 'fg', round(d['config']['foreground_mv_blocks'],4), d['config']['bgr_passes_per_step'][:3])"; }
for rep in $(seq $reps); do
  for cfg in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16 C2-720p-3L-dct8; do
    for mode in "" "--two-bgr-passes" "--always-speculate" "--wire" "--wire --two-bgr-passes"; do row --config $cfg $mode; done
  done
done
