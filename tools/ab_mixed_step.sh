#!/bin/bash
# round 6: a step into an empty pipeline that knows nothing about the clip -- the mixed form (first half two passes, second half reading its
# frames once, blind) against two-pass halves (--no-mixed-steps) and against a whole-shard step (--whole-shard-steps): first_encode of the bench line.
# usage: tools/ab_mixed_step.sh [reps]
cd "$GRAFT_REPO_ROOT"
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); fe=d["first_encode"]
o,w,v=fe["once_through"],fe["with_prior"],fe["policy_voided_steps"]
print("  %-16s steady ms %.3f | once-through ms %.3f (min %.3f, %d of %d chunk launches blind) with prior %.3f policy-voided steps %.3f" % (d["config"]["workload"][:16], d["ms_per_step"], o["ms_median"], o["ms_min"], o["chunk_launches_speculated"], o["chunk_launches"], w["ms_median"], v["ms_per_step"]))'
for r in $(seq 1 ${1:-3}); do
for c in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do
  for f in "--mixed-steps" "" "--whole-shard-steps"; do
    echo "== $c ${f:-as built (two-pass halves)}"
    python3 bench.py --config $c $f --no-cpu-baseline --no-end-to-end --sustain-seconds 0 --no-hbm-probe 2>/dev/null | python3 -c "$pick"
  done
done
done
