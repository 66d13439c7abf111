#!/bin/bash
# Same-box A/B (round 6): the idle-pipeline rule (a step that finds the pipeline empty runs in two chunks on big shards in the two-pass
# order) against whole-shard steps everywhere (--whole-shard-steps): first_encode.once_through and the steady state, C3 / C3b / C5.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); f=d['first_encode']
print(' ', d['config']['workload'][:16], 'steady ms', round(d['ms_per_step'],3), '| once-through ms', round(f['once_through']['ms_median'],3), '(min', round(f['once_through']['ms_min'],3), ') with prior', round(f['with_prior']['ms_median'],3), 'policy-voided steps', round(f['policy_voided_steps']['ms_per_step'],3))"; }
for rep in 1 2 3; do
  for c in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant; do
    echo "== $c whole-shard steps"; run --config $c --whole-shard-steps
    echo "== $c as built (idle-pipeline rule)"; run --config $c
  done
done
echo "== C3 --wire, --two-bgr-passes (as built)"; run --wire; run --two-bgr-passes; run --two-bgr-passes --whole-shard-steps
