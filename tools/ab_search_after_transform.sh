#!/bin/bash
# Same-box A/B (round 6): the motion search, not a pyramid pass, right behind the transform kernel (--search-after-transform: a one-pass
# micro-step runs transform(m) | search(m - 1) | pyramid levels(m), a two-pass one luma(m) | transform(m - 3) | search(m)) against the
# shipped order.  Whatever follows the transform shares the memory system with the write-back of what it left dirty.
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 --first-encode-reps 0 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']
print(' ', d['config']['workload'][:16], 'ms/step', round(d['ms_per_step'],3), {a: round(b,3) for a,b in k.items()})"; }
for rep in 1 2 3; do
  for args in "" "--two-bgr-passes" "--wire" "--config C3b-1080p-4L-dct8-quant" "--config C5-4k-4L-dct16"; do
    echo "== [$args] as built"; run $args
    echo "== [$args] --search-after-transform"; run $args --search-after-transform
  done
done
