#!/bin/bash
# Same-box A/B (round 6): chunks per step of the pipelined schedule at one rank (--chunk-pairs: 299 = whole-shard launches, the schedule of
# rounds 2-5).  Per setting: the steady-state figure (`value`: K back-to-back steps) and what a clip encoded once costs (first_encode).
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms_per_step']; f=d['first_encode']
print(' ', d['config']['workload'][:16], 'chunks', d['config']['chunks_per_step'], 'steady ms', round(d['ms_per_step'],3), 'kernels', round(sum(k.values()),3),
      '| once-through ms', round(f['once_through']['ms_median'],3), '(min', round(f['once_through']['ms_min'],3), ') with prior', round(f['with_prior']['ms_median'],3),
      '(min', round(f['with_prior']['ms_min'],3), ') policy-voided steps', round(f['policy_voided_steps']['ms_per_step'],3), 'sets', d['config']['output_sets'])"; }
for rep in 1 2; do
  for cp in 299 150 100 60; do echo "== C3 --chunk-pairs $cp"; run --chunk-pairs $cp; run --chunk-pairs $cp --wire; done
  for cp in 63 32 21; do echo "== C5 --chunk-pairs $cp"; run --config C5-4k-4L-dct16 --chunk-pairs $cp; done
done
