#!/bin/bash
# One MI355X, pipelined schedule: the step time of the shard one rank of an N-GPU strong-scaling run of C3 (and C5) holds.
# Writes one JSON row per shard size to stdout; DESIGN.md 6's predicted table is ms_per_step(N = 1) / ms_per_step(shard).
set -eu
cd "$GRAFT_REPO_ROOT"
row() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({'workload': d['config']['workload'], 'frames_per_gpu': d['config']['frames_per_gpu'], 'encoded_frames_per_step': d['config']['encoded_frames_per_step'], 'ms_per_step': d['ms_per_step'], 'kernel_ms_per_step': d['kernel_ms_per_step'], 'overlapped_ms_per_step': {k: v for k, v in d.get('overlapped_ms_per_step', {}).items() if k != 'note'}}))"; }
for rep in 1 2; do
  for f in 300 150 75 38; do row --frames $f; done
  for f in 64 32 16 8; do row --config C5-4k-4L-dct16 --frames $f; done
done
