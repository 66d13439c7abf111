#!/bin/bash
# Same-box A/B of the 4-level motion search: lane-per-block kernel (no LDS) vs the LDS-tiled kernel.
# usage (on the GPU box): tools/ab_hbma_tiled.sh [out file]
out=${1:-gpurun_out/ab_hbma_tiled.txt}
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-probe "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']
print('  value %8.0f  ms/step %.3f  hbma %.4f ms  frac %.3f   %s' % (d['value'], d['ms_per_step'], k.get('hbma',0), d['roofline']['frac'], d['roofline']['kernel'][:24]))"; }
{
for rep in 1 2; do
for c in ${CONFIGS:-C5-4k-4L-dct16 C3b-1080p-4L-dct8-quant C3-1080p-3L-dct8-quant}; do
  for k in lane tiled; do
    for s in serial pipelined; do echo "== $c kernel=$k schedule=$s"; run --config $c --hbma-kernel $k --schedule $s; done
  done
done
done
} 2>&1 | tee $out
