#!/bin/bash
# Same-box A/B: RANSAC's in-order RMSE sum inside its kernel (--inline-rmse) vs as a launch of its own beside the
# segmentation (default in the pipelined schedule).
row() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); o=d.get('overlapped_ms_per_step',{})
print('  ms/step %.3f  main %s  ransac %.3f segment %.3f' % (d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, o.get('ransac',0), o.get('segment',0)))"; }
for rep in 1 2; do
for a in "--config C5-4k-4L-dct16 --frames 8" "--config C5-4k-4L-dct16 --frames 16" "--config C5-4k-4L-dct16" "--frames 38" "--frames 75" "" "--config C3b-1080p-4L-dct8-quant"; do
  echo "== $a inline"; row $a --inline-rmse
  echo "== $a deferred"; row $a
done
done
