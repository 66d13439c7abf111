#!/bin/bash
# Same-box A/B: a heavy frame's k-means attempts as one workgroup each (--narrow-attempts) vs as launch sequences over
# several workgroups (default when frames x attempts <= CUs and the sequence is at most ~32 launches: small shards of large
# fields AND C5's whole 64-frame clip).  C5 shards of an 8 / 4-GPU run, and the whole clip.
row() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); o=d.get('overlapped_ms_per_step',{})
print('  ms/step %.3f  main %s  ransac %.3f segment %.3f' % (d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, o.get('ransac',0), o.get('segment',0)))"; }
for rep in 1 2; do
for f in 8 16 64; do
  echo "== C5 frames=$f narrow"; row --config C5-4k-4L-dct16 --frames $f --narrow-attempts
  echo "== C5 frames=$f default"; row --config C5-4k-4L-dct16 --frames $f
done
done
echo "== serial, frames=8"; for v in "--narrow-attempts" ""; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 --config C5-4k-4L-dct16 --frames 8 --schedule serial $v 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('  ', '$v', 'ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; done
