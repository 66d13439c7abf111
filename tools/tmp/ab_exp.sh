#!/bin/bash
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-probe "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']
print('  value %8.0f  ms/step %.3f  hbma %.4f ms  frac %.3f   %s' % (d['value'], d['ms_per_step'], k.get('hbma',0), d['roofline']['frac'], d['roofline']['kernel'][:24]))"; }
for fl in "-DSVC_TILED_EXP=1" "-DSVC_TILED_EXP=2" "-DSVC_TILED_EXP=3"; do
  SVC_EXTRA_HIPCC_FLAGS="$fl" python -c "
import sys; sys.path.insert(0,'.')
from scalable_video_codec_amd import build; build.build_hip(force=True); build.build_motion(force=True)" 2>&1 | grep -v warning | tail -2
  for c in C5-4k-4L-dct16 C3b-1080p-4L-dct8-quant; do
    echo "== $fl $c tiled serial"; run --config $c --hbma-kernel tiled --schedule serial
  done
done
