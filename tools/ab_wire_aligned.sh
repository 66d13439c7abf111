#!/bin/bash
# Same-box A/B (round 4): the record-emitting transform (bench.py --wire) with per-channel 16-byte stores at the records' own
# 4-byte-aligned addresses (the previous build, scalable_video_codec_amd/_ab_prev_libsvc_hip.so) against the build under test
# (as built).  Three results are kept under profiles/: r04_ab_wire_aligned_experiment.txt (768-byte pseudo-records: what
# line-aligned runs would cost), r04_ab_wire_aligned_chunks.txt (aligned 16-byte chunks + end dwords per run: slower) and
# r04_ab_wire_stretch.txt (whole records of a wave's segment columns staged in LDS and written as one aligned stretch: what
# ships).  Serial schedule: the event time is the kernel alone.  Same bytes (tests/test_gpu_wire.py).
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --wire --steps 30 --warmup 5 --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(' ', d['config']['workload'][:24], 'transform ms', round(d['kernel_ms_per_step']['dct_quant'],4), 'step', round(d['ms_per_step'],3))"; }
pip() { python3 bench.py --wire --steps 30 --warmup 5 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(' ', d['config']['workload'][:24], 'pipelined: transform ms', round(d['kernel_ms_per_step']['dct_quant'],4), 'step', round(d['ms_per_step'],3), 'frames/s', round(d['value']))"; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/asbuilt.so
for rep in 1 2; do
  cp scalable_video_codec_amd/_ab_prev_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== previous build (stores at the records' own alignment)"; run; run --config C5-4k-4L-dct16; run --config C2-720p-3L-dct8; pip
  cp /tmp/asbuilt.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built"; run; run --config C5-4k-4L-dct16; run --config C2-720p-3L-dct8; pip
done
