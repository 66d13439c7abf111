#!/bin/bash
# A/B on the GPU box: the transform kernel of the previous build (a copy at scalable_video_codec_amd/_ab_prev_libsvc_hip.so:
# f32 row butterflies, scalar quantiser) against the one as built (integer butterflies, packed-f32 quantiser).
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule "$1" --no-cpu-baseline --no-hbm-probe "${@:2}" | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['config']['workload'][:24], 'dct ms', round(d['kernel_ms_per_step']['dct_quant'],4), 'step', round(d['ms_per_step'],3))"; }
all() { for sch in serial pipelined; do run $sch; run $sch --config C5-4k-4L-dct16; run $sch --config C2-720p-3L-dct8; done; }
cp scalable_video_codec_amd/libsvc_hip.so /tmp/new.so
for rep in 1 2; do
  cp scalable_video_codec_amd/_ab_prev_libsvc_hip.so scalable_video_codec_amd/libsvc_hip.so; echo "== previous build"; all
  cp /tmp/new.so scalable_video_codec_amd/libsvc_hip.so; echo "== as built"; all
done
python3 -m pytest tests/test_gpu_dct_quant.py tests/test_gpu_golden.py tests/test_gpu_decode.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -3
