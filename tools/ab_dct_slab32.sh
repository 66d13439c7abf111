#!/bin/bash
# A/B on the GPU box: the transform kernel as built (row-pass results parked in LDS as f64: 36 KB per workgroup, 4 waves per
# SIMD) against -DSVC_DCT_SLAB32 (parked as 2^-21 fixed point in 32 bits: half the LDS, 8 waves per SIMD).
set -eu
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --steps 30 --warmup 5 --schedule "$1" --no-cpu-baseline --no-hbm-probe "${@:2}" | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['config']['workload'][:24], 'dct ms', round(d['kernel_ms_per_step']['dct_quant'],4), 'step', round(d['ms_per_step'],3))"; }
all() { for sch in serial pipelined; do run $sch; run $sch --config C5-4k-4L-dct16; run $sch --config C2-720p-3L-dct8; done; }
echo "== as built"; all
touch scalable_video_codec_amd/csrc/dct.hip
SVC_EXTRA_HIPCC_FLAGS=-DSVC_DCT_SLAB32 python3 -m scalable_video_codec_amd.build > /dev/null
echo "== SVC_DCT_SLAB32"; all
python3 -m pytest tests/test_gpu_dct_quant.py -m gpu -q 2>&1 | tail -5
