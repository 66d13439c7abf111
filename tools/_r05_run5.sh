set -o pipefail
mkdir -p gpurun_out/r05e
timeout -k 10 600 python -m pytest tests/test_gpu_ransac_pyramid.py tests/test_gpu_wire.py tests/test_gpu_clip.py tests/test_gpu_imageops.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r05e/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r05e/tests.log
tail -3 gpurun_out/r05e/tests.log
run() {  # label
for rep in 1 2; do
for cfg in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do
for mode in "--wire" ""; do
  python3 bench.py --config $cfg $mode --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 2>> gpurun_out/r05e/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['config']['workload'][:4], '$mode', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/r05e/ab.txt
done; done; done
}
run rpt4
for r in 2 1; do
touch scalable_video_codec_amd/csrc/luma_pyramid.hip
SVC_EXTRA_HIPCC_FLAGS="-DSVC_PLANE_RPT=$r" python3 -c "from scalable_video_codec_amd import build as b; b.build_hip(verbose=True)" >> gpurun_out/r05e/build.log 2>&1
run rpt$r
done
cat gpurun_out/r05e/ab.txt
