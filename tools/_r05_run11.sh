set -o pipefail
mkdir -p gpurun_out/r05l
timeout -k 10 600 python -m pytest tests/test_gpu_ransac_pyramid.py tests/test_gpu_golden.py tests/test_gpu_wire.py tests/test_gpu_imageops.py tests/test_gpu_misc_property.py -m gpu -x -q > gpurun_out/r05l/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r05l/tests.log
tail -3 gpurun_out/r05l/tests.log
run() {
for rep in 1 2; do
for cfg in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do
for mode in "--wire" "--two-bgr-passes"; do
  python3 bench.py --config $cfg $mode --schedule serial --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 2>> gpurun_out/r05l/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['config']['workload'][:4], '$mode', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items() if k in ('luma_pyramid','dct_quant')})" >> gpurun_out/r05l/ab.txt
done; done; done
}
run grid1536
for g in 2048 1024; do
touch scalable_video_codec_amd/csrc/luma_pyramid.hip
SVC_EXTRA_HIPCC_FLAGS="-DSVC_PLANE_GRID=$g" python3 -c "from scalable_video_codec_amd import build as b; b.build_hip(verbose=True)" >> gpurun_out/r05l/build.log 2>&1
run grid$g
done
cat gpurun_out/r05l/ab.txt
