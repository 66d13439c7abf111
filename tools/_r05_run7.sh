set -o pipefail
mkdir -p gpurun_out/r05g
run() {
for rep in 1 2 3; do
for cfg in C3-1080p-3L-dct8-quant C5-4k-4L-dct16; do
for sched in serial pipelined; do
  python3 bench.py --config $cfg --schedule $sched --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 2>> gpurun_out/r05g/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['config']['workload'][:4], '$sched', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items() if k in ('luma_pyramid','hbma','dct_quant')})" >> gpurun_out/r05g/ab.txt
done; done; done
}
run registers
touch scalable_video_codec_amd/csrc/luma_pyramid.hip
SVC_EXTRA_HIPCC_FLAGS="-DSVC_LUMA_DMA" python3 -c "from scalable_video_codec_amd import build as b; b.build_hip(verbose=True)" >> gpurun_out/r05g/build.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_ransac_pyramid.py tests/test_gpu_golden.py tests/test_gpu_clip.py tests/test_gpu_misc_property.py -m gpu -x -q > gpurun_out/r05g/tests_dma.log 2>&1; echo "tests rc $?" >> gpurun_out/r05g/tests_dma.log
tail -3 gpurun_out/r05g/tests_dma.log
run lds_dma
cat gpurun_out/r05g/ab.txt
