set -o pipefail
mkdir -p gpurun_out/r05h
timeout -k 10 900 python -m pytest tests/test_gpu_dct_quant.py tests/test_gpu_clip.py tests/test_gpu_wire.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r05h/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r05h/tests.log
tail -4 gpurun_out/r05h/tests.log
for rep in 1 2; do
for cfg in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do
for mode in "" "--two-bgr-passes"; do
for sched in pipelined serial; do
  python3 bench.py --config $cfg $mode --schedule $sched --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 2>> gpurun_out/r05h/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['workload'][:4], '$sched', '$mode', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items() if k not in ('ransac','segment')}, 'fg', round(d['config']['foreground_mv_blocks'],4))" >> gpurun_out/r05h/ab.txt
done; done; done; done
cat gpurun_out/r05h/ab.txt
