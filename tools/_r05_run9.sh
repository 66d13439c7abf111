set -o pipefail
mkdir -p gpurun_out/r05j
timeout -k 10 900 python -m pytest tests/test_gpu_dct_quant.py tests/test_gpu_clip.py tests/test_gpu_wire.py tests/test_gpu_fullsize.py tests/test_gpu_bench_contract.py -m gpu -x -q > gpurun_out/r05j/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r05j/tests.log
tail -4 gpurun_out/r05j/tests.log
for rep in 1 2; do
for cfg in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16 C2-720p-3L-dct8; do
for mode in "" "--two-bgr-passes" "--always-speculate"; do
  python3 bench.py --config $cfg $mode --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 2>> gpurun_out/r05j/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['workload'][:4], '$mode', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, round(d.get('overlapped_ms_per_step',{}).get('type_patch',0),4), 'fg', round(d['config']['foreground_mv_blocks'],4), d['config']['bgr_passes_per_step'][:3], d['config']['bgr_passes_per_step'][-14:])" >> gpurun_out/r05j/ab.txt
done; done; done
cat gpurun_out/r05j/ab.txt
