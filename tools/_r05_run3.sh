set -o pipefail
mkdir -p gpurun_out/r05c
timeout -k 10 900 python -m pytest tests/test_gpu_ransac_pyramid.py tests/test_gpu_wire.py tests/test_gpu_clip.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c5.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/r05c/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r05c/tests.log
tail -4 gpurun_out/r05c/tests.log
for rep in 1 2; do
for cfg in C3-1080p-3L-dct8-quant C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do
for mode in "--wire" "--wire --two-bgr-passes" ""; do
  python3 bench.py --config $cfg $mode --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 >> gpurun_out/r05c/ab.jsonl 2>> gpurun_out/r05c/ab.err
done; done; done
python3 - <<'P'
import json
for l in open('gpurun_out/r05c/ab.jsonl'):
    d=json.loads(l)
    print(d['config']['workload'][:4], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})
P
