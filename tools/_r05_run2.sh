set -o pipefail
mkdir -p gpurun_out/r05b
timeout -k 10 900 python -m pytest tests/test_gpu_wire.py tests/test_gpu_clip.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r05b/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r05b/tests.log
tail -4 gpurun_out/r05b/tests.log
for rep in 1 2; do
for mode in "" "--two-bgr-passes"; do
for sched in pipelined serial; do
  python3 bench.py --wire $mode --schedule $sched --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 >> gpurun_out/r05b/ab_wire.jsonl 2>> gpurun_out/r05b/ab_wire.err
done; done; done
python3 - <<'P'
import json
for l in open('gpurun_out/r05b/ab_wire.jsonl'):
    d=json.loads(l)
    print(d['config']['schedule'][:12], round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})
P
