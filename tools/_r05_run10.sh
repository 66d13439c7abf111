set -o pipefail
mkdir -p gpurun_out/r05k
for rep in 1 2; do
for frames in 150 75 38 19; do
for mode in "--always-speculate" "--two-bgr-passes" "--wire" "--wire --two-bgr-passes"; do
  python3 bench.py --frames $frames $mode --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 2>> gpurun_out/r05k/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C3 frames $frames', '$mode', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/r05k/ab.txt
done; done; done
for rep in 1 2; do
for mode in "--always-speculate" "--two-bgr-passes" "--wire" "--wire --two-bgr-passes"; do
  python3 bench.py --config C2-720p-3L-dct8 $mode --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe --sustain-seconds 0 2>> gpurun_out/r05k/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C2', '$mode', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/r05k/ab.txt
done; done
cat gpurun_out/r05k/ab.txt
