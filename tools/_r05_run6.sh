set -o pipefail
mkdir -p gpurun_out/r05f
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05f/suite.log 2>&1; echo "suite rc $?" >> gpurun_out/r05f/suite.log
tail -3 gpurun_out/r05f/suite.log
for rep in 1 2 3; do
for mode in "--wire" "--wire --two-bgr-passes" ""; do
  python3 bench.py $mode --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 2>> gpurun_out/r05f/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pipelined', '$mode', round(d['ms_per_step'],4), round(d['value']), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, round(d['roofline_step']['frac'],3))" >> gpurun_out/r05f/ab.txt
done; done
cat gpurun_out/r05f/ab.txt
