#!/bin/bash
# round 6: what the GPU's clocks, power and temperatures do while the step runs back to back (bench.py's sustained run, 8 s): is the run-to-run and
# box-to-box spread of the store-bound transform kernel a matter of power / thermal state?  rocm-smi sampled every 0.4 s beside the run.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
rocm-smi --showclocks --showpower --showtemp --showuse 2>&1 | grep -v "^=\|^$" | head -30 > gpurun_out/r06/probe_clocks_idle.txt
python3 bench.py --no-cpu-baseline --no-end-to-end --no-hbm-probe --first-encode-reps 0 --sustain-seconds 8 > gpurun_out/r06/probe_clocks_bench.json 2>/dev/null &
B=$!
for i in $(seq 1 40); do
  sleep 0.4
  echo "--- t=$(python3 -c "print(round($i*0.4,1))") s"
  rocm-smi --showclocks --showpower --showtemp --showuse 2>&1 | grep -i "sclk\|mclk\|fclk\|power\|temperature\|busy" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  kill -0 $B 2>/dev/null || break
done
wait $B
python3 -c "import json; d=json.loads(open('gpurun_out/r06/probe_clocks_bench.json').read().strip().splitlines()[-1]); print('bench: value %.1f k, sustained %.1f k' % (d['value']/1e3, d['sustained']['value']/1e3))"
