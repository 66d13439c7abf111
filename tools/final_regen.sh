#!/bin/bash
# Regenerates the round's end-state evidence on the GPU box into gpurun_out/final/ (copied to profiles/ by hand):
# default bench line (with the CPU leg), the other configs, rocprofv3 kernel stats of the pipelined and the serial
# schedule.  usage: tools/final_regen.sh <round tag, e.g. r03>
set -u
tag=${1:-r03}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/${tag}_z_final_bench.json 2> gpurun_out/final/bench.err; echo "default bench rc=$?"
for c in C2-720p-3L-dct8 C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do python3 bench.py --config $c --no-cpu-baseline > gpurun_out/final/${tag}_bench_$c.json 2>>gpurun_out/final/bench.err; echo "$c rc=$?"; done
bash tools/prof_bench.sh gpurun_out/final/${tag}_z_final_pipelined_profiled --steps 20 --warmup 5 --sustain-seconds 0 > /dev/null; echo prof1 done
bash tools/prof_bench.sh gpurun_out/final/${tag}_z_final_serial --steps 20 --warmup 5 --schedule serial --sustain-seconds 0 > /dev/null; echo prof2 done
ls -la gpurun_out/final
