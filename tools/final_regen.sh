#!/bin/bash
# Regenerates the round's end-state evidence on the GPU box into gpurun_out/final/ (copied to profiles/ by hand):
# default bench line (with the CPU leg and the end-to-end object), the other configs, the --wire line, rocprofv3 kernel stats
# of the pipelined and the serial schedule (C3, and C3b / C5 serial), the shard-size table behind DESIGN.md section 6.
# usage: tools/final_regen.sh <round tag, e.g. r04>
set -u
tag=${1:-r04}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/${tag}_z_final_bench.json 2> gpurun_out/final/bench.err; echo "default bench rc=$?"
for c in C2-720p-3L-dct8 C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do python3 bench.py --config $c --no-cpu-baseline --no-end-to-end > gpurun_out/final/${tag}_bench_$c.json 2>>gpurun_out/final/bench.err; echo "$c rc=$?"; done
python3 bench.py --wire --no-cpu-baseline --no-end-to-end > gpurun_out/final/${tag}_bench_C3_wire.json 2>>gpurun_out/final/bench.err; echo "wire rc=$?"
python3 bench.py --config C3b-1080p-4L-dct8-quant > gpurun_out/final/${tag}_bench_C3b_with_cpu.json 2>>gpurun_out/final/bench.err; echo "C3b+cpu rc=$?"
bash tools/prof_bench.sh gpurun_out/final/${tag}_z_final_pipelined_profiled --steps 20 --warmup 5 --sustain-seconds 0 > /dev/null; echo prof1 done
bash tools/prof_bench.sh gpurun_out/final/${tag}_z_final_serial --steps 20 --warmup 5 --schedule serial --sustain-seconds 0 > /dev/null; echo prof2 done
bash tools/prof_bench.sh gpurun_out/final/${tag}_z_final_C3b_serial --steps 20 --warmup 5 --schedule serial --sustain-seconds 0 --config C3b-1080p-4L-dct8-quant > /dev/null; echo prof3 done
bash tools/prof_bench.sh gpurun_out/final/${tag}_z_final_C5_serial --steps 20 --warmup 5 --schedule serial --sustain-seconds 0 --config C5-4k-4L-dct16 > /dev/null; echo prof4 done
bash tools/prof_bench.sh gpurun_out/final/${tag}_z_final_wire_serial --steps 20 --warmup 5 --schedule serial --sustain-seconds 0 --wire > /dev/null; echo prof5 done
bash tools/shard_sizes.sh > gpurun_out/final/${tag}_shard_rows.jsonl 2>>gpurun_out/final/bench.err; echo shard sizes done
ls -la gpurun_out/final
