#!/bin/bash
# Regenerates the round's end-state evidence on the GPU box into gpurun_out/final/ (copied to profiles/ by hand):
# default bench line (with the CPU leg and the end-to-end object), the other configs, the --wire line, rocprofv3 kernel stats
# of the pipelined and the serial schedule (C3, and C3b / C5 serial), the counter traffic of the shipped kernels (with the source
# hashes bench.py ties it to), the shard-size table behind DESIGN.md section 6.
# usage: tools/final_regen.sh <round tag, e.g. r05> [part ...]   parts: bench prof pmc shards (default: all)
set -u
tag=${1:-r06}; shift || true
parts=${*:-bench prof pmc shards}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
F=gpurun_out/final
for part in $parts; do case $part in
bench)
  python3 bench.py > $F/${tag}_z_final_bench.json 2> $F/bench.err; echo "default bench rc=$?"
  python3 bench.py --two-bgr-passes --no-cpu-baseline --no-end-to-end > $F/${tag}_bench_C3_two_passes.json 2>>$F/bench.err; echo "two passes rc=$?"
  for c in C2-720p-3L-dct8 C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do python3 bench.py --config $c --no-cpu-baseline --no-end-to-end > $F/${tag}_bench_$c.json 2>>$F/bench.err; echo "$c rc=$?"; done
  python3 bench.py --wire --no-cpu-baseline --no-end-to-end > $F/${tag}_bench_C3_wire.json 2>>$F/bench.err; echo "wire rc=$?"
  python3 bench.py --config C3b-1080p-4L-dct8-quant --no-end-to-end > $F/${tag}_bench_C3b_with_cpu.json 2>>$F/bench.err; echo "C3b+cpu rc=$?"
  # round 6: two chunks per step (--chunk-pairs 150) beside the default (one chunk: whole-shard launches)
  python3 bench.py --chunk-pairs 150 --no-cpu-baseline --no-end-to-end > $F/${tag}_bench_C3_two_chunks.json 2>>$F/bench.err; echo "two chunks rc=$?" ;;
prof)
  bash tools/prof_bench.sh $F/${tag}_z_final_pipelined_profiled --steps 20 --warmup 8 --sustain-seconds 0 > /dev/null; echo prof1 done
  bash tools/prof_bench.sh $F/${tag}_z_final_serial --steps 20 --warmup 8 --schedule serial --sustain-seconds 0 > /dev/null; echo prof2 done
  bash tools/prof_bench.sh $F/${tag}_z_final_two_passes_serial --steps 20 --warmup 8 --schedule serial --sustain-seconds 0 --two-bgr-passes > /dev/null; echo prof2b done
  bash tools/prof_bench.sh $F/${tag}_z_final_C3b_serial --steps 20 --warmup 8 --schedule serial --sustain-seconds 0 --config C3b-1080p-4L-dct8-quant > /dev/null; echo prof3 done
  bash tools/prof_bench.sh $F/${tag}_z_final_C5_serial --steps 20 --warmup 8 --schedule serial --sustain-seconds 0 --config C5-4k-4L-dct16 > /dev/null; echo prof4 done
  bash tools/prof_bench.sh $F/${tag}_z_final_wire_serial --steps 20 --warmup 8 --schedule serial --sustain-seconds 0 --wire > /dev/null; echo prof5 done ;;
pmc)
  # groups 6 + 7: the L2's memory-side request counters by size.  --always-speculate: a 3-step counter run has no foreground share yet
  PMC_GROUPS="6 7" bash tools/pmc_passes.sh $F/pmc_C3 --always-speculate > $F/pmc_C3.log 2>&1; cp $F/pmc_C3/summary.csv $F/${tag}_pmc_C3_traffic_summary.csv; cp $F/pmc_C3/summary.csv.meta.json $F/${tag}_pmc_C3_traffic_summary.csv.meta.json
  PMC_GROUPS="6 7" bash tools/pmc_passes.sh $F/pmc_C3_two --two-bgr-passes > $F/pmc_C3_two.log 2>&1; cp $F/pmc_C3_two/summary.csv $F/${tag}_pmc_C3_two_passes_traffic_summary.csv; cp $F/pmc_C3_two/summary.csv.meta.json $F/${tag}_pmc_C3_two_passes_traffic_summary.csv.meta.json
  PMC_GROUPS="6 7" bash tools/pmc_passes.sh $F/pmc_C3_wire --wire > $F/pmc_C3_wire.log 2>&1; cp $F/pmc_C3_wire/summary.csv $F/${tag}_pmc_C3_wire_traffic_summary.csv; cp $F/pmc_C3_wire/summary.csv.meta.json $F/${tag}_pmc_C3_wire_traffic_summary.csv.meta.json
  PMC_GROUPS="6 7" bash tools/pmc_passes.sh $F/pmc_C3b --config C3b-1080p-4L-dct8-quant > $F/pmc_C3b.log 2>&1; cp $F/pmc_C3b/summary.csv $F/${tag}_pmc_C3b_traffic_summary.csv; cp $F/pmc_C3b/summary.csv.meta.json $F/${tag}_pmc_C3b_traffic_summary.csv.meta.json
  PMC_GROUPS="6 7" bash tools/pmc_passes.sh $F/pmc_C5 --config C5-4k-4L-dct16 > $F/pmc_C5.log 2>&1; cp $F/pmc_C5/summary.csv $F/${tag}_pmc_C5_traffic_summary.csv; cp $F/pmc_C5/summary.csv.meta.json $F/${tag}_pmc_C5_traffic_summary.csv.meta.json
  PMC_GROUPS="6 7" bash tools/pmc_passes.sh $F/pmc_C2 --config C2-720p-3L-dct8 > $F/pmc_C2.log 2>&1; cp $F/pmc_C2/summary.csv $F/${tag}_pmc_C2_traffic_summary.csv; cp $F/pmc_C2/summary.csv.meta.json $F/${tag}_pmc_C2_traffic_summary.csv.meta.json
  echo pmc done ;;
pmcpyr)
  # round 6: SQ / TCP counters of the plane-to-plane pyramid pass before (LDS-tiled, variant library pyr_tiled) and after (pyr_strip_kernel)
  PMC_KERNEL_FILTER="pyr_strip\|luma_pyr1_kernel<false" bash tools/pmc_variants.sh $F/pmc_pyr "3 4 9" pyr_tiled asbuilt -- --always-speculate > $F/pmc_pyr.txt 2>&1
  cp $F/pmc_pyr/pyr_tiled_summary.csv $F/${tag}_pmc_pyr_tiled_summary.csv; cp $F/pmc_pyr/asbuilt_summary.csv $F/${tag}_pmc_pyr_strip_summary.csv; echo pmcpyr done ;;
shards)
  bash tools/shard_sizes.sh > $F/${tag}_shard_rows.jsonl 2>>$F/bench.err; echo shard sizes done ;;
esac; done
ls -la $F
