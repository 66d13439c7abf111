set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/r02_z_final_bench.json 2> gpurun_out/final/bench.err; echo "default bench rc=$?"
for c in C2-720p-3L-dct8 C3b-1080p-4L-dct8-quant C5-4k-4L-dct16; do python3 bench.py --config $c --no-cpu-baseline > gpurun_out/final/r02_bench_$c.json 2>>gpurun_out/final/bench.err; echo "$c rc=$?"; done
bash tools/prof_bench.sh gpurun_out/final/r02_z_final_pipelined_profiled --steps 20 --warmup 5 > /dev/null; echo prof1 done
bash tools/prof_bench.sh gpurun_out/final/r02_z_final_serial --steps 20 --warmup 5 --schedule serial > /dev/null; echo prof2 done
ls -la gpurun_out/final
