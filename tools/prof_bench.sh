#!/bin/bash
# rocprofv3 kernel trace + stats of one bench.py run, condensed into <out>_kernel_stats.csv and the bench line.
# (--first-encode-reps 0: the once-through repetitions launch half-shard chunks, which would mix into the per-kernel averages that the
# bench line's avg_launch_ms -- whole-shard launches of the timed region -- is checked against)
# usage: tools/prof_bench.sh <out prefix under gpurun_out/> [bench args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$(dirname "$out")"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "${out}_raw" -- python3 bench.py --no-cpu-baseline --no-hbm-probe --no-end-to-end --first-encode-reps 0 "$@" > "${out}_bench.json" 2> "${out}_bench.err"
rc=$?
echo "rocprofv3 exit $rc"
python3 tools/summarize_rocprof.py "${out}_raw" "${out}_kernel_stats.csv" "bench.py $*" > /dev/null
rm -rf "${out}_raw"
head -30 "${out}_kernel_stats.csv"
