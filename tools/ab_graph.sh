#!/bin/bash
# HISTORICAL (round 4): the script behind profiles/r04_ab_graph.txt, after which the --graph option and the capture / replay
# branch of svc::ClipEncoder were removed (the graph form lost to the eager pipelined schedule on every workload below; it needs
# a build from before that commit).
# Same-box A/B of --graph (the steady-state iteration replayed from a hipGraph) on the launch-bound workloads: C2 (30 frames
# of 720p: nine launches in 0.16 ms), tiny 1080p shards, C1 (CIF).  Three rounds measured "no gain" on the large clips; this
# asks the question where a graph could matter at all.
cd "$GRAFT_REPO_ROOT"
row() { python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('  ms/step %.4f  frames/s %.0f' % (d['ms_per_step'], d['value']))"; }
for rep in 1 2; do
  for cfg in "--config C2-720p-3L-dct8" "--frames 19" "--frames 8" "--config C2-720p-3L-dct8 --frames 8"; do
    echo "== $cfg eager (pipelined)"; row $cfg
    echo "== $cfg --graph"; row $cfg --graph
    echo "== $cfg serial"; row $cfg --schedule serial
  done
done
