#!/bin/bash
# round 6: RANSAC + segmentation of the previous step forked BEHIND the front-of-step transform (--fork-behind-front: beside the pyramid pass and the
# motion search) instead of in front of it (beside the store-bound transform's first third).  Same box, three repetitions, the one-pass orders.
cd "$GRAFT_REPO_ROOT"
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; o=d["overlapped_ms_per_step"]
print("  %-16s ms/step %.3f  transform %.3f  pyramid %.3f  search %.3f | beside: ransac %.3f segment %.3f | stream of clips %.3f" % (d["config"]["workload"][:16], d["ms_per_step"], k["dct_quant"], k["luma_pyramid"], k["hbma"], o["ransac"], o["segment"], d["first_encode"]["stream_of_clips"]["ms_per_clip"]))'
for r in 1 2 3; do for a in "" "--wire" "--config C5-4k-4L-dct16 --always-speculate"; do for f in "" "--fork-behind-front"; do
  echo "== [$a] ${f:-as built}"
  python3 bench.py $a $f --steps 30 --no-cpu-baseline --no-end-to-end --sustain-seconds 0 --no-hbm-probe --first-encode-reps 1 2>/dev/null | python3 -c "$pick"
done; done; done
