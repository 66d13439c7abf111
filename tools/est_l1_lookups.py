#!/usr/bin/env python3
"""Would ordering a tile's lanes by their level-1 vector's row make the lanes of a wave share tracked cache lines at level 0 of the
4-level search?  Answered from a motion field the REFERENCE produced (tests/golden/hbma_C3b-*.npz, frame
pair 0 -> 1 of the 1080p clip) with the vector L1's measured cost model (profiles/r03_ubench_tcp.txt: a wave instruction costs ~0.55
cycles per distinct 64-byte segment it touches, 16 cycles at least): the tag lookups of level 0's first tracked load (16 bytes per lane
at the window origin, dword-aligned) per wave instruction, lanes dealt (a) as shipped -- a wave = 64 blocks of one block row -- and
(b) after sorting the 64 x 4 tile's 256 lanes by (window row, window column).

    python3 tools/est_l1_lookups.py [tests/golden/hbma_C3b-1080p-4L-dct8-quant.npz]
"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "hbma_C3b-1080p-4L-dct8-quant.npz")
z = np.load(path)
pw, ph = [int(v) for v in z["padded"]]
mfw, mfh = pw // 16, ph // 16
mv = z["mv"].astype(int)
mvx, mvy = mv[:, 0].reshape(mfh, mfw), mv[:, 1].reshape(mfh, mfw)
# the vector a block enters level 0 with is twice its level-1 vector: the final one rounded to even is within the +-1 refinement of it
mix, miy = 2 * np.round(mvx / 2).astype(int), 2 * np.round(mvy / 2).astype(int)


def lookups(lanes):
    segs = set()
    for r, a0 in lanes:
        segs.add((r, a0 // 64))
        segs.add((r, (a0 + 15) // 64))
    return len(segs)


shipped, ordered = [], []
for ty in range(0, mfh, 4):
    for tx in range(0, mfw, 64):
        tile = []
        for by in range(ty, min(ty + 4, mfh)):
            lanes = [(by * 16 + miy[by, bx] - 1, (bx * 16 + mix[by, bx] - 1) & ~3) for bx in range(tx, min(tx + 64, mfw))]
            tile += lanes
            shipped.append(lookups(lanes) * 64 / len(lanes))
        tile.sort()
        for i in range(0, len(tile), 64):
            w = tile[i:i + 64]
            ordered.append(lookups(w) * 64 / len(w))
print(f"{os.path.basename(path)}: {mfw} x {mfh} blocks; level-0 entry vectors, rows: {collections.Counter(miy.ravel()).most_common(6)}")
print(f"tag lookups per 64-lane tracked load: wave = one block row (shipped) {np.mean(shipped):.1f}; tile's lanes ordered by (row, column) {np.mean(ordered):.1f}; "
      "all lanes on one row would be 16-17")
print("reading: the lookups of an instruction are the distinct (row, 64-byte segment) pairs its lanes touch.  Re-dealing lanes to waves does not "
      "change which pairs the TILE touches, and lanes from other block rows bring rows of their own: nothing to gain.  What is redundant is the "
      "SAME pair looked up again at another step t by a lane whose window starts on another row -- removable only by staging level 0 "
      "(built twice in round 3, slower).")
