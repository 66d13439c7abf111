#!/usr/bin/env python3
"""tests/golden/hbma_shapes.npz: EstimateMotionHierarchical of the UNMODIFIED reference (oracle/_ref/libsvc_ref.so =
/root/reference/libs/motion.cpp compiled in place) for every (MV block, levels, search range) shape the lane-per-block GPU
kernel is instantiated for -- 8x8 / 16x16 / 32x32 blocks, 2 .. log2(block) levels, R_top 1 .. 4 (apps/encoder.cpp:75-104).
Inputs are regenerated from the seed by tests/golden_util.py::shape_case and pinned by sha256; the fixture holds the
reference's MVs and min-MADs only.  Run where /root/reference exists:

    make -C oracle all && python tests/golden/make_hbma_shapes_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.binding import Reference  # noqa: E402
from tests.golden_util import SHAPE_CASES, sha, shape_case  # noqa: E402


def main():
    ref = Reference()
    out = {}
    for mb, levels, r in SHAPE_CASES:
        t, a = shape_case(mb, levels, r)
        mv, mad = ref.hbma(t, a, r, mb, mb)
        key = f"b{mb}_l{levels}_r{r}"
        out[f"{key}/mv"], out[f"{key}/mad"] = mv, mad
        out[f"{key}/sha"] = np.frombuffer(bytes.fromhex(sha(t + a)), np.uint8)
        print(key, mv.shape, "distinct MVs", len(np.unique(mv, axis=0)), "max |mv|", float(np.abs(mv).max()))
    np.savez_compressed(os.path.join(HERE, "hbma_shapes.npz"), **out)


if __name__ == "__main__":
    main()
