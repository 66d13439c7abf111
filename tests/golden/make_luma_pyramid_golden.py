"""Writes tests/golden/luma_pyramid.npz: inputs + expected outputs of the luma / pyrDown pre-step
(libs/encoder.cpp:468-470, OpenCV's 8-bit fixed-point definitions -- parity unpinned, OpenCV is absent).

The expected values come from an INDEPENDENT formulation (numpy integer arithmetic + scipy.ndimage.correlate1d
with mode="mirror", which is BORDER_REFLECT_101), not from oracle/ and not from the product package, so the
fixture pins both.  Re-running reproduces the file byte for byte (fixed seed)."""
import os

import numpy as np
from scipy.ndimage import correlate1d

HERE = os.path.dirname(os.path.abspath(__file__))


def luma(bgr):
    p = bgr.astype(np.int64)
    return ((p[..., 0] * 1868 + p[..., 1] * 9617 + p[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def pyr_down(plane):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = correlate1d(correlate1d(plane.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    return ((full[::2, ::2] + 128) >> 8).astype(np.uint8)


def main():
    rng = np.random.default_rng(0x50C0DEC)
    out = {}
    # known answers: pure primaries and grey levels (Y of pure B / G / R at 255: 29 / 150 / 76; white 255; black 0)
    known = np.zeros((6, 4, 4, 3), np.uint8)
    known[0, ..., 0] = 255
    known[1, ..., 1] = 255
    known[2, ..., 2] = 255
    known[3] = 255
    known[5] = 128
    out["known/bgr"] = known
    out["known/y"] = np.stack([luma(f) for f in known])
    assert [int(out["known/y"][i, 0, 0]) for i in range(6)] == [29, 150, 76, 255, 0, 128]
    # whole pyramids
    for name, (w, h, levels) in {"a": (48, 32, 3), "b": (20, 12, 3), "c": (64, 16, 4), "d": (16, 2, 2)}.items():
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out[f"pyr_{name}/bgr"] = bgr
        lvl = luma(bgr)
        for l in range(levels):
            out[f"pyr_{name}/level{l}"] = lvl
            lvl = pyr_down(lvl)
    # pyrDown alone: odd sizes ((n + 1) / 2 outputs), a constant plane (stays constant), a single row / column
    for name, (w, h) in {"odd": (7, 5), "row": (9, 1), "col": (1, 6), "two": (2, 2)}.items():
        p = rng.integers(0, 256, (h, w), dtype=np.uint8)
        out[f"down_{name}/in"], out[f"down_{name}/out"] = p, pyr_down(p)
    c = np.full((6, 10), 201, np.uint8)
    out["down_const/in"], out["down_const/out"] = c, pyr_down(c)
    assert (out["down_const/out"] == 201).all()
    np.savez_compressed(os.path.join(HERE, "luma_pyramid.npz"), **out)
    print("wrote luma_pyramid.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
