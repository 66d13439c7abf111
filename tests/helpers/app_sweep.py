#!/usr/bin/env python3
"""Discovery tool: random command lines the reference's Validate admits, through the reference's UNCHANGED apps/encoder.cpp with this
repo's class Encoder (tests/dropin/ref_app_svc_encoder_generic), the stream on stdout against the oracle run stage by stage with the same
options (tests/test_gpu_encoder_class.py::_expected).  Random frame sizes, level counts, square and non-square MV / transform blocks,
search ranges, RANSAC and segmentation parameters, clip lengths around the batch size.

usage (GPU box): python tests/helpers/app_sweep.py [--count 40] [--seed 1]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402
from scalable_video_codec_amd import synth  # noqa: E402
from tests import test_gpu_encoder_class as ec  # noqa: E402
from tests import test_gpu_ref_encoder as rc  # noqa: E402
from tests.test_gpu_ref_encoder import _check, _write_clip  # noqa: E402


def random_case(rng):
    levels = int(rng.integers(1, 5))
    f = 1 << (levels - 1)
    sides = [b for b in (8, 16, 32) if b >= 2 * f]
    bw, bh = int(rng.choice(sides)), int(rng.choice(sides))
    if rng.random() < 0.6:
        bh = bw
    tws = [t for t in (2, 4, 8, 16, 32) if bw % t == 0]
    ths = [t for t in (2, 4, 8, 16, 32) if bh % t == 0]
    tw, th = int(rng.choice(tws)), int(rng.choice(ths))
    if rng.random() < 0.6 and tw in ths:
        th = tw
    # the reference's serialiser swaps the transform block's sides in its loops (libs/encoder.cpp:230-262): with non-square tiles its
    # reads must stay inside the planes, which holds when the taller-than-wide excess is small; keep to shapes the reference itself survives
    if tw > th and tw - th > 8 and tw in ths:
        th = tw
    search = int(rng.choice([r for r in (2, 4, 8, 8, 16) if r >= f]))
    w = int(rng.integers(2 * bw + 1, 400))
    h = int(rng.integers(2 * bh + 1, 300))
    n = int(rng.choice([2, 3, 5, 17, 18]))
    opts = ["--pyr-lvl-count", str(levels), "--mv-block-w", str(bw), "--mv-block-h", str(bh), "--transform-block-w", str(tw),
            "--transform-block-h", str(th), "--mv-search-range", str(search)]
    kw = dict(levels=levels, tb=(tw, th), mv_block=(bw, bh), search_range=search)
    if rng.random() < 0.5:
        sub = int(rng.integers(1, 4))
        thr = float(rng.choice([1.5, 2.5, 7.5]))
        opts += ["--ransac-subset-sz", str(sub), "--ransac-inlier-thresh", str(thr)]
        kw["ransac"] = dict(subset_sz=sub, inlier_thresh=thr)
    if rng.random() < 0.5:
        seg = dict(morph_w=int(rng.integers(1, 6)), morph_h=int(rng.integers(1, 6)), cluster_count=int(rng.integers(1, 12)),
                   attempts=int(rng.integers(1, 6)), max_iter=int(rng.integers(1, 12)), connectivity=int(rng.choice([4, 8])))
        opts += ["--morph-rect-w", str(seg["morph_w"]), "--morph-rect-h", str(seg["morph_h"]), "--kmeans-cluster-count", str(seg["cluster_count"]),
                 "--kmeans-attempt-count", str(seg["attempts"]), "--kmeans-max-iter-count", str(seg["max_iter"]),
                 "--connected-components-connectivity", str(seg["connectivity"])]
        kw["seg"] = seg
    return (w, h, n), opts, kw


def reference_reads_inside(w, h, levels, mv_block, tb):
    """SerializeEncodedFrame (libs/encoder.cpp:241-266) is handed the UNPADDED size, walks transform_block_w ROWS of transform_block_h
    floats per tile and channel, with the unpadded width as the row stride: do its reads stay inside a padded plane?"""
    (bw, bh), (tw, th) = mv_block, tb
    pw, ph = synth.padded_dims(w, h, bw, bh, levels)
    last_x, last_y = (w - 1) // tw * tw, (h - 1) // th * th
    return (last_y + tw - 1) * w + last_x + th <= pw * ph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--adapter", action="store_true", help="the reference's own libs/encoder.cpp on compat/opencv2 (tests/dropin/ref_encoder_generic) "
                                                          "instead of class Encoder: its draws and seeds are the reference's own generators'")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    oracle = binding.Oracle()
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(args.count):
            (w, h, n), opts, kw = random_case(rng)
            t0 = time.perf_counter()
            name = f"{w}x{h} n={n} {' '.join(opts)}"
            inside = reference_reads_inside(w, h, kw["levels"], kw["mv_block"], kw["tb"])
            if not inside and args.adapter:
                # the reference's own serialiser then reads heap beyond its matrices: whatever it emits is not a function of the clip
                print(f"SKIP (the reference reads past its planes here) {name}", flush=True)
                continue
            try:
                clip = synth.SynthClip(w, h, n, seed=1000 + i)
                frames = [clip.frame_bgr(t).numpy() for t in range(n)]
                path = os.path.join(tmp, "clip.svcbgr")
                _write_clip(path, frames)
                levels, tb = kw.pop("levels"), kw.pop("tb")
                if args.adapter:
                    got, _, _ = rc._encode("ref_encoder_generic", path, "--verbose", "0", *opts)
                    header, expected = rc._expected_stream(oracle, frames, levels, False, tb, search_range=kw["search_range"], mvb=kw["mv_block"],
                                                           seg=kw.get("seg"), ransac=kw.get("ransac"))
                else:
                    got, _ = ec._encode("ref_app_svc_encoder_generic", path, "--verbose", "0", *opts)
                    if not inside:  # must be refused: the oracle's restated loops would read past their planes too
                        raise AssertionError("the class Encoder emitted a stream for a configuration whose serialiser reads past its planes")
                    header, expected = ec._expected(oracle, frames, levels, tb, **kw)
                _check(got, header, expected, tb)
                verdict = None
            except BaseException as e:  # noqa: BLE001 (pytest.skip / AssertionError / CalledProcessError alike)
                verdict = f"{type(e).__name__}: {str(e)[:300]}"
            # non-square tiles wider than tall: the reference's serialiser walks transform_block_w ROWS per tile (libs/encoder.cpp:257), past
            # the padded plane's end on the last tile row -- it reads out of bounds there; this build refuses with a message instead
            refused = verdict is not None and "planes of" in verdict and "too small" in verdict and not inside
            bad += verdict is not None and not refused
            tag = "ok  " if verdict is None else "REFUSED (the reference reads past its planes here)" if refused else "FAIL"
            print(f"{tag} {name} {time.perf_counter() - t0:.1f}s {'' if refused else verdict or ''}", flush=True)
    print(f"{args.count - bad} of {args.count} command lines give the oracle's stream", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
