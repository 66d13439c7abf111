"""Live comparison of the C restatement with the unmodified reference (oracle/_ref,
built from /root/reference by oracle/Makefile).  Skipped where _ref was never built."""
import numpy as np
import pytest

from oracle.binding import DEFAULT_RANSAC
from tests import util


@pytest.mark.parametrize("levels,r,bw,bh", [(1, 8, 16, 16), (2, 8, 16, 16), (3, 8, 16, 16), (4, 8, 16, 16),
                                            (3, 5, 8, 8), (2, 7, 16, 8), (1, 3, 6, 10), (3, 12, 32, 16)])
def test_hbma_random(oracle, reference, levels, r, bw, bh):
    rng = np.random.default_rng(levels * 31 + r)
    f = 1 << (levels - 1)
    w, h = bw * 7 * f // f * f, bh * 5 * f // f * f
    w, h = (w // (bw * f) + 1) * bw * f, (h // (bh * f) + 1) * bh * f
    t, a = util.random_planes(rng, w, h, levels), util.random_planes(rng, w, h, levels)
    mv_o, mad_o = oracle.hbma(t, a, r, bw, bh)
    mv_r, mad_r = reference.hbma(t, a, r, bw, bh)
    assert np.array_equal(mv_o, mv_r) and np.array_equal(mad_o, mad_r)


def test_hbma_synthetic_clip_and_sse2(oracle, reference):
    _, pyrs, _ = util.clip_frames(352, 288, 3, 77, 4)
    for i in range(2):
        t, a = util.np_pyr(pyrs[i]), util.np_pyr(pyrs[i + 1])
        mv_r, mad_r = reference.hbma(t, a, 8, 16, 16)
        mv_s, mad_s = reference.hbma16_sse2(t, a, 8)
        mv_o, mad_o = oracle.hbma16_sse2(t, a, 8)
        assert np.array_equal(mv_r, mv_s) and np.array_equal(mad_r, mad_s)  # SSE2 == generic L=4
        assert np.array_equal(mv_o, mv_s) and np.array_equal(mad_o, mad_s)


def test_ebma(oracle, reference):
    rng = np.random.default_rng(3)
    t = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    a = np.roll(t, (1, -2), (0, 1))
    for r, bw, bh in ((8, 16, 16), (2, 4, 4), (20, 8, 8)):
        mv_o, mad_o = oracle.ebma(t, a, r, bw, bh)
        mv_r, mad_r = reference.ebma(t, a, r, bw, bh)
        assert np.array_equal(mv_o, mv_r) and np.array_equal(mad_o, mad_r)


def test_ransac_lockstep(oracle, reference):
    """Several calls in a row: the mirrored engine stays in step with the reference's static one."""
    rng = np.random.default_rng(11)
    for trial in range(6):
        n = 500 + 37 * trial
        mv = np.tile(np.array([[2.0, 1.0]], np.float32), (n + 1, 1))
        mv[rng.choice(n, n // 4, replace=False)] += rng.integers(-12, 13, (n // 4, 2)).astype(np.float32)
        p = dict(DEFAULT_RANSAC, subset_sz=1 + trial % 4)
        k = oracle.ransac_iter_count(**p)
        gm_r, rmse_r, inl_r = reference.ransac(mv, n, **p)
        s = reference.ransac_draw(n, p["subset_sz"], k)
        gm_o, rmse_o, inl_o = oracle.ransac(mv, s, n=n, **p)
        assert gm_o.tobytes() == gm_r.tobytes() and rmse_o.tobytes() == rmse_r.tobytes()
        assert np.array_equal(inl_o, inl_r)


def test_global_motion_functions(oracle, reference):
    """libs/motion.hpp:38-59.  The running mean is pinned bit for bit.  The exhaustive search is pinned in the
    reference's LITERAL form: its `int dy <= unsigned search_range` loops (libs/motion.cpp:72, :81) never run for
    R > 0, so the unmodified reference answers {0, 0}, FLT_MAX for every input -- shown here on frames with an
    obvious (+3, -2) shift -- and only R = 0 computes anything (the zero-displacement whole-frame MAD)."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 396, 8160):
        mv = (rng.integers(-16, 17, (n, 2)) + rng.random((n, 2)) * (n % 3)).astype(np.float32)
        assert oracle.global_avg(mv).tobytes() == reference.global_avg(mv).tobytes()
    base = rng.integers(0, 256, (80, 120), dtype=np.uint8)
    t, a = np.ascontiguousarray(base[8:72, 8:104]), np.ascontiguousarray(base[10:74, 5:101])
    for r in (0, 1, 4, 8):
        gm_r, mad_r = reference.global_ebma(t, a, r)
        gm_o, mad_o = oracle.global_ebma(t, a, r, reference_loop=True)
        assert gm_o.tobytes() == gm_r.tobytes() and mad_o.tobytes() == mad_r.tobytes(), r
        if r > 0:
            assert gm_r.tolist() == [0.0, 0.0] and mad_r == np.finfo(np.float32).max  # the reference's loops never run
    # the search as evidently meant finds the planted shift: tracked(y, x) = anchor(y - dy, x - dx)
    gm, mad = oracle.global_ebma(t, a, 4)
    assert gm.tolist() == [-3.0, 2.0] and mad == 0.0
    pyr_t = [np.ascontiguousarray(t[:: 1 << l, :: 1 << l]) for l in range(3)]
    pyr_a = [np.ascontiguousarray(a[:: 1 << l, :: 1 << l]) for l in range(3)]
    assert reference.global_hbma(pyr_t, pyr_a, 8).tobytes() == oracle.global_hbma(pyr_t, pyr_a, 8, reference_loop=True).tobytes()
    assert reference.global_hbma(pyr_t, pyr_a, 3).tobytes() == oracle.global_hbma(pyr_t, pyr_a, 3, reference_loop=True).tobytes()


def test_random_search_configurations_against_the_unmodified_reference(oracle, reference):
    """Seeded random block shapes (square and not), level counts 1 ... 5, search ranges (R_top 1 ... 9), field sizes, shifted noise and
    periodic texture (exact ties: the last minimum wins at the top level, the first at every refinement): the restatement == the
    unmodified reference, MVs and min-MADs bit for bit.  (The GPU suite sweeps the kernels against the restatement the same way.)"""
    rng = np.random.default_rng(2024)
    done = 0
    while done < 60:
        levels = int(rng.integers(1, 6))
        f = 1 << (levels - 1)
        bw, bh = f * int(rng.integers(1, max(2, 40 // f))), f * int(rng.integers(1, max(2, 40 // f)))
        if rng.random() < 0.5:
            bh = bw
        if bw * bh > 256 * 16:
            continue
        r = f * int(rng.integers(1, 10))
        if r > 48:
            continue
        w, h = bw * int(rng.integers(1, 8)), bh * int(rng.integers(1, 6))
        if w * h > 320 * 240:
            continue
        if rng.random() < 0.5:
            base = rng.integers(0, 256, (h + 64, w + 64), dtype=np.uint8)
            dx, dy = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
            a0, b0 = base[32:32 + h, 32:32 + w], base[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]
        else:
            period = int(rng.choice([2, 4, 8]))
            yy, xx = np.mgrid[0:h, 0:w]
            a0 = (((xx // period) + (yy // period)) % 2 * 200 + 20).astype(np.uint8)
            b0 = np.roll(a0, int(rng.integers(0, period)), axis=1)
        pa, pb = [np.ascontiguousarray(a0)], [np.ascontiguousarray(b0)]
        ok = True
        for _ in range(1, levels):
            if min(pa[-1].shape) < 3:
                ok = False
                break
            pa.append(oracle.pyr_down(pa[-1]))
            pb.append(oracle.pyr_down(pb[-1]))
        if not ok:
            continue
        mv_o, mad_o = oracle.hbma(pa, pb, r, bw, bh)
        mv_r, mad_r = reference.hbma(pa, pb, r, bw, bh)
        assert np.array_equal(mv_o, mv_r) and np.array_equal(mad_o, mad_r), (levels, bw, bh, r, w, h)
        done += 1


def test_random_ransac_parameters_against_the_unmodified_reference(oracle, reference):
    """Seeded random RansacParams and fields through the reference's EstimateGlobalMotionRansac (its own engine, mirrored draw for draw)
    and the restatement fed those draws: global motion, RMSE, inlier list bit for bit."""
    rng = np.random.default_rng(2025)
    done = 0
    while done < 40:
        n = int(rng.choice([30, 396, 1500, 8160]))
        p = dict(subset_sz=int(rng.integers(1, 6)), inlier_thresh=float(rng.choice([0.4, 1.5, 7.5, 40.0])),
                 success_prob=float(rng.choice([0.5, 0.9, 0.99, 0.999])), inlier_ratio=float(rng.choice([0.2, 0.5, 0.8, 0.95])))
        k = oracle.ransac_iter_count(**p)
        if not 1 <= k <= 300:
            continue
        mv = np.zeros((n + 1, 2), np.float32)  # n + 1: the reference draws from [0, n] (libs/motion.cpp:208)
        mv[:] = rng.integers(-6, 7, 2)
        cut = int(n * rng.random())
        mv[cut:] = rng.integers(-14, 15, (n + 1 - cut, 2))
        if rng.random() < 0.3:
            mv += (rng.random((n + 1, 2)) * 0.5).astype(np.float32)
        gm_r, rmse_r, inl_r = reference.ransac(mv, n, **p)
        s = reference.ransac_draw(n, p["subset_sz"], k)
        if (np.asarray(s) >= n).any():
            continue  # a draw of index n: the reference read one vector past its field; the restatement takes indices inside it only
        gm_o, rmse_o, inl_o = oracle.ransac(mv, s, n=n, **p)
        assert gm_o.tobytes() == gm_r.tobytes() and rmse_o.tobytes() == rmse_r.tobytes() and np.array_equal(inl_o, inl_r), (n, p)
        done += 1
