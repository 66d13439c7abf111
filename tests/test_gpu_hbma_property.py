"""Randomised shapes for the motion search (hypothesis): any block size, search range, level
count and frame size the reference accepts must give the oracle's MVs and min-MADs bit for bit,
through whichever kernel the dispatcher picks and through the forced general kernel."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests.util import FUZZ_RANDOM, fuzz_examples

pytestmark = pytest.mark.gpu


@st.composite
def hbma_case(draw):
    levels = draw(st.integers(1, 4))
    f = 1 << (levels - 1)
    bw = f * draw(st.integers(1, 32 // f))
    bh = f * draw(st.integers(1, 32 // f))
    nbx, nby = draw(st.integers(1, 6)), draw(st.integers(1, 5))
    r = draw(st.integers(f, max(f, 12)))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    kind = draw(st.sampled_from(["noise", "shifted", "flat", "periodic"]))
    return levels, bw, bh, bw * nbx, bh * nby, r, seed, kind


def _planes(kind, rng, w, h, levels):
    if kind == "flat":
        base_t = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        base_a = base_t.copy()
    elif kind == "periodic":
        yy, xx = np.mgrid[0:h, 0:w]
        p = int(rng.integers(1, 4))
        base_t = ((((xx // p) % 2) * 150 + ((yy // p) % 2) * 70) % 256).astype(np.uint8)
        base_a = np.roll(base_t, (int(rng.integers(-2, 3)), int(rng.integers(-2, 3))), (0, 1))
    else:
        base_t = rng.integers(0, 256, (h, w), dtype=np.uint8)
        base_a = (np.roll(base_t, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), (0, 1))
                  if kind == "shifted" else rng.integers(0, 256, (h, w), dtype=np.uint8))

    def pyr(b):
        out = [b]
        for _ in range(levels - 1):
            out.append(np.ascontiguousarray(out[-1][::2, ::2]))
        return out
    return pyr(base_t), pyr(base_a)


@settings(max_examples=fuzz_examples(60), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture], derandomize=not FUZZ_RANDOM)
@given(case=hbma_case())
def test_hbma_random_shapes(native, oracle, case):
    levels, bw, bh, w, h, r, seed, kind = case
    t, a = _planes(kind, np.random.default_rng(seed), w, h, levels)
    exp_mv, exp_mad = oracle.hbma(t, a, r, bw, bh)
    for flags in (native.HBMA_AUTO, native.HBMA_FORCE_WAVE_PER_BLOCK):
        mv, mad = native.hbma_host(t, a, r, bw, bh, flags=flags)
        assert np.array_equal(mv, exp_mv) and np.array_equal(mad, exp_mad), (case, flags)


@settings(max_examples=fuzz_examples(40), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture], derandomize=not FUZZ_RANDOM)
@given(mb=st.sampled_from([8, 16, 32]), levels=st.integers(2, 5), rt=st.integers(1, 4), nbx=st.integers(2, 9), nby=st.integers(1, 6),
       seed=st.integers(0, 2 ** 31 - 1), kind=st.sampled_from(["noise", "shifted", "flat", "periodic"]))
def test_fused_kernel_random_frames(native, oracle, mb, levels, rt, nbx, nby, seed, kind):
    """The fused lane-per-block kernel (8x8 / 16x16 / 32x32 blocks, 2 .. 5 levels, R_top 1 .. 4) on small odd-shaped
    fields: every block near a border."""
    if (mb >> (levels - 1)) < 1:
        levels = mb.bit_length() - 1
    w, h = mb * nbx, mb * nby
    r = rt << (levels - 1)
    t, a = _planes(kind, np.random.default_rng(seed), w, h, levels)
    exp_mv, exp_mad = oracle.hbma(t, a, r, mb, mb)
    try:
        mv, mad = native.hbma_host(t, a, r, mb, mb, flags=native.HBMA_FORCE_FUSED)
    except native.SvcError as e:  # outside the instantiations / too small for the candidate grid: must say so, not guess
        assert e.status == native.SVC_ERR_UNSUPPORTED
        mv, mad = native.hbma_host(t, a, r, mb, mb)
    assert np.array_equal(mv, exp_mv) and np.array_equal(mad, exp_mad), (mb, levels, rt, w, h, kind)


@settings(max_examples=fuzz_examples(30), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture], derandomize=not FUZZ_RANDOM)
@given(nbx4=st.integers(2, 20), nby=st.integers(2, 36), r=st.integers(8, 15), seed=st.integers(0, 2 ** 31 - 1),
       kind=st.sampled_from(["noise", "shifted", "flat", "periodic"]))
def test_tiled_kernel_random_frames(native, oracle, nbx4, nby, r, seed, kind):
    """The LDS-tiled kernel (4 levels, R_top 1 -- search ranges 8 .. 15 --, frame widths that are multiples of 64) on fields
    of 8 .. 80 x 2 .. 36 blocks (smaller ones have no room for the top level's candidate grid and take the per-level kernel): every tile shape the launcher picks, whole and partial tiles, every window clamp."""
    w, h = 64 * nbx4, 16 * nby
    t, a = _planes(kind, np.random.default_rng(seed), w, h, 4)
    exp_mv, exp_mad = oracle.hbma(t, a, r, 16, 16)
    assert native.hbma_kernel_name(4, w, h, r) == "hbma_tiled16_kernel"
    for flags in (native.HBMA_FORCE_TILED, native.HBMA_AUTO, native.HBMA_FORCE_LANE):
        mv, mad = native.hbma_host(t, a, r, 16, 16, flags=flags)
        assert np.array_equal(mv, exp_mv) and np.array_equal(mad, exp_mad), (w, h, r, kind, flags)
