"""GPU results against the committed golden vectors (outputs of the unmodified reference,
tests/golden/make_golden.py).  Bit-exact for MVs, min-MADs, inlier sets and global motion;
DCT within 1e-4 * max(1, |ref|) of the float64 DCT-II."""
import numpy as np
import pytest
import torch

from tests import golden_util as G
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", G.HBMA_CONFIGS, ids=lambda c: c.name)
def test_hbma_configs(native, cfg):
    z = G.load(f"hbma_{cfg.name}.npz")
    (_, p0), (_, p1) = G.config_pair(cfg)
    assert G.sha(p0) == str(z["sha_tracked"]) and G.sha(p1) == str(z["sha_anchor"])
    mv, mad = native.hbma_host(p0, p1, cfg.search_range, cfg.mv_block, cfg.mv_block)
    assert np.array_equal(mv, z["mv"]) and np.array_equal(mad, z["mad"]), cfg.name
    # the per-level LDS-staged kernel must agree too (for C1 it IS the kernel: 289 candidates)
    mvw, madw = native.hbma_host(p0, p1, cfg.search_range, cfg.mv_block, cfg.mv_block,
                                 flags=native.HBMA_FORCE_WAVE_PER_BLOCK)
    assert np.array_equal(mvw, z["mv"]) and np.array_equal(madw, z["mad"]), cfg.name


@pytest.mark.parametrize("case", list(G.micro_cases()), ids=lambda c: c[0])
def test_hbma_micro(native, case):
    name, t, a, r, bw, bh, mv_ref, mad_ref = case
    for flags in (native.HBMA_AUTO, native.HBMA_FORCE_WAVE_PER_BLOCK):
        mv, mad = native.hbma_host(t, a, r, bw, bh, flags=flags)
        assert np.array_equal(mv, mv_ref) and np.array_equal(mad, mad_ref), (name, flags)


@pytest.mark.parametrize("case", list(G.shape_cases()), ids=lambda c: c[0])
def test_hbma_every_fused_shape_against_the_reference(native, case):
    """tests/golden/hbma_shapes.npz: the unmodified reference's EstimateMotionHierarchical (libs/motion.cpp:412-465) for every
    (MV block, levels, search range) the lane-per-block kernel is instantiated for; the forced fused kernel, the dispatcher's
    choice and the per-level kernel must all reproduce it."""
    key, mb, levels, r, t, a, mv_ref, mad_ref = case
    for flags in (native.HBMA_FORCE_FUSED, native.HBMA_AUTO, native.HBMA_FORCE_WAVE_PER_BLOCK):
        mv, mad = native.hbma_host(t, a, r, mb, mb, flags=flags)
        assert np.array_equal(mv, mv_ref) and np.array_equal(mad, mad_ref), (key, flags)


@pytest.mark.parametrize("case", list(G.ransac_cases()), ids=lambda c: c[0])
def test_ransac(native, case):
    name, mv, params, samples, gm_ref, rmse_ref, inl_ref = case
    n = len(mv) - 1
    if int(samples.max()) >= n:
        # the reference drew index n (its off-by-one, motion.cpp:208): rejected, not reproduced
        with pytest.raises(native.SvcError) as e:
            native.ransac_host(mv[:n], samples, gm_in=(0.25, -0.75), **params)
        assert e.value.status == native.SVC_ERR_INVALID_ARG
        return
    gm, rmse, inl = native.ransac_host(mv[:n], samples, gm_in=(0.25, -0.75), **params)
    assert gm.tobytes() == gm_ref.tobytes() and rmse.tobytes() == rmse_ref.tobytes(), name
    assert np.array_equal(inl, inl_ref)


def test_dct_tiles(native):
    z = G.load("dct_tiles.npz")
    for key in sorted({k.split("/")[0] for k in z.files}):
        tin, tout = z[f"{key}/in"], z[f"{key}/out"]
        n, blk = tin.shape[0], tin.shape[1]
        # lay the tiles side by side as one frame (width must be a multiple of 16)
        cols = n if (n * blk) % 16 == 0 else n + 1
        frame = np.zeros((blk, cols * blk, 3), np.uint8)
        for i in range(n):
            frame[:, i * blk:(i + 1) * blk] = tin[i]
        got = native.dct_host(frame, blk)
        for i in range(n):
            ref = tout[i]
            err = np.abs(got[:, :, i * blk:(i + 1) * blk].astype(np.float64) - ref)
            assert (err <= 1e-4 * np.maximum(1.0, np.abs(ref))).all(), (key, i, err.max())


def test_quant_hand_vectors(native):
    z = G.load("quant.npz")
    assert native.quant_host(z["step640/in"], 640).tobytes() == z["step640/out"].tobytes()
    assert native.quant_host(z["step1/in"], 1).tobytes() == z["step1/out"].tobytes()
