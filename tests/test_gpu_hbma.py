"""GPU parity of the motion search against the oracle (bit-exact MVs and min-MADs).

Reference behaviour under test: libs/motion.cpp:268-340 (EBMA), :342-410
(refinement), :412-465 (HBMA), :691-749 (the fixed 4-level 16x16 entry)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _assert_same(got_mv, got_mad, exp_mv, exp_mad, what):
    got_mv, got_mad = np.asarray(got_mv), np.asarray(got_mad)
    bad = np.flatnonzero((got_mv != exp_mv).any(axis=-1) | (got_mad != exp_mad))
    assert bad.size == 0, (f"{what}: {bad.size}/{len(exp_mad)} blocks differ; first {bad[:5]}: "
                           f"got {got_mv[bad[:5]].tolist()} {got_mad[bad[:5]].tolist()} "
                           f"want {exp_mv[bad[:5]].tolist()} {exp_mad[bad[:5]].tolist()}")


@pytest.mark.parametrize("levels", [1, 2, 3, 4])
@pytest.mark.parametrize("flags", ["auto", "wave"])
def test_hbma_host_cif(native, oracle, levels, flags):
    """C1-shaped clip through the host-pointer entry (what motion.hpp's wrapper calls)."""
    _, pyrs, _ = util.clip_frames(352, 288, 2, 0x5C0DEC01, levels)
    t, a = util.np_pyr(pyrs[0]), util.np_pyr(pyrs[1])
    exp_mv, exp_mad = oracle.hbma(t, a, 8, 16, 16)
    f = native.HBMA_AUTO if flags == "auto" else native.HBMA_FORCE_WAVE_PER_BLOCK
    mv, mad = native.hbma_host(t, a, 8, 16, 16, flags=f)
    _assert_same(mv, mad, exp_mv, exp_mad, f"L={levels} {flags}")


@pytest.mark.parametrize("levels,rng_range", [(3, 8), (3, 4), (4, 8), (4, 16)])
def test_hbma_fused_vs_oracle_noise(native, oracle, levels, rng_range):
    """Uncorrelated noise planes: every window clamp and carried-MAD path gets exercised."""
    rng = np.random.default_rng(levels * 100 + rng_range)
    w, h = 256, 192
    t, a = util.random_planes(rng, w, h, levels), util.random_planes(rng, w, h, levels)
    exp_mv, exp_mad = oracle.hbma(t, a, rng_range, 16, 16)
    mv, mad = native.hbma_host(t, a, rng_range, 16, 16, flags=native.HBMA_FORCE_FUSED)
    _assert_same(mv, mad, exp_mv, exp_mad, f"fused L={levels} R={rng_range}")
    mv, mad = native.hbma_host(t, a, rng_range, 16, 16, flags=native.HBMA_FORCE_WAVE_PER_BLOCK)
    _assert_same(mv, mad, exp_mv, exp_mad, f"wave L={levels} R={rng_range}")


FUSED_SHAPES = [(mb, L, rt) for mb in (8, 16, 32) for L in range(2, 6) if (mb >> (L - 1)) >= 2
                for rt in range(1, (2 if mb == 32 else 4) + 1)]


@pytest.mark.parametrize("mb,levels,rt", FUSED_SHAPES)
def test_hbma_fused_every_instantiation(native, oracle, mb, levels, rt):
    """Every (MV block, levels, R_top) the lane-per-block kernel is instantiated for -- 8x8 / 16x16 / 32x32 blocks, 2 ..
    log2(block) levels, R_top 1 .. 4 (apps/encoder.cpp:75-104: mv-block-w/h, pyr-lvl-count, mv-search-range) -- against
    the oracle (libs/motion.cpp:412-465): uncorrelated noise (every clamp, 32x32 SADs beyond 16 bits, 49 / 81
    candidates), a shifted copy (the planted vector must come back) and a frame of the minimum admitted size."""
    f = 1 << (levels - 1)
    r = rt * f + (f // 2 if f > 1 else 0)  # R_top = r >> (levels - 1) = rt with a remainder that must be ignored
    rng = np.random.default_rng(mb * 100 + levels * 10 + rt)
    for nbx, nby, kind in ((8, 5, "noise"), (6, 4, "shifted"), (None, None, "min")):  # even counts: the top plane's rows stay dword multiples
        if kind == "min":  # the smallest frame fused_supported admits for this shape
            tb = mb >> (levels - 1)
            tw = max(tb + 8, 12)
            tw = (tw + 3) // 4 * 4
            w = tw * f
            w = (w + mb - 1) // mb * mb
            h = max((tb + 2 * rt) * f, mb)
            h = (h + mb - 1) // mb * mb
        else:
            tb = mb >> (levels - 1)
            nby = max(nby, -(-(tb + 2 * rt) * f // mb))  # the top plane must hold a candidate grid
            w, h = mb * nbx, mb * nby
        base_t = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if kind == "shifted":
            dy, dx = int(rng.integers(-rt * f + 1, rt * f)), int(rng.integers(-rt * f + 1, rt * f))
            base_a = np.roll(base_t, (dy, dx), (0, 1))
        else:
            base_a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        t = [np.ascontiguousarray(base_t[:: 1 << l, :: 1 << l]) for l in range(levels)]
        a = [np.ascontiguousarray(base_a[:: 1 << l, :: 1 << l]) for l in range(levels)]
        exp_mv, exp_mad = oracle.hbma(t, a, r, mb, mb)
        mv, mad = native.hbma_host(t, a, r, mb, mb, flags=native.HBMA_FORCE_FUSED)
        _assert_same(mv, mad, exp_mv, exp_mad, f"fused {mb}x{mb} L={levels} R={r} {kind} {w}x{h}")
        mv, mad = native.hbma_host(t, a, r, mb, mb, flags=native.HBMA_FORCE_WAVE_PER_BLOCK)
        _assert_same(mv, mad, exp_mv, exp_mad, f"wave {mb}x{mb} L={levels} R={r} {kind} {w}x{h}")


def test_hbma_fused_says_unsupported(native):
    """Shapes outside the instantiations are UNSUPPORTED when the fused kernel is forced and take the per-level kernel
    otherwise: non-square blocks, one level, R_top 5, 32x32 at R_top 3, a top level of 1x1 blocks."""
    for levels, w, h, r, bw, bh in ((2, 256, 128, 4, 16, 8), (1, 128, 128, 2, 16, 16), (2, 256, 128, 10, 16, 16),
                                    (3, 512, 256, 12, 32, 32), (4, 256, 128, 8, 8, 8)):
        t = util.random_planes(np.random.default_rng(1), w, h, levels)
        with pytest.raises(native.SvcError) as e:
            native.hbma_host(t, t, r, bw, bh, flags=native.HBMA_FORCE_FUSED)
        assert e.value.status in (native.SVC_ERR_UNSUPPORTED, native.SVC_ERR_INVALID_ARG), (levels, bw, bh)
        if bw >> (levels - 1) >= 1 and bh >> (levels - 1) >= 1:
            native.hbma_host(t, t, r, bw, bh)


@pytest.mark.parametrize("bw,bh,r,levels", [(8, 8, 4, 1), (16, 8, 8, 2), (4, 4, 2, 1), (32, 16, 8, 3), (6, 10, 3, 1)])
def test_hbma_wave_generic_shapes(native, oracle, bw, bh, r, levels):
    """Block shapes outside the fused kernel, including a non-multiple-of-4 width (byte path)."""
    rng = np.random.default_rng(bw * 1000 + bh)
    w, h = bw * 12, bh * 9
    f = 1 << (levels - 1)
    w, h = w * f, h * f
    t, a = util.random_planes(rng, w, h, levels), util.random_planes(rng, w, h, levels)
    exp_mv, exp_mad = oracle.hbma(t, a, r, bw, bh)
    mv, mad = native.hbma_host(t, a, r, bw, bh)
    _assert_same(mv, mad, exp_mv, exp_mad, f"{bw}x{bh} R={r} L={levels}")


@pytest.mark.parametrize("bw,bh,r", [(512, 4, 2), (260, 4, 3), (1024, 2, 1)])
def test_hbma_wave_blocks_wider_than_256(native, oracle, bw, bh, r):
    """One row of a block wider than 256 pixels can exceed 65535 in a packed-u16 SAD lane: near-maximal differences
    (bright anchor, dark tracked frame) must still give the oracle's MVs and MADs (these blocks take the byte path)."""
    rng = np.random.default_rng(bw)
    w, h = bw * 3, bh * 5
    t = [rng.integers(0, 6, (h, w), dtype=np.uint8)]
    a = [(250 + rng.integers(0, 6, (h, w))).astype(np.uint8)]
    exp_mv, exp_mad = oracle.hbma(t, a, r, bw, bh)
    assert exp_mad.min() > 240
    mv, mad = native.hbma_host(t, a, r, bw, bh)
    _assert_same(mv, mad, exp_mv, exp_mad, f"{bw}x{bh}")


def test_ebma_host(native, oracle):
    rng = np.random.default_rng(7)
    t = rng.integers(0, 256, (96, 160), dtype=np.uint8)
    a = np.roll(t, (2, -3), axis=(0, 1))
    exp_mv, exp_mad = oracle.ebma(t, a, 8, 16, 16)
    mv, mad = native.ebma_host(t, a, 8, 16, 16)
    _assert_same(mv, mad, exp_mv, exp_mad, "ebma")


def test_flat_frames_zero_reset(native, oracle):
    """Flat frames: every candidate ties, the top level zero-resets the MV (motion.cpp:333-337)."""
    t = [np.full((128 >> l, 192 >> l), 77, np.uint8) for l in range(3)]
    a = [np.full((128 >> l, 192 >> l), 77, np.uint8) for l in range(3)]
    exp_mv, exp_mad = oracle.hbma(t, a, 8, 16, 16)
    assert not exp_mv.any() and not exp_mad.any()
    for f in (native.HBMA_FORCE_FUSED, native.HBMA_FORCE_WAVE_PER_BLOCK):
        mv, mad = native.hbma_host(t, a, 8, 16, 16, flags=f)
        _assert_same(mv, mad, exp_mv, exp_mad, "flat")


def test_periodic_ties(native, oracle):
    """Exactly periodic texture: many exact ties; top picks the LAST raster minimum, refinement the FIRST."""
    yy, xx = np.mgrid[0:256, 0:384]
    base = (((xx // 2) % 2) * 120 + ((yy // 2) % 2) * 60 + 20).astype(np.uint8)
    for levels in (3, 4):
        t = [np.ascontiguousarray(base[:: 1 << l, :: 1 << l]) for l in range(levels)]
        a = [np.ascontiguousarray(np.roll(base, (1, 1), (0, 1))[:: 1 << l, :: 1 << l]) for l in range(levels)]
        exp_mv, exp_mad = oracle.hbma(t, a, 8, 16, 16)
        for f in (native.HBMA_FORCE_FUSED, native.HBMA_FORCE_WAVE_PER_BLOCK):
            mv, mad = native.hbma_host(t, a, 8, 16, 16, flags=f)
            _assert_same(mv, mad, exp_mv, exp_mad, f"periodic L={levels}")


@pytest.mark.parametrize("levels", [3, 4])
def test_hbma_pairs_batched_clip(native, oracle, levels):
    """Device-resident clip layout: pair p = (frame p, frame p+1) of one packed buffer."""
    n = 5
    _, pyrs, (pw, ph) = util.clip_frames(320, 240, n, 0xABC + levels, levels)
    stride = native.pyramid_stride(pw, ph, levels)
    buf = util.pack_clip(pyrs, stride, "cuda")
    mv, mad = native.hbma_pairs(buf, buf[stride:], stride, n - 1, levels, pw, ph, 8)
    mvw, madw = native.hbma_pairs(buf, buf[stride:], stride, n - 1, levels, pw, ph, 8,
                                  flags=native.HBMA_FORCE_WAVE_PER_BLOCK)
    torch.cuda.synchronize()
    for p in range(n - 1):
        exp_mv, exp_mad = oracle.hbma(util.np_pyr(pyrs[p]), util.np_pyr(pyrs[p + 1]), 8, 16, 16)
        _assert_same(mv[p].cpu().numpy(), mad[p].cpu().numpy(), exp_mv, exp_mad, f"fused pair {p}")
        _assert_same(mvw[p].cpu().numpy(), madw[p].cpu().numpy(), exp_mv, exp_mad, f"wave pair {p}")


@pytest.mark.parametrize("w,h,n", [(512, 128, 3), (640, 400, 3), (1024, 256, 2), (128, 128, 2), (1152, 640, 2), (192, 144, 2), (576, 1088, 2),
                                   (1920, 1088, 2), (2048, 64, 2), (320, 528, 2), (704, 528, 2), (960, 544, 2)])
@pytest.mark.parametrize("kind", ["clip", "noise"])
def test_hbma_tiled_kernel(native, oracle, w, h, n, kind):
    """The LDS-tiled form of the 4-level search (what SVC_HBMA_AUTO takes for 16 x 16 blocks, 4 levels, R_top = 1 and a
    frame width that is a multiple of 64; SVC_HBMA_FORCE_TILED names it): levels 2 and 1 searched from LDS tiles.  Frames
    of one tile, of partial tiles in both directions (640 = 1.25 tiles of 32 blocks, 400 / 16 = 25 block rows = 3.1 tiles
    of 8), of several tiles, smaller than a tile, and shapes that take each of the three tile forms (64 x 4 blocks: 1080p,
    2048 x 64, 960 x 544, 704 x 528; 32 x 8: 512 x 128, 1152 x 640, 320 x 528; 16 x 16: 640 x 400, 192 x 144);
    coherent clips and uncorrelated noise (every window clamp, vectors
    up to the +-6 / +-2 the tile margins are sized for).  Against the oracle, the lane-per-block kernel and the per-level
    kernel, libs/motion.cpp:691-749."""
    levels, r = 4, 8
    if kind == "clip":
        _, pyrs, (pw, ph) = util.clip_frames(w, h, n, 0x711E + w, levels)
        pyrs = [util.np_pyr(p) for p in pyrs]
    else:
        rng = np.random.default_rng(w * 7 + h)
        pw, ph = w, h
        pyrs = [util.random_planes(rng, w, h, levels) for _ in range(n)]
    assert (pw, ph) == (w, h)
    stride = native.pyramid_stride(pw, ph, levels)
    buf = util.pack_clip(pyrs, stride, "cuda")
    assert native.hbma_kernel_name(levels, pw, ph, r) == "hbma_tiled16_kernel"
    out = {}
    for name, f in (("tiled", native.HBMA_FORCE_TILED), ("lane", native.HBMA_FORCE_LANE), ("auto", native.HBMA_AUTO),
                    ("wave", native.HBMA_FORCE_WAVE_PER_BLOCK)):
        out[name] = native.hbma_pairs(buf, buf[stride:], stride, n - 1, levels, pw, ph, r, flags=f)
    torch.cuda.synchronize()
    for p in range(n - 1):
        exp_mv, exp_mad = oracle.hbma(pyrs[p], pyrs[p + 1], r, 16, 16)
        for name, (mv, mad) in out.items():
            _assert_same(mv[p].cpu().numpy(), mad[p].cpu().numpy(), exp_mv, exp_mad, f"{name} {w}x{h} {kind} pair {p}")


def test_hbma_tiled_kernel_says_unsupported(native):
    """Shapes the tiled form does not cover are UNSUPPORTED when it is forced (3 levels; a frame width that is not a multiple
    of 64; R_top = 2), and run through the lane-per-block form otherwise."""
    for levels, w, h, r in ((3, 256, 128, 8), (4, 352, 256, 8), (4, 256, 128, 16)):
        t = util.random_planes(np.random.default_rng(1), w, h, levels)
        with pytest.raises(native.SvcError) as e:
            native.hbma_host(t, t, r, 16, 16, flags=native.HBMA_FORCE_TILED)
        assert e.value.status == native.SVC_ERR_UNSUPPORTED
        native.hbma_host(t, t, r, 16, 16, flags=native.HBMA_FORCE_LANE)


def test_invalid_args(native):
    t = [np.zeros((64, 64), np.uint8)]
    with pytest.raises(native.SvcError) as e:
        native.hbma_host(t, t, 8, 16, 24)  # 64 % 24 != 0 (motion.cpp:428-429)
    assert e.value.status == native.SVC_ERR_INVALID_ARG
    with pytest.raises(native.SvcError) as e:
        native.hbma_host(t * 3, t * 3, 2, 16, 16)  # search_range < 2^(L-1) (motion.cpp:433)
    assert e.value.status == native.SVC_ERR_INVALID_ARG


def test_empty_batches_are_noops(native):
    """Zero pairs / frames: every batched entry point returns OK without touching memory."""
    e8 = torch.empty(0, dtype=torch.uint8, device="cuda")
    mv, mad = native.hbma_pairs(e8, e8, 256, 0, 3, 64, 64, 8)
    assert mv.shape == (0, 16, 2) and mad.shape == (0, 16)
    assert native.dct_frames(torch.empty((0, 32, 32, 3), dtype=torch.uint8, device="cuda"), 8).shape == (0, 3, 32, 32)
    gm, rmse, mask, count = native.ransac_frames(torch.empty((0, 16, 2), device="cuda"),
                                                 torch.empty((0, 7, 1), dtype=torch.int32, device="cuda"))
    assert gm.shape == (0, 2) and mask.shape == (0, 16)
    assert native.segment_frames(torch.empty((0, 16), dtype=torch.uint8, device="cuda"),
                                 torch.empty((0, 16, 2), device="cuda"), 4, 4).shape == (0, 16)
    buf, _ = native.luma_pyramid_frames(torch.empty((0, 32, 32, 3), dtype=torch.uint8, device="cuda"), 2)
    assert buf.numel() == 0


def test_tiny_frames_fall_back_to_the_general_kernel(native, oracle):
    """Frames too small for the fused kernel's candidate grid (top plane < block + 2 R_top) still work."""
    rng = np.random.default_rng(4)
    # (112, 32, 4): top plane 14 px wide, not dword-aligned -> the fused kernel must decline
    # (regression, found by tests/test_gpu_hbma_property.py)
    for w, h, levels in ((16, 16, 3), (32, 16, 3), (48, 32, 4), (16, 48, 1), (112, 32, 4), (176, 64, 4)):
        t, a = util.random_planes(rng, w, h, levels), util.random_planes(rng, w, h, levels)
        exp_mv, exp_mad = oracle.hbma(t, a, 8, 16, 16)
        mv, mad = native.hbma_host(t, a, 8, 16, 16)
        _assert_same(mv, mad, exp_mv, exp_mad, f"{w}x{h} L={levels}")


@pytest.mark.parametrize("w,h,mb,levels", [(720, 576, 16, 4), (176, 144, 16, 4), (336, 272, 16, 4), (1360, 768, 16, 4), (112, 32, 16, 4),
                                           (120, 72, 8, 3), (224, 96, 32, 5)])
def test_top_level_rows_that_are_not_whole_dwords(native, oracle, w, h, mb, levels):
    """Frames 16 mod 32 pixels wide at 4 levels of 16 x 16 (PAL, QCIF, 1360 x 768 ...; likewise 8 mod 16 at 3 levels of 8 x 8, 32 mod 64 at
    5 of 32 x 32): the top plane's rows are 2 mod 4 bytes, so its row starts are not dword-aligned.  The lane-per-block kernel takes
    them since round 4 (top-level dwords read where they lie, only the plane's end guarded) -- against the oracle and the general kernel,
    with the pyramids packed at their exact size so that the last pair's top planes end where the buffer's data ends."""
    rng = np.random.default_rng(w + levels)
    n = 3
    pyrs = [util.random_planes(rng, w, h, levels) for _ in range(n + 1)]
    # a few flat / periodic rows at the bottom right: ties and clamped windows next to the plane's end
    for p in pyrs:
        p[levels - 1][-3:, -6:] = 77
    exact = sum(p.size for p in pyrs[0])
    stride = (exact + 15) & ~15
    buf = util.pack_clip(pyrs, stride, torch.device("cuda"))[: n * stride + exact].clone()
    assert native.hbma_kernel_name(levels, w, h, 8 if mb <= 16 else 16, mb, mb) == "hbma_fused_kernel"
    r = 8 if mb <= 16 else 16
    mv, mad = native.hbma_pairs(buf, buf[stride:], stride, n, levels, w, h, r, mb, mb)
    mvw, madw = native.hbma_pairs(buf, buf[stride:], stride, n, levels, w, h, r, mb, mb, flags=native.HBMA_FORCE_WAVE_PER_BLOCK)
    torch.cuda.synchronize()
    assert torch.equal(mv, mvw) and torch.equal(mad, madw)
    for p in range(n):
        exp_mv, exp_mad = oracle.hbma(pyrs[p], pyrs[p + 1], r, mb, mb)
        _assert_same(mv[p].cpu().numpy(), mad[p].cpu().numpy(), exp_mv, exp_mad, f"{w}x{h} pair {p}")


def test_random_search_configurations(native, oracle):
    """Seeded random block shapes (square and not, 2 ... 48 pixels a side), level counts, search ranges (R_top 1 ... 9) and field sizes
    through svc_hip_hbma_host, whichever kernel the dispatch picks, against the oracle: MVs and min-MADs bit for bit.  Periodic texture
    in half of the cases (exact ties: last minimum at the top level, first at the refinements)."""
    rng = np.random.default_rng(77)
    done = 0
    while done < 50:
        levels = int(rng.integers(1, 6))
        f = 1 << (levels - 1)
        bw, bh = f * int(rng.integers(1, max(2, 48 // f))), f * int(rng.integers(1, max(2, 48 // f)))
        if rng.random() < 0.5:
            bh = bw
        if bw // f < 1 or bh // f < 1 or bw * bh > 4096:
            continue
        r = f * int(rng.integers(1, 10))
        if r > 64:
            continue
        nx, ny = int(rng.integers(1, 9)), int(rng.integers(1, 7))
        w, h = bw * nx, bh * ny
        if (w >> (levels - 1)) < (bw >> (levels - 1)) + 0 or w * h > 400 * 300:
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
        for l in range(1, levels):
            if min(pa[-1].shape) < 3:
                ok = False
                break
            pa.append(oracle.pyr_down(pa[-1]))
            pb.append(oracle.pyr_down(pb[-1]))
        if not ok:
            continue
        mv, mad = native.hbma_host(pa, pb, r, bw, bh)
        emv, emad = oracle.hbma(pa, pb, r, bw, bh)
        assert np.array_equal(mv, emv) and np.array_equal(mad, emad), (levels, bw, bh, r, w, h)
        done += 1
