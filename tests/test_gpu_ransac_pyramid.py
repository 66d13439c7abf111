"""GPU parity of RANSAC (libs/motion.cpp:182-266, draws made explicit) and of the
luma/pyramid pre-step (this repo's fixed-point definitions, synth.py)."""
import os

import numpy as np
import pytest
import torch

from oracle.binding import DEFAULT_RANSAC
from scalable_video_codec_amd import synth
from tests import util

pytestmark = pytest.mark.gpu


def _field(rng, n, outlier_frac):
    mv = np.tile(np.array([[3.0, -2.0]], np.float32), (n, 1))
    k = int(n * outlier_frac)
    idx = rng.choice(n, k, replace=False)
    mv[idx] += rng.integers(-14, 15, (k, 2)).astype(np.float32)
    return mv


@pytest.mark.parametrize("subset", [1, 2, 5])
@pytest.mark.parametrize("outliers", [0.0, 0.3, 0.9])
@pytest.mark.parametrize("fractional", [False, True])
def test_ransac_host_vs_oracle(native, oracle, subset, outliers, fractional):
    """Integral MVs take the exact integer-sum path, fractional ones the in-order f32 walk
    (order-dependent rounding, motion.cpp:156-159); both must be bit-identical."""
    rng = np.random.default_rng(subset * 10 + int(outliers * 10))
    n = 8160 if not fractional else 9001  # 9001 > 2 LDS chunks
    mv = _field(rng, n, outliers)
    if fractional:
        mv += (rng.random((n, 2)) * 0.37).astype(np.float32)
    p = dict(DEFAULT_RANSAC, subset_sz=subset)
    k = oracle.ransac_iter_count(**p)
    assert k == native.ransac_iter_count(**p)
    samples = np.stack([rng.choice(n, subset, replace=False) for _ in range(k)]).astype(np.uint32)
    gm_o, rmse_o, inl_o = oracle.ransac(mv, samples, gm_in=(1.5, 2.5), **p)
    gm, rmse, inl = native.ransac_host(mv, samples, gm_in=(1.5, 2.5), **p)
    assert gm.tobytes() == gm_o.tobytes() and rmse.tobytes() == rmse_o.tobytes()
    assert np.array_equal(inl, inl_o)


def test_ransac_too_few_inliers_branch(native, oracle):
    """All MVs far apart, subset 3: no iteration gathers 3 inliers -> motion.cpp:240-242 branch."""
    n = 64
    mv = (np.arange(n * 2, dtype=np.float32).reshape(n, 2) * 100.0)
    p = dict(DEFAULT_RANSAC, subset_sz=3, inlier_thresh=1.0)
    k = oracle.ransac_iter_count(**p)
    rng = np.random.default_rng(5)
    samples = np.stack([rng.choice(n, 3, replace=False) for _ in range(k)]).astype(np.uint32)
    gm_o, rmse_o, inl_o = oracle.ransac(mv, samples, gm_in=(7.0, -1.0), **p)
    gm, rmse, inl = native.ransac_host(mv, samples, gm_in=(7.0, -1.0), **p)
    assert len(inl_o) < 3
    assert gm.tobytes() == gm_o.tobytes() and rmse.tobytes() == rmse_o.tobytes() and np.array_equal(inl, inl_o)


def test_ransac_frames_batched(native, oracle):
    rng = np.random.default_rng(99)
    frames, n = 6, 3600
    mv = np.stack([_field(rng, n, 0.1 * f) for f in range(frames)])
    k = oracle.ransac_iter_count(**DEFAULT_RANSAC)
    samples = rng.integers(0, n, (frames, k, 1)).astype(np.int32)
    gm, rmse, mask, count = native.ransac_frames(torch.from_numpy(mv).cuda(), torch.from_numpy(samples).cuda())
    torch.cuda.synchronize()
    for f in range(frames):
        gm_o, rmse_o, inl_o = oracle.ransac(mv[f], samples[f].astype(np.uint32), **DEFAULT_RANSAC)
        assert gm[f].cpu().numpy().tobytes() == gm_o.tobytes()
        assert rmse[f].cpu().numpy().tobytes() == rmse_o.tobytes()
        assert np.array_equal(np.flatnonzero(mask[f].cpu().numpy()), inl_o)
        assert int(count[f]) == len(inl_o)
        # fg mask of libs/encoder.cpp:507-513 is the complement
        assert np.array_equal(oracle.fg_mask(inl_o, n) == 0, mask[f].cpu().numpy() == 1)


@pytest.mark.parametrize("frames,n", [(1, 2048), (3, 2049), (257, 8160), (300, 3600), (259, 8192), (5, 8200), (3, 32768), (2, 33000)])
@pytest.mark.parametrize("subset", [1, 3])
@pytest.mark.parametrize("flags", [0, 1], ids=["alone", "beside"])
def test_ransac_frames_every_launch_shape(native, oracle, frames, n, subset, flags):
    """Every kernel variant (256 / 1024 lanes, 8 / 16 / 32 blocks per lane, one or two frames per workgroup with an odd
    frame left over, the L2-walking fallback), frames that take the integer fast path next to frames that need the
    in-order walk, and frames where no iteration gathers a subset (motion.cpp:240-242)."""
    rng = np.random.default_rng(frames * 31 + n + subset)
    p = dict(DEFAULT_RANSAC, subset_sz=subset)
    k = oracle.ransac_iter_count(**p)
    check = sorted(set([0, 1, frames // 2, frames - 2, frames - 1]) & set(range(frames)))
    mv = np.empty((frames, n, 2), np.float32)
    for f in range(frames):
        mv[f] = _field(rng, n, 0.05 * (f % 7))
        if f % 3 == 1:
            mv[f] += (rng.random((n, 2)) * 0.37).astype(np.float32)       # fractional: the serial sums
        if f % 5 == 4:
            mv[f] = (rng.random((n, 2)) * 1e4).astype(np.float32)          # scattered: too few inliers for subset 3
    samples = np.stack([np.stack([rng.choice(n, subset, replace=False) for _ in range(k)]) for _ in range(frames)]).astype(np.int32)
    gm_in = rng.integers(-3, 4, (frames, 2)).astype(np.float32)
    # flags = SVC_LAUNCH_BESIDE: 256 lanes x 32 blocks in registers for fields of 2 049 .. 8 192 blocks
    gm, rmse, mask, count = native.ransac_frames(torch.from_numpy(mv).cuda(), torch.from_numpy(samples).cuda(),
                                                 gm_in=torch.from_numpy(gm_in).cuda(), flags=flags, **p)
    torch.cuda.synchronize()
    # SVC_LAUNCH_DEFER_RMSE + svc_hip_ransac_rmse_frames (the in-order RMSE sum as a launch of its own, what the pipelined
    # driver runs beside the segmentation): the same bytes, every frame -- those that keep the best subset's model
    # (count < subset: their RMSE comes from the first launch and must be left alone) included
    gm2, rmse2, mask2, count2 = native.ransac_frames(torch.from_numpy(mv).cuda(), torch.from_numpy(samples).cuda(),
                                                     gm_in=torch.from_numpy(gm_in).cuda(), flags=flags | native.LAUNCH_DEFER_RMSE, **p)
    native.ransac_rmse_frames(torch.from_numpy(mv).cuda(), gm2, mask2, count2, rmse2, **p)
    torch.cuda.synchronize()
    assert torch.equal(gm2, gm) and torch.equal(mask2, mask) and torch.equal(count2, count)
    assert rmse2.cpu().numpy().tobytes() == rmse.cpu().numpy().tobytes()
    native.ransac_rmse_frames(torch.from_numpy(mv).cuda(), gm, mask, count, rmse2, **p)  # idempotent after a full call
    torch.cuda.synchronize()
    assert rmse2.cpu().numpy().tobytes() == rmse.cpu().numpy().tobytes()
    gm, rmse, mask, count = gm.cpu().numpy(), rmse.cpu().numpy(), mask.cpu().numpy(), count.cpu().numpy()
    for f in check:
        gm_o, rmse_o, inl_o = oracle.ransac(mv[f], samples[f].astype(np.uint32), gm_in=tuple(gm_in[f]), **p)
        assert gm[f].tobytes() == gm_o.tobytes(), f
        assert rmse[f].tobytes() == rmse_o.tobytes(), f
        assert np.array_equal(np.flatnonzero(mask[f]), inl_o) and int(count[f]) == len(inl_o), f


@pytest.mark.parametrize("frames,n,subset", [(3, 3600, 1), (2, 8160, 3), (5, 40000, 2)])
def test_ransac_device_draw_past_the_field_stays_inside(native, oracle, frames, n, subset):
    """A device-side sample index >= blocks (the reference's inclusive draw, motion.cpp:208) cannot be rejected by the
    device entry point; it must act as index blocks - 1 -- never the next frame's first MV, never past the allocation
    on the last frame of a batch (include/svc_hip.h).  Every kernel variant, every frame incl. the last."""
    rng = np.random.default_rng(n + subset)
    p = dict(DEFAULT_RANSAC, subset_sz=subset)
    k = oracle.ransac_iter_count(**p)
    mv = np.stack([_field(rng, n, 0.2 * f) for f in range(frames)])
    mv[:, -1] = [40.0, -35.0]            # entry blocks - 1 is distinctive, and so is the next frame's entry 0
    mv[:, 0] = [-50.0, 45.0]
    samples = np.stack([np.stack([rng.choice(n - 1, subset, replace=False) for _ in range(k)]) for _ in range(frames)]).astype(np.int64)
    samples[:, -1, 0] = n                # the last iteration wins ties (>=): make it the out-of-range one
    samples[:, 0, subset - 1] = n + 7
    clamped = np.minimum(samples, n - 1)
    dev_mv = torch.from_numpy(mv).cuda()
    got = native.ransac_frames(dev_mv, torch.from_numpy(samples.astype(np.int32)).cuda(), **p)
    want = native.ransac_frames(dev_mv, torch.from_numpy(clamped.astype(np.int32)).cuda(), **p)
    torch.cuda.synchronize()
    for g, w_ in zip(got, want):
        assert torch.equal(g, w_)
    for f in (0, frames - 1):
        gm_o, rmse_o, inl_o = oracle.ransac(mv[f], clamped[f].astype(np.uint32), **p)
        assert got[0][f].cpu().numpy().tobytes() == gm_o.tobytes() and got[1][f].cpu().numpy().tobytes() == rmse_o.tobytes()
        assert np.array_equal(np.flatnonzero(got[2][f].cpu().numpy()), inl_o)


# the last three are regressions: frames shorter than one 32-row LDS tile (rows beyond the frame
# must not be reflected twice), found by tests/test_gpu_misc_property.py
# 720 x 576 (PAL), 1360 x 768, 176 x 144 (QCIF), 336 x 272: frames 16 mod 32 pixels wide, whose level-3 plane is not a whole number
# of dwords wide (90, 170, 22, 42) -- what the reference's DEFAULT 4-level build makes of them (refused until round 4); 5 levels too
@pytest.mark.parametrize("w,h,levels", [(352, 288, 1), (320, 208, 3), (640, 368, 4), (64, 16, 2), (32, 8, 3), (128, 2, 2),
                                        (720, 576, 4), (1360, 768, 4), (176, 144, 4), (336, 272, 4), (720, 576, 5), (48, 16, 4), (96, 48, 5),
                                        (360, 200, 3), (40, 24, 4), (8, 8, 1), (24, 12, 2)])  # the last four: not a multiple of 16 wide (8 x 8 MV blocks pad to 8)
def test_luma_pyramid(native, oracle, w, h, levels):
    """svc_hip_luma_pyramid_frames against the oracle's restatement of cvtColor + buildPyramid (libs/encoder.cpp:468-470)."""
    rng = np.random.default_rng(w * 31 + h)
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(2)]
    bgr = torch.from_numpy(np.stack(frames)).cuda()
    buf, stride = native.luma_pyramid_frames(bgr, levels)
    torch.cuda.synchronize()
    offs = synth.level_offsets(w, h, levels)
    for i, f in enumerate(frames):
        for l, p in enumerate(oracle.luma_pyramid(f, levels)):
            got = buf[i * stride + offs[l]: i * stride + offs[l] + p.size].cpu().numpy().reshape(p.shape)
            assert np.array_equal(got, p), f"frame {i} level {l}: {(got != p).sum()} px differ"


# cv::buildPyramid from planes that exist (the steps that read the BGR clip once leave level 0 from the transform kernel): one tile, partial tiles
# in x / in y / both, planes whose edge falls exactly on a tile boundary, 1080p and 4K level-0 shapes, small planes whose mirrored samples all
# come from one tile, widths that are not whole 16-byte segments (the gather kernel), 1 - 5 levels
@pytest.mark.parametrize("w,h,levels", [(256, 64, 3), (512, 128, 3), (272, 72, 3), (1920, 1088, 3), (1920, 1088, 4), (3840, 2160, 4), (3840, 2160, 5),
                                        (16, 8, 3), (32, 16, 3), (48, 12, 3), (240, 68, 3), (768, 192, 5), (1008, 500, 3), (264, 64, 3),
                                        (128, 8, 2), (720, 576, 4)])
def test_pyramid_levels_from_an_existing_plane(native, oracle, w, h, levels):
    """svc_hip_pyramid_levels_frames (cv::buildPyramid from given level-0 planes, libs/encoder.cpp:470) against the oracle, level by level."""
    rng = np.random.default_rng(w * 7 + h + levels)
    n = 3
    planes = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(n)]
    planes[1][:] = 255  # saturated: the rounding at the top of the range
    stride = native.pyramid_stride(w, h, levels)
    buf = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
    for i, p in enumerate(planes):
        buf[i * stride:i * stride + w * h] = torch.from_numpy(p.ravel()).cuda()
    native.pyramid_levels_frames(buf, stride, n, w, h, levels)
    torch.cuda.synchronize()
    offs = synth.level_offsets(w, h, levels)
    for i, p in enumerate(planes):
        want = p
        for l in range(1, levels):
            want = oracle.pyr_down(want)
            got = buf[i * stride + offs[l]: i * stride + offs[l] + want.size].cpu().numpy().reshape(want.shape)
            assert np.array_equal(got, want), f"frame {i} level {l}: {(got != want).sum()} of {want.size} px differ"
        assert np.array_equal(buf[i * stride:i * stride + w * h].cpu().numpy().reshape(h, w), p)  # level 0 untouched


def test_luma_pyramid_golden(native):
    """The committed fixture (tests/golden/luma_pyramid.npz, expected values from an independent numpy / scipy formulation):
    the frames wide enough for the device entry point, through the C ABI."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "luma_pyramid.npz"))
    ran = 0
    for n in sorted({k.split("/")[0] for k in z.files if k.startswith("pyr_")}):
        bgr = z[f"{n}/bgr"]
        levels = sum(1 for k in z.files if k.startswith(f"{n}/level"))
        h, w, _ = bgr.shape
        try:
            buf, stride = native.luma_pyramid_frames(torch.from_numpy(bgr[None]).cuda(), levels)
        except native.SvcError as e:  # widths that are not a multiple of 16: a clean refusal
            assert e.status in (native.SVC_ERR_UNSUPPORTED, native.SVC_ERR_INVALID_ARG)
            continue
        offs = synth.level_offsets(w, h, levels)
        for l in range(levels):
            want = z[f"{n}/level{l}"]
            assert np.array_equal(buf[offs[l]:offs[l] + want.size].cpu().numpy().reshape(want.shape), want), (n, l)
        ran += 1
    assert ran >= 2


def test_random_ransac_parameters(native, oracle):
    """Seeded random RansacParams (subset 1 ... 6, thresholds from sub-pixel to larger than any vector, success probabilities and inlier
    ratios that give 1 ... 200 iterations), field sizes from a handful of vectors to 4K's, integral and fractional vectors, one / two /
    no dominant motion: global motion, RMSE and the inlier list through svc_hip_ransac_host against the oracle, bit for bit."""
    rng = np.random.default_rng(99)
    for case in range(60):
        n = int(rng.choice([7, 50, 396, 3600, 8160, 8193, 32400]))
        subset = int(rng.integers(1, 7))
        p = dict(subset_sz=subset, inlier_thresh=float(rng.choice([0.4, 1.5, 7.5, 40.0])), success_prob=float(rng.choice([0.5, 0.9, 0.99, 0.999])),
                 inlier_ratio=float(rng.choice([0.2, 0.5, 0.8, 0.95])))
        iters = oracle.ransac_iter_count(**p)
        if not 1 <= iters <= 200 or subset > n:
            continue
        kind = int(rng.integers(0, 3))
        mv = rng.integers(-14, 15, (n, 2)).astype(np.float32) if kind == 0 else np.zeros((n, 2), np.float32)
        if kind >= 1:
            mv[:] = rng.integers(-6, 7, 2)
            cut = int(n * rng.random())
            mv[cut:] = rng.integers(-14, 15, (n - cut, 2)) if kind == 1 else rng.integers(-6, 7, 2)
        if rng.random() < 0.3:
            mv += rng.random((n, 2)).astype(np.float32) * 0.5  # fractional vectors: the f32 sums are order-dependent
        samples = np.stack([rng.permutation(n)[:subset] for _ in range(iters)]).astype(np.uint32)
        gm, rmse, inl = native.ransac_host(mv, samples, **p)
        egm, ermse, einl = oracle.ransac(mv, samples.ravel(), **p)
        assert gm.tobytes() == egm.tobytes() and np.float32(rmse).tobytes() == np.float32(ermse).tobytes() and np.array_equal(inl, einl), (case, n, p, kind)
