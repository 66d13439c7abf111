"""GPU parity of the transform path.

DCT: within 1e-4 * max(1, |ref|) of the float64 orthonormal DCT-II (the oracle of
record; cv::dct itself is unavailable offline -- parity with OpenCV is unpinned).
Quant: bit-exact against libs/decoder.cpp:140-144 restated."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: float DCT within 1e-4 (relative to max(1, |ref|))


def _dct_close(got, ref64):
    err = np.abs(got.astype(np.float64) - ref64)
    lim = TOL * np.maximum(1.0, np.abs(ref64))
    worst = float((err / lim).max())
    assert worst <= 1.0, f"worst error / tolerance = {worst}"
    return float(err.max())


@pytest.mark.parametrize("block", [8, 16])
def test_dct_host_random(native, oracle, block):
    rng = np.random.default_rng(block)
    bgr = rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)
    ref = oracle.dct_frame_f64(bgr, block, block)
    got = native.dct_host(bgr, block)
    _dct_close(got, ref)


@pytest.mark.parametrize("block", [8, 16])
def test_dct_known_answers(native, block):
    """Constant tile v -> DC = v * block, all AC = 0; extremes 0 and 255."""
    for v in (0, 1, 128, 255):
        bgr = np.full((32, 48, 3), v, np.uint8)
        got = native.dct_host(bgr, block)
        dc = got[:, ::block, ::block]
        assert np.allclose(dc, v * block, rtol=1e-6, atol=1e-4), (v, dc.ravel()[:4])
        ac = got.copy()
        ac[:, ::block, ::block] = 0
        assert np.abs(ac).max() <= 1e-4, (v, np.abs(ac).max())


@pytest.mark.parametrize("block", [8, 16])
def test_dct_parseval_and_planes(native, oracle, block):
    """Orthonormal transform preserves energy per tile; plane order is B, G, R (cv::split)."""
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    got = native.dct_host(bgr, block).astype(np.float64)
    for c in range(3):
        e_in = (bgr[..., c].astype(np.float64) ** 2).sum()
        assert abs(e_in - (got[c] ** 2).sum()) <= 1e-6 * e_in
    one = np.zeros((16, 16, 3), np.uint8)
    one[..., 1] = 200  # G only
    g = native.dct_host(one, block)
    assert g[0].max() == 0 and g[2].max() == 0 and g[1, 0, 0] > 0


def test_dct_frames_batched_synthetic(native, oracle):
    frames, _, (pw, ph) = util.clip_frames(320, 200, 3, 0xD0, 3)
    assert ph == 208  # zero padding rows are part of the transform input (encoder.cpp:459-461, :638)
    bgr = torch.stack(frames).cuda()
    got = native.dct_frames(bgr, 8).cpu().numpy()
    for i, f in enumerate(frames):
        _dct_close(got[i], oracle.dct_frame_f64(f.numpy(), 8, 8))


def test_quant_hand_vectors(native):
    """round half away from zero, step 640 and step 1."""
    v = np.array([319.9, 320.0, -320.0, 959.9, 0.0, -319.9, 1e6], np.float32)
    assert native.quant_host(v, 640).tolist() == [0.0, 640.0, -640.0, 640.0, 0.0, -0.0, 1000320.0]
    w = np.array([2.5, -2.5, 7.0, -0.4, 0.5], np.float32)
    assert native.quant_host(w, 1).tolist() == [3.0, -3.0, 7.0, -0.0, 1.0]


@pytest.mark.parametrize("step", [1, 3, 7, 640, 65535])
def test_quant_bit_exact(native, oracle, step):
    rng = np.random.default_rng(step)
    c = (rng.standard_normal(100003) * 900).astype(np.float32)
    c[:7] = [0.0, -0.0, 0.5 * step, -0.5 * step, 1.5 * step, 4080.0, -4080.0]
    got, exp = native.quant_host(c, step), oracle.quant(c, step)
    assert got.tobytes() == exp.tobytes()


@pytest.mark.parametrize("block", [8, 16])
def test_dct_quant_fused(native, oracle, block):
    """Fused DCT+quant == oracle quant applied to the device DCT (bit-exact), and the
    device DCT itself is within tolerance of the f64 oracle."""
    rng = np.random.default_rng(11)
    h, w = 64, 96
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    types = rng.integers(0, 3, (h // 16) * (w // 16)).astype(np.uint32)
    dct = native.dct_host(bgr, block)
    exp = oracle.quant_frame(dct, 16, 16, types, 2, 640)
    got = native.dct_quant_host(bgr, block, types, 16, 2, 640)
    assert got.tobytes() == exp.tobytes()
    planes = torch.from_numpy(dct).cuda().unsqueeze(0).contiguous()
    native.quant_frames_(planes, torch.from_numpy(types.astype(np.int32)).cuda().unsqueeze(0).contiguous(), 16, 2, 640)
    assert planes[0].cpu().numpy().tobytes() == exp.tobytes()


GENERAL_BLOCKS = [(4, 4), (2, 2), (16, 8), (8, 16), (32, 32), (2, 16), (64, 64), (8, 1), (1, 4), (6, 10), (8, 8), (16, 16)]


@pytest.mark.parametrize("bw,bh", GENERAL_BLOCKS)
def test_dct_general_blocks(native, oracle, bw, bh):
    """static Dct takes ANY block_w x block_h (libs/encoder.cpp:323-339) and Validate admits every transform block that
    divides the MV block (:62-142): 4x4, 2x2, non-square, 32x32 on 32x32 MV blocks, single rows / columns.  8x8 / 16x16
    on a frame that is NOT a multiple of 16 wide takes the general kernel too."""
    rng = np.random.default_rng(bw * 100 + bh)
    w = {(8, 8): 88, (16, 16): 112}.get((bw, bh), bw * max(1, 200 // bw))  # 88 = 8 * 11: not a multiple of 16
    h = bh * max(2, 70 // bh)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    bgr[:bh, :bw] = 255  # a saturated tile: the largest DC
    got = native.dct_host(bgr, (bw, bh))
    _dct_close(got, oracle.dct_frame_f64(bgr, bw, bh))
    # energy per tile (orthonormal) and the DC of the saturated tile
    e_in = (bgr.astype(np.float64) ** 2).sum()
    assert abs(e_in - (got.astype(np.float64) ** 2).sum()) <= 1e-6 * e_in
    assert np.allclose(got[:, 0, 0], 255.0 * np.sqrt(bw * bh), rtol=1e-6)


@pytest.mark.parametrize("bw,bh,mv", [(4, 4, 16), (16, 8, 16), (8, 16, 16), (2, 2, 16), (32, 32, 32), (4, 8, (8, 16))])
def test_dct_quant_general_blocks(native, oracle, bw, bh, mv):
    """Fused DCT + quant on the general kernel == the oracle's quant lines applied to the device DCT, bit for bit;
    batched device entry point == the host one; the reference's serialiser runs on the result (wire records)."""
    rng = np.random.default_rng(bw + 7 * bh)
    mvw, mvh = (mv, mv) if isinstance(mv, int) else mv
    h, w = 3 * mvh * 2, 5 * mvw * 2
    frames = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    types = rng.integers(0, 3, (2, (h // mvh) * (w // mvw))).astype(np.uint32)
    dev = torch.device("cuda")
    bgr = torch.from_numpy(frames).to(dev)
    t = torch.from_numpy(types.astype(np.int32)).to(dev)
    n, hh, ww, _ = bgr.shape
    planes = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=dev)
    native._check(native.load().svc_hip_dct_quant_frames(bgr.data_ptr(), hh * ww * 3, n, ww, hh, bw, bh, t.data_ptr(), mvw, mvh,
                                                         3, 640, planes.data_ptr(), torch.cuda.current_stream().cuda_stream))
    got = planes.cpu().numpy()
    for f in range(2):
        dct = native.dct_host(frames[f], (bw, bh))
        exp = oracle.quant_frame(dct, mvw, mvh, types[f], 3, 640)
        assert got[f].tobytes() == exp.tobytes()
        assert native.dct_quant_host(frames[f], (bw, bh), types[f], (mvw, mvh), 3, 640).tobytes() == exp.tobytes()
    # SerializeEncodedFrame (libs/encoder.cpp:222-269) on these planes; its tile loops swap w / h for the rows, so the
    # reference itself only handles square transform blocks consistently -- serialise the square cases
    if bw == bh and mvw == mvh:
        rec = native.serialize_frames(planes, t, ww, hh, bw, bh, ww // mvw, hh // mvh, mvw).cpu().numpy()
        exp = oracle.serialize_frame(got[0], types[0], ww, hh, bw, bh, ww // mvw, mvw, mvh)
        assert rec[0].tobytes() == exp.tobytes()


def test_dct_block_preconditions(native):
    for block, status in (((3, 4), native.SVC_ERR_INVALID_ARG),      # cv::dct: odd sizes are not implemented
                          ((1, 1), native.SVC_ERR_INVALID_ARG),
                          ((128, 128), native.SVC_ERR_UNSUPPORTED)):  # beyond the general kernel's 64 x 64
        with pytest.raises(native.SvcError) as e:
            native.dct_host(np.zeros((256, 384, 3), np.uint8), block)
        assert e.value.status == status, block
    with pytest.raises(native.SvcError) as e:
        native.dct_host(np.zeros((30, 32, 3), np.uint8), 8)
    assert e.value.status == native.SVC_ERR_INVALID_ARG
    with pytest.raises(native.SvcError) as e:  # the fused record emitter takes square transform blocks up to 64 x 64
        native.dct_records_frames(torch.zeros((1, 256, 256, 3), dtype=torch.uint8, device="cuda"), 128,
                                  torch.zeros((1, 4), dtype=torch.int32, device="cuda"), mv_block=128)
    assert e.value.status == native.SVC_ERR_UNSUPPORTED
    native.dct_records_frames(torch.zeros((1, 32, 32, 3), dtype=torch.uint8, device="cuda"), 4,
                              torch.zeros((1, 4), dtype=torch.int32, device="cuda"))  # 4 x 4: the general kernel emits records


def test_random_transform_shapes(native, oracle):
    """Seeded random transform blocks -- every even side up to 64, single rows and columns of even length -- on random frame sizes they
    divide: within the parity tolerance of the f64 DCT-II, energy preserved, and the fused quantiser bit-identical to the oracle's quant
    lines applied to the device DCT."""
    rng = np.random.default_rng(424242)
    sides = [1] + list(range(2, 66, 2))
    for _ in range(40):
        bw, bh = int(rng.choice(sides)), int(rng.choice(sides))
        if bw == 1 and bh == 1:
            continue  # cv::dct has no 1 x 1 transform
        if (bw == 1 or bh == 1) and rng.random() < 0.7:
            bw = bh = int(rng.choice(sides[1:]))  # mostly two-dimensional blocks
        w, h = bw * int(rng.integers(1, max(2, 160 // bw))), bh * int(rng.integers(1, max(2, 120 // bh)))
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = native.dct_host(bgr, (bw, bh))
        _dct_close(got, oracle.dct_frame_f64(bgr, bw, bh))
        e_in = (bgr.astype(np.float64) ** 2).sum()
        assert abs(e_in - (got.astype(np.float64) ** 2).sum()) <= 1e-6 * max(e_in, 1.0), (bw, bh, w, h)
        mvw, mvh = bw * int(rng.integers(1, 3)), bh * int(rng.integers(1, 3))
        if w % mvw or h % mvh:
            continue
        types = rng.integers(0, 3, (h // mvh) * (w // mvw)).astype(np.uint32)
        fg, bg = int(rng.choice([1, 2, 5])), int(rng.choice([16, 640]))
        exp = oracle.quant_frame(got, mvw, mvh, types, fg, bg)
        assert native.dct_quant_host(bgr, (bw, bh), types, (mvw, mvh), fg, bg).tobytes() == exp.tobytes(), (bw, bh, w, h, mvw, mvh)


@pytest.mark.parametrize("block,w,h,levels,mvb", [(8, 160, 96, 3, 16), (16, 160, 96, 3, 16), (8, 1920, 64, 2, 16), (16, 352, 288, 4, 16), (8, 256, 128, 3, 32),
                                                   (16, 256, 128, 2, 32)])
@pytest.mark.parametrize("fg_share", [0.0, 0.03, 0.5, 1.0])
def test_speculative_quant_plus_redo_equals_dct_quant(native, block, w, h, levels, mvb, fg_share):
    """One pass over the BGR bytes (every tile quantised as background + the luma plane) followed by the redo of the foreground MV
    blocks' tiles leaves exactly the planes of svc_hip_dct_quant_frames with the region ids up front (libs/encoder.cpp:323-339,
    libs/decoder.cpp:130-144) and the pyramid of svc_hip_luma_pyramid_frames -- for no, few, many and only foreground blocks."""
    rng = np.random.default_rng(block + w + int(fg_share * 100))
    n = 3
    bgr = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
    mfw, mfh = w // mvb, h // mvb
    ids = rng.integers(1, 40, (n, mfw * mfh)) * (rng.random((n, mfw * mfh)) < fg_share)
    types = torch.from_numpy(ids.astype(np.int32)).cuda()
    for fg, bg in ((1, 640), (3, 17)):
        want = native.dct_quant_frames(bgr, block, types, mvb, fg, bg)
        want_pyr, stride = native.luma_pyramid_frames(bgr, levels)
        planes, pyr, stride2 = native.dct_quant_luma_frames(bgr, block, levels, bg_step=bg)
        torch.cuda.synchronize()
        assert torch.equal(planes, native.dct_quant_frames(bgr, block, torch.zeros_like(types), mvb, fg, bg))  # all background so far
        used = sum((w >> l) * (h >> l) for l in range(levels))
        for f in range(n):
            assert torch.equal(pyr[f * stride:f * stride + used], want_pyr[f * stride:f * stride + used]), f
        native.dct_quant_redo_frames(bgr, planes, block, types, mvb, fg_step=fg)
        torch.cuda.synchronize()
        assert torch.equal(planes, want), (fg, bg)


def test_speculative_quant_refuses_what_it_does_not_cover(native):
    bgr = torch.zeros((1, 64, 160, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="speculative form"):
        native.dct_quant_luma_frames(bgr, 4, 1)
    planes = torch.zeros((1, 3, 64, 160), dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="whole"):
        native.dct_quant_redo_frames(bgr, planes, 8, torch.zeros((1, 160), dtype=torch.int32, device="cuda"), mv_block=8)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 255, 1000, 8160 * 7, 2439840])
def test_count_foreground(native, n):
    """svc_hip_count_foreground: how many region ids of a batch are not 0 (what the driver's speculation policy runs on)."""
    rng = np.random.default_rng(n)
    ids = torch.from_numpy((rng.integers(1, 300, n) * (rng.random(n) < 0.13)).astype(np.int32)).cuda()
    assert native.count_foreground(ids) == int((ids != 0).sum())
    assert native.count_foreground(torch.zeros_like(ids)) == 0
    assert native.count_foreground(torch.ones_like(ids)) == n


def test_one_pass_forms_on_random_shapes(native):
    """Seeded random sweep of the calls that read a frame's bytes once (round 5): widths of whole 16-pixel segments, heights of whole MV
    blocks, 1-4 pyramid levels where the size divides, 8 / 16 transform blocks, MV blocks 16 / 32 / 48 / 64 wide and 8 ... 64 tall, any
    foreground share, emitted heights below the frame's -- records + type patch == svc_hip_dct_records_frames, speculative planes + redo ==
    svc_hip_dct_quant_frames, and both leave svc_hip_luma_pyramid_frames' pyramid."""
    rng = np.random.default_rng(20261005)
    done = 0
    while done < 30:
        block = int(rng.choice([8, 16]))
        mvw = int(rng.choice([16, 32, 48, 64]))
        mvh = int(rng.choice([m for m in (8, 16, 32, 48, 64) if m % block == 0]))
        levels = int(rng.integers(1, 5))
        f = 1 << (levels - 1)
        w = mvw * int(rng.integers(1, 9))
        h = mvh * int(rng.integers(1, 7))
        if w % f or h % f or (w >> (levels - 1)) < 3 or (h >> (levels - 1)) < 3 or mvw % block:
            continue
        n = int(rng.integers(1, 4))
        bgr = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        mfw, mfh = w // mvw, h // mvh
        share = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
        types = torch.from_numpy((rng.integers(1, 500, (n, mfw * mfh)) * (rng.random((n, mfw * mfh)) < share)).astype(np.int32)).cuda()
        fg, bg = int(rng.integers(1, 9)), int(rng.integers(2, 900))
        emit_h = h - int(rng.integers(0, 2)) * min(block, h - 1)
        lib = native.load()
        want_pyr, stride = native.luma_pyramid_frames(bgr, levels)
        used = sum((w >> l) * (h >> l) for l in range(levels))
        # records
        per = native.serialized_frame_bytes(w, emit_h, block, block)
        want_rec = torch.empty((n, per), dtype=torch.uint8, device="cuda")
        native._check(lib.svc_hip_dct_records_frames(bgr.data_ptr(), h * w * 3, n, w, h, block, types.data_ptr(), mvw, mvh, 0, 0, emit_h,
                                                     want_rec.data_ptr(), per, None))
        rec, pyr, _ = native.dct_records_luma_frames(bgr, block, levels, emit_h=emit_h)
        native._check(lib.svc_hip_wire_patch_types_frames(types.data_ptr(), n, w, h, emit_h, block, mvw, mvh, rec.data_ptr(), per, 0, None))
        # planes
        want = torch.empty((n, 3, h, w), dtype=torch.float32, device="cuda")
        native._check(lib.svc_hip_dct_quant_frames(bgr.data_ptr(), h * w * 3, n, w, h, block, block, types.data_ptr(), mvw, mvh, fg, bg, want.data_ptr(), None))
        planes, pyr2, _ = native.dct_quant_luma_frames(bgr, block, levels, bg_step=bg)
        nbytes = lib.svc_hip_dct_redo_workspace_bytes(n, w, h, mvw, mvh)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        native._check(lib.svc_hip_dct_quant_redo_frames(bgr.data_ptr(), h * w * 3, n, w, h, block, types.data_ptr(), mvw, mvh, fg, planes.data_ptr(),
                                                        ws.data_ptr(), nbytes, None))
        torch.cuda.synchronize()
        tag = (block, mvw, mvh, levels, w, h, n, share, fg, bg, emit_h)
        assert torch.equal(rec, want_rec), tag
        assert torch.equal(planes, want), tag
        for f_ in range(n):
            assert torch.equal(pyr[f_ * stride:f_ * stride + used], want_pyr[f_ * stride:f_ * stride + used]), tag
            assert torch.equal(pyr2[f_ * stride:f_ * stride + used], want_pyr[f_ * stride:f_ * stride + used]), tag
        done += 1
