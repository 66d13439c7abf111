"""The CPU baseline's fast legs (oracle/svc_cpu_dct.c: f32 separable DCT with an AVX2 + FMA path, vectorised quantiser)
against the oracle of record, and the per-call oracle functions (oracle/svc_imageops.c) against the fused segmentation
oracle.  No GPU."""
import numpy as np
import pytest


@pytest.mark.parametrize("block", [8, 16])
@pytest.mark.parametrize("w,h", [(64, 48), (1920, 1088), (352, 288)])
def test_cpu_dct_within_the_parity_tolerance_of_the_f64_oracle(oracle, block, w, h):
    if w % block or h % block:
        pytest.skip("size not divisible")
    rng = np.random.default_rng(w + block)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    bgr[:block, :block] = 255      # the largest DC term (2040 / 4080)
    bgr[:block, block:2 * block] = 0
    bgr[block:2 * block, :block, 0] = (np.arange(block) % 2 * 255)[None, :]  # the highest horizontal frequency
    ref = oracle.dct_frame_f64(bgr, block, block)
    got = oracle.cpu_dct_frame_f32(bgr, block)
    assert (np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    assert abs(got[0, 0, 0] - 255.0 * block) < 1e-2 and np.abs(got[0, 0, 1:block]).max() < 1e-3  # constant tile -> DC only
    with pytest.raises(ValueError):
        oracle.cpu_dct_frame_f32(bgr, 4)


def test_cpu_quant_is_the_oracle_quant_bit_for_bit(oracle):
    rng = np.random.default_rng(5)
    planes = (rng.standard_normal((3, 64, 96)) * 700).astype(np.float32)
    planes[0, 0, :8] = [319.9, 320.0, -320.0, 959.9, 2.5, -2.5, 0.49999997, 0.5]  # SURVEY 8c's hand vectors + the tie trick's edge
    planes[1, 0, :4] = [8388609.0, -8388609.0, 1e20, -0.0]
    types = rng.integers(0, 3, (64 // 16) * (96 // 16)).astype(np.uint32)
    for fg, bg in ((1, 640), (3, 7), (65535, 1), (640, 640)):
        want = oracle.quant_frame(planes, 16, 16, types, fg, bg)
        got = oracle.cpu_quant_frame_f32(planes.copy(), 16, 16, types, fg, bg)
        assert got.tobytes() == want.tobytes(), (fg, bg)


@pytest.mark.parametrize("mfw,mfh,density,conn", [(120, 68, 0.4, 4), (120, 68, 0.9, 8), (33, 31, 0.7, 4), (22, 18, 0.05, 8), (240, 135, 0.5, 4)])
def test_per_call_oracle_composition_equals_the_fused_oracle(oracle, mfw, mfh, density, conn):
    """libs/encoder.cpp:507-623 composed from svc_oracle_morph_rect / _kmeans / _connected_components with the reference's glue
    (BuildMvFeatures' (0, m.x, x, y), one labelling per cluster, offset += count) == svc_oracle_segment."""
    rng = np.random.default_rng(mfw + conn)
    n = mfw * mfh
    mask = (~(rng.random((mfh, mfw)) < density)).astype(np.uint8).reshape(-1)
    mv = np.stack([rng.integers(-8, 9, n), rng.integers(-8, 9, n)], -1).astype(np.float32)
    want = oracle.segment(mask, mv, mfw, mfh, connectivity=conn, seed=7)
    got = oracle.segment_by_calls(np.nonzero(mask)[0], mv, mfw, mfh, connectivity=conn, seed=7)
    assert np.array_equal(want, got)


def test_per_call_oracle_known_answers(oracle):
    img = np.zeros((5, 7), np.uint8)
    img[1:4, 1:3] = 9
    img[0, 6] = 1
    img[4, 4] = 200
    img[3, 3] = 1  # touches the first blob's corner (2, 2)... diagonal of (3, 2)? no: 4-neighbour of (3, 2)
    lab4, n4 = oracle.connected_components(img, 4)
    assert n4 == 4 and lab4[0, 6] == 1 and lab4[1, 1] == 2 and lab4[3, 3] == 2 and lab4[4, 4] == 3
    lab8, n8 = oracle.connected_components(img, 8)
    assert n8 == 3 and lab8[4, 4] == lab8[1, 1] == 2  # (4, 4) is a diagonal neighbour of (3, 3)
    one = np.zeros((6, 6), np.uint8)
    one[2, 3] = 255
    assert oracle.morph_rect(one, 3, 3, 1).sum() == 9 * 255 and oracle.morph_rect(one, 3, 3, 0).sum() == 0
    assert np.array_equal(oracle.morph_rect(one, 3, 3, 3), one) and not oracle.morph_rect(one, 3, 3, 2).any()
    bgr = np.array([[[255, 0, 0], [0, 0, 255], [0, 255, 0], [128, 128, 128]]], np.uint8)
    yuv = oracle.bgr2yuv(bgr)
    assert yuv[0, :, 0].tolist() == [29, 76, 150, 128] and yuv[0, 3].tolist() == [128, 128, 128]
    assert yuv[0, 0, 1] > 200 and yuv[0, 1, 2] == 255 and yuv[0, 2, 1] < 60  # blue -> high U, red -> V saturates, green -> low U
    pts = np.array([[0, 0], [1, 0], [0, 1], [100, 100], [101, 100], [100, 101]], np.float32)
    labels, compact = oracle.kmeans(pts, 2, seed=3)
    assert len(set(labels[:3])) == 1 and len(set(labels[3:])) == 1 and labels[0] != labels[3]
    assert abs(compact - 6 * (1 / 9 + 4 / 9) * 1.0 - 0.0) < 1.0
    with pytest.raises(ValueError):
        oracle.kmeans(np.array([[0.5, 0.0]], np.float32), 1)
