"""The per-call image operations of include/svc_hip.h (what compat/opencv2/ forwards the reference's cv:: calls to),
each against its statement in oracle/svc_imageops.c, and composed the way libs/encoder.cpp:507-623 composes the cv::
calls against the fused svc_hip_segment_frames."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(64, 48), (1920, 1088), (33, 7), (5, 3)])
def test_bgr2yuv(native, oracle, w, h):
    rng = np.random.default_rng(w * 31 + h)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    bgr[0, :min(w, 4)] = [[255, 0, 0], [0, 0, 255], [0, 255, 0], [255, 255, 255]][:min(w, 4)]  # saturating corners of U / V
    got = native.bgr2yuv_host(bgr)
    want = oracle.bgr2yuv(bgr)
    assert np.array_equal(got, want)
    assert np.array_equal(got[..., 0], oracle.luma(bgr))  # channel 0 is the luma the fused pyramid kernel computes


@pytest.mark.parametrize("w,h,levels", [(1920, 1088, 3), (1920, 1088, 4), (3840, 2160, 4), (352, 288, 1), (64, 32, 2), (128, 64, 4),
                                        (720, 576, 4), (176, 144, 4), (48, 16, 4)])  # the last three: level planes 90 / 22 / 6 pixels wide
def test_build_pyramid(native, oracle, w, h, levels):
    rng = np.random.default_rng(levels * 7 + w)
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    planes = native.build_pyramid_host(y, levels)
    src = y
    for l in range(1, levels):
        want = oracle.pyr_down(src)
        assert np.array_equal(planes[l], want), f"level {l}"
        src = want
    with pytest.raises(native.SvcError) as e:
        native.build_pyramid_host(y[:h - 1], levels + 1)
    assert e.value.status == native.SVC_ERR_INVALID_ARG


@pytest.mark.parametrize("w,h,kw,kh", [(120, 68, 3, 3), (240, 135, 3, 3), (22, 18, 5, 2), (120, 68, 1, 4), (7, 5, 9, 9)])
def test_morph_rect(native, oracle, w, h, kw, kh):
    rng = np.random.default_rng(kw * 13 + w)
    binary = np.where(rng.random((h, w)) < 0.4, 255, 0).astype(np.uint8)
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for img in (binary, grey):
        for op in (native.MORPH_ERODE, native.MORPH_DILATE, native.MORPH_OPEN, native.MORPH_CLOSE):
            assert np.array_equal(native.morph_rect_host(img, kw, kh, op), oracle.morph_rect(img, kw, kh, op)), (op, kw, kh)


@pytest.mark.parametrize("n,dims,k,kw", [(600, 4, 10, {}), (8160, 4, 10, {}), (30000, 4, 10, {}), (5, 4, 5, {}), (64, 1, 3, {}),
                                          (2000, 2, 64, dict(attempts=1, max_iter=30)), (900, 3, 7, dict(epsilon=40.0)),
                                          (1500, 4, 1, {}), (777, 4, 12, dict(max_iter=1, attempts=16)),
                                          (3000, 4, 200, dict(attempts=2)), (400, 2, 5, dict(attempts=40, max_iter=3))])  # more clusters / attempts than the fused path takes
def test_kmeans(native, oracle, n, dims, k, kw):
    rng = np.random.default_rng(n + dims)
    f = np.zeros((n, dims), np.float32)
    centres = rng.integers(-3000, 3000, (max(k, 2), dims))
    which = rng.integers(0, len(centres), n)
    f[:] = centres[which] + rng.integers(-40, 41, (n, dims))
    if dims == 4:
        f[:, 0] = 0  # the reference's features: (0, mv.x, x_px, y_px)
    if n == 5:
        f[1] = f[0]  # coinciding points: a zero k-means++ total before all centres are drawn
        f[3] = f[0]
        f[4] = f[0]
    got, gc = native.kmeans_host(f, k, seed=99, **kw)
    want, wc = oracle.kmeans(f, k, seed=99, **kw)
    assert np.array_equal(got, want)
    assert gc == wc
    assert got.min() >= 0 and got.max() < k


def test_kmeans_rejects_what_the_definition_does_not_cover(native):
    f = np.zeros((10, 4), np.float32)
    f[3, 1] = 0.5
    with pytest.raises(native.SvcError) as e:
        native.kmeans_host(f, 2)
    assert e.value.status == native.SVC_ERR_UNSUPPORTED and "integer" in str(e.value)
    f[3, 1] = 40000.0
    with pytest.raises(native.SvcError):
        native.kmeans_host(f, 2)
    with pytest.raises(native.SvcError) as e:
        native.kmeans_host(np.zeros((3, 4), np.float32), 4)  # cv::kmeans asserts N >= K
    assert e.value.status == native.SVC_ERR_INVALID_ARG


@pytest.mark.parametrize("w,h,density", [(120, 68, 0.3), (120, 68, 0.62), (240, 135, 0.55), (22, 18, 0.9), (513, 7, 0.5), (1, 1, 1.0)])
@pytest.mark.parametrize("conn", [4, 8])
def test_connected_components(native, oracle, w, h, density, conn):
    rng = np.random.default_rng(w + conn)
    img = np.where(rng.random((h, w)) < density, rng.integers(1, 256, (h, w)), 0).astype(np.uint8)
    got, gn = native.connected_components_host(img, conn)
    want, wn = oracle.connected_components(img, conn)
    assert gn == wn and np.array_equal(got, want)
    empty, en = native.connected_components_host(np.zeros((h, w), np.uint8), conn)
    assert en == 1 and not empty.any()


@pytest.mark.parametrize("w,h,bw,bh", [(64, 48, 8, 8), (1920, 1088, 8, 8), (256, 128, 16, 16), (64, 64, 64, 64), (48, 32, 4, 2),
                                       (96, 16, 32, 16), (40, 6, 8, 1)])
def test_dct_tiles(native, w, h, bw, bh):
    """cv::dct over a tile list == the float64 orthonormal DCT-II (scipy) within 1e-4 max(1, |ref|); tiles that are not
    listed keep their samples; the regular-grid form == listing every tile."""
    from scipy.fft import dctn
    rng = np.random.default_rng(w + bw)
    img = rng.integers(0, 256, (h, w)).astype(np.float32)
    nx, ny = w // bw, h // bh
    full = native.dct_tiles_host(img, bw, bh)
    ref = np.empty((h, w))
    for ty in range(ny):
        for tx in range(nx):
            t = img[ty * bh:(ty + 1) * bh, tx * bw:(tx + 1) * bw].astype(np.float64)
            ref[ty * bh:(ty + 1) * bh, tx * bw:(tx + 1) * bw] = dctn(t, type=2, norm="ortho")
    assert (np.abs(full - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    xy = np.array([(tx * bw, ty * bh) for ty in range(ny) for tx in range(nx)], np.uint32)
    assert native.dct_tiles_host(img, bw, bh, xy).tobytes() == full.tobytes()
    some = xy[rng.permutation(len(xy))[:max(1, len(xy) // 3)]]
    part = native.dct_tiles_host(img, bw, bh, some)
    done = np.zeros((h, w), bool)
    for x, y in some:
        done[y:y + bh, x:x + bw] = True
    assert np.array_equal(part[done], full[done]) and np.array_equal(part[~done], img[~done])
    with pytest.raises(native.SvcError):
        native.dct_tiles_host(img, bw, bh, np.array([[w - bw + 1, 0]], np.uint32))


def test_dct_tiles_equal_the_frame_kernels_input_for_input(native):
    """The f32-plane tile form on a converted + split frame (what the reference's Dct does with cv::split + cv::dct)
    against the u8 frame form: same coefficients within the parity tolerance (both round an f64 result once)."""
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    for b in (8, 16, 4):
        planes = native.dct_planes_host(bgr, b, b)
        for c in range(3):
            t = native.dct_tiles_host(bgr[..., c].astype(np.float32), b, b)
            assert (np.abs(t - planes[c]) <= 1e-4 * np.maximum(1.0, np.abs(planes[c]))).all()
        packed = np.empty((3, 64, 96), np.float32)
        native._check(native.load().svc_hip_dct_host(bgr.ctypes.data, 96, 64, b, b, packed.ctypes.data))
        assert packed.tobytes() == planes.tobytes()  # the planes form is the packed form, D2H into three buffers


@pytest.mark.parametrize("mfw,mfh,density,conn", [(120, 68, 0.25, 4), (120, 68, 0.8, 8), (240, 135, 0.5, 4), (22, 18, 0.1, 4)])
def test_per_call_composition_equals_the_fused_segmentation(native, oracle, mfw, mfh, density, conn):
    """libs/encoder.cpp:507-623 run the reference's way -- morphologyEx x 2, kmeans, one connectedComponents per
    cluster, each a call into the per-call C ABI with the reference's glue in between -- gives the region ids of the
    fused device form svc_hip_segment_frames (and of oracle/svc_segment.c)."""
    rng = np.random.default_rng(mfw + conn)
    n = mfw * mfh
    mask = (~(rng.random((mfh, mfw)) < density)).astype(np.uint8).reshape(-1)  # 1 = RANSAC inlier
    mv = np.stack([rng.integers(-8, 9, n), rng.integers(-8, 9, n)], -1).astype(np.float32)
    seed = 4242

    class ByCalls:  # oracle.segment_by_calls with every arithmetic step replaced by its GPU entry point
        morph_rect = staticmethod(native.morph_rect_host)
        kmeans = staticmethod(lambda f, k, attempts, max_iter, epsilon, seed: native.kmeans_host(f, k, attempts, max_iter, epsilon, seed))
        connected_components = staticmethod(native.connected_components_host)
    got = type(oracle).segment_by_calls(ByCalls, np.nonzero(mask)[0], mv, mfw, mfh, connectivity=conn, seed=seed)
    want = oracle.segment(mask, mv, mfw, mfh, connectivity=conn, seed=seed)
    assert np.array_equal(got, want)
    fused = native.segment_frames(torch.from_numpy(mask[None]).cuda(), torch.from_numpy(mv[None]).cuda(), mfw, mfh, seed=seed,
                                  connectivity=conn).cpu().numpy()[0]
    assert np.array_equal(fused.astype(np.uint32), got)


def test_random_sweep_of_the_per_call_operations(native, oracle):
    """Seeded random arguments for every per-call operation against its statement in oracle/svc_imageops.c: sizes from 1 x 1 up, odd
    and degenerate structuring elements, every point dimension and cluster count the definition covers, both connectivities."""
    rng = np.random.default_rng(20260404)
    for _ in range(60):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        kw, kh = int(rng.integers(1, 8)), int(rng.integers(1, 8))
        op = int(rng.integers(0, 4))
        assert np.array_equal(native.morph_rect_host(img, kw, kh, op), oracle.morph_rect(img, kw, kh, op)), ("morph", w, h, kw, kh, op)
        binary = np.where(rng.random((h, w)) < rng.random(), img | 1, 0).astype(np.uint8)
        conn = int(rng.choice([4, 8]))
        got, gn = native.connected_components_host(binary, conn)
        want, wn = oracle.connected_components(binary, conn)
        assert gn == wn and np.array_equal(got, want), ("components", w, h, conn)
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(native.bgr2yuv_host(bgr), oracle.bgr2yuv(bgr)), ("bgr2yuv", w, h)
    for _ in range(40):
        n, dims = int(rng.integers(1, 4000)), int(rng.integers(1, 5))
        k = int(rng.integers(1, min(n, 40) + 1))
        f = rng.integers(-2000, 2000, (n, dims)).astype(np.float32)
        if rng.random() < 0.3:
            f[rng.integers(0, n, n // 2)] = f[0]  # many coinciding points
        kw = dict(attempts=int(rng.integers(1, 7)), max_iter=int(rng.integers(1, 15)), epsilon=float(rng.choice([0.5, 1.0, 30.0])))
        seed = int(rng.integers(0, 2 ** 40))
        got, gc = native.kmeans_host(f, k, seed=seed, **kw)
        want, wc = oracle.kmeans(f, k, seed=seed, **kw)
        assert np.array_equal(got, want) and gc == wc, ("kmeans", n, dims, k, kw, seed)
    for _ in range(25):
        levels = int(rng.integers(1, 5))
        f = 1 << (levels - 1)
        w, h = int(rng.integers(1, 40)) * f * 2, int(rng.integers(1, 30)) * f * 2
        if min(w, h) >> (levels - 1) < 3 and levels > 1:
            continue  # reflect-101 needs three pixels on the level that is reduced last
        y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        planes = native.build_pyramid_host(y, levels)
        src = y
        for l in range(1, levels):
            src = oracle.pyr_down(src)
            assert np.array_equal(planes[l], src), ("pyramid", w, h, levels, l)
