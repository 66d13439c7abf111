"""The three whole-frame estimators of libs/motion.hpp:38-59 on the GPU, through the C ABI and through the C++ symbols
of libsvc_motion.so.  EstimateGlobalMotionAvg: bit-exact against reference-produced goldens.  The exhaustive /
hierarchical searches: the reference's own loops never run for R > 0 (libs/motion.cpp:72, :81 -- pinned in
tests/test_oracle_vs_ref.py and the golden), so the product implements the search as evidently meant and is checked
against the oracle's restatement of THAT (reference_loop=False), and against the reference's literal output at R = 0."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu


def test_global_avg_golden_and_random(native, oracle):
    avgs, _, _, _ = G.global_motion_cases()
    for mv, want in avgs:
        assert native.global_avg_host(mv).tobytes() == want.tobytes()
    rng = np.random.default_rng(9)
    for n in (3, 1023, 1024, 1025, 32400):
        mv = (rng.standard_normal((n, 2)) * 9).astype(np.float32)
        assert native.global_avg_host(mv).tobytes() == oracle.global_avg(mv).tobytes()
    batch = (rng.standard_normal((5, 777, 2)) * 3).astype(np.float32)
    got = native.global_avg_frames(torch.from_numpy(batch).cuda()).cpu().numpy()
    for f in range(5):
        assert got[f].tobytes() == oracle.global_avg(batch[f]).tobytes()


def test_global_ebma_reference_literal_r0(native):
    """R = 0 is the one search range at which the reference's loop runs: same single candidate, same MAD."""
    _, t, a, literal = G.global_motion_cases()
    gm, mad = native.global_ebma_host(t, a, 0)
    assert np.array([gm[0], gm[1], mad], np.float32).tobytes() == literal[0].tobytes()


@pytest.mark.parametrize("w,h,r", [(96, 64, 4), (96, 64, 8), (97, 61, 5), (130, 70, 20), (64, 48, 0), (1920, 1088, 8), (30, 20, 19),
                                   (256, 64, 1)])
def test_global_ebma_vs_oracle(native, oracle, w, h, r):
    rng = np.random.default_rng(w * 7 + r)
    base = rng.integers(0, 256, (h + 64, w + 64), dtype=np.uint8)
    dx, dy = (int(v) for v in rng.integers(-min(r, 6), min(r, 6) + 1, 2)) if r else (0, 0)
    t = np.ascontiguousarray(base[32:32 + h, 32:32 + w])
    a = np.ascontiguousarray(base[32 + dy:32 + dy + h, 32 + dx:32 + dx + w])   # tracked(y, x) = anchor(y - dy, x - dx)
    a = (a.astype(np.int16) + rng.integers(-1, 2, a.shape)).clip(0, 255).astype(np.uint8)
    gm, mad = native.global_ebma_host(t, a, r)
    gm_o, mad_o = oracle.global_ebma(t, a, r)
    assert gm.tobytes() == gm_o.tobytes() and np.float32(mad).tobytes() == np.float32(mad_o).tobytes()
    assert gm.tolist() == [float(dx), float(dy)]
    # flat frames: every candidate ties at 0 -> the first in raster order, (-r, -r)
    flat = np.full((h, w), 77, np.uint8)
    gm, mad = native.global_ebma_host(flat, flat, r)
    assert gm.tolist() == [float(-r), float(-r)] and mad == 0.0


def test_global_ebma_batched_device(native, oracle):
    rng = np.random.default_rng(2)
    w, h, r, n = 128, 72, 6, 3
    planes = rng.integers(0, 256, (n + 1, h, w), dtype=np.uint8)
    planes[2] = np.roll(planes[1], (2, -3), (0, 1))
    d = torch.from_numpy(planes).cuda().reshape(-1)
    gm, mad = native.global_ebma_pairs(d, d[w * h:], w * h, n, w, h, r)
    for p in range(n):
        g, m = oracle.global_ebma(planes[p], planes[p + 1], r)
        assert gm[p].cpu().numpy().tobytes() == g.tobytes() and np.float32(mad[p].item()).tobytes() == np.float32(m).tobytes()


@pytest.mark.parametrize("levels,r", [(1, 5), (2, 8), (3, 8), (3, 3), (4, 16)])
def test_global_hbma_vs_oracle(native, oracle, levels, r):
    rng = np.random.default_rng(levels * 10 + r)
    w, h = 256, 160
    base = rng.integers(0, 256, ((h + 64) // 8, (w + 64) // 8), dtype=np.uint8)
    big = np.kron(base, np.ones((8, 8), np.uint8))  # blocky: the coarse levels still carry the shift
    t0 = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
    a0 = np.ascontiguousarray(big[32 - 4:32 - 4 + h, 32 + 8:32 + 8 + w])
    t = [np.ascontiguousarray(t0[:: 1 << l, :: 1 << l]) for l in range(levels)]
    a = [np.ascontiguousarray(a0[:: 1 << l, :: 1 << l]) for l in range(levels)]
    assert native.global_hbma_host(t, a, r).tobytes() == oracle.global_hbma(t, a, r).tobytes()


def test_global_motion_preconditions(native):
    z = np.zeros((16, 16), np.uint8)
    for args in ((z, z, 16), (z, z, 40)):  # motion.cpp:63-64; at R == side the overlap is empty
        with pytest.raises(native.SvcError) as e:
            native.global_ebma_host(*args)
        assert e.value.status == native.SVC_ERR_INVALID_ARG


def test_cpp_symbols_of_the_reference_header(native, oracle):
    """The mangled C++ entry points (libs/motion.hpp:38, :45-49, :55-59) in libsvc_motion.so."""
    lib = C.CDLL(native.MOTION_LIB_PATH)
    rng = np.random.default_rng(4)
    mv = (rng.standard_normal((500, 2)) * 5).astype(np.float32)

    class V(C.Structure):
        _fields_ = [("x", C.c_float), ("y", C.c_float)]
    f = lib._Z23EstimateGlobalMotionAvgPK5Vec2fj
    f.restype, f.argtypes = V, [C.c_void_p, C.c_uint]
    v = f(mv.ctypes.data, len(mv))
    assert np.array([v.x, v.y], np.float32).tobytes() == oracle.global_avg(mv).tobytes()
    t = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    a = np.roll(t, (1, 2), (0, 1))
    g = lib._Z36EstimateGlobalMotionExhaustiveSearchPKhS0_jjjP5Vec2fPf
    g.restype, g.argtypes = None, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
    gm, mad = np.zeros(2, np.float32), np.zeros(1, np.float32)
    g(t.ctypes.data, a.ctypes.data, 64, 48, 3, gm.ctypes.data, mad.ctypes.data)
    go, mo = oracle.global_ebma(t, a, 3)
    assert gm.tobytes() == go.tobytes() and mad[0].tobytes() == np.float32(mo).tobytes()
    hfn = lib._Z32EstimateGlobalMotionHierarchicalPKPKhS2_jjjjP5Vec2f
    hfn.restype, hfn.argtypes = None, [C.c_void_p, C.c_void_p] + [C.c_uint] * 4 + [C.c_void_p]
    tp = [np.ascontiguousarray(t[:: 1 << l, :: 1 << l]) for l in range(2)]
    ap = [np.ascontiguousarray(a[:: 1 << l, :: 1 << l]) for l in range(2)]
    tpp = (C.c_void_p * 2)(*[x.ctypes.data for x in tp])
    app = (C.c_void_p * 2)(*[x.ctypes.data for x in ap])
    hfn(tpp, app, 2, 64, 48, 4, gm.ctypes.data)
    assert gm.tobytes() == oracle.global_hbma(tp, ap, 4).tobytes()


def test_reference_literal_switch_reproduces_the_reference(native):
    """SvcReferenceLiteralGlobalSearch(true): the two whole-frame searches answer what the unmodified reference answers
    (its `int dy <= uint search_range` loops never run, libs/motion.cpp:72, :81) -- the committed reference outputs of
    tests/golden/global_motion.npz -- and nothing else changes; off again, the search as meant."""
    lib = C.CDLL(native.MOTION_LIB_PATH)
    sw = lib._Z31SvcReferenceLiteralGlobalSearchb
    sw.restype, sw.argtypes = None, [C.c_bool]
    _, t, a, literal = G.global_motion_cases()
    h, w = t.shape
    g = lib._Z36EstimateGlobalMotionExhaustiveSearchPKhS0_jjjP5Vec2fPf
    g.restype, g.argtypes = None, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
    hfn = lib._Z32EstimateGlobalMotionHierarchicalPKPKhS2_jjjjP5Vec2f
    hfn.restype, hfn.argtypes = None, [C.c_void_p, C.c_void_p] + [C.c_uint] * 4 + [C.c_void_p]
    gm, mad = np.full(2, 9.0, np.float32), np.full(1, 9.0, np.float32)
    sw(True)
    try:
        for r, want in sorted(literal.items()):
            g(t.ctypes.data, a.ctypes.data, w, h, r, gm.ctypes.data, mad.ctypes.data)
            assert np.array([gm[0], gm[1], mad[0]], np.float32).tobytes() == want.tobytes(), r
            if r > 0:
                assert gm.tolist() == [0.0, 0.0] and mad[0] == np.finfo(np.float32).max
        tp = [np.ascontiguousarray(t[:: 1 << l, :: 1 << l]) for l in range(2)]
        ap = [np.ascontiguousarray(a[:: 1 << l, :: 1 << l]) for l in range(2)]
        tpp = (C.c_void_p * 2)(*[x.ctypes.data for x in tp])
        app = (C.c_void_p * 2)(*[x.ctypes.data for x in ap])
        gm[:] = 9.0
        hfn(tpp, app, 2, w, h, 4, gm.ctypes.data)
        assert gm.tolist() == [0.0, 0.0]
    finally:
        sw(False)
    g(t.ctypes.data, a.ctypes.data, w, h, 4, gm.ctypes.data, mad.ctypes.data)
    assert mad[0] < np.finfo(np.float32).max  # the search runs again
