"""ctypes doorway to the checkers: oracle/libsvc_oracle.so (C restatement) and,
when it has been built, oracle/_ref/libsvc_ref.so (the unmodified reference
libs/motion.cpp compiled in place).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from scalable_video_codec_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libsvc_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsvc_ref.so")

_u8p = C.POINTER(C.c_uint8)
_u8pp = C.POINTER(_u8p)
_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)


class RansacParams(C.Structure):
    _fields_ = [("subset_sz", C.c_uint32), ("inlier_thresh", C.c_float),
                ("success_prob", C.c_float), ("inlier_ratio", C.c_float)]


DEFAULT_RANSAC = dict(subset_sz=1, inlier_thresh=7.5, success_prob=0.99, inlier_ratio=0.5)


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def _pyr_ptrs(pyr: Sequence[np.ndarray]):
    arr = (_u8p * len(pyr))()
    for i, p in enumerate(pyr):
        assert p.dtype == np.uint8 and p.flags["C_CONTIGUOUS"]
        arr[i] = _ptr(p, _u8p)
    return arr


class Oracle:
    """The C restatement (svc_oracle.h)."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle oracle`")
        L = self.lib = C.CDLL(path)
        L.svc_oracle_mad.restype = C.c_float
        L.svc_oracle_mad.argtypes = [_u8p, _u8p] + [C.c_uint32] * 7
        L.svc_oracle_ebma.restype = None
        L.svc_oracle_ebma.argtypes = [_u8p, _u8p] + [C.c_uint32] * 5 + [_f32p, _f32p]
        L.svc_oracle_refine.restype = None
        L.svc_oracle_refine.argtypes = [_u8p, _u8p] + [C.c_uint32] * 5 + [_f32p, _f32p]
        L.svc_oracle_hbma.restype = C.c_int
        L.svc_oracle_hbma.argtypes = [_u8pp, _u8pp] + [C.c_uint32] * 6 + [_f32p, _f32p]
        L.svc_oracle_hbma16_sse2.restype = C.c_int
        L.svc_oracle_hbma16_sse2.argtypes = [_u8pp, _u8pp] + [C.c_uint32] * 3 + [_f32p, _f32p]
        L.svc_oracle_ransac_iter_count.restype = C.c_uint32
        L.svc_oracle_ransac_iter_count.argtypes = [RansacParams]
        L.svc_oracle_ransac.restype = None
        L.svc_oracle_ransac.argtypes = [_f32p, C.c_uint32, RansacParams, _u32p, C.c_uint32,
                                        _f32p, _f32p, _u32p, _u32p]
        L.svc_oracle_fg_mask.restype = None
        L.svc_oracle_fg_mask.argtypes = [_u32p, C.c_uint32, C.c_uint32, _u8p]
        L.svc_oracle_luma.restype = None
        L.svc_oracle_luma.argtypes = [_u8p, C.c_uint32, C.c_uint32, _u8p]
        L.svc_oracle_pyr_down.restype = None
        L.svc_oracle_pyr_down.argtypes = [_u8p, C.c_uint32, C.c_uint32, _u8p]
        L.svc_oracle_luma_pyramid.restype = None
        L.svc_oracle_luma_pyramid.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _u8p]
        L.svc_oracle_global_avg.restype = None
        L.svc_oracle_global_avg.argtypes = [_f32p, C.c_uint32, _f32p]
        L.svc_oracle_global_ebma.restype = None
        L.svc_oracle_global_ebma.argtypes = [_u8p, _u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _f32p, _f32p]
        L.svc_oracle_global_hbma.restype = None
        L.svc_oracle_global_hbma.argtypes = [_u8pp, _u8pp] + [C.c_uint32] * 4 + [C.c_int, _f32p]
        L.svc_oracle_segment.restype = C.c_int
        L.svc_oracle_segment.argtypes = [_u8p, _f32p] + [C.c_uint32] * 9 + [C.c_float, C.c_uint32, C.c_uint64, _u32p]
        _i32p = C.POINTER(C.c_int32)
        L.svc_oracle_bgr2yuv.restype = None
        L.svc_oracle_bgr2yuv.argtypes = [_u8p, C.c_uint32, C.c_uint32, _u8p]
        L.svc_oracle_morph_rect.restype = None
        L.svc_oracle_morph_rect.argtypes = [_u8p] + [C.c_uint32] * 5 + [_u8p]
        L.svc_oracle_kmeans.restype = C.c_int
        L.svc_oracle_kmeans.argtypes = [_f32p] + [C.c_uint32] * 5 + [C.c_float, C.c_uint64, _i32p, _f64p]
        L.svc_oracle_connected_components.restype = C.c_uint32
        L.svc_oracle_connected_components.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _i32p]
        L.svc_cpu_dct_frame_f32.restype = C.c_int
        L.svc_cpu_dct_frame_f32.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _f32p]
        L.svc_cpu_dct_isa.restype = C.c_int
        L.svc_cpu_quant_frame_f32.restype = None
        L.svc_cpu_quant_frame_f32.argtypes = [_f32p] + [C.c_uint32] * 4 + [_u32p, C.c_uint32, C.c_uint32]
        L.svc_oracle_serialize_frame.restype = C.c_uint64
        L.svc_oracle_serialize_frame.argtypes = [_f32p, C.c_uint64, C.c_uint32, _u32p] + [C.c_uint32] * 7 + [_u8p]
        L.svc_oracle_decode_frame.restype = None
        L.svc_oracle_decode_frame.argtypes = [_f32p] + [C.c_uint32] * 4 + [_u32p] + [C.c_uint32] * 8 + [_f64p]
        L.svc_oracle_sse_frame.restype = C.c_uint64
        L.svc_oracle_sse_frame.argtypes = [_u8p, _f32p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.svc_oracle_quant.restype = None
        L.svc_oracle_quant.argtypes = [_f32p, C.c_uint64, C.c_uint32]
        L.svc_oracle_quant_frame.restype = None
        L.svc_oracle_quant_frame.argtypes = [_f32p] + [C.c_uint32] * 4 + [_u32p, C.c_uint32, C.c_uint32]
        L.svc_oracle_dct_frame_f64.restype = None
        L.svc_oracle_dct_frame_f64.argtypes = [_u8p] + [C.c_uint32] * 4 + [_f64p]
        L.svc_oracle_dct_frame_f32.restype = None
        L.svc_oracle_dct_frame_f32.argtypes = [_u8p] + [C.c_uint32] * 4 + [_f32p]

    # -- motion --
    def ebma(self, tracked, anchor, r, bw, bh):
        h, w = tracked.shape
        n = (w // bw) * (h // bh)
        mv = np.empty((n, 2), np.float32)
        mad = np.empty(n, np.float32)
        self.lib.svc_oracle_ebma(_ptr(tracked, _u8p), _ptr(anchor, _u8p), w, h, r, bw, bh,
                                 _ptr(mv, _f32p), _ptr(mad, _f32p))
        return mv, mad

    def refine(self, tracked, anchor, bw, bh, r, mv, mad):
        h, w = tracked.shape
        mv = np.ascontiguousarray(mv, np.float32).copy()
        mad = np.ascontiguousarray(mad, np.float32).copy()
        self.lib.svc_oracle_refine(_ptr(tracked, _u8p), _ptr(anchor, _u8p), w, h, bw, bh, r,
                                   _ptr(mv, _f32p), _ptr(mad, _f32p))
        return mv, mad

    def hbma(self, tracked_pyr, anchor_pyr, r, bw, bh):
        h, w = tracked_pyr[0].shape
        n = (w // bw) * (h // bh)
        mv = np.empty((n, 2), np.float32)
        mad = np.empty(n, np.float32)
        rc = self.lib.svc_oracle_hbma(_pyr_ptrs(tracked_pyr), _pyr_ptrs(anchor_pyr),
                                      len(tracked_pyr), w, h, r, bw, bh,
                                      _ptr(mv, _f32p), _ptr(mad, _f32p))
        if rc:
            raise ValueError("svc_oracle_hbma: precondition violated")
        return mv, mad

    def hbma16_sse2(self, tracked_pyr, anchor_pyr, r):
        assert len(tracked_pyr) == 4
        h, w = tracked_pyr[0].shape
        n = (w // 16) * (h // 16)
        mv = np.empty((n, 2), np.float32)
        mad = np.empty(n, np.float32)
        rc = self.lib.svc_oracle_hbma16_sse2(_pyr_ptrs(tracked_pyr), _pyr_ptrs(anchor_pyr),
                                             w, h, r, _ptr(mv, _f32p), _ptr(mad, _f32p))
        if rc:
            raise ValueError("svc_oracle_hbma16_sse2: precondition violated")
        return mv, mad

    # -- luma + pyramid (libs/encoder.cpp:468-470; OpenCV steps, parity unpinned) --
    def luma(self, bgr):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        y = np.empty((h, w), np.uint8)
        self.lib.svc_oracle_luma(_ptr(bgr, _u8p), w, h, _ptr(y, _u8p))
        return y

    def pyr_down(self, plane):
        plane = np.ascontiguousarray(plane, np.uint8)
        h, w = plane.shape
        out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.svc_oracle_pyr_down(_ptr(plane, _u8p), w, h, _ptr(out, _u8p))
        return out

    def luma_pyramid(self, bgr, levels):
        """Level planes of one B,G,R frame, level 0 first (w, h divisible by 2^(levels - 1))."""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        sizes = [((h >> l), (w >> l)) for l in range(levels)]
        packed = np.empty(sum(a * b for a, b in sizes), np.uint8)
        self.lib.svc_oracle_luma_pyramid(_ptr(bgr, _u8p), w, h, levels, _ptr(packed, _u8p))
        out, o = [], 0
        for a, b in sizes:
            out.append(packed[o:o + a * b].reshape(a, b))
            o += a * b
        return out

    # -- whole-frame global motion (libs/motion.cpp:45-142) --
    def global_avg(self, mv):
        mv = np.ascontiguousarray(mv, np.float32)
        out = np.zeros(2, np.float32)
        self.lib.svc_oracle_global_avg(_ptr(mv, _f32p), len(mv), _ptr(out, _f32p))
        return out

    def global_ebma(self, tracked, anchor, r, reference_loop=False):
        h, w = tracked.shape
        gm, mad = np.zeros(2, np.float32), np.zeros(1, np.float32)
        self.lib.svc_oracle_global_ebma(_ptr(np.ascontiguousarray(tracked), _u8p), _ptr(np.ascontiguousarray(anchor), _u8p),
                                        w, h, r, int(reference_loop), _ptr(gm, _f32p), _ptr(mad, _f32p))
        return gm, mad[0]

    def global_hbma(self, tracked_pyr, anchor_pyr, r, reference_loop=False):
        h, w = tracked_pyr[0].shape
        gm = np.zeros(2, np.float32)
        self.lib.svc_oracle_global_hbma(_pyr_ptrs(tracked_pyr), _pyr_ptrs(anchor_pyr), len(tracked_pyr), w, h, r,
                                        int(reference_loop), _ptr(gm, _f32p))
        return gm

    # -- RANSAC --
    def ransac_iter_count(self, **p) -> int:
        return int(self.lib.svc_oracle_ransac_iter_count(RansacParams(**p)))

    def ransac(self, mv, samples, gm_in=(0.0, 0.0), n: Optional[int] = None, **p):
        """`mv` may hold n + 1 vectors (see svc_oracle.h); n defaults to len(mv)."""
        mv = np.ascontiguousarray(mv, np.float32)
        n = len(mv) if n is None else n
        samples = np.ascontiguousarray(samples, np.uint32)
        params = RansacParams(**p)
        iters = samples.size // params.subset_sz
        rmse = C.c_float(0)
        gm = np.array(gm_in, np.float32)
        inl = np.empty(max(n, 1), np.uint32)
        cnt = C.c_uint32(0)
        self.lib.svc_oracle_ransac(_ptr(mv, _f32p), n, params, _ptr(samples, _u32p), iters,
                                   C.byref(rmse), _ptr(gm, _f32p), _ptr(inl, _u32p), C.byref(cnt))
        return gm, np.float32(rmse.value), inl[:cnt.value].copy()

    def fg_mask(self, inliers, n):
        inliers = np.ascontiguousarray(inliers, np.uint32)
        mask = np.empty(n, np.uint8)
        self.lib.svc_oracle_fg_mask(_ptr(inliers, _u32p), inliers.size, n, _ptr(mask, _u8p))
        return mask

    def segment(self, inlier_mask, mv, mfw, mfh, mv_bw=16, mv_bh=16, morph_w=3, morph_h=3, cluster_count=10,
                attempts=3, max_iter=10, epsilon=1.0, connectivity=4, seed=0):
        """Region ids per MV block (defaults: apps/encoder.cpp:47-56)."""
        m = np.ascontiguousarray(inlier_mask, np.uint8)
        v = np.ascontiguousarray(mv, np.float32)
        out = np.empty(mfw * mfh, np.uint32)
        rc = self.lib.svc_oracle_segment(_ptr(m, _u8p), _ptr(v, _f32p), mfw, mfh, mv_bw, mv_bh, morph_w, morph_h,
                                         cluster_count, attempts, max_iter, epsilon, connectivity, seed,
                                         _ptr(out, _u32p))
        if rc:
            raise ValueError("svc_oracle_segment: invalid parameter")
        return out

    def serialize_frame(self, planes, block_types, frame_w, frame_h, tbw, tbh, mv_field_w, mv_bw=16, mv_bh=16):
        """planes (C, H, W) f32 -> bytes exactly as libs/encoder.cpp:222-269 emits them for these arguments."""
        pl = np.ascontiguousarray(planes, np.float32)
        bt = np.ascontiguousarray(block_types, np.uint32)
        ch = pl.shape[0]
        cap = ((frame_h + tbh - 1) // tbh) * ((frame_w + tbw - 1) // tbw) * (4 + 4 * ch * tbw * tbh)
        out = np.empty(cap, np.uint8)
        n = self.lib.svc_oracle_serialize_frame(_ptr(pl, _f32p), pl.shape[1] * pl.shape[2], ch, _ptr(bt, _u32p),
                                                frame_w, frame_h, tbw, tbh, mv_field_w, mv_bw, mv_bh, _ptr(out, _u8p))
        return out[:n].copy()

    def decode_frame(self, planes, block, block_types, mv_block=16, fg_step=1, bg_step=640, gaze=(0, 0, 0, 0)):
        pl = np.ascontiguousarray(planes, np.float32)
        _, h, w = pl.shape
        bt = np.ascontiguousarray(block_types, np.uint32)
        out = np.empty((h, w, 3), np.float64)
        self.lib.svc_oracle_decode_frame(_ptr(pl, _f32p), w, h, block, block, _ptr(bt, _u32p), mv_block, mv_block,
                                         fg_step, bg_step, *gaze, _ptr(out, _f64p))
        return out

    def sse_frame(self, src_bgr, rec_bgr, region_w, region_h):
        s = np.ascontiguousarray(src_bgr, np.uint8)
        r = np.ascontiguousarray(rec_bgr, np.float32)
        return int(self.lib.svc_oracle_sse_frame(_ptr(s, _u8p), _ptr(r, _f32p), s.shape[1], region_w, region_h))

    # -- the per-call image operations (svc_imageops.c) --
    def bgr2yuv(self, bgr):
        src = np.ascontiguousarray(bgr, np.uint8)
        out = np.empty_like(src)
        self.lib.svc_oracle_bgr2yuv(_ptr(src, _u8p), src.shape[1], src.shape[0], _ptr(out, _u8p))
        return out

    def morph_rect(self, img, kw, kh, op):
        src = np.ascontiguousarray(img, np.uint8)
        out = np.empty_like(src)
        self.lib.svc_oracle_morph_rect(_ptr(src, _u8p), src.shape[1], src.shape[0], kw, kh, op, _ptr(out, _u8p))
        return out

    def kmeans(self, features, k, attempts=3, max_iter=10, epsilon=1.0, seed=0):
        f = np.ascontiguousarray(features, np.float32)
        n, dims = f.shape
        labels = np.empty(n, np.int32)
        compact = C.c_double(0.0)
        rc = self.lib.svc_oracle_kmeans(_ptr(f, _f32p), n, dims, k, attempts, max_iter, epsilon, seed,
                                        labels.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(compact))
        if rc:
            raise ValueError("svc_oracle_kmeans: invalid parameter")
        return labels, compact.value

    def connected_components(self, img, connectivity=4):
        src = np.ascontiguousarray(img, np.uint8)
        labels = np.empty(src.shape, np.int32)
        count = self.lib.svc_oracle_connected_components(_ptr(src, _u8p), src.shape[1], src.shape[0], connectivity,
                                                         labels.ctypes.data_as(C.POINTER(C.c_int32)))
        return labels, int(count)

    def segment_by_calls(self, inlier_indices, mv, mfw, mfh, mv_bw=16, mv_bh=16, morph_w=3, morph_h=3, cluster_count=10,
                         attempts=3, max_iter=10, epsilon=1.0, connectivity=4, seed=0):
        """libs/encoder.cpp:507-623 composed from the per-call functions exactly as the reference composes the cv::
        calls (mask, close, open, index list, BuildMvFeatures with its m.y overwrite, kmeans, one connectedComponents per
        cluster, offset numbering): must equal segment()."""
        n = mfw * mfh
        fg = np.full(n, 255, np.uint8)
        fg[np.asarray(inlier_indices, np.int64)] = 0                               # :507-513
        fg = self.morph_rect(fg.reshape(mfh, mfw), morph_w, morph_h, 3)            # :524-525 close
        fg = self.morph_rect(fg, morph_w, morph_h, 2).reshape(-1)                  # :526-527 open
        idx = np.nonzero(fg == 255)[0]                                             # :538-546
        types = np.zeros(n, np.uint32)                                             # :549-551
        if len(idx) == 0:
            return types
        k = min(cluster_count, len(idx))                                           # :555
        mvv = np.asarray(mv, np.float32).reshape(n, 2)
        # :300-321 with libs/math.hpp:285-291: Vec4f is {w, x, y, z} and operator[](i) is (&x)[i], so features[i][0..2]
        # write memory slots 1..3 and slot 0 (w) keeps the 0 of vector::resize: (0, m.x, x_px, y_px), m.y overwritten
        feats = np.zeros((len(idx), 4), np.float32)
        feats[:, 1] = mvv[idx, 0]
        feats[:, 2] = (idx % mfw) * mv_bw
        feats[:, 3] = (idx // mfw) * mv_bh
        labels, _ = self.kmeans(feats, k, attempts, max_iter, epsilon, seed)       # :575-576
        offset = 0
        for cid in range(k):                                                       # :597-623
            m = np.zeros(n, np.uint8)
            m[idx[labels == cid]] = 255
            cc, count = self.connected_components(m.reshape(mfh, mfw), connectivity)
            cc = cc.reshape(-1)
            sel = idx[cc[idx] != 0]
            types[sel] = cc[sel].astype(np.uint32) + offset
            offset += count
        return types

    # -- the CPU baseline's transform (svc_cpu_dct.c): f32 separable, AVX2 where available --
    def cpu_dct_frame_f32(self, bgr, block, out=None):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        if out is None:
            out = np.empty((3, h, w), np.float32)
        if self.lib.svc_cpu_dct_frame_f32(_ptr(bgr, _u8p), w, h, block, _ptr(out, _f32p)):
            raise ValueError("svc_cpu_dct_frame_f32: block must be 8 or 16 and divide the frame")
        return out

    def cpu_quant_frame_f32(self, planes, mv_bw, mv_bh, block_types, fg_step, bg_step):
        """In place on a C-contiguous (3, H, W) f32 array; returns it."""
        assert planes.dtype == np.float32 and planes.flags["C_CONTIGUOUS"]
        bt = np.ascontiguousarray(block_types, np.uint32)
        self.lib.svc_cpu_quant_frame_f32(_ptr(planes, _f32p), planes.shape[2], planes.shape[1], mv_bw, mv_bh, _ptr(bt, _u32p),
                                         fg_step, bg_step)
        return planes

    def cpu_dct_isa(self) -> str:
        return "avx2+fma" if self.lib.svc_cpu_dct_isa() == 2 else "sse2 (baseline x86-64)"

    # -- quant / DCT --
    def quant(self, coeffs, step):
        out = np.ascontiguousarray(coeffs, np.float32).copy()
        self.lib.svc_oracle_quant(_ptr(out, _f32p), out.size, step)
        return out

    def quant_frame(self, planes, mv_bw, mv_bh, block_types, fg_step, bg_step):
        out = np.ascontiguousarray(planes, np.float32).copy()
        _, h, w = out.shape
        bt = np.ascontiguousarray(block_types, np.uint32)
        self.lib.svc_oracle_quant_frame(_ptr(out, _f32p), w, h, mv_bw, mv_bh, _ptr(bt, _u32p),
                                        fg_step, bg_step)
        return out

    def dct_frame_f64(self, bgr, bw, bh):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        out = np.empty((3, h, w), np.float64)
        self.lib.svc_oracle_dct_frame_f64(_ptr(bgr, _u8p), w, h, bw, bh, _ptr(out, _f64p))
        return out

    def dct_frame_f32(self, bgr, bw, bh):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        out = np.empty((3, h, w), np.float32)
        self.lib.svc_oracle_dct_frame_f32(_ptr(bgr, _u8p), w, h, bw, bh, _ptr(out, _f32p))
        return out


class Reference:
    """The unmodified reference libs/motion.cpp (oracle/_ref/libsvc_ref.so)."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
        L = self.lib = C.CDLL(path)
        L.svc_ref_seed.restype = C.c_uint
        L.svc_ref_has_sse2.restype = C.c_int
        L.svc_ref_ebma.restype = None
        L.svc_ref_ebma.argtypes = [_u8p, _u8p] + [C.c_uint32] * 5 + [_f32p, _f32p]
        L.svc_ref_hbma.restype = None
        L.svc_ref_hbma.argtypes = [_u8pp, _u8pp] + [C.c_uint32] * 6 + [_f32p, _f32p]
        L.svc_ref_hbma16_sse2.restype = None
        L.svc_ref_hbma16_sse2.argtypes = [_u8pp, _u8pp] + [C.c_uint32] * 3 + [_f32p, _f32p]
        L.svc_ref_ransac.restype = C.c_uint32
        L.svc_ref_ransac.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                     _f32p, _f32p, _u32p]
        L.svc_ref_ransac_draw.restype = None
        L.svc_ref_ransac_draw.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _u32p]

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def ebma(self, tracked, anchor, r, bw, bh):
        h, w = tracked.shape
        n = (w // bw) * (h // bh)
        mv = np.empty((n, 2), np.float32)
        mad = np.empty(n, np.float32)
        self.lib.svc_ref_ebma(_ptr(tracked, _u8p), _ptr(anchor, _u8p), w, h, r, bw, bh,
                              _ptr(mv, _f32p), _ptr(mad, _f32p))
        return mv, mad

    def hbma(self, tracked_pyr, anchor_pyr, r, bw, bh):
        h, w = tracked_pyr[0].shape
        n = (w // bw) * (h // bh)
        mv = np.empty((n, 2), np.float32)
        mad = np.empty(n, np.float32)
        self.lib.svc_ref_hbma(_pyr_ptrs(tracked_pyr), _pyr_ptrs(anchor_pyr), len(tracked_pyr),
                              w, h, r, bw, bh, _ptr(mv, _f32p), _ptr(mad, _f32p))
        return mv, mad

    def hbma16_sse2(self, tracked_pyr, anchor_pyr, r):
        assert len(tracked_pyr) == 4 and self.lib.svc_ref_has_sse2()
        h, w = tracked_pyr[0].shape
        n = (w // 16) * (h // 16)
        mv = np.empty((n, 2), np.float32)
        mad = np.empty(n, np.float32)
        self.lib.svc_ref_hbma16_sse2(_pyr_ptrs(tracked_pyr), _pyr_ptrs(anchor_pyr), w, h, r,
                                     _ptr(mv, _f32p), _ptr(mad, _f32p))
        return mv, mad

    def global_avg(self, mv):
        mv = np.ascontiguousarray(mv, np.float32)
        out = np.zeros(2, np.float32)
        self.lib.svc_ref_global_avg.restype = None
        self.lib.svc_ref_global_avg.argtypes = [_f32p, C.c_uint32, _f32p]
        self.lib.svc_ref_global_avg(_ptr(mv, _f32p), len(mv), _ptr(out, _f32p))
        return out

    def global_ebma(self, tracked, anchor, r):
        h, w = tracked.shape
        gm, mad = np.zeros(2, np.float32), np.zeros(1, np.float32)
        self.lib.svc_ref_global_ebma.restype = None
        self.lib.svc_ref_global_ebma.argtypes = [_u8p, _u8p, C.c_uint32, C.c_uint32, C.c_uint32, _f32p, _f32p]
        self.lib.svc_ref_global_ebma(_ptr(np.ascontiguousarray(tracked), _u8p), _ptr(np.ascontiguousarray(anchor), _u8p), w, h, r,
                                     _ptr(gm, _f32p), _ptr(mad, _f32p))
        return gm, mad[0]

    def global_hbma(self, tracked_pyr, anchor_pyr, r):
        h, w = tracked_pyr[0].shape
        gm = np.zeros(2, np.float32)
        self.lib.svc_ref_global_hbma.restype = None
        self.lib.svc_ref_global_hbma.argtypes = [_u8pp, _u8pp] + [C.c_uint32] * 4 + [_f32p]
        self.lib.svc_ref_global_hbma(_pyr_ptrs(tracked_pyr), _pyr_ptrs(anchor_pyr), len(tracked_pyr), w, h, r, _ptr(gm, _f32p))
        return gm

    def ransac(self, mv_n_plus_1, n, gm_in=(0.0, 0.0), **p):
        """`mv_n_plus_1` holds n + 1 vectors (the reference may read entry n)."""
        mv = np.ascontiguousarray(mv_n_plus_1, np.float32)
        assert len(mv) == n + 1
        rmse = C.c_float(0)
        gm = np.array(gm_in, np.float32)
        inl = np.empty(max(n, 1), np.uint32)
        cnt = self.lib.svc_ref_ransac(_ptr(mv, _f32p), n, p["subset_sz"], p["inlier_thresh"],
                                      p["success_prob"], p["inlier_ratio"], C.byref(rmse),
                                      _ptr(gm, _f32p), _ptr(inl, _u32p))
        return gm, np.float32(rmse.value), inl[:cnt].copy()

    def ransac_draw(self, n, subset_sz, iter_count):
        s = np.empty(iter_count * subset_sz, np.uint32)
        self.lib.svc_ref_ransac_draw(n, subset_sz, iter_count, _ptr(s, _u32p))
        return s
