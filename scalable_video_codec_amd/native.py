"""ctypes binding of the C ABI (include/svc_hip.h) for the Python harness.

PyTorch is used for device memory and streams only; every computation below runs
in the hand-written HIP kernels of csrc/ through libsvc_hip.so.  There is no
fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libsvc_hip.so")
MOTION_LIB_PATH = os.path.join(PKG, "libsvc_motion.so")

SVC_OK, SVC_ERR_INVALID_ARG, SVC_ERR_UNSUPPORTED, SVC_ERR_HIP, SVC_ERR_NO_DEVICE = range(5)
HBMA_AUTO, HBMA_FORCE_WAVE_PER_BLOCK, HBMA_FORCE_FUSED, HBMA_FORCE_TILED, HBMA_FORCE_LANE = 0, 1, 2, 4, 8


class SvcError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"svc_hip status {status}: {message}")
        self.status = status


class RansacParams(C.Structure):
    _fields_ = [("subset_sz", C.c_uint32), ("inlier_thresh", C.c_float),
                ("success_prob", C.c_float), ("inlier_ratio", C.c_float)]


class SegmentParams(C.Structure):
    _fields_ = [("morph_rect_w", C.c_uint32), ("morph_rect_h", C.c_uint32), ("cluster_count", C.c_uint32),
                ("attempt_count", C.c_uint32), ("max_iter_count", C.c_uint32), ("epsilon", C.c_float),
                ("connectivity", C.c_uint32)]


class WireHeader(C.Structure):  # libs/codec.hpp:8-17
    _fields_ = [(n, C.c_uint32) for n in ("frame_count", "frame_w", "frame_h", "frame_excess_w", "frame_excess_h",
                                          "transform_block_w", "transform_block_h", "channel_count")]


# apps/encoder.cpp:47-56
DEFAULT_SEGMENT = dict(morph_rect_w=3, morph_rect_h=3, cluster_count=10, attempt_count=3, max_iter_count=10,
                       epsilon=1.0, connectivity=4)

_vp = C.c_void_p
_u32, _u64 = C.c_uint32, C.c_uint64

# name -> (restype, argtypes); mirrors include/svc_hip.h one to one
SIGNATURES = {
    "svc_hip_last_error": (C.c_char_p, []),
    "svc_hip_abi_version": (C.c_int, []),
    "svc_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "svc_hip_tune_host_allocator": (C.c_int, [_u32]),
    "svc_hip_host_tuning_requested": (C.c_int, []),
    "svc_hip_pyramid_bytes": (_u64, [_u32, _u32, _u32]),
    "svc_hip_hbma_pairs": (C.c_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32, _vp]),
    "svc_hip_hbma_kernel_name": (C.c_char_p, [_u32, _u32, _u32, _u32, _u32, _u32, _u32]),
    "svc_hip_ebma_pairs": (C.c_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "svc_hip_ransac_iter_count": (_u32, [RansacParams]),
    "svc_hip_ransac_frames": (C.c_int, [_vp, _u32, _u32, RansacParams, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "svc_hip_ransac_frames_ex": (C.c_int, [_vp, _u32, _u32, RansacParams, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _vp]),
    "svc_hip_ransac_rmse_frames": (C.c_int, [_vp, _u32, _u32, RansacParams, _vp, _vp, _vp, _vp, _vp]),
    "svc_hip_segment_frames_ex": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, SegmentParams, _u64, _vp, _u64, _vp, _u32, _vp]),
    "svc_hip_block_types_frames": (C.c_int, [_vp, _u32, _u32, _vp, _vp]),
    "svc_hip_probe_stream": (C.c_int, [_vp, _vp, _u64, _u32, _u32, _vp]),
    "svc_hip_segment_workspace_bytes": (_u64, [_u32, _u32, _u32, _u32]),
    "svc_hip_segment_frames": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, SegmentParams, _u64, _vp, _u64, _vp, _vp]),
    "svc_hip_wire_header": (C.c_int, [_u32] * 8 + [C.POINTER(WireHeader)]),
    "svc_hip_serialized_frame_bytes": (_u64, [_u32, _u32, _u32, _u32]),
    "svc_hip_serialize_frames": (C.c_int, [_vp, _u64, _u32, _vp] + [_u32] * 8 + [_vp, _u64, _vp]),
    "svc_hip_dct_records_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _u64, _vp]),
    "svc_hip_decode_frames": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _vp] + [_u32] * 8 + [_vp, _vp]),
    "svc_hip_sse_frames": (C.c_int, [_vp, _u64, _vp] + [_u32] * 5 + [_vp, _vp]),
    "svc_hip_dct_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "svc_hip_dct_quant_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "svc_hip_quant": (C.c_int, [_vp, _u64, _u32, _vp]),
    "svc_hip_quant_frames": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _vp]),
    "svc_hip_luma_pyramid_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _vp, _u64, _vp]),
    "svc_hip_pyramid_levels_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _vp]),
    "svc_hip_dct_records_luma_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _u64, _vp]),
    "svc_hip_wire_patch_types_frames": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _u64, C.c_int, _vp]),
    "svc_hip_dct_quant_luma_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u64, _vp]),
    "svc_hip_dct_redo_workspace_bytes": (_u64, [_u32, _u32, _u32, _u32, _u32]),
    "svc_hip_count_foreground": (C.c_int, [_vp, _u64, _vp, _vp]),
    "svc_hip_dct_quant_redo_frames": (C.c_int, [_vp, _u64, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _vp, _vp, _u64, _vp]),
    "svc_hip_hbma_host": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32]),
    "svc_hip_ebma_host": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "svc_hip_ransac_host": (C.c_int, [_vp, _u32, RansacParams, _vp, _u32, _vp, _vp, _vp, _vp]),
    "svc_hip_dct_host": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _vp]),
    "svc_hip_dct_quant_host": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32, _vp]),
    "svc_hip_quant_host": (C.c_int, [_vp, _u64, _u32]),
    "svc_hip_global_ebma_workspace_bytes": (_u64, [_u32, _u32]),
    "svc_hip_global_ebma_pairs": (C.c_int, [_vp, _vp, _u64, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp, _vp]),
    "svc_hip_global_avg_frames": (C.c_int, [_vp, _u32, _u32, _vp, _vp]),
    "svc_hip_global_ebma_host": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp]),
    "svc_hip_global_hbma_host": (C.c_int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    "svc_hip_global_avg_host": (C.c_int, [_vp, _u32, _vp]),
    "svc_hip_comm_available": (C.c_int, []),
    "svc_hip_comm_info": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_int32)]),
    "svc_hip_comm_unique_id": (C.c_int, [_vp]),
    "svc_hip_comm_create": (C.c_int, [_vp, _u32, _u32, C.POINTER(_vp)]),
    "svc_hip_comm_destroy": (C.c_int, [_vp]),
    "svc_hip_halo_shift": (C.c_int, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    # round 4: the per-call image operations behind compat/opencv2/ (host pointers)
    "svc_hip_dct_planes_host": (C.c_int, [_vp, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    "svc_hip_bgr2yuv_host": (C.c_int, [_vp, _u32, _u32, _vp]),
    "svc_hip_build_pyramid_host": (C.c_int, [_vp, _u32, _u32, _u32, C.POINTER(_vp)]),
    "svc_hip_morph_rect_host": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _vp]),
    "svc_hip_kmeans_host": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, C.c_float, _u64, _vp, C.POINTER(C.c_double)]),
    "svc_hip_connected_components_host": (C.c_int, [_vp, _u32, _u32, _u32, _vp, C.POINTER(_u32)]),
    "svc_hip_dct_tiles_host": (C.c_int, [_vp, _u32, _u32, _u32, _u32, _vp, _u32]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads libsvc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `python -m scalable_video_codec_amd.build` "
                "(or __graft_entry__.build()); the MI355X path has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc: int) -> None:
    if rc != SVC_OK:
        raise SvcError(rc, load().svc_hip_last_error().decode())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _bwbh(block) -> Tuple[int, int]:
    """A transform block given as one side (square) or as (block_w, block_h)."""
    return (block, block) if isinstance(block, int) else (int(block[0]), int(block[1]))


def _dev(t: torch.Tensor, dtype) -> int:
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


def ransac_iter_count(subset_sz=1, inlier_thresh=7.5, success_prob=0.99, inlier_ratio=0.5) -> int:
    return int(load().svc_hip_ransac_iter_count(RansacParams(subset_sz, inlier_thresh, success_prob, inlier_ratio)))


def pyramid_bytes(w: int, h: int, levels: int) -> int:
    return int(load().svc_hip_pyramid_bytes(w, h, levels))


def pyramid_stride(w: int, h: int, levels: int) -> int:
    """Packed-pyramid stride used by the harness: rounded up to 256 B."""
    return (pyramid_bytes(w, h, levels) + 255) // 256 * 256


# ---- device-resident, batched ---------------------------------------------------

def hbma_pairs(tracked: torch.Tensor, anchor: torch.Tensor, pair_stride: int, n_pairs: int,
               levels: int, w: int, h: int, search_range: int, block_w: int = 16, block_h: int = 16,
               flags: int = HBMA_AUTO, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """tracked/anchor: u8 CUDA tensors whose data_ptr is pair 0's packed pyramid."""
    blocks = (w // block_w) * (h // block_h)
    if out is None:
        mv = torch.empty((n_pairs, blocks, 2), dtype=torch.float32, device=tracked.device)
        mad = torch.empty((n_pairs, blocks), dtype=torch.float32, device=tracked.device)
    else:
        mv, mad = out
    _check(load().svc_hip_hbma_pairs(_dev(tracked, torch.uint8), _dev(anchor, torch.uint8), pair_stride, n_pairs,
                                     levels, w, h, search_range, block_w, block_h,
                                     _dev(mv, torch.float32), _dev(mad, torch.float32), flags, _stream()))
    return mv, mad


def hbma_kernel_name(levels: int, w: int, h: int, search_range: int, block_w: int = 16, block_h: int = 16,
                     flags: int = HBMA_AUTO) -> str:
    """The kernel hbma_pairs launches for this shape and these flags (aligned pyramids); no GPU work."""
    name = load().svc_hip_hbma_kernel_name(levels, w, h, search_range, block_w, block_h, flags)
    if name is None:  # invalid parameters (the asserts of motion.cpp:417-433) or a forced kernel that does not cover the shape
        msg = load().svc_hip_last_error().decode()
        raise SvcError(SVC_ERR_UNSUPPORTED if "does not cover" in msg else SVC_ERR_INVALID_ARG, msg)
    return name.decode()


def ebma_pairs(tracked: torch.Tensor, anchor: torch.Tensor, pair_stride: int, n_pairs: int, w: int, h: int,
               search_range: int, block_w: int, block_h: int):
    blocks = (w // block_w) * (h // block_h)
    mv = torch.empty((n_pairs, blocks, 2), dtype=torch.float32, device=tracked.device)
    mad = torch.empty((n_pairs, blocks), dtype=torch.float32, device=tracked.device)
    _check(load().svc_hip_ebma_pairs(_dev(tracked, torch.uint8), _dev(anchor, torch.uint8), pair_stride, n_pairs,
                                     w, h, search_range, block_w, block_h,
                                     _dev(mv, torch.float32), _dev(mad, torch.float32), _stream()))
    return mv, mad


LAUNCH_BESIDE = 1
LAUNCH_NO_FORK = 2  # segmentation keeps its heavy attempts on the caller's stream (include/svc_hip.h)


def ransac_frames(mv: torch.Tensor, samples: torch.Tensor, gm_in: Optional[torch.Tensor] = None,
                  subset_sz=1, inlier_thresh=7.5, success_prob=0.99, inlier_ratio=0.5, out=None, flags: int = 0):
    """mv: (frames, blocks, 2) f32; samples: (frames, iters, subset) i32/u32 as int32 storage."""
    frames, blocks, _ = mv.shape
    iters = samples.shape[1] if samples.numel() else 0
    if out is None:
        gm = torch.zeros((frames, 2), dtype=torch.float32, device=mv.device) if gm_in is None else gm_in.clone()
        rmse = torch.empty(frames, dtype=torch.float32, device=mv.device)
        mask = torch.empty((frames, blocks), dtype=torch.uint8, device=mv.device)
        count = torch.empty(frames, dtype=torch.int32, device=mv.device)
    else:
        gm, rmse, mask, count = out
    p = RansacParams(subset_sz, inlier_thresh, success_prob, inlier_ratio)
    _check(load().svc_hip_ransac_frames_ex(_dev(mv, torch.float32), blocks, frames, p, _dev(samples, torch.int32),
                                           iters, _dev(gm, torch.float32), _dev(rmse, torch.float32),
                                           _dev(mask, torch.uint8), _dev(count, torch.int32), flags, _stream()))
    return gm, rmse, mask, count


def ransac_rmse_frames(mv: torch.Tensor, gm: torch.Tensor, mask: torch.Tensor, count: torch.Tensor, rmse: torch.Tensor,
                       subset_sz=1, inlier_thresh=7.5, success_prob=0.99, inlier_ratio=0.5) -> torch.Tensor:
    """Completes a ransac_frames(..., flags=LAUNCH_DEFER_RMSE) call: the in-order RMSE over the inliers, in place."""
    frames, blocks = mv.shape[0], mv.shape[1]
    p = RansacParams(subset_sz, inlier_thresh, success_prob, inlier_ratio)
    _check(load().svc_hip_ransac_rmse_frames(_dev(mv, torch.float32), blocks, frames, p, _dev(gm, torch.float32),
                                             _dev(mask, torch.uint8), _dev(count, torch.int32), _dev(rmse, torch.float32), _stream()))
    return rmse


LAUNCH_BESIDE, LAUNCH_NO_FORK, LAUNCH_WIDE, LAUNCH_NO_WIDE, LAUNCH_DEFER_RMSE = 1, 2, 4, 8, 16


def block_types_frames(mask: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mask: (frames, blocks) u8 inlier mask -> (frames, blocks) i32 block types (0 = background)."""
    frames, blocks = mask.shape
    if out is None:
        out = torch.empty((frames, blocks), dtype=torch.int32, device=mask.device)
    _check(load().svc_hip_block_types_frames(_dev(mask, torch.uint8), blocks, frames, _dev(out, torch.int32), _stream()))
    return out


def probe_stream(src: torch.Tensor, dst: torch.Tensor, reads: int, writes: int) -> None:
    """One launch of the plain streaming kernel (measurement aid): reads x 16 B in, writes x 16 B out per lane; one 4 KiB unit per workgroup."""
    _check(load().svc_hip_probe_stream(_dev(src, torch.uint8), _dev(dst, torch.uint8), min(src.numel(), dst.numel()),
                                       reads, writes, _stream()))


def segment_workspace_bytes(mfw: int, mfh: int, frames: int, attempts: int = 3) -> int:
    return int(load().svc_hip_segment_workspace_bytes(mfw, mfh, frames, attempts))


def segment_frames(mask: torch.Tensor, mv: torch.Tensor, mfw: int, mfh: int, mv_block: int = 16, seed: int = 0,
                   out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, flags: int = 0,
                   **params) -> torch.Tensor:
    """mask (frames, blocks) u8 inlier mask + mv (frames, blocks, 2) f32 -> (frames, blocks) i32 region ids."""
    frames, blocks = mask.shape
    assert blocks == mfw * mfh
    p = SegmentParams(**{**DEFAULT_SEGMENT, **params})
    need = int(load().svc_hip_segment_workspace_bytes(mfw, mfh, frames, p.attempt_count))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=mask.device)
    if out is None:
        out = torch.empty((frames, blocks), dtype=torch.int32, device=mask.device)
    _check(load().svc_hip_segment_frames_ex(_dev(mask, torch.uint8), _dev(mv, torch.float32), mfw, mfh, frames,
                                            mv_block, mv_block, p, seed, _dev(workspace, torch.uint8),
                                            workspace.numel(), _dev(out, torch.int32), flags, _stream()))
    return out


def wire_header(clip_frames: int, w: int, h: int, mv_block: int, levels: int, tb: int) -> bytes:
    hdr = WireHeader()
    _check(load().svc_hip_wire_header(clip_frames, w, h, mv_block, mv_block, levels, tb, tb, C.byref(hdr)))
    return bytes(hdr)


def serialized_frame_bytes(frame_w: int, frame_h: int, tbw: int, tbh: int) -> int:
    return int(load().svc_hip_serialized_frame_bytes(frame_w, frame_h, tbw, tbh))


def serialize_frames(planes: torch.Tensor, block_types: torch.Tensor, frame_w: int, frame_h: int, tbw: int,
                     tbh: int, mfw: int, mfh: int, mv_block: int = 16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """planes (frames, 3, H, W) f32 + types (frames, blocks) i32 -> (frames, bytes) u8, the records of
    libs/encoder.cpp:222-269 for tile loops over frame_w x frame_h (also the row stride, as in the reference)."""
    n, _, ph, pw = planes.shape
    per = serialized_frame_bytes(frame_w, frame_h, tbw, tbh)
    if out is None:
        out = torch.empty((n, per), dtype=torch.uint8, device=planes.device)
    _check(load().svc_hip_serialize_frames(_dev(planes, torch.float32), ph * pw, n, _dev(block_types, torch.int32),
                                           frame_w, frame_h, tbw, tbh, mfw, mfh, mv_block, mv_block,
                                           _dev(out, torch.uint8), out.stride(0), _stream()))
    return out


def dct_records_frames(bgr: torch.Tensor, block: int, block_types: torch.Tensor, mv_block: int = 16,
                       fg_step: int = 0, bg_step: int = 0, emit_h: Optional[int] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bgr (frames, H, W, 3) u8 -> (frames, bytes) u8 serialised records, DCT (+quant) and
    SerializeEncodedFrame in one kernel."""
    n, h, w, _ = bgr.shape
    emit_h = h if emit_h is None else emit_h
    per = serialized_frame_bytes(w, emit_h, block, block)
    if out is None:
        out = torch.empty((n, per), dtype=torch.uint8, device=bgr.device)
    _check(load().svc_hip_dct_records_frames(_dev(bgr, torch.uint8), h * w * 3, n, w, h, block,
                                             _dev(block_types, torch.int32), mv_block, mv_block, fg_step, bg_step,
                                             emit_h, _dev(out, torch.uint8), out.stride(0), _stream()))
    return out


def decode_frames(planes: torch.Tensor, block: int, block_types: torch.Tensor, mv_block: int = 16, fg_step: int = 1,
                  bg_step: int = 640, gaze=(0, 0, 0, 0), out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Coefficient planes (frames, 3, H, W) f32 -> reconstructed (frames, H, W, 3) f32 B,G,R
    (libs/decoder.cpp:128-149 over every tile)."""
    n, _, h, w = planes.shape
    if out is None:
        out = torch.empty((n, h, w, 3), dtype=torch.float32, device=planes.device)
    _check(load().svc_hip_decode_frames(_dev(planes, torch.float32), n, w, h, block, _dev(block_types, torch.int32),
                                        mv_block, mv_block, fg_step, bg_step, *gaze, _dev(out, torch.float32), _stream()))
    return out


def sse_frames(src_bgr: torch.Tensor, rec: torch.Tensor, region_w: int, region_h: int) -> torch.Tensor:
    """Exact per-frame SSE (int64) of u8 source frames vs an f32 reconstruction rounded to u8."""
    n, h, w, _ = src_bgr.shape
    out = torch.empty(n, dtype=torch.int64, device=src_bgr.device)
    _check(load().svc_hip_sse_frames(_dev(src_bgr, torch.uint8), h * w * 3, _dev(rec, torch.float32), n, w, h,
                                     region_w, region_h, _dev(out, torch.int64), _stream()))
    return out


def dct_frames(bgr: torch.Tensor, block, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bgr: (frames, H, W, 3) u8 -> (frames, 3, H, W) f32 coefficient planes.  block: side or (block_w, block_h)."""
    n, h, w, _ = bgr.shape
    bw, bh = _bwbh(block)
    if out is None:
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=bgr.device)
    _check(load().svc_hip_dct_frames(_dev(bgr, torch.uint8), h * w * 3, n, w, h, bw, bh,
                                     _dev(out, torch.float32), _stream()))
    return out


def dct_quant_frames(bgr: torch.Tensor, block, block_types: torch.Tensor, mv_block: int, fg_step: int,
                     bg_step: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, h, w, _ = bgr.shape
    bw, bh = _bwbh(block)
    if out is None:
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=bgr.device)
    _check(load().svc_hip_dct_quant_frames(_dev(bgr, torch.uint8), h * w * 3, n, w, h, bw, bh,
                                           _dev(block_types, torch.int32), mv_block, mv_block, fg_step, bg_step,
                                           _dev(out, torch.float32), _stream()))
    return out


def quant_(coeffs: torch.Tensor, step: int) -> torch.Tensor:
    _check(load().svc_hip_quant(_dev(coeffs, torch.float32), coeffs.numel(), step, _stream()))
    return coeffs


def quant_frames_(planes: torch.Tensor, block_types: torch.Tensor, mv_block: int, fg_step: int, bg_step: int):
    n, _, h, w = planes.shape
    _check(load().svc_hip_quant_frames(_dev(planes, torch.float32), n, w, h, mv_block, mv_block,
                                       _dev(block_types, torch.int32), fg_step, bg_step, _stream()))
    return planes


def luma_pyramid_frames(bgr: torch.Tensor, levels: int, out: Optional[torch.Tensor] = None,
                        stride: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """bgr: (frames, H, W, 3) u8 -> (flat u8 buffer of `frames` packed pyramids, stride)."""
    n, h, w, _ = bgr.shape
    stride = pyramid_stride(w, h, levels) if stride is None else stride
    if out is None:
        out = torch.empty(n * stride, dtype=torch.uint8, device=bgr.device)
    _check(load().svc_hip_luma_pyramid_frames(_dev(bgr, torch.uint8), h * w * 3, n, w, h, levels,
                                              _dev(out, torch.uint8), stride, _stream()))
    return out, stride


def dct_records_luma_frames(bgr: torch.Tensor, block: int, levels: int, emit_h: Optional[int] = None,
                            records: Optional[torch.Tensor] = None, pyr: Optional[torch.Tensor] = None,
                            stride: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """ONE pass over bgr (frames, H, W, 3) u8: the raw-coefficient records (type words 0) AND level 0 of each frame's packed pyramid,
    then the pyramid's other levels from it.  -> (records (frames, bytes) u8, flat pyramid buffer, pyramid stride)."""
    n, h, w, _ = bgr.shape
    emit_h = h if emit_h is None else emit_h
    per = serialized_frame_bytes(w, emit_h, block, block)
    if records is None:
        records = torch.empty((n, per), dtype=torch.uint8, device=bgr.device)
    stride = pyramid_stride(w, h, levels) if stride is None else stride
    if pyr is None:
        pyr = torch.empty(n * stride, dtype=torch.uint8, device=bgr.device)
    _check(load().svc_hip_dct_records_luma_frames(_dev(bgr, torch.uint8), h * w * 3, n, w, h, block, emit_h, _dev(records, torch.uint8),
                                                  records.stride(0), _dev(pyr, torch.uint8), stride, _stream()))
    _check(load().svc_hip_pyramid_levels_frames(_dev(pyr, torch.uint8), stride, n, w, h, levels, _stream()))
    return records, pyr, stride


def pyramid_levels_frames(pyr: torch.Tensor, stride: int, n: int, w: int, h: int, levels: int) -> torch.Tensor:
    """Levels 1 .. levels - 1 of `n` packed pyramids whose level-0 planes are in place (cv::buildPyramid, libs/encoder.cpp:470)."""
    _check(load().svc_hip_pyramid_levels_frames(_dev(pyr, torch.uint8), stride, n, w, h, levels, _stream()))
    return pyr


def dct_quant_luma_frames(bgr: torch.Tensor, block: int, levels: int, bg_step: int = 640, planes: Optional[torch.Tensor] = None,
                          pyr: Optional[torch.Tensor] = None, stride: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """ONE pass over bgr: coefficient planes with EVERY tile quantised as background AND level 0 of each frame's pyramid (then its other
    levels).  -> (planes (frames, 3, H, W) f32, flat pyramid buffer, stride); dct_quant_redo_frames finishes the foreground tiles."""
    n, h, w, _ = bgr.shape
    if planes is None:
        planes = torch.empty((n, 3, h, w), dtype=torch.float32, device=bgr.device)
    stride = pyramid_stride(w, h, levels) if stride is None else stride
    if pyr is None:
        pyr = torch.empty(n * stride, dtype=torch.uint8, device=bgr.device)
    _check(load().svc_hip_dct_quant_luma_frames(_dev(bgr, torch.uint8), h * w * 3, n, w, h, block, bg_step, _dev(planes, torch.float32),
                                                _dev(pyr, torch.uint8), stride, _stream()))
    _check(load().svc_hip_pyramid_levels_frames(_dev(pyr, torch.uint8), stride, n, w, h, levels, _stream()))
    return planes, pyr, stride


def dct_quant_redo_frames(bgr: torch.Tensor, planes: torch.Tensor, block: int, block_types: torch.Tensor, mv_block: int = 16,
                          fg_step: int = 1) -> torch.Tensor:
    """The tiles of every foreground MV block (region id != 0) transformed again and quantised with fg_step, in place."""
    n, h, w, _ = bgr.shape
    nbytes = load().svc_hip_dct_redo_workspace_bytes(n, w, h, mv_block, mv_block)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=bgr.device)
    _check(load().svc_hip_dct_quant_redo_frames(_dev(bgr, torch.uint8), h * w * 3, n, w, h, block, _dev(block_types, torch.int32), mv_block,
                                                mv_block, fg_step, _dev(planes, torch.float32), _dev(ws, torch.uint8), nbytes, _stream()))
    return planes


def count_foreground(block_types: torch.Tensor) -> int:
    """How many of the region ids are not 0."""
    out = torch.zeros(1, dtype=torch.int32, device=block_types.device)
    _check(load().svc_hip_count_foreground(_dev(block_types, torch.int32) if block_types.numel() else None, block_types.numel(),
                                           _dev(out, torch.int32), _stream()))
    return int(out.item())


def wire_patch_types_frames(records: torch.Tensor, block_types: torch.Tensor, w: int, h: int, block: int, mv_block: int = 16,
                            emit_h: Optional[int] = None, all_tiles: bool = False) -> torch.Tensor:
    """Stores the region id of every foreground MV block into the type words of its tiles' records (in place)."""
    n = records.shape[0]
    emit_h = h if emit_h is None else emit_h
    _check(load().svc_hip_wire_patch_types_frames(_dev(block_types, torch.int32), n, w, h, emit_h, block, mv_block, mv_block,
                                                  _dev(records, torch.uint8), records.stride(0), 1 if all_tiles else 0, _stream()))
    return records


# ---- host-pointer forms (numpy in, numpy out): what include/svc/motion.hpp calls ----

def _np_ptr(a):
    return a.ctypes.data_as(_vp)


def hbma_host(tracked_pyr, anchor_pyr, search_range: int, block_w: int = 16, block_h: int = 16,
              flags: int = HBMA_AUTO):
    import numpy as np
    levels = len(tracked_pyr)
    h, w = tracked_pyr[0].shape
    tp = (_vp * levels)(*[_np_ptr(np.ascontiguousarray(p)) for p in tracked_pyr])
    ap = (_vp * levels)(*[_np_ptr(np.ascontiguousarray(p)) for p in anchor_pyr])
    blocks = (w // block_w) * (h // block_h) if block_w and block_h else 0
    mv = np.empty((max(blocks, 1), 2), np.float32)
    mad = np.empty(max(blocks, 1), np.float32)
    _check(load().svc_hip_hbma_host(C.cast(tp, _vp), C.cast(ap, _vp), levels, w, h, search_range, block_w, block_h,
                                    _np_ptr(mv), _np_ptr(mad), flags))
    return mv[:blocks], mad[:blocks]


def ebma_host(tracked, anchor, search_range: int, block_w: int, block_h: int):
    import numpy as np
    h, w = tracked.shape
    blocks = (w // block_w) * (h // block_h)
    mv = np.empty((blocks, 2), np.float32)
    mad = np.empty(blocks, np.float32)
    _check(load().svc_hip_ebma_host(_np_ptr(np.ascontiguousarray(tracked)), _np_ptr(np.ascontiguousarray(anchor)),
                                    w, h, search_range, block_w, block_h, _np_ptr(mv), _np_ptr(mad)))
    return mv, mad


def ransac_host(mv, samples, gm_in=(0.0, 0.0), subset_sz=1, inlier_thresh=7.5, success_prob=0.99,
                inlier_ratio=0.5):
    import numpy as np
    mv = np.ascontiguousarray(mv, np.float32)
    samples = np.ascontiguousarray(samples, np.uint32)
    n = len(mv)
    gm = np.array(gm_in, np.float32)
    rmse = C.c_float(0)
    inl = np.empty(max(n, 1), np.uint32)
    cnt = C.c_uint32(0)
    p = RansacParams(subset_sz, inlier_thresh, success_prob, inlier_ratio)
    iters = samples.size // subset_sz if subset_sz else 0
    _check(load().svc_hip_ransac_host(_np_ptr(mv), n, p, _np_ptr(samples), iters, _np_ptr(gm),
                                      C.cast(C.byref(rmse), _vp), _np_ptr(inl), C.cast(C.byref(cnt), _vp)))
    return gm, np.float32(rmse.value), inl[:cnt.value].copy()


def dct_host(bgr, block):
    import numpy as np
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    bw, bh = _bwbh(block)
    out = np.empty((3, h, w), np.float32)
    _check(load().svc_hip_dct_host(_np_ptr(bgr), w, h, bw, bh, _np_ptr(out)))
    return out


def dct_quant_host(bgr, block, block_types, mv_block, fg_step: int, bg_step: int):
    import numpy as np
    bgr = np.ascontiguousarray(bgr, np.uint8)
    bt = np.ascontiguousarray(block_types, np.uint32)
    h, w, _ = bgr.shape
    bw, bh = _bwbh(block)
    mvw, mvh = _bwbh(mv_block)
    out = np.empty((3, h, w), np.float32)
    _check(load().svc_hip_dct_quant_host(_np_ptr(bgr), w, h, bw, bh, _np_ptr(bt), mvw, mvh,
                                         fg_step, bg_step, _np_ptr(out)))
    return out


def quant_host(coeffs, step: int):
    import numpy as np
    out = np.ascontiguousarray(coeffs, np.float32).copy()
    _check(load().svc_hip_quant_host(_np_ptr(out), out.size, step))
    return out


def global_ebma_host(tracked, anchor, search_range: int):
    """EstimateGlobalMotionExhaustiveSearch (libs/motion.hpp:45-49) -> ((dx, dy), min_mad)."""
    import numpy as np
    h, w = tracked.shape
    gm = np.zeros(2, np.float32)
    mad = C.c_float(0)
    _check(load().svc_hip_global_ebma_host(_np_ptr(np.ascontiguousarray(tracked)), _np_ptr(np.ascontiguousarray(anchor)), w, h,
                                           search_range, _np_ptr(gm), C.cast(C.byref(mad), _vp)))
    return gm, np.float32(mad.value)


def global_hbma_host(tracked_pyr, anchor_pyr, search_range: int):
    """EstimateGlobalMotionHierarchical (libs/motion.hpp:55-59) -> (dx, dy)."""
    import numpy as np
    levels = len(tracked_pyr)
    h, w = tracked_pyr[0].shape
    tp = (_vp * levels)(*[_np_ptr(np.ascontiguousarray(p)) for p in tracked_pyr])
    ap = (_vp * levels)(*[_np_ptr(np.ascontiguousarray(p)) for p in anchor_pyr])
    gm = np.zeros(2, np.float32)
    _check(load().svc_hip_global_hbma_host(C.cast(tp, _vp), C.cast(ap, _vp), levels, w, h, search_range, _np_ptr(gm)))
    return gm


def global_avg_host(mv):
    """EstimateGlobalMotionAvg (libs/motion.hpp:38)."""
    import numpy as np
    mv = np.ascontiguousarray(mv, np.float32)
    out = np.zeros(2, np.float32)
    _check(load().svc_hip_global_avg_host(_np_ptr(mv), len(mv), _np_ptr(out)))
    return out


def global_ebma_pairs(tracked: torch.Tensor, anchor: torch.Tensor, pair_stride: int, n_pairs: int, w: int, h: int,
                      search_range: int):
    ws = torch.empty(int(load().svc_hip_global_ebma_workspace_bytes(search_range, n_pairs)), dtype=torch.uint8, device=tracked.device)
    gm = torch.empty((n_pairs, 2), dtype=torch.float32, device=tracked.device)
    mad = torch.empty(n_pairs, dtype=torch.float32, device=tracked.device)
    _check(load().svc_hip_global_ebma_pairs(_dev(tracked, torch.uint8), _dev(anchor, torch.uint8), pair_stride, n_pairs, w, h,
                                            search_range, _dev(ws, torch.uint8), ws.numel(), _dev(gm, torch.float32),
                                            _dev(mad, torch.float32), _stream()))
    return gm, mad


def global_avg_frames(mv: torch.Tensor) -> torch.Tensor:
    frames, blocks, _ = mv.shape
    out = torch.empty((frames, 2), dtype=torch.float32, device=mv.device)
    _check(load().svc_hip_global_avg_frames(_dev(mv, torch.float32), blocks, frames, _dev(out, torch.float32), _stream()))
    return out


# ---- per-call image operations (host numpy arrays in, numpy arrays out): what compat/opencv2/ forwards to ----------
def _np(a, dtype):
    import numpy as np
    return np.ascontiguousarray(a, dtype)


def bgr2yuv_host(bgr):
    import numpy as np
    src = _np(bgr, np.uint8)
    h, w, _ = src.shape
    out = np.empty_like(src)
    _check(load().svc_hip_bgr2yuv_host(src.ctypes.data, w, h, out.ctypes.data))
    return out


def build_pyramid_host(level0, levels: int):
    import numpy as np
    src = _np(level0, np.uint8)
    h, w = src.shape
    planes = [src] + [np.empty((h >> l, w >> l), np.uint8) for l in range(1, levels)]
    ptrs = (_vp * levels)(*[p.ctypes.data for p in planes])
    _check(load().svc_hip_build_pyramid_host(src.ctypes.data, w, h, levels, ptrs))
    return planes


MORPH_ERODE, MORPH_DILATE, MORPH_OPEN, MORPH_CLOSE = 0, 1, 2, 3


def morph_rect_host(img, kw: int, kh: int, op: int):
    import numpy as np
    src = _np(img, np.uint8)
    h, w = src.shape
    out = np.empty_like(src)
    _check(load().svc_hip_morph_rect_host(src.ctypes.data, w, h, kw, kh, op, out.ctypes.data))
    return out


def kmeans_host(features, k: int, attempts: int = 3, max_iter: int = 10, epsilon: float = 1.0, seed: int = 0):
    import numpy as np
    f = _np(features, np.float32)
    n, dims = f.shape
    labels = np.empty(n, np.int32)
    compact = C.c_double(0.0)
    _check(load().svc_hip_kmeans_host(f.ctypes.data, n, dims, k, attempts, max_iter, epsilon, seed, labels.ctypes.data,
                                      C.byref(compact)))
    return labels, compact.value


def connected_components_host(img, connectivity: int = 4):
    import numpy as np
    src = _np(img, np.uint8)
    h, w = src.shape
    labels = np.empty((h, w), np.int32)
    count = _u32(0)
    _check(load().svc_hip_connected_components_host(src.ctypes.data, w, h, connectivity, labels.ctypes.data, C.byref(count)))
    return labels, int(count.value)


def dct_tiles_host(image, bw: int, bh: int, tiles_xy=None):
    """In-place cv::dct over the listed tiles (or the whole regular grid) of an f32 image; returns the image."""
    import numpy as np
    img = np.array(image, np.float32, order="C")
    h, w = img.shape
    if tiles_xy is None:
        _check(load().svc_hip_dct_tiles_host(img.ctypes.data, w, h, bw, bh, None, 0))
    else:
        xy = _np(tiles_xy, np.uint32)
        _check(load().svc_hip_dct_tiles_host(img.ctypes.data, w, h, bw, bh, xy.ctypes.data, len(xy)))
    return img


def dct_planes_host(bgr, bw: int, bh: int):
    import numpy as np
    src = _np(bgr, np.uint8)
    h, w, _ = src.shape
    planes = [np.empty((h, w), np.float32) for _ in range(3)]
    ptrs = (_vp * 3)(*[p.ctypes.data for p in planes])
    _check(load().svc_hip_dct_planes_host(src.ctypes.data, w, h, bw, bh, ptrs))
    return np.stack(planes)
