"""ctypes binding of include/svc_clip.h: the C++ driver (svc::ClipEncoder) of one rank's shard of
an HBM-resident clip -- buffers, streams, the step schedule and the RCCL halo all live in C++
(csrc/host/clip_encoder.cpp); Python only loads frames, calls step() and reads results.

There is no fallback: a missing library or a failing call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, Optional, Tuple

import torch

from . import native
from .configs import CodecConfig
from .native import RansacParams, SegmentParams

_vp, _u32, _u64 = C.c_void_p, C.c_uint32, C.c_uint64

SERIAL, PIPELINED = 0, 1
TUNE_STANDALONE_SHAPES, TUNE_SEGMENT_FORK, TUNE_NARROW_ATTEMPTS, TUNE_INLINE_RMSE, TUNE_TWO_BGR_PASSES, TUNE_ALWAYS_SPECULATE = 1, 2, 4, 8, 16, 32  # svc_clip_config.tuning bits
TUNE_IDLE_RULE_ANY_SIZE = 512  # the idle-pipeline rule whatever the shard's size (tests)
TUNE_FORK_BEHIND_FRONT = 4096  # one rank, A/B: RANSAC + segmentation fork behind the front-of-step transform
TUNE_RANDOM_POLICY = 2048  # tests: the speculation policy answers yes / no by a fixed pseudo-random sequence over the chunk launches
TUNE_MIXED_STEPS = 1024  # a step into an empty pipeline that knows nothing about the clip takes the mixed form (A/B, off by default)
TUNE_SEARCH_AFTER_TRANSFORM = 256  # one rank: the motion search right behind the transform kernel (A/B)
TUNE_WHOLE_SHARD_STEPS = 128  # never the idle-pipeline rule (A/B)
KEEP_FOREGROUND_PRIOR = 64  # same field: the clips loaded are consecutive pieces of one stream (the policy keeps its prior across load_frames)
STAGES = ("luma_pyramid", "halo_exchange", "hbma", "ransac", "segment", "dct_quant", "type_patch")
BUFFERS = {"mv": (0, torch.float32), "min_mad": (1, torch.float32), "global_motion": (2, torch.float32),
           "rmse": (3, torch.float32), "inlier_mask": (4, torch.uint8), "inlier_count": (5, torch.int32),
           "block_types": (6, torch.int32), "coeffs": (7, torch.float32), "records": (8, torch.uint8),
           "pyramids": (9, torch.uint8), "bgr": (10, torch.uint8)}
COMM_ID_BYTES = 128


class ClipConfig(C.Structure):
    _fields_ = [("struct_size", _u32), ("width", _u32), ("height", _u32), ("levels", _u32), ("mv_block", _u32), ("search_range", _u32),
                ("dct_block_w", _u32), ("dct_block_h", _u32), ("fg_step", _u32), ("bg_step", _u32), ("wire", _u32),
                ("segmentation", _u32), ("seed", _u64), ("ransac", RansacParams), ("segment", SegmentParams),
                ("clip_frames", _u32), ("rank", _u32), ("world", _u32), ("schedule", _u32), ("chunk_pairs", _u32),
                ("hbma_flags", _u32), ("lat_depth", _u32), ("tuning", _u32)]


class ClipInfo(C.Structure):
    _fields_ = [(n, _u32) for n in ("padded_w", "padded_h", "mv_field_w", "mv_field_h", "blocks", "ransac_iters")] + \
               [(n, _u64) for n in ("pyramid_stride", "frame_bytes", "record_bytes")] + \
               [(n, _u32) for n in ("first_frame", "frames", "pairs", "first_encoded", "needs_halo", "chunks_per_step", "output_sets", "reserved")]


HALO_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _u64, _vp, _vp)

# name -> (restype, argtypes); mirrors include/svc_clip.h one to one
SIGNATURES = {
    "svc_clip_last_error": (C.c_char_p, []),
    "svc_clip_plan_shard": (C.c_int, [_u32, _u32, _u32] + [C.POINTER(_u32)] * 4),
    "svc_clip_create": (C.c_int, [C.POINTER(ClipConfig), C.POINTER(_vp)]),
    "svc_clip_destroy": (None, [_vp]),
    "svc_clip_get_info": (C.c_int, [_vp, C.POINTER(ClipInfo)]),
    "svc_clip_load_frames": (C.c_int, [_vp, _vp, _u32, _u32, C.c_int]),
    "svc_clip_set_comm": (C.c_int, [_vp, _vp]),
    "svc_clip_set_halo_callback": (C.c_int, [_vp, HALO_FN, _vp]),
    "svc_clip_step": (C.c_int, [_vp, C.c_int]),
    "svc_clip_step_frames": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_u32)]),
    "svc_clip_wait_step": (C.c_int, [_vp, _u32]),
    "svc_clip_flush": (C.c_int, [_vp]),
    "svc_clip_sync": (C.c_int, [_vp]),
    "svc_clip_stage_time": (C.c_int, [_vp, _u32, C.POINTER(C.c_double), C.POINTER(_u32)]),
    "svc_clip_stage_pairs": (C.c_int, [_vp, _u32, C.POINTER(_u64)]),
    "svc_clip_reset_timers": (C.c_int, [_vp]),
    "svc_clip_reset_policy": (C.c_int, [_vp]),
    "svc_clip_policy_info": (C.c_int, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(C.c_double)]),
    "svc_clip_output": (C.c_int, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u64)]),
    "svc_clip_read": (C.c_int, [_vp, _u32, _u64, _vp, _u64, C.c_int]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads libsvc_motion.so (the C++ layer above the C ABI); raises if it has not been built."""
    global _lib
    if _lib is None:
        native.load()  # libsvc_hip.so first: libsvc_motion.so links against it
        if not os.path.exists(native.MOTION_LIB_PATH):
            raise FileNotFoundError(f"{native.MOTION_LIB_PATH} is missing: build it with `python -m scalable_video_codec_amd.build`")
        lib = C.CDLL(native.MOTION_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class ClipError(RuntimeError):
    pass


def _check(rc: int) -> None:
    if rc:
        raise ClipError(load().svc_clip_last_error().decode())


def plan_shard(clip_frames: int, world: int, rank: int) -> Tuple[int, int, int, int]:
    """(first_frame, frames, pairs, first_encoded) of shard `rank` -- svc::PlanShard."""
    a, b, c, d = _u32(), _u32(), _u32(), _u32()
    _check(load().svc_clip_plan_shard(clip_frames, world, rank, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
    return a.value, b.value, c.value, d.value


# ---- RCCL communicator of the C ABI (include/svc_hip.h, csrc/comm.hip) --------------------------

def comm_available() -> bool:
    """librccl binds in this process (no communicator is created: not a collective)."""
    return native.load().svc_hip_comm_available() == 0


def comm_info(comm: int) -> Tuple[int, int, int]:
    """(ranks, my rank, device) as the communicator itself reports them (ncclCommCount / UserRank / CuDevice)."""
    n, r, d = _u32(), _u32(), C.c_int32()
    native._check(native.load().svc_hip_comm_info(_vp(comm), C.byref(n), C.byref(r), C.byref(d)))
    return n.value, r.value, d.value


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    native._check(native.load().svc_hip_comm_unique_id(buf))
    return bytes(buf)


def comm_create(uid: bytes, rank: int, world: int) -> int:
    assert len(uid) == COMM_ID_BYTES
    buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
    comm = _vp()
    native._check(native.load().svc_hip_comm_create(buf, rank, world, C.byref(comm)))
    return comm.value


def comm_destroy(comm: int) -> None:
    native._check(native.load().svc_hip_comm_destroy(_vp(comm)))


def halo_shift(comm: int, send: Optional[torch.Tensor], recv: Optional[torch.Tensor], nbytes: int, rank: int, world: int,
               cyclic: bool = False) -> None:
    """svc_hip_halo_shift on torch's current stream."""
    native._check(native.load().svc_hip_halo_shift(_vp(comm), _vp(send.data_ptr() if send is not None else 0),
                                                   _vp(recv.data_ptr() if recv is not None else 0), nbytes, rank, world,
                                                   1 if cyclic else 0, _vp(torch.cuda.current_stream().cuda_stream)))


class Clip:
    """One rank's shard of a clip, resident in HBM, driven by svc::ClipEncoder."""

    def __init__(self, cfg: CodecConfig, clip_frames: int, rank: int = 0, world: int = 1, schedule: int = PIPELINED,
                 segmentation: bool = True, wire: bool = False, seed: Optional[int] = None,
                 ransac: Optional[dict] = None, segment: Optional[dict] = None, dct_block: Optional[Tuple[int, int]] = None,
                 hbma_flags: int = 0, lat_depth: int = 0, tuning: int = 0, chunk_pairs: int = 0):
        self.cfg = cfg
        r = dict(subset_sz=1, inlier_thresh=7.5, success_prob=0.99, inlier_ratio=0.5)
        r.update(ransac or {})
        s = dict(native.DEFAULT_SEGMENT)
        s.update(segment or {})
        bw, bh = dct_block if dct_block is not None else (cfg.dct_block, cfg.dct_block)
        self.config = ClipConfig(C.sizeof(ClipConfig), cfg.width, cfg.height, cfg.levels, cfg.mv_block, cfg.search_range, bw, bh,
                                 cfg.fg_step, cfg.bg_step, int(wire), int(segmentation),
                                 cfg.seed if seed is None else seed, RansacParams(**r), SegmentParams(**s),
                                 clip_frames, rank, world, schedule, chunk_pairs, hbma_flags, lat_depth, tuning)
        self._h = _vp()
        self._cb = None  # keeps the ctypes callback alive
        _check(load().svc_clip_create(C.byref(self.config), C.byref(self._h)))
        self.info = ClipInfo()
        _check(load().svc_clip_get_info(self._h, C.byref(self.info)))

    def close(self) -> None:
        if self._h:
            load().svc_clip_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs ---------------------------------------------------------------------------------
    def load_frames(self, frames, first_local: int = 0) -> None:
        """frames: (n, padded_h, padded_w, 3) u8 tensor (CUDA or CPU), or a list of such frames."""
        if isinstance(frames, (list, tuple)):
            for i, f in enumerate(frames):
                self.load_frames(f.unsqueeze(0), first_local + i)
            return
        assert frames.dtype == torch.uint8 and frames.is_contiguous()
        assert tuple(frames.shape[1:]) == (self.info.padded_h, self.info.padded_w, 3), frames.shape
        if frames.is_cuda:
            torch.cuda.current_stream().synchronize()
        _check(load().svc_clip_load_frames(self._h, _vp(frames.data_ptr()), first_local, frames.shape[0], int(frames.is_cuda)))

    def set_comm(self, comm: int) -> None:
        _check(load().svc_clip_set_comm(self._h, _vp(comm)))

    def set_halo_transport(self, fn: Optional[Callable[[int, int, int, int], None]]) -> None:
        """fn(send_ptr, recv_ptr, nbytes, stream): enqueue the neighbour shift on `stream` (tests, and the
        torch.distributed transport of the harness)."""
        if fn is None:
            self._cb = None
            _check(load().svc_clip_set_halo_callback(self._h, C.cast(None, HALO_FN), None))
            return

        def tramp(send, recv, nbytes, stream, _user):
            try:
                fn(send or 0, recv or 0, nbytes, stream or 0)
                return 0
            except Exception as e:  # noqa: BLE001 -- reported through the C ABI's status
                import traceback
                traceback.print_exc()
                return 1
        self._cb = HALO_FN(tramp)
        _check(load().svc_clip_set_halo_callback(self._h, self._cb, None))

    # -- running ----------------------------------------------------------------------------------
    def step(self, timed: bool = False) -> None:
        _check(load().svc_clip_step(self._h, int(timed)))

    def step_frames(self, frames: torch.Tensor, timed: bool = False) -> int:
        """One pass over the shard's frames where THEY are (a CUDA tensor laid out as load_frames fills the resident buffer): a stream of
        clips, each encoded once, without a copy and without draining the pipeline.  Returns the step's number; the tensor must stay alive
        and untouched until wait_step(number) or sync()."""
        i = self.info
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous()
        assert tuple(frames.shape) == (i.frames, i.padded_h, i.padded_w, 3), frames.shape
        torch.cuda.current_stream().synchronize()  # whoever produced the frames has finished (the encoder's streams do not order behind torch's)
        step = _u32()
        _check(load().svc_clip_step_frames(self._h, _vp(frames.data_ptr()), int(timed), C.byref(step)))
        return step.value

    def wait_step(self, step: int) -> None:
        _check(load().svc_clip_wait_step(self._h, step))

    def flush(self) -> None:
        _check(load().svc_clip_flush(self._h))

    def sync(self) -> None:
        _check(load().svc_clip_sync(self._h))

    def reset_timers(self) -> None:
        _check(load().svc_clip_reset_timers(self._h))

    def output_sets(self) -> int:
        """Sets the coefficient planes / records exist in NOW (planes: the extra sets appear with the first speculative step)."""
        i = ClipInfo()
        _check(load().svc_clip_get_info(self._h, C.byref(i)))
        return i.output_sets

    def reset_policy(self) -> None:
        """Forget what the speculation policy has measured (what load_frames does unless KEEP_FOREGROUND_PRIOR)."""
        _check(load().svc_clip_reset_policy(self._h))

    def policy_info(self) -> Dict[str, float]:
        """Chunk launches that had the choice to speculate, those that did, the newest foreground share known (-1: none)."""
        a, b, f = _u64(), _u64(), C.c_double()
        _check(load().svc_clip_policy_info(self._h, C.byref(a), C.byref(b), C.byref(f)))
        return {"chunks_decided": a.value, "chunks_speculated": b.value, "foreground_share": f.value}

    def stage_times_ms(self) -> Dict[str, Tuple[float, int]]:
        """stage -> (summed HIP-event ms over the timed steps, launches covered)."""
        out = {}
        for i, name in enumerate(STAGES):
            t, n = C.c_double(), _u32()
            _check(load().svc_clip_stage_time(self._h, i, C.byref(t), C.byref(n)))
            if n.value:
                out[name] = (t.value, n.value)
        return out

    def stage_pairs(self) -> Dict[str, int]:
        """stage -> frame pairs its timed launches covered (a step's launches are its chunks): ms per step = total ms x pairs per step / this."""
        out = {}
        for i, name in enumerate(STAGES):
            n = _u64()
            _check(load().svc_clip_stage_pairs(self._h, i, C.byref(n)))
            out[name] = n.value
        return out

    # -- outputs ----------------------------------------------------------------------------------
    def read(self, name: str, device=None) -> torch.Tensor:
        """A copy of the newest finished step's output buffer as a flat tensor (CPU unless `device`)."""
        idx, dtype = BUFFERS[name]
        ptr, nbytes = _vp(), _u64()
        _check(load().svc_clip_output(self._h, idx, C.byref(ptr), C.byref(nbytes)))
        n = nbytes.value // torch.empty((), dtype=dtype).element_size()
        out = torch.empty(n, dtype=dtype, device=device or "cpu")
        if n:
            _check(load().svc_clip_read(self._h, idx, 0, _vp(out.data_ptr()), nbytes.value, int(out.is_cuda)))
        return out

    def outputs(self, device=None) -> Dict[str, torch.Tensor]:
        i = self.info
        p = i.pairs
        o = {"mv": self.read("mv", device).view(p, i.blocks, 2), "min_mad": self.read("min_mad", device).view(p, i.blocks),
             "global_motion": self.read("global_motion", device).view(p, 2), "rmse": self.read("rmse", device),
             "inlier_mask": self.read("inlier_mask", device).view(p, i.blocks),
             "inlier_count": self.read("inlier_count", device), "block_types": self.read("block_types", device).view(p, i.blocks)}
        return o
