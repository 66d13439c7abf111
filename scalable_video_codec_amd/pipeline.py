"""Stage-by-stage harness over the C ABI: the part of the reference's per-frame loop
(Encoder::operator(), libs/encoder.cpp:453-664) that touches motion search, global
motion, block types and the transform, batched over a clip that lives in HBM, one
C-ABI call per stage on torch's current stream.  The product driver -- schedules,
streams, RCCL halo -- is C++ (svc::ClipEncoder, include/svc/clip_encoder.hpp, bound in
clip.py); this module is what the tests compare it with, plus the torch.distributed
halo transport that the CPU (gloo) tests and the bench's SVC_HALO=torch switch use.

Frame order follows the reference: the tracked frame of encoded frame t is the
previous SOURCE frame (libs/encoder.cpp:661-663 swaps source pyramids; there is no
decoded-frame feedback), and frame 0 of a clip is tracked-only (:361-367).

Multi-GPU: a long clip is cut into consecutive chunks, one per rank.
The only cross-rank dependency is the pyramid of the frame just before a chunk, so
each step rank r sends the pyramid of its last frame to rank r+1 (one RCCL
send/recv of ~2.7 MB at 1080p over one xGMI link) and rank r+1 parks it in a halo
slot in front of its own pyramids.  No other data crosses ranks; outputs stay
sharded.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import native
from .configs import CodecConfig

_PMC_JSON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")


def load_pmc_traffic() -> Dict[str, float]:
    """HBM bytes per launch measured offline with rocprofv3 --pmc (profiles/), if recorded."""
    try:
        with open(_PMC_JSON) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


# The kernel sources each counter-traffic figure depends on (svc_common.hpp, shared with the host staging code, is left out): a figure is only quoted for a build whose sources hash the same as the
# build it was collected on (tools/summarize_pmc.py records the hashes next to every summary; bench.py compares).
_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
KERNEL_SOURCE_GROUPS = {
    "hbma": ("hbma_fused.hip", "hbma_fused8.hip", "hbma_fused32.hip", "hbma_fused_kernel.hpp", "hbma_search.hpp", "hbma_tiled.hip",
             "hbma_wave.hip"),
    "dct": ("dct.hip", "dct_tables.inc", "luma16.hpp"),
    "luma_pyr1": ("luma_pyramid.hip", "luma16.hpp"),
}


def kernel_source_hashes() -> Dict[str, Optional[str]]:
    """sha256[:16] over the sources of each kernel group as they are in THIS checkout (None where a file is missing)."""
    import hashlib
    out = {}
    for group, files in KERNEL_SOURCE_GROUPS.items():
        h = hashlib.sha256()
        try:
            for fn in files:
                with open(os.path.join(_CSRC, fn), "rb") as f:
                    h.update(fn.encode() + b"\0" + f.read() + b"\0")
            out[group] = h.hexdigest()[:16]
        except OSError:
            out[group] = None
    return out


def pmc_traffic_for(config_name: str, group: str, key: str, launched_kernel: Optional[str] = None):
    """(bytes per recorded launch, recorded pairs, source) of profiles/pmc_traffic.json for `key` -- or (None, None, reason) when the
    figure cannot be tied to this build: no record, no source hash recorded with it, sources of `group` changed since it was
    collected, or (motion search) the library now dispatches another kernel than the one that was measured."""
    rec = load_pmc_traffic().get(config_name)
    if not rec or rec.get(key) is None or not rec.get("pairs"):
        return None, None, f"no counter traffic recorded for {config_name} / {key} in profiles/pmc_traffic.json"
    on = rec.get("collected_on") or {}
    want = (on.get("source_sha16") or {}).get(group)
    if not want:
        return None, None, f"{rec.get('source')}: no source hash was recorded with it (collected before round 5): not tied to this build"
    have = kernel_source_hashes().get(group)
    if have != want:
        return None, None, (f"{rec.get('source')}: collected on {group} sources {want}, this build has {have}: the kernels changed since; "
                            "re-collect with tools/pmc_passes.sh")
    measured = (on.get("kernels") or {}).get(group)
    if launched_kernel is not None and measured and not any(launched_kernel in m for m in measured):
        return None, None, f"{rec.get('source')}: collected on {measured}, this run launches {launched_kernel}"
    return rec[key], rec["pairs"], rec.get("source")


def plan_shards(total_frames: int, world: int) -> List[Tuple[int, int, bool]]:
    """Cuts a clip of `total_frames` into `world` consecutive chunks, the first `total_frames % world` ranks one frame
    longer -- the plan of svc::PlanShard (include/svc/clip_encoder.hpp), restated here in plain Python so that the CPU
    (gloo) tests need no built library; tests/test_host_logic.py holds the two to each other.  Returns, per rank,
    (first_frame, n_frames, needs_halo): every rank but the first needs the pyramid of frame first_frame - 1 from its
    predecessor.  Encoded frames: total_frames - 1."""
    base, extra = divmod(total_frames, world)
    out = []
    for r in range(world):
        n = base + (1 if r < extra else 0)
        first = r * base + min(r, extra)
        out.append((first, n, first > 0 and n > 0))
    return out


def ransac_samples(n_frames: int, iters: int, subset: int, blocks: int, seed: int, device) -> torch.Tensor:
    """Deterministic, distinct-within-an-iteration sample indices (the explicit draws of
    include/svc_hip.h): a counter hash, then offsets that keep a subset distinct."""
    from .synth import hash32
    idx = torch.arange(n_frames * iters, dtype=torch.int64, device=device)
    first = hash32(idx * 0x9E3779B1 + seed) % blocks
    step = 1 + hash32(idx * 0x85EBCA6B + seed + 1) % max(1, (blocks - 1) // max(1, subset))
    k = torch.arange(subset, dtype=torch.int64, device=device)
    s = (first.unsqueeze(1) + step.unsqueeze(1) * k) % blocks
    return s.reshape(n_frames, iters, subset).to(torch.int32).contiguous()


def halo_exchange(pyr: torch.Tensor, stride: int, n_frames: int, rank: int, world: int) -> None:
    """The one cross-rank step.  `pyr` holds n_frames + 1 packed-pyramid
    slots: slot 0 is the halo, slots 1..n_frames this rank's frames.  Rank r sends its
    last pyramid to rank r + 1, which receives it into slot 0; rank 0 has no predecessor
    (frame 0 of the clip is tracked-only, libs/encoder.cpp:361-367).  Works on any
    backend: RCCL ("nccl") on device buffers in the product, gloo on CPU in the tests."""
    if world <= 1:
        return
    send_buf = pyr[n_frames * stride:(n_frames + 1) * stride]
    recv_buf = pyr[:stride]
    staged = pyr.is_cuda and dist.get_backend() == "gloo"  # rehearsal mode only: gloo moves host memory
    if staged:
        send_buf = send_buf.cpu()
        recv_host = torch.empty(stride, dtype=torch.uint8)
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, send_buf, rank + 1))
    if rank > 0:
        ops.append(dist.P2POp(dist.irecv, recv_host if staged else recv_buf, rank - 1))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if staged and rank > 0:
        recv_buf.copy_(recv_host)


class ClipEncoder:
    """Owns the device buffers of one rank's chunk and runs one hot-path pass per step()."""

    def __init__(self, cfg: CodecConfig, n_frames: int, device, rank: int = 0, world: int = 1,
                 ransac: Optional[dict] = None, segment: Optional[dict] = None, segmentation: bool = True,
                 wire: bool = False):
        self.cfg, self.n, self.dev, self.rank, self.world = cfg, n_frames, device, rank, world
        self.pw, self.ph = cfg.padded
        self.levels = cfg.levels
        self.blocks = cfg.blocks
        self.has_halo = world > 1 and rank > 0
        self.pairs_per_step = n_frames - 1 + (1 if self.has_halo else 0)
        self.encoded_per_step = self.pairs_per_step
        self.first_encoded = 0 if self.has_halo else 1
        self.stride = native.pyramid_stride(self.pw, self.ph, self.levels)
        self.ransac = dict(subset_sz=1, inlier_thresh=7.5, success_prob=0.99, inlier_ratio=0.5)
        if ransac:
            self.ransac.update(ransac)
        p = self.pairs_per_step
        # slot 0 = halo (previous rank's last frame), slots 1..n = own frames
        self.pyr = torch.zeros((n_frames + 1) * self.stride, dtype=torch.uint8, device=device)
        self.bgr = torch.empty((n_frames, self.ph, self.pw, 3), dtype=torch.uint8, device=device)
        self.mv = torch.empty((p, self.blocks, 2), dtype=torch.float32, device=device)
        self.mad = torch.empty((p, self.blocks), dtype=torch.float32, device=device)
        self.gm = torch.zeros((p, 2), dtype=torch.float32, device=device)
        self.rmse = torch.empty(p, dtype=torch.float32, device=device)
        self.mask = torch.empty((p, self.blocks), dtype=torch.uint8, device=device)
        self.count = torch.empty(p, dtype=torch.int32, device=device)
        self.types = torch.empty((p, self.blocks), dtype=torch.int32, device=device)
        # wire=True: the transform emits the serialised records of libs/encoder.cpp:222-269 directly
        # (padded tile counts: the layout the reference's decoder parses) instead of planes
        self.wire = wire and bool(cfg.dct_block)
        self.coeffs = (torch.empty((p, 3, self.ph, self.pw), dtype=torch.float32, device=device)
                       if cfg.dct_block and not self.wire else None)
        self.records = (torch.empty((p, native.serialized_frame_bytes(self.pw, self.ph, cfg.dct_block, cfg.dct_block)),
                                    dtype=torch.uint8, device=device) if self.wire else None)
        # region ids: the full segmentation glue (libs/encoder.cpp:507-623) or, with
        # segmentation=False, only its in-repo part (foreground = one region)
        self.segmentation = segmentation
        self.segment = dict(native.DEFAULT_SEGMENT)
        if segment:
            self.segment.update(segment)
        self.mfw, self.mfh = cfg.mv_field
        self.seg_ws = (torch.empty(native.segment_workspace_bytes(self.mfw, self.mfh, p, self.segment["attempt_count"]), dtype=torch.uint8, device=device)
                       if segmentation else None)
        self.seg_seed = cfg.seed * 1000003 + rank * 100003
        self.iters = native.ransac_iter_count(**self.ransac)
        self.samples = ransac_samples(p, self.iters, self.ransac["subset_sz"], self.blocks,
                                      cfg.seed + 7919 * rank, device)
        self._ev: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event]]] = {}
        self._steps_timed = 0
        self.hbma_kernel_name = native.hbma_kernel_name(self.levels, self.pw, self.ph, cfg.search_range, cfg.mv_block, cfg.mv_block)

    def load_frames(self, frames: List[torch.Tensor]) -> None:
        assert len(frames) == self.n
        for i, f in enumerate(frames):
            self.bgr[i].copy_(f)

    # -- timing: HIP events on the stream the kernels are launched on ----------------
    def reset_kernel_timers(self) -> None:
        self._ev = {}
        self._steps_timed = 0

    def _timed(self, name: str, timed: bool):
        enc = self

        class _Ctx:
            def __enter__(self_inner):
                if timed:
                    self_inner.a = torch.cuda.Event(enable_timing=True)
                    self_inner.b = torch.cuda.Event(enable_timing=True)
                    self_inner.a.record(torch.cuda.current_stream())

            def __exit__(self_inner, *exc):
                if timed:
                    self_inner.b.record(torch.cuda.current_stream())
                    enc._ev.setdefault(name, []).append((self_inner.a, self_inner.b))
        return _Ctx()

    def kernel_times_ms(self) -> Dict[str, float]:
        """Per-step duration of each stage (summed over its launches when the step is chunked),
        from HIP events recorded on the stream each stage is launched on."""
        torch.cuda.synchronize()
        steps = max(1, self._steps_timed)
        return {k: sum(a.elapsed_time(b) for a, b in v) / steps for k, v in self._ev.items()}

    def launches_per_step(self) -> Dict[str, float]:
        steps = max(1, self._steps_timed)
        return {k: len(v) / steps for k, v in self._ev.items()}

    # -- one pass of the hot path -----------------------------------------------------
    def exchange_halo(self) -> None:
        """Ring-less neighbour shift: my last pyramid -> rank+1's halo slot (RCCL over xGMI)."""
        halo_exchange(self.pyr, self.stride, self.n, self.rank, self.world)

    def step(self, timed: bool = False) -> None:
        """One pass over the clip on torch's current stream, one C-ABI call per stage."""
        c = self.cfg
        t0 = 0 if self.has_halo else 1        # slot of the first tracked pyramid
        p = self.pairs_per_step
        with self._timed("luma_pyramid", timed):
            native.luma_pyramid_frames(self.bgr, self.levels, out=self.pyr[self.stride:], stride=self.stride)
        with self._timed("halo_exchange", timed and self.world > 1):
            self.exchange_halo()
        with self._timed("hbma", timed):
            native.hbma_pairs(self.pyr[t0 * self.stride:], self.pyr[(t0 + 1) * self.stride:], self.stride, p,
                              self.levels, self.pw, self.ph, c.search_range, c.mv_block, c.mv_block,
                              out=(self.mv, self.mad))
        with self._timed("ransac", timed):
            self.gm.zero_()
            native.ransac_frames(self.mv, self.samples, out=(self.gm, self.rmse, self.mask, self.count), **self.ransac)
        if self.segmentation:
            with self._timed("segment", timed):
                native.segment_frames(self.mask, self.mv, self.mfw, self.mfh, c.mv_block, seed=self.seg_seed,
                                      out=self.types, workspace=self.seg_ws, **self.segment)
        else:
            with self._timed("block_types", timed):
                native.block_types_frames(self.mask, out=self.types)
        if c.dct_block:
            f0 = self.first_encoded       # encoded frame of pair p is own frame first_encoded + p
            with self._timed("dct_quant", timed):
                if self.wire:
                    # records carry RAW coefficients, as the reference's encoder serialises them
                    # (libs/encoder.cpp:638-650; the decoder picks the step per tile, libs/decoder.cpp:130-135)
                    native.dct_records_frames(self.bgr[f0:f0 + p], c.dct_block, self.types, c.mv_block,
                                              0, 0, out=self.records)
                else:
                    native.dct_quant_frames(self.bgr[f0:f0 + p], c.dct_block, self.types, c.mv_block,
                                            c.fg_step, c.bg_step, out=self.coeffs)
        self._steps_timed += 1 if timed else 0
