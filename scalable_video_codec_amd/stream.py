"""A clip that lives in HOST memory, through the same hot path: frames cross PCIe in batches on a
copy stream while the previous batch is in the kernels and the one before is on its way back
(frame batching + H2D / compute / D2H overlap).

The reference encodes synchronously, one frame per call, from pageable memory
(libs/encoder.cpp:472-498); the C++ drop-in wrappers of include/svc/motion.hpp keep that contract
and pay a PCIe round trip per call.  This is the batched form a host application would use
instead: it writes unpadded BGR frames into pinned input buffers handed out by the encoder and
reads motion vectors, region ids and coefficient planes (or wire records) from pinned output
buffers -- no pageable staging copy on either side.  Results are identical to ClipEncoder's on
the resident clip (same kernels, same RANSAC draws, same segmentation seeds).

The bench's `value` never comes from here (inputs resident in HBM is the contract); DESIGN.md
quotes this path's PCIe-inclusive rate next to it."""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional

import numpy as np
import torch

from .configs import CodecConfig
from . import native
from .pipeline import ClipEncoder, ransac_samples


class _Slot:
    def __init__(self, cfg: CodecConfig, batch: int, device, wire: bool, segmentation: bool):
        self.enc = ClipEncoder(cfg, batch + 1, device, segmentation=segmentation, wire=wire)
        pw, ph = cfg.padded
        self.pin_in = torch.zeros((batch + 1, ph, pw, 3), dtype=torch.uint8).pin_memory()  # padding stays zero
        e = self.enc
        self.pin_mv = torch.empty(e.mv.shape, dtype=torch.float32).pin_memory()
        self.pin_types = torch.empty(e.types.shape, dtype=torch.int32).pin_memory()
        self.pin_gm = torch.empty(e.gm.shape, dtype=torch.float32).pin_memory()
        big = e.records if e.wire else e.coeffs
        self.pin_big = torch.empty(big.shape, dtype=big.dtype).pin_memory() if big is not None else None
        self.h2d_done = torch.cuda.Event()
        self.compute_done = torch.cuda.Event()
        self.d2h_done = torch.cuda.Event()
        self.busy = False
        self.count = 0      # encoded frames in flight in this slot
        self.first = 0      # index of its first encoded frame in the clip


class HostStreamEncoder:
    """encode(frames) yields, batch by batch, dicts of numpy views into pinned memory:
    first (index of the batch's first encoded frame; frame 0 of a clip is tracked-only,
    libs/encoder.cpp:361-367), mv (n, blocks, 2) f32, types (n, blocks) i32, gm (n, 2) f32 and
    coeffs (n, 3, H, W) f32 (quantised) or records (n, bytes) u8 -- the serialised RAW coefficients over the
    PADDED tile grid, what the reference's decoder parses (libs/decoder.cpp:185-186); the first batch of a wire
    stream also carries "header", the 32 bytes of libs/codec.hpp:8-17.  A view is valid until depth - 2 more
    batches have been yielded (its slot is re-staged one iteration before its turn to be yielded comes again)."""

    def __init__(self, cfg: CodecConfig, batch: int = 32, device=None, wire: bool = False,
                 segmentation: bool = True, depth: int = 3):
        self.cfg, self.batch, self.depth = cfg, batch, max(3, depth)  # 2 would leave nothing overlapped
        self.dev = device or torch.device("cuda")
        self.slots = [_Slot(cfg, batch, self.dev, wire, segmentation) for _ in range(self.depth)]
        self.copy_in = torch.cuda.Stream(device=self.dev)
        self.copy_out = torch.cuda.Stream(device=self.dev)
        self.compute = torch.cuda.Stream(device=self.dev)
        e = self.slots[0].enc
        self._iters, self._subset, self._blocks = e.iters, e.ransac["subset_sz"], e.blocks
        self._seg_seed0 = e.seg_seed

    def _stage(self, slot: _Slot, frames: np.ndarray, lo: int, hi: int, carry: bool) -> int:
        """Frames [lo, hi) of the clip into the slot's pinned input; returns the encoded-frame count."""
        h, w = frames.shape[1:3]
        n = hi - lo
        dst = slot.pin_in.numpy()
        off = 1 if carry else 0  # entry 0 = the previous batch's last frame, copied on the device
        dst[off:off + n, :h, :w] = frames[lo:hi]
        return n if carry else n - 1

    def encode(self, frames: np.ndarray) -> Iterator[Dict[str, np.ndarray]]:
        """frames: (N, h, w, 3) u8, unpadded, anywhere in host memory."""
        n_total = frames.shape[0]
        assert frames.dtype == np.uint8 and frames.shape[3] == 3 and n_total >= 2
        assert frames.shape[1] <= self.cfg.padded[1] and frames.shape[2] <= self.cfg.padded[0]
        samples = ransac_samples(n_total - 1, self._iters, self._subset, self._blocks, self.cfg.seed, self.dev)
        # the draws (and the slots' zero-initialised buffers) were produced on the caller's current stream:
        # the worker streams start behind it
        cur = torch.cuda.current_stream(self.dev)
        for s in (self.copy_in, self.compute, self.copy_out):
            s.wait_stream(cur)
        self._n_total = n_total
        B = self.batch
        pending: List[_Slot] = []
        prev: Optional[_Slot] = None
        lo, k, first = 0, 0, 1
        while lo < n_total:
            slot = self.slots[k % self.depth]
            if slot.busy:  # its results were handed out `depth` batches ago: the caller is done with the views
                slot.d2h_done.synchronize()
                slot.busy = False
            carry = prev is not None
            hi = min(n_total, lo + (B if carry else B + 1))
            cnt = self._stage(slot, frames, lo, hi, carry)
            e = slot.enc
            with torch.cuda.stream(self.copy_in):
                off = 1 if carry else 0
                e.bgr[off:off + (hi - lo)].copy_(slot.pin_in[off:off + (hi - lo)], non_blocking=True)
                if carry:
                    self.copy_in.wait_event(prev.h2d_done)
                    e.bgr[0].copy_(prev.enc.bgr[prev.count], non_blocking=True)  # its last frame, either way it was laid out
                slot.h2d_done.record(self.copy_in)
            g0 = first - 1  # global pair index of this batch's first pair
            e.seg_seed = self._seg_seed0 + g0
            with torch.cuda.stream(self.compute):
                # sliced on the stream that consumes them (torch.cat is a kernel)
                e.samples = samples[g0:g0 + e.pairs_per_step] if g0 + e.pairs_per_step <= n_total - 1 else \
                    torch.cat([samples[g0:], samples[:e.pairs_per_step - (n_total - 1 - g0)]])
                self.compute.wait_event(slot.h2d_done)
                e.step()
                slot.compute_done.record(self.compute)
            with torch.cuda.stream(self.copy_out):
                self.copy_out.wait_event(slot.compute_done)
                slot.pin_mv[:cnt].copy_(e.mv[:cnt], non_blocking=True)
                slot.pin_types[:cnt].copy_(e.types[:cnt], non_blocking=True)
                slot.pin_gm[:cnt].copy_(e.gm[:cnt], non_blocking=True)
                if slot.pin_big is not None:
                    big = e.records if e.wire else e.coeffs
                    slot.pin_big[:cnt].copy_(big[:cnt], non_blocking=True)
                slot.d2h_done.record(self.copy_out)
            slot.busy, slot.count, slot.first = True, cnt, first
            pending.append(slot)
            prev, lo, k, first = slot, hi, k + 1, first + cnt
            if len(pending) >= self.depth - 1:
                yield self._finish(pending.pop(0))
        while pending:
            yield self._finish(pending.pop(0))

    def _finish(self, slot: _Slot) -> Dict[str, np.ndarray]:
        slot.d2h_done.synchronize()
        c = slot.count
        out = {"first": slot.first, "mv": slot.pin_mv.numpy()[:c], "types": slot.pin_types.numpy()[:c],
               "gm": slot.pin_gm.numpy()[:c]}
        if slot.pin_big is not None:
            out["records" if slot.enc.wire else "coeffs"] = slot.pin_big.numpy()[:c]
        if slot.enc.wire and slot.first == 1:
            c_ = self.cfg
            out["header"] = native.wire_header(self._n_total, c_.width, c_.height, c_.mv_block, c_.levels, c_.dct_block)
        return out
