"""Shared helpers for the parity tests: seeded clips -> numpy pyramids."""
from __future__ import annotations

import os

import numpy as np
import torch

from scalable_video_codec_amd import synth


def clip_frames(w, h, n, seed, levels, block=16, device="cpu"):
    """Returns (padded BGR frames [n] as (H,W,3) u8 tensors, pyramids [n][levels] u8 tensors, (pw, ph))."""
    pw, ph = synth.padded_dims(w, h, block, block, levels)
    clip = synth.SynthClip(w, h, n, seed, device=device)
    frames = [synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(n)]
    pyrs = [synth.build_pyramid(synth.bgr_to_y(f), levels) for f in frames]
    return frames, pyrs, (pw, ph)


def np_pyr(pyr):
    return [p.cpu().numpy() for p in pyr]


def random_planes(rng, w, h, levels):
    """Independent uniform-noise pyramids (worst case for ties: none; stress for windows)."""
    return [rng.integers(0, 256, (h >> l, w >> l), dtype=np.uint8) for l in range(levels)]


def pack_clip(pyrs, stride, device):
    """pyrs[n][levels] (tensors or arrays) -> one flat u8 device buffer of n packed pyramids."""
    n = len(pyrs)
    buf = torch.zeros(n * stride, dtype=torch.uint8)
    for i, pyr in enumerate(pyrs):
        flat = torch.cat([torch.as_tensor(np.ascontiguousarray(p) if isinstance(p, np.ndarray) else p.cpu()).reshape(-1)
                          for p in pyr])
        buf[i * stride:i * stride + flat.numel()] = flat
    return buf.to(device)


# One-off deep fuzz of the hypothesis suites: SVC_FUZZ_SCALE=10 SVC_FUZZ_RANDOM=1 python -m pytest tests -m gpu -k "property or random"
# (default: the committed example counts, derandomised so that every run checks the same cases).
FUZZ_RANDOM = os.environ.get("SVC_FUZZ_RANDOM", "0") == "1"


def fuzz_examples(n: int) -> int:
    return n * max(1, int(os.environ.get("SVC_FUZZ_SCALE", "1")))
