"""Seed-addressed synthetic clips and the luma / pyramid pre-steps.

Everything here is integer arithmetic on torch tensors, so the same call gives
bit-identical frames on the CPU (tests, golden fixtures) and on the GPU (bench).
Randomness is a stateless 32-bit counter hash (no sequential generator), which
is what lets a 300-frame 1080p clip be produced on the device in a blink.

The clip follows SURVEY.md 8(d): a bilinear-upsampled noise canvas sampled at a
global offset that drifts (+3, -2) px per frame and turns around inside a +-24 px
margin (so every frame-to-frame global displacement is within the default search
range of 8), a few textured rectangles with their own integer velocities
(<= 6 px/frame), and +-2 uniform noise per sample.

The luma and pyramid definitions stand in for cv::cvtColor(BGR2YUV) +
cv::extractChannel + cv::buildPyramid (reference libs/encoder.cpp:449-451,
:468-470).  OpenCV is not available offline, so parity of THESE pre-steps with
OpenCV is unpinned; motion-estimation parity does not depend on it because the
reference takes the pyramids as inputs (libs/motion.hpp:134-138).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import torch

_M32 = 0xFFFFFFFF
SEED_BASE = 0x5C0DEC00
MARGIN = 32  # canvas border; the drift stays within +-24 of its centre


def hash32(x: torch.Tensor) -> torch.Tensor:
    """Stateless 32-bit mixer on int64 tensors (values taken mod 2**32)."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _hash_scalar(x: int) -> int:
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def _noise_u8(seed: int, shape: Tuple[int, ...], device) -> torch.Tensor:
    n = math.prod(shape)
    idx = torch.arange(n, dtype=torch.int64, device=device)
    return (hash32(idx + (seed & _M32) * 0x9E3779B1) >> 11 & 0xFF).reshape(shape)


def _upsample_bilinear_int(lat: torch.Tensor, factor: int, out_h: int, out_w: int) -> torch.Tensor:
    """Exact integer bilinear upsample of a (C, h, w) int64 lattice by `factor`."""
    ys = torch.arange(out_h, device=lat.device)
    xs = torch.arange(out_w, device=lat.device)
    y0, fy = ys // factor, (ys % factor).view(1, -1, 1)
    x0, fx = xs // factor, (xs % factor).view(1, 1, -1)
    a = lat[:, y0][:, :, x0]
    b = lat[:, y0][:, :, x0 + 1]
    c = lat[:, y0 + 1][:, :, x0]
    d = lat[:, y0 + 1][:, :, x0 + 1]
    f = factor
    top = a * (f - fx) + b * fx
    bot = c * (f - fx) + d * fx
    return (top * (f - fy) + bot * fy + (f * f) // 2) // (f * f)


@dataclass
class _Rect:
    w: int
    h: int
    x: int
    y: int
    vx: int
    vy: int
    tex: torch.Tensor  # (3, h, w) int64


class SynthClip:
    """A deterministic BGR u8 clip of `n_frames` frames of `width` x `height`."""

    def __init__(self, width: int, height: int, n_frames: int, seed: int, device="cpu"):
        self.w, self.h, self.n = int(width), int(height), int(n_frames)
        self.seed = int(seed) & _M32
        self.device = torch.device(device)
        cw, ch = self.w + 2 * MARGIN, self.h + 2 * MARGIN
        lat = _noise_u8(self.seed ^ 0xA5A5, (3, ch // 8 + 2, cw // 8 + 2), self.device)
        self.canvas = _upsample_bilinear_int(lat, 8, ch, cw)  # (3, ch, cw)
        # drift trajectory (host ints): step (+3, -2), reflect inside +-24
        self.offsets: List[Tuple[int, int]] = []
        ox = oy = 0
        sx, sy = 3, -2
        for _ in range(self.n):
            self.offsets.append((ox, oy))
            if abs(ox + sx) > 24:
                sx = -sx
            if abs(oy + sy) > 24:
                sy = -sy
            ox += sx
            oy += sy
        # moving rectangles
        k = 2 + _hash_scalar(self.seed ^ 0x51) % 3
        self.rects: List[_Rect] = []
        for r in range(k):
            hsh = lambda j: _hash_scalar(self.seed * 31 + r * 977 + j)  # noqa: E731
            rw = min(96 + hsh(1) % 161, max(8, self.w // 3))
            rh = min(64 + hsh(2) % 97, max(8, self.h // 3))
            x = hsh(3) % (self.w - rw)
            y = hsh(4) % (self.h - rh)
            vx = hsh(5) % 13 - 6
            vy = hsh(6) % 13 - 6
            tl = _noise_u8(self.seed ^ (0xBEEF + r), (3, rh // 4 + 2, rw // 4 + 2), self.device)
            tex = _upsample_bilinear_int(tl, 4, rh, rw)
            tex = (tex + 64 * ((r % 3) - 1)).clamp(0, 255)
            self.rects.append(_Rect(rw, rh, x, y, vx, vy, tex))

    def _rect_pos(self, r: _Rect, t: int) -> Tuple[int, int]:
        """Position after t steps with reflection at the frame border."""
        def refl(p0: int, v: int, span: int) -> int:
            if span <= 0:
                return 0
            p = (p0 + v * t) % (2 * span)
            return p if p <= span else 2 * span - p
        return refl(r.x, r.vx, self.w - r.w), refl(r.y, r.vy, self.h - r.h)

    def frame_bgr(self, t: int) -> torch.Tensor:
        """Frame t as an (H, W, 3) uint8 tensor in B, G, R order."""
        ox, oy = self.offsets[t]
        x0, y0 = MARGIN + ox, MARGIN + oy
        img = self.canvas[:, y0:y0 + self.h, x0:x0 + self.w].clone()
        for r in self.rects:
            rx, ry = self._rect_pos(r, t)
            img[:, ry:ry + r.h, rx:rx + r.w] = r.tex
        n = 3 * self.h * self.w
        idx = torch.arange(n, dtype=torch.int64, device=self.device)
        noise = hash32(idx + ((self.seed + 0x1234567 * (t + 1)) & _M32) * 0x85EBCA6B) >> 9
        img = (img.reshape(-1) + (noise % 5) - 2).clamp(0, 255).reshape(3, self.h, self.w)
        return img.permute(1, 2, 0).contiguous().to(torch.uint8)


# ---- reference pre-steps (our fixed-point definitions) -----------------------

def closest_larger_divisible(a: int, x: int, y: int) -> int:
    """libs/math.hpp:276-283 (ClosestLargerDivisible): round up to lcm(x, y)."""
    l = math.lcm(x, y)
    return (a + l - 1) // l * l


def padded_dims(w: int, h: int, block_w: int, block_h: int, levels: int) -> Tuple[int, int]:
    """libs/encoder.cpp:164-168: padded frame size for a level count."""
    f = 1 << (levels - 1)
    return closest_larger_divisible(w, block_w, f), closest_larger_divisible(h, block_h, f)


def pad_frame(bgr: torch.Tensor, pw: int, ph: int) -> torch.Tensor:
    """libs/encoder.cpp:459-461: zero border on the right and bottom."""
    h, w, _ = bgr.shape
    if (pw, ph) == (w, h):
        return bgr
    out = torch.zeros((ph, pw, 3), dtype=bgr.dtype, device=bgr.device)
    out[:h, :w] = bgr
    return out


def bgr_to_y(bgr: torch.Tensor) -> torch.Tensor:
    """Fixed-point BT.601 luma, Y = (1868 B + 9617 G + 4899 R + 8192) >> 14."""
    p = bgr.to(torch.int32)
    y = (p[..., 0] * 1868 + p[..., 1] * 9617 + p[..., 2] * 4899 + 8192) >> 14
    return y.to(torch.uint8)


def pyr_down(plane: torch.Tensor) -> torch.Tensor:
    """5x5 [1 4 6 4 1]^2 kernel, BORDER_REFLECT_101, (s + 128) >> 8, even samples."""
    h, w = plane.shape
    p = plane.to(torch.int32)
    taps = (1, 4, 6, 4, 1)

    def refl(i: torch.Tensor, n: int) -> torch.Tensor:
        i = i.abs()
        return torch.where(i >= n, 2 * (n - 1) - i, i)

    xs = torch.arange(0, w, 2, device=plane.device)
    ys = torch.arange(0, h, 2, device=plane.device)
    rows = sum(t * p[:, refl(xs + d - 2, w)] for d, t in enumerate(taps))
    out = sum(t * rows[refl(ys + d - 2, h), :] for d, t in enumerate(taps))
    return ((out + 128) >> 8).to(torch.uint8)


def build_pyramid(y: torch.Tensor, levels: int) -> List[torch.Tensor]:
    """libs/encoder.cpp:470 (cv::buildPyramid, maxlevel = levels - 1)."""
    pyr = [y.contiguous()]
    for _ in range(levels - 1):
        pyr.append(pyr_down(pyr[-1]).contiguous())
    return pyr


def pyramid_bytes(w: int, h: int, levels: int) -> int:
    return sum((w >> l) * (h >> l) for l in range(levels))


def pack_pyramid(pyr: List[torch.Tensor]) -> torch.Tensor:
    """Level planes back to back (level 0 first): the packed layout the device
    entry points take (include/svc_hip.h)."""
    return torch.cat([p.reshape(-1) for p in pyr])


def level_offsets(w: int, h: int, levels: int) -> List[int]:
    offs, o = [], 0
    for l in range(levels):
        offs.append(o)
        o += (w >> l) * (h >> l)
    return offs
