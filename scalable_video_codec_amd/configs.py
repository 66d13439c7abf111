"""The BASELINE.json workload configurations.

Defaults that the configs do not override are the reference's
(apps/encoder.cpp:28-58): 16x16 MV blocks, search range 8, RANSAC
{n=1, thresh=7.5, p=0.99, w=0.5}; decoder quant steps fg 1 / bg 640
(apps/decoder.cpp:22-23).
"""
from __future__ import annotations

from dataclasses import dataclass

from .synth import SEED_BASE, padded_dims


@dataclass(frozen=True)
class CodecConfig:
    name: str
    cfg_id: int
    width: int          # source size
    height: int
    frames: int         # clip length; an N-frame clip encodes N-1 frames (encoder.cpp:361-367)
    levels: int         # pyr-lvl-count
    mv_block: int = 16
    search_range: int = 8
    dct_block: int = 8  # transform-block-w/h; 0 = no DCT in this config
    fg_step: int = 1
    bg_step: int = 640

    @property
    def seed(self) -> int:
        return SEED_BASE + self.cfg_id

    @property
    def padded(self):
        return padded_dims(self.width, self.height, self.mv_block, self.mv_block, self.levels)

    @property
    def mv_field(self):
        pw, ph = self.padded
        return pw // self.mv_block, ph // self.mv_block

    @property
    def blocks(self) -> int:
        fw, fh = self.mv_field
        return fw * fh

    @property
    def r_top(self) -> int:
        return self.search_range >> (self.levels - 1)

    def hbma_bytes_per_frame(self) -> int:
        """Algorithmic bytes of one HBMA frame pair: each level of
        both pyramids read once + 12 B of output per MV block."""
        pw, ph = self.padded
        return 2 * sum((pw >> l) * (ph >> l) for l in range(self.levels)) + 12 * self.blocks

    def luma_pyramid_bytes_per_frame(self) -> int:
        """u8 BGR in, every pyramid level out once (level l + 1 is built from level l while it is on chip)."""
        pw, ph = self.padded
        return 3 * pw * ph + sum((pw >> l) * (ph >> l) for l in range(self.levels))

    def dct_bytes_per_frame(self) -> int:
        """u8 BGR in, f32 planar out: 5 * 3 * W * H (+ 4 B per MV block of types)."""
        pw, ph = self.padded
        return 15 * pw * ph + 4 * self.blocks


C1 = CodecConfig("C1-cif-1L", 1, 352, 288, 2, levels=1, dct_block=0)
C2 = CodecConfig("C2-720p-3L-dct8", 2, 1280, 720, 30, levels=3, dct_block=8)
C3 = CodecConfig("C3-1080p-3L-dct8-quant", 3, 1920, 1080, 300, levels=3, dct_block=8)
C5 = CodecConfig("C5-4k-4L-dct16", 5, 3840, 2160, 64, levels=4, dct_block=16)
# the reference's default build (SSE2 path): 4 levels, 16x16 (libs/motion.hpp:143-147)
C3_L4 = CodecConfig("C3b-1080p-4L-dct8-quant", 6, 1920, 1080, 300, levels=4, dct_block=8)

# not a BASELINE configuration: PAL with the reference's default build -- a frame 16 mod 32 pixels wide, whose level-3 plane (90 wide) is
# not a whole number of dwords: the shape the fast motion-search kernels do not take (the general per-level kernel serves it)
X_PAL = CodecConfig("X1-pal-720x576-4L-dct8", 7, 720, 576, 300, levels=4, dct_block=8)

ALL = {c.name: c for c in (C1, C2, C3, C5, C3_L4, X_PAL)}
