#!/usr/bin/env python3
"""Does the plane-to-plane pyramid pass pay for the kernel in front of it?  (round 6)

In the step it follows the transform, which has just written 10 GB (81 % of the step's bytes); whatever runs next shares the memory system with
the write-back of what that kernel left dirty in L2 / Infinity Cache.  This times svc_hip_pyramid_levels_frames over a C3-sized clip of luma
planes three ways, HIP events around each launch sequence:
  alone      ten calls back to back (nothing else in flight);
  after_fill each call right after a 7.5 GB store-only kernel (torch fill_ of a coefficient-sized buffer) -- the transform's write stream;
  after_read each call right after a 7.5 GB read-only reduction (torch sum) -- a big kernel that leaves nothing dirty.
usage: python tools/ubench_pyr_standalone.py [frames] [width] [height] [levels]
"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from scalable_video_codec_amd import native

n, w, h, levels = (int(a) for a in (sys.argv[1:5] + ["299", "1920", "1088", "3"][len(sys.argv) - 1:]))
native.load()
dev = torch.device("cuda")
stride = native.pyramid_stride(w, h, levels)
pyr = torch.randint(0, 255, (n * stride,), dtype=torch.uint8, device=dev)
big = torch.empty(n * 3 * w * h, dtype=torch.float32, device=dev)


def timed(before):
    out = []
    for _ in range(10):
        if before:
            before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.pyramid_levels_frames(pyr, stride, n, w, h, levels)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


native.pyramid_levels_frames(pyr, stride, n, w, h, levels)
torch.cuda.synchronize()
bytes_alg = n * sum((w >> l) * (h >> l) + (w >> (l + 1)) * (h >> (l + 1)) for l in range(levels - 1))
for name, before in (("alone", None), ("after_fill", lambda: big.fill_(1.0)), ("after_read", lambda: big.sum()), ("alone again", None)):
    med, lo, hi = timed(before)
    print(f"{name:12s} median {med:.4f} ms (min {lo:.4f}, max {hi:.4f})  = {bytes_alg / med / 1e6:.0f} GB/s of the pass's {bytes_alg / 1e9:.3f} GB")
