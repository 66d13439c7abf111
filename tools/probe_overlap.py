#!/usr/bin/env python3
"""Probe (round 5): would the pyramid-from-the-plane pass (VALU / LDS-bound, 0.46 of the HBM peak) hide beside the front-of-step transform
(store-bound, VALU 61 % busy) if the two ran concurrently on two streams?  Times, on independent buffers of C3's size: the transform alone, the
pyramid levels alone, both launched back to back on two streams (wall time of the pair)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from scalable_video_codec_amd import configs, native, synth

cfg = configs.C3
lib = native.load()
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 299
pw, ph = cfg.padded
src = synth.SynthClip(cfg.width, cfg.height, 8, cfg.seed, device=dev)
one = torch.stack([synth.pad_frame(src.frame_bgr(t), pw, ph) for t in range(8)])
bgr = one.repeat((n + 7) // 8, 1, 1, 1)[:n].contiguous()
planes = torch.empty((n, 3, ph, pw), dtype=torch.float32, device=dev)
stride = native.pyramid_stride(pw, ph, cfg.levels)
pyr_a = torch.empty(n * stride, dtype=torch.uint8, device=dev)
pyr_b = torch.empty(n * stride, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
vp = C.c_void_p


def f(stream):
    native._check(lib.svc_hip_dct_quant_luma_frames(bgr.data_ptr(), ph * pw * 3, n, pw, ph, 8, 640, planes.data_ptr(), pyr_a.data_ptr(), stride, vp(stream.cuda_stream)))


def p(stream):
    native._check(lib.svc_hip_pyramid_levels_frames(pyr_b.data_ptr(), stride, n, pw, ph, cfg.levels, vp(stream.cuda_stream)))


f(s1); torch.cuda.synchronize(); pyr_b.copy_(pyr_a); torch.cuda.synchronize()


def timed(fn, reps=8):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        s1.wait_event(e0); s2.wait_event(e0)
        fn()
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(s1); eb.record(s2)
        torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    out.sort()
    return out[len(out) // 2], out[0]


for name, fn in (("transform alone", lambda: f(s1)), ("pyramid levels alone", lambda: p(s2)), ("sequential on one stream", lambda: (f(s1), p(s1))),
                 ("both, two streams (transform first)", lambda: (f(s1), p(s2))), ("both, two streams (pyramid first)", lambda: (p(s2), f(s1)))):
    med, best = timed(fn)
    print(f"{name:42s} median {med:.4f} ms  best {best:.4f} ms", flush=True)
