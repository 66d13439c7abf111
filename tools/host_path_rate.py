"""Times the host-pointer entry points (pinned staging + H2D + kernel + D2H + sync per call), i.e. what
the C++ drop-in wrappers of include/svc/motion.hpp cost per frame at 1080p.  PCIe-inclusive."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scalable_video_codec_amd import configs, native, synth
cfg = configs.C3
clip = synth.SynthClip(cfg.width, cfg.height, 2, cfg.seed)
pw, ph = cfg.padded
fr = [synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(2)]
pyr = [[p.numpy() for p in synth.build_pyramid(synth.bgr_to_y(f), cfg.levels)] for f in fr]
bgr = fr[1].numpy()
types = np.zeros(cfg.blocks, np.uint32)
def t(fn, n=20):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
a = t(lambda: native.hbma_host(pyr[0], pyr[1], 8, 16, 16))
b = t(lambda: native.dct_quant_host(bgr, 8, types, 16, 1, 640))
mv, _ = native.hbma_host(pyr[0], pyr[1], 8, 16, 16)
s = (np.arange(7, dtype=np.uint32) * 977) % cfg.blocks
c = t(lambda: native.ransac_host(mv, s))
print(f"hbma_host {a:.3f} ms  dct_quant_host {b:.3f} ms  ransac_host {c:.3f} ms  -> {1e3 / (a + b + c):.0f} frames/s through the synchronous host-pointer API")
