"""PCIe-inclusive rate of the batched host-memory encoder (scalable_video_codec_amd/stream.py) at 1080p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scalable_video_codec_amd import configs, stream, synth
cfg = configs.C3
n = 257
dev = torch.device("cuda")
clip = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
host = torch.stack([clip.frame_bgr(t) for t in range(n)]).cpu().numpy()
for wire in (False, True):
    for batch in (16, 32, 64):
        enc = stream.HostStreamEncoder(cfg, batch=batch, device=dev, wire=wire)
        for _ in enc.encode(host[:2 * batch + 2]): pass
        t0 = time.perf_counter()
        got = 0
        for out in enc.encode(host): got += out["mv"].shape[0]
        dt = time.perf_counter() - t0
        print(f"wire={wire} batch={batch}: {got / dt:.0f} frames/s PCIe-inclusive ({got} frames in {dt * 1e3:.0f} ms)", flush=True)
        del enc
