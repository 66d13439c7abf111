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

# the same schedule as a C++ host application (tests/dropin/stream_main.cpp, include/svc/stream_encoder.hpp)
import subprocess, tempfile
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "dropin", "stream_main")
if os.path.exists(exe):
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "clip.raw")
        host[:193].tofile(raw)
        for wire in (0, 1):
            r = subprocess.run([exe, raw, str(cfg.width), str(cfg.height), "193", str(cfg.levels), str(cfg.dct_block), str(wire), "16",
                                str(cfg.seed), os.path.join(d, "out")], capture_output=True, text=True)
            print(f"C++ StreamEncoder wire={wire} batch=16:", (r.stdout + r.stderr).strip(), flush=True)
