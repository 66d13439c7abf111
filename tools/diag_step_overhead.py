"""What do the per-stage timing events cost per step?  Same clip, pipelined schedule, steps enqueued back to back:
timed (two HIP events around every stage) vs untimed vs hipGraph replay, at the shard sizes of 1 and 8 ranks."""
import sys
import time

import torch

sys.path.insert(0, ".")
from scalable_video_codec_amd import clip as clipmod, configs, synth  # noqa: E402

# usage: diag_step_overhead.py [config name [frames ...]]   (default: C3 at 38 and 300 frames)
cfg = configs.ALL[sys.argv[1]] if len(sys.argv) > 1 else configs.C3
sizes = [int(a) for a in sys.argv[2:]] or [38, 300]
dev = torch.device("cuda")
pw, ph = cfg.padded
for n in sizes:
    src = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
    frames = torch.stack([synth.pad_frame(src.frame_bgr(t), pw, ph) for t in range(n)]).contiguous()
    for label, kw, timed in (("timed", {}, True), ("untimed", {}, False),
                             ("serial timed", {"schedule": clipmod.SERIAL}, True), ("serial untimed", {"schedule": clipmod.SERIAL}, False)):
        enc = clipmod.Clip(cfg, n, **kw)
        enc.load_frames(frames)
        for _ in range(10):
            enc.step()
        enc.sync()
        torch.cuda.synchronize()
        k = 100 if n <= 38 else 30
        t0 = time.perf_counter()
        for _ in range(k):
            enc.step(timed=timed)
        t_enq = time.perf_counter() - t0
        enc.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"frames {n:3d} {label:15s} {dt / k * 1e3:.4f} ms/step   host enqueue {t_enq / k * 1e3:.4f} ms/step", flush=True)
        enc.close()
    del frames
