"""Where does the hipGraph replay of the steady-state iteration fall over?  Steps one at a time with a sync and a
progress line after each, faulthandler on."""
import faulthandler
import sys

import torch

sys.path.insert(0, ".")
from scalable_video_codec_amd import clip as clipmod, configs, synth  # noqa: E402

faulthandler.enable()
cfg = configs.ALL[sys.argv[3]] if len(sys.argv) > 3 else configs.CodecConfig("t-360p-3L-dct8", 41, 640, 360, 11, levels=3, dct_block=8)
dev = torch.device("cuda")
n = 10
src = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
pw, ph = cfg.padded
frames = torch.stack([synth.pad_frame(src.frame_bgr(t), pw, ph) for t in range(n)]).contiguous()
plan = sys.argv[1] if len(sys.argv) > 1 else "4,6"
nosync = len(sys.argv) > 2 and sys.argv[2] == "nosync"
enc = clipmod.Clip(cfg, n, graph=True)
enc.load_frames(frames)
for phase, k in enumerate(int(x) for x in plan.split(",")):
    for s in range(k):
        enc.step()
        print(f"phase {phase} step {s} enqueued", flush=True)
        if not nosync:
            torch.cuda.synchronize()
            print(f"phase {phase} step {s} done", flush=True)
    enc.sync()
    print(f"phase {phase} synced", flush=True)
enc.close()
print("closed", flush=True)
