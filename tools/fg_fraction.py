#!/usr/bin/env python3
"""Foreground share of the BASELINE clips: the fraction of MV blocks (and of encoded frames) whose region id is not 0 after RANSAC +
segmentation, per configuration -- what a speculative 'quantise as background, redo the foreground tiles' transform would have to redo."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_video_codec_amd import clip as clipmod, configs, native, synth
native.load(); clipmod.load()
dev = torch.device("cuda")
for name in sys.argv[1:] or ["C3-1080p-3L-dct8-quant", "C3b-1080p-4L-dct8-quant", "C5-4k-4L-dct16", "C2-720p-3L-dct8"]:
    cfg = configs.ALL[name]
    enc = clipmod.Clip(cfg, cfg.frames, schedule=clipmod.SERIAL)
    src = synth.SynthClip(cfg.width, cfg.height, cfg.frames, cfg.seed, device=dev)
    pw, ph = cfg.padded
    for j in range(cfg.frames):
        enc.load_frames(synth.pad_frame(src.frame_bgr(j), pw, ph).unsqueeze(0).contiguous(), j)
    enc.step(); enc.sync()
    t = enc.read("block_types").view(enc.info.pairs, -1)
    fg = (t != 0)
    per = fg.float().mean(dim=1)
    print(f"{name}: foreground MV blocks {fg.float().mean().item()*100:.2f} % of {t.numel()} (per frame: min {per.min().item()*100:.2f} %, median {per.median().item()*100:.2f} %, "
          f"max {per.max().item()*100:.2f} %); frames with any foreground {int((per > 0).sum())} of {t.shape[0]}; region ids up to {int(t.max())}")
    enc.close()
