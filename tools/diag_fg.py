"""Foreground-block distribution per frame after one encode pass (diagnostic for the segmentation critical path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_video_codec_amd import configs, pipeline, synth
for name in sys.argv[1:]:
    cfg = configs.ALL[name]
    n = cfg.frames
    dev = torch.device("cuda")
    clip = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
    pw, ph = cfg.padded
    enc = pipeline.ClipEncoder(cfg, n, dev)
    enc.load_frames([synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(n)])
    enc.step(); torch.cuda.synchronize()
    fg = (enc.types != 0).sum(1).cpu()
    out = (enc.mask == 0).sum(1).cpu() if enc.mask.dtype != torch.bool else (~enc.mask).sum(1).cpu()
    q = torch.tensor([0.0, 0.5, 0.9, 0.99, 1.0])
    print(name, "blocks", cfg.blocks, "fg quantiles (0,50,90,99,100):", torch.quantile(fg.float(), q).tolist(),
          "outliers:", torch.quantile(out.float(), q).tolist(), "frames>2000:", int((fg > 2000).sum()))
