"""Segmentation latency of single frames (heaviest / median foreground) of a config's synthetic clip, and the
throughput of 300 copies of each (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_video_codec_amd import configs, native, pipeline, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C3b-1080p-4L-dct8-quant"
cfg = configs.ALL[name]
n = cfg.frames
dev = torch.device("cuda")
clip = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
pw, ph = cfg.padded
enc = pipeline.ClipEncoder(cfg, n, dev)
enc.load_frames([synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(n)])
enc.step(); torch.cuda.synchronize()
fg = (enc.types != 0).sum(1)
order = torch.argsort(fg)
def t(mask, mv, **kw):
    native.segment_frames(mask, mv, enc.mfw, enc.mfh, seed=1, **kw); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): native.segment_frames(mask, mv, enc.mfw, enc.mfh, seed=1, **kw)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 5 * 1e3, 1)
for label, f in (("heaviest", int(order[-1])), ("p90", int(order[int(0.9 * (len(order) - 1))])), ("median", int(order[len(order) // 2])), ("lightest", int(order[0]))):
    m1, v1 = enc.mask[f:f + 1].contiguous(), enc.mv[f:f + 1].contiguous()
    mN, vN = m1.repeat(300, 1).contiguous(), v1.repeat(300, 1, 1).contiguous()
    print(label, "frame", f, "fg", int(fg[f]), "us: 1 frame", t(m1, v1), "| 1 attempt", t(m1, v1, attempt_count=1),
          "| 1 attempt 1 iter", t(m1, v1, attempt_count=1, max_iter_count=1), "| k=1 1 iter", t(m1, v1, attempt_count=1, max_iter_count=1, cluster_count=1),
          "| 300 copies", t(mN, vN), "| 300 copies 1 attempt", t(mN, vN, attempt_count=1))
