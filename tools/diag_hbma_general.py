"""Rates of the motion search outside the fused kernel's shapes (the per-level wave-per-block kernel) at 1080p (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_video_codec_amd import native, synth
dev = torch.device("cuda")
n = 33
clip = synth.SynthClip(1920, 1080, n, 1234, device=dev)
def run(levels, block, rng, flags=0):
    pw, ph = synth.padded_dims(1920, 1080, block, block, levels)
    frames = torch.stack([synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(n)])
    stride = native.pyramid_stride(pw, ph, levels)
    pyr = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    native.luma_pyramid_frames(frames, levels, out=pyr, stride=stride)
    f = lambda: native.hbma_pairs(pyr, pyr[stride:], stride, n - 1, levels, pw, ph, rng, block, block, flags=flags)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"L={levels} block={block} R={rng} flags={flags}: {ms / (n - 1) * 1e3:.1f} us/pair  ({(n - 1) / ms * 1e3:.0f} pairs/s)", flush=True)
run(3, 16, 8); run(3, 16, 8, 1); run(4, 16, 8); run(4, 16, 8, 1)
run(2, 16, 8); run(1, 16, 8); run(3, 8, 8); run(2, 32, 8); run(3, 16, 16)
