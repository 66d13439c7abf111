"""Per-pair cost of the motion search at 1080p for every shape apps/encoder.cpp:75-104 admits at low cost: the lane-per-block
all-level kernel (hbma_fused_kernel<MB, L, RT>) against the per-level LDS-staged kernel (hbma_wave.hip) on the same clip.
Diagnostic (GPU box): python tools/diag_hbma_general.py > gpurun_out/hbma_shapes.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_video_codec_amd import native, synth
dev = torch.device("cuda")
n = 65
clip = synth.SynthClip(1920, 1080, n, 1234, device=dev)


def run(levels, block, rng, flags):
    pw, ph = synth.padded_dims(1920, 1080, block, block, levels)
    frames = torch.stack([synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(n)])
    stride = native.pyramid_stride(pw, ph, levels)
    pyr = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
    native.luma_pyramid_frames(frames, levels, out=pyr, stride=stride)
    f = lambda: native.hbma_pairs(pyr, pyr[stride:], stride, n - 1, levels, pw, ph, rng, block, block, flags=flags)
    try:
        f()
    except native.SvcError:
        return None
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / (n - 1) * 1e3


print("| MV block | levels | search range (R_top) | lane-per-block, us/pair | per-level kernel, us/pair |")
print("|---|---|---|---|---|")
for block in (8, 16, 32):
    for levels in (1, 2, 3, 4, 5):
        if (block >> (levels - 1)) < 1:
            continue
        for rng in (4, 8, 16):
            if rng < (1 << (levels - 1)):
                continue
            fused = run(levels, block, rng, native.HBMA_FORCE_FUSED)
            wave = run(levels, block, rng, native.HBMA_FORCE_WAVE_PER_BLOCK)
            fs = f"{fused:.2f}" if fused is not None else "-- (not instantiated)"
            print(f"| {block} | {levels} | {rng} ({rng >> (levels - 1)}) | {fs} | {wave:.2f} |", flush=True)
