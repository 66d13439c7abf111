"""DCT kernel time with and without the quantiser, 8x8 and 16x16, on a 1080p clip (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scalable_video_codec_amd import native
dev = torch.device("cuda")
n, h, w = 299, 1088, 1920
bgr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev)
types = torch.randint(0, 3, (n, (h // 16) * (w // 16)), dtype=torch.int32, device=dev)
out = torch.empty(n * 3 * h * w + 4 * 1152 * n, dtype=torch.float32, device=dev)[:n * 3 * h * w].view(n, 3, h, w)
def t(f):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5
by = n * h * w * 15
for blk in (8, 16):
    a = t(lambda: native.dct_frames(bgr, blk, out=out))
    b = t(lambda: native.dct_quant_frames(bgr, blk, types, 16, 1, 640, out=out))
    print(f"dct{blk}: plain {a:.3f} ms ({by / a / 1e9:.2f} TB/s)   +quant {b:.3f} ms ({by / b / 1e9:.2f} TB/s)")
print(f"fill of the same output: {t(lambda: out.fill_(0.0)):.3f} ms")
