"""Achievable HBM rates on this GPU with plain torch ops: fill (write only), copy (1:1), read-mostly reduction (diagnostic,
puts the per-kernel roofline fractions in context)."""
import torch
dev = torch.device("cuda")
n = 2 * 1024 ** 3  # 8 GiB of f32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
by = 4 * n
print(f"fill   {by / t(lambda: a.fill_(1.0)) / 1e12:.2f} TB/s written")
print(f"copy   {2 * by / t(lambda: b.copy_(a)) / 1e12:.2f} TB/s read+written")
print(f"sum    {by / t(lambda: a.sum()) / 1e12:.2f} TB/s read")
c = torch.empty(n // 4, dtype=torch.float32, device=dev)
print(f"1r:4w  {(by + by // 4) / t(lambda: torch.mul(c.repeat(4), 2.0, out=a)) / 1e12:.2f} TB/s (repeat+mul: upper bound only)")
