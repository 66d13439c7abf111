"""Times the reference / restated CPU motion search on this host (diagnostic)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.binding import Reference, Oracle
o = Oracle()
r = Reference() if Reference.available() else None
rng = np.random.default_rng(0)
def pl(L): return [rng.integers(0, 256, (1088 >> l, 1920 >> l), dtype=np.uint8) for l in range(L)]
t4, a4 = pl(4), pl(4)
cases = [("oracle sse2 L4", lambda: o.hbma16_sse2(t4, a4, 8)), ("oracle generic L3", lambda: o.hbma(t4[:3], a4[:3], 8, 16, 16))]
if r:
    cases += [("ref sse2 L4", lambda: r.hbma16_sse2(t4, a4, 8)), ("ref generic L4", lambda: r.hbma(t4, a4, 8, 16, 16)),
              ("ref generic L3", lambda: r.hbma(t4[:3], a4[:3], 8, 16, 16))]
for name, fn in cases:
    fn(); t0 = time.perf_counter()
    for _ in range(3): fn()
    print(name, round((time.perf_counter() - t0) / 3 * 1e3, 2), "ms", flush=True)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|MHz' | head -5; nproc")
