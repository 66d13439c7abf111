"""RANSAC launch time against the number of frames in the batch (diagnostic: one workgroup per frame, so the time
should not grow until the frames outnumber the workgroups the chip can hold at once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scalable_video_codec_amd import native
rng = np.random.default_rng(0)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8160
for frames in (1, 32, 128, 256, 299, 512, 1024):
    mv = torch.from_numpy(rng.integers(-8, 9, (frames, blocks, 2)).astype(np.float32)).cuda()
    samples = torch.from_numpy(rng.integers(0, blocks, (frames, 7, 1)).astype(np.int32)).cuda()
    out = native.ransac_frames(mv, samples)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): native.ransac_frames(mv, samples, out=out)
    b.record(); torch.cuda.synchronize()
    print(f"{frames:5d} frames: {a.elapsed_time(b) / 10 * 1e3:7.1f} us per launch", flush=True)
