"""Phase timeline of the labelling kernel (segment_label_kernel) on the frames of a shard.  Needs a library whose
segment.hip was compiled with -DSVC_SEG_TIMING (diagnostic only):
  SVC_EXTRA_HIPCC_FLAGS=-DSVC_SEG_TIMING python -c "from scalable_video_codec_amd import build; build.build_hip(force=True)"
usage: diag_label_phases.py [config [frames]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scalable_video_codec_amd import configs, native, pipeline, synth

cfg = configs.ALL[sys.argv[1] if len(sys.argv) > 1 else "C5-4k-4L-dct16"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda")
clip = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
pw, ph = cfg.padded
enc = pipeline.ClipEncoder(cfg, n, dev)
enc.load_frames([synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(n)])
enc.step()
torch.cuda.synchronize()
nb, A = enc.mfw * enc.mfh, 3
a16 = lambda v: (v + 15) & ~15
# Workspace (segment.hip): header 256 | idx 4n | pk 4n | lab A n | pts 12n | dmin 4 A n | cl n | parent 4n | roots 4n | ...
off_roots = 256 + a16(4 * nb) + a16(4 * nb) + a16(A * nb) + a16(12 * nb) + a16(4 * A * nb) + a16(nb) + a16(4 * nb)
names = ["start", "scatter", "runs+unions", "flatten", "roots", "numbered", "end"]
fg = (enc.types != 0).sum(1)
for f in range(enc.mask.shape[0]):
    m1, v1 = enc.mask[f:f + 1].contiguous(), enc.mv[f:f + 1].contiguous()
    ws = torch.zeros(native.segment_workspace_bytes(enc.mfw, enc.mfh, 1, A), dtype=torch.uint8, device=dev)
    for _ in range(2):
        native.segment_frames(m1, v1, enc.mfw, enc.mfh, seed=1, workspace=ws)
    torch.cuda.synchronize()
    raw = ws.cpu().numpy()
    st = raw[off_roots + 4 * ((nb - 64) & ~1):][:64].view(np.uint64).astype(np.int64)
    print(f"pair {f}: fg blocks {int(fg[f])}  " + " ".join(f"{nm}={st[i] - st[0]}" for i, nm in enumerate(names) if i) + f"  components={st[7]}", flush=True)
