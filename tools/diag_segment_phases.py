"""Phase timeline of the attempt kernel on single frames of a config's clip.  Needs a library whose segment.hip was
compiled with -DSVC_SEG_TIMING (diagnostic only):
  hipcc <flags of build.py> -DSVC_SEG_TIMING -c csrc/segment.hip -o _obj/segment.o && relink libsvc_hip.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
nb = enc.mfw * enc.mfh
A = 3
off_lab = 256 + ((4 * nb + 15) & ~15)
for label, f in (("heaviest", int(order[-1])), ("p90", int(order[int(0.9 * (len(order) - 1))])), ("median", int(order[len(order) // 2]))):
    m1, v1 = enc.mask[f:f + 1].contiguous(), enc.mv[f:f + 1].contiguous()
    ws = torch.zeros(native.segment_workspace_bytes(enc.mfw, enc.mfh, 1, A), dtype=torch.uint8, device=dev)
    for _ in range(2):
        native.segment_frames(m1, v1, enc.mfw, enc.mfh, seed=1, workspace=ws)
    torch.cuda.synchronize()
    raw = ws.cpu().numpy()
    print(label, "frame", f, "fg", int(fg[f]))
    for att in range(A):
        o = off_lab + att * nb + ((nb - 256) & ~7)
        st = raw[o:o + 256].view(np.uint64).astype(np.int64)
        rel = (st - st[0])
        names = ["start", "morph", "list", "pre-kmeans", "seeded"] + [f"it{i//2}{'a' if i%2==0 else 'b'}" for i in range(20)]
        line = " ".join(f"{names[i]}={rel[i]}" for i in range(1, 25) if st[i] > 0)
        print(f"  att {att}: {line} end={rel[31]}")
    off_roots = raw.size  # roots is the last array of the (single-frame) workspace
    o = ((raw.size - 255) // 256) * 256  # recompute from the layout: header 256 | idx | lab | pts | dmin | cl | parent | roots
    a16 = lambda v: (v + 15) & ~15
    off = 256 + a16(4 * nb); off += a16(A * nb); off += a16(12 * A * nb); off += a16(4 * A * nb); off += a16(nb); off += a16(4 * nb)
    st = raw[off + 4 * ((nb - 64) & ~1):][:64].view(np.uint64).astype(np.int64)
    print("  label: " + " ".join(f"{nm}={st[i] - st[0]}" for i, nm in enumerate(["start", "scatter", "merge", "flatten", "roots", "numbered", "end"]) if i) + f" components={st[7]}")
