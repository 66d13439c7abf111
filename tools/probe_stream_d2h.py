#!/usr/bin/env python3
"""round 6: is svc::StreamEncoder's D2H rate the BOX's or the encoder's?  On one box, alternately: tools/_bin/ubench_pcie (a bare 400 MiB
hipMemcpyAsync D2H into pinned memory, and a copy kernel) and tests/dropin/stream_main over the 65-frame 1080p sample clip bench.py's end_to_end
uses (its phases line: d2h_GBps from HIP events around the batch's D2H copies).  Extra arguments are passed to stream_main through the
environment (SVC_STREAM_* switches), one run per value of --env."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scalable_video_codec_amd import configs, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE for an extra stream_main run per repetition")
    args = ap.parse_args()
    cfg = configs.ALL["C3-1080p-3L-dct8-quant"]
    dev = torch.device("cuda")
    src = synth.SynthClip(cfg.width, cfg.height, 66, cfg.seed, device=dev)
    frames = np.stack([src.frame_bgr(t).cpu().numpy() for t in range(65)])
    ubench = os.path.join(ROOT, "tools", "_bin", "ubench_pcie")
    exe = os.path.join(ROOT, "tests", "dropin", "stream_main")
    with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
        raw = os.path.join(d, "clip.raw")
        frames.tofile(raw)
        for rep in range(args.reps):
            r = subprocess.run([ubench, "400"], capture_output=True, text=True, timeout=120)
            for ln in r.stdout.splitlines():
                if "hipMemcpyAsync, one call" in ln or "idle" in ln or "slice per workgroup, 64" in ln or "beside" in ln:
                    print("  ubench:", " ".join(ln.split()), flush=True)
            for env in [""] + args.env:
                e = dict(os.environ)
                if env:
                    k, v = env.split("=", 1)
                    e[k] = v
                r = subprocess.run([exe, raw, str(cfg.width), str(cfg.height), "65", str(cfg.levels), str(cfg.dct_block), "0", "16", str(cfg.seed), "-"],
                                   capture_output=True, text=True, timeout=300, env=e)
                if r.returncode != 0:
                    print("  stream_main failed:", (r.stderr or r.stdout)[-300:], flush=True)
                    continue
                fps = float(r.stdout.split("encoded frames,")[1].split("frames/s")[0])
                ph = [ln for ln in r.stdout.splitlines() if ln.startswith("phases ")]
                p = json.loads(ph[-1][len("phases "):])
                for ln in r.stdout.splitlines():
                    if ln.startswith("d2h_GBps_by_pass"):
                        print("   ", ln, flush=True)
                ls = [ln for ln in r.stdout.splitlines() if ln.startswith("long_stream ")]
                if ls:
                    q = json.loads(ls[-1][len("long_stream "):])
                    print(f"  stream_main [{env or 'as built'}] as ONE stream: {q['frames_per_s']:.0f} frames/s  wall per batch {q['host_ms_per_batch']['wall']} ms  "
                          f"d2h {q['device_ms_per_batch']['d2h']} ms = {q['d2h_GBps']} GB/s", flush=True)
                print(f"  stream_main [{env or 'as built'}]: {fps:.0f} frames/s  d2h {p['d2h_GBps']} GB/s  h2d {p['h2d_GBps']} GB/s  device ms {p['device_ms_per_batch']}  "
                      f"host ms {p['host_ms_per_batch']}", flush=True)


if __name__ == "__main__":
    main()
