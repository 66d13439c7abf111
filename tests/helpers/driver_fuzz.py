#!/usr/bin/env python3
"""Random call sequences against svc::ClipEncoder (through its C handle API), every result checked against the serial two-pass encoder's.

The pipelined driver is the most stateful piece of the host side: a software pipeline over micro-steps, buffer sets rotating by step,
events reused by slot, coefficient sets that appear on the first speculation, a policy fed by measurements that arrive when they arrive.
Its tests walk chosen sequences; this walks RANDOM ones: per encoder a random configuration (output form, chunk plan, pipeline depth, the
switches of round 6) and a random sequence of load(a|b|c) / step x k / flush / sync / reset_policy / read -- after every read the outputs
must be the serial two-pass encoder's for the clip that is resident, bit for bit.  A chunk served from a stale set, an event waited on too
early, a coefficient set rewritten before its redo, a policy decision taken on another clip's share would each surface as the wrong clip's
bytes.  usage: driver_fuzz.py [--encoders N] [--ops M] [--seed S]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scalable_video_codec_amd import clip as clipmod  # noqa: E402
from scalable_video_codec_amd import configs, synth  # noqa: E402

KEYS = ("mv", "min_mad", "global_motion", "rmse", "inlier_mask", "inlier_count", "block_types")


def frames_of(cfg, n, dev):
    src = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
    pw, ph = cfg.padded
    return torch.stack([synth.pad_frame(src.frame_bgr(t), pw, ph) for t in range(n)]).contiguous()


def reference(cfg, n, frames, wire, ransac, dev):
    s = clipmod.Clip(cfg, n, schedule=clipmod.SERIAL, wire=wire, tuning=clipmod.TUNE_TWO_BGR_PASSES, ransac=ransac)
    s.load_frames(frames)
    s.step()
    s.sync()
    # (the pyramids also on the device: a shard's halo is handed over from there, as RCCL would deliver it)
    out = (s.outputs(), s.read("records" if wire else "coeffs"), s.read("pyramids"), s.read("pyramids", device=dev))
    s.close()
    return out


def same(enc, want, wire):
    """None, or the name of the first output of the encoder's shard that is not the unsharded serial encoder's rows."""
    i = enc.info
    g0, p = i.first_encoded - 1, i.pairs
    out = enc.outputs()  # syncs
    for k in KEYS:
        a, b = out[k], want[0][k][g0:g0 + p]
        if a.dtype.is_floating_point:
            if a.numpy().tobytes() != b.numpy().tobytes():
                return k
        elif not torch.equal(a, b):
            return k
    big = want[1]
    per = big.numel() // want[0]["mv"].shape[0]  # elements per encoded frame (planes or records)
    if not torch.equal(enc.read("records" if wire else "coeffs"), big[g0 * per:(g0 + p) * per]):
        return "records" if wire else "coeffs"
    st = i.pyramid_stride
    lo = 0 if i.needs_halo else 1  # slot 0 is the halo (the previous rank's last frame) where there is one
    if not torch.equal(enc.read("pyramids")[lo * st:], want[2][(i.first_frame + lo) * st:(i.first_frame + i.frames + 1) * st]):
        return "pyramids"
    return None


_hip = None


def hip_copy_async(dst, src, nbytes, stream):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so.7")  # the HIP runtime already in the process
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert _hip.hipMemcpyAsync(dst, src, nbytes, 3, stream) == 0


def one_encoder(rng, dev, shapes, clips, refs, ops, log, max_world=4):
    si = int(rng.integers(len(shapes)))
    cfg, n = shapes[si]
    wire = bool(rng.integers(3) == 0)
    tuning = 0
    form = "wire" if wire else ["policy", "two_passes", "always_speculate"][int(rng.integers(3))]
    if form == "two_passes":
        tuning |= clipmod.TUNE_TWO_BGR_PASSES
    if form == "always_speculate":
        tuning |= clipmod.TUNE_ALWAYS_SPECULATE
    # (small shards can speculate by the POLICY only with the idle rule's size waiver: SVC_CLIP_TUNE_IDLE_RULE_ANY_SIZE)
    for bit, p in ((clipmod.TUNE_IDLE_RULE_ANY_SIZE, 0.85 if form == "policy" else 0.5), (clipmod.TUNE_MIXED_STEPS, 0.4), (clipmod.TUNE_SEARCH_AFTER_TRANSFORM, 0.3),
                   (clipmod.KEEP_FOREGROUND_PRIOR, 0.3), (clipmod.TUNE_WHOLE_SHARD_STEPS, 0.15), (clipmod.TUNE_INLINE_RMSE, 0.2),
                   (clipmod.TUNE_NARROW_ATTEMPTS, 0.2), (clipmod.TUNE_FORK_BEHIND_FRONT, 0.3), (clipmod.TUNE_RANDOM_POLICY, 0.35 if form == "policy" else 0.0)):
        if rng.random() < p:
            tuning |= bit
    chunk_pairs = int(rng.integers(1, n)) if rng.random() < 0.4 else 0
    lat_depth = int(rng.integers(0, 4))
    ransac = dict(inlier_thresh=1.5)
    # one rank of a multi-rank run now and then (its own iteration order: the motion search one micro-step behind its pyramids, the halo on
    # the communication stream); the halo comes from the unsharded reference's pyramids through the transport hook
    world = int(rng.integers(2, 5)) if rng.random() < (0.0 if max_world < 2 else 0.3) else 1
    world = min(world, max_world, n)
    rank = int(rng.integers(world))
    desc = f"{cfg.name} n={n} rank {rank}/{world} {form} tuning={tuning} chunk_pairs={chunk_pairs} lat_depth={lat_depth}"
    enc = clipmod.Clip(cfg, n, rank=rank, world=world, schedule=clipmod.PIPELINED, wire=wire, lat_depth=lat_depth, tuning=tuning,
                       chunk_pairs=chunk_pairs, ransac=ransac)
    first, cnt = enc.info.first_frame, enc.info.frames
    resident, stepped, trace = None, False, []  # resident: the clip the LAST step encoded (or the one loaded, before any step)
    loaded = None                                # the clip in the encoder's own buffer
    bad = None
    halo = {"src": 0}
    shard = {}  # clip -> its frames of this shard, contiguous on the device (step_frames reads them where they are)

    def frames_of_shard(r):
        if r not in shard:
            shard[r] = clips[si][r][first:first + cnt].contiguous()
        return shard[r]

    def halo_for(r):
        if world > 1:  # the pyramid of clip frame first - 1 (slot = frame + 1) of the clip about to be stepped
            halo["src"] = ref_of(r)[3].data_ptr() + first * enc.info.pyramid_stride

    def ref_of(r):
        key = (si, r, wire)
        if key not in refs:
            refs[key] = reference(cfg, n, clips[si][r], wire, ransac, dev)
        return refs[key]
    if world > 1:
        enc.set_halo_transport(lambda send, recv, nbytes, stream: hip_copy_async(recv, halo["src"], nbytes, stream) if recv else None)
    try:
        for _ in range(ops):
            r = rng.random()
            if loaded is None or r < 0.15:
                loaded = resident = int(rng.integers(len(clips[si])))
                enc.load_frames(frames_of_shard(loaded))
                stepped = False
                trace.append(f"load{loaded}")
            elif r < 0.50:
                k = int(rng.integers(1, 6))
                halo_for(loaded)
                for _ in range(k):
                    enc.step(timed=bool(rng.integers(2)))
                resident, stepped = loaded, True
                trace.append(f"step{k}")
            elif r < 0.62:  # a stream of clips: steps over frames where they are, no load, no drain
                k = int(rng.integers(1, 5))
                for _ in range(k):
                    resident = int(rng.integers(len(clips[si])))
                    halo_for(resident)
                    last_ext = enc.step_frames(frames_of_shard(resident), timed=bool(rng.integers(2)))
                    trace.append(f"ext{resident}")
                    if rng.random() < 0.2:
                        enc.wait_step(int(rng.integers(0, last_ext + 1)))
                        trace.append("wait")
                stepped = True
            elif r < 0.70:
                enc.flush()
                trace.append("flush")
            elif r < 0.78:
                enc.sync()
                trace.append("sync")
            elif r < 0.84:
                enc.reset_policy()
                trace.append("reset")
            elif r < 0.88:
                enc.reset_timers()
                trace.append("timers")
            elif stepped:
                trace.append("read")
                bad = same(enc, ref_of(resident), wire)
                if bad:
                    break
        if not bad and stepped:
            bad = same(enc, ref_of(resident), wire)
    finally:
        pol = enc.policy_info()
        enc.close()
    log(f"{'FAIL ' + bad if bad else 'ok  '} {desc} | decided {pol['chunks_decided']} speculated {pol['chunks_speculated']} | {' '.join(trace[-24:])}")
    return bad is None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoders", type=int, default=40)
    ap.add_argument("--ops", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-world", type=int, default=4, help="largest world size a shard's encoder is drawn from (1: unsharded only)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda")
    clipmod.load()
    # two shapes (one whose padded width differs from the frame's), three clips each: different content AND one with the same content as
    # another but a different seed, so that stale bytes cannot pass for fresh ones
    shapes = [(configs.CodecConfig("fz-640x360-3L-dct8", 41, 640, 360, 9, levels=3, dct_block=8), 9),
              (configs.CodecConfig("fz-416x234-2L-dct16", 42, 416, 234, 6, levels=2, dct_block=16), 6)]
    clips = []
    for cfg, n in shapes:
        cs = []
        for seed in (cfg.seed, cfg.seed + 100, cfg.seed + 200):
            c2 = configs.CodecConfig(cfg.name, seed, cfg.width, cfg.height, n, levels=cfg.levels, dct_block=cfg.dct_block)
            cs.append(frames_of(c2, n, dev))
        # ... and a STILL clip (one frame repeated: no foreground at all), so that the policy's answer flips between loads: it speculates
        # on this one, not on the others (whose share is far above 2 %), and with a kept prior it acts on the WRONG clip's share for a step
        cs.append(cs[0][:1].repeat(n, 1, 1, 1).contiguous())
        clips.append(cs)
    refs = {}
    t0 = time.perf_counter()
    good = 0
    for i in range(args.encoders):
        good += one_encoder(rng, dev, shapes, clips, refs, args.ops, lambda s: print(s, flush=True), args.max_world)
    print(f"{good} of {args.encoders} random call sequences gave the serial two-pass encoder's bytes ({time.perf_counter() - t0:.0f} s)", flush=True)
    return 0 if good == args.encoders else 1


if __name__ == "__main__":
    sys.exit(main())
