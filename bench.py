#!/usr/bin/env python3
"""bench.py -- encoded frames/s of the MI355X encode hot path on the BASELINE.json
headline workload: 1080p (padded 1920x1088), 300-frame synthetic clip, 16x16 MV
blocks, 3-level HBMA + RANSAC + 8x8 DCT + quant (fg 1 / bg 640).

One "step" = one pass of the hot path over this rank's shard of the clip, resident in HBM:
  luma + pyramid (all frames) -> [N>1: halo shift of the previous rank's last pyramid, RCCL
  send/recv on its own stream] -> fused HBMA (all frame pairs) -> RANSAC (per frame) ->
  segmentation (mask, close/open, k-means, connected components -> region ids) -> fused DCT +
  quant (per encoded frame).  Default schedule: consecutive steps are software-pipelined; every
  step's work is enqueued and finished inside the timed region.
The driver of a step is C++ (svc::ClipEncoder, include/svc/clip_encoder.hpp) behind the C handle
API of include/svc_clip.h; this file loads frames, calls step() and reports.

N GPUs (BASELINE configs 4/5): the SAME clip is cut into N consecutive chunks, one per rank
("scaling": "strong": 300 frames -> 37/38 per GPU at N = 8); the weak-scaling figure (300 frames per
GPU) is measured after it and reported under "weak".  `value` = encoded frames of all ranks /
max-over-ranks time, BGR frames already resident in HBM when the timed region starts (PCIe
excluded).

Contract: python bench.py --gpus N --steps K --warmup W ; for N > 1 the driver launches it under
torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from scalable_video_codec_amd import clip as clipmod  # noqa: E402
from scalable_video_codec_amd import configs, native, pipeline, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(cfg: configs.CodecConfig, frames_bgr, budget_s: float = 7.0):
    """Times the CPU path on THIS host on the first frames of the same clip, one core and all cores, in TWO rows:
    the configuration's own level count, and the reference's default build (its SSE2 entry exists for 4 levels / 16x16 only,
    libs/motion.hpp:143-147).  Motion search: the UNMODIFIED reference (oracle/_ref) when it was built, else the C
    restatement.  RANSAC / segmentation / quant: the restatement.  Transform: oracle/svc_cpu_dct.c, an f32 separable DCT
    with an AVX2 + FMA path (cv::dct cannot be built offline, so the reference's own transform cannot be timed;
    `dct_is_reference` says so).  Only this leg and tests may touch oracle/."""
    import concurrent.futures as cf
    import numpy as np
    from oracle import binding
    orc = binding.Oracle()
    ref = binding.Reference() if binding.Reference.available() else None
    impl = ref if ref is not None else orc
    n = min(len(frames_bgr), 128)
    host = [f.cpu() for f in frames_bgr[:n]]
    host_np = [f.numpy() for f in host]
    k = orc.ransac_iter_count(**binding.DEFAULT_RANSAC)
    mfw, mfh = cfg.mv_field
    cores = max(1, min(len(os.sched_getaffinity(0)), 32))
    fast_dct = cfg.dct_block in (8, 16)

    def run_row(levels: int, sse2: bool, label: str):
        pyrs = {}

        def pyr(i):  # built lazily; the pre-step is not part of the timed CPU work (the GPU side's input is the same frames)
            if i not in pyrs:
                pyrs[i] = [p.numpy() for p in synth.build_pyramid(synth.bgr_to_y(host[i]), levels)]
            return pyrs[i]
        search = (lambda t, a: impl.hbma16_sse2(t, a, cfg.search_range)) if sse2 else \
                 (lambda t, a: impl.hbma(t, a, cfg.search_range, cfg.mv_block, cfg.mv_block))
        local = __import__("threading").local()

        def encode_frame(i, ta, tb_):  # everything the hot path does for encoded frame i, on the CPU
            t0 = time.perf_counter()
            mv, _ = search(ta, tb_)
            t1 = time.perf_counter()
            samples = (np.arange(k, dtype=np.uint32) * 2654435761 % len(mv)).astype(np.uint32)
            _, _, inl = orc.ransac(mv, samples, **binding.DEFAULT_RANSAC)
            inl_mask = np.zeros(len(mv), np.uint8)
            inl_mask[inl] = 1
            types = orc.segment(inl_mask, mv, mfw, mfh, cfg.mv_block, cfg.mv_block, seed=i)
            t2 = time.perf_counter()
            if cfg.dct_block:
                if fast_dct:
                    if not hasattr(local, "planes"):
                        local.planes = np.empty((3,) + host_np[i].shape[:2], np.float32)
                    planes = orc.cpu_dct_frame_f32(host_np[i], cfg.dct_block, local.planes)
                else:
                    planes = orc.dct_frame_f32(host_np[i], cfg.dct_block, cfg.dct_block)
                t3 = time.perf_counter()
                orc.cpu_quant_frame_f32(np.ascontiguousarray(planes), cfg.mv_block, cfg.mv_block, types, cfg.fg_step, cfg.bg_step)
            else:
                t3 = t2
            return t1 - t0, t2 - t1, t3 - t2, time.perf_counter() - t3

        search(pyr(0), pyr(1))  # warm-up: page in, let the core clock up
        tot = [0.0, 0.0, 0.0, 0.0]
        done, busy = 0, 0.0
        for i in range(1, n):
            ta, tb_ = pyr(i - 1), pyr(i)
            pyrs.pop(i - 2, None)
            dts = encode_frame(i, ta, tb_)
            tot = [a + b for a, b in zip(tot, dts)]
            busy += sum(dts)
            done += 1
            if busy > budget_s:
                break
        # the same work frame-parallel on the host's cores (the reference itself encodes on one thread, apps/encoder.cpp:228;
        # this is what a frame-parallel CPU deployment of it would get): the C entry points release the GIL
        all_cores = None
        if cores > 1 and done:
            m = min(n - 1, 3 * cores)
            pp = [pyr(i) for i in range(m + 1)]  # pre-step, untimed as above
            with cf.ThreadPoolExecutor(max_workers=cores) as ex:
                list(ex.map(lambda i: encode_frame(i, pp[i - 1], pp[i]), range(1, min(m, cores) + 1)))  # warm the pool
                t0 = time.perf_counter()
                list(ex.map(lambda i: encode_frame(i, pp[i - 1], pp[i]), range(1, m + 1)))
                wall = time.perf_counter() - t0
            all_cores = {"value": m / wall, "unit": "frames/s", "cores": cores,
                         "sample": f"{m} encoded frames of the same clip, one frame per task on {cores} threads"}
        total = sum(tot)
        return {
            "label": label, "levels": levels,
            "value": done / total if total > 0 else None, "unit": "frames/s", "cores": 1, "frames_timed": done,
            "hbma_ms_per_frame": tot[0] / done * 1e3 if done else None,
            "ransac_segment_ms_per_frame": tot[1] / done * 1e3 if done else None,
            "dct_ms_per_frame": tot[2] / done * 1e3 if done else None,
            "quant_ms_per_frame": tot[3] / done * 1e3 if done else None,
            "all_cores": all_cores,
        }

    who = "unmodified reference" if ref is not None else "C restatement of"
    own_sse2 = cfg.levels == 4 and cfg.mv_block == 16
    rows = {}
    if own_sse2:
        rows["config"] = run_row(4, True, f"{who} EstimateMotionHierarchical16x16Sse2 (the reference's default build) + the CPU transform")
        rows["generic"] = run_row(4, False, f"{who} EstimateMotionHierarchical, 4 levels, generic path (SVC_MOTION_SSE2 off) + the CPU transform")
    else:
        rows["config"] = run_row(cfg.levels, False, f"{who} EstimateMotionHierarchical, {cfg.levels} levels (generic path: the reference's SSE2 path "
                                 "exists for 4 levels only) + the CPU transform")
        pw, ph = cfg.padded
        if cfg.mv_block == 16 and pw % 8 == 0 and ph % 8 == 0:
            rows["sse2_4level"] = run_row(4, True, f"{who} EstimateMotionHierarchical16x16Sse2 -- the reference's default build, 4 levels, "
                                          "NOT this configuration's level count -- + the CPU transform")
    main = rows["config"]
    return {
        "value": main["value"],
        "unit": "frames/s",
        "cores": 1,
        "kind": "reference" if ref is not None else "port",
        "sample": (f"first {main['frames_timed']} encoded frames of the same clip, 1 thread: motion search = {main['label']}; "
                   "RANSAC / segmentation = C restatement; transform + quant = oracle/svc_cpu_dct.c (f32 separable, "
                   f"{orc.cpu_dct_isa() if fast_dct else 'f64 from the definition: block size outside 8 / 16'}), not cv::dct"),
        "rows": rows,
        "hbma_ms_per_frame": main["hbma_ms_per_frame"],
        "ransac_dct_quant_ms_per_frame": (main["ransac_segment_ms_per_frame"] or 0) + (main["dct_ms_per_frame"] or 0) + (main["quant_ms_per_frame"] or 0),
        "dct_leg": ("own f32 separable DCT-II with an AVX2 + FMA path (oracle/svc_cpu_dct.c; checked against the f64 oracle at 1e-4 max(1, |ref|)), "
                    "not cv::dct: OpenCV is not installed, the reference's transform cannot be timed"),
        "dct_isa": orc.cpu_dct_isa(),
        "dct_is_reference": False,
        "reference_sse2_4level_hbma_ms_per_frame": (rows.get("sse2_4level") or rows["config"])["hbma_ms_per_frame"] if (own_sse2 or "sse2_4level" in rows) else None,
        "all_cores": main["all_cores"],
    }


def end_to_end_rates(cfg: configs.CodecConfig, frames_padded) -> dict:
    """PCIe-INCLUSIVE rates of the three host-facing ways into the path ("plus a separate end-to-end
    number including PCIe"), measured on this box OUTSIDE the timed region, frames starting in host memory and every
    output landing back in host memory:
      reference_signatures_fps   one frame per call through the reference's own signatures (svc_hip_hbma_host + _ransac_host +
                                 _dct_quant_host, what libsvc_motion.so's wrappers call), synchronous, one host thread;
      stream_encoder_fps         svc::StreamEncoder (C++, tests/dropin/stream_main): batches, H2D / kernels / D2H on three streams;
      reference_application_fps  the reference's UNCHANGED apps/encoder.cpp + libs/encoder.cpp on compat/opencv2
                                 (tests/dropin/ref_encoder_*), two clip lengths so that process start-up cancels;
      reference_application_batched_encoder_fps  the same unchanged apps/encoder.cpp with this repo's class Encoder
                                 (the reference's encoder.hpp on svc::StreamEncoder, tests/dropin/ref_app_svc_encoder*).
    `value` of the line never includes any of this."""
    import subprocess
    import tempfile
    import numpy as np
    out = {"unit": "frames/s", "pcie_inclusive": True, "config": cfg.name,
           "bound": "PCIe D2H of the f32 coefficients (25 MB per 1080p frame: 63 GB/s caps a host-fed pipeline near 2 000 frames/s) for the batched "
                    "driver; host-side copies and per-call round trips for the two per-frame forms"}
    pw, ph = cfg.padded
    host = [f.cpu().numpy() for f in frames_padded[:66]]
    # (a) the reference's one-frame-per-call signatures
    try:
        pyr = [[p.numpy() for p in synth.build_pyramid(synth.bgr_to_y(torch.from_numpy(f)), cfg.levels)] for f in host[:2]]
        types = np.zeros(cfg.blocks, np.uint32)
        smp = (np.arange(native.ransac_iter_count(), dtype=np.uint32) * 977) % cfg.blocks

        def one_frame():
            mv, _ = native.hbma_host(pyr[0], pyr[1], cfg.search_range, cfg.mv_block, cfg.mv_block)
            native.ransac_host(mv, smp)
            if cfg.dct_block:
                native.dct_quant_host(host[1], cfg.dct_block, types, cfg.mv_block, cfg.fg_step, cfg.bg_step)
        one_frame(); one_frame()
        t0 = time.perf_counter()
        for _ in range(12):
            one_frame()
        out["reference_signatures_fps"] = 12 / (time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001
        out["reference_signatures_fps"] = None
        out["reference_signatures_note"] = f"not measured: {e}"
    bin_dir = os.path.join(ROOT, "tests", "dropin")
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=tmp_root) as d:
        src = np.stack([f[:cfg.height, :cfg.width] for f in host])  # the unpadded source frames
        # (b) the batched C++ driver.  Round 6: THIS process has just used the GPU (the per-call leg above), and for most of a second after
        # a process's last GPU work its idle copy queues still share the SDMA engines with the child's: the child's D2H then runs at exactly
        # 29.1 GB/s and jumps to 54.9 in mid-run (`d2h_GBps_by_pass`; profiles/r06_probe_e2e_twice.txt) -- which is what moved round 5's
        # figure between 1 087 and 1 956 frames/s.  A harness artefact, not the encoder's or the box's: the parent goes quiet first.
        exe = os.path.join(bin_dir, "stream_main")
        try:
            torch.cuda.synchronize()
            time.sleep(float(os.environ.get("SVC_BENCH_QUIET_SECONDS", "2.0")))
            raw = os.path.join(d, "clip.raw")
            n = 65  # fixed length whatever the run's --frames (the sample frames, repeated, as for the application rows below)
            with open(raw, "wb") as f:
                for lo in range(0, n, len(src)):
                    src[:min(len(src), n - lo)].tofile(f)
            r = subprocess.run([exe, raw, str(cfg.width), str(cfg.height), str(n), str(cfg.levels), str(cfg.dct_block), "0", "16",
                                str(cfg.seed), "-"], capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                raise RuntimeError((r.stderr or r.stdout).strip()[-300:])
            out["stream_encoder_fps"] = float(r.stdout.split("encoded frames,")[1].split("frames/s")[0])
            ph = [ln for ln in r.stdout.splitlines() if ln.startswith("phases ")]
            # the encoder's own clocks (svc::EncodeStats): per batch of 16, host phases of the calling thread and HIP-event time per stream --
            # which of H2D / kernels / D2H / staging bounds this box's figure, and how many cores the copy crew had
            out["stream_encoder_phases"] = json.loads(ph[-1][len("phases "):]) if ph else None
            # round 6: the same frames as ONE stream of the same length (a single Encode call over a cycling source): the pipeline fills and
            # drains once instead of once per 65-frame clip, so wall per batch -> the D2H time of a batch
            bp = [ln for ln in r.stdout.splitlines() if ln.startswith("d2h_GBps_by_pass ")]
            if bp and out["stream_encoder_phases"] is not None:
                out["stream_encoder_phases"]["d2h_GBps_by_pass"] = [float(x) for x in bp[-1].split()[1:]]
            ls = [ln for ln in r.stdout.splitlines() if ln.startswith("long_stream ")]
            out["stream_encoder_long_stream"] = json.loads(ls[-1][len("long_stream "):]) if ls else None
            out["stream_encoder_sample"] = (f"{n - 1} encoded frames per pass ({min(len(src), n)} distinct sample frames"
                                            f"{', repeated' if len(src) < n else ''}), batch 16, passes over the clip repeated for >= 1 s after one warm-up pass")
            os.remove(raw)
        except Exception as e:  # noqa: BLE001
            out["stream_encoder_fps"] = None
            out["stream_encoder_note"] = f"not measured: {e}"
        # (c) the reference's own application.  Both application rows run with SVC_KEEP_LARGE_BLOCKS=1: the opt-in by which a host process
        # asks these libraries to keep frame-sized blocks on the heap (nothing is tuned unasked)
        tuned_env = dict(os.environ, SVC_KEEP_LARGE_BLOCKS="1")
        sse2 = cfg.levels == 4 and cfg.mv_block == 16
        exe = os.path.join(bin_dir, "ref_encoder_sse2" if sse2 else "ref_encoder_generic")
        try:
            if not os.path.exists(exe):
                raise RuntimeError("tests/dropin/ref_encoder_* not built (needs /root/reference at build time)")
            args = ["--verbose", "0", "--transform-block-w", str(cfg.dct_block or 8), "--transform-block-h", str(cfg.dct_block or 8)]
            if not sse2:
                args += ["--pyr-lvl-count", str(cfg.levels), "--mv-block-w", str(cfg.mv_block), "--mv-block-h", str(cfg.mv_block)]
            times = {}
            for n in (5, 69):  # 64 frames apart whatever the run's --frames: the sample frames, repeated
                path = os.path.join(d, f"clip{n}.svcbgr")
                with open(path, "wb") as f:
                    f.write(b"SVCBGR1\0" + np.array([cfg.width, cfg.height, n, 0], np.uint32).tobytes())
                    for lo in range(0, n, len(src)):
                        src[:min(len(src), n - lo)].tofile(f)
                t0 = time.perf_counter()
                with open(os.devnull, "wb") as sink:
                    r = subprocess.run([exe, *args, path], stdout=sink, stderr=subprocess.PIPE, timeout=600, env=tuned_env)
                times[n] = time.perf_counter() - t0
                os.remove(path)
                if r.returncode != 0:
                    raise RuntimeError(r.stderr.decode()[-300:])
            (n1, t1), (n2, t2) = sorted(times.items())
            if t2 - t1 < 0.1:  # a difference of two process lifetimes: below this it measures the scheduler, not the encoder
                raise RuntimeError(f"{n2 - n1} more frames took {t2 - t1:.3f} s more: too short to tell")
            out["reference_application_fps"] = (n2 - n1) / (t2 - t1)
            out["reference_application_sample"] = (f"{os.path.basename(exe)} {' '.join(args)}: {n1} and {n2} frame clips, stdout to /dev/null; "
                                                   f"({n2} - {n1}) frames / ({t2:.2f} - {t1:.2f}) s, so process start-up and GPU initialisation cancel; SVC_KEEP_LARGE_BLOCKS=1")
        except Exception as e:  # noqa: BLE001
            out["reference_application_fps"] = None
            out["reference_application_note"] = f"not measured: {e}"
        # (d) the same unchanged application with THIS repo's implementation of the reference's class Encoder (svc::StreamEncoder
        # behind libs/encoder.hpp: csrc/host/encoder_hip.cpp) in place of the reference's libs/encoder.cpp
        exe = os.path.join(bin_dir, "ref_app_svc_encoder" if sse2 else "ref_app_svc_encoder_generic")
        try:
            if not os.path.exists(exe):
                raise RuntimeError("tests/dropin/ref_app_svc_encoder* not built (needs /root/reference at build time)")
            if cfg.dct_block == 0:
                raise RuntimeError("this configuration has no transform")
            n = 193
            path = os.path.join(d, f"clip{n}.svcbgr")
            with open(path, "wb") as f:
                f.write(b"SVCBGR1\0" + np.array([cfg.width, cfg.height, n, 0], np.uint32).tobytes())
                for lo in range(0, n, len(src)):  # the sample frames, repeated
                    src[:min(len(src), n - lo)].tofile(f)
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as sink:  # SVC_ENCODER_REPORT: this repo's Encoder prints its loop's own clock (set-up apart)
                r = subprocess.run([exe, *args, path], stdout=sink, stderr=subprocess.PIPE, timeout=600, env=dict(tuned_env, SVC_ENCODER_REPORT="1"))
            wall = time.perf_counter() - t0
            os.remove(path)
            if r.returncode != 0:
                raise RuntimeError(r.stderr.decode()[-300:])
            line = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("svc Encoder:") and "frames/s" in ln][-1]
            out["reference_application_batched_encoder_fps"] = float(line.split("(")[1].split(" frames/s")[0])
            out["reference_application_batched_encoder_sample"] = (
                f"{os.path.basename(exe)} (the reference's unchanged apps/encoder.cpp + libs/cli.cpp, class Encoder = csrc/host/encoder_hip.cpp on "
                f"svc::StreamEncoder), a {n}-frame clip (the sample frames repeated), stdout to /dev/null, SVC_KEEP_LARGE_BLOCKS=1: '{line}'; whole process {wall:.2f} s")
        except Exception as e:  # noqa: BLE001
            out["reference_application_batched_encoder_fps"] = None
            out["reference_application_batched_encoder_note"] = f"not measured: {e}"
    return out


def predicted_step(cfg: configs.CodecConfig, frames_per_gpu: int, measured_ms: float, world: int) -> dict:
    """What DESIGN.md section 6 PREDICTS for this run, next to what it measured: the repo's scaling table is built from the step
    time ONE MI355X needs for the shard a rank holds (tools/shard_sizes.sh -> profiles/r0N_shard_sizes.json, halo assumed off
    the critical path, eight ranks assumed to behave like one).  The first real N-GPU run grades that model by itself:
    ratio = measured / predicted (1.0 = the model holds; > 1 = the ranks, the halo or the launcher cost what the 1-GPU
    model does not see)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_shard_sizes.json")))
    if not files:
        return {"predicted_ms_per_step": None, "note": "no profiles/r0N_shard_sizes.json in this checkout"}
    with open(files[-1]) as f:
        rows = [r for r in json.load(f)["rows"] if r["workload"].startswith(cfg.name)]
    same = [r["ms_per_step"] for r in rows if r["frames_per_gpu"] == frames_per_gpu]
    whole = [r["ms_per_step"] for r in rows if r["frames_per_gpu"] == cfg.frames]
    if not same:
        return {"predicted_ms_per_step": None, "source": os.path.basename(files[-1]),
                "note": f"no 1-GPU measurement of a {frames_per_gpu}-frame shard of {cfg.name} in that file"}
    pred = sum(same) / len(same)
    return {"predicted_ms_per_step": pred, "measured_ms_per_step": measured_ms, "ratio_measured_over_predicted": measured_ms / pred,
            "predicted_speedup_vs_1_gpu": (sum(whole) / len(whole)) / pred if whole else None,
            "frames_per_gpu": frames_per_gpu, "n_gpus": world, "source": "profiles/" + os.path.basename(files[-1]),
            "model": "step time of the largest shard on ONE MI355X, pipelined schedule; halo off the critical path"}


def hbm_streaming_rates(device) -> dict:
    """What a plain streaming kernel (svc_hip_probe_stream: dwordx4 per lane, contiguous) reaches on THIS GPU, in GB/s,
    for the read/write mixes of the three HBM-bound kernels.  Measured after the timed region (N = 1 only), with
    HIP events around three launches each over 2 GiB buffers."""
    n = 2 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    a.fill_(1)
    out = {}
    for name, r, w in (("read_only", 3, 0), ("read_only_16B_per_lane", 1, 0), ("write_only", 0, 1), ("copy_1_read_1_write", 1, 1), ("3_read_1_write", 3, 1), ("1_read_4_write", 1, 4)):
        native.probe_stream(a, b, r, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            native.probe_stream(a, b, r, w)
        e1.record()
        torch.cuda.synchronize()
        iters = n // (16 * max(r, w))
        out[name] = iters * 16 * (r + w) / (e0.elapsed_time(e1) / 3 * 1e-3) / 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.fill_(0)
    e0.record()
    for _ in range(3):
        b.fill_(0)
    e1.record()
    torch.cuda.synchronize()
    out["write_only_torch_fill"] = n / (e0.elapsed_time(e1) / 3 * 1e-3) / 1e9
    del a, b
    return out




class _DevMem:
    """A raw device range as a __cuda_array_interface__ object (zero-copy torch view of a C++-owned buffer)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _torch_halo_transport(rank: int, world: int, staged: bool):
    """The neighbour shift through torch.distributed P2P (RCCL under backend "nccl"; `staged` = gloo rehearsal,
    through host memory).  Alternative to the C ABI's own RCCL entry point (SVC_HALO=torch)."""
    def fn(send_ptr, recv_ptr, nbytes, stream_ptr):
        stream = torch.cuda.ExternalStream(stream_ptr)
        with torch.cuda.stream(stream):
            send = torch.as_tensor(_DevMem(send_ptr, nbytes), device="cuda")
            recv = torch.as_tensor(_DevMem(recv_ptr, nbytes), device="cuda")
            if staged:
                stream.synchronize()
                send_h, recv_h = send.cpu(), torch.empty(nbytes, dtype=torch.uint8)
            ops = []
            if rank + 1 < world:
                ops.append(dist.P2POp(dist.isend, send_h if staged else send, rank + 1))
            if rank > 0:
                ops.append(dist.P2POp(dist.irecv, recv_h if staged else recv, rank - 1))
            for w in dist.batch_isend_irecv(ops):
                w.wait()  # nccl: orders the stream behind the transfer, does not block the host
            if staged and rank > 0:
                recv.copy_(recv_h)
                stream.synchronize()
    return fn


def _checksum(x: torch.Tensor) -> int:
    w = (torch.arange(x.numel(), device=x.device, dtype=torch.int64) % 251) + 1
    return int((x.to(torch.int64) * w).sum().item())


def hbma_flags_of(args) -> int:
    return {"auto": native.HBMA_AUTO, "tiled": native.HBMA_FORCE_TILED, "lane": native.HBMA_FORCE_LANE,
            "wave": native.HBMA_FORCE_WAVE_PER_BLOCK}[args.hbma_kernel]


def run_mode(args, cfg, mode: str, rank: int, world: int, dev, backend: str, comm):
    """Builds this rank's shard for `mode` ("strong": cfg.frames cut over the ranks; "weak": cfg.frames per rank),
    runs W warm-up and exactly K timed steps between barriers, returns the measurements of this rank."""
    n_cfg = args.frames or cfg.frames
    clip_frames = n_cfg if mode == "strong" else n_cfg * world
    schedule = clipmod.SERIAL if args.schedule == "serial" else clipmod.PIPELINED
    hbma_flags = hbma_flags_of(args)
    tuning = (clipmod.TUNE_STANDALONE_SHAPES if args.standalone_shapes else 0) | (clipmod.TUNE_SEGMENT_FORK if args.segment_fork else 0) | \
             (clipmod.TUNE_NARROW_ATTEMPTS if args.narrow_attempts else 0) | (clipmod.TUNE_INLINE_RMSE if args.inline_rmse else 0) | \
             (clipmod.TUNE_TWO_BGR_PASSES if args.two_bgr_passes else 0) | (clipmod.TUNE_ALWAYS_SPECULATE if args.always_speculate else 0) | \
             (clipmod.TUNE_WHOLE_SHARD_STEPS if args.whole_shard_steps else 0) | (clipmod.TUNE_SEARCH_AFTER_TRANSFORM if args.search_after_transform else 0) | (clipmod.TUNE_MIXED_STEPS if args.mixed_steps else 0) | (clipmod.TUNE_FORK_BEHIND_FRONT if args.fork_behind_front else 0)
    enc = clipmod.Clip(cfg, clip_frames, rank=rank, world=world, schedule=schedule,
                       segmentation=not args.no_segmentation, wire=args.wire, hbma_flags=hbma_flags,
                       lat_depth=args.lat_depth, tuning=tuning, chunk_pairs=args.chunk_pairs)
    info = enc.info
    src = synth.SynthClip(cfg.width, cfg.height, clip_frames, cfg.seed, device=dev)
    pw, ph = cfg.padded
    want_first = world == 1 and mode == "strong" and args.first_encode_reps > 0
    held = []  # N = 1: the padded clip stays on the device (1.9 GB at C3) so that first_encode can load it again
    for j in range(info.frames):
        f = synth.pad_frame(src.frame_bgr(info.first_frame + j), pw, ph).unsqueeze(0).contiguous()
        enc.load_frames(f, j)
        if want_first:
            held.append(f)
    sample_frames = [synth.pad_frame(src.frame_bgr(t), pw, ph) for t in range(min(info.frames, 129))] \
        if (rank == 0 and mode == "strong" and world == 1 and not (args.no_cpu_baseline and args.no_end_to_end)) else None
    del src
    halo = None
    halo_check = None
    if world > 1:
        if comm is not None:
            enc.set_comm(comm)
            halo = "svc_hip_halo_shift (RCCL send/recv, C ABI)"
        else:
            enc.set_halo_transport(_torch_halo_transport(rank, world, staged=backend != "nccl"))
            halo = f"torch.distributed P2P ({'RCCL' if backend == 'nccl' else backend + ', staged through host memory'})"
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1 if world > 1 else 0)):
        enc.step()
    enc.sync()
    if world > 1:
        # the halo that arrived must be the predecessor's last pyramid: checked once, outside the timed region
        pyr = enc.read("pyramids", device=dev)
        stride = info.pyramid_stride
        mine = torch.tensor([_checksum(pyr[info.frames * stride:(info.frames + 1) * stride])], dtype=torch.int64,
                            device=dev if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        got = _checksum(pyr[:stride]) if rank > 0 else 0
        bad = torch.tensor([1 if (rank > 0 and got != int(every[rank - 1].item())) else 0], dtype=torch.int64, device=mine.device)
        verdicts = [torch.zeros_like(bad) for _ in range(world)]
        dist.all_gather(verdicts, bad)  # collective: every rank learns every rank's verdict and they leave together
        failed = [q for q in range(world) if int(verdicts[q].item())]
        if failed:
            if rank in failed:
                print(f"bench.py: rank {rank}: the halo received (checksum {got}) is not rank {rank - 1}'s last pyramid "
                      f"(checksum {int(every[rank - 1].item())})", file=sys.stderr, flush=True)
            raise SystemExit(f"halo self-check failed on rank(s) {failed} (transport: {halo})")
        halo_check = {"verdict": "ok", "ranks_checked": world - 1,
                      "what": "position-weighted checksum of the pyramid in halo slot 0 == the predecessor's last pyramid, all-gathered"}
        del pyr
    barrier()
    enc.reset_timers()
    pol0 = enc.policy_info()
    barrier()
    t0 = time.perf_counter()
    # HIP events around every stage cost 20-45 us per step (tools/diag_step_overhead.py): they are recorded on every
    # `stride`-th step of the timed region, which is what the per-kernel averages below are taken over
    stride = 1 if args.steps < 8 else args.time_every
    for k in range(args.steps):
        enc.step(timed=k % stride == 0)
    enc.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    pol1 = enc.policy_info()
    chunks = enc.info.chunks_per_step
    policy = {"chunk_launches": args.steps * chunks, "had_the_choice": pol1["chunks_decided"] - pol0["chunks_decided"],
              "speculated": pol1["chunks_speculated"] - pol0["chunks_speculated"], "foreground_share_known": pol1["foreground_share"]}

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    encoded = torch.tensor([float(info.pairs)], dtype=torch.float64, device=red_dev)
    st = enc.stage_times_ms()
    sp = enc.stage_pairs()
    steps_timed = {k: (sp[k] / info.pairs if info.pairs and sp.get(k) else float(v[1])) for k, v in st.items()}  # halo: once per step
    per_rank = None
    if world > 1:
        # every rank's own clock and halo time: a straggler or a slow link shows up by rank in the line
        halo_ms = st["halo_exchange"][0] / st["halo_exchange"][1] if "halo_exchange" in st else -1.0
        mine_row = torch.tensor([elapsed / args.steps * 1e3, halo_ms, float(info.frames), float(info.pairs), float(policy["had_the_choice"]),
                                 float(policy["speculated"])], dtype=torch.float64, device=red_dev)
        rows = [torch.zeros_like(mine_row) for _ in range(world)]
        dist.all_gather(rows, mine_row)
        per_rank = [[float(v) for v in r_.tolist()] for r_ in rows]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(encoded, op=dist.ReduceOp.SUM)
    timed_steps = len(range(0, args.steps, stride))
    res = {
        "elapsed": float(t.item()), "encoded_per_step": float(encoded.item()), "info": info, "halo": halo,
        # every stage is one launch sequence per step: its per-step time is the average over the launches that were
        # timed (in the pipelined schedule a timed call covers stages of four different steps, and the first call
        # after a drain times fewer of them, so the counts differ from stage to stage)
        # a stage is launched once per chunk of a step (one chunk by default; a step that finds the pipeline empty may run in two): the driver
        # counts the frame pairs its timed launches covered, so the per-step time is exact whatever the chunking: total x pairs per step / pairs
        "stage_ms_per_step": {k: v[0] / steps_timed[k] for k, v in st.items()},
        "launches_per_step": {k: v[1] / steps_timed[k] for k, v in st.items()},
        "launches_timed": {k: v[1] for k, v in st.items()},
        "chunks": chunks, "policy": policy, "output_sets": enc.output_sets(),
        "timed_steps": timed_steps,
        "sample_frames": sample_frames, "clip_frames": clip_frames,
        "per_rank": per_rank, "halo_check": halo_check,
    }
    if args.sustain_seconds > 0 and world == 1 and mode == "strong":
        # untimed for `value`: back-to-back steps for a few seconds, so that a sampler outside this process sees the
        # GPU busy and the line carries a figure that is not a 50 ms burst
        n_sus, t_sus = 0, time.perf_counter()
        while time.perf_counter() - t_sus < args.sustain_seconds:
            for _ in range(50):
                enc.step()
            enc.sync()
            n_sus += 50
        dt_sus = time.perf_counter() - t_sus
        res["sustained"] = {"ms_per_step": dt_sus / n_sus * 1e3, "steps": n_sus, "seconds": dt_sus}
    if want_first and info.pairs:
        res["first_encode"] = first_encode(args, enc, torch.cat(held), info, chunks)
        del held
    if info.pairs:
        # what the speculative transform had to redo: the share of MV blocks whose region id is not 0 (outside the timed region)
        res["foreground_share"] = float((enc.read("block_types", device=dev) != 0).float().mean().item())
    enc.close()
    torch.cuda.empty_cache()
    return res


def first_encode(args, enc, clip_dev, info, chunks: int) -> dict:
    """What a clip that is encoded ONCE costs, next to `value` (K steps over a resident clip, back to back, the speculation policy taught by
    the warm-up steps -- the steady state of a stream of such clips).  Outside the timed region, N = 1:
      once_through   R x { load_frames of the whole clip (outside the clock; it voids what the policy knew, the driver's default),
                           ONE step(), sync() } -- the pipeline fills and drains inside the clock, nothing is known about the clip;
      with_prior     R x { ONE step(), sync() } on the resident clip with the policy's last measurement kept: the next piece of a stream
                           (what SVC_CLIP_KEEP_FOREGROUND_PRIOR gives a caller across load_frames);
      policy_voided_steps  reset_policy(), then K back-to-back steps: the steady state while the policy learns again.
    The reference encodes a clip once, frame by frame (libs/encoder.cpp:453-664)."""
    import statistics
    reps = args.first_encode_reps
    out = {"unit": "frames/s", "reps": reps, "encoded_frames": info.pairs, "chunks_per_step": chunks}

    def timed_once(reload: bool):
        ms, spec, had = [], 0, 0
        for _ in range(reps):
            if reload:
                enc.load_frames(clip_dev)
            torch.cuda.synchronize()
            p0 = enc.policy_info()
            t0 = time.perf_counter()
            enc.step()
            enc.sync()
            ms.append((time.perf_counter() - t0) * 1e3)
            p1 = enc.policy_info()
            spec += p1["chunks_speculated"] - p0["chunks_speculated"]
            had += p1["chunks_decided"] - p0["chunks_decided"]
        med = statistics.median(ms)
        # chunk launches: what the driver counted where the configuration can read the clip once (a step into an empty pipeline runs in two
        # chunks whatever chunks_per_step says: the idle-pipeline rule), else the configured plan
        return {"ms_median": med, "ms_min": min(ms), "ms_max": max(ms), "value": info.pairs / (med * 1e-3),
                "chunk_launches_speculated": spec, "chunk_launches": had or reps * chunks}
    enc.sync()
    out["with_prior"] = timed_once(False)
    out["once_through"] = timed_once(True)
    # A STREAM of clips, each encoded ONCE: K different clips in device memory (made from the resident one: mirrored, upside down, played
    # backwards, negated, channels permuted, and combinations -- up to 24 of them, 1.9 GB each at C3; beyond that they rotate), stepped where
    # they are (svc_clip_step_frames: no copy into the resident buffer and no drain of the pipeline between clips -- load_frames
    # synchronises, so load / step / load / step is once_through every time), the policy's last measurement carried from clip to clip as in
    # any stream.  Five further clips, untimed, in front (the policy meets the stream; the coefficient sets exist).  Wall clock, sync at the end.
    try:
        import itertools
        perms = list(itertools.permutations(range(3)))

        def variant(i):
            v = clip_dev if i % 6 == 0 else clip_dev[..., list(perms[i % 6])]
            i //= 6
            for axis in (2, 1, 0):  # mirrored, upside down, backwards
                if i & 1:
                    v = v.flip(axis)
                i >>= 1
            if i & 1:
                v = 255 - v
            return v.contiguous() if v is not clip_dev else clip_dev
        distinct = min(args.steps, 24)
        lead = [variant(96 - 1 - k) for k in range(5)]
        for v in lead:
            enc.step_frames(v)
        enc.sync()
        del lead
        variants = [variant(k) for k in range(distinct)]
        p0 = enc.policy_info()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            enc.step_frames(variants[i % distinct])
        enc.sync()
        dt = time.perf_counter() - t0
        p1 = enc.policy_info()
        out["stream_of_clips"] = {"distinct_clips": distinct, "steps": args.steps, "ms_per_clip": dt / args.steps * 1e3, "value": info.pairs * args.steps / dt,
                                  "chunk_launches_speculated": p1["chunks_speculated"] - p0["chunks_speculated"],
                                  "chunk_launches": (p1["chunks_decided"] - p0["chunks_decided"]) or args.steps * chunks,
                                  "foreground_share_last": p1["foreground_share"],
                                  "note": "different clips resident in HBM, each encoded ONCE where it is (svc_clip_step_frames; clips rotate only beyond 24 "
                                          "steps); the speculation policy acts on the previous clips' measurements; five more clips untimed in front"}
        del variants
    except Exception as e:  # noqa: BLE001
        out["stream_of_clips"] = None
        out["stream_of_clips_note"] = f"not measured: {e}"
    enc.load_frames(clip_dev)  # (the resident clip again, policy voided, as after once_through)
    enc.reset_policy()
    torch.cuda.synchronize()
    p0 = enc.policy_info()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc.step()
    enc.sync()
    dt = time.perf_counter() - t0
    p1 = enc.policy_info()
    out["policy_voided_steps"] = {"steps": args.steps, "ms_per_step": dt / args.steps * 1e3, "value": info.pairs * args.steps / dt,
                                  "chunk_launches_speculated": p1["chunks_speculated"] - p0["chunks_speculated"],
                                  "chunk_launches": (p1["chunks_decided"] - p0["chunks_decided"]) or args.steps * chunks}
    out["note"] = ("outside the timed region; wall clock around step() + sync() (the pipeline's fill and drain included).  once_through: "
                   "load_frames before every repetition voids the speculation policy, so the step runs the two-pass order (in two halves on a big "
                   "shard: the idle-pipeline rule; --mixed-steps: the second half reading its frames once, blind); with_prior: the resident clip "
                   "again with the last measurement kept -- an ISOLATED one-pass step, whose RANSAC + segmentation + redo are exposed at its end (it can "
                   "be slower than once_through's two-pass halves); stream_of_clips: the same policy state in a stream, where nothing is exposed")
    return out


def _launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a CHILD process (never an exec: this process has not touched the GPU and does not from here on), relay its
    stdout (the one JSON line of rank 0), stderr and exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"bench.py: --gpus {n} without a launcher (WORLD_SIZE unset): starting {n} ranks as a child: {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SVC_BENCH_SELF_LAUNCHED="1")
    return subprocess.run(cmd, env=env).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3-1080p-3L-dct8-quant", choices=sorted(configs.ALL))
    ap.add_argument("--frames", type=int, default=0, help="override the clip length (0 = the config's)")
    ap.add_argument("--schedule", choices=("pipelined", "serial"), default="pipelined",
                    help="pipelined: software pipeline over consecutive steps (HBM-bound kernels back to back on one stream, RANSAC + "
                         "segmentation of the previous step beside them, halo in flight meanwhile); serial: one stream, stages back to back")
    ap.add_argument("--scaling", choices=("both", "strong", "weak"), default="both",
                    help="N > 1: which shardings to measure (default: strong = the BASELINE config 4/5 workload, then weak)")
    ap.add_argument("--no-segmentation", action="store_true",
                    help="region ids from the in-repo part only (foreground = one region) instead of the full segmentation glue")
    ap.add_argument("--wire", action="store_true", help="emit the serialised records of libs/encoder.cpp:222-269 (fused into the DCT kernel) instead of coefficient planes")
    ap.add_argument("--search-after-transform", action="store_true",
                    help="one rank: the motion search, not a pyramid pass, runs right behind the transform kernel (A/B)")
    ap.add_argument("--fork-behind-front", action="store_true",
                    help="A/B: RANSAC + segmentation of the previous step fork behind the front-of-step transform (beside the pyramid pass and the search)")
    ap.add_argument("--mixed-steps", action="store_true",
                    help="A/B: a step into an empty pipeline that knows nothing about the clip takes the mixed form (first half two passes, second half one pass, blind)")
    ap.add_argument("--whole-shard-steps", action="store_true",
                    help="never the idle-pipeline rule (a step that finds the pipeline empty -- first_encode's once-through step -- runs in two chunks on "
                         "big shards in the two-pass order); A/B")
    ap.add_argument("--two-bgr-passes", action="store_true",
                    help="luma + pyramid and the transform as two passes over the BGR clip (A/B of the default, which reads it once: the transform at the "
                         "front of the step also leaves the luma plane; what needs region ids -- the records' type words with --wire, the foreground "
                         "tiles' quantisation otherwise -- follows after the segmentation)")
    ap.add_argument("--always-speculate", action="store_true",
                    help="planes + quant: the speculative one-pass form on every step (default: only while the newest foreground share that has arrived "
                         "is at most 2 %%; A/B)")
    ap.add_argument("--chunk-pairs", type=int, default=0,
                    help="pipelined, one rank: frame pairs per chunk of EVERY step (0 = whole-shard launches, with the idle-pipeline rule for a step "
                         "that finds the pipeline empty)")
    ap.add_argument("--first-encode-reps", type=int, default=8, help="N = 1: repetitions of {load the clip, ONE step, sync} behind first_encode (0 = skip)")
    ap.add_argument("--time-every", type=int, default=4, help="record the per-stage HIP events on every n-th timed step (every step when --steps < 8)")
    # A/B switches (svc_clip_config tuning fields): kernel choice and launch shapes only, results never change.  The
    # environment variables of the round-2 scripts under tools/ are honoured HERE as defaults, not inside the library.
    ap.add_argument("--hbma-kernel", choices=("auto", "tiled", "lane", "wave"), default=os.environ.get("SVC_HBMA_KERNEL", "auto"),
                    help="motion search kernel: auto (the library's choice: LDS-tiled for 4 levels / r_top 1, lane-per-block for the other fused shapes, "
                         "else the per-level kernel), tiled (levels 2 and 1 from LDS tiles), lane (lane-per-block, no LDS), wave (per-level general kernel)")
    ap.add_argument("--lat-depth", type=int, default=int(os.environ.get("SVC_LAT_DEPTH", "0")),
                    help="pipelined: iterations RANSAC + segmentation get to finish (1..3; 0 = default 2)")
    ap.add_argument("--standalone-shapes", action="store_true", default=os.environ.get("SVC_LAUNCH_BESIDE", "1") == "0",
                    help="pipelined: keep the stand-alone launch shapes of RANSAC / segmentation")
    ap.add_argument("--segment-fork", action="store_true", default=os.environ.get("SVC_LAUNCH_NO_FORK", "1") == "0",
                    help="pipelined: let the segmentation fork its heavy attempts to a side stream")
    ap.add_argument("--sustain-seconds", type=float, default=3.0,
                    help="N = 1: after the timed steps, run back-to-back steps for this long and report sustained_ms_per_step (0 = skip)")
    ap.add_argument("--inline-rmse", action="store_true",
                    help="pipelined: RANSAC keeps its in-order RMSE sum inside its kernel (A/B of the deferred form)")
    ap.add_argument("--narrow-attempts", action="store_true",
                    help="segmentation: one workgroup per (frame, k-means attempt) even on small shards of large fields (A/B of the multi-launch form)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="N = 1: skip the PCIe-inclusive end_to_end object (measured outside the timed region)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="N = 1: measure end_to_end also on a shortened clip (--frames): by default only the full-length run does (it writes "
                         "~1.2 GB of clips to /dev/shm and starts four child processes on the GPU)")
    args = ap.parse_args()
    if args.frames and not args.end_to_end:
        args.no_end_to_end = True

    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: this process becomes the launcher and NEVER touches the GPU (nothing above this line has:
        # importing torch and the package does not initialise HIP) -- a run must never report fewer ranks than it was asked for
        raise SystemExit(_launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would report a rank count the run did not have")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # SVC_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks (RCCL
    # refuses two ranks on one device): ranks then share GPUs and the halo goes through gloo.
    backend = os.environ.get("SVC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    native.load()
    clipmod.load()
    torch.set_num_threads(min(8, torch.get_num_threads()))  # CPU share of a 1-GPU box is small
    comm_note = None
    if world > 1 and backend == "nccl" and os.environ.get("SVC_HALO", "rccl") == "rccl":
        # the C ABI's own communicator: rank 0 draws the id, torch.distributed carries the 128 bytes.  Every rank
        # learns whether EVERY rank got its communicator; if not, all of them use the torch.distributed transport
        # (also RCCL) -- the decision must be collective or the ranks would wait on different transports.
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        # ncclCommInitRank is a blocking collective: the ranks enter it all together or not at all, so whether librccl
        # binds (a per-process matter: dlopen, symbols) is agreed on FIRST
        if not clipmod.comm_available():
            ok.zero_()
            comm_note = f"librccl does not bind on rank {rank}: {native.load().svc_hip_last_error().decode()}"
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        box = [None]
        if int(ok.item()):
            try:
                box = [clipmod.comm_unique_id() if rank == 0 else None]
            except Exception as e:  # noqa: BLE001
                box, comm_note = [None], f"svc_hip_comm_unique_id failed: {e}"
            dist.broadcast_object_list(box, src=0, device=dev)
        if box[0] is not None:
            try:
                comm = clipmod.comm_create(box[0], rank, world)
                n_ranks, my_rank, _ = clipmod.comm_info(comm)
                if (n_ranks, my_rank) != (world, rank):
                    raise RuntimeError(f"the communicator reports rank {my_rank} of {n_ranks}, expected {rank} of {world}")
            except Exception as e:  # noqa: BLE001
                ok.zero_()
                comm_note = f"svc_hip_comm_create failed on rank {rank}: {e}"
        else:
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                clipmod.comm_destroy(comm)
            comm = None
            comm_note = comm_note or "another rank could not create its communicator"
            if rank == 0:
                print(f"bench.py: C-ABI RCCL communicator unavailable ({comm_note}); halo via torch.distributed P2P", file=sys.stderr)

    cfg = configs.ALL[args.config]
    modes = ["strong"] if world == 1 else (["strong", "weak"] if args.scaling == "both" else [args.scaling])
    results = {m: run_mode(args, cfg, m, rank, world, dev, backend, comm) for m in modes}

    rccl_mismatch = False
    if rank == 0:
        main_mode = modes[0]
        r = results[main_mode]
        info = r["info"]
        pw, ph = cfg.padded
        kt, nl = r["stage_ms_per_step"], r["launches_per_step"]
        # type_patch (what a one-pass step owes once its region ids exist) rides the latency stream behind the segmentation
        side = ("ransac", "segment", "type_patch", "halo_exchange") if args.schedule == "pipelined" else ("halo_exchange",)
        main_kt = {k: v for k, v in kt.items() if k not in side}
        side_kt = {k: v for k, v in kt.items() if k in side}
        elapsed = r["elapsed"]
        pol = r["policy"]
        # the share of the timed region's chunk launches that read the BGR clip once (wire: all of them where the tuned emitter applies)
        one_pass_frac = (1.0 if "type_patch" in kt else 0.0) if args.wire or not pol["had_the_choice"] else pol["speculated"] / pol["chunk_launches"]
        out = {
            "metric": "encoded frames/sec (1080p, 16x16 HBMA+DCT)" if cfg.name.startswith("C3") else
                      f"encoded frames/sec ({cfg.name})",
            "value": r["encoded_per_step"] * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": main_mode if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u8 (SAD, integer argmin) / f64 accumulate -> f32 (DCT, quant)",
            "data": "synthetic",
            "config": {
                "workload": cfg.name + (" (BASELINE config 4: the 300-frame clip cut over the ranks)" if world > 1 and main_mode == "strong" and cfg.name.startswith("C3-") else ""),
                "frame": f"{cfg.width}x{cfg.height} -> padded {pw}x{ph}",
                "clip_frames": r["clip_frames"],
                "frames_per_gpu": info.frames if world == 1 else [clipmod.plan_shard(r["clip_frames"], world, q)[1] for q in range(world)],
                "encoded_frames_per_step": r["encoded_per_step"],
                "foreground_mv_blocks": r.get("foreground_share"),
                "value_is": ("steady state: K back-to-back steps over the resident clip, the speculation policy taught by the warm-up steps.  A STREAM of "
                             "different clips, each encoded once where it is (svc_clip_step_frames), runs at the same rate: first_encode.stream_of_clips; "
                             "ONE clip encoded alone (load, step, sync: nothing to overlap with, nothing known) costs first_encode.once_through"),
                "bgr_passes_per_step": ("two (luma + pyramid, later the transform)" if one_pass_frac == 0 else
                                        "one (records + luma plane from one kernel; type words stored after the segmentation)" if args.wire else
                                        "one, speculative (every tile quantised as background + the luma plane at the front of the step; foreground tiles redone "
                                        "in type_patch)" if one_pass_frac == 1 else
                                        f"MIXED: {pol['speculated']} of the timed region's {pol['chunk_launches']} chunk launches speculated (one pass), the others ran "
                                        "the two-pass order: the transform's time and bytes below are the launch-weighted mix"),
                "speculation_policy": pol,
                "chunks_per_step": r["chunks"],
                "output_sets": r["output_sets"],
                "pyr_levels": cfg.levels, "mv_block": cfg.mv_block, "search_range": cfg.search_range,
                "dct_block": cfg.dct_block, "quant": {"fg": cfg.fg_step, "bg": cfg.bg_step},
                "schedule": (("software pipeline over the CHUNKS of consecutive steps (a step = " + str(r["chunks"]) + " chunk(s) of frame pairs): luma+pyramid(m), motion "
                              "search(m), transform(m-3) back to back on one stream; RANSAC+segmentation(m-1) beside them for up to two iterations on two "
                              "alternating streams" if world == 1 else
                              "software pipeline over consecutive steps: luma+pyramid(s), motion search(s-1), transform(s-4) back to back on one stream; "
                              "RANSAC+segmentation(s-2) beside them for up to two iterations, consecutive steps on two alternating streams; halo(s) on its own stream")
                             if args.schedule == "pipelined" else "one stream, stages back to back"),
                "driver": "svc::ClipEncoder (C++, include/svc/clip_encoder.hpp)",
                "parallelism": f"frame-sharded x{world}" + (f", halo = 1 pyramid/rank/step via {r['halo']}" if world > 1 else ""),
            },
            "kernel_ms_per_step": main_kt,
            "kernel_timing": f"HIP events on the launch streams, on {r['timed_steps']} of the {args.steps} timed steps; launches timed per stage: {r['launches_timed']}",
        }
        if side_kt:
            out["overlapped_ms_per_step"] = {**side_kt, "note": "event-to-event time on the second / communication stream; these stages run BESIDE "
                                             "the main stream's kernels (and wait for CUs there), so they are not additive with kernel_ms_per_step"}
        if world > 1:
            out["halo_exchange_ms"] = kt.get("halo_exchange")
            out["rank0_kernel_ms_per_step"] = main_kt
            pr = r["per_rank"]
            out["multi_gpu"] = {
                "transport": r["halo"],
                # ncclCommCount of the C ABI's communicator; torch's RCCL process group when the halo went through it; None = gloo rehearsal
                "rccl_ranks": clipmod.comm_info(comm)[0] if comm is not None else (dist.get_world_size() if backend == "nccl" else None),
                "rccl_ranks_source": "ncclCommCount (svc_hip_comm_info)" if comm is not None else
                                     ("torch.distributed process group, backend nccl = RCCL" if backend == "nccl" else None),
                "launcher": "bench.py itself (torch.distributed.run as a child process)" if os.environ.get("SVC_BENCH_SELF_LAUNCHED") else "external",
                "transport_note": comm_note,
                "halo_check": r["halo_check"],
                "ms_per_step_by_rank": [x[0] for x in pr],
                "ms_per_step_min": min(x[0] for x in pr), "ms_per_step_max": max(x[0] for x in pr),
                "halo_exchange_ms_by_rank": [x[1] if x[1] >= 0 else None for x in pr],
                "frames_by_rank": [int(x[2]) for x in pr], "encoded_by_rank": [int(x[3]) for x in pr],
                # which order each rank ran: the policy is per rank and involves no collective, so ranks may differ in order (never in bytes);
                # a straggler that did not speculate shows here
                "per_rank": [{"rank": q, "ms_per_step": x[0], "halo_exchange_ms": x[1] if x[1] >= 0 else None, "frames": int(x[2]), "encoded": int(x[3]),
                              "launches_with_the_choice": int(x[4]), "launches_speculated": int(x[5]),
                              "bgr_passes_per_step": "two" if x[5] == 0 else ("one" if x[5] == x[4] else "mixed")} for q, x in enumerate(pr)],
                "note": "each rank's own wall clock over the timed steps / steps; `ms_per_step` of the line is the max over ranks "
                        "between barriers.  halo_exchange_ms: event-to-event on the communication stream (includes waiting for the "
                        "neighbour's pyramid kernel)",
            }
            if backend == "nccl" and out["multi_gpu"]["rccl_ranks"] != world:
                print(f"bench.py: RCCL reports {out['multi_gpu']['rccl_ranks']} ranks, the line would say n_gpus = {world}", file=sys.stderr, flush=True)
                rccl_mismatch = True
            out["multi_gpu"]["prediction"] = predicted_step(cfg, max(int(x[2]) for x in pr), out["ms_per_step"], world) if main_mode == "strong" else None
        if "weak" in results and main_mode != "weak":
            w = results["weak"]
            out["weak"] = {"value": w["encoded_per_step"] * args.steps / w["elapsed"], "unit": "frames/s",
                           "ms_per_step": w["elapsed"] / args.steps * 1e3, "clip_frames": w["clip_frames"],
                           "frames_per_gpu": w["info"].frames, "encoded_frames_per_step": w["encoded_per_step"],
                           "rank0_kernel_ms_per_step": {k: v for k, v in w["stage_ms_per_step"].items() if k not in side}}
        if "hbma" in kt:
            # the library's own dispatch names the kernel this run launched (svc_hip_hbma_kernel_name)
            kname = native.hbma_kernel_name(cfg.levels, pw, ph, cfg.search_range, cfg.mv_block, cfg.mv_block, hbma_flags_of(args))

            def traffic(group, key, pairs, launched=None):
                """Counter traffic is collected offline (rocprofv3 --pmc passes cannot share a run with the timed region) and only quoted
                when it is tied to THIS build: same kernel dispatched, same sources (hash recorded at collection time).  Recorded for a
                whole-clip launch, scaled to this launch's pairs.  Returns (bytes or None, where it came from or why it is null)."""
                v, base, why = pipeline.pmc_traffic_for(cfg.name, group, key, launched)
                if v is None:
                    return None, f"null: {why}"
                return v * pairs / base, f"offline PMC ({why}), same {group} sources as this build, scaled from {base} to {pairs} frame pairs; not measured in this run"
            hbma_bytes = cfg.hbma_bytes_per_frame() * info.pairs   # per step; a launch covers 1 / chunks of it
            hbma_ms = kt["hbma"] / nl["hbma"]                       # average launch
            hbma_gbps = hbma_bytes / (kt["hbma"] * 1e-3) / 1e9
            hbma_traffic, hbma_traffic_source = traffic("hbma", "hbma_bytes_per_launch", info.pairs, kname)
            if hbma_traffic is not None:
                hbma_traffic /= nl["hbma"]
            out["roofline"] = {
                "kernel": {"hbma_tiled16_kernel": "hbma_tiled16_kernel (MAD search, all pyramid levels, windows of levels 2 and 1 staged in LDS)",
                           "hbma_fused_kernel": "hbma_fused_kernel (MAD search, all pyramid levels, lane per block)",
                           "hbma_wave_level_kernel": "hbma_wave_level_kernel (LDS-staged wave-per-block search)"}[kname],
                "bound": "hbm", "achieved": hbma_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": hbma_gbps / HBM_PEAK_GBPS,
                "traffic": hbma_traffic,
                "traffic_source": hbma_traffic_source,
                "algorithmic_bytes_per_launch": hbma_bytes / nl["hbma"],
                "pairs_per_launch": info.pairs / nl["hbma"],
                "avg_launch_ms": hbma_ms,
                "launches_per_step": nl["hbma"],
                "note": "HBM is the stated bound; measured VALU busy ~86 % (byte-SAD ops issue at 4 cycles/wave): VALU time ~= HBM floor, DESIGN.md 4.1"
                        + ("; RANSAC + segmentation of an earlier step may still be running beside it (pipelined schedule)" if args.schedule == "pipelined" else ""),
            }
            if "dct_quant" in kt:
                # one pass over the BGR clip: the transform kernel at the front of the step also stores the luma plane; a timed region that mixed both
                # orders (the adaptive policy switching inside it) gets the launch-weighted bytes and says so in its label
                one_pass = one_pass_frac > 0
                dct_bytes = (cfg.dct_bytes_per_frame() + pw * ph * one_pass_frac) * info.pairs
                dct_ms = kt["dct_quant"] / nl["dct_quant"]
                dct_gbps = dct_bytes / (kt["dct_quant"] * 1e-3) / 1e9
                # one figure per kernel variant: records (+ luma plane) with --wire read once, planes + luma plane when the step speculated,
                # the plain transform in the two-pass order (none was collected for --wire --two-bgr-passes)
                dct_key = ("dct_records_bytes_per_launch" if one_pass else "dct_records_two_passes_bytes_per_launch") if args.wire else \
                    ("dct_luma_bytes_per_launch" if one_pass else "dct_bytes_per_launch")
                dct_traffic, dct_traffic_source = traffic("dct", dct_key, info.pairs) if one_pass_frac in (0.0, 1.0) else (None, "null: the timed region mixed both transform kernels")
                if dct_traffic is not None:
                    dct_traffic /= nl["dct_quant"]
                out["roofline_dct"] = {
                    "kernel": ((f"MIXED over the timed region ({pol['speculated']} of {pol['chunk_launches']} launches speculative): " if 0 < one_pass_frac < 1 else "") +
                               f"dct_kernel<{cfg.dct_block}, quant, luma, speculative> (every tile quantised as background + the luma plane from one pass over the BGR clip; "
                               "the foreground tiles are redone in type_patch)" if one_pass and not args.wire else
                               f"dct_kernel<{cfg.dct_block}, records, luma> (records of the raw coefficients + the luma plane from one pass over the BGR clip)" if one_pass else
                               f"dct_kernel<{cfg.dct_block}, records> (raw coefficients, libs/encoder.cpp:638-650)" if args.wire else
                               f"dct_kernel<{cfg.dct_block}, quant> (the step's longest kernel)"),
                    "bound": "hbm", "achieved": dct_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": dct_gbps / HBM_PEAK_GBPS,
                    "traffic": dct_traffic,
                    "traffic_source": dct_traffic_source,
                    "algorithmic_bytes_per_launch": dct_bytes / nl["dct_quant"], "frames_per_launch": info.pairs / nl["dct_quant"],
                    "avg_launch_ms": dct_ms, "launches_per_step": nl["dct_quant"],
                }
        if world == 1 and "hbma" in kt and "dct_quant" in kt:
            # the whole step against the same peak: the three main-stream kernels run back to back and are all
            # HBM-bound, so (their algorithmic bytes) / (step time) says how far the STEP is from the roofline
            step_bytes = cfg.luma_pyramid_bytes_per_frame() * info.frames + hbma_bytes + dct_bytes
            if one_pass_frac == 1:
                # BGR in once, pyramid out, coefficients out: what the one-pass form must move (the transform's figure above already holds the BGR
                # read and the luma plane's store; left of the pyramid stage are the first frame's own luma pass and the levels above 0)
                step_bytes = hbma_bytes + dct_bytes + cfg.luma_pyramid_bytes_per_frame() * (info.frames - info.pairs) + \
                    (cfg.luma_pyramid_bytes_per_frame() - 4 * pw * ph) * info.pairs
            step_gbps = step_bytes / (r["elapsed"] / args.steps) / 1e9
            out["roofline_step"] = {"bound": "hbm", "achieved": step_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": step_gbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": step_bytes,
                                    "note": ("one pass over the BGR clip: BGR in once, pyramid out, motion search, coefficients out / ms_per_step (the redo of the foreground "
                                             "tiles is not counted as algorithmic bytes); the same step as two passes moves 3 W H more per frame" if one_pass else
                                             "luma+pyramid, motion search and transform of one step / ms_per_step; "
                                             "RANSAC + segmentation move < 1 % of these bytes")}
        if r.get("first_encode"):
            out["first_encode"] = r["first_encode"]
        if r.get("sustained"):
            out["sustained_ms_per_step"] = r["sustained"]["ms_per_step"]
            out["sustained"] = {**r["sustained"], "value": r["encoded_per_step"] / (r["sustained"]["ms_per_step"] * 1e-3), "unit": "frames/s",
                                "note": "untimed for `value`: back-to-back steps after the timed region"}
        if world == 1 and not args.no_hbm_probe:
            # context only: what plain streaming kernels get from this box's HBM, in the form that gets the most out of it (one 4 KiB
            # unit per short-lived workgroup, XCD-contiguous order: tools/ubench_stream_oneshot.hip)
            out["hbm_streaming_measured"] = {"unit": "GB/s", **hbm_streaming_rates(dev),
                                             "note": "svc_hip_probe_stream on this GPU (one unit per workgroup, XCD-contiguous), context only; roofline "
                                                     "fractions are against the 8 TB/s peak"}
        if r["sample_frames"] is not None and not args.no_end_to_end:
            out["end_to_end"] = end_to_end_rates(cfg, r["sample_frames"])
        if r["sample_frames"] is not None and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, r["sample_frames"])
            cb = out["cpu_baseline"]
            if cb.get("hbma_ms_per_frame") and "hbma" in kt:
                # the one like-for-like ratio: the unmodified reference's motion search vs the MAD kernel, per frame pair
                out["speedup_hbma_vs_cpu_1core"] = cb["hbma_ms_per_frame"] / (kt["hbma"] / info.pairs)
        if not rccl_mismatch:
            print(json.dumps(out), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        clipmod.comm_destroy(comm)
    if world > 1:
        dist.destroy_process_group()
    if rccl_mismatch:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
