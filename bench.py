#!/usr/bin/env python3
"""bench.py -- encoded frames/s of the MI355X encode hot path on the BASELINE.json
headline workload: 1080p (padded 1920x1088), 300-frame synthetic clip, 16x16 MV
blocks, 3-level HBMA + RANSAC + 8x8 DCT + quant (fg 1 / bg 640).

One "step" = one pass of the hot path over the whole clip resident in HBM:
  luma + pyramid (all frames) -> [N>1: halo exchange of the previous rank's last
  pyramid, RCCL send/recv] -> fused HBMA (all frame pairs) -> RANSAC (per frame) ->
  segmentation (mask, close/open, k-means, connected components -> region ids) ->
  fused DCT + quant (per encoded frame).
`value` = encoded frames of all ranks / max-over-ranks time, BGR frames already
resident in HBM when the timed region starts (PCIe excluded).

Contract: python bench.py --gpus N --steps K --warmup W ; for N > 1 the driver
launches it under torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints
ONE JSON line.
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

from scalable_video_codec_amd import configs, native, pipeline, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(cfg: configs.CodecConfig, frames_bgr, budget_s: float = 12.0):
    """Times the CPU path on THIS host, one core, on the first few frames of the same
    clip.  Motion search: the unmodified reference (oracle/_ref) when it was built,
    else the C restatement; RANSAC / DCT / quant: the restatement (cv::dct cannot be
    built offline).  Only this leg and tests may touch oracle/."""
    import numpy as np
    from oracle import binding
    orc = binding.Oracle()
    ref = binding.Reference() if binding.Reference.available() else None
    pw, ph = cfg.padded
    # a bounded sample of the same clip: enough frames for ~10-15 s of single-core work
    n = min(len(frames_bgr), 128)
    host = [f.cpu() for f in frames_bgr[:n]]
    pyrs = {}

    def pyr(i):  # built lazily (the pre-step is not part of the timed CPU work, as on the GPU side's input)
        if i not in pyrs:
            pyrs[i] = [p.numpy() for p in synth.build_pyramid(synth.bgr_to_y(host[i]), cfg.levels)]
        return pyrs[i]
    k = orc.ransac_iter_count(**binding.DEFAULT_RANSAC)
    t_hbma = t_rest = 0.0
    done = 0
    t_start = time.perf_counter()
    # the reference's default build runs its SSE2 entry, which exists for 4 levels / 16x16 only
    # (libs/motion.hpp:143-147, libs/encoder.cpp:472-476); other level counts run the generic path
    use_sse2 = cfg.levels == 4 and cfg.mv_block == 16
    if ref is not None:
        search = (lambda t, a: ref.hbma16_sse2(t, a, cfg.search_range)) if use_sse2 else \
                 (lambda t, a: ref.hbma(t, a, cfg.search_range, cfg.mv_block, cfg.mv_block))
        search_name = "unmodified reference " + ("EstimateMotionHierarchical16x16Sse2" if use_sse2 else
                                                "EstimateMotionHierarchical (generic path; the SSE2 path only exists for 4 levels)")
    else:
        search = (lambda t, a: orc.hbma16_sse2(t, a, cfg.search_range)) if use_sse2 else \
                 (lambda t, a: orc.hbma(t, a, cfg.search_range, cfg.mv_block, cfg.mv_block))
        search_name = "C restatement" + (" (SSE2 path)" if use_sse2 else "")
    search(pyr(0), pyr(1))  # warm-up: page in, let the core clock up
    mfw, mfh = cfg.mv_field

    def encode_frame(i, ta, tb_):  # everything the hot path does for encoded frame i, on the CPU
        t0 = time.perf_counter()
        mv, _ = search(ta, tb_)
        t1 = time.perf_counter()
        samples = (np.arange(k, dtype=np.uint32) * 2654435761 % len(mv)).astype(np.uint32)
        _, _, inl = orc.ransac(mv, samples, **binding.DEFAULT_RANSAC)
        inl_mask = np.zeros(len(mv), np.uint8)
        inl_mask[inl] = 1
        types = orc.segment(inl_mask, mv, mfw, mfh, cfg.mv_block, cfg.mv_block, seed=i)
        if cfg.dct_block:
            planes = orc.dct_frame_f32(host[i].numpy(), cfg.dct_block, cfg.dct_block)
            orc.quant_frame(planes, cfg.mv_block, cfg.mv_block, types, cfg.fg_step, cfg.bg_step)
        return t1 - t0, time.perf_counter() - t1

    busy = 0.0
    for i in range(1, n):
        ta, tb_ = pyr(i - 1), pyr(i)
        pyrs.pop(i - 2, None)
        dt_search, dt_rest = encode_frame(i, ta, tb_)
        t_hbma += dt_search
        t_rest += dt_rest
        busy += dt_search + dt_rest
        done += 1
        if busy > budget_s:
            break
    # the same work frame-parallel on the host's cores (the reference itself encodes on one thread,
    # apps/encoder.cpp:228; this is what a frame-parallel CPU deployment of it would get): the C entry
    # points release the GIL, so plain threads scale
    import concurrent.futures as cf
    cores = max(1, min(len(os.sched_getaffinity(0)), 32))
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
        del pp
    # the reference's own fast path (SSE2, fixed 4 levels) on the same frames, for context
    sse2_ms = None
    if ref is not None and not use_sse2 and cfg.mv_block == 16 and cfg.padded[0] % 8 == 0 and cfg.padded[1] % 8 == 0:
        p4 = [[p.numpy() for p in synth.build_pyramid(synth.bgr_to_y(f), 4)] for f in host[:2]]
        ref.hbma16_sse2(p4[0], p4[1], cfg.search_range)
        t0 = time.perf_counter()
        for _ in range(3):
            ref.hbma16_sse2(p4[0], p4[1], cfg.search_range)
        sse2_ms = (time.perf_counter() - t0) / 3 * 1e3
    total = t_hbma + t_rest
    return {
        "value": done / total if total > 0 else None,
        "unit": "frames/s",
        "cores": 1,
        "kind": "reference" if ref is not None else "port",
        "sample": (f"first {done} encoded frames of the same clip, 1 thread: motion search = {search_name}"
                   f", RANSAC/segmentation/DCT(f64 separable)/quant = C restatement (cv::dct is not buildable offline)"),
        "hbma_ms_per_frame": t_hbma / done * 1e3 if done else None,
        "ransac_dct_quant_ms_per_frame": t_rest / done * 1e3 if done else None,
        "reference_sse2_4level_hbma_ms_per_frame": sse2_ms,
        "all_cores": all_cores,
    }


def hbm_streaming_rates(device) -> dict:
    """What a plain streaming kernel (svc_hip_probe_stream: dwordx4 per lane, contiguous) reaches on THIS GPU, in GB/s,
    for the read/write mixes of the three HBM-bound kernels.  Measured after the timed region (N = 1 only), with
    HIP events around three launches each over 2 GiB buffers."""
    n = 2 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    a.fill_(1)
    out = {}
    for name, r, w in (("read_only", 3, 0), ("write_only", 0, 1), ("3_read_1_write", 3, 1), ("1_read_4_write", 1, 4)):
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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3-1080p-3L-dct8-quant", choices=sorted(configs.ALL))
    ap.add_argument("--frames", type=int, default=0, help="override the clip length (0 = the config's)")
    ap.add_argument("--chunks", type=int, default=1,
                    help="experiment kept for the record: cut the clip into this many chunks, the transform of chunk k overlapping the front of "
                         "chunk k+1 on a second stream -- slower than whole-clip launches (the latency-bound kernels are paid per chunk)")
    ap.add_argument("--no-segmentation", action="store_true",
                    help="region ids from the in-repo part only (foreground = one region) instead of the full segmentation glue")
    ap.add_argument("--wire", action="store_true", help="emit the serialised records of libs/encoder.cpp:222-269 (fused into the DCT kernel) instead of coefficient planes")
    ap.add_argument("--overlap", action="store_true",
                    help="software-pipeline consecutive passes on two streams (back end of pass s beside the front end of pass s+1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # SVC_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks (RCCL
    # refuses two ranks on one device): ranks then share GPUs and the halo goes through gloo.
    backend = os.environ.get("SVC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    native.load()
    torch.set_num_threads(min(8, torch.get_num_threads()))  # CPU share of a 1-GPU box is small

    cfg = configs.ALL[args.config]
    n_frames = args.frames or cfg.frames
    dev = torch.device("cuda", local_rank)

    # this rank's slice of a (world * n_frames)-frame clip: one generator, offset frames
    clip = synth.SynthClip(cfg.width, cfg.height, world * n_frames, cfg.seed, device=dev)
    pw, ph = cfg.padded
    frames = [synth.pad_frame(clip.frame_bgr(rank * n_frames + t), pw, ph) for t in range(n_frames)]
    enc = pipeline.ClipEncoder(cfg, n_frames, dev, rank=rank, world=world, segmentation=not args.no_segmentation,
                               wire=args.wire)
    enc.load_frames(frames)
    del clip
    torch.cuda.synchronize()

    def one_step(timed=False):
        if args.overlap:
            enc.step_overlapped(timed=timed)
        else:
            enc.step(timed=timed, chunks=args.chunks)

    for _ in range(args.warmup):
        one_step()
    enc.finish_overlapped()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    enc.reset_kernel_timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(timed=True)
    enc.finish_overlapped()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    encoded = torch.tensor([enc.encoded_per_step], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(encoded, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_encoded = float(encoded.item())

    if rank == 0:
        kt = enc.kernel_times_ms()  # per-launch averages from HIP events on the launch stream
        hbma_bytes = cfg.hbma_bytes_per_frame() * enc.pairs_per_step
        dct_bytes = cfg.dct_bytes_per_frame() * enc.encoded_per_step
        hbma_gbps = hbma_bytes / (kt["hbma"] * 1e-3) / 1e9
        has_dct = "dct_quant" in kt
        dct_gbps = dct_bytes / (kt["dct_quant"] * 1e-3) / 1e9 if has_dct else None
        nl = enc.launches_per_step()  # launches per step of each stage (= --chunks)
        pmc = pipeline.load_pmc_traffic()

        def per_launch_traffic(key, stage):
            # PMC traffic was recorded for whole-clip launches; a chunked launch moves 1/chunks of it
            v = pmc.get(key)
            return v / nl[stage] if v is not None else None
        out = {
            "metric": "encoded frames/sec (1080p, 16x16 HBMA+DCT)" if cfg.name.startswith("C3") else
                      f"encoded frames/sec ({cfg.name})",
            "value": total_encoded * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 (SAD, integer argmin) / f64 accumulate -> f32 (DCT, quant)",
            "data": "synthetic",
            "config": {
                "workload": cfg.name,
                "frame": f"{cfg.width}x{cfg.height} -> padded {pw}x{ph}",
                "frames_per_gpu": n_frames,
                "encoded_frames_per_step": total_encoded,
                "pyr_levels": cfg.levels, "mv_block": cfg.mv_block, "search_range": cfg.search_range,
                "dct_block": cfg.dct_block, "quant": {"fg": cfg.fg_step, "bg": cfg.bg_step},
                "chunks_per_step": args.chunks,
                "schedule": "two-stream software pipeline across passes" if args.overlap else "one stream, passes back to back",
                "parallelism": f"frame-sharded x{world}" + (f" + {'RCCL' if backend == 'nccl' else backend} halo (1 pyramid/rank/step)" if world > 1 else ""),
            },
            "roofline": {
                "kernel": "hbma_fused16_kernel (MAD search, all pyramid levels)",
                "bound": "hbm", "achieved": hbma_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": hbma_gbps / HBM_PEAK_GBPS,
                "traffic": per_launch_traffic("hbma_bytes_per_launch", "hbma") if cfg.name.startswith("C3-") else None,
                "algorithmic_bytes_per_launch": hbma_bytes / nl["hbma"],
                "avg_launch_ms": kt["hbma"] / nl["hbma"],
                "launches_per_step": nl["hbma"],
                "note": "HBM is the stated bound; measured VALU busy ~86 % (byte-SAD ops issue at 4 cycles/wave): "
                        "VALU time ~= HBM floor, see DESIGN.md 4.1",
            },
            "roofline_dct": {
                "kernel": f"dct_kernel<{cfg.dct_block}, quant> (the step's longest kernel)",
                "bound": "hbm", "achieved": dct_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": dct_gbps / HBM_PEAK_GBPS,
                "traffic": per_launch_traffic("dct_bytes_per_launch", "dct_quant") if cfg.name.startswith("C3-") else None,
                "algorithmic_bytes_per_launch": dct_bytes / nl["dct_quant"],
                "avg_launch_ms": kt["dct_quant"] / nl["dct_quant"],
                "launches_per_step": nl["dct_quant"],
            } if has_dct else None,
            "kernel_ms_per_step": kt,
        }
        if not has_dct:
            out.pop("roofline_dct")
        if "hbma_wave" in (enc.hbma_kernel_name or ""):
            out["roofline"]["kernel"] = "hbma_wave_level_kernel (LDS-staged wave-per-block search)"
        if world == 1:
            # context for the roofline fractions: what plain streaming kernels get from this box's HBM
            rates = hbm_streaming_rates(dev)
            out["hbm_streaming_measured"] = {"unit": "GB/s", **rates,
                                             "note": "svc_hip_probe_stream on this GPU; the MAD kernel is read-only, luma+pyramid 3:1, DCT+quant 1:4"}
            out["roofline"]["frac_of_streaming_read_rate"] = hbma_gbps / rates["read_only"]
            if has_dct:
                out["roofline_dct"]["frac_of_streaming_fill_rate"] = dct_gbps / rates["write_only_torch_fill"]
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cfg, frames)
            if out["cpu_baseline"]["value"]:
                out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
