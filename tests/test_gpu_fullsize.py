"""BASELINE.json's full-size workload (C3: 1080p, 300 frames, 3 levels, 8x8 DCT + quant)
through size-independent properties, plus sampled pairs against the oracle."""
import numpy as np
import pytest
import torch

from scalable_video_codec_amd import clip as clipmod
from scalable_video_codec_amd import configs, pipeline, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def encoded(native):
    cfg = configs.C3
    dev = torch.device("cuda")
    clip = synth.SynthClip(cfg.width, cfg.height, cfg.frames, cfg.seed, device=dev)
    pw, ph = cfg.padded
    frames = [synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(cfg.frames)]
    enc = pipeline.ClipEncoder(cfg, cfg.frames, dev)
    enc.load_frames(frames)
    enc.step()
    torch.cuda.synchronize()
    enc.frames_bgr = frames  # the padded source frames, for the driver tests below
    return cfg, enc


def _planes(pyr_flat, slot, stride, pw, ph, levels):
    offs = synth.level_offsets(pw, ph, levels)
    flat = pyr_flat[slot * stride:(slot + 1) * stride].cpu().numpy()
    return [flat[offs[l]:offs[l] + (pw >> l) * (ph >> l)].reshape(ph >> l, pw >> l) for l in range(levels)]


def test_bench_driver_at_the_headline_size(native, oracle, encoded):
    """What bench.py times -- svc::ClipEncoder (C++), pipelined schedule, the 300-frame C3 clip, several steps in flight --
    against (a) the stage-by-stage calls of the fixture, every output buffer bit for bit, and (b) the oracle directly on
    the DRIVER's outputs: motion search of pairs 0 / 150 / 298 (libs/motion.cpp:412-465), RANSAC (:182-266) and region
    ids (libs/encoder.cpp:507-623) of the same pairs.  Frame order per libs/encoder.cpp:472-498, 661-663."""
    cfg, ref = encoded
    dev = torch.device("cuda")
    drv = clipmod.Clip(cfg, cfg.frames, schedule=clipmod.PIPELINED)
    assert (drv.info.frames, drv.info.pairs) == (cfg.frames, cfg.frames - 1)
    drv.load_frames(torch.stack(ref.frames_bgr).contiguous())
    for _ in range(3):
        drv.step()
    drv.sync()  # the foreground share of these steps has landed: the next ones read the clip once (speculative transform, round 5)
    before = drv.policy_info()
    for _ in range(2):
        drv.step()
    drv.sync()
    after = drv.policy_info()
    # the step whose outputs are compared below DID speculate: every chunk launch of the last two steps took the one-pass form
    assert after["chunks_speculated"] - before["chunks_speculated"] == 2 * drv.info.chunks_per_step, (before, after)
    assert 0 <= after["foreground_share"] <= 0.02
    out = drv.outputs(device=dev)
    assert torch.equal(out["mv"], ref.mv) and torch.equal(out["min_mad"], ref.mad)
    assert out["global_motion"].cpu().numpy().tobytes() == ref.gm.cpu().numpy().tobytes()
    assert out["rmse"].cpu().numpy().tobytes() == ref.rmse.cpu().numpy().tobytes()
    assert torch.equal(out["inlier_mask"], ref.mask) and torch.equal(out["inlier_count"], ref.count)
    assert torch.equal(out["block_types"], ref.types)
    coeffs = drv.read("coeffs", device=dev)
    assert torch.equal(coeffs.view(ref.coeffs.shape), ref.coeffs)
    # (c) the speculated step's coefficient planes against the oracle DIRECTLY (libs/encoder.cpp:323-339 + libs/decoder.cpp:130-144): tiles of
    # foreground MV blocks (redone with fg_step after the segmentation) and of background blocks (quantised at the front of the step), sampled
    # over the frames that have foreground; 1e-4 * max(1, |ref|) on the raw coefficient means equality away from the quantiser's rounding edges
    types = out["block_types"].cpu().numpy().astype(np.uint32)
    fg_frames = np.flatnonzero((types != 0).any(axis=1))
    assert len(fg_frames) >= 4
    rng = np.random.default_rng(6)
    planes = coeffs.view(ref.coeffs.shape)
    n_fg = n_bg = n_equal = n_total = 0
    for p in rng.choice(fg_frames, 4, replace=False):
        frame = ref.frames_bgr[p + 1].cpu().numpy()
        want = oracle.quant_frame(oracle.dct_frame_f64(frame, 8, 8).astype(np.float32), 16, 16, types[p], cfg.fg_step, cfg.bg_step)
        got = planes[p].cpu().numpy()
        fg_blocks = np.flatnonzero(types[p] != 0)
        bg_blocks = np.flatnonzero(types[p] == 0)
        for blocks, is_fg in ((rng.choice(fg_blocks, min(12, len(fg_blocks)), replace=False), True), (rng.choice(bg_blocks, 12, replace=False), False)):
            for b in blocks:
                by, bx = divmod(int(b), ref.mfw)
                tile_got = got[:, by * 16:by * 16 + 16, bx * 16:bx * 16 + 16]   # the four 8x8 tiles of the MV block, three channels
                tile_want = want[:, by * 16:by * 16 + 16, bx * 16:bx * 16 + 16]
                step = cfg.fg_step if is_fg else cfg.bg_step
                assert np.abs(tile_got - tile_want).max() <= step, (p, b, is_fg)
                n_equal += int((tile_got == tile_want).sum()); n_total += tile_got.size
                n_fg += 4 * is_fg; n_bg += 4 * (not is_fg)
    assert n_fg >= 32 and n_bg >= 32 and n_fg + n_bg >= 64, (n_fg, n_bg)
    assert n_equal / n_total > 0.999, (n_equal, n_total)
    del coeffs, planes
    pyr = drv.read("pyramids", device=dev)
    assert torch.equal(pyr[drv.info.pyramid_stride:], ref.pyr[ref.stride:])  # slot 0 = halo: unused at world 1
    from oracle.binding import DEFAULT_RANSAC
    iters = drv.info.ransac_iters
    for p in (0, 150, 298):  # pair p = frames p, p + 1 = pyramid slots p + 1, p + 2
        mv, mad = oracle.hbma(_planes(pyr, p + 1, ref.stride, ref.pw, ref.ph, cfg.levels),
                              _planes(pyr, p + 2, ref.stride, ref.pw, ref.ph, cfg.levels), cfg.search_range, 16, 16)
        assert np.array_equal(out["mv"][p].cpu().numpy(), mv) and np.array_equal(out["min_mad"][p].cpu().numpy(), mad)
        samples = pipeline.ransac_samples(cfg.frames - 1, iters, 1, cfg.blocks, cfg.seed, "cpu")[p].numpy().astype(np.uint32).ravel()
        gm, rmse, inl = oracle.ransac(mv, samples, **DEFAULT_RANSAC)
        assert out["global_motion"][p].cpu().numpy().tobytes() == gm.tobytes()
        assert np.float32(out["rmse"][p].item()).tobytes() == rmse.tobytes()
        assert np.array_equal(np.flatnonzero(out["inlier_mask"][p].cpu().numpy()), inl)
        want = oracle.segment(out["inlier_mask"][p].cpu().numpy(), mv, ref.mfw, ref.mfh, seed=ref.seg_seed + p)
        assert np.array_equal(out["block_types"][p].cpu().numpy().astype(np.uint32), want)
    drv.close()


def test_a_step_into_an_empty_pipeline_runs_in_two_chunks(native, encoded):
    """The idle-pipeline rule (round 6): a step that finds the pipeline EMPTY -- the first after load_frames / sync: a clip encoded once,
    libs/encoder.cpp:453-664 -- has no earlier step to overlap its RANSAC + segmentation with, so on a big shard in the two-pass order it
    runs in two chunks and overlaps them with its own second half; steps that follow each other keep whole-shard launches; the one-pass
    orders and SVC_CLIP_TUNE_WHOLE_SHARD_STEPS never chunk.  Same bytes either way; the driver's pair counts make per-step times exact."""
    cfg, ref = encoded
    dev = torch.device("cuda")
    frames = torch.stack(ref.frames_bgr).contiguous()
    for tuning, first, wire in ((0, 2, False), (clipmod.TUNE_MIXED_STEPS, 2, False), (clipmod.TUNE_WHOLE_SHARD_STEPS, 1, False), (0, 1, True)):
        drv = clipmod.Clip(cfg, cfg.frames, schedule=clipmod.PIPELINED, tuning=tuning, wire=wire)
        assert drv.info.chunks_per_step == 1
        drv.load_frames(frames)
        # into an empty pipeline; nothing is known about the clip: planes + quant runs two-pass halves (with SVC_CLIP_TUNE_MIXED_STEPS the second
        # half reads its frames once, blind), wire is one pass anyway
        drv.step(timed=True)
        drv.sync()
        t, sp = drv.stage_times_ms(), drv.stage_pairs()
        for k in ("luma_pyramid", "hbma", "ransac", "segment", "dct_quant"):
            assert t[k][1] == first and sp[k] == drv.info.pairs, (tuning, wire, k, t[k], sp[k])
        blind = 1 if tuning == clipmod.TUNE_MIXED_STEPS and not wire else 0
        if not wire:
            assert drv.policy_info()["chunks_speculated"] == blind and ("type_patch" in t) == bool(blind)
            assert torch.equal(drv.read("coeffs", device=dev).view(ref.coeffs.shape), ref.coeffs)  # the mixed step's planes are the reference's
        drv.step(timed=True)  # again into an empty pipeline (sync above) -- but now the clip's foreground share is known (0.5 %): planes + quant
        drv.step(timed=True)  # reads the clip once, and the one-pass orders keep whole-shard launches; the third step follows the second anyway
        drv.sync()
        t, sp = drv.stage_times_ms(), drv.stage_pairs()
        assert t["hbma"][1] == first + 2 and sp["hbma"] == 3 * drv.info.pairs
        if not wire:
            assert drv.policy_info()["chunks_speculated"] == blind + 2
        out = drv.outputs(device=dev)
        assert torch.equal(out["mv"], ref.mv) and torch.equal(out["block_types"], ref.types) and torch.equal(out["inlier_mask"], ref.mask)
        if not wire:
            assert torch.equal(drv.read("coeffs", device=dev).view(ref.coeffs.shape), ref.coeffs)
        drv.close()


def test_bench_driver_on_one_shard_of_eight(native, encoded):
    """BASELINE config 4's rank 3 of 8 (38 frames of the 300, pipelined, the halo handed over through the transport hook
    as RCCL would deliver it): its outputs are the unsharded clip's rows, bit for bit."""
    import ctypes as C
    cfg, ref = encoded
    dev = torch.device("cuda")
    first, cnt, pairs, first_encoded = clipmod.plan_shard(cfg.frames, 8, 3)
    assert cnt in (37, 38) and pairs == cnt
    drv = clipmod.Clip(cfg, cfg.frames, rank=3, world=8, schedule=clipmod.PIPELINED)
    drv.load_frames(torch.stack(ref.frames_bgr[first:first + cnt]).contiguous())
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    halo_src = ref.pyr.data_ptr() + first * ref.stride  # pyramid slot of clip frame first - 1 (slot = frame + 1)
    calls = []

    def transport(send, recv, nbytes, stream):
        calls.append(nbytes)
        assert hip.hipMemcpyAsync(recv, halo_src, nbytes, 3, stream) == 0
    drv.set_halo_transport(transport)
    for _ in range(5):
        drv.step()
    drv.sync()
    assert calls == [ref.stride] * 5
    g0 = first_encoded - 1
    out = drv.outputs(device=dev)
    for k, want in (("mv", ref.mv), ("min_mad", ref.mad), ("inlier_mask", ref.mask), ("inlier_count", ref.count),
                    ("block_types", ref.types)):
        assert torch.equal(out[k], want[g0:g0 + pairs]), k
    assert out["global_motion"].cpu().numpy().tobytes() == ref.gm[g0:g0 + pairs].cpu().numpy().tobytes()
    assert out["rmse"].cpu().numpy().tobytes() == ref.rmse[g0:g0 + pairs].cpu().numpy().tobytes()
    assert torch.equal(drv.read("coeffs", device=dev).view(pairs, *ref.coeffs.shape[1:]), ref.coeffs[g0:g0 + pairs])
    drv.close()


def test_all_eight_ranks_of_config_4_at_full_size(native, encoded):
    """BASELINE config 4 (the 300-frame 1080p clip over 8 ranks) with EVERY rank run at full size, one after the other on
    this GPU: rank 0 (no halo, 37 pairs from 38 frames), ranks 1-3 (38 frames, 38 pairs), ranks 4-7 (37 / 37) -- each with
    the halo handed over through the transport hook as its predecessor would send it; the concatenation of the ranks'
    outputs is the unsharded clip, every buffer bit for bit, and the shard plan tiles the clip without gap or overlap."""
    import ctypes as C
    cfg, ref = encoded
    dev = torch.device("cuda")
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    world, done, next_frame = 8, 0, 0
    for r in range(world):
        first, cnt, pairs, first_encoded = clipmod.plan_shard(cfg.frames, world, r)
        assert first == next_frame and cnt == (38 if r < 4 else 37) and pairs == (cnt - 1 if r == 0 else cnt)
        next_frame += cnt
        drv = clipmod.Clip(cfg, cfg.frames, rank=r, world=world, schedule=clipmod.PIPELINED)
        assert bool(drv.info.needs_halo) == (r > 0)
        drv.load_frames(torch.stack(ref.frames_bgr[first:first + cnt]).contiguous())
        calls = []

        def transport(send, recv, nbytes, stream, first=first, calls=calls):
            calls.append(nbytes)
            if first > 0:  # what rank r - 1 sends: the pyramid of clip frame first - 1 (slot = frame + 1)
                assert hip.hipMemcpyAsync(recv, ref.pyr.data_ptr() + first * ref.stride, nbytes, 3, stream) == 0
        drv.set_halo_transport(transport)
        for _ in range(4):
            drv.step()
        drv.sync()
        assert calls == [ref.stride] * 4
        g0 = first_encoded - 1
        assert g0 == done
        out = drv.outputs(device=dev)
        for k, want in (("mv", ref.mv), ("min_mad", ref.mad), ("inlier_mask", ref.mask), ("inlier_count", ref.count),
                        ("block_types", ref.types)):
            assert torch.equal(out[k], want[g0:g0 + pairs]), (r, k)
        assert out["global_motion"].cpu().numpy().tobytes() == ref.gm[g0:g0 + pairs].cpu().numpy().tobytes(), r
        assert out["rmse"].cpu().numpy().tobytes() == ref.rmse[g0:g0 + pairs].cpu().numpy().tobytes(), r
        assert torch.equal(drv.read("coeffs", device=dev).view(pairs, *ref.coeffs.shape[1:]), ref.coeffs[g0:g0 + pairs]), r
        done += pairs
        drv.close()
    assert done == cfg.frames - 1 and next_frame == cfg.frames


def test_two_kernels_agree_on_every_pair(native, encoded):
    """The fused lane-per-block kernel and the per-level LDS-staged kernel are independent
    implementations; on all 299 x 8160 blocks they must give identical MVs and min-MADs."""
    cfg, enc = encoded
    t0 = enc.stride
    mv, mad = native.hbma_pairs(enc.pyr[t0:], enc.pyr[2 * t0:], enc.stride, enc.pairs_per_step, cfg.levels,
                                enc.pw, enc.ph, cfg.search_range, flags=native.HBMA_FORCE_WAVE_PER_BLOCK)
    torch.cuda.synchronize()
    assert torch.equal(mv, enc.mv) and torch.equal(mad, enc.mad)
    bound = cfg.r_top * ((1 << cfg.levels) - 1)  # |mv| <= R_top (2^L - 1), SURVEY.md 8a row 5
    assert float(enc.mv.abs().max()) <= bound
    assert bool((enc.mv == enc.mv.round()).all()) and float(enc.mad.min()) >= 0.0


def test_sampled_pairs_against_oracle(oracle, encoded):
    cfg, enc = encoded
    offs = synth.level_offsets(enc.pw, enc.ph, cfg.levels)

    def planes(slot):
        flat = enc.pyr[slot * enc.stride:(slot + 1) * enc.stride].cpu().numpy()
        return [flat[offs[l]:offs[l] + (enc.pw >> l) * (enc.ph >> l)].reshape(enc.ph >> l, enc.pw >> l)
                for l in range(cfg.levels)]
    for p in (0, 150, 298):  # pair p = frames p, p+1 = slots p+1, p+2
        mv, mad = oracle.hbma(planes(p + 1), planes(p + 2), cfg.search_range, 16, 16)
        assert np.array_equal(enc.mv[p].cpu().numpy(), mv) and np.array_equal(enc.mad[p].cpu().numpy(), mad)
    # device pyramids == the oracle's luma + pyrDown (frame 7)
    for l, ref in enumerate(oracle.luma_pyramid(enc.bgr[7].cpu().numpy(), cfg.levels)):
        got = enc.pyr[8 * enc.stride + offs[l]: 8 * enc.stride + offs[l] + ref.size].cpu().numpy().reshape(ref.shape)
        assert np.array_equal(got, ref)


def test_ransac_and_types_consistent(oracle, encoded):
    cfg, enc = encoded
    from oracle.binding import DEFAULT_RANSAC
    for p in (0, 149, 298):
        gm, rmse, inl = oracle.ransac(enc.mv[p].cpu().numpy(), enc.samples[p].cpu().numpy().astype(np.uint32).ravel(),
                                      **DEFAULT_RANSAC)
        assert enc.gm[p].cpu().numpy().tobytes() == gm.tobytes()
        assert np.float32(enc.rmse[p].item()).tobytes() == rmse.tobytes()
        assert np.array_equal(np.flatnonzero(enc.mask[p].cpu().numpy()), inl) and int(enc.count[p]) == len(inl)
    assert bool(((enc.count > 0) & (enc.count <= cfg.blocks)).all())
    # region ids: the oracle's statement of libs/encoder.cpp:507-623 on the same masks / MVs
    for p in (0, 149, 298):
        want = oracle.segment(enc.mask[p].cpu().numpy(), enc.mv[p].cpu().numpy(), enc.mfw, enc.mfh,
                              seed=enc.seg_seed + p)
        assert np.array_equal(enc.types[p].cpu().numpy().astype(np.uint32), want)
    # inliers are always background; a region id > 0 implies "not an inlier" before the morphology only
    assert int(enc.types.min()) == 0


def test_dct_energy_dc_and_quant_properties(native, encoded):
    cfg, enc = encoded
    n = enc.encoded_per_step
    raw = native.dct_frames(enc.bgr[1:], cfg.dct_block)           # un-quantised, all 299 frames
    px = enc.bgr[1:].to(torch.float64)
    e_in = (px * px).sum(dim=(1, 2, 3))
    e_out = (raw.to(torch.float64) ** 2).sum(dim=(1, 2, 3))
    assert float(((e_in - e_out).abs() / e_in).max()) < 1e-6       # Parseval, every frame
    dc = raw[:, :, ::8, ::8].to(torch.float64)
    means = px.permute(0, 3, 1, 2).reshape(n, 3, enc.ph // 8, 8, enc.pw // 8, 8).mean(dim=(3, 5))
    assert float((dc - 8 * means).abs().max()) < 2e-3              # DC = 8 x tile mean
    # fused DCT+quant == quant applied to the raw DCT, bit for bit, on the whole clip
    q = raw.clone()
    native.quant_frames_(q, enc.types, cfg.mv_block, cfg.fg_step, cfg.bg_step)
    torch.cuda.synchronize()
    assert torch.equal(q, enc.coeffs)
    # idempotence and lattice membership
    q2 = q.clone()
    native.quant_frames_(q2, enc.types, cfg.mv_block, cfg.fg_step, cfg.bg_step)
    assert torch.equal(q2, q)
    bg = (enc.types == 0).reshape(n, 1, enc.ph // 16, 1, enc.pw // 16, 1).expand(n, 3, enc.ph // 16, 16, enc.pw // 16, 16)
    bgc = q.reshape(n, 3, enc.ph // 16, 16, enc.pw // 16, 16)[bg]
    assert bool((bgc % cfg.bg_step == 0).all())


def test_wire_mode_agrees_with_the_plain_step(native):
    """The fused record output is a layout choice: the records must be the raw transform's coefficients serialised
    (schedules -- serial, pipelined, hipGraph -- are checked in tests/test_gpu_clip.py)."""
    cfg = configs.C2
    dev = torch.device("cuda")
    n = 9
    clip = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
    pw, ph = cfg.padded
    frames = [synth.pad_frame(clip.frame_bgr(t), pw, ph) for t in range(n)]
    base = pipeline.ClipEncoder(cfg, n, dev)
    base.load_frames(frames)
    base.step()
    wired = pipeline.ClipEncoder(cfg, n, dev, wire=True)
    wired.load_frames(frames)
    wired.step()
    torch.cuda.synchronize()
    assert torch.equal(base.types, wired.types)
    # records carry the RAW coefficients, as the reference's encoder serialises them (libs/encoder.cpp:638-650)
    raw = native.dct_frames(base.bgr[1:].contiguous(), 8)
    want = native.serialize_frames(raw, base.types, pw, ph, 8, 8, base.mfw, base.mfh)
    assert torch.equal(wired.records, want)
    q = raw.clone()
    native.quant_frames_(q, base.types, cfg.mv_block, cfg.fg_step, cfg.bg_step)
    assert torch.equal(q, base.coeffs)  # ... and the planar output is their quantised form


def test_8k_frame_pair(native, oracle):
    """Largest realistic frame: 7680 x 4320 (129 600 MV blocks, 4-level pyramid), one pair."""
    rng = np.random.default_rng(8)
    w, h, levels = 7680, 4320, 4
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2), dtype=np.uint8)
    big = np.kron(base, np.ones((8, 8), np.uint8))  # blocky texture: plenty of exact ties
    t0 = np.ascontiguousarray(big[4:4 + h, 5:5 + w])
    a0 = np.ascontiguousarray(big[6:6 + h, 2:2 + w])
    a0 = (a0.astype(np.int16) + rng.integers(-2, 3, a0.shape)).clip(0, 255).astype(np.uint8)
    t = [np.ascontiguousarray(t0[:: 1 << l, :: 1 << l]) for l in range(levels)]
    a = [np.ascontiguousarray(a0[:: 1 << l, :: 1 << l]) for l in range(levels)]
    mv, mad = native.hbma_host(t, a, 8, 16, 16)
    exp_mv, exp_mad = oracle.hbma16_sse2(t, a, 8)  # the reference's 4-level path, restated
    assert np.array_equal(mv, exp_mv) and np.array_equal(mad, exp_mad)
    bgr = np.stack([t0, a0, t0], axis=-1)
    planes = native.dct_host(np.ascontiguousarray(bgr[:256]), 16)  # a 256-row band of it through the host path
    ref = oracle.dct_frame_f64(np.ascontiguousarray(bgr[:256]), 16, 16)
    assert (np.abs(planes - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
