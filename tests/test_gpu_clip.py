"""svc::ClipEncoder (include/svc/clip_encoder.hpp) through its C handle API: the C++ driver of one rank's
shard.  Checked here: every schedule (serial, pipelined at every depth) gives the bytes of the
stage-by-stage C-ABI calls; a clip cut into shards -- the halo moved by the encoder's own transport hook --
encodes to exactly the unsharded clip; the RCCL entry points bind and move bytes."""
import ctypes as C

import numpy as np
import pytest
import torch

from scalable_video_codec_amd import clip as clipmod
from scalable_video_codec_amd import configs, pipeline, synth

pytestmark = pytest.mark.gpu

CFG = configs.CodecConfig("t-360p-3L-dct8", 41, 640, 360, 11, levels=3, dct_block=8)


def _frames(cfg, n, dev):
    src = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
    pw, ph = cfg.padded
    return torch.stack([synth.pad_frame(src.frame_bgr(t), pw, ph)
                        for t in range(n)]).contiguous()


_hip = None


def _hip_memcpy_async(dst, src, nbytes, stream):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so.7")  # the HIP runtime already in the process
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    rc = _hip.hipMemcpyAsync(dst, src, nbytes, 3, stream)  # hipMemcpyDeviceToDevice
    assert rc == 0, rc


def _reference_outputs(native, cfg, frames):
    """The same clip through the stage-by-stage C-ABI calls of the Python harness (pipeline.ClipEncoder)."""
    n = frames.shape[0]
    ref = pipeline.ClipEncoder(cfg, n, frames.device)
    ref.load_frames(list(frames))
    ref.step()
    torch.cuda.synchronize()
    return ref


def _assert_same(out, ref, coeffs=None):
    assert torch.equal(out["mv"], ref.mv.cpu())
    assert torch.equal(out["min_mad"], ref.mad.cpu())
    assert out["global_motion"].numpy().tobytes() == ref.gm.cpu().numpy().tobytes()
    assert out["rmse"].numpy().tobytes() == ref.rmse.cpu().numpy().tobytes()
    assert torch.equal(out["inlier_mask"], ref.mask.cpu())
    assert torch.equal(out["inlier_count"], ref.count.cpu())
    assert torch.equal(out["block_types"], ref.types.cpu())
    if coeffs is not None:
        assert torch.equal(coeffs.view(ref.coeffs.shape), ref.coeffs.cpu())


@pytest.mark.parametrize("schedule,steps", [(clipmod.SERIAL, 1), (clipmod.SERIAL, 3), (clipmod.PIPELINED, 1), (clipmod.PIPELINED, 5),
                                            (clipmod.PIPELINED, 7)])
def test_schedules_equal_stagewise_calls(native, schedule, steps):
    dev = torch.device("cuda")
    n = CFG.frames
    frames = _frames(CFG, n, dev)
    ref = _reference_outputs(native, CFG, frames)
    enc = clipmod.Clip(CFG, n, schedule=schedule)
    assert (enc.info.frames, enc.info.pairs, enc.info.first_encoded, enc.info.needs_halo) == (n, n - 1, 1, 0)
    enc.load_frames(frames)
    for s in range(steps):
        enc.step(timed=True)
    enc.sync()
    _assert_same(enc.outputs(), ref, enc.read("coeffs"))
    t = enc.stage_times_ms()
    # type_patch (the foreground tiles redone) shows on the steps that speculated: none before a foreground share has arrived (adaptive policy)
    assert set(t) - {"type_patch"} == {"luma_pyramid", "hbma", "ransac", "segment", "dct_quant"}
    assert all(ms > 0 and launches == steps for k, (ms, launches) in t.items() if k != "type_patch")
    assert "type_patch" not in t or 0 < t["type_patch"][1] <= steps
    enc.close()


@pytest.mark.parametrize("lat_depth", [0, 1, 3])
def test_pipeline_state_machine_fuzz(native, lat_depth):
    """Bursts of steps of random length, random timing flags, syncs and output reads at random points, a halo that
    arrives through the transport hook on every step: whatever the interleaving, the newest finished step's outputs are
    the clip's (the double-buffered sets, the deferred join of the second stream and the drain must never mix steps)."""
    dev = torch.device("cuda")
    n = 9
    frames = _frames(CFG, n, dev)
    whole = clipmod.Clip(CFG, n, schedule=clipmod.SERIAL)
    whole.load_frames(frames)
    whole.step()
    whole.sync()
    want = whole.outputs()
    want_coeffs = whole.read("coeffs")
    pyr = whole.read("pyramids", device=dev)
    first, cnt, pairs, first_encoded = clipmod.plan_shard(n, 2, 1)
    enc = clipmod.Clip(CFG, n, rank=1, world=2, lat_depth=lat_depth)
    enc.load_frames(frames[first:first + cnt].contiguous())
    stride = enc.info.pyramid_stride
    enc.set_halo_transport(lambda send, recv, nbytes, stream: _hip_memcpy_async(recv, pyr.data_ptr() + first * stride, nbytes, stream))
    rng = np.random.default_rng(17)
    g0 = first_encoded - 1
    per = 3 * enc.info.padded_w * enc.info.padded_h
    for burst in range(12):
        for _ in range(int(rng.integers(1, 8))):
            enc.step(timed=bool(rng.integers(0, 2)))
        if rng.integers(0, 3) == 0:
            enc.flush()
        out = enc.outputs()  # syncs
        for k in want:
            assert torch.equal(out[k], want[k][g0:g0 + pairs]), (burst, k)
        assert torch.equal(enc.read("coeffs").view(pairs, per), want_coeffs.view(n - 1, per)[g0:g0 + pairs]), burst
        if rng.integers(0, 2):
            enc.reset_timers()
    enc.close()


@pytest.mark.parametrize("lat_depth", [0, 1, 3])
@pytest.mark.parametrize("steps_after", [1, 2, 3, 4, 5, 6])
def test_buffer_sets_never_serve_stale_steps(native, lat_depth, steps_after):
    """The pipelined schedule keeps the small per-step buffers in depth + 2 sets and the pyramids in two; re-encoding the
    same clip leaves every set with identical bytes after a few steps, so a missing join (the transform of step d reading
    region ids before RANSAC + segmentation of d have run, the motion search of step h overwriting a field its reader has
    not finished with) would hand back stale bytes that EQUAL the right ones.  Here the clip changes between bursts:
    after a burst on clip A every set holds A's data; then 1 .. nsets + 1 steps of clip B must give B's outputs exactly
    (a stale set would surface A's motion field, region ids or coefficients)."""
    dev = torch.device("cuda")
    n = 9
    cfg_b = configs.CodecConfig("t-360p-3L-dct8-b", 77, 640, 360, n, levels=3, dct_block=8)
    fa, fb = _frames(CFG, n, dev), _frames(cfg_b, n, dev)
    assert not torch.equal(fa, fb)
    want = {}
    for name, f in (("a", fa), ("b", fb)):
        s = clipmod.Clip(CFG, n, schedule=clipmod.SERIAL)
        s.load_frames(f)
        s.step()
        s.sync()
        want[name] = (s.outputs(), s.read("coeffs"))
        s.close()
    assert not torch.equal(want["a"][0]["mv"], want["b"][0]["mv"]) and not torch.equal(want["a"][0]["block_types"], want["b"][0]["block_types"])
    enc = clipmod.Clip(CFG, n, schedule=clipmod.PIPELINED, lat_depth=lat_depth)
    enc.load_frames(fa)
    for _ in range(7):
        enc.step()
    enc.load_frames(fb)  # drains the pipeline; every buffer set now holds clip A
    for _ in range(steps_after):
        enc.step()
    out = enc.outputs()  # syncs
    for k in want["b"][0]:
        assert torch.equal(out[k], want["b"][0][k]), k
    assert torch.equal(enc.read("coeffs"), want["b"][1])
    enc.load_frames(fa)  # and back, mid-stream (no drain before the switch other than load_frames' own)
    enc.step()
    enc.step()
    out = enc.outputs()
    for k in want["a"][0]:
        assert torch.equal(out[k], want["a"][0][k]), k
    assert torch.equal(enc.read("coeffs"), want["a"][1])
    enc.close()


@pytest.mark.parametrize("form", ["two_passes", "wire", "speculative", "two_passes+search_after_transform", "speculative+search_after_transform",
                                  "wire+search_after_transform"])
@pytest.mark.parametrize("chunk_pairs", [1, 2, 3, 4, 7])
def test_chunked_steps_equal_whole_steps(native, form, chunk_pairs):
    """Round 6: a pipelined step at one rank is cut into chunks of frame pairs and the pipeline runs over the CHUNKS (RANSAC + segmentation
    of a chunk beside the motion search of the next and the transform of the previous one: a clip encoded once no longer pays the
    latency-bound stages end to end).  Every chunk size -- a pair per chunk, chunks that do not divide the clip, one chunk -- at every
    pipeline depth, in all three output forms, in bursts of steps over clips that CHANGE between bursts (a chunk served from a stale set,
    a join event reused too early, a coefficient set rewritten before its foreground tiles were redone would surface the other clip's
    bytes): the outputs of the serial whole-shard step, bit for bit.  Launch counts: every stage once per chunk and step."""
    dev = torch.device("cuda")
    n = 9
    cfg_b = configs.CodecConfig("t-360p-3L-dct8-b", 77, 640, 360, n, levels=3, dct_block=8)
    fa, fb = _frames(CFG, n, dev), _frames(cfg_b, n, dev)
    form, _, order = form.partition("+")  # "+search_after_transform": the A/B order of the main stream's kernels (same bytes)
    wire = form == "wire"
    buf = "records" if wire else "coeffs"
    tuning = {"two_passes": clipmod.TUNE_TWO_BGR_PASSES, "wire": 0, "speculative": clipmod.TUNE_ALWAYS_SPECULATE}[form] | \
        (clipmod.TUNE_SEARCH_AFTER_TRANSFORM if order else 0)
    want = {}
    for name, f in (("a", fa), ("b", fb)):
        s = clipmod.Clip(CFG, n, schedule=clipmod.SERIAL, wire=wire, tuning=clipmod.TUNE_TWO_BGR_PASSES, ransac=dict(inlier_thresh=1.5))
        s.load_frames(f)
        s.step()
        s.sync()
        want[name] = (s.outputs(), s.read(buf), s.read("pyramids"))
        s.close()
    assert want["a"][0]["block_types"].count_nonzero() > 0 and not torch.equal(want["a"][1], want["b"][1])
    chunks = -(-(n - 1) // chunk_pairs)
    for lat_depth in (0, 1, 3):
        enc = clipmod.Clip(CFG, n, schedule=clipmod.PIPELINED, wire=wire, lat_depth=lat_depth, tuning=tuning, chunk_pairs=chunk_pairs,
                           ransac=dict(inlier_thresh=1.5))
        assert enc.info.chunks_per_step == chunks
        steps = 0
        for burst, (name, f, k) in enumerate((("a", fa, 1), ("b", fb, 1), ("a", fa, 6), ("b", fb, 2), ("a", fa, 3), ("b", fb, 5))):
            enc.load_frames(f)
            for _ in range(k):
                enc.step(timed=True)
            steps += k
            out = enc.outputs()  # syncs
            for key in want[name][0]:
                assert torch.equal(out[key], want[name][0][key]), (lat_depth, burst, key)
            assert torch.equal(enc.read(buf), want[name][1]), (lat_depth, burst)
            assert torch.equal(enc.read("pyramids")[enc.info.pyramid_stride:], want[name][2][enc.info.pyramid_stride:]), (lat_depth, burst)
        t = enc.stage_times_ms()
        assert all(launches == steps * chunks for _, launches in t.values()), t
        assert ("type_patch" in t) == (form != "two_passes")
        if form == "speculative":
            assert enc.policy_info()["chunks_speculated"] == enc.policy_info()["chunks_decided"] == steps * chunks
        enc.close()


@pytest.mark.parametrize("n", [3, 6, 9])
def test_a_step_that_knows_nothing_takes_the_mixed_form(native, n):
    """Round 6, SVC_CLIP_TUNE_MIXED_STEPS (an A/B switch, off by default: it loses where the foreground share is high): a step into an EMPTY
    pipeline with nothing known about the clip's foreground share (a clip encoded once) runs its first half in the two-pass order and its
    second half reading its frames once, blind (ClipEncoder::Step; from 400 M pixels x frames, here forced by SVC_CLIP_TUNE_IDLE_RULE_ANY_SIZE).  Same bytes as the serial two-pass
    step, for odd and even pair counts, clips that change between steps, every pipeline depth; the next step (share known by then: the
    synthetic clip's is far above 2 %) is two passes again, in two halves when it finds the pipeline empty."""
    dev = torch.device("cuda")
    cfg_b = configs.CodecConfig("t-360p-3L-dct8-b", 77, 640, 360, n, levels=3, dct_block=8)
    fa, fb = _frames(CFG, n, dev), _frames(cfg_b, n, dev)
    want = {}
    for name, f in (("a", fa), ("b", fb)):
        s = clipmod.Clip(CFG, n, schedule=clipmod.SERIAL, tuning=clipmod.TUNE_TWO_BGR_PASSES, ransac=dict(inlier_thresh=1.5))
        s.load_frames(f)
        s.step()
        s.sync()
        want[name] = (s.outputs(), s.read("coeffs"), s.read("pyramids"))
        s.close()
    assert want["a"][0]["block_types"].count_nonzero() > 0
    for lat_depth in (0, 1, 3):
        for extra in (clipmod.TUNE_MIXED_STEPS, clipmod.TUNE_MIXED_STEPS | clipmod.TUNE_SEARCH_AFTER_TRANSFORM, 0):
            enc = clipmod.Clip(CFG, n, schedule=clipmod.PIPELINED, lat_depth=lat_depth, tuning=clipmod.TUNE_IDLE_RULE_ANY_SIZE | extra,
                               ransac=dict(inlier_thresh=1.5))
            assert enc.info.chunks_per_step == 1
            mixed = bool(extra & clipmod.TUNE_MIXED_STEPS)
            for burst, (name, f) in enumerate((("a", fa), ("b", fb), ("a", fa))):
                enc.load_frames(f)  # voids the policy: the step knows nothing
                p0 = enc.policy_info()
                enc.step(timed=True)
                out = enc.outputs()  # syncs
                p1 = enc.policy_info()
                assert p1["chunks_decided"] - p0["chunks_decided"] == 2 and p1["chunks_speculated"] - p0["chunks_speculated"] == (1 if mixed else 0)
                for key in want[name][0]:
                    assert torch.equal(out[key], want[name][0][key]), (lat_depth, extra, burst, key)
                assert torch.equal(enc.read("coeffs"), want[name][1]), (lat_depth, extra, burst)
                assert torch.equal(enc.read("pyramids")[enc.info.pyramid_stride:], want[name][2][enc.info.pyramid_stride:]), (lat_depth, extra, burst)
            t = enc.stage_times_ms()
            assert t["hbma"][1] == 6 and t["dct_quant"][1] == 6 and (("type_patch" in t) == mixed), t
            assert enc.output_sets() == 1  # the mixed form needs no extra coefficient sets
            # the share is known now: a further step into the empty pipeline follows the policy (high share: two passes in two halves, then
            # whole-shard steps behind it; low: whole-shard one-pass steps)
            p0 = enc.policy_info()
            assert p0["foreground_share"] >= 0
            for _ in range(3):
                enc.step()
            enc.sync()
            p1 = enc.policy_info()
            if p0["foreground_share"] > 0.02:
                assert p1["chunks_speculated"] == p0["chunks_speculated"] and p1["chunks_decided"] - p0["chunks_decided"] == 4
            else:
                assert p1["chunks_speculated"] - p0["chunks_speculated"] == p1["chunks_decided"] - p0["chunks_decided"] == 3
            assert torch.equal(enc.read("coeffs"), want["a"][1])
            enc.close()


def test_the_foreground_prior_survives_a_load_only_on_request(native):
    """LoadFrames voids what the speculation policy knew (other frames: the first step over them is two passes) unless the caller states
    that the clips are consecutive pieces of one stream (SVC_CLIP_KEEP_FOREGROUND_PRIOR): then the step right after a load speculates
    on the last piece's share.  svc_clip_reset_policy voids it by hand.  Bytes never depend on any of it."""
    dev = torch.device("cuda")
    big = configs.CodecConfig("t-1080p-3L", 43, 1920, 1080, 26, levels=3, dct_block=8)  # 25 encoded frames x 1920 x 1088 = 52 M: may speculate
    frames = _frames(big, big.frames, dev)
    ref = None
    for keep in (False, True):
        enc = clipmod.Clip(big, big.frames, tuning=clipmod.KEEP_FOREGROUND_PRIOR if keep else 0)
        enc.load_frames(frames)
        for _ in range(2):  # the host enqueues faster than the GPU measures: the first burst learns the share, the second acts on it
            for _ in range(3):
                enc.step()
            enc.sync()
        warm = enc.policy_info()
        assert 0 <= warm["foreground_share"] <= 0.02 and warm["chunks_speculated"] > 0
        enc.load_frames(frames)          # "the next piece of the stream"
        enc.step()
        enc.sync()
        after = enc.policy_info()
        spec_now = after["chunks_speculated"] - warm["chunks_speculated"]
        assert (spec_now == enc.info.chunks_per_step) if keep else (spec_now == 0), (keep, warm, after)
        enc.reset_policy()
        assert enc.policy_info()["foreground_share"] == -1.0
        enc.step()
        enc.sync()
        assert enc.policy_info()["chunks_speculated"] == after["chunks_speculated"]  # nothing known: two passes
        got = (enc.outputs(), enc.read("coeffs"))
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(got[0][k], ref[0][k]) for k in ref[0]) and torch.equal(got[1], ref[1])
        enc.close()


def test_wire_and_no_segmentation(native):
    dev = torch.device("cuda")
    n = 6
    frames = _frames(CFG, n, dev)
    ref = pipeline.ClipEncoder(CFG, n, dev, wire=True, segmentation=False)
    ref.load_frames(list(frames))
    ref.step()
    torch.cuda.synchronize()
    enc = clipmod.Clip(CFG, n, wire=True, segmentation=False)
    enc.load_frames(frames)
    enc.step()
    enc.step()
    enc.sync()
    assert torch.equal(enc.read("block_types").view(n - 1, -1), ref.types.cpu())
    assert torch.equal(enc.read("records").view(ref.records.shape), ref.records.cpu())


@pytest.mark.parametrize("block,wire", [((16, 8), False), ((4, 4), True), ((8, 16), True)])
def test_general_transform_blocks(native, block, wire):
    """Transform blocks beyond 8x8 / 16x16 through the driver: planes from the general kernel, records from
    Dct + SerializeEncodedFrame (two C-ABI calls) -- equal to the stage-by-stage calls."""
    dev = torch.device("cuda")
    n = 4
    frames = _frames(CFG, n, dev)
    enc = clipmod.Clip(CFG, n, wire=wire, dct_block=block)
    enc.load_frames(frames)
    enc.step()
    enc.sync()
    types = enc.read("block_types", device=dev).view(n - 1, -1)
    i = enc.info
    if wire:
        planes = native.dct_frames(frames[1:].contiguous(), block)
        want = native.serialize_frames(planes, types, i.padded_w, i.padded_h, block[0], block[1], i.mv_field_w, i.mv_field_h)
        assert i.record_bytes == want.shape[1]
        assert torch.equal(enc.read("records", device=dev).view(n - 1, -1), want)
    else:
        want = native.dct_quant_frames(frames[1:].contiguous(), block, types, 16, CFG.fg_step, CFG.bg_step)
        assert torch.equal(enc.read("coeffs", device=dev).view(want.shape), want)


@pytest.mark.parametrize("world,total,schedule", [(2, 11, clipmod.SERIAL), (3, 11, clipmod.PIPELINED), (4, 5, clipmod.PIPELINED),
                                                  (4, 4, clipmod.SERIAL), (8, 11, clipmod.PIPELINED)])
def test_sharded_clip_equals_unsharded(native, world, total, schedule):
    """All ranks of a sharded clip in ONE process: rank r's halo transport copies rank r - 1's last pyramid (the
    bytes RCCL would deliver).  Concatenated shard outputs == the unsharded clip's, bit for bit -- uneven shards,
    a rank with a single frame, rank 0 with a single frame (no pair at all) included."""
    _check_sharding(CFG, world, total, schedule)


def test_random_shardings_equal_the_unsharded_clip(native):
    """Seeded random clip lengths, rank counts (2 ... 8), schedules and configurations (sizes, levels, MV and transform blocks): every
    sharding of every clip gives the unsharded clip's outputs, bit for bit."""
    rng = np.random.default_rng(808)
    for i in range(14):
        levels = int(rng.integers(1, 5))
        f = 1 << (levels - 1)
        mv_block = int(rng.choice([b for b in (8, 16, 32) if b >= 2 * f]))
        dct_block = int(rng.choice([b for b in (4, 8, 16) if mv_block % b == 0]))
        w, h = int(rng.integers(2 * mv_block + 1, 300)), int(rng.integers(2 * mv_block + 1, 220))
        world = int(rng.integers(2, 9))
        total = int(rng.integers(world, 30))
        cfg = configs.CodecConfig(f"t-shard{i}-{w}x{h}-{levels}L", 60 + i, w, h, total, levels=levels, mv_block=mv_block,
                                  search_range=int(rng.choice([r for r in (4, 8, 16) if r >= f])), dct_block=dct_block)
        _check_sharding(cfg, world, total, int(rng.choice([clipmod.SERIAL, clipmod.PIPELINED])))


def _check_sharding(CFG, world, total, schedule):  # noqa: N803 (the body below was written against the module's CFG)
    dev = torch.device("cuda")
    frames = _frames(CFG, total, dev)
    whole = clipmod.Clip(CFG, total, schedule=clipmod.SERIAL)
    whole.load_frames(frames)
    whole.step()
    whole.sync()
    want = whole.outputs()
    per = 3 * whole.info.padded_w * whole.info.padded_h
    want_coeffs = whole.read("coeffs").view(total - 1, per)
    steps = 1 if schedule == clipmod.SERIAL else 4
    got = {k: [] for k in want}
    got_coeffs = []
    prev = None
    seen_pairs = 0
    for r in range(world):
        enc = clipmod.Clip(CFG, total, rank=r, world=world, schedule=schedule)
        i = enc.info
        assert (i.first_frame, i.frames, i.pairs, i.first_encoded) == clipmod.plan_shard(total, world, r)
        assert i.first_encoded - 1 == seen_pairs
        enc.load_frames(frames[i.first_frame:i.first_frame + i.frames].contiguous())
        calls = []

        def transport(send, recv, nbytes, stream, r=r, prev=prev, calls=calls, stride=i.pyramid_stride):
            assert nbytes == stride and send
            calls.append(r)
            if r > 0:  # what rank r - 1 sends: its last pyramid, finished and synced below
                src, have = C.c_void_p(), C.c_uint64()
                clipmod._check(clipmod.load().svc_clip_output(prev._h, clipmod.BUFFERS["pyramids"][0], C.byref(src), C.byref(have)))
                _hip_memcpy_async(recv, src.value + prev.info.frames * stride, nbytes, stream)
        enc.set_halo_transport(transport)
        for _ in range(steps):
            enc.step()
        enc.sync()
        assert len(calls) == steps
        o = enc.outputs()
        for k in want:
            got[k].append(o[k])
        got_coeffs.append(enc.read("coeffs").view(i.pairs, per))
        seen_pairs += i.pairs
        prev = enc
    assert seen_pairs == total - 1
    for k in want:
        assert torch.equal(torch.cat(got[k]), want[k]), k
    assert torch.equal(torch.cat(got_coeffs), want_coeffs)


def test_world_needs_a_transport(native):
    enc = clipmod.Clip(CFG, 6, rank=1, world=2)
    enc.load_frames(_frames(CFG, 6, torch.device("cuda"))[3:6].contiguous())
    with pytest.raises(clipmod.ClipError, match="SetComm"):
        enc.step()


def test_rccl_entry_points_move_bytes(native):
    """One rank, cyclic shift = a send to and a receive from itself through RCCL: the library binds at run time,
    a communicator comes up on this GPU, the group enqueues on the caller's stream."""
    dev = torch.device("cuda")
    uid = clipmod.comm_unique_id()
    assert len(uid) == clipmod.COMM_ID_BYTES and any(uid)
    comm = clipmod.comm_create(uid, 0, 1)
    try:
        n = 2_741_760  # one C3 pyramid
        send = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
        recv = torch.zeros(n, dtype=torch.uint8, device=dev)
        clipmod.halo_shift(comm, send, recv, n, 0, 1)  # non-cyclic, single rank: no neighbour, nothing moves
        torch.cuda.synchronize()
        assert int(recv.max()) == 0
        clipmod.halo_shift(comm, send, recv, n, 0, 1, cyclic=True)
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
    finally:
        clipmod.comm_destroy(comm)


def test_invalid_configurations(native):
    with pytest.raises(clipmod.ClipError):
        clipmod.Clip(CFG, 1)  # a clip needs two frames
    with pytest.raises(clipmod.ClipError):
        clipmod.Clip(CFG, 3, rank=0, world=4)  # fewer frames than ranks
    with pytest.raises(clipmod.ClipError):
        clipmod.Clip(CFG, 8, rank=2, world=2)


@pytest.mark.parametrize("w,h", [(720, 576), (176, 144)])
def test_default_four_level_build_on_widths_16_mod_32(native, oracle, w, h):
    """PAL (720 x 576) and QCIF (176 x 144) with the reference's DEFAULT configuration (4 levels, 16 x 16, range 8): the level-3
    plane is 90 / 22 pixels wide, not a whole number of dwords -- the pyramid kernels refused that until round 4.  The C++ driver's
    pyramids and motion field (the general per-level kernel serves this shape) against the oracle, pipelined schedule."""
    cfg = configs.CodecConfig(f"t-{w}x{h}-4L-dct8", 5, w, h, 5, levels=4, dct_block=8)
    dev = torch.device("cuda")
    frames = _frames(cfg, cfg.frames, dev)
    enc = clipmod.Clip(cfg, cfg.frames, schedule=clipmod.PIPELINED)
    enc.load_frames(frames)
    for _ in range(4):
        enc.step()
    enc.sync()
    out = enc.outputs()
    pw, ph = cfg.padded
    pyrs = [oracle.luma_pyramid(frames[t].cpu().numpy(), 4) for t in range(cfg.frames)]
    pyr = enc.read("pyramids")
    stride = enc.info.pyramid_stride
    for t in (0, cfg.frames - 1):  # slot = frame + 1 (slot 0 is the halo)
        off = 0
        for l, p in enumerate(pyrs[t]):
            got = pyr[(t + 1) * stride + off:(t + 1) * stride + off + p.size].numpy().reshape(p.shape)
            assert np.array_equal(got, p), (t, l)
            off += p.size
    for p in range(cfg.frames - 1):
        mv, mad = oracle.hbma16_sse2(pyrs[p], pyrs[p + 1], 8)
        assert np.array_equal(out["mv"][p].numpy(), mv) and np.array_equal(out["min_mad"][p].numpy(), mad), p
    ref = _reference_outputs(native, cfg, frames)
    _assert_same(out, ref, enc.read("coeffs"))
    enc.close()


def test_eight_pixel_mv_blocks_on_a_frame_not_16_wide(native, oracle):
    """--mv-block-w 8 --mv-block-h 8 pads a 360 x 200 frame to itself (multiples of 8 and of 2^(L-1) = 4): not a whole number of the
    luma kernel's 16-pixel segments wide.  The C++ driver end to end (general-width luma, bytewise level 2, lane-per-block search of
    8 x 8 blocks at 3 levels, general transform kernel for 4 x 4 blocks) against the stage-by-stage calls and the oracle."""
    cfg = configs.CodecConfig("t-360x200-8x8-3L-dct4", 9, 360, 200, 5, levels=3, mv_block=8, dct_block=4)
    assert cfg.padded == (360, 200)
    dev = torch.device("cuda")
    frames = _frames(cfg, cfg.frames, dev)
    enc = clipmod.Clip(cfg, cfg.frames, schedule=clipmod.PIPELINED)
    enc.load_frames(frames)
    for _ in range(4):
        enc.step()
    enc.sync()
    out = enc.outputs()
    pyrs = [oracle.luma_pyramid(frames[t].cpu().numpy(), 3) for t in range(cfg.frames)]
    for p in range(cfg.frames - 1):
        mv, mad = oracle.hbma(pyrs[p], pyrs[p + 1], cfg.search_range, 8, 8)
        assert np.array_equal(out["mv"][p].numpy(), mv) and np.array_equal(out["min_mad"][p].numpy(), mad), p
    ref = _reference_outputs(native, cfg, frames)
    _assert_same(out, ref, enc.read("coeffs"))
    want = oracle.dct_frame_f64(frames[2].cpu().numpy(), 4, 4)
    got = oracle.quant_frame(want.astype(np.float32), 8, 8, out["block_types"][1].numpy().astype(np.uint32), cfg.fg_step, cfg.bg_step)
    c = enc.read("coeffs").view(cfg.frames - 1, 3, 200, 360)[1].numpy()
    assert np.mean(c == got) > 0.999  # away from rounding boundaries the quantised values agree exactly
    enc.close()


@pytest.mark.parametrize("wire", [True, False])
@pytest.mark.parametrize("dct", [8, 16])
@pytest.mark.parametrize("schedule,steps,rank", [(clipmod.SERIAL, 2, 0), (clipmod.PIPELINED, 1, 0), (clipmod.PIPELINED, 6, 0), (clipmod.PIPELINED, 7, 1)])
def test_one_bgr_pass_equals_two_passes(native, wire, dct, schedule, steps, rank):
    """Both output forms read the BGR clip once per step by default: the transform runs at the front of the step and leaves the luma
    plane; what needs the region ids follows once the segmentation has them -- the type words of the records (wire), the foreground
    tiles' quantisation (planes: every tile is quantised as background first, the tiles of foreground MV blocks are redone with fg_step).
    SVC_CLIP_TUNE_TWO_BGR_PASSES keeps the two-pass order.  Same bytes in every buffer, on an unsharded clip (frame 0 tracked only: its
    pyramid comes from the plain luma kernel) and on a shard behind a halo."""
    dev = torch.device("cuda")
    n = 9
    cfg = configs.CodecConfig("t-360p-3L", 41, 640, 360, n, levels=3, dct_block=dct)
    frames = _frames(cfg, n, dev)
    world = 2 if rank else 1
    first, cnt, pairs, _ = clipmod.plan_shard(n, world, rank)
    halo_src = None
    if rank:
        whole = clipmod.Clip(cfg, n, schedule=clipmod.SERIAL)
        whole.load_frames(frames)
        whole.step()
        whole.sync()
        halo_src = whole.read("pyramids", device=dev)
    got = {}
    for name, tuning in (("two", clipmod.TUNE_TWO_BGR_PASSES), ("one", 0 if wire else clipmod.TUNE_ALWAYS_SPECULATE)):
        # a tight inlier threshold: the moving rectangles become foreground regions (at 7.5 px this clip has none, and the type words would
        # all stay 0)
        enc = clipmod.Clip(cfg, n, rank=rank, world=world, schedule=schedule, wire=wire, tuning=tuning, ransac=dict(inlier_thresh=1.5))
        enc.load_frames(frames[first:first + cnt].contiguous())
        if rank:
            stride = enc.info.pyramid_stride
            enc.set_halo_transport(lambda send, recv, nbytes, stream: _hip_memcpy_async(recv, halo_src.data_ptr() + first * stride, nbytes, stream))
        for _ in range(steps):
            enc.step(timed=True)
        enc.sync()
        got[name] = (enc.outputs(), enc.read("records" if wire else "coeffs"), enc.read("pyramids"), enc.stage_times_ms())
        enc.close()
    for k in got["two"][0]:
        assert torch.equal(got["one"][0][k], got["two"][0][k]), k
    assert got["two"][0]["block_types"].count_nonzero() > 0  # there are foreground ids to store
    assert torch.equal(got["one"][1], got["two"][1])
    assert torch.equal(got["one"][2], got["two"][2])
    assert "type_patch" in got["one"][3] and "type_patch" not in got["two"][3]
    assert all(launches == steps for _, launches in got["one"][3].values())


@pytest.mark.parametrize("wire", [True, False])
def test_one_bgr_pass_output_sets_never_serve_stale_steps(native, wire):
    """A step's records / coefficients are written at its front and completed (type words / foreground tiles) 2 + depth iterations later,
    in a set of their own; the clip changes between bursts, so the output of a stale set -- or region ids finished into another step's
    output -- would show."""
    dev = torch.device("cuda")
    n = 7
    cfg_b = configs.CodecConfig("t-360p-3L-dct8-b", 77, 640, 360, n, levels=3, dct_block=8)
    fa, fb = _frames(CFG, n, dev), _frames(cfg_b, n, dev)
    buf = "records" if wire else "coeffs"
    want = {}
    for name, f in (("a", fa), ("b", fb)):
        s = clipmod.Clip(CFG, n, schedule=clipmod.SERIAL, wire=wire, tuning=clipmod.TUNE_TWO_BGR_PASSES, ransac=dict(inlier_thresh=1.5))
        s.load_frames(f)
        s.step()
        s.sync()
        want[name] = s.read(buf)
        s.close()
    assert not torch.equal(want["a"], want["b"])
    for lat_depth in (0, 1, 3):
        enc = clipmod.Clip(CFG, n, schedule=clipmod.PIPELINED, wire=wire, lat_depth=lat_depth, ransac=dict(inlier_thresh=1.5),
                           tuning=0 if wire else clipmod.TUNE_ALWAYS_SPECULATE)
        for burst, (name, f, k) in enumerate((("a", fa, 8), ("b", fb, 1), ("a", fa, 2), ("b", fb, 3), ("a", fa, 5), ("b", fb, 6))):
            enc.load_frames(f)
            for _ in range(k):
                enc.step()
            assert torch.equal(enc.read(buf), want[name]), (lat_depth, burst)
        enc.close()


def test_speculation_follows_the_foreground_share(native):
    """Planes + quant: a step speculates (transform at the front, every tile as background, foreground tiles redone) only if the newest
    foreground share that has ARRIVED is at most 2 % (none has before the first steps are through) and the shard is big enough to pay
    (50 M pixels x frames).  A clip without foreground switches over once its first measurement is in; a clip with a fifth of its blocks
    foreground never does; a small shard never does; the bytes never depend on it."""
    dev = torch.device("cuda")
    big = configs.CodecConfig("t-1080p-3L", 43, 1920, 1080, 26, levels=3, dct_block=8)  # 25 encoded frames x 1920 x 1088 = 52 M
    other = configs.CodecConfig("t-1080p-3L-other", 44, 1920, 1080, 26, levels=3, dct_block=8)
    calm = _frames(big, big.frames, dev)
    busy = calm.clone()
    busy[1::2] = _frames(other, big.frames, dev)[1::2]  # every second frame from another scene: a clip of scene cuts
    small = _frames(CFG, 7, dev)
    outs = {}
    for name, cfg, frames, tuning in (("calm", big, calm, 0), ("calm_plain", big, calm, clipmod.TUNE_TWO_BGR_PASSES), ("busy", big, busy, 0),
                                      ("busy_always", big, busy, clipmod.TUNE_ALWAYS_SPECULATE), ("small", CFG, small, 0)):
        enc = clipmod.Clip(cfg, frames.shape[0], tuning=tuning)
        enc.load_frames(frames)
        for _ in range(3):
            for _ in range(5):
                enc.step(timed=True)
            enc.sync()  # every measurement enqueued so far has arrived
        t = enc.stage_times_ms()
        outs[name] = (enc.outputs(), enc.read("coeffs"), t.get("type_patch", (0.0, 0))[1], t["dct_quant"][1])
        enc.close()
    share = lambda o: float((o[0]["block_types"] != 0).float().mean())  # noqa: E731
    assert share(outs["calm"]) <= 0.02 < 0.1 < share(outs["busy"]) and share(outs["small"]) <= 0.02
    assert outs["calm"][3] == outs["busy"][3] == outs["small"][3] == 15
    assert outs["calm"][2] >= 9 and outs["calm_plain"][2] == 0  # the steps after the first sync speculated
    assert outs["busy"][2] == 0 and outs["busy_always"][2] == 15 and outs["small"][2] == 0
    for a, b in (("calm", "calm_plain"), ("busy", "busy_always")):
        for k in outs[a][0]:
            assert torch.equal(outs[a][0][k], outs[b][0][k]), (a, k)
        assert torch.equal(outs[a][1], outs[b][1]), a


@pytest.mark.parametrize("chunk_pairs", [0, 1, 2, 3, 7])
def test_the_policy_may_flip_at_every_chunk(native, chunk_pairs):
    """SVC_CLIP_TUNE_RANDOM_POLICY (a test switch): the speculation policy answers yes / no by a fixed pseudo-random sequence over the chunk
    launches -- flips inside steps, across steps, across loads.  Found by tests/helpers/driver_fuzz.py: with chunked steps the FIRST speculation of a shard
    (which allocates the extra coefficient sets and starts their rotation) could fall on a later chunk of a step whose first chunk had
    already written set 0, leaving the step's planes in two sets; the rotation now starts with a step's first chunk.  Bytes of the serial
    two-pass step in every burst, whatever flipped where; with whole-shard steps the idle-pipeline rule (forced at this size) cuts the
    steps that find the pipeline empty in two, so flips fall inside those too."""
    dev = torch.device("cuda")
    n = 9
    cfg_b = configs.CodecConfig("t-360p-3L-dct8-b", 77, 640, 360, n, levels=3, dct_block=8)
    fa, fb = _frames(CFG, n, dev), _frames(cfg_b, n, dev)
    want = {}
    for name, f in (("a", fa), ("b", fb)):
        s = clipmod.Clip(CFG, n, schedule=clipmod.SERIAL, tuning=clipmod.TUNE_TWO_BGR_PASSES, ransac=dict(inlier_thresh=1.5))
        s.load_frames(f)
        s.step()
        s.sync()
        want[name] = (s.outputs(), s.read("coeffs"), s.read("pyramids"))
        s.close()
    for lat_depth in (0, 1, 3):
        for extra in (0, clipmod.TUNE_SEARCH_AFTER_TRANSFORM):
            enc = clipmod.Clip(CFG, n, schedule=clipmod.PIPELINED, lat_depth=lat_depth, chunk_pairs=chunk_pairs,
                               tuning=clipmod.TUNE_RANDOM_POLICY | clipmod.TUNE_IDLE_RULE_ANY_SIZE | extra, ransac=dict(inlier_thresh=1.5))
            # (the switch's sequence: step 0 never, step 1 on every chunk but its first, then pseudo-random -- the second burst is read right
            # after the step in which the shard speculates for the first time, inside that step where it has more than one chunk)
            for burst, (name, f, k, flush) in enumerate((("a", fa, 1, False), ("b", fb, 1, False), ("a", fa, 5, False), ("b", fb, 2, True),
                                                         ("a", fa, 3, True), ("b", fb, 4, False))):
                enc.load_frames(f)
                for i in range(k):
                    enc.step()
                    if flush and i == 0:
                        enc.flush()  # every stage enqueued, nothing waited for: the next step finds the pipeline "empty"
                out = enc.outputs()  # syncs
                for key in want[name][0]:
                    assert torch.equal(out[key], want[name][0][key]), (lat_depth, extra, burst, key)
                assert torch.equal(enc.read("coeffs"), want[name][1]), (lat_depth, extra, burst)
                assert torch.equal(enc.read("pyramids")[enc.info.pyramid_stride:], want[name][2][enc.info.pyramid_stride:]), (lat_depth, extra, burst)
            p = enc.policy_info()
            assert 0 < p["chunks_speculated"] < p["chunks_decided"]
            enc.close()


@pytest.mark.parametrize("form", ["two_passes", "wire", "speculative", "policy"])
def test_a_stream_of_clips_encoded_where_they_are(native, form):
    """svc_clip_step_frames (round 6): a stream of clips, each encoded ONCE (libs/encoder.cpp:453-664 never looks at a clip twice), stepped
    where the caller has them in device memory -- no copy into the resident buffer and no drain of the pipeline between clips (load_frames
    synchronises, so load / step / load / step runs every step into an empty pipeline).  Four clips in rotation without a sync in between,
    mixed with steps over the resident buffer: after any sync the outputs are those of the LAST clip stepped, bit for bit (the serial
    two-pass encoder's); wait_step(s) returns once step s's frames are free -- they are then overwritten with noise while later steps are
    still in flight, and nothing changes."""
    dev = torch.device("cuda")
    n = 9
    wire = form == "wire"
    buf = "records" if wire else "coeffs"
    tuning = {"two_passes": clipmod.TUNE_TWO_BGR_PASSES, "wire": 0, "speculative": clipmod.TUNE_ALWAYS_SPECULATE,
              "policy": clipmod.TUNE_IDLE_RULE_ANY_SIZE | clipmod.TUNE_RANDOM_POLICY}[form]
    cfgs = [CFG] + [configs.CodecConfig(f"t-360p-3L-dct8-{k}", 70 + k, 640, 360, n, levels=3, dct_block=8) for k in range(3)]
    clips = [_frames(c, n, dev) for c in cfgs]
    want = []
    for f in clips:
        s = clipmod.Clip(CFG, n, schedule=clipmod.SERIAL, wire=wire, tuning=clipmod.TUNE_TWO_BGR_PASSES, ransac=dict(inlier_thresh=1.5))
        s.load_frames(f)
        s.step()
        s.sync()
        want.append((s.outputs(), s.read(buf), s.read("pyramids")))
        s.close()

    def check(enc, k, tag):
        out = enc.outputs()  # syncs
        for key in want[k][0]:
            assert torch.equal(out[key], want[k][0][key]), (tag, key)
        assert torch.equal(enc.read(buf), want[k][1]), tag
        assert torch.equal(enc.read("pyramids")[enc.info.pyramid_stride:], want[k][2][enc.info.pyramid_stride:]), tag
    for schedule, lat_depth, chunk_pairs in ((clipmod.SERIAL, 0, 0), (clipmod.PIPELINED, 0, 0), (clipmod.PIPELINED, 1, 3), (clipmod.PIPELINED, 3, 0)):
        enc = clipmod.Clip(CFG, n, schedule=schedule, wire=wire, lat_depth=lat_depth, chunk_pairs=chunk_pairs, tuning=tuning,
                           ransac=dict(inlier_thresh=1.5))
        work = [c.clone() for c in clips]  # the stream's buffers (clobbered below)
        enc.load_frames(clips[0])
        enc.step()
        steps = [enc.step_frames(work[k]) for k in (1, 2, 3, 1)]
        check(enc, 1, (schedule, lat_depth, "first burst"))
        assert steps == [1, 2, 3, 4]
        # a longer run; the early buffers are given back and destroyed while the late steps are in flight
        seq = [2, 3, 0, 1, 2, 3, 0, 2]
        ids = [enc.step_frames(work[k]) for k in seq]
        enc.wait_step(ids[3])  # steps over work[2], work[3], work[0], work[1] have let go of their frames ...
        enc.step()             # (the resident clip in between)
        ids.append(enc.step_frames(work[1]))
        enc.wait_step(ids[6])  # ... and so have the next three
        for k in (0, 3):       # not needed again below: work[2] (ids[7]) and work[1] (the last step) may still be read
            work[k].random_(0, 256)
        check(enc, 1, (schedule, lat_depth, "second burst"))
        work[1].random_(0, 256)
        enc.step()
        check(enc, 0, (schedule, lat_depth, "the resident clip again"))
        with pytest.raises(clipmod.ClipError):
            enc.wait_step(10 ** 6)  # never submitted
        odd = torch.empty(clips[0].numel() + 16, dtype=torch.uint8, device=dev)[4:4 + clips[0].numel()].view(clips[0].shape)
        with pytest.raises(clipmod.ClipError):
            enc.step_frames(odd)  # 4 bytes off a 16-byte boundary: refused before anything is enqueued ...
        enc.step()
        check(enc, 0, (schedule, lat_depth, "after a refused step"))  # ... so the pipeline is as it was
        enc.close()


def test_random_call_sequences_give_the_serial_encoders_bytes(native):
    """A slice of tests/helpers/driver_fuzz.py: random configurations of the pipelined driver (output form, chunk plan, pipeline depth, the
    switches of round 6) under random sequences of load / step x k / flush / sync / reset_policy / read over clips whose foreground share
    makes the policy's answer flip -- after every read the resident clip's outputs are the serial two-pass encoder's, bit for bit."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("driver_fuzz", os.path.join(os.path.dirname(__file__), "helpers", "driver_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    import sys
    argv, sys.argv = sys.argv, ["driver_fuzz.py", "--encoders", "30", "--ops", "40", "--seed", "3"]
    try:
        assert fz.main() == 0
    finally:
        sys.argv = argv
