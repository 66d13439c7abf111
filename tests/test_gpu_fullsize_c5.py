"""BASELINE config 5 at full size on one GPU: 4K (3840x2160), 64 frames, 4-level pyramid (R_top = 1), 16x16 DCT + quant,
through the C++ driver (svc::ClipEncoder, pipelined schedule) -- size-independent properties on all 63 pairs, sampled
pairs against the oracle, and the 8-rank sharding of the same clip (BASELINE's 8-GPU form, every rank's shard run here
one after the other with the halo handed over by the transport hook) against the unsharded result."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.binding import DEFAULT_RANSAC
from scalable_video_codec_amd import clip as clipmod
from scalable_video_codec_amd import configs, pipeline, synth

pytestmark = pytest.mark.gpu
CFG = configs.C5


@pytest.fixture(scope="module")
def encoded(native):
    dev = torch.device("cuda")
    cfg = CFG
    src = synth.SynthClip(cfg.width, cfg.height, cfg.frames, cfg.seed, device=dev)
    pw, ph = cfg.padded
    enc = clipmod.Clip(cfg, cfg.frames)
    for t in range(cfg.frames):
        enc.load_frames(synth.pad_frame(src.frame_bgr(t), pw, ph).unsqueeze(0).contiguous(), t)
    for _ in range(4):  # the pipeline in steady state
        enc.step()
    enc.sync()
    out = enc.outputs(device=dev)
    out["pyr"] = enc.read("pyramids", device=dev)
    out["bgr"] = enc.read("bgr", device=dev).view(cfg.frames, ph, pw, 3)
    yield cfg, enc, out
    enc.close()


def test_two_kernels_agree_on_every_pair(native, encoded):
    cfg, enc, out = encoded
    i = enc.info
    s = i.pyramid_stride
    mv, mad = native.hbma_pairs(out["pyr"][s:], out["pyr"][2 * s:], s, i.pairs, cfg.levels, i.padded_w, i.padded_h,
                                cfg.search_range, flags=native.HBMA_FORCE_WAVE_PER_BLOCK)
    torch.cuda.synchronize()
    assert torch.equal(mv, out["mv"]) and torch.equal(mad, out["min_mad"])
    bound = cfg.r_top * ((1 << cfg.levels) - 1)  # |mv| <= R_top (2^L - 1)
    assert float(out["mv"].abs().max()) <= bound and bool((out["mv"] == out["mv"].round()).all())


def test_sampled_pairs_against_oracle(oracle, encoded):
    cfg, enc, out = encoded
    i = enc.info
    offs = synth.level_offsets(i.padded_w, i.padded_h, cfg.levels)

    def planes(slot):
        flat = out["pyr"][slot * i.pyramid_stride:(slot + 1) * i.pyramid_stride].cpu().numpy()
        return [flat[offs[l]:offs[l] + (i.padded_w >> l) * (i.padded_h >> l)].reshape(i.padded_h >> l, i.padded_w >> l)
                for l in range(cfg.levels)]
    for p in (0, 31, 62):
        mv, mad = oracle.hbma16_sse2(planes(p + 1), planes(p + 2), cfg.search_range)  # the reference's default-build path, restated
        assert np.array_equal(out["mv"][p].cpu().numpy(), mv) and np.array_equal(out["min_mad"][p].cpu().numpy(), mad)
    for l, ref in enumerate(oracle.luma_pyramid(out["bgr"][9].cpu().numpy(), cfg.levels)):  # device pyramid == the oracle's
        o = 10 * i.pyramid_stride + offs[l]
        assert np.array_equal(out["pyr"][o:o + ref.size].cpu().numpy().reshape(ref.shape), ref)


def test_ransac_and_region_ids(oracle, encoded):
    cfg, enc, out = encoded
    i = enc.info
    samples = pipeline.ransac_samples(i.pairs, i.ransac_iters, 1, i.blocks, cfg.seed, "cpu").numpy().astype(np.uint32)
    for p in (0, 40, 62):
        mv = out["mv"][p].cpu().numpy()
        gm, rmse, inl = oracle.ransac(mv, samples[p].ravel(), **DEFAULT_RANSAC)
        assert out["global_motion"][p].cpu().numpy().tobytes() == gm.tobytes()
        assert np.float32(out["rmse"][p].item()).tobytes() == rmse.tobytes()
        mask = out["inlier_mask"][p].cpu().numpy()
        assert np.array_equal(np.flatnonzero(mask), inl) and int(out["inlier_count"][p]) == len(inl)
        want = oracle.segment(mask, mv, i.mv_field_w, i.mv_field_h, seed=cfg.seed * 1000003 + p)
        assert np.array_equal(out["block_types"][p].cpu().numpy().astype(np.uint32), want)


def test_dct16_energy_dc_and_quant(native, encoded):
    cfg, enc, out = encoded
    i = enc.info
    n, pw, ph = i.pairs, i.padded_w, i.padded_h
    coeffs = enc.read("coeffs", device=out["mv"].device).view(n, 3, ph, pw)
    types = out["block_types"]
    worst_e = worst_dc = 0.0
    for f0 in range(0, n, 9):  # in slices: the raw planes of the whole clip would be another 6 GB
        f1 = min(n, f0 + 9)
        frames = out["bgr"][1 + f0:1 + f1].contiguous()
        raw = native.dct_frames(frames, 16)
        px = frames.to(torch.float64)
        e_in = (px * px).sum(dim=(1, 2, 3))
        e_out = (raw.to(torch.float64) ** 2).sum(dim=(1, 2, 3))
        worst_e = max(worst_e, float(((e_in - e_out).abs() / e_in).max()))
        dc = raw[:, :, ::16, ::16].to(torch.float64)
        means = px.permute(0, 3, 1, 2).reshape(f1 - f0, 3, ph // 16, 16, pw // 16, 16).mean(dim=(3, 5))
        worst_dc = max(worst_dc, float((dc - 16 * means).abs().max()))
        native.quant_frames_(raw, types[f0:f1].contiguous(), cfg.mv_block, cfg.fg_step, cfg.bg_step)
        assert torch.equal(raw, coeffs[f0:f1])  # fused DCT + quant == quant of the raw DCT, bit for bit
        del raw, px
    assert worst_e < 1e-6 and worst_dc < 4e-3  # Parseval; DC = 16 x tile mean


_hip = None


def _copy_async(dst, src, nbytes, stream):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so.7")
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert _hip.hipMemcpyAsync(dst, src, nbytes, 3, stream) == 0


def test_eight_rank_sharding_equals_the_clip(native, encoded):
    cfg, enc, out = encoded
    world = 8
    stride = enc.info.pyramid_stride
    coeffs = enc.read("coeffs", device=out["mv"].device).view(enc.info.pairs, -1)
    done = 0
    for r in range(world):
        first, frames, pairs, first_encoded = clipmod.plan_shard(cfg.frames, world, r)
        shard = clipmod.Clip(cfg, cfg.frames, rank=r, world=world)
        shard.load_frames(out["bgr"][first:first + frames].contiguous())

        def transport(send, recv, nbytes, stream, first=first):
            if first > 0:  # what rank r - 1 would send: the pyramid of clip frame first - 1 (slot first of the whole clip)
                _copy_async(recv, out["pyr"].data_ptr() + first * stride, nbytes, stream)
        shard.set_halo_transport(transport)
        for _ in range(3):
            shard.step()
        shard.sync()
        o = shard.outputs(device=out["mv"].device)
        g0 = first_encoded - 1
        assert g0 == done
        for k in ("mv", "min_mad", "global_motion", "rmse", "inlier_mask", "inlier_count", "block_types"):
            assert torch.equal(o[k], out[k][g0:g0 + pairs]), (r, k)
        assert torch.equal(shard.read("coeffs", device=out["mv"].device).view(pairs, -1), coeffs[g0:g0 + pairs])
        done += pairs
        shard.close()
    assert done == cfg.frames - 1
