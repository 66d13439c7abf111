"""A fixed slice of tests/helpers/shape_sweep.py in the suite: random configurations the reference's Validate admits (odd frame sizes,
1 - 4 levels, 8 / 16 / 32-pixel MV blocks, every dividing transform block, search ranges 2 - 32) through svc::ClipEncoder, pipelined
schedule, every output against the oracle -- pyramids, motion field, RANSAC, region ids bit for bit, coefficients within the parity
tolerance.  (The tool itself ran 400 of them and the named big shapes on the GPU box: DESIGN.md section 7.)"""
import numpy as np
import pytest
import torch

from tests.helpers import shape_sweep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,speculate", [(3, False), (11, False), (5, True)])
def test_random_validate_admitted_configurations_equal_the_oracle(native, oracle, seed, speculate):
    """speculate: the same sweep with the speculative one-pass transform forced on wherever it applies (round 5)."""
    from scalable_video_codec_amd import clip as clipmod
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda")
    failures = []
    for i in range(20):
        cfg = shape_sweep.random_config(rng, i, 360)
        verdict = shape_sweep.check(cfg, oracle, dev, clipmod.TUNE_ALWAYS_SPECULATE if speculate else 0)
        if verdict is not None:
            failures.append((cfg.name, verdict))
    assert not failures, failures


def test_random_configurations_wire_one_pass_equals_two_passes(native):
    """The record output through svc::ClipEncoder on random Validate-admitted configurations: reading the BGR clip once per step (where the
    tuned record emitter applies; everything else falls back by itself) gives the bytes of the two-pass order -- records, pyramids, region ids."""
    from scalable_video_codec_amd import clip as clipmod, synth
    rng = np.random.default_rng(77)
    dev = torch.device("cuda")
    took_one_pass = 0
    for i in range(24):
        cfg = shape_sweep.random_config(rng, i, 300)
        pw, ph = cfg.padded
        src = synth.SynthClip(cfg.width, cfg.height, cfg.frames, cfg.seed, device=dev)
        frames = torch.stack([synth.pad_frame(src.frame_bgr(t), pw, ph) for t in range(cfg.frames)]).contiguous()
        got = {}
        for name, tuning in (("one", 0), ("two", clipmod.TUNE_TWO_BGR_PASSES)):
            enc = clipmod.Clip(cfg, cfg.frames, wire=True, tuning=tuning, ransac=dict(inlier_thresh=1.0))
            enc.load_frames(frames)
            for _ in range(6):
                enc.step(timed=True)
            enc.sync()
            got[name] = (enc.read("records"), enc.read("pyramids"), enc.read("block_types"), "type_patch" in enc.stage_times_ms())
            enc.close()
        took_one_pass += got["one"][3]
        assert not got["two"][3]
        for k in range(3):
            assert torch.equal(got["one"][k], got["two"][k]), (cfg.name, k)
    assert took_one_pass >= 5  # the sweep does reach the one-pass form (8 / 16 transform blocks on widths of whole segments)
