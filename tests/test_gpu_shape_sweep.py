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
