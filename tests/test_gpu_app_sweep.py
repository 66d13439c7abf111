"""A fixed slice of tests/helpers/app_sweep.py in the suite: random command lines the reference's Validate admits -- frame sizes, level
counts, square and non-square MV / transform blocks, search ranges, RANSAC and segmentation parameters, clip lengths around the batch size --
through the reference's unchanged application, both ways (class Encoder on svc::StreamEncoder; the reference's own libs/encoder.cpp on
compat/opencv2), the stream on stdout against the oracle run stage by stage with the same options.  Configurations whose serialiser reads
past its planes in the reference itself (non-square tiles wider than tall) must be refused by the first and are not compared for the second."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [[], ["--adapter"]], ids=["class-encoder", "adapter"])
def test_random_command_lines_give_the_oracles_stream(native, oracle, flags):
    if not os.path.exists(os.path.join(ROOT, "tests", "dropin", "ref_app_svc_encoder_generic")):
        pytest.skip("the reference's application is not built (needs /root/reference at build time)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "app_sweep.py"), "--count", "14", "--seed", "3", *flags],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAIL" not in r.stdout and "command lines give the oracle's stream" in r.stdout
