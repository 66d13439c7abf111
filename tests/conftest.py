import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU suite (the driver runs `pytest -x`): the HIP-vs-oracle parity files first, in the order of SURVEY 8a's
# rows, then the property / sweep files, and LAST everything that spawns bench.py or the application binaries -- a failure there
# can no longer hide a parity result (round 4: one stopwatch assertion in the bench contract stopped the run before any of them).
_FIRST = ("test_gpu_golden", "test_gpu_hbma", "test_gpu_dct_quant", "test_gpu_ransac_pyramid", "test_gpu_segment", "test_gpu_imageops",
          "test_gpu_wire", "test_gpu_decode", "test_gpu_fullsize", "test_gpu_fullsize_c5", "test_gpu_clip")
_LAST = ("test_gpu_dropin", "test_gpu_stream", "test_gpu_compat", "test_gpu_encoder_class", "test_gpu_ref_encoder", "test_gpu_app_sweep",
         "test_gpu_bench_contract")


def _order_key(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _FIRST:
        return (0, _FIRST.index(name))
    if name in _LAST:
        return (2, _LAST.index(name))
    return (1, 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_order_key)  # stable: the order inside a file, and of the files in the middle group, is untouched


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once, here, rather than fail every
    test that loads them (hipcc cross-compiles gfx950 without a GPU; a minute the first time)."""
    from scalable_video_codec_amd import build as b
    if not (os.path.exists(b.LIB_HIP) and os.path.exists(b.LIB_MOTION)):
        try:
            b.build_all(verbose=True)
        except RuntimeError as e:  # no hipcc on this box: the oracle / golden / host-logic tests still run,
            print(f"[conftest] native libraries not built: {e}", file=sys.stderr)  # the `native` fixture skips


@pytest.fixture(scope="session")
def oracle():
    """The C restatement (oracle/libsvc_oracle.so); built on demand with gcc."""
    import subprocess
    from oracle import binding
    if not os.path.exists(binding.ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return binding.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference motion.cpp (oracle/_ref); skipped where it was never built."""
    from oracle import binding
    if not binding.Reference.available():
        pytest.skip("oracle/_ref/libsvc_ref.so not built (needs /root/reference)")
    return binding.Reference()


@pytest.fixture(scope="session")
def native():
    """The product library; GPU tests must go through it (no fallback)."""
    import torch
    from scalable_video_codec_amd import build as b
    from scalable_video_codec_amd import native as n
    if not os.path.exists(b.LIB_HIP) and not torch.cuda.is_available():
        pytest.skip("libsvc_hip.so is not built (no hipcc) and no GPU is visible")
    n.load()  # on a GPU box a missing library is an error, not a skip: the product path has no fallback
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return n
