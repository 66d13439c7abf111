"""the host side under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only -- GPU
sanitizers are not available on this pool).  tests/sanitize/Makefile compiles the C-ABI library's host code, the
C++ layer and the oracle with -fsanitize=address,undefined into one executable; asan_main.cpp walks the oracle over
exactly-sized heap buffers and every entry point's argument validation.  Where /root/reference exists, the
unmodified reference RANSAC is built the same way and shown to read one vector past its field (libs/motion.cpp:208)
-- the out-of-bounds draw this build deliberately does not reproduce."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _make(target):
    r = subprocess.run(["make", "-s", "-j8", "-C", SAN, target], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.timeout(1200)
def test_host_side_is_clean_under_asan_and_ubsan():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc: the sanitized host build needs the ROCm clang")
    _make("all")
    r = subprocess.run([os.path.join(SAN, "_build", "svc_asan_main")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "checks passed" in r.stdout
    assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-4000:]


@pytest.mark.timeout(600)
def test_reference_ransac_reads_past_its_field():
    if not os.path.exists("/root/reference/libs/motion.cpp") or not os.path.exists(HIPCC):
        pytest.skip("needs the reference tree (this container only)")
    _make("reference")
    r = subprocess.run([os.path.join(SAN, "_build", "ref_ransac_oob")], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and "heap-buffer-overflow" in out and "motion.cpp" in out, out[-2000:]
