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


@pytest.mark.timeout(1200)
def test_compat_adapter_host_side_is_clean(tmp_path):
    """compat/opencv2 under the same sanitizers: the matrix semantics program of tests/test_compat_host.py (sharing, views, create()
    reuse, the buffer pool, copyMakeBorder / split / convertTo / extractChannel, both clip containers)."""
    import struct
    import numpy as np
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc: the sanitized host build needs the ROCm clang")
    _make("compat")
    frames = [((np.arange(4 * 6 * 3) + 50 * t) & 255).astype(np.uint8).reshape(4, 6, 3) for t in range(3)]
    raw, ppm = tmp_path / "c.svcbgr", tmp_path / "c.ppm"
    with open(raw, "wb") as f:
        f.write(b"SVCBGR1\0" + struct.pack("<4I", 6, 4, 3, 0))
        for fr in frames:
            f.write(fr.tobytes())
    with open(ppm, "wb") as f:
        for fr in frames:
            f.write(b"P6\n6 4\n255\n" + np.ascontiguousarray(fr[..., ::-1]).tobytes())
    r = subprocess.run([os.path.join(SAN, "_build", "compat_host_asan"), str(raw), str(ppm)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "compat host semantics ok" in r.stdout, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-4000:]


@pytest.mark.timeout(600)
def test_copy_crew_under_thread_and_address_sanitizers():
    """csrc/host/copy_crew.hpp -- the threads that stage source frames for svc::StreamEncoder and move the host-pointer entry points'
    results -- under ThreadSanitizer and under ASan + UBSan: flat and pitched copies of awkward sizes, concurrent callers, start / stop."""
    _make("crew")
    for exe, env in (("copy_crew_tsan", {"TSAN_OPTIONS": "halt_on_error=1"}),
                     ("copy_crew_asan", {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})):
        r = subprocess.run([os.path.join(SAN, "_build", exe)], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        out = r.stdout + r.stderr
        assert r.returncode == 0 and "copy crew ok" in r.stdout, out[-4000:]
        assert "ThreadSanitizer" not in out and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]


@pytest.mark.timeout(600)
def test_reference_ransac_reads_past_its_field():
    if not os.path.exists("/root/reference/libs/motion.cpp") or not os.path.exists(HIPCC):
        pytest.skip("needs the reference tree (this container only)")
    _make("reference")
    r = subprocess.run([os.path.join(SAN, "_build", "ref_ransac_oob")], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and "heap-buffer-overflow" in out and "motion.cpp" in out, out[-2000:]
