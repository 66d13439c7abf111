"""compat/opencv2 on the GPU (tests/compat/compat_gpu_main.cpp): every forwarding cv:: function == the C ABI called directly, the
reuse of preallocated planes, cv::theRNG's step per cv::kmeans call, and the deferred cv::dct list -- calls that depend on each other
(the same tile twice, another shape over collected tiles, off-grid and out-of-order tiles) run in turn, as OpenCV's eager cv::dct would."""
import os
import shutil
import subprocess

import pytest

from scalable_video_codec_amd import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compat_forwarding_and_deferred_dct(native, tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++ on this box")
    exe = tmp_path / "compat_gpu"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", f"-I{os.path.join(ROOT, 'compat')}", f"-I{os.path.join(ROOT, 'include')}",
                           "-o", str(exe), os.path.join(ROOT, "tests", "compat", "compat_gpu_main.cpp"), f"-L{build.PKG}",
                           "-lsvc_opencv_compat", "-lsvc_hip", f"-Wl,-rpath,{build.PKG}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "compat gpu semantics ok" in r.stdout, r.stdout + r.stderr
