"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU).

  libsvc_hip.so     HIP kernels + the C ABI of include/svc_hip.h
  libsvc_motion.so  C++ wrappers with the reference's own signatures
                    (include/svc/motion.hpp), on top of the C ABI
  libsvc_opencv_compat.so  compat/opencv2: the OpenCV-shaped adapter the reference's own
                    encoder application compiles against, on top of the C ABI

Both land next to this file; they are git-ignored but travel to the GPU box.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from typing import List

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(PKG, "_obj")
LIB_HIP = os.path.join(PKG, "libsvc_hip.so")
LIB_MOTION = os.path.join(PKG, "libsvc_motion.so")

HIP_SOURCES = ["capi.hip", "hbma_wave.hip", "hbma_fused.hip", "hbma_fused8.hip", "hbma_fused32.hip", "hbma_tiled.hip", "dct.hip", "ransac.hip", "luma_pyramid.hip", "segment.hip", "wire.hip", "idct.hip", "probe.hip", "comm.hip", "global_motion.hip", "imageops.hip"]
HOST_SOURCES = [os.path.join("host", "motion_hip.cpp")]

# -ffp-contract=off: the reference's float expressions (RANSAC inlier test, quant) are
# evaluated without FMA on baseline x86-64; the DCT asks for its FMAs explicitly.
# -Wno-inline-asm: hbma_tiled.hip's LDS-DMA asm names m0 in its clobber list ON PURPOSE (it writes m0; the backend must not
# merge one of its own m0 initialisations across it); LLVM answers every such asm with "clobber list contains reserved registers".
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-Wno-inline-asm", f"-I{INCLUDE}", f"-I{CSRC}"]


# A/B experiments: extra device-compile flags (-D... of a variant under test) without editing the sources
EXTRA_FLAGS = os.environ.get("SVC_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X path cannot be built")
    return exe


def _newer(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {os.path.basename(cmd[-1])}")


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "svc_common.hpp"), os.path.join(CSRC, "union_find.hpp"), os.path.join(CSRC, "hbma_search.hpp"), os.path.join(CSRC, "hbma_fused_kernel.hpp"), os.path.join(CSRC, "dct_tables.inc"), os.path.join(CSRC, "luma16.hpp"),
               os.path.join(CSRC, "host", "copy_crew.hpp"), os.path.join(INCLUDE, "svc_hip.h")]
    jobs, objs = [], []
    for s in HIP_SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            jobs.append([_hipcc(), *HIPCC_FLAGS, *EXTRA_FLAGS, "-c", src, "-o", obj])
    if jobs:
        if verbose:
            print(f"[build] compiling {len(jobs)} HIP translation unit(s) for gfx950", flush=True)
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if jobs or not os.path.exists(LIB_HIP):
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP, *objs])
    return LIB_HIP


STREAM_SRC = os.path.join(CSRC, "host", "stream_encoder.cpp")
CLIP_SRC = os.path.join(CSRC, "host", "clip_encoder.cpp")


def build_motion(force: bool = False) -> str:
    """The C++ layer above the C ABI: the reference's motion.hpp entry points (plain C++, g++) and the
    batched host-memory encoder (uses the HIP runtime for buffers, streams and events: hipcc, host only)."""
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    deps = srcs + [STREAM_SRC, CLIP_SRC, os.path.join(CSRC, "host", "copy_crew.hpp"), os.path.join(INCLUDE, "svc_hip.h"), os.path.join(INCLUDE, "svc_clip.h")] + \
        [os.path.join(INCLUDE, "svc", h) for h in ("motion.hpp", "math.hpp", "types.hpp", "stream_encoder.hpp",
                                                    "clip_encoder.hpp")]
    if force or not _newer(LIB_MOTION, deps + [LIB_HIP]):
        cxx = shutil.which("g++") or "g++"
        stream_obj = os.path.join(OBJ, "stream_encoder.o")
        clip_obj = os.path.join(OBJ, "clip_encoder.o")
        _run([_hipcc(), "-std=c++17", "-O2", "-fPIC", "-Wall", f"-I{INCLUDE}", "-c", STREAM_SRC, "-o", stream_obj])
        _run([_hipcc(), "-std=c++17", "-O2", "-fPIC", "-Wall", f"-I{INCLUDE}", "-c", CLIP_SRC, "-o", clip_obj])
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib")
        _run([cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", f"-I{INCLUDE}", f"-I{os.path.join(INCLUDE, 'svc')}",
              "-o", LIB_MOTION, *srcs, stream_obj, clip_obj, f"-L{PKG}", "-lsvc_hip", f"-L{rocm_lib}", "-lamdhip64",
              "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{rocm_lib}"])
    return LIB_MOTION


DROPIN_SRC = os.path.join(ROOT, "tests", "dropin", "dropin_main.cpp")
REFERENCE_LIBS = os.environ.get("SVC_REFERENCE_DIR", "/root/reference") + "/libs"


def build_dropin(force: bool = False) -> List[str]:
    """A caller written against the reference's motion.hpp only, linked to libsvc_motion.so:
    once with the reference's OWN header (where /root/reference exists; the binary travels
    to the GPU box), once with include/svc/motion.hpp."""
    cxx = shutil.which("g++") or "g++"
    out = []
    variants = [("dropin_own_hdr", os.path.join(INCLUDE, "svc"), [])]
    if os.path.exists(os.path.join(REFERENCE_LIBS, "motion.hpp")):
        variants.append(("dropin_ref_hdr", REFERENCE_LIBS, []))
        variants.append(("dropin_ref_hdr_sse2", REFERENCE_LIBS, ["-DDROPIN_USE_SSE2_ENTRY"]))
    for name, inc, extra in variants:
        exe = os.path.join(os.path.dirname(DROPIN_SRC), name)
        if force or not _newer(exe, [DROPIN_SRC, LIB_MOTION]):
            _run([cxx, "-std=c++17", "-O2", "-msse2", f"-I{inc}", *extra, "-o", exe, DROPIN_SRC, f"-L{PKG}",
                  "-lsvc_motion", "-lsvc_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/../../scalable_video_codec_amd"])
        out.append(exe)
    # a host application written against include/svc/stream_encoder.hpp only
    exe = os.path.join(os.path.dirname(DROPIN_SRC), "stream_main")
    src = os.path.join(os.path.dirname(DROPIN_SRC), "stream_main.cpp")
    if force or not _newer(exe, [src, LIB_MOTION, os.path.join(INCLUDE, "svc", "stream_encoder.hpp")]):
        _run([cxx, "-std=c++17", "-O2", f"-I{INCLUDE}", "-o", exe, src, f"-L{PKG}", "-lsvc_motion", "-lsvc_hip",
              f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/../../scalable_video_codec_amd", "-Wl,--allow-shlib-undefined"])
    out.append(exe)
    # ... and one that uses it at random against itself (batch sizes, depths, entry points, reuse): tests/test_gpu_stream.py
    exe = os.path.join(os.path.dirname(DROPIN_SRC), "stream_fuzz")
    src = os.path.join(os.path.dirname(DROPIN_SRC), "stream_fuzz.cpp")
    if force or not _newer(exe, [src, LIB_MOTION, os.path.join(INCLUDE, "svc", "stream_encoder.hpp")]):
        _run([cxx, "-std=c++17", "-O2", f"-I{INCLUDE}", "-o", exe, src, f"-L{PKG}", "-lsvc_motion", "-lsvc_hip",
              f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/../../scalable_video_codec_amd", "-Wl,--allow-shlib-undefined"])
    out.append(exe)
    return out


COMPAT = os.path.join(ROOT, "compat")
LIB_COMPAT = os.path.join(PKG, "libsvc_opencv_compat.so")


def build_compat(force: bool = False) -> str:
    """compat/opencv2: the slice of the OpenCV API the reference's encoder uses, every arithmetic call forwarding to
    include/svc_hip.h (a product-side adapter -- never an oracle).  Plain C++ on top of the C ABI."""
    srcs = [os.path.join(COMPAT, "src", f) for f in ("core.cpp", "imgproc.cpp", "videoio.cpp")]
    hdrs = [os.path.join(COMPAT, "src", "internal.hpp"), os.path.join(CSRC, "host", "copy_crew.hpp"), os.path.join(INCLUDE, "svc_hip.h")] + \
        [os.path.join(COMPAT, "opencv2", f) for f in ("core.hpp", "imgproc.hpp", "videoio.hpp", os.path.join("core", "mat.hpp"))]
    if force or not _newer(LIB_COMPAT, srcs + hdrs + [LIB_HIP]):
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", f"-I{COMPAT}", f"-I{INCLUDE}", f"-I{CSRC}", "-o", LIB_COMPAT,
              *srcs, f"-L{PKG}", "-lsvc_hip", "-Wl,-rpath,$ORIGIN"])
    return LIB_COMPAT


REFERENCE_APPS = os.environ.get("SVC_REFERENCE_DIR", "/root/reference") + "/apps"


def reference_encoder_command(exe: str, sse2: bool, extra_sources: List[str] = ()) -> List[str]:
    """THE build line of INTEGRATION.md section 6: the reference's own apps/encoder.cpp + libs/encoder.cpp + libs/cli.cpp,
    compiled where they lie (never copied, never edited), against compat/opencv2 instead of OpenCV, with this repo's
    ThreadGuard in place of the reference's thread.cpp (which does not compile), linked to the three product libraries."""
    cxx = shutil.which("g++") or "g++"
    return [cxx, "-std=c++17", "-O2", "-DNDEBUG", "-msse2", *(["-DSVC_MOTION_SSE2"] if sse2 else []),
            f"-I{COMPAT}", f"-I{REFERENCE_LIBS}", "-o", exe,
            os.path.join(REFERENCE_APPS, "encoder.cpp"), os.path.join(REFERENCE_LIBS, "encoder.cpp"),
            os.path.join(REFERENCE_LIBS, "cli.cpp"), os.path.join(COMPAT, "src", "thread_guard.cpp"), *extra_sources,
            f"-L{PKG}", "-lsvc_opencv_compat", "-lsvc_motion", "-lsvc_hip", "-pthread",
            f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/../../scalable_video_codec_amd"]


def svc_encoder_app_command(exe: str, sse2: bool = True, extra_sources: List[str] = ()) -> List[str]:
    """INTEGRATION.md section 3, second build line: the reference's unchanged apps/encoder.cpp + libs/cli.cpp with THIS repo's
    implementation of the reference's class Encoder (csrc/host/encoder_hip.cpp, compiled against the reference's own encoder.hpp)
    in place of libs/encoder.cpp -- the batched GPU form of the same application."""
    cxx = shutil.which("g++") or "g++"
    return [cxx, "-std=c++17", "-O2", "-DNDEBUG", "-msse2", *(["-DSVC_MOTION_SSE2"] if sse2 else []), f"-I{COMPAT}", f"-I{REFERENCE_LIBS}",
            f"-I{INCLUDE}", "-o", exe, os.path.join(REFERENCE_APPS, "encoder.cpp"), os.path.join(REFERENCE_LIBS, "cli.cpp"),
            os.path.join(CSRC, "host", "encoder_hip.cpp"), os.path.join(COMPAT, "src", "thread_guard.cpp"), *extra_sources,
            f"-L{PKG}", "-lsvc_opencv_compat", "-lsvc_motion", "-lsvc_hip", "-pthread",
            f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/../../scalable_video_codec_amd"]


def build_reference_encoder(force: bool = False) -> List[str]:
    """tests/dropin/ref_encoder_{sse2,generic}: the reference's unchanged encoder application on the HIP path,
    built where /root/reference exists; the binaries travel to the GPU box like the other drop-in callers.  The only extra
    object is tests/dropin/ransac_seed.cpp, a TEST seam (it seeds the wrapper's RANSAC engine before main() so that the
    output stream is repeatable; the reference seeds from std::random_device)."""
    if not os.path.exists(os.path.join(REFERENCE_APPS, "encoder.cpp")):
        return []
    here = os.path.dirname(DROPIN_SRC)
    seed_src = os.path.join(here, "ransac_seed.cpp")
    deps = [LIB_COMPAT, LIB_MOTION, seed_src, os.path.join(COMPAT, "src", "thread_guard.cpp")] + \
        [os.path.join(COMPAT, "opencv2", f) for f in ("core.hpp", "imgproc.hpp", "videoio.hpp", os.path.join("core", "mat.hpp"))]
    out = []
    for name, sse2 in (("ref_encoder_sse2", True), ("ref_encoder_generic", False)):
        exe = os.path.join(here, name)
        if force or not _newer(exe, deps):
            _run(reference_encoder_command(exe, sse2, [seed_src]))
        out.append(exe)
    # the reference's unchanged apps/encoder.cpp on the BATCHED Encoder (csrc/host/encoder_hip.cpp: class Encoder of the reference's own
    # encoder.hpp implemented on svc::StreamEncoder) instead of the reference's libs/encoder.cpp
    enc_src = os.path.join(CSRC, "host", "encoder_hip.cpp")
    seed2 = os.path.join(here, "encoder_seed.cpp")
    for name, sse2 in (("ref_app_svc_encoder", True), ("ref_app_svc_encoder_generic", False)):
        exe = os.path.join(here, name)
        if force or not _newer(exe, [enc_src, seed2, LIB_COMPAT, LIB_MOTION, os.path.join(INCLUDE, "svc", "stream_encoder.hpp")]):
            _run(svc_encoder_app_command(exe, sse2, [seed2]))
        out.append(exe)
    # the libstdc++ engine + distribution the C++ RANSAC wrapper draws with, for tests that mirror its draws
    exe = os.path.join(here, "ransac_draws")
    src = os.path.join(here, "ransac_draws.cpp")
    if force or not _newer(exe, [src]):
        _run([shutil.which("g++") or "g++", "-std=c++17", "-O2", "-o", exe, src])
    out.append(exe)
    return out


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_motion(force)
    build_compat(force)
    build_dropin(force)
    build_reference_encoder(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print(LIB_HIP)
    print(LIB_MOTION)
