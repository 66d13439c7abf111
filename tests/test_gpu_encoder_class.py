"""The drop-in one level above libs/motion.hpp: the reference's `class Encoder` (libs/encoder.hpp:52-95) implemented on
svc::StreamEncoder (scalable_video_codec_amd/csrc/host/encoder_hip.cpp, compiled against the reference's OWN encoder.hpp) and driven by
the reference's UNCHANGED apps/encoder.cpp + libs/cli.cpp (tests/dropin/ref_app_svc_encoder[_generic], built by build.py where
/root/reference exists).  Same command line, same reader / writer threads and queues, same stdout format -- but the per-frame loop is
the batched GPU pipeline instead of the reference's libs/encoder.cpp.

Checked: the stream (Header + one record block per frame, the reference's unpadded tile loops and row stride included) against the
oracle run stage by stage with the draws and seeds this Encoder uses (a function of one seed and the frame index, like the harness);
and that it is the same stream whatever the frame count does to the batching."""
import os
import struct
import subprocess
import time

import numpy as np
import pytest

from scalable_video_codec_amd import pipeline, synth
from tests.test_gpu_ref_encoder import _check, _write_clip, _write_ppm_stream

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")
SEED = 9001


def _encode(exe, clip, *args):
    path = os.path.join(BIN, exe)
    if not os.path.exists(path):
        pytest.skip(f"{exe} not built (needs /root/reference at build time)")
    t0 = time.time()
    r = subprocess.run([path, *args, str(clip)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, SVC_TEST_ENCODER_SEED=str(SEED)))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout, time.time() - t0


def _expected(oracle, frames, levels, tb, mv_block=16, search_range=8, seg=None, ransac=None):
    """Header + per frame (tile types, tile coefficients): the oracle's stages with THIS Encoder's draws (pipeline.ransac_samples) and
    segmentation seeds (seed * 1000003 + clip-wide pair index), serialised with the reference's own arguments.  tb / mv_block: a side,
    or (width, height)."""
    h, w, _ = frames[0].shape
    bw, bh = (mv_block, mv_block) if isinstance(mv_block, int) else mv_block
    tw, th = (tb, tb) if isinstance(tb, int) else tb
    pw, ph = synth.padded_dims(w, h, bw, bh, levels)
    mfw, mfh = pw // bw, ph // bh
    header = struct.pack("<8I", len(frames) - 1, w, h, pw - w, ph - h, tw, th, 3)
    padded = []
    for fr in frames:
        p = np.zeros((ph, pw, 3), np.uint8)
        p[:h, :w] = fr
        padded.append(p)
    pyrs = [oracle.luma_pyramid(p, levels) for p in padded]
    rp = {**dict(subset_sz=1, inlier_thresh=7.5, success_prob=0.99, inlier_ratio=0.5), **(ransac or {})}
    iters = oracle.ransac_iter_count(**rp)
    samples = pipeline.ransac_samples(len(frames) - 1, iters, rp["subset_sz"], mfw * mfh, SEED, "cpu").numpy().astype(np.uint32)
    out = []
    for t in range(1, len(frames)):
        mv, _ = oracle.hbma(pyrs[t - 1], pyrs[t], search_range, bw, bh)
        _, _, inliers = oracle.ransac(mv, samples[t - 1].ravel(), **rp)
        mask = np.zeros(mfw * mfh, np.uint8)
        mask[inliers] = 1
        types = oracle.segment(mask, mv, mfw, mfh, bw, bh, seed=SEED * 1000003 + (t - 1), **(seg or {}))
        planes = oracle.dct_frame_f32(padded[t], tw, th)
        rec = oracle.serialize_frame(planes, types, w, h, tw, th, mfw, bw, bh)
        rec = rec.view(np.uint32).reshape(-1, 1 + 3 * tw * th)
        out.append((rec[:, 0].copy(), rec[:, 1:].copy().view(np.float32)))
    return header, out


def test_batched_encoder_class_1080p(native, oracle, tmp_path):
    """1080p through the reference's unchanged main() with the batched Encoder: 21 frames = one full batch of 16 + a short one."""
    n = 21
    clip = synth.SynthClip(1920, 1080, n, seed=0x5C0DEC02)
    frames = [clip.frame_bgr(t).numpy() for t in range(n)]
    path = tmp_path / "clip.svcbgr"
    _write_clip(path, frames)
    got, secs = _encode("ref_app_svc_encoder", path, "--verbose", "0")
    header, expected = _expected(oracle, frames, 4, 8)
    fg = _check(got, header, expected, 8)
    assert fg > 0
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "ref_app_svc_encoder.txt"), "w") as f:
        f.write(f"ref_app_svc_encoder: {n - 1} encoded 1080p frames in {secs:.2f} s (process start to exit), {len(got)} bytes\n")


@pytest.mark.parametrize("n", [2, 3, 17, 18, 34])
def test_batched_encoder_class_any_clip_length(native, oracle, tmp_path, n):
    """Clip lengths around the batch size (16 encoded frames per batch: 17 frames = exactly one batch, 18 = one frame into the second,
    2 = a single pair) on a 344 x 280 clip (pads to 352 x 288: the serialiser's unpadded-stride quirk, so the planes + serialiser route)
    from a PPM stream, with 16 x 16 transform blocks."""
    clip = synth.SynthClip(344, 280, n, seed=5)
    frames = [clip.frame_bgr(t).numpy() for t in range(n)]
    path = tmp_path / "clip.ppm"
    _write_ppm_stream(path, frames)
    got, _ = _encode("ref_app_svc_encoder", path, "--verbose", "0", "--transform-block-w", "16", "--transform-block-h", "16")
    header, expected = _expected(oracle, frames, 4, 16)
    _check(got, header, expected, 16)


def test_batched_encoder_class_generic_build_and_limits(native, oracle, tmp_path):
    """The build without -DSVC_MOTION_SSE2: --pyr-lvl-count 3 and 8 x 8 MV blocks from the command line; a single-frame clip is a header
    and nothing else."""
    n = 6
    clip = synth.SynthClip(360, 200, n, seed=11)
    frames = [clip.frame_bgr(t).numpy() for t in range(n)]
    path = tmp_path / "clip.svcbgr"
    _write_clip(path, frames)
    args = ["--verbose", "0", "--pyr-lvl-count", "3", "--mv-block-w", "8", "--mv-block-h", "8", "--transform-block-w", "4",
            "--transform-block-h", "4", "--kmeans-cluster-count", "4"]
    got, _ = _encode("ref_app_svc_encoder_generic", path, *args)
    header, expected = _expected(oracle, frames, 3, 4, mv_block=8, seg=dict(cluster_count=4))
    _check(got, header, expected, 4)
    one = tmp_path / "one.svcbgr"
    _write_clip(one, frames[:1])
    got, _ = _encode("ref_app_svc_encoder_generic", one, "--verbose", "0")
    assert len(got) == 32 and struct.unpack("<3I", got[:12]) == (0, 360, 200)


@pytest.mark.parametrize("size,mv,tb,levels", [((344, 280), (32, 16), (16, 8), 3), ((352, 288), (16, 32), (8, 16), 2), ((360, 200), (8, 16), (8, 4), 3),
                                               ((352, 288), (8, 32), (8, 16), 2)])  # the last: tile height 16 > MV block width 8 -- the reference's swapped assert
                                                                                   # (libs/encoder.cpp:235) would fail, its Release build runs
def test_batched_encoder_class_non_square_blocks(native, oracle, tmp_path, size, mv, tb, levels):
    """--mv-block-w != --mv-block-h and --transform-block-w != --transform-block-h (Validate admits them, libs/encoder.cpp:62-142): the
    per-level search kernel and the planes + serialiser route instead of the tuned kernels, same stream as the oracle's stages
    serialised with the reference's own arguments."""
    n = 5
    clip = synth.SynthClip(size[0], size[1], n, seed=17)
    frames = [clip.frame_bgr(t).numpy() for t in range(n)]
    path = tmp_path / "clip.svcbgr"
    _write_clip(path, frames)
    args = ["--verbose", "0", "--pyr-lvl-count", str(levels), "--mv-block-w", str(mv[0]), "--mv-block-h", str(mv[1]),
            "--transform-block-w", str(tb[0]), "--transform-block-h", str(tb[1])]
    got, _ = _encode("ref_app_svc_encoder_generic", path, *args)
    header, expected = _expected(oracle, frames, levels, tb, mv_block=mv)
    _check(got, header, expected, tb)


@pytest.mark.parametrize("opts,kw", [
    (["--mv-search-range", "16", "--ransac-subset-sz", "3", "--ransac-inlier-thresh", "2.5", "--ransac-success-prob", "0.999", "--ransac-inlier-ratio", "0.4"],
     dict(search_range=16, ransac=dict(subset_sz=3, inlier_thresh=2.5, success_prob=0.999, inlier_ratio=0.4))),
    (["--morph-rect-w", "5", "--morph-rect-h", "1", "--kmeans-cluster-count", "6", "--kmeans-attempt-count", "5", "--kmeans-max-iter-count", "4",
      "--kmeans-epsilon", "2.5", "--connected-components-connectivity", "8"],
     dict(seg=dict(morph_w=5, morph_h=1, cluster_count=6, attempts=5, max_iter=4, epsilon=2.5, connectivity=8))),
], ids=["search-range-and-ransac", "segmentation"])
def test_batched_encoder_class_every_other_option(native, oracle, tmp_path, opts, kw):
    """The options the tests above leave at their defaults -- search range, the four RANSAC parameters, structuring element, the four
    k-means parameters, connectivity (apps/encoder.cpp:75-104) -- reach the kernels: same stream as the oracle run with those values."""
    n = 5
    clip = synth.SynthClip(352, 288, n, seed=23)
    frames = [clip.frame_bgr(t).numpy() for t in range(n)]
    path = tmp_path / "clip.svcbgr"
    _write_clip(path, frames)
    got, _ = _encode("ref_app_svc_encoder", path, "--verbose", "0", *opts)
    header, expected = _expected(oracle, frames, 4, 8, **kw)
    assert _check(got, header, expected, 8) >= 0
