"""The C++ drop-in: a caller compiled against the REFERENCE'S OWN motion.hpp (tests/dropin/,
built by scalable_video_codec_amd/build.py where /root/reference exists) and linked against
libsvc_motion.so runs on the GPU and reproduces the reference's golden outputs."""
import os
import struct
import subprocess

import numpy as np
import pytest

from scalable_video_codec_amd import configs
from tests import golden_util as G

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def _run(exe, cfg, tmp_path):
    path = os.path.join(BIN, exe)
    if not os.path.exists(path):
        pytest.skip(f"{exe} not built")
    (_, p0), (_, p1) = G.config_pair(cfg)
    pw, ph = cfg.padded
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<7I3f", cfg.levels, pw, ph, cfg.search_range, 16, 16, 1, 7.5, 0.99, 0.5))
        for p in p0 + p1:
            f.write(np.ascontiguousarray(p).tobytes())
    subprocess.run([path, str(fin), str(fout)], check=True, timeout=120)
    raw = open(fout, "rb").read()
    b = cfg.blocks
    mv = np.frombuffer(raw, np.float32, 2 * b).reshape(b, 2)
    mad = np.frombuffer(raw, np.float32, b, 8 * b)
    gm = np.frombuffer(raw, np.float32, 2, 12 * b)
    rmse = np.frombuffer(raw, np.float32, 1, 12 * b + 8)[0]
    n = int(np.frombuffer(raw, np.uint32, 1, 12 * b + 12)[0])
    inl = np.frombuffer(raw, np.uint32, n, 12 * b + 16)
    _run.extra = raw[12 * b + 16 + 4 * n:]
    return mv, mad, gm, rmse, inl


@pytest.mark.parametrize("exe,cfg", [("dropin_ref_hdr", configs.C2), ("dropin_ref_hdr_sse2", configs.C3_L4),
                                     ("dropin_own_hdr", configs.C3)], ids=["ref-header-C2", "ref-header-sse2-entry", "own-header-C3"])
def test_dropin_binary(native, exe, cfg, tmp_path):
    z = G.load(f"hbma_{cfg.name}.npz")
    mv, mad, gm, rmse, inl = _run(exe, cfg, tmp_path)
    assert np.array_equal(mv, z["mv"]) and np.array_equal(mad, z["mad"])
    # RANSAC through the wrapper draws its own samples (as the reference does); check what must
    # hold for any draw: ascending distinct inliers, gm = in-order f32 mean, rmse = motion.cpp:165-180
    assert len(inl) > cfg.blocks // 2 and np.all(np.diff(inl.astype(np.int64)) > 0)
    s = np.zeros(2, np.float32)
    for i in inl:
        s = (s + mv[i]).astype(np.float32)
    want_gm = (s * np.float32(1.0 / np.float32(len(inl)))).astype(np.float32)
    assert gm.tobytes() == want_gm.tobytes()
    acc = np.float32(0)
    for i in inl:
        d = (mv[i] - gm).astype(np.float32)
        acc = np.float32(acc + np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])))
    assert np.float32(np.sqrt(np.float32(acc / np.float32(len(inl))))).tobytes() == np.float32(rmse).tobytes()


def test_dropin_additions(native, oracle, tmp_path):
    """SvcSeedRansac makes a thread's draws repeat; Dct / QuantizeDequantize through the C++ wrappers."""
    _run("dropin_own_hdr", configs.C2, tmp_path)
    extra = _run.extra
    assert struct.unpack("<I", extra[:4])[0] == 1
    dw, dh = 48, 32
    yy, xx, cc = np.meshgrid(np.arange(dh), np.arange(dw), np.arange(3), indexing="ij")
    bgr = ((xx * 7 + yy * 13 + cc * 29) & 255).astype(np.uint8)
    planes = np.frombuffer(extra, np.float32, 3 * dw * dh, 4).reshape(3, dh, dw)
    ref = oracle.dct_frame_f64(bgr, 8, 8)
    assert (np.abs(planes - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all()
    q = np.frombuffer(extra, np.float32, 3 * dw * dh, 4 + 4 * 3 * dw * dh).reshape(3, dh, dw)
    assert q.tobytes() == oracle.quant(planes, 640).tobytes()


def test_host_entry_points_from_several_threads(native):
    """include/svc_hip.h:476: the host-pointer entry points (what libsvc_motion.so's reference signatures call) are thread-safe, each calling
    thread with its own staging buffers and stream.  Six threads, each its own frame pair of its own size, 25 rounds of
    hbma_host -> ransac_host -> dct_quant_host at once (ctypes drops the GIL inside a call): every result equals the one the same call
    gave alone; an error text raised on one thread never shows on another."""
    import threading
    from scalable_video_codec_amd import configs, synth
    import torch
    shapes = [(320, 192, 3), (256, 128, 2), (416, 240, 3), (192, 192, 1), (640, 352, 3), (208, 112, 2)]
    work = []
    for i, (w, h, levels) in enumerate(shapes):
        rng = np.random.default_rng(100 + i)
        f0 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        f1 = np.roll(f0, (1 + i % 3, 2), (0, 1))
        pyr = [[p.numpy() for p in synth.build_pyramid(synth.bgr_to_y(torch.from_numpy(f)), levels)] for f in (f0, f1)]
        blocks = (w // 16) * (h // 16)
        smp = (np.arange(native.ransac_iter_count(), dtype=np.uint32) * 977) % blocks
        types = (np.arange(blocks, dtype=np.uint32) % 5 == 0).astype(np.uint32)

        def run(pyr=pyr, f1=f1, smp=smp, types=types):
            mv, mad = native.hbma_host(pyr[0], pyr[1], 8, 16, 16)
            gm, rmse, inl = native.ransac_host(mv, smp)
            co = native.dct_quant_host(f1, 8, types, 16, 1, 640)
            return mv.tobytes(), mad.tobytes(), gm.tobytes(), np.float32(rmse).tobytes(), inl.tobytes(), co.tobytes()
        work.append((run, run()))  # the result of the call made alone
    errors = []

    def worker(k):
        run, want = work[k]
        try:
            for r in range(25):
                if r % 7 == 3:  # a refused call in between: its message belongs to this thread only
                    with pytest.raises(native.SvcError):
                        native.hbma_host([np.zeros((16, 16), np.uint8)], [np.zeros((16, 16), np.uint8)], 8, 0, 16)
                got = run()
                if got != want:
                    errors.append((k, r, [i for i, (a, b) in enumerate(zip(got, want)) if a != b]))
                    return
        except Exception as ex:  # noqa: BLE001
            errors.append((k, "exception", repr(ex)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(work))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
