"""Host-resident clip through the batched, PCIe-overlapped encoder (scalable_video_codec_amd/stream.py):
results must equal the resident ClipEncoder's, whatever the batch size."""
import numpy as np
import pytest
import torch

from scalable_video_codec_amd import configs, pipeline, stream, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch,wire", [(5, False), (16, False), (7, True)])
def test_stream_equals_resident(native, batch, wire):
    cfg = configs.ALL["C2-720p-3L-dct8"]
    n = 19
    dev = torch.device("cuda")
    clip = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
    frames = [clip.frame_bgr(t) for t in range(n)]
    pw, ph = cfg.padded
    ref = pipeline.ClipEncoder(cfg, n, dev, wire=wire)
    ref.load_frames([synth.pad_frame(f, pw, ph) for f in frames])
    ref.step()
    torch.cuda.synchronize()
    host = torch.stack(frames).cpu().numpy()
    enc = stream.HostStreamEncoder(cfg, batch=batch, device=dev, wire=wire)
    seen = 0
    held = None  # a yielded view stays valid until depth - 2 = 1 more batch has been yielded (stream.py)
    for out in enc.encode(host):
        if held is not None:
            assert np.array_equal(held[0], held[1]), "the previous batch's view changed inside its documented lifetime"
        held = (out["mv"], out["mv"].copy())
        assert ("header" in out) == (wire and out["first"] == 1)
        if "header" in out:
            assert np.frombuffer(out["header"], np.uint32).tolist() == [n - 1, cfg.width, cfg.height, pw - cfg.width, ph - cfg.height, 8, 8, 3]
        a = out["first"] - 1
        c = out["mv"].shape[0]
        assert np.array_equal(out["mv"], ref.mv[a:a + c].cpu().numpy())
        assert np.array_equal(out["types"], ref.types[a:a + c].cpu().numpy())
        assert np.array_equal(out["gm"], ref.gm[a:a + c].cpu().numpy())
        if wire:
            assert np.array_equal(out["records"], ref.records[a:a + c].cpu().numpy())
        else:
            assert np.array_equal(out["coeffs"], ref.coeffs[a:a + c].cpu().numpy())
        seen += c
    assert seen == n - 1


@pytest.mark.parametrize("batch,wire", [(6, False), (16, True)])
def test_cpp_stream_encoder_equals_resident(native, tmp_path, batch, wire):
    """The C++ host application (tests/dropin/stream_main.cpp against include/svc/stream_encoder.hpp) on a raw clip
    file: every output equals the resident path's."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "dropin", "stream_main")
    if not os.path.exists(exe):
        pytest.fail("tests/dropin/stream_main is not built (python -m scalable_video_codec_amd.build)")
    cfg = configs.ALL["C2-720p-3L-dct8"]
    n = 21
    dev = torch.device("cuda")
    clip = synth.SynthClip(cfg.width, cfg.height, n, cfg.seed, device=dev)
    frames = [clip.frame_bgr(t) for t in range(n)]
    pw, ph = cfg.padded
    ref = pipeline.ClipEncoder(cfg, n, dev, wire=wire)
    ref.load_frames([synth.pad_frame(f, pw, ph) for f in frames])
    ref.step()
    torch.cuda.synchronize()
    raw = tmp_path / "clip.raw"
    torch.stack(frames).cpu().numpy().tofile(raw)
    prefix = str(tmp_path / "out")
    r = subprocess.run([exe, str(raw), str(cfg.width), str(cfg.height), str(n), str(cfg.levels), str(cfg.dct_block),
                        "1" if wire else "0", str(batch), str(cfg.seed), prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    p = n - 1
    assert np.array_equal(np.fromfile(prefix + ".mv", np.float32).reshape(p, cfg.blocks, 2), ref.mv.cpu().numpy())
    assert np.array_equal(np.fromfile(prefix + ".types", np.int32).reshape(p, cfg.blocks), ref.types.cpu().numpy())
    assert np.array_equal(np.fromfile(prefix + ".gm", np.float32).reshape(p, 2), ref.gm.cpu().numpy())
    if wire:
        assert np.array_equal(np.fromfile(prefix + ".big", np.uint8).reshape(p, -1), ref.records.cpu().numpy())
        assert np.fromfile(prefix + ".hdr", np.uint32).tolist() == [n - 1, cfg.width, cfg.height, pw - cfg.width, ph - cfg.height, 8, 8, 3]
    else:
        assert np.array_equal(np.fromfile(prefix + ".big", np.float32).reshape(p, 3, ph, pw), ref.coeffs.cpu().numpy())


@pytest.mark.parametrize("shape", ["200 120 3 8", "416 234 2 16"])
def test_stream_encoder_used_at_random_agrees_with_itself(shape):
    """tests/dropin/stream_fuzz.cpp: random batch sizes (1-8), batches in flight (3-5), copy threads, pointer / Source entry points, encoders
    reused over clips of 2 to 17 frames in random order (short last batches, clips that end on a batch boundary), all three output forms --
    every clip encodes to the bytes of its first encoding (the results are a function of clip and seed alone, stream_encoder.hpp:12-14),
    and the three forms agree on motion fields and region ids.  The test above ties one such result to the resident path."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "dropin", "stream_fuzz")
    if not os.path.exists(exe):
        pytest.fail("tests/dropin/stream_fuzz is not built (python -m scalable_video_codec_amd.build)")
    r = subprocess.run([exe, *shape.split(), "40", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])
    assert "all equal" in r.stdout.splitlines()[-1]
