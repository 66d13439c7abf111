"""The drop-in boundary, checked without a GPU: the C-ABI library loads and exports
every symbol include/svc_hip.h declares; the C++ library exports the reference's own
mangled names (libs/motion.hpp:100-153); argument validation answers before any device
work; and with no GPU the product fails loudly instead of falling back."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from scalable_video_codec_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "svc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svc_hip_\w+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = native.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/svc_hip.h but not exported"
    assert set(names) == set(native.SIGNATURES), set(names) ^ set(native.SIGNATURES)
    assert lib.svc_hip_abi_version() == 5


def test_clip_header_symbols_are_exported():
    """include/svc_clip.h (C handle API of svc::ClipEncoder) vs libsvc_motion.so vs the ctypes binding."""
    from scalable_video_codec_amd import clip
    lib = clip.load()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "svc_clip.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(svc_clip_\w+)\s*\(", text)) - {"svc_clip_halo_fn"})
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/svc_clip.h but not exported"
    assert set(names) == set(clip.SIGNATURES), set(names) ^ set(clip.SIGNATURES)
    assert ctypes.sizeof(clip.ClipConfig) == 136 and ctypes.sizeof(clip.ClipInfo) == 80  # struct layout of the header
    out = subprocess.check_output(["nm", "-DC", "--defined-only", native.MOTION_LIB_PATH], text=True)
    for sym in ("svc::ClipEncoder::ClipEncoder(svc::ClipEncoderConfig const&)", "svc::ClipEncoder::Step(bool)",
                "svc::PlanShard(unsigned int, unsigned int, unsigned int)", "svc::ClipEncoder::SetComm(void*)"):
        assert sym in out, sym


def test_clip_api_without_a_gpu():
    """Host logic of the C++ driver: the shard plan, and a loud failure (no fallback) when no device is there."""
    import torch
    from scalable_video_codec_amd import clip, configs
    assert clip.plan_shard(300, 8, 0) == (0, 38, 37, 1) and clip.plan_shard(300, 8, 3) == (114, 38, 38, 114)
    assert clip.plan_shard(300, 8, 4) == (152, 37, 37, 152) and clip.plan_shard(300, 8, 7) == (263, 37, 37, 263)
    assert clip.plan_shard(2, 2, 0) == (0, 1, 0, 1) and clip.plan_shard(2, 2, 1) == (1, 1, 1, 1)
    with pytest.raises(clip.ClipError):
        clip.plan_shard(10, 2, 2)
    if not torch.cuda.is_available():
        with pytest.raises(clip.ClipError, match="no ROCm-capable device|hipStreamCreate|invalid device"):
            clip.Clip(configs.C3, 300)
        with pytest.raises(native.SvcError):
            clip.comm_create(bytes(128), 0, 1)  # RCCL cannot bring up a communicator without a device


def test_motion_library_exports_reference_symbols():
    """Itanium-mangled names of the reference's public functions (nm -C of the compiled
    reference object) -- what apps/encoder.cpp links against."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", native.MOTION_LIB_PATH], text=True)
    for sym in ("_Z26EstimateMotionHierarchicalPKPKhS2_jjjjjjP5Vec2fPf",
                "_Z35EstimateMotionHierarchical16x16Sse2PKPKhS2_jjjP5Vec2fPf",
                "_Z30EstimateMotionExhaustiveSearchPKhS0_jjjjjP5Vec2fPf",
                "_Z26EstimateGlobalMotionRansacPK5Vec2fj12RansacParamsPfPS_PSt6vectorIjSaIjEE",
                "_Z23EstimateGlobalMotionAvgPK5Vec2fj",                      # libs/motion.hpp:38
                "_Z36EstimateGlobalMotionExhaustiveSearchPKhS0_jjjP5Vec2fPf",  # :45-49
                "_Z32EstimateGlobalMotionHierarchicalPKPKhS2_jjjjP5Vec2f"):     # :55-59
        assert sym in out, sym
    ctypes.CDLL(native.MOTION_LIB_PATH)  # resolves its libsvc_hip.so dependency via $ORIGIN


def test_motion_library_exports_stream_encoder():
    """include/svc/stream_encoder.hpp: the batched host-memory encoder is part of the C++ layer."""
    out = subprocess.check_output(["nm", "-DC", "--defined-only", native.MOTION_LIB_PATH], text=True)
    for sym in ("svc::StreamEncoder::StreamEncoder(svc::StreamEncoderConfig const&)", "svc::StreamEncoder::~StreamEncoder()",
                "svc::StreamEncoder::Encode(unsigned char const*, unsigned int, std::function<void (svc::EncodedBatch const&)> const&)",
                "svc::StreamEncoder::padded_width() const", "svc::StreamEncoder::padded_height() const"):
        assert sym in out, sym


def test_pyramid_bytes_and_iter_count():
    assert native.pyramid_bytes(1920, 1088, 3) == 2741760
    assert native.pyramid_bytes(3840, 2160, 4) == 11016000
    assert native.pyramid_bytes(352, 288, 1) == 101376
    assert native.ransac_iter_count() == 7                      # defaults, SURVEY.md 3.3
    assert native.ransac_iter_count(subset_sz=3) == 35


def test_hbma_kernel_choice_is_queryable():
    """svc_hip_hbma_kernel_name: the dispatch of svc_hip_hbma_pairs as a string (bench.py labels its roofline with it)."""
    name = native.hbma_kernel_name
    assert name(4, 3840, 2160, 8) == "hbma_tiled16_kernel"       # C5: the reference's default search, levels 2 and 1 from LDS tiles
    assert name(4, 1920, 1088, 8) == "hbma_tiled16_kernel"       # C3b
    assert name(3, 1920, 1088, 8) == "hbma_fused_kernel"         # C3 (headline): lane per block
    assert name(4, 1920, 1088, 16) == "hbma_fused_kernel"        # r_top 2
    assert name(4, 1936, 1088, 8) == "hbma_fused_kernel"         # top plane 242 wide, not whole dwords: taken since round 4 (2 x 2 top blocks)
    assert name(4, 720, 576, 8) == "hbma_fused_kernel"           # PAL with the reference's default build
    assert name(1, 352, 288, 8) == "hbma_wave_level_kernel"      # C1: one level is EBMA-shaped
    assert name(3, 1920, 1088, 8, 8, 8) == "hbma_fused_kernel"
    assert name(3, 1920, 1088, 8, 16, 8) == "hbma_wave_level_kernel"   # non-square blocks
    assert name(4, 3840, 2160, 8, flags=native.HBMA_FORCE_LANE) == "hbma_fused_kernel"
    assert name(4, 3840, 2160, 8, flags=native.HBMA_FORCE_WAVE_PER_BLOCK) == "hbma_wave_level_kernel"
    assert name(4, 3840, 2160, 8, flags=native.HBMA_FORCE_TILED) == "hbma_tiled16_kernel"
    for args, flags, status in (((3, 1920, 1088, 8), native.HBMA_FORCE_TILED, native.SVC_ERR_UNSUPPORTED),
                                ((3, 1920, 1088, 8, 16, 8), native.HBMA_FORCE_LANE, native.SVC_ERR_UNSUPPORTED),   # non-square blocks
                                ((4, 1920, 1080, 8), native.HBMA_AUTO, native.SVC_ERR_INVALID_ARG),     # 1080 % 16
                                ((5, 1920, 1088, 8), native.HBMA_AUTO, native.SVC_ERR_INVALID_ARG)):    # range < 2^(L-1)
        with pytest.raises(native.SvcError) as e:
            name(*args, flags=flags)
        assert e.value.status == status, (args, flags)


def test_preconditions_are_statuses_not_ub():
    """The reference only asserts these (libs/motion.cpp:417-433); here they are
    SVC_ERR_INVALID_ARG, raised before any device call (so this runs without a GPU)."""
    z = [np.zeros((64, 64), np.uint8)]
    for args in ((z, z, 8, 16, 24),          # frame % block (motion.cpp:428-429)
                 (z, z, 8, 0, 16),           # zero block (:423)
                 (z * 3, z * 3, 2, 16, 16),  # range < 2^(L-1) (:433)
                 (z * 2, z * 2, 8, 3, 16)):  # block not divisible by 2^(L-1)
        with pytest.raises(native.SvcError) as e:
            native.hbma_host(*args)
        assert e.value.status == native.SVC_ERR_INVALID_ARG, args[2:]
    with pytest.raises(native.SvcError) as e:
        native.quant_host(np.ones(4, np.float32), 0)
    assert e.value.status == native.SVC_ERR_INVALID_ARG
    with pytest.raises(native.SvcError) as e:
        native.ransac_host(np.zeros((4, 2), np.float32), np.array([4], np.uint32))  # index == N: the reference's OOB draw
    assert e.value.status == native.SVC_ERR_INVALID_ARG
    with pytest.raises(native.SvcError) as e:
        native.dct_host(np.zeros((30, 32, 3), np.uint8), 8)
    assert e.value.status == native.SVC_ERR_INVALID_ARG


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    z = [np.zeros((64, 64), np.uint8)]
    with pytest.raises(native.SvcError) as e:
        native.hbma_host(z, z, 8, 16, 16)
    assert e.value.status == native.SVC_ERR_NO_DEVICE
    with pytest.raises(native.SvcError) as e:
        native.dct_host(np.zeros((32, 32, 3), np.uint8), 8)
    assert e.value.status == native.SVC_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under scalable_video_codec_amd/, include/ or compat/ may reference it."""
    for base in ("scalable_video_codec_amd", "include", "compat"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".inc")):
                    text = open(os.path.join(dp, f)).read()
                    assert "svc_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
