"""Host-side checks that need no GPU: the HIP library builds/loads, exports every symbol declared in
include/svcmi.h, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_so():
    import importlib.util
    spec = importlib.util.spec_from_file_location("svcmi_build", os.path.join(ROOT, "whisper-vits-svc_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_hip()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "svcmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svcmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_so):
    lib = ctypes.CDLL(hip_so)
    syms = header_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/svcmi.h but not exported"
    lib.svcmi_build_info.restype = ctypes.c_char_p
    assert lib.svcmi_build_info() == b"hip:gfx950"
    from svcmi import _lib
    assert lib.svcmi_abi_version() == _lib.ABI_VERSION == 22


def test_binding_table_matches_header():
    from svcmi import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_argument_validation_needs_no_gpu(hip_so):
    from svcmi import _lib
    lib = _lib.load_library(hip_so)
    assert lib.svcmi_conv_gemm_f32(None, None) == -1                    # SVCMI_EINVAL, nothing launched
    d = _lib.ConvDesc()
    assert lib.svcmi_conv_gemm_f32(ctypes.byref(d), None) == -1
    assert lib.svcmi_source2wav_i16(None, None, 10, None) == -1
    # grouped / retrieval / decoding entry points reject bad arguments before touching the device
    descs = (_lib.ConvDesc * 3)()
    assert lib.svcmi_conv_gemm_group_f32(descs, 0, None) == -1 and lib.svcmi_conv_gemm_group_f32(descs, 4, None) == -1
    assert lib.svcmi_conv_gemm_group_f32(descs, 2, None) == -1                    # null operands inside the descriptors
    assert lib.svcmi_snake_conv_group_f32((_lib.SnakeConvDesc * 3)(), 0, None, 1, 10, 20, 20, None) == -1
    # the narrow stages' half-step on the fp16 matrix cores: same descriptors; 10 / 20 channels, fp16 modes only, float4-aligned rows
    sd = (_lib.SnakeConvDesc * 3)()
    assert lib.svcmi_snake_conv_group_lp(sd, 0, None, 1, 10, 20, 20, 3, None) == -1 and lib.svcmi_snake_conv_group_lp(sd, 1, None, 1, 10, 20, 20, 3, None) == -1
    assert lib.svcmi_snake_conv_lp_supported(20, 20, 11, 5, 3) == 1 and lib.svcmi_snake_conv_lp_supported(10, 12, 3, 1, 8) == 1
    assert lib.svcmi_snake_conv_lp_supported(40, 40, 3, 1, 3) == 0 and lib.svcmi_snake_conv_lp_supported(20, 20, 3, 1, 2) == 0 \
        and lib.svcmi_snake_conv_lp_supported(20, 20, 5, 1, 3) == 0 and lib.svcmi_snake_conv_lp_supported(20, 20, 3, 6, 3) == 0
    fb = (ctypes.c_float * 4096)()
    base = ctypes.addressof(fb) + (-ctypes.addressof(fb)) % 16          # a 16-byte aligned address inside the buffer
    sd[0].x, sd[0].w, sd[0].y, sd[0].alpha_log, sd[0].beta_log = base, base + 1024, base + 2048, base + 4096, base + 4096
    sd[0].ldw, sd[0].ksize, sd[0].dilation, sd[0].alpha = 60, 3, 1, 1.0
    flt = ctypes.c_void_p(base + 8192)
    assert lib.svcmi_snake_conv_group_lp(sd, 1, flt, 1, 4, 20, 20, 2, None) == -2       # bf16 is not a mode of this kernel: SVCMI_EUNSUPPORTED
    assert lib.svcmi_snake_conv_group_lp(sd, 1, flt, 1, 4, 40, 40, 3, None) == -2       # 40 channels stay on the GEMM path
    sd[0].y = base + 2048 + 8
    assert lib.svcmi_snake_conv_group_lp(sd, 1, flt, 1, 4, 20, 20, 3, None) == -3       # float4 epilogue: SVCMI_EALIGN
    sd[0].y = base + 2048
    sd[0].ldw = 58
    assert lib.svcmi_snake_conv_group_lp(sd, 1, flt, 1, 4, 20, 20, 3, None) == -1       # ldw < ksize * ld / not a multiple of 4
    sd[0].ldw, sd[0].y = 60, base
    assert lib.svcmi_snake_conv_group_lp(sd, 1, flt, 1, 4, 20, 20, 3, None) == -1       # x == y: halo reads, not an in-place op
    # reduced-precision entry points: unknown precision / null descriptors / bad image geometry are rejected before any launch
    assert lib.svcmi_conv_gemm_lp(None, 1, None) == -1 and lib.svcmi_conv_gemm_lp(ctypes.byref(d), 0, None) == -1
    assert lib.svcmi_conv_gemm_lp(ctypes.byref(d), 7, None) == -1
    assert lib.svcmi_conv_gemm_group_lp(descs, 2, 1, None) == -1 and lib.svcmi_conv_gemm_group_lp(descs, 2, 9, None) == -1
    buf16 = (ctypes.c_float * 256)()
    p16 = ctypes.cast(buf16, ctypes.c_void_p)
    assert lib.svcmi_pack_weights_lp(None, 4, 32, 1, p16, 32, None) == -1            # null source
    assert lib.svcmi_pack_weights_lp(p16, 4, 32, 0, p16, 32, None) == -1             # fp32 is not a packable precision
    assert lib.svcmi_pack_weights_lp(p16, 4, 40, 2, p16, 48, None) == -1             # ldw16 must be a multiple of 32
    assert lib.svcmi_pack_weights_lp(p16, 4, 40, 2, p16, 32, None) == -1             # ldw16 < ldw
    assert lib.svcmi_block_mean_f32(None, 3, None, 16, None) == -1
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.svcmi_knn_blend_f32(p, 16, p, 16, p, 16, p, p, 16, 1, 16, 16, 9, 0.5, None) == -2   # k > 8: SVCMI_EUNSUPPORTED
    assert lib.svcmi_knn_blend_f32(p, 16, p, 16, p, 4, p, p, 16, 1, 4, 16, 5, 0.5, None) == -1   # k > n
    assert lib.svcmi_ivf_assign_f32(p, 16, p, 2, p, 1, 4, 16, p, None, None) == -1                # ldd < nlist
    assert lib.svcmi_ivf_assign_f32(p, 16, p, 4, p, 1, 4, 16, None, None, None) == -1             # no output
    assert lib.svcmi_ivf_assign_f32(p, 18, p, 4, p, 1, 4, 18, p, None, None) == -3                # d % 4: SVCMI_EALIGN
    assert lib.svcmi_ivf_blend_f32(p, 16, p, p, p, 16, p, 16, 1, 16, 9, 0.5, None, None, None) == -2   # k > 8
    assert lib.svcmi_ivf_blend_f32(p, 16, p, p, p, 16, p, 16, 1, 16, 0, 0.5, None, None, None) == -1   # k < 1
    assert lib.svcmi_ivf_blend_f32(p, 16, None, p, p, 16, p, 16, 1, 16, 1, 0.5, None, None, None) == -1  # no assignment
    assert lib.svcmi_segment_mean_f32(p, 16, p, p, p, 8, 2, 16, None) == -1                       # ldo < d
    assert lib.svcmi_viterbi_decode(p, p, p, p, p, 4, 4, 0, 360, 16, None) == -2                 # band > 15
    assert lib.svcmi_viterbi_decode(p, p, p, p, p, 4, 4, 10, 5, 0, None) == -1                   # empty bin range
    assert lib.svcmi_tune_set(b"no_such_knob", 1) == -1


def test_product_refuses_to_run_without_gpu(hip_so):
    import torch
    from svcmi import Ops, SvcmiError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(SvcmiError):
        Ops()


def test_missing_library_fails_loudly(tmp_path):
    from svcmi import SvcmiError, load_library
    with pytest.raises(SvcmiError):
        load_library(str(tmp_path / "libsvcmi.so"))


def test_product_never_imports_oracle_or_emulator():
    pkg = os.path.join(ROOT, "whisper-vits-svc_amd", "svcmi")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", src, flags=re.M), f
                assert "libsvcmi_emu" not in src, f
