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
    assert lib.svcmi_abi_version() == 14


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
