"""The shipped binary holds no packed-fp32 instruction whose LOW lane takes the HIGH half of src1 (op_sel:[.,1,..]).

Round 6: on MI355X `v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1,0]` computes lanes 48..63 of its low result wrongly while another
wave of the SIMD executes `v_mfma_f32_16x16x32_f16` or `_bf16` -- found as SnakeAlias outputs of one clip differing by 1e-2 beside the fp16 fused
half-step of another clip in flight, reproduced with two ten-line kernels (scripts/probes/pkfma_mfma_corun.hip, profiles/r06y_pkfma_mfma_corun.log).
The select on src0 (op_sel:[1,0,0]), op_sel_hi and the other matrix-core shapes are not affected, so the kernels keep the swizzled value in src0
(snake_math.h, amp_fused.hip) and avoid the libm routine the compiler vectorised into the form (conv_gemm_body.h: Mish).  This test disassembles
every gfx950 code object of libsvcmi.so and fails on the first such instruction, whoever generated it (our packed code or the SLP vectoriser).
CPU only: llvm-objdump from the ROCm image."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "whisper-vits-svc_amd", "svcmi", "libsvcmi.so")


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "whisper-vits-svc_amd", rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


A = _load("svcmi_isa_audit", "isa_audit.py")       # the scanner build.py runs on every fresh library
scan, OBJDUMP = A.scan, A.OBJDUMP


def test_the_scanner_sees_the_form():
    txt = ("0000000000001000 <k>:\n\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0] // 000000001000: D3B04000\n"
           "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,0,0]\n\tv_pk_add_f32 v[0:1], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n"
           "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]\n\tv_pk_fma_f32 v[0:1], s[2:3], v[4:5], v[0:1] op_sel:[1,1,0]\n")
    n, bad = scan(txt)
    assert n == 5 and len(bad["k"]) == 3


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image")
def test_no_packed_fp32_instruction_selects_the_high_half_of_src1():
    assert _load("svcmi_build", "build.py").build_hip() == LIB          # (rebuilds only when a source is newer than the library)
    n_obj, total, offenders = A.audit(LIB)
    assert n_obj >= 8, f"{n_obj} device code objects found in {LIB}"
    assert total > 10000, f"only {total} packed fp32 instructions seen: the disassembly did not work"
    assert not offenders, "packed fp32 instructions with the src1 half-select (MI355X: wrong in lanes 48..63 beside v_mfma_f32_16x16x32_f16/bf16): " + \
        A.describe(offenders)
