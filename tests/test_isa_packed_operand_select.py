"""The shipped binary holds no packed-fp32 instruction whose LOW lane takes the HIGH half of src1 (op_sel:[.,1,..]).

Round 6: on MI355X `v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1,0]` computes lanes 48..63 of its low result wrongly while another
wave of the SIMD executes `v_mfma_f32_16x16x32_f16` or `_bf16` -- found as SnakeAlias outputs of one clip differing by 1e-2 beside the fp16 fused
half-step of another clip in flight, reproduced with two ten-line kernels (scripts/probes/pkfma_mfma_corun.hip, profiles/r06y_pkfma_mfma_corun.log).
The select on src0 (op_sel:[1,0,0]), op_sel_hi and the other matrix-core shapes are not affected, so the kernels keep the swizzled value in src0
(snake_math.h, amp_fused.hip) and avoid the libm routine the compiler vectorised into the form (conv_gemm_body.h: Mish).  This test disassembles
every gfx950 code object of libsvcmi.so and fails on the first such instruction, whoever generated it (our packed code or the SLP vectoriser).
CPU only: llvm-objdump from the ROCm image."""
import os, re, struct, subprocess, tempfile
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "whisper-vits-svc_amd", "svcmi", "libsvcmi.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
SRC1_HIGH_FOR_LOW_LANE = re.compile(r"op_sel:\[[01],1")


def code_objects(path):
    """(triple, bytes) of every device code object bundled into the shared library (one clang offload bundle per translation unit)."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "amdgcn" in triple and size:
                out.append((triple, data[i + off:i + off + size]))
        pos = i + 24


def scan(disassembly):
    """-> (packed fp32 instructions, {kernel: offending instructions})"""
    name, n, bad = None, 0, {}
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
        elif PACKED.search(line):
            n += 1
            if SRC1_HIGH_FOR_LOW_LANE.search(line):
                bad.setdefault(name, []).append(line.split("//")[0].strip())
    return n, bad


def test_the_scanner_sees_the_form():
    txt = ("0000000000001000 <k>:\n\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0] // 000000001000: D3B04000\n"
           "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,0,0]\n\tv_pk_add_f32 v[0:1], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n"
           "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]\n\tv_pk_fma_f32 v[0:1], s[2:3], v[4:5], v[0:1] op_sel:[1,1,0]\n")
    n, bad = scan(txt)
    assert n == 5 and len(bad["k"]) == 3


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image")
def test_no_packed_fp32_instruction_selects_the_high_half_of_src1():
    import importlib.util
    spec = importlib.util.spec_from_file_location("svcmi_build", os.path.join(ROOT, "whisper-vits-svc_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.build_hip() == LIB          # (rebuilds only when a source is newer than the library)
    objs = code_objects(LIB)
    assert len(objs) >= 8, f"{len(objs)} device code objects found in {LIB}"
    from concurrent.futures import ThreadPoolExecutor

    def disassemble(item):
        triple, blob = item
        assert "gfx950" in triple, triple
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            return scan(subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout)

    total, offenders = 0, {}
    with ThreadPoolExecutor(max_workers=4) as pool:
        for n, bad in pool.map(disassemble, objs):
            total += n
            offenders.update(bad)
    assert total > 10000, f"only {total} packed fp32 instructions seen: the disassembly did not work"
    assert not offenders, "packed fp32 instructions with the src1 half-select (MI355X: wrong in lanes 48..63 beside v_mfma_f32_16x16x32_f16/bf16): " + \
        "; ".join(f"{k}: {len(v)} e.g. {v[0]}" for k, v in list(offenders.items())[:5])
