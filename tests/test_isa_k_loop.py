"""The steady-state K loop of the fp32 implicit GEMM as the compiler emits it for gfx950 (no GPU needed: hipcc cross-compiles).

What the pinned placement of conv_gemm_body.h (SVCMI_GEMM_SPREAD) promises is a property of the INSTRUCTION STREAM: between two consecutive
matrix instructions of a K-step there is never a longer run of other instructions than one MFMA can cover for long, and no accumulator is
copied.  A compiler update or an innocent edit of the loop can silently undo that (round 5 saw both: 27-instruction clumps from the
scheduler, 16 v_accvgpr copies per K-step from the register allocator) without changing a single result bit, so the stream itself is
checked here.  Takes a few seconds; skipped where hipcc is absent."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "whisper-vits-svc_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not installed")


def _file_flags():
    sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
    import build
    return build.FILE_FLAGS.get("conv_gemm.hip", [])


def _steady_loop(asm, kernel_substr):
    """Instructions of the loop with the most MFMAs inside the kernel whose mangled name contains `kernel_substr` (tightest such loop that
    issues a whole tile of LDS-DMAs: the drain loop holds as many MFMAs and no DMA, and a backward branch of the prologue can enclose a
    straight-line K-step)."""
    src = asm.split("\n")
    start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l) and kernel_substr in l)
    end = next(i for i in range(start + 1, len(src)) if src[i].startswith(".Lfunc_end"))
    body = src[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            seg = body[labels[m.group(1)]:i + 1]
            n = (sum("v_mfma" in s for s in seg), sum(s.strip().startswith("buffer_load") for s in seg) >= 4, -len(seg))
            if best is None or n > best[0]:
                best = (n, seg)
    seg = [s.split(";")[0].strip() for s in best[1]]
    return [s for s in seg if s and not s.endswith(":")]


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    hip = d / "k_loop.hip"
    hip.write_text(f'#include "{CSRC}/conv_gemm_body.h"\n'
                   "namespace {\n"
                   "template __global__ void conv_gemm_kernel<1, 5, MODE_CHUNK, true, PREC_F32, 0>(ConvArgs);      // Whisper MLP GEMMs, 3-deep ring\n"
                   "template __global__ void conv_gemm_kernel<1, 5, MODE_CHUNK, true, PREC_F32, 2>(ConvArgs);      // ... with clips in flight: 2-deep ring\n"
                   "template __global__ void conv_gemm_kernel<1, 1, MODE_CHUNK, false, PREC_F32, 0>(ConvArgs);     // QKV, prior encoder, flow\n"
                   "template __global__ void conv_gemm_kernel<2, 2, MODE_CHUNK, false, PREC_F32, 0>(ConvArgs);     // 128 x 128: chip-filling launches\n"
                   "template __global__ void conv_gemm_kernel<1, 5, MODE_CHUNK, true, PREC_F32, 3, 8>(ConvArgs);   // round 6: the 64x80 wave tile on eight waves (128 x 80, opt-in)\n"
                   "}\n")
    out = d / "k_loop.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", str(hip), "-o", str(out)] + _file_flags()
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text()


# (kernel, MFMAs per K-step, issue slots one MFMA covers: 32 / 64 cycles at one instruction per ~4 cycles, longest tolerated run)
CASES = [("ILi1ELi5ELi0ELb1ELi0ELi0E", 40, 7, 14), ("ILi1ELi5ELi0ELb1ELi0ELi2E", 40, 7, 14),
         ("ILi1ELi1ELi0ELb0ELi0ELi0E", 16, 15, 24), ("ILi2ELi2ELi0ELb0ELi0ELi0E", 64, 15, 18),
         ("ILi1ELi5ELi0ELb1ELi0ELi3ELi8E", 40, 7, 14)]


@pytest.mark.parametrize("kernel,n_mfma,slots,longest", CASES, ids=["64x80", "64x80_ring2", "64x64", "128x128", "128x80_eight_waves"])
def test_k_loop_instruction_stream(asm, kernel, n_mfma, slots, longest):
    seg = _steady_loop(asm, kernel)
    seq = "".join("M" if s.startswith("v_mfma") else "x" for s in seg)
    assert seq.count("M") == n_mfma, "the steady K-step is one basic block holding every MFMA of the tile"
    gaps = [len(g) for g in seq.split("M")]
    gaps[0] += gaps.pop()                                   # the loop wraps around
    assert max(gaps) <= longest, f"a run of {max(gaps)} non-MFMA instructions between two MFMAs (pinned placement lost?): {gaps}"
    overflow = sum(max(0, g - slots) for g in gaps)
    assert overflow * 4 <= 0.06 * n_mfma * (32 if slots == 7 else 64), f"issue-slot overflow {overflow} slots: {gaps}"
    assert not any("accvgpr" in s for s in seg), "accumulator copies inside the K loop"
    assert sum(s.startswith("s_barrier") for s in seg) == 1 and sum(s.startswith("buffer_load") for s in seg) >= 4


def test_nothing_but_the_dma_statements_touches_m0(tmp_path):
    """build.py compiles conv_gemm.hip with SVCMI_DMA_M0_RAW: its LDS-DMA statements write M0 without saving / restoring it, which is sound
    only while the compiler itself never reads or writes M0 in this translation unit (hipcc reserves the register and cannot be told).
    Checked on the assembly of the WHOLE file: every mention of m0 is the destination of an `s_mov_b32 m0, s<N>` of our own statement."""
    out = tmp_path / "conv_gemm.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(CSRC, "conv_gemm.hip"), "-o", str(out)] + _file_flags()
    assert "-DSVCMI_DMA_M0_RAW=1" in cmd
    subprocess.run(cmd, check=True, capture_output=True)
    lines = [l.split(";")[0].strip() for l in out.read_text().split("\n")]
    m0 = [l for l in lines if re.search(r"\bm0\b", l) and not l.startswith(".")]
    assert len(m0) > 100, "the LDS-DMA statements are gone?"
    bad = [l for l in m0 if not re.fullmatch(r"s_mov_b32 m0, s\d+", l)]
    assert not bad, f"M0 used outside the DMA statements: {bad[:5]}"
    dma = [i for i, l in enumerate(lines) if l.startswith("buffer_load_dword") and l.endswith("lds")]
    assert dma and all(any(re.fullmatch(r"s_mov_b32 m0, s\d+", lines[j]) for j in range(max(0, i - 3), i)) for i in dma), "an LDS-DMA without its own M0 write"
