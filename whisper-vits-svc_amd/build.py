"""Build libsvcmi.so (HIP, gfx950) in-tree:  python whisper-vits-svc_amd/build.py

hipcc cross-compiles without a GPU; the .so lands next to the Python package
(whisper-vits-svc_amd/svcmi/libsvcmi.so) so it travels with the source tree.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "svcmi", "libsvcmi.so")


# Per-file compiler flags.  amp_fused.hip: MFMA accumulators in ARCHITECTURAL registers -- in snake_conv16_group_kernel they are dead
# during the activation phase that sets the kernel's VGPR count, while accumulation registers come ON TOP of it in the unified file
# (117 + 32 -> occupancy 3 instead of 4).  The file holds no other matrix-core kernel.
# conv_gemm.hip (round 5, the pinned K loop of conv_gemm_body.h): with the fragment requests and DMA issues tied into the MFMA stream the
# accumulator-file form rotates three accumulator tuples through 16 v_accvgpr copies (+ s_nop 5) at the top of every K-step; in
# architectural registers there is nothing to rotate and the unified register count is the same (92 / 123 against 100 / 124).
# SVCMI_ACC_IN_VGPRS makes svcmi_pin name a "v" register there.  The 16-bit kernels (conv_gemm_lp.hip) keep the accumulator file.
# SVCMI_DMA_M0_RAW (round 6, conv_gemm.hip only): the LDS-DMA statements set M0 without saving / restoring it (2 scalar moves less per
# 1-KiB piece).  hipcc reserves M0 and rejects an "m0" clobber, so this is sound only where the compiler itself never uses M0:
# tests/test_isa_k_loop.py checks the assembly of this translation unit for exactly that.
FILE_FLAGS = {"amp_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
              "conv_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-DSVCMI_ACC_IN_VGPRS=1", "-DSVCMI_DMA_M0_RAW=1"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = list(srcs) + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "svcmi.h")]
    return any(os.path.getmtime(s) > t for s in deps)


def build_hip(force=False, verbose=False):
    """One hipcc -c per kernel file, in parallel, then one link: ~25 s instead of ~75 s for the eight translation units."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sources()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not force and not needs_build(OUT, srcs):
        ex_src, ex = os.path.join(HERE, "examples", "stage_host.cpp"), os.path.join(HERE, "examples", "stage_host")
        if not os.path.exists(ex) or max(os.path.getmtime(ex_src), os.path.getmtime(OUT)) > os.path.getmtime(ex):      # the example has its own source
            build_example(hipcc, verbose)
        return OUT
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + flags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    audit_packed_operand_select(verbose)
    build_example(hipcc, verbose)
    return OUT


def audit_packed_operand_select(verbose=False):
    """The fresh library must not hold a packed-fp32 instruction with the src1 half-select (isa_audit.py, DESIGN 4.6: MI355X computes it wrongly
    beside the 16x16x32 16-bit matrix-core shapes; hipcc emits it from innocent source).  ~10 s; SVCMI_SKIP_ISA_AUDIT=1 skips, and so does a
    missing llvm-objdump (the test suite then skips its copy of the check as well)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("svcmi_isa_audit", os.path.join(HERE, "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if os.environ.get("SVCMI_SKIP_ISA_AUDIT") or not os.path.exists(mod.OBJDUMP):
        return
    n_obj, n_packed, offenders = mod.audit(OUT)
    if verbose:
        print(f"isa audit: {n_obj} code objects, {n_packed} packed fp32 instructions, {sum(len(v) for v in offenders.values())} with the src1 half-select", flush=True)
    if offenders:
        raise RuntimeError("libsvcmi.so holds packed fp32 instructions with the src1 half-select (op_sel:[.,1,..]) -- rewrite the source so the "
                           "swizzled operand is src0 (snake_math.h, svcmi_hsum2): " + mod.describe(offenders))


def build_example(hipcc, verbose=False):
    """examples/stage_host: the C++ host without Python (packed model -> svcmi_synth_infer_fwd), linked against the in-tree library.
    Best effort: the library build every import and test depends on must not fail because the example does not compile or link
    (the two GPU tests that run the example assert that the binary exists and say who builds it)."""
    try:
        return _build_example(hipcc, verbose)
    except (subprocess.CalledProcessError, OSError) as e:
        print(f"svcmi build: examples/stage_host not built ({e})", file=sys.stderr)
        return None


def _build_example(hipcc, verbose=False):
    src = os.path.join(HERE, "examples", "stage_host.cpp")
    exe = os.path.join(HERE, "examples", "stage_host")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", src, "-o", exe, "-L", os.path.dirname(OUT), "-lsvcmi",
           "-Wl,-rpath,$ORIGIN/../svcmi"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return exe


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
