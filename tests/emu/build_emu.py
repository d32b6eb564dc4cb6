"""Build the TEST-ONLY CPU emulation of the kernels: same csrc/*.hip sources, g++ -DSVCMI_EMU.

Three builds (tests/emu/hip_emu.cpp):
  build_emu()               libsvcmi_emu.so       -O2, blocks of a launch one after another (the everyday emulator tests)
  build_emu(sanitize=True)  libsvcmi_emu_san.so   -O1 -fsanitize=address,undefined: `__shared__` arrays and device buffers with red zones
                                                  (load it in a process started with LD_PRELOAD=<libasan.so>: tests/test_emu_hardened.py)
  build_emu(tls=True)       libsvcmi_emu_tls.so   -DSVCMI_EMU_TLS: per-block copies of `__shared__` -> SVCMI_EMU_BLOCKS=K interleaves K blocks
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "whisper-vits-svc_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libsvcmi_emu.so")


def asan_runtime():
    """Path of libasan.so (for LD_PRELOAD), or None when this gcc has none."""
    try:
        p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:       # noqa: BLE001
        return None
    return os.path.realpath(p) if os.path.isabs(p) and os.path.exists(p) else None


def build_emu(force=False, sanitize=False, tls=False, harness_only=False):
    """harness_only: hip_emu.cpp alone (the scheduler + its self-test kernels, tests/test_emu_hardened.py) -- seconds instead of minutes."""
    kind = "san" if sanitize else ("tls" if tls else "")
    extra = os.environ.get("SVCMI_EMU_CXXFLAGS", "").split()      # build experiments (-DSVCMI_...=...): the same switches as scripts/build_variant.sh
    if extra:
        kind += "x"                                                # its own library / object directory, rebuilt every time
        force = True
    out = os.path.join(OUT_DIR, f"lib{'emu_harness' if harness_only else 'svcmi_emu'}{'_' + kind if kind else ''}.so")
    objdir = os.path.join(OUT_DIR, ("h" if harness_only else "") + kind) if (kind or harness_only) else OUT_DIR
    srcs = [] if harness_only else sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "hip_emu.*")) + \
        [os.path.join(ROOT, "include", "svcmi.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(objdir, exist_ok=True)
    flags = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-Wno-psabi", "-pthread"]
    if sanitize:
        # -fno-sanitize=alignment: the kernels' float4 / 16-byte vector accesses are checked by the hardware's own rules (the -m gpu tests),
        # and torch hands out 64-byte aligned tensors; signed overflow, shifts, bounds of static arrays and every memory access stay checked
        flags = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-psabi", "-pthread", "-fno-omit-frame-pointer",
                 "-fsanitize=address,undefined", "-fno-sanitize=alignment,vptr", "-fno-sanitize-recover=undefined"]
    if tls:
        flags.append("-DSVCMI_EMU_TLS")
    flags += extra
    jobs = [(flags + ["-DSVCMI_EMU", "-I", HERE, "-I", CSRC, "-x", "c++", "-c", s, "-o", os.path.join(objdir, os.path.basename(s) + ".o")])
            for s in srcs]
    jobs.append(flags + ["-DSVCMI_EMU", "-I", HERE, "-c", os.path.join(HERE, "hip_emu.cpp"), "-o", os.path.join(objdir, "hip_emu.o")])
    procs = [subprocess.Popen(j) for j in jobs]          # one compiler per file, in parallel (a fresh checkout builds this once per test run)
    if any([p.wait() for p in procs]):
        raise RuntimeError("emulator build failed")
    objs = [j[-1] for j in jobs]
    link = ["g++", "-shared", "-pthread", "-o", out] + objs
    if sanitize:
        link[1:1] = ["-fsanitize=address,undefined"]
    subprocess.run(link, check=True)
    return out


if __name__ == "__main__":
    import sys
    print(build_emu(force=True, sanitize="san" in sys.argv[1:], tls="tls" in sys.argv[1:]))
