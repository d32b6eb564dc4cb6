"""Build the TEST-ONLY CPU emulation of the kernels: same csrc/*.hip sources, g++ -DSVCMI_EMU."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "whisper-vits-svc_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libsvcmi_emu.so")


def build_emu(force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "hip_emu.*")) + \
        [os.path.join(ROOT, "include", "svcmi.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    flags = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-Wno-psabi"]
    jobs = [(flags + ["-DSVCMI_EMU", "-I", HERE, "-I", CSRC, "-x", "c++", "-c", s, "-o", os.path.join(OUT_DIR, os.path.basename(s) + ".o")])
            for s in srcs]
    jobs.append(flags + ["-I", HERE, "-c", os.path.join(HERE, "hip_emu.cpp"), "-o", os.path.join(OUT_DIR, "hip_emu.o")])
    procs = [subprocess.Popen(j) for j in jobs]          # one compiler per file, in parallel (a fresh checkout builds this once per test run)
    if any([p.wait() for p in procs]):
        raise RuntimeError("emulator build failed")
    objs = [j[-1] for j in jobs[:-1]]
    o = jobs[-1][-1]
    subprocess.run(["g++", "-shared", "-o", OUT] + objs + [o], check=True)
    return OUT


if __name__ == "__main__":
    print(build_emu(force=True))
