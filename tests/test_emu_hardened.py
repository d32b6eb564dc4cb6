"""The CPU SIMT emulator's hardened modes ("make the emulator able to see it").

  * sanitizer build (g++ -fsanitize=address,undefined, process under LD_PRELOAD=libasan.so): every `__shared__` array is a static
    object of exactly its declared bytes with red zones, every device buffer (torch CPU tensor) has red zones, the LDS-DMA copies are
    checked.  A canary kernel that writes one float past its LDS IS reported (at the kernel's source line); the kernel suites run clean.
    First full run (round 5, profiles/r05_emu_hardened.log): 336 cases clean, ONE finding -- the whole-block AMP kernel (amp_block.hip, off
    by default since round 4) evaluated its end-of-sequence fix-up values through LDS rows outside its tile (values it then discarded);
    the kernel was removed.
  * concurrent blocks (-DSVCMI_EMU_TLS build, SVCMI_EMU_BLOCKS=K): K blocks of a launch resident at once with private LDS, interleaved
    round by round.  A canary whose result depends on the block order changes; the kernel suite gives the same results.

Cost control: the canaries build only the scheduler (seconds).  The kernel suites under the two modes are sub-process runs with
xdist: SVCMI_HARDENED=quick (the LDS-tile kernels, 3 resident blocks: the default), =full (every emulator case under both modes,
~10 min on 8 cores incl. the 3-minute sanitizer build), =off (canaries only)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.emu.build_emu import asan_runtime, build_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEVEL = os.environ.get("SVCMI_HARDENED", "quick")
# the kernels with LDS tiles, halos and multi-problem launches
QUICK = "snake_conv_group or two_deep or snake_post"
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:redzone=128", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}


def _run_suite(env_extra, select, files=("tests/test_kernels_emu.py",)):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "pytest", *files, "-x", "-q", "-p", "no:cacheprovider", "-n", "8"]
    if select:
        cmd += ["-k", select]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3400)


def test_concurrent_blocks_canary_depends_on_the_interleaving(monkeypatch):
    lib = ctypes.CDLL(build_emu(tls=True, harness_only=True))
    n = 6
    res = {}
    for k in ("1", "2", "3"):
        monkeypatch.setenv("SVCMI_EMU_BLOCKS", k)
        flags, seen = np.zeros(n, np.int32), np.zeros(n, np.int32)
        assert lib.emu_selftest_order(flags.ctypes.data_as(ctypes.c_void_p), seen.ctypes.data_as(ctypes.c_void_p), n) == 0
        res[k] = seen.tolist()
    assert res["1"] == [0, 0, 0, 0, 0, -1]                   # one after another: the right neighbour has not run yet
    assert res["2"][0] == 2 and res["2"] != res["1"]         # blocks 0 and 1 resident together: block 0 sees block 1's word
    assert res["3"][:2] == [2, 3]


@pytest.mark.skipif(asan_runtime() is None, reason="no libasan.so for this gcc")
def test_sanitizer_build_reports_a_write_one_float_past_the_lds_array():
    san = build_emu(sanitize=True, harness_only=True)
    env = dict(os.environ, LD_PRELOAD=asan_runtime(), **SAN_ENV)
    probe = ("import ctypes, numpy as np, sys; lib = ctypes.CDLL(%r); out = np.zeros(128, np.float32); "
             "rc = lib.emu_selftest_lds(out.ctypes.data_as(ctypes.c_void_p), 2, int(sys.argv[1])); print('rc', rc, out[:3].tolist())" % san)
    ok = subprocess.run([sys.executable, "-c", probe, "63"], env=env, capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0 and "rc 0" in ok.stdout, ok.stdout + ok.stderr[-2000:]
    bad = subprocess.run([sys.executable, "-c", probe, "64"], env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "AddressSanitizer" in bad.stderr and "global-buffer-overflow" in bad.stderr, bad.stderr[-2000:]
    assert "canary_lds_kernel" in bad.stderr                     # reported at the kernel's own source line


@pytest.mark.skipif(LEVEL == "off", reason="SVCMI_HARDENED=off")
def test_kernel_suite_with_three_resident_blocks():
    """The emulator kernel cases again with 3 blocks of each launch resident at once (private LDS, round-robin interleaving): same
    references, same tolerances, the bit-identity assertions included."""
    build_emu(tls=True)
    r = _run_suite({"SVCMI_EMU_BUILD": "tls", "SVCMI_EMU_BLOCKS": "3"}, None if LEVEL == "full" else QUICK)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.skipif(LEVEL != "full", reason="SVCMI_HARDENED=full runs every emulator case under ASan + UBSan (~10 min; the round-5 run: profiles/r05_emu_hardened.log)")
@pytest.mark.skipif(asan_runtime() is None, reason="no libasan.so for this gcc")
def test_every_emulator_case_is_clean_under_the_sanitizers():
    build_emu(sanitize=True)
    r = _run_suite(dict(SAN_ENV, LD_PRELOAD=asan_runtime(), SVCMI_EMU_BUILD="san"), None, files=("tests/test_kernels_emu.py", "tests/test_engine_emu.py"))
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout[-3000:] + r.stderr[-3000:]
