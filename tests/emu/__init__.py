"""TEST-ONLY: CPU SIMT emulation of the svcmi kernels (see hip_emu.h)."""
import functools
import os


@functools.lru_cache(maxsize=1)
def emu_ops():
    """SVCMI_EMU_BUILD = san | tls selects the sanitizer / concurrent-blocks build of the emulator (tests/test_emu_hardened.py runs the
    kernel tests in sub-processes under both); default: the plain -O2 build."""
    from svcmi import _lib
    from svcmi.ops import Ops
    from .build_emu import build_emu
    kind = os.environ.get("SVCMI_EMU_BUILD", "")
    lib = _lib.load_library(build_emu(sanitize=kind == "san", tls=kind == "tls"))
    ops = Ops(lib)
    assert ops.build == "emu"
    return ops
