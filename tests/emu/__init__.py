"""TEST-ONLY: CPU SIMT emulation of the svcmi kernels (see hip_emu.h)."""
import functools


@functools.lru_cache(maxsize=1)
def emu_ops():
    from svcmi import _lib
    from svcmi.ops import Ops
    from .build_emu import build_emu
    lib = _lib.load_library(build_emu())
    ops = Ops(lib)
    assert ops.build == "emu"
    return ops
