"""Packed-model files (include/svcmi.h: svcmi_packed_model_info / svcmi_packed_model_bind): the kernel-ready weights of one model as ONE
file a host without Python can mmap, upload and run through the stage-level entry points.

    python -m svcmi.tools pack --config configs/base.yaml --model sovits5.0.pth --out synth.svcmi
    python -m svcmi.tools pack --whisper whisper_pretrain/large-v2.pt --out whisper.svcmi

Layout: header | relocation table (byte offsets of the pointer fields inside the struct) | the C model struct with every pointer stored as
(byte offset into the arena + 1; 0 = NULL) | padding to 256 | the flat fp32 arena of ``svcmi.dist.pack_arena`` (tensors 256-byte aligned).
fp32 operands only (the 16-bit weight images of the reduced-precision modes are made on the device by svcmi_pack_weights_lp)."""
import ctypes
import struct

import numpy as np
import torch

from . import _lib, cmodel
from . import dist as D

MAGIC = b"SVCMIPK1"
KINDS = {_lib.SynthModel: 1, _lib.WhisperModel: 2}


def _pointer_fields(ctype, base=0):
    """Byte offsets of every c_void_p inside a ctypes struct / array type."""
    if ctype is ctypes.c_void_p:
        yield base
    elif isinstance(ctype, type) and issubclass(ctype, ctypes.Array):
        n, et = ctype._length_, ctype._type_
        for i in range(n):
            yield from _pointer_fields(et, base + i * ctypes.sizeof(et))
    elif isinstance(ctype, type) and issubclass(ctype, ctypes.Structure):
        for name, ft in ctype._fields_:
            yield from _pointer_fields(ft, base + getattr(ctype, name).offset)


def pack_model(weights, build=None):
    """``weights``: svcmi.weights.VitsWeights / WhisperWeights (any device).  Returns the file image as bytes."""
    from .weights import VitsWeights
    skel, arena = D.pack_arena(weights)
    arena = arena.cpu().contiguous()
    w = D.unpack_arena(skel, arena)                      # the same object, every tensor a view of `arena`
    build = build or (cmodel.synth_cmodel if isinstance(weights, VitsWeights) else cmodel.whisper_cmodel)
    cm = build(w, None, _lib.PREC_F32)
    st = cm.struct
    image = bytearray(ctypes.string_at(ctypes.addressof(st), ctypes.sizeof(st)))
    base, nbytes = arena.data_ptr(), arena.numel() * 4
    relocs = sorted(_pointer_fields(type(st)))
    for off in relocs:
        (v,) = struct.unpack_from("<Q", image, off)
        if v:
            if not (base <= v < base + nbytes):
                raise _lib.SvcmiError(f"pointer field at struct offset {off} does not point into the packed arena")
            v = v - base + 1
        struct.pack_into("<Q", image, off, v)
    head_bytes = 8 + 4 + 4 + 8 * 4
    arena_off = (head_bytes + 8 * len(relocs) + len(image) + 255) // 256 * 256
    head = MAGIC + struct.pack("<IIQQQQ", KINDS[type(st)], _lib.ABI_VERSION, len(image), len(relocs), arena_off, nbytes)
    body = head + struct.pack(f"<{len(relocs)}Q", *relocs) + bytes(image)
    return body + b"\0" * (arena_off - len(body)) + arena.numpy().tobytes()


def load_packed(path_or_bytes, lib, device):
    """The Python side of what examples/stage_host.cpp does in C++: file -> (CModel, kind) with the arena on ``device``."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    kind, off, nb = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
    if lib.svcmi_packed_model_info(buf, len(data), ctypes.byref(kind), ctypes.byref(off), ctypes.byref(nb)) != 0:
        raise _lib.SvcmiError("not a packed svcmi model (or written for another ABI version)")
    host = torch.from_numpy(np.frombuffer(data, dtype=np.uint8, count=nb.value, offset=off.value).copy())
    raw = torch.empty(nb.value + 256, dtype=torch.uint8, device=device)
    shift = (-raw.data_ptr()) % 256
    arena = raw[shift:shift + nb.value]
    arena.copy_(host)
    st = (_lib.SynthModel if kind.value == 1 else _lib.WhisperModel)()
    if lib.svcmi_packed_model_bind(buf, len(data), arena.data_ptr(), ctypes.byref(st), ctypes.sizeof(st)) != 0:
        raise _lib.SvcmiError("svcmi_packed_model_bind failed")
    return cmodel.CModel(st, [raw, arena]), kind.value
