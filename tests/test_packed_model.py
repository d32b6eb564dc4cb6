"""Packed-model files (svcmi/packed.py, svcmi_packed_model_info / svcmi_packed_model_bind): what a host without Python loads.  On the CPU
SIMT emulator: pack the tiny synthesizer and the tiny Whisper encoder, bind them through the C entry points, run the stage-level forward
passes on the bound structs and compare with the facade's own (weights object -> struct) path bit for bit."""
import ctypes

import pytest
import torch

from svcmi import _lib, packed
from svcmi import weights as PW
from tests import engine_cases as E
from tests.emu import emu_ops
from workload import config as C, inputs as I, weights as W


@pytest.fixture(scope="module")
def ops():
    return emu_ops()


def test_header_and_relocations_are_checked(ops):
    w = PW.WhisperWeights(W.make_whisper_state(C.WHISPER_TINY_TEST), "cpu")
    data = packed.pack_model(w)
    assert data[:8] == b"SVCMIPK1"
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    kind, off, nb = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
    assert ops.lib.svcmi_packed_model_info(buf, len(data), ctypes.byref(kind), ctypes.byref(off), ctypes.byref(nb)) == 0
    assert kind.value == 2 and off.value % 256 == 0 and off.value + nb.value == len(data)
    bad = (ctypes.c_char * len(data)).from_buffer_copy(b"X" + data[1:])
    assert ops.lib.svcmi_packed_model_info(bad, len(data), None, None, None) == -1            # wrong magic
    assert ops.lib.svcmi_packed_model_info(buf, len(data) - 1, None, None, None) == -1        # truncated file
    st = _lib.SynthModel()                                                                   # wrong struct for this kind
    arena = torch.empty(nb.value + 256, dtype=torch.uint8)
    base = (arena.data_ptr() + 255) & ~255
    assert ops.lib.svcmi_packed_model_bind(buf, len(data), base, ctypes.byref(st), ctypes.sizeof(st)) == -1


def test_mangled_header_and_relocation_table_are_rejected(ops):
    """File-controlled 64-bit fields must not wrap the bounds checks: n_reloc = 2^61 makes 8 * n_reloc == 0, relocation
    offsets near 2^64 wrap `offset + 8`, an arena past the end of the file must be refused -- all before anything is read or written."""
    import struct
    w = PW.WhisperWeights(W.make_whisper_state(C.WHISPER_TINY_TEST), "cpu")
    data = bytearray(packed.pack_model(w))
    kind, abi, struct_bytes, n_reloc, arena_off, arena_bytes = struct.unpack_from("<IIQQQQ", data, 8)
    assert n_reloc > 0

    def info(img):
        buf = (ctypes.c_char * len(img)).from_buffer_copy(bytes(img))
        return ops.lib.svcmi_packed_model_info(buf, len(img), None, None, None), buf

    def header(**kw):
        f = dict(struct_bytes=struct_bytes, n_reloc=n_reloc, arena_off=arena_off, arena_bytes=arena_bytes)
        f.update(kw)
        img = bytearray(data)
        struct.pack_into("<IIQQQQ", img, 8, kind, abi, f["struct_bytes"], f["n_reloc"], f["arena_off"], f["arena_bytes"])
        return img

    assert info(data)[0] == 0
    for bad in (header(n_reloc=1 << 61), header(n_reloc=(1 << 61) + n_reloc), header(arena_off=(1 << 64) - 256),
                header(arena_bytes=(1 << 64) - arena_off + 16), header(arena_bytes=arena_bytes + 1), header(arena_off=40),
                header(n_reloc=(arena_off - 48 - struct_bytes) // 8 + 1)):
        assert info(bad)[0] == -1
    # relocation entries: past the struct, wrapping, misaligned
    st = _lib.WhisperModel()
    arena = torch.empty(arena_bytes + 256, dtype=torch.uint8)
    base = (arena.data_ptr() + 255) & ~255
    guard = bytes(ctypes.string_at(ctypes.addressof(st), ctypes.sizeof(st)))
    for v in ((1 << 64) - 4, (1 << 64) - 8, struct_bytes - 4, struct_bytes, 4):
        img = bytearray(data)
        struct.pack_into("<Q", img, 48, v)
        rc, buf = info(img)
        assert rc == 0                                        # the header itself is fine
        assert ops.lib.svcmi_packed_model_bind(buf, len(img), base, ctypes.byref(st), ctypes.sizeof(st)) == -1
    # an arena offset stored in a pointer field that lies outside the arena
    img = bytearray(data)
    (first,) = struct.unpack_from("<Q", img, 48)
    struct.pack_into("<Q", img, 48 + 8 * n_reloc + first, arena_bytes + 1)
    rc, buf = info(img)
    assert rc == 0 and ops.lib.svcmi_packed_model_bind(buf, len(img), base, ctypes.byref(st), ctypes.sizeof(st)) == -1
    del guard


def test_packed_whisper_runs_like_the_facade(ops):
    from svcmi.whisper.inference import load_model
    ck = W.make_whisper_state(C.WHISPER_TINY_TEST)
    wm = load_model(ck, "cpu", ops=ops)
    g = torch.Generator().manual_seed(1)
    mel = (torch.randn(1, 80, 60, generator=g) * 0.5).clamp(-1, 1.5)
    want = wm.encoder(mel)
    cm, kind = packed.load_packed(packed.pack_model(wm.weights), ops.lib, "cpu")
    assert kind == 2
    got = ops.whisper_encoder_fwd(cm, mel, None, 0.0)
    assert torch.equal(got, want)


def test_packed_synthesizer_runs_like_the_facade(ops):
    hp = C.tiny_hp()
    m, _ = E.make_model(hp, ops, "cpu")
    d = I.synth_clip(T=3, hp=hp, seed=9, B=1)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    want = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"])
    cm, kind = packed.load_packed(packed.pack_model(m._weights()), ops.lib, "cpu")
    assert kind == 1 and cm.struct.hop == 320 and cm.struct.n_stages == 5
    src2 = ops.pitch2source_fwd(cm, d["pit"], d["rand_ini"], d["src_noise"])
    assert torch.equal(src2.view(-1), src.view(-1))
    got = ops.synth_infer_fwd(cm, d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"].to(torch.int32), src2, d["enc_noise"])
    assert torch.equal(got, want)


def test_per_stage_entry_points_compose_to_the_whole(ops):
    """svcmi_text_encoder_fwd -> svcmi_flow_reverse_fwd -> svcmi_generator_fwd == svcmi_synth_infer_fwd, and the intermediate tensors are
    the ones `return_parts` hands out."""
    hp = C.tiny_hp()
    m, _ = E.make_model(hp, ops, "cpu")
    d = I.synth_clip(T=3, hp=hp, seed=4, B=2)
    lens = d["lengths"].clone()
    lens[-1] = 2
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    want, parts = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"], return_parts=True)
    wave, z_p, z = ops.synth_stages_fwd(m._cmodel(), d["ppg"], d["vec"], d["pit"], d["spk"], lens.to(torch.int32), src.view(2, -1), d["enc_noise"])
    assert torch.equal(wave, want)
    assert torch.equal(z_p.transpose(1, 2), parts["z_p"]) and torch.equal(z.transpose(1, 2), parts["z"])
    # a workspace smaller than svcmi_synth_workspace_bytes says is refused before anything is launched
    cm = m._cmodel()
    io = _lib.SynthIO()
    io.batch, io.t = 2, 3
    small = torch.empty(4096, dtype=torch.uint8)
    base = (small.data_ptr() + 255) & ~255
    assert ops.lib.svcmi_flow_reverse_fwd(ctypes.byref(cm.struct), ctypes.byref(io), z.data_ptr(), base, 1024, 0) == -1
