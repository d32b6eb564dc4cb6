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
