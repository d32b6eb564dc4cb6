"""Import the real reference modules from /root/reference (build container only).

Used by ``oracle/make_golden.py`` and the ``needs_reference`` tests to pin the oracle against the
reference itself.  /root/reference does not exist on the GPU box: nothing on the gpu-marked path
imports this file.  Recipe from SURVEY.md section 8c: OmegaConf shim (attr-dict over yaml) and a
stub ``librosa`` so that ``whisper.model`` imports without the (absent) audio stack.
"""
import contextlib
import os
import sys
import types
import warnings

import torch

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "vits"))


def _prepare():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "librosa" not in sys.modules:
        import importlib.machinery
        lib = types.ModuleType("librosa")
        fil = types.ModuleType("librosa.filters")
        lib.__spec__ = importlib.machinery.ModuleSpec("librosa", None)     # transformers probes find_spec
        fil.__spec__ = importlib.machinery.ModuleSpec("librosa.filters", None)
        fil.mel = lambda *a, **k: None
        lib.filters = fil
        lib.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("librosa stub"))
        sys.modules["librosa"] = lib
        sys.modules["librosa.filters"] = fil
    warnings.filterwarnings("ignore", category=FutureWarning)


def ref_synthesizer(hp, state_dict):
    """The reference ``SynthesizerInfer`` (vits/models.py:211) loaded strictly with ``state_dict``."""
    _prepare()
    from vits.models import SynthesizerInfer
    m = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp)
    m.load_state_dict(state_dict, strict=True)
    m.train(False)          # Generator.eval() returns None (generator.py:154-158)
    return m


def ref_whisper_encoder(ckpt):
    """The truncated reference encoder exactly as whisper/inference.py:11-29 builds it (cpu, fp32)."""
    _prepare()
    from whisper.model import Whisper, ModelDimensions
    dims = ModelDimensions(**ckpt["dims"])
    model = Whisper(dims)
    del model.decoder
    cut = len(model.encoder.blocks) // 4
    del model.encoder.blocks[-cut:]
    model.load_state_dict(ckpt["model_state_dict"], strict=False)
    model.eval()
    return model


@contextlib.contextmanager
def injected_noise(randn_like_queue=(), rand_queue=()):
    """Feed host-drawn noise to the reference's three stochastic call sites (vits/models.py:51,
    vits_decoder/nsf.py:232-235,311, whisper/inference.py:46,58) by shadowing torch.randn_like /
    torch.rand for the duration of the call; shapes are checked so a mis-ordered queue fails."""
    rl, rq = list(randn_like_queue), list(rand_queue)
    orig_rl, orig_r = torch.randn_like, torch.rand

    def fake_randn_like(t, *a, **k):
        n = rl.pop(0)
        assert tuple(n.shape) == tuple(t.shape), (n.shape, t.shape)
        return n.to(t.dtype)

    def fake_rand(*shape, **k):
        n = rq.pop(0)
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(n.shape) == shp, (n.shape, shp)
        return n.clone()

    torch.randn_like, torch.rand = fake_randn_like, fake_rand
    try:
        yield
    finally:
        torch.randn_like, torch.rand = orig_rl, orig_r
    assert not rl and not rq, "unused injected noise"
