"""Checkpoint and speaker tooling around the hot path (row N4): the formats the engine loads.

    export_checkpoint  svc_export.py:10-57   training ckpt -> {'model_g': SynthesizerInfer.state_dict()}
    save_pretrain      svc_export.py:31-37   keep only model_g / model_d
    merge_model        svc_merge.py:33-39    rate * m1 + (1 - rate) * m2 per key
    average_model      svc_merge.py:17-25    mean of several state dicts
    mix_speakers       svc_eva.py:6-20       weighted sum of speaker embeddings

Pure state-dict arithmetic on the host (torch CPU tensors / numpy), no kernels involved.
"""
import collections
import functools
import operator
import os

import numpy as np
import torch


def load_model_g(checkpoint_path):
    """svc_merge.py:7-11."""
    assert os.path.isfile(checkpoint_path)
    return torch.load(checkpoint_path, map_location="cpu")["model_g"]


def save_model_g(state_dict, checkpoint_path):
    """svc_merge.py:14-15 / svc_export.py:40-45."""
    torch.save({"model_g": state_dict}, checkpoint_path)


def export_checkpoint(hp, checkpoint_path, save_path="sovits5.0.pth"):
    """svc_export.py:47-57: build the inference model, take every key it has from the training checkpoint's ``model_g``
    (keys the checkpoint lacks keep the model's initial value, silently, :18-23), drop everything else (posterior encoder,
    discriminator, optimiser state) and save ``{'model_g': ...}``.  Returns the saved state dict."""
    from .vits.models import SynthesizerInfer
    assert os.path.isfile(checkpoint_path)
    saved = torch.load(checkpoint_path, map_location="cpu")["model_g"]
    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp)
    new_state = collections.OrderedDict((k, saved[k] if k in saved else v) for k, v in model.state_dict().items())
    save_model_g(new_state, save_path)
    return new_state


def save_pretrain(checkpoint_path, save_path):
    """svc_export.py:31-37."""
    assert os.path.isfile(checkpoint_path)
    ckpt = torch.load(checkpoint_path, map_location="cpu")
    torch.save({"model_g": ckpt["model_g"], "model_d": ckpt["model_d"]}, save_path)


def average_model(model_list):
    """svc_merge.py:17-25: per key, ``(0 + m_0 + m_1 + ...) / n`` in list order."""
    n = float(len(model_list))
    return collections.OrderedDict(
        (key, torch.div(functools.reduce(operator.add, (m[key] for m in model_list), 0), n)) for key in model_list[0].keys())


def merge_model(model1, model2, rate):
    """svc_merge.py:33-39."""
    assert 0 < rate < 1, f"{rate} should be in range (0, 1)"
    return collections.OrderedDict((k, rate * model1[k] + (1 - rate) * model2[k]) for k in model1.keys())


def mix_speakers(eva_conf, save_path=None, dim=256):
    """svc_eva.py:6-20: ``eva_conf`` maps speaker .npy paths to weights; float64 accumulation like the reference."""
    eva = np.zeros(dim)
    for path, weight in eva_conf.items():
        assert os.path.isfile(path), path
        eva = eva + np.load(path) * weight
    if save_path is not None:
        np.save(save_path, eva, allow_pickle=False)
    return eva


def _main_export(argv=None):
    import argparse
    from .svc_inference import load_config
    p = argparse.ArgumentParser(description="svc_export.py")
    p.add_argument("-c", "--config", type=str, required=True)
    p.add_argument("-p", "--checkpoint_path", type=str, required=True)
    a = p.parse_args(argv)
    export_checkpoint(load_config(a.config), a.checkpoint_path, "sovits5.0.pth")


def _main_merge(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="svc_merge.py")
    p.add_argument("-m1", "--model1", type=str, required=True)
    p.add_argument("-m2", "--model2", type=str, required=True)
    p.add_argument("-r1", "--rate", type=float, required=True)
    a = p.parse_args(argv)
    save_model_g(merge_model(load_model_g(a.model1), load_model_g(a.model2), a.rate), "sovits5.0_merge.pth")


def _main_pack(argv=None):
    """Kernel-ready weights of one model as a packed-model file for hosts without Python (svcmi/packed.py, include/svcmi.h)."""
    import argparse
    from . import packed, weights as PW
    from .svc_inference import load_config
    p = argparse.ArgumentParser(description="pack a checkpoint into a .svcmi file (svcmi_packed_model_bind)")
    p.add_argument("--config", type=str, help="yaml config of the synthesizer (with --model)")
    p.add_argument("--model", type=str, help="sovits5.0.pth: {'model_g': SynthesizerInfer.state_dict()}")
    p.add_argument("--whisper", type=str, help="OpenAI Whisper checkpoint {'dims', 'model_state_dict'}")
    p.add_argument("--out", type=str, required=True)
    a = p.parse_args(argv)
    if bool(a.model) == bool(a.whisper) or (a.model and not a.config):
        p.error("give either --config + --model or --whisper")
    if a.model:
        w = PW.VitsWeights(load_model_g(a.model), load_config(a.config), "cpu")
    else:
        w = PW.WhisperWeights(torch.load(a.whisper, map_location="cpu"), "cpu")
    with open(a.out, "wb") as f:
        f.write(packed.pack_model(w))


if __name__ == "__main__":
    import sys
    cmd, rest = (sys.argv[1], sys.argv[2:]) if len(sys.argv) > 1 else ("", [])
    {"export": _main_export, "merge": _main_merge, "pack": _main_pack}.get(cmd, lambda _: sys.exit("usage: python -m svcmi.tools export|merge|pack ..."))(rest)
