"""CPU oracle of the HuBERT-Soft content-unit extractor (TEST INFRASTRUCTURE; row N3 of SURVEY.md 8f).

Functional fp32 restatement of ``HubertSoft.units`` (hubert/hubert_model.py:64-72 -> encode :41-50 ->
FeatureExtractor :75-96, FeatureProjection :99-110, PositionalConvEmbedding :113-131, the 12 post-LN
``nn.TransformerEncoderLayer(768, 12, 3072, activation="gelu", batch_first=True)`` of :20-25,134-158, ``proj`` :26) on
a plain state dict; dimensions are read from the tensor shapes so the small test configuration runs through the same
code.  Pinned against the reference module itself by oracle/make_golden.py (tests/golden/hubert_soft_*.npz).
"""
import math

import torch
import torch.nn.functional as F


def units(sd, wav, heads):
    """wav [B, 1, n] at 16 kHz -> soft units [B, T, proj], T = (n + 80 - 400) // 320 + 1."""
    x = F.pad(wav, ((400 - 320) // 2, (400 - 320) // 2))                              # :70
    x = F.conv1d(x, sd["feature_extractor.conv0.weight"], None, stride=5)             # :88
    Cc = x.shape[1]
    x = F.gelu(F.group_norm(x, Cc, sd["feature_extractor.norm0.weight"], sd["feature_extractor.norm0.bias"], 1e-5))
    for i, s in zip(range(1, 7), (2, 2, 2, 2, 2, 2)):                                  # :89-94
        x = F.gelu(F.conv1d(x, sd[f"feature_extractor.conv{i}.weight"], None, stride=s))
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (Cc,), sd["feature_projection.norm.weight"], sd["feature_projection.norm.bias"], 1e-5)
    x = F.linear(x, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    # positional conv: weight_norm over dim=2 (one norm per kernel tap), grouped, last frame dropped (:123-131)
    v, g = sd["positional_embedding.conv.weight_v"], sd["positional_embedding.conv.weight_g"]
    w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    E, Kp = v.shape[0], v.shape[2]
    G = E // v.shape[1]
    pos = F.conv1d(x.transpose(1, 2), w, sd["positional_embedding.conv.bias"], padding=Kp // 2, groups=G)[:, :, :-1]
    x = x + F.gelu(pos).transpose(1, 2)                                                # :46
    x = F.layer_norm(x, (E,), sd["norm.weight"], sd["norm.bias"], 1e-5)                # :47
    B, T, _ = x.shape
    dh = E // heads
    i = 0
    while f"encoder.layers.{i}.norm1.weight" in sd:                                    # post-LN encoder layers
        p = f"encoder.layers.{i}."
        qkv = F.linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, vv = [t.view(B, T, heads, dh).transpose(1, 2) for t in qkv.split(E, dim=-1)]
        a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1) @ vv
        a = a.transpose(1, 2).reshape(B, T, E)
        x = F.layer_norm(x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"]), (E,),
                         sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        h = F.linear(F.gelu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = F.layer_norm(x + h, (E,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        i += 1
    return F.linear(x, sd["proj.weight"], sd["proj.bias"])                             # :72


def window_plan(n_samples, sr=16000, window_s=20):
    """(start, stop) per window, hubert/inference.py:29-49: full 20 s windows while idx + 20 s < len, then the rest."""
    plan, idx, step = [], 0, window_s * sr
    while idx + step < n_samples:
        plan.append((idx, idx + step))
        idx += step
    if idx < n_samples:
        plan.append((idx, n_samples))
    return plan
