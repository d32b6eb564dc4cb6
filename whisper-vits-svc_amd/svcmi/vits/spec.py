"""Parameter inventory of ``SynthesizerInfer`` -- names and shapes of the reference's
``state_dict()`` (903 tensors at configs/base.yaml), derived from the
hyper-parameters only, so checkpoints made by svc_export.py:40-57 load by key.
"""
import math
from collections import OrderedDict

import torch

from . import consts as K

# vits_decoder/nsf.py:378-381: the fixed (buffer) merge of the 11 harmonics
NSF_MERGE_W = [0.2942, -0.2243, 0.0033, -0.0056, -0.0020, -0.0046, 0.0221, -0.0083, -0.0241, -0.0036, -0.0581]
NSF_MERGE_B = 0.0008


def kaiser_sinc_taps(cutoff=0.25, half_width=0.3, taps=12):
    """12-tap Kaiser-windowed sinc low-pass registered as a buffer by every UpSample1d/LowPassFilter1d
    (vits_decoder/alias/filter.py:28-57 with cutoff 0.5/2, half_width 0.6/2)."""
    half = taps // 2
    att = 2.285 * (half - 1) * math.pi * 4 * half_width + 7.95
    beta = 0.1102 * (att - 8.7) if att > 50 else (0.5842 * (att - 21) ** 0.4 + 0.07886 * (att - 21) if att >= 21 else 0.0)
    win = torch.kaiser_window(taps, beta=beta, periodic=False)
    t = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * win * torch.sinc(2 * cutoff * t)
    return (f / f.sum()).view(1, 1, taps)


def param_shapes(hp):
    s = OrderedDict()
    H, F_, I = hp.vits.hidden_channels, hp.vits.filter_channels, hp.vits.inter_channels
    dk = H // K.ENC_HEADS

    def conv(name, co, ci, k, bias=True):
        s[name + ".weight"] = (co, ci, k)
        if bias:
            s[name + ".bias"] = (co,)

    def wn(name, d0, d1, k, nb):
        s[name + ".bias"] = (nb,)
        s[name + ".weight_g"] = (d0, 1, 1)
        s[name + ".weight_v"] = (d0, d1, k)

    conv("enc_p.pre", H, hp.vits.ppg_dim, 5)
    conv("enc_p.hub", H, hp.vits.vec_dim, 5)
    s["enc_p.pit.weight"] = (256, H)
    for i in range(K.ENC_LAYERS):
        a = f"enc_p.enc.attn_layers.{i}"
        s[a + ".emb_rel_k"] = (1, 2 * K.ENC_WINDOW + 1, dk)
        s[a + ".emb_rel_v"] = (1, 2 * K.ENC_WINDOW + 1, dk)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            conv(f"{a}.{n}", H, H, 1)
    for i in range(K.ENC_LAYERS):
        s[f"enc_p.enc.norm_layers_1.{i}.gamma"] = (H,)
        s[f"enc_p.enc.norm_layers_1.{i}.beta"] = (H,)
    for i in range(K.ENC_LAYERS):
        conv(f"enc_p.enc.ffn_layers.{i}.conv_1", F_, H, K.ENC_FFN_KERNEL)
        conv(f"enc_p.enc.ffn_layers.{i}.conv_2", H, F_, K.ENC_FFN_KERNEL)
    for i in range(K.ENC_LAYERS):
        s[f"enc_p.enc.norm_layers_2.{i}.gamma"] = (H,)
        s[f"enc_p.enc.norm_layers_2.{i}.beta"] = (H,)
    conv("enc_p.proj", 2 * I, H, 1)
    half = I // 2
    for f in range(K.FLOW_N):
        p = f"flow.flows.{2 * f}"
        conv(p + ".pre", H, half, 1)
        for l in range(K.FLOW_WN_LAYERS):
            wn(f"{p}.enc.in_layers.{l}", 2 * H, H, K.FLOW_KERNEL, 2 * H)
        for l in range(K.FLOW_WN_LAYERS):
            rs = 2 * H if l < K.FLOW_WN_LAYERS - 1 else H
            wn(f"{p}.enc.res_skip_layers.{l}", rs, H, 1, rs)
        conv(p + ".post", half, H, 1)
        conv(p + ".snac", 2 * half, hp.vits.spk_dim, 1)
    U, C0 = hp.gen.upsample_input, hp.gen.upsample_initial_channel
    s["dec.adapter.W_scale.weight"] = (U, hp.vits.spk_dim)
    s["dec.adapter.W_scale.bias"] = (U,)
    s["dec.adapter.W_bias.weight"] = (U, hp.vits.spk_dim)
    s["dec.adapter.W_bias.bias"] = (U,)
    conv("dec.conv_pre", C0, U, 7)
    s["dec.m_source.merge_w"] = (1, 11)
    s["dec.m_source.merge_b"] = (1,)
    rates, ksz = list(hp.gen.upsample_rates), list(hp.gen.upsample_kernel_sizes)
    n_up = len(rates)
    for i in range(n_up):
        cout = C0 // (2 ** (i + 1))
        st = int(math.prod(rates[i + 1:])) if i + 1 < n_up else 1
        conv(f"dec.noise_convs.{i}", cout, 1, 2 * st if i + 1 < n_up else 1)
    for i in range(n_up):
        wn(f"dec.ups.{i}", C0 // (2 ** i), C0 // (2 ** (i + 1)), ksz[i], C0 // (2 ** (i + 1)))
    rk = list(hp.gen.resblock_kernel_sizes)
    for i in range(n_up):
        ch = C0 // (2 ** (i + 1))
        for j, k in enumerate(rk):
            b = f"dec.resblocks.{i * len(rk) + j}"
            for q in range(3):
                wn(f"{b}.convs1.{q}", ch, ch, k, ch)
            for q in range(3):
                wn(f"{b}.convs2.{q}", ch, ch, k, ch)
            for q in range(6):
                s[f"{b}.activations.{q}.act.alpha"] = (ch,)
                s[f"{b}.activations.{q}.act.beta"] = (ch,)
                s[f"{b}.activations.{q}.upsample.filter"] = (1, 1, 12)
                s[f"{b}.activations.{q}.downsample.lowpass.filter"] = (1, 1, 12)
    ch = C0 // (2 ** n_up)
    s["dec.activation_post.act.alpha"] = (ch,)
    s["dec.activation_post.act.beta"] = (ch,)
    s["dec.activation_post.upsample.filter"] = (1, 1, 12)
    s["dec.activation_post.downsample.lowpass.filter"] = (1, 1, 12)
    s["dec.conv_post.weight"] = (1, ch, 7)
    return s


def default_state_dict(hp, seed=1234):
    """Deterministic placeholder parameters (what a freshly constructed module would hold before
    ``load_state_dict``): small seeded normals, unit gains, exact buffer constants."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    filt = kaiser_sinc_taps()
    for name, shape in param_shapes(hp).items():
        if name.endswith(".filter"):
            sd[name] = filt.clone()
        elif name == "dec.m_source.merge_w":
            sd[name] = torch.tensor([NSF_MERGE_W], dtype=torch.float32)
        elif name == "dec.m_source.merge_b":
            sd[name] = torch.tensor([NSF_MERGE_B], dtype=torch.float32)
        elif name.endswith(".gamma") or name == "dec.adapter.W_scale.bias":
            sd[name] = torch.ones(shape)
        elif name.endswith((".bias", ".beta", ".alpha")):
            sd[name] = torch.zeros(shape)
        elif name.endswith(".weight_g"):
            sd[name] = None      # filled from weight_v below
        else:
            sd[name] = torch.randn(shape, generator=g) * 0.02
    for name in list(sd):
        if name.endswith(".weight_g"):
            v = sd[name[:-1] + "v"]
            sd[name] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
    return sd
