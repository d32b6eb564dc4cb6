"""Seeded synthetic checkpoints with the reference's key names (workload definition; no model arithmetic).

No pretrained weights ship with the reference, so parity runs on
random-but-seeded tensors.  Key names and shapes follow ``SynthesizerInfer.state_dict()``
(903 tensors at base.yaml; layout documented in SURVEY.md appendix A.2) and the Whisper
checkpoint format ``{"dims", "model_state_dict"}`` (whisper/inference.py:12-20).

Parameters the reference initialises to zero/identity (ResidualCouplingLayer.post,
vits/modules.py:283-284; SpeakerAdapter, vits_decoder/generator.py:30-34; SnakeBeta alpha/beta,
vits_decoder/alias/act.py:69-71) are drawn non-trivially here so every branch carries signal.
Distributions are ours (fan-in scaled so activations stay O(1)); they do not mimic the
reference's initialisers -- parity only needs both sides to see the same tensors.
"""
import math

import torch

from workload import config as C


def kaiser_sinc_filter(cutoff=0.25, half_width=0.3, taps=12):
    """The one 12-tap low-pass shared by every up/down sampler (vits_decoder/alias/filter.py:28-57,
    called with cutoff=0.5/ratio, half_width=0.6/ratio, ratio=2 -- resample.py:18-20,41-44)."""
    half = taps // 2
    delta_f = 4 * half_width
    att = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if att > 50.0:
        beta = 0.1102 * (att - 8.7)
    elif att >= 21.0:
        beta = 0.5842 * (att - 21.0) ** 0.4 + 0.07886 * (att - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(taps, beta=beta, periodic=False)
    time = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    f = f / f.sum()
    return f.view(1, 1, taps)


class _Rng:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)

    def normal(self, *shape, std=1.0, mean=0.0):
        return torch.randn(*shape, generator=self.g) * std + mean

    def uniform(self, *shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, generator=self.g) * (hi - lo) + lo


def _conv(sd, r, name, cout, cin, k, bias=True, gain=1.0):
    sd[name + ".weight"] = r.normal(cout, cin, k, std=gain / math.sqrt(cin * k))
    if bias:
        sd[name + ".bias"] = r.normal(cout, std=0.05)


def _wn_conv(sd, r, name, d0, d1, k, fan_in, nbias, gain=1.0):
    """weight-normed conv: tensors ``weight_v`` [d0,d1,k] and ``weight_g`` [d0,1,1] (norm over dims 1,2)."""
    v = r.normal(d0, d1, k, std=gain / math.sqrt(fan_in))
    sd[name + ".bias"] = r.normal(nbias, std=0.05)
    sd[name + ".weight_g"] = v.flatten(1).norm(dim=1).view(d0, 1, 1) * r.uniform(d0, 1, 1, lo=0.8, hi=1.2)
    sd[name + ".weight_v"] = v


def make_vits_state(hp=None, seed=1234):
    """state_dict of ``SynthesizerInfer`` (vits/models.py:211-239) filled with seeded tensors."""
    hp = hp or C.base_hp()
    r = _Rng(seed)
    sd = {}
    H, F_, I = hp.vits.hidden_channels, hp.vits.filter_channels, hp.vits.inter_channels
    dk = H // C.ENC_HEADS
    # enc_p  (vits/models.py:26-37)
    _conv(sd, r, "enc_p.pre", H, hp.vits.ppg_dim, 5)
    _conv(sd, r, "enc_p.hub", H, hp.vits.vec_dim, 5)
    sd["enc_p.pit.weight"] = r.normal(256, H, std=0.5)
    for i in range(C.ENC_LAYERS):
        a = f"enc_p.enc.attn_layers.{i}"
        sd[a + ".emb_rel_k"] = r.normal(1, 2 * C.ENC_WINDOW + 1, dk, std=dk ** -0.5)
        sd[a + ".emb_rel_v"] = r.normal(1, 2 * C.ENC_WINDOW + 1, dk, std=dk ** -0.5)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            _conv(sd, r, f"{a}.{n}", H, H, 1)
        sd[f"enc_p.enc.norm_layers_1.{i}.gamma"] = r.normal(H, std=0.1, mean=1.0)
        sd[f"enc_p.enc.norm_layers_1.{i}.beta"] = r.normal(H, std=0.1)
        _conv(sd, r, f"enc_p.enc.ffn_layers.{i}.conv_1", F_, H, C.ENC_FFN_KERNEL, gain=1.4)
        _conv(sd, r, f"enc_p.enc.ffn_layers.{i}.conv_2", H, F_, C.ENC_FFN_KERNEL)
        sd[f"enc_p.enc.norm_layers_2.{i}.gamma"] = r.normal(H, std=0.1, mean=1.0)
        sd[f"enc_p.enc.norm_layers_2.{i}.beta"] = r.normal(H, std=0.1)
    _conv(sd, r, "enc_p.proj", 2 * I, H, 1, gain=0.5)
    # flow (vits/models.py:66-78, vits/modules.py:250-286); Flip modules at odd indices hold no tensors
    half = I // 2
    for f in range(C.FLOW_N):
        p = f"flow.flows.{2 * f}"
        _conv(sd, r, p + ".pre", H, half, 1)
        for l in range(C.FLOW_WN_LAYERS):
            _wn_conv(sd, r, f"{p}.enc.in_layers.{l}", 2 * H, H, C.FLOW_KERNEL, H * C.FLOW_KERNEL, 2 * H)
            rs = 2 * H if l < C.FLOW_WN_LAYERS - 1 else H
            _wn_conv(sd, r, f"{p}.enc.res_skip_layers.{l}", rs, H, 1, H, rs, gain=0.7)
        _conv(sd, r, p + ".post", half, H, 1, gain=0.5)      # zero-init in the reference: de-zeroed
        sd[p + ".snac.weight"] = r.normal(2 * half, hp.vits.spk_dim, 1, std=0.3 / math.sqrt(hp.vits.spk_dim))
        sd[p + ".snac.bias"] = r.normal(2 * half, std=0.1)
    # dec (vits_decoder/generator.py:52-112)
    U = hp.gen.upsample_input
    C0 = hp.gen.upsample_initial_channel
    sd["dec.adapter.W_scale.weight"] = r.normal(U, hp.vits.spk_dim, std=0.2 / math.sqrt(hp.vits.spk_dim))
    sd["dec.adapter.W_scale.bias"] = r.normal(U, std=0.1, mean=1.0)
    sd["dec.adapter.W_bias.weight"] = r.normal(U, hp.vits.spk_dim, std=0.2 / math.sqrt(hp.vits.spk_dim))
    sd["dec.adapter.W_bias.bias"] = r.normal(U, std=0.1)
    _conv(sd, r, "dec.conv_pre", C0, U, 7)
    sd["dec.m_source.merge_w"] = torch.tensor([C.NSF_MERGE_W], dtype=torch.float32)
    sd["dec.m_source.merge_b"] = torch.tensor([C.NSF_MERGE_B], dtype=torch.float32)
    rates, ksz = list(hp.gen.upsample_rates), list(hp.gen.upsample_kernel_sizes)
    filt = kaiser_sinc_filter()
    n_up = len(rates)
    for i in range(n_up):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        if i + 1 < n_up:
            s = 1
            for u in rates[i + 1:]:
                s *= u
            _conv(sd, r, f"dec.noise_convs.{i}", cout, 1, 2 * s, gain=1.0)
        else:
            _conv(sd, r, f"dec.noise_convs.{i}", cout, 1, 1, gain=1.0)
        # ConvTranspose1d weight is [C_in, C_out, k]; weight-norm g is per INPUT channel (dim 0)
        _wn_conv(sd, r, f"dec.ups.{i}", cin, cout, ksz[i], cin * ksz[i] / rates[i], cout)
    rk = list(hp.gen.resblock_kernel_sizes)
    for i in range(n_up):
        ch = C0 // (2 ** (i + 1))
        for j, k in enumerate(rk):
            b = f"dec.resblocks.{i * len(rk) + j}"
            for q in range(3):
                _wn_conv(sd, r, f"{b}.convs1.{q}", ch, ch, k, ch * k, ch, gain=1.0)
                _wn_conv(sd, r, f"{b}.convs2.{q}", ch, ch, k, ch * k, ch, gain=0.4)
            for q in range(6):
                sd[f"{b}.activations.{q}.act.alpha"] = r.normal(ch, std=0.3)
                sd[f"{b}.activations.{q}.act.beta"] = r.normal(ch, std=0.3)
                sd[f"{b}.activations.{q}.upsample.filter"] = filt.clone()
                sd[f"{b}.activations.{q}.downsample.lowpass.filter"] = filt.clone()
    ch = C0 // (2 ** n_up)
    sd["dec.activation_post.act.alpha"] = r.normal(ch, std=0.3)
    sd["dec.activation_post.act.beta"] = r.normal(ch, std=0.3)
    sd["dec.activation_post.upsample.filter"] = filt.clone()
    sd["dec.activation_post.downsample.lowpass.filter"] = filt.clone()
    sd["dec.conv_post.weight"] = r.normal(1, ch, 7, std=0.15 / math.sqrt(ch * 7))
    return {k: v.float().contiguous() for k, v in sd.items()}


def sinusoids(length, channels, max_timescale=10000.0):
    """Positional table of the audio encoder (whisper/model.py:48-54)."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def make_whisper_state(dims=None, seed=4321, n_layers_present=None):
    """A Whisper checkpoint dict ``{"dims", "model_state_dict"}`` holding the ENCODER keys only.

    whisper/inference.py:16-20 deletes the decoder and the last quarter of the encoder blocks and loads
    with strict=False, so decoder.* keys and blocks >= 3/4 n_audio_layer never matter; by default we
    materialise just the blocks that survive (24 for large-v2) to keep the 1.9 GB build fast.
    """
    dims = dict(dims or C.WHISPER_LARGE_V2)
    S, L = dims["n_audio_state"], dims["n_audio_layer"]
    keep = L - L // 4 if n_layers_present is None else n_layers_present
    r = _Rng(seed)
    sd = {}
    sd["encoder.conv1.weight"] = r.normal(S, dims["n_mels"], 3, std=1.0 / math.sqrt(dims["n_mels"] * 3))
    sd["encoder.conv1.bias"] = r.normal(S, std=0.05)
    sd["encoder.conv2.weight"] = r.normal(S, S, 3, std=1.0 / math.sqrt(S * 3))
    sd["encoder.conv2.bias"] = r.normal(S, std=0.05)
    sd["encoder.positional_embedding"] = sinusoids(dims["n_audio_ctx"], S)
    for i in range(keep):
        b = f"encoder.blocks.{i}"
        for n in ("query", "key", "value", "out"):
            g = 0.5 if n == "out" else 1.0
            sd[f"{b}.attn.{n}.weight"] = r.normal(S, S, std=g / math.sqrt(S))
            if n != "key":                       # key has no bias (whisper/model.py:62)
                sd[f"{b}.attn.{n}.bias"] = r.normal(S, std=0.05)
        sd[f"{b}.attn_ln.weight"] = r.normal(S, std=0.1, mean=1.0)
        sd[f"{b}.attn_ln.bias"] = r.normal(S, std=0.1)
        sd[f"{b}.mlp.0.weight"] = r.normal(4 * S, S, std=1.0 / math.sqrt(S))
        sd[f"{b}.mlp.0.bias"] = r.normal(4 * S, std=0.05)
        sd[f"{b}.mlp.2.weight"] = r.normal(S, 4 * S, std=0.5 / math.sqrt(4 * S))
        sd[f"{b}.mlp.2.bias"] = r.normal(S, std=0.05)
        sd[f"{b}.mlp_ln.weight"] = r.normal(S, std=0.1, mean=1.0)
        sd[f"{b}.mlp_ln.bias"] = r.normal(S, std=0.1)
    sd["encoder.ln_post.weight"] = r.normal(S, std=0.1, mean=1.0)
    sd["encoder.ln_post.bias"] = r.normal(S, std=0.1)
    return {"dims": dims, "model_state_dict": {k: v.float().contiguous() for k, v in sd.items()}}


def stress_whisper_state(ck, seed=97):
    """Outlier-stress variant of a Whisper checkpoint (a modified copy): trained Whisper-large encoders carry a handful of
    residual-stream channels two orders of magnitude above the rest, LayerNorm gains far from 1 and saturated GELU inputs,
    which N(0, sigma) weights never produce.  Deterministic post-hoc edits, so the plain fixtures keep their RNG stream:
      * 4 residual channels driven to |x| ~ 50-100 from block 1 on (mlp.2 rows x50 and a +-20 bias),
      * 6 LayerNorm gains per block raised to 10..30 (attn_ln and mlp_ln),
      * 8 mlp.0 biases per block at +-15 (erf-GELU far in both tails),
      * one head's query/key rows x3 in every third block (peaked softmax)."""
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in ck["model_state_dict"].items()}
    S = ck["dims"]["n_audio_state"]
    dh = S // ck["dims"]["n_audio_head"]
    big = torch.randperm(S, generator=g)[:4]
    nb = len([k for k in sd if k.endswith(".mlp_ln.weight")])
    for i in range(nb):
        b = f"encoder.blocks.{i}"
        if i == 1:
            sd[f"{b}.mlp.2.weight"][big] *= 50.0
            sd[f"{b}.mlp.2.bias"][big] += torch.tensor([20.0, -20.0, 20.0, -20.0])
        for ln in ("attn_ln", "mlp_ln"):
            idx = torch.randperm(S, generator=g)[:6]
            sd[f"{b}.{ln}.weight"][idx] = 10.0 + 20.0 * torch.rand(6, generator=g)
        idx = torch.randperm(4 * S, generator=g)[:8]
        sd[f"{b}.mlp.0.bias"][idx] = 15.0 * torch.sign(torch.randn(8, generator=g))
        if i % 3 == 0:
            h = int(torch.randint(0, S // dh, (1,), generator=g))
            sd[f"{b}.attn.query.weight"][h * dh:(h + 1) * dh] *= 3.0
            sd[f"{b}.attn.key.weight"][h * dh:(h + 1) * dh] *= 3.0
    idx = torch.randperm(S, generator=g)[:6]
    sd["encoder.ln_post.weight"][idx] = 10.0 + 20.0 * torch.rand(6, generator=g)
    return {"dims": dict(ck["dims"]), "model_state_dict": sd}


def stress_vits_state(sd, hp=None, seed=98):
    """Outlier-stress variant of a SynthesizerInfer state dict (a modified copy): a few generator channels x50 at conv_pre
    (|x| >> 1 into the first SnakeBeta stages), SnakeBeta frequencies e^alpha up to e and amplitudes 1/e^beta up to 1.6 on a
    fifth of the channels, and prior-encoder LayerNorm gains up to 30 (|z_p| reaches hundreds).  The set is tuned to stay
    WELL-CONDITIONED: x50 channels combined with frequencies e^2.5 and amplitudes e^2 make the fp32 and fp64 oracles themselves
    disagree by 5e-2 on the waveform (each SnakeBeta then has slope ~100 and 90 of them are chained), which would test chaos,
    not kernels; sin^2 at arguments of thousands of radians is covered at kernel level instead (kernel_cases.check_snake)."""
    hp = hp or C.base_hp()
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    C0 = hp.gen.upsample_initial_channel
    idx = torch.randperm(C0, generator=g)[:3]
    sd["dec.conv_pre.weight"][idx] *= 50.0
    for k in sd:
        if k.endswith(".act.alpha") or k.endswith(".act.beta"):
            n = sd[k].shape[0]
            m = max(1, n // 5)
            sel = torch.randperm(n, generator=g)[:m]
            sd[k][sel] = (1.0 if k.endswith("alpha") else -0.5) * torch.rand(m, generator=g)
        # (not the last norm_layers_2: it feeds `proj`, whose log-sigma half is exponentiated -- gains of 30 there make
        # |z_p| ~ 1e7 and the comparison a test of exp() conditioning)
        if ".norm_layers_" in k and k.endswith(".gamma") and k != f"enc_p.enc.norm_layers_2.{C.ENC_LAYERS - 1}.gamma":
            sel = torch.randperm(sd[k].shape[0], generator=g)[:4]
            sd[k][sel] = 10.0 + 20.0 * torch.rand(4, generator=g)
    return sd


def make_hubert_state(dims=None, seed=2468):
    """``HubertSoft().state_dict()`` key layout (hubert/hubert_model.py; 166 tensors at the reference dims), seeded.
    ``masked_spec_embed`` / ``label_embedding`` are training-only and included so a strict load succeeds."""
    d = dict(C.HUBERT_SOFT if dims is None else dims)
    r = _Rng(seed)
    sd = {}
    Cc, E, Fd = d["conv_dim"], d["embed"], d["ffn"]
    sd["masked_spec_embed"] = r.uniform(E)
    sd["feature_extractor.conv0.weight"] = r.normal(Cc, 1, 10, std=1.0 / math.sqrt(10))
    sd["feature_extractor.norm0.weight"] = r.normal(Cc, std=0.2, mean=1.0)
    sd["feature_extractor.norm0.bias"] = r.normal(Cc, std=0.1)
    for i, k in zip(range(1, 7), (3, 3, 3, 3, 2, 2)):
        sd[f"feature_extractor.conv{i}.weight"] = r.normal(Cc, Cc, k, std=1.4 / math.sqrt(Cc * k))
    sd["feature_projection.norm.weight"] = r.normal(Cc, std=0.2, mean=1.0)
    sd["feature_projection.norm.bias"] = r.normal(Cc, std=0.1)
    sd["feature_projection.projection.weight"] = r.normal(E, Cc, std=1.0 / math.sqrt(Cc))
    sd["feature_projection.projection.bias"] = r.normal(E, std=0.05)
    G, Kp = d["pos_groups"], d["pos_kernel"]
    sd["positional_embedding.conv.bias"] = r.normal(E, std=0.05)
    sd["positional_embedding.conv.weight_g"] = r.uniform(1, 1, Kp, lo=0.5, hi=1.5)
    sd["positional_embedding.conv.weight_v"] = r.normal(E, E // G, Kp, std=1.0)
    sd["norm.weight"] = r.normal(E, std=0.2, mean=1.0)
    sd["norm.bias"] = r.normal(E, std=0.1)
    for i in range(d["layers"]):
        p = f"encoder.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = r.normal(3 * E, E, std=1.0 / math.sqrt(E))
        sd[p + "self_attn.in_proj_bias"] = r.normal(3 * E, std=0.05)
        sd[p + "self_attn.out_proj.weight"] = r.normal(E, E, std=1.0 / math.sqrt(E))
        sd[p + "self_attn.out_proj.bias"] = r.normal(E, std=0.05)
        sd[p + "linear1.weight"] = r.normal(Fd, E, std=1.0 / math.sqrt(E))
        sd[p + "linear1.bias"] = r.normal(Fd, std=0.05)
        sd[p + "linear2.weight"] = r.normal(E, Fd, std=1.0 / math.sqrt(Fd))
        sd[p + "linear2.bias"] = r.normal(E, std=0.05)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = r.normal(E, std=0.2, mean=1.0)
            sd[p + n + ".bias"] = r.normal(E, std=0.1)
    sd["proj.weight"] = r.normal(d["proj"], E, std=1.0 / math.sqrt(E))
    sd["proj.bias"] = r.normal(d["proj"], std=0.05)
    sd["label_embedding.weight"] = r.normal(100, d["proj"])
    return sd


CREPE_CAPACITY = {"full": ([1, 1024, 128, 128, 128, 256], [1024, 128, 128, 128, 256, 512], 2048),
                  "tiny": ([1, 128, 16, 16, 16, 32], [128, 16, 16, 16, 32, 64], 256)}


def make_crepe_state(capacity="full", seed=1357):
    """``crepe.Crepe(capacity).state_dict()`` key layout (crepe/model.py:14-101), seeded; BatchNorm running statistics are
    drawn non-trivially (eval-mode BatchNorm is the affine map they define)."""
    cin, cout, feat = CREPE_CAPACITY[capacity]
    r = _Rng(seed)
    sd = {}
    for i, (a, b) in enumerate(zip(cin, cout), start=1):
        k = 512 if i == 1 else 64
        sd[f"conv{i}.weight"] = r.normal(b, a, k, 1, std=1.4 / math.sqrt(a * k))
        sd[f"conv{i}.bias"] = r.normal(b, std=0.05)
        sd[f"conv{i}_BN.weight"] = r.uniform(b, lo=0.6, hi=1.4)
        sd[f"conv{i}_BN.bias"] = r.normal(b, std=0.1)
        sd[f"conv{i}_BN.running_mean"] = r.normal(b, std=0.2, mean=0.3)
        sd[f"conv{i}_BN.running_var"] = r.uniform(b, lo=0.5, hi=1.5)
        sd[f"conv{i}_BN.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    sd["classifier.weight"] = r.normal(360, feat, std=2.0 / math.sqrt(feat))
    sd["classifier.bias"] = r.normal(360, std=0.1)
    return sd
