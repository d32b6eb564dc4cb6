"""CPU oracle: functional fp32 restatement of the SVC inference hot path (TEST INFRASTRUCTURE).

Every function works on a plain ``state_dict`` (reference key names) and takes the stochastic
draws of the reference as EXPLICIT arguments, so that oracle, reference and HIP engine can be
fed identical noise.  Each function cites the reference lines it follows.
The restatement is checked against the imported reference by ``oracle/make_golden.py`` and
``tests/test_oracle_golden.py``.  Layout convention is the reference's: NCL ``[B, C, T]``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from workload import config as C


# ----------------------------------------------------------------------------- helpers
def fold_weight_norm(sd, name):
    """w = g * v / ||v||, norm over every dim but 0 (torch weight_norm default dim=0).
    The reference never removes weight-norm at inference (generator.py:154-158), so the fold is
    what its forward computes on every call."""
    v, g = sd[name + ".weight_v"], sd[name + ".weight_g"]
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / n


def sequence_mask(lengths, T):
    """vits/commons.py:147-151 -> float mask [B,1,T]."""
    return (torch.arange(T)[None, :] < lengths[:, None]).float().unsqueeze(1)


def f0_to_coarse(f0):
    """vits/utils.py:20-33: Hz -> mel-scale bin in 1..255."""
    f0_mel_min = 1127 * np.log(1 + 50.0 / 700)
    f0_mel_max = 1127 * np.log(1 + 1100.0 / 700)
    mel = 1127 * (1 + f0 / 700).log()
    voiced = mel > 0
    mel = torch.where(voiced, (mel - f0_mel_min) * (256 - 2) / (f0_mel_max - f0_mel_min) + 1, mel)
    mel = torch.where(mel <= 1, torch.ones_like(mel), mel)
    mel = torch.where(mel > 255, torch.full_like(mel, 255.0), mel)
    return (mel + 0.5).long()


def channel_layer_norm(x, gamma, beta, eps=1e-5):
    """vits/modules.py:19-22 -- LN over the channel dim of [B,C,T]."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), gamma, beta, eps).transpose(1, 2)


# ----------------------------------------------------------------------------- prior encoder
def relpos_attention(sd, p, x, attn_mask, n_heads, window):
    """vits/attentions.py:215-274 with the relative-position terms written as a band
    (|j-i| <= window) instead of the reference's pad/reshape skew (attentions.py:312-347);
    SURVEY.md A.3.  x: [B,C,T]; attn_mask: [B,1,T,T]."""
    B, Cc, T = x.shape
    dk = Cc // n_heads
    q = F.conv1d(x, sd[p + ".conv_q.weight"], sd[p + ".conv_q.bias"])
    k = F.conv1d(x, sd[p + ".conv_k.weight"], sd[p + ".conv_k.bias"])
    v = F.conv1d(x, sd[p + ".conv_v.weight"], sd[p + ".conv_v.bias"])
    q = q.view(B, n_heads, dk, T).transpose(2, 3) / math.sqrt(dk)
    k = k.view(B, n_heads, dk, T).transpose(2, 3)
    v = v.view(B, n_heads, dk, T).transpose(2, 3)
    scores = q @ k.transpose(-2, -1)                                  # [B,h,T,T]
    Ek, Ev = sd[p + ".emb_rel_k"][0], sd[p + ".emb_rel_v"][0]         # [2w+1, dk]
    idx = torch.arange(T)
    rel = idx[None, :] - idx[:, None]                                 # j - i
    band = rel.abs() <= window
    rel_c = (rel + window).clamp(0, 2 * window)
    qe = q @ Ek.t()                                                   # [B,h,T,2w+1]
    scores = scores + torch.where(band, qe.gather(-1, rel_c.expand(B, n_heads, T, T)), scores.new_zeros(()))
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    pa = F.softmax(scores, dim=-1)
    out = pa @ v
    # relative values: sum_{|j-i|<=w} P[i,j] * Ev[j-i+w]
    pb = torch.where(band, pa, pa.new_zeros(()))
    rw = pa.new_zeros(B, n_heads, T, 2 * window + 1)
    rw.scatter_add_(-1, rel_c.expand(B, n_heads, T, T), pb)
    out = out + rw @ Ev
    out = out.transpose(2, 3).contiguous().view(B, Cc, T)
    return F.conv1d(out, sd[p + ".conv_o.weight"], sd[p + ".conv_o.bias"])


def ffn(sd, p, x, mask, ksz):
    """vits/attentions.py:390-416 (activation None -> relu, same padding)."""
    pl, pr = (ksz - 1) // 2, ksz // 2
    h = F.conv1d(F.pad(x * mask, (pl, pr)), sd[p + ".conv_1.weight"], sd[p + ".conv_1.bias"])
    h = torch.relu(h)
    h = F.conv1d(F.pad(h * mask, (pl, pr)), sd[p + ".conv_2.weight"], sd[p + ".conv_2.bias"])
    return h * mask


def text_encoder(sd, ppg, vec, f0c, lengths, noise):
    """vits/models.py:39-52.  ppg [B,T,ppg_dim], vec [B,T,vec_dim], f0c long [B,T], noise [B,I,T].
    Returns z, m, logs, mask."""
    T = ppg.shape[1]
    mask = sequence_mask(lengths, T)
    x = F.conv1d(ppg.transpose(1, 2), sd["enc_p.pre.weight"], sd["enc_p.pre.bias"], padding=2) * mask
    v = F.conv1d(vec.transpose(1, 2), sd["enc_p.hub.weight"], sd["enc_p.hub.bias"], padding=2) * mask
    x = x + v + sd["enc_p.pit.weight"][f0c].transpose(1, 2)
    # attentions.Encoder.forward, attentions.py:60-72 (post-LN)
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    x = x * mask
    for i in range(C.ENC_LAYERS):
        y = relpos_attention(sd, f"enc_p.enc.attn_layers.{i}", x, attn_mask, C.ENC_HEADS, C.ENC_WINDOW)
        x = channel_layer_norm(x + y, sd[f"enc_p.enc.norm_layers_1.{i}.gamma"], sd[f"enc_p.enc.norm_layers_1.{i}.beta"])
        y = ffn(sd, f"enc_p.enc.ffn_layers.{i}", x, mask, C.ENC_FFN_KERNEL)
        x = channel_layer_norm(x + y, sd[f"enc_p.enc.norm_layers_2.{i}.gamma"], sd[f"enc_p.enc.norm_layers_2.{i}.beta"])
    x = x * mask
    stats = F.conv1d(x, sd["enc_p.proj.weight"], sd["enc_p.proj.bias"]) * mask
    m, logs = stats.chunk(2, dim=1)
    z = (m + noise * torch.exp(logs)) * mask          # models.py:51 with the randn_like externalised
    return z, m, logs, mask


# ----------------------------------------------------------------------------- flow
def wn(sd, p, x, mask):
    """vits/modules.py:178-203 with g=None (no global conditioning inside the flow, modules.py:296)."""
    H = x.shape[1]
    out = torch.zeros_like(x)
    for l in range(C.FLOW_WN_LAYERS):
        w_in = fold_weight_norm(sd, f"{p}.in_layers.{l}")
        a = F.conv1d(x, w_in, sd[f"{p}.in_layers.{l}.bias"], padding=(C.FLOW_KERNEL - 1) // 2)
        acts = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])          # commons.py:126-133
        w_rs = fold_weight_norm(sd, f"{p}.res_skip_layers.{l}")
        rs = F.conv1d(acts, w_rs, sd[f"{p}.res_skip_layers.{l}.bias"])
        if l < C.FLOW_WN_LAYERS - 1:
            x = (x + rs[:, :H]) * mask
            out = out + rs[:, H:]
        else:
            out = out + rs
    return out * mask


def coupling_reverse(sd, p, x, mask, spk):
    """vits/modules.py:288-321, reverse branch, mean_only=True (logs == 0); logdet dropped."""
    half = x.shape[1] // 2
    s = F.conv1d(spk.unsqueeze(-1), sd[p + ".snac.weight"], sd[p + ".snac.bias"])
    m_s, v_s = s.chunk(2, dim=1)
    x0, x1 = x[:, :half], x[:, half:]
    x0n = (x0 - m_s) * torch.exp(-v_s) * mask
    h = F.conv1d(x0n, sd[p + ".pre.weight"], sd[p + ".pre.bias"]) * mask
    h = wn(sd, p + ".enc", h, mask)
    m = F.conv1d(h, sd[p + ".post.weight"], sd[p + ".post.bias"]) * mask
    x1 = (x1 - m) * mask
    x1 = (m_s + x1 * torch.exp(v_s)) * mask
    return torch.cat([x0, x1], dim=1)


def flow_reverse(sd, z, mask, spk):
    """vits/models.py:89-94: reversed(flows) = Flip, RCL3, Flip, RCL2, Flip, RCL1, Flip, RCL0."""
    x = z
    for f in reversed(range(C.FLOW_N)):
        x = torch.flip(x, [1])                                           # modules.py:225-229
        x = coupling_reverse(sd, f"flow.flows.{2 * f}", x, mask, spk)
    return x


# ----------------------------------------------------------------------------- generator
def snake_alias(x, alpha_log, beta_log, filt):
    """vits_decoder/alias/act.py:124-129 with the reference's own operator sequence, so that the CPU baseline costs what
    the reference costs: UpSample1d (resample.py:25-33: replicate-pad 5, depthwise conv_transpose1d stride 2, x2, crop
    15 / 15) -> SnakeBeta (act.py:79-92, log-scale alpha / beta) -> LowPassFilter1d with stride 2 (filter.py:86-95:
    replicate-pad 5 / 6, depthwise conv1d).  ``snake_alias_polyphase`` is the same function in the gather form of
    SURVEY.md A.5 (the form the kernels implement); tests/test_oracle_golden.py checks the two against each other."""
    Cc = x.shape[1]
    f = filt.view(1, 1, -1).expand(Cc, -1, -1)
    up = 2.0 * F.conv_transpose1d(F.pad(x, (5, 5), mode="replicate"), f, stride=2, groups=Cc)[..., 15:-15]
    a = torch.exp(alpha_log).view(1, -1, 1)
    b = torch.exp(beta_log).view(1, -1, 1)
    s = up + (1.0 / (b + 1e-9)) * torch.pow(torch.sin(up * a), 2)
    return F.conv1d(F.pad(s, (5, 6), mode="replicate"), f, stride=2, groups=Cc)


def snake_alias_polyphase(x, alpha_log, beta_log, filt):
    """vits_decoder/alias/act.py:124-129 in the polyphase form of SURVEY.md A.5:
    2x Kaiser-sinc upsample (resample.py:25-33) -> SnakeBeta (act.py:79-92) -> 2x low-pass
    decimation (filter.py:86-95), replicate padding at the SEQUENCE ends."""
    B, Cc, n = x.shape
    f = filt.view(-1)
    xp = F.pad(x, (5, 5), mode="replicate")            # xp[i] = x[clamp(i-5)]
    win = xp.unfold(-1, 6, 1)                          # win[..., q, j] = x[cl(q-5+j)], q = 0..n+4
    # even phase y[2t] = 2*sum_j f[2j+1]*x[cl(t+2-j)] ; odd phase y[2t+1] = 2*sum_j f[2j]*x[cl(t+3-j)]
    fo = torch.flip(f[1::2], [0])                      # index m=5-j over x[t-3+m]
    fe = torch.flip(f[0::2], [0])
    ye = 2 * (win[..., 2:2 + n, :] * fo).sum(-1)       # window starting at x[t-3]
    yo = 2 * (win[..., 3:3 + n, :] * fe).sum(-1)       # window starting at x[t-2]
    y = torch.stack([ye, yo], dim=-1).reshape(B, Cc, 2 * n)
    a = torch.exp(alpha_log).view(1, -1, 1)
    b = torch.exp(beta_log).view(1, -1, 1)
    s = y + (1.0 / (b + 1e-9)) * torch.sin(y * a) ** 2
    sp = F.pad(s, (5, 6), mode="replicate")
    z = (sp.unfold(-1, 12, 2) * f).sum(-1)             # z[t] = sum_k f[k] * s[cl(2t+k-5)]
    return z


def amp_block(sd, p, x, ksz, dilations=(1, 3, 5)):
    """vits_decoder/bigv.py:50-58."""
    for q, d in enumerate(dilations):
        filt = sd[f"{p}.activations.{2 * q}.upsample.filter"]
        xt = snake_alias(x, sd[f"{p}.activations.{2 * q}.act.alpha"], sd[f"{p}.activations.{2 * q}.act.beta"], filt)
        xt = F.conv1d(xt, fold_weight_norm(sd, f"{p}.convs1.{q}"), sd[f"{p}.convs1.{q}.bias"],
                      dilation=d, padding=(ksz * d - d) // 2)
        xt = snake_alias(xt, sd[f"{p}.activations.{2 * q + 1}.act.alpha"], sd[f"{p}.activations.{2 * q + 1}.act.beta"], filt)
        xt = F.conv1d(xt, fold_weight_norm(sd, f"{p}.convs2.{q}"), sd[f"{p}.convs2.{q}.bias"],
                      padding=(ksz - 1) // 2)
        x = xt + x
    return x


def speaker_adapter(sd, x, spk, eps=1e-5):
    """vits_decoder/generator.py:36-47: un-affine LN over channels, then speaker scale/bias."""
    xt = x.transpose(1, 2)
    mean = xt.mean(-1, keepdim=True)
    var = ((xt - mean) ** 2).mean(-1, keepdim=True)
    y = (xt - mean) / (var + eps).sqrt()
    scale = F.linear(spk, sd["dec.adapter.W_scale.weight"], sd["dec.adapter.W_scale.bias"])
    bias = F.linear(spk, sd["dec.adapter.W_bias.weight"], sd["dec.adapter.W_bias.bias"])
    return (y * scale.unsqueeze(1) + bias.unsqueeze(1)).transpose(1, 2)


def generator_inference(sd, hp, spk, x, source, return_stages=False):
    """vits_decoder/generator.py:175-200.  x [B,192,T], source [B,1,320T] -> [B,1,320T]."""
    rates, ksz = list(hp.gen.upsample_rates), list(hp.gen.upsample_kernel_sizes)
    rk = list(hp.gen.resblock_kernel_sizes)
    rd = [list(d) for d in hp.gen.resblock_dilation_sizes]
    stages = []
    x = speaker_adapter(sd, x, spk)
    x = F.conv1d(x, sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], padding=3)
    x = x * torch.tanh(F.softplus(x))
    stages.append(x)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.conv_transpose1d(x, fold_weight_norm(sd, f"dec.ups.{i}"), sd[f"dec.ups.{i}.bias"],
                               stride=u, padding=(k - u) // 2)
        if i + 1 < len(rates):
            s = int(np.prod(rates[i + 1:]))
            xs = F.conv1d(source, sd[f"dec.noise_convs.{i}.weight"], sd[f"dec.noise_convs.{i}.bias"],
                          stride=s, padding=s // 2)
        else:
            xs = F.conv1d(source, sd[f"dec.noise_convs.{i}.weight"], sd[f"dec.noise_convs.{i}.bias"])
        x = x + xs
        acc = None
        for j in range(len(rk)):
            y = amp_block(sd, f"dec.resblocks.{i * len(rk) + j}", x, rk[j], rd[j])
            acc = y if acc is None else acc + y
        x = acc / len(rk)
        stages.append(x)
    x = snake_alias(x, sd["dec.activation_post.act.alpha"], sd["dec.activation_post.act.beta"],
                    sd["dec.activation_post.upsample.filter"])
    x = F.conv1d(x, sd["dec.conv_post.weight"], None, padding=3)
    x = torch.tanh(x)
    return (x, stages) if return_stages else x


def pitch2source(sd, hp, f0, rand_ini, noise):
    """vits_decoder/generator.py:160-165 -> nsf.py:383-394 -> SineGen nsf.py:223-253,284-316, with
    ``torch.rand(B,11)`` (nsf.py:232-235) and ``randn_like(sine_waves)`` (nsf.py:311) passed in
    (SURVEY.md A.6).  f0 [B,T] Hz; rand_ini [B,11] (column 0 is forced to 0 here as in :235);
    noise [B,L,11].  Returns [B,1,L]."""
    hop = int(np.prod(list(hp.gen.upsample_rates)))
    sr = float(hp.data.sampling_rate)
    up = f0.repeat_interleave(hop, dim=1)                            # nearest upsample (generator.py:63-64)
    B, L = up.shape
    harm = torch.arange(1, C.NSF_HARMONICS + 1, dtype=torch.float32)
    f0_buf = up.unsqueeze(-1) * harm                                  # nsf.py:293-297
    rad = (f0_buf / sr) % 1
    ini = rand_ini.clone()
    ini[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ini
    over = torch.cumsum(rad, 1) % 1
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = ((over[:, 1:, :] - over[:, :-1, :]) < 0) * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * C.NSF_SINE_AMP
    uv = (up > 0).float().unsqueeze(-1)
    noise_amp = uv * C.NSF_NOISE_STD + (1 - uv) * C.NSF_SINE_AMP / 3
    sines = sines * uv + noise_amp * noise
    merged = torch.tanh(F.linear(sines, sd["dec.m_source.merge_w"]) + sd["dec.m_source.merge_b"])
    return merged.transpose(1, 2)


def source2wav(audio):
    """vits_decoder/generator.py:167-173 -> int16 numpy."""
    a = 32768.0 * audio.squeeze()
    return a.clamp(min=-32768.0, max=32767.0).short().cpu().numpy()


def synth_inference(sd, hp, ppg, vec, pit, spk, lengths, source, enc_noise, return_parts=False):
    """SynthesizerInfer.inference, vits/models.py:251-256."""
    z_p, m_p, logs_p, mask = text_encoder(sd, ppg, vec, f0_to_coarse(pit), lengths, enc_noise)
    z = flow_reverse(sd, z_p, mask, spk)
    o = generator_inference(sd, hp, spk, z * mask, source)
    if return_parts:
        return o, {"z_p": z_p, "m_p": m_p, "logs_p": logs_p, "z": z}
    return o


# ----------------------------------------------------------------------------- whisper encoder
def audio_encoder(wsd, mel, n_head, n_layer):
    """whisper/model.py:144-163 over the first ``n_layer`` blocks (whisper/inference.py:16-19 keeps
    24 of 32).  mel [B,80,n] -> [B, ceil(n/2), state]."""
    x = F.gelu(F.conv1d(mel, wsd["encoder.conv1.weight"], wsd["encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, wsd["encoder.conv2.weight"], wsd["encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    Tw, S = x.shape[1], x.shape[2]
    x = x + wsd["encoder.positional_embedding"][:Tw]
    dh = S // n_head
    scale = dh ** -0.25
    for i in range(n_layer):
        b = f"encoder.blocks.{i}"
        h = F.layer_norm(x, (S,), wsd[b + ".attn_ln.weight"], wsd[b + ".attn_ln.bias"])
        q = F.linear(h, wsd[b + ".attn.query.weight"], wsd[b + ".attn.query.bias"])
        k = F.linear(h, wsd[b + ".attn.key.weight"])
        v = F.linear(h, wsd[b + ".attn.value.weight"], wsd[b + ".attn.value.bias"])
        Bq = q.shape[0]
        q = q.view(Bq, Tw, n_head, dh).permute(0, 2, 1, 3) * scale       # model.py:88-101
        k = k.view(Bq, Tw, n_head, dh).permute(0, 2, 3, 1) * scale
        v = v.view(Bq, Tw, n_head, dh).permute(0, 2, 1, 3)
        w = F.softmax((q @ k).float(), dim=-1)
        a = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
        x = x + F.linear(a, wsd[b + ".attn.out.weight"], wsd[b + ".attn.out.bias"])
        h = F.layer_norm(x, (S,), wsd[b + ".mlp_ln.weight"], wsd[b + ".mlp_ln.bias"])
        h = F.gelu(F.linear(h, wsd[b + ".mlp.0.weight"], wsd[b + ".mlp.0.bias"]))
        x = x + F.linear(h, wsd[b + ".mlp.2.weight"], wsd[b + ".mlp.2.bias"])
    return F.layer_norm(x, (S,), wsd["encoder.ln_post.weight"], wsd["encoder.ln_post.bias"])


def whisper_kept_layers(dims):
    """whisper/inference.py:16-19: drop the last quarter of the encoder blocks."""
    L = dims["n_audio_layer"]
    return L - L // 4


def pred_ppg_from_mel(wsd, dims, mels, mel_noises, kept_frames):
    """whisper/inference.py:32-62 from the mel tensor on (the contract starts at the mel; the librosa
    front-end is row N2).  ``mels``: list of [80,n] windows (15 s each + remainder), ``mel_noises``:
    matching N(0,1) draws (scaled by 0.1 as in :46,58), ``kept_frames``: len//320 per window."""
    out = []
    for mel, nz, keep in zip(mels, mel_noises, kept_frames):
        x = (mel + nz * 0.1).unsqueeze(0)
        ppg = audio_encoder(wsd, x, dims["n_audio_head"], whisper_kept_layers(dims))[0]
        out.append(ppg[:keep])
    return torch.cat(out, 0)


# ----------------------------------------------------------------------------- driver
def chunk_schedule(all_frame, hop_size, out_chunk=C.CHUNK_FRAMES, hop_frame=C.HALO_FRAMES):
    """The chunk/halo arithmetic of svc_inference.py:94-131 as (cut_s, cut_e, cut_s_out, cut_e_out) tuples."""
    plan, out_index = [], 0
    while out_index < all_frame:
        if out_index == 0:
            cut_s, cut_s_out = 0, 0
        else:
            cut_s, cut_s_out = out_index - hop_frame, hop_frame * hop_size
        if out_index + out_chunk + hop_frame > all_frame:
            cut_e, cut_e_out = all_frame, -1
        else:
            cut_e, cut_e_out = out_index + out_chunk + hop_frame, -1 * hop_frame * hop_size
        plan.append((cut_s, cut_e, cut_s_out, cut_e_out))
        out_index += out_chunk
    return plan


def svc_infer(sd, hp, spk, pit, ppg, vec, rand_ini, src_noise, enc_noises):
    """svc_inference.py:77-134 (DummyRetrieval, :25-28).  spk [256], pit [T], ppg [T,1280], vec [T,256]
    already repeated x2 (:175-182).  ``enc_noises``: one [1,I,len] draw per chunk.  Returns
    (float32 waveform [L-1], int16 pitch wav)."""
    n = min(pit.shape[0], vec.shape[0], ppg.shape[0])
    pit, vec, ppg = pit[:n], vec[:n], ppg[:n]
    spk = spk.unsqueeze(0)
    source = pitch2source(sd, hp, pit.unsqueeze(0), rand_ini, src_noise)
    pitwav = source2wav(source)
    hop = hp.data.hop_length
    out = []
    for (cs, ce, cso, ceo), nz in zip(chunk_schedule(n, hop), enc_noises):
        sub = synth_inference(sd, hp, ppg[cs:ce].unsqueeze(0), vec[cs:ce].unsqueeze(0),
                              pit[cs:ce].unsqueeze(0), spk, torch.LongTensor([ce - cs]),
                              source[:, :, cs * hop:ce * hop], nz)
        out.append(sub[0, 0].numpy()[cso:ceo])
    return np.concatenate(out), pitwav


class SynthOracle:
    """Object with the reference's method names, for tests that read like calls on SynthesizerInfer."""

    def __init__(self, sd, hp=None):
        self.sd, self.hp = sd, hp or C.base_hp()

    def pitch2source(self, f0, rand_ini, noise):
        return pitch2source(self.sd, self.hp, f0, rand_ini, noise)

    def source2wav(self, source):
        return source2wav(source)

    def inference(self, ppg, vec, pit, spk, ppg_l, source, enc_noise):
        with torch.no_grad():
            return synth_inference(self.sd, self.hp, ppg, vec, pit, spk, ppg_l, source, enc_noise)
