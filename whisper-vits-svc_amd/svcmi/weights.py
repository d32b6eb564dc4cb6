"""Checkpoint -> kernel-ready weight arena (one-off, at load time; torch used as plumbing).

Consumes the reference checkpoint formats unchanged:
  * ``{"model_g": SynthesizerInfer.state_dict()}``  (svc_export.py:40-57, loaded by svc_inference.py:61-74)
  * ``{"dims", "model_state_dict"}``                 (whisper/inference.py:12-20)
and produces time-major GEMM operands:
  * weight-norm folded, ``w = g * v / ||v||`` over all dims but 0 -- the reference never removes it at
    inference (vits_decoder/generator.py:154-158) and recomputes it on every call;
  * Conv1d weight [N, C_in, K] -> [N, K*C_in] (tap-major, channel fastest), C_in/N zero-padded to x4;
  * ConvTranspose1d [C_in, C_out, K] (stride u, padding p) -> polyphase form: one GEMM with N = u*C_out
    output columns (phase-major) over the taps j of  y[u*q + r] = sum_j x[q - j] * w[:, :, u*j + r + p];
  * q/k/v projections concatenated into one [3C, C] operand; Flip (vits/modules.py:225-229) folded into
    row/column reversals of the coupling layers that see a flipped tensor.
"""
import math

import torch


def _round4(n):
    return (n + 3) // 4 * 4


def fold_weight_norm(sd, name):
    v, g = sd[name + ".weight_v"].float(), sd[name + ".weight_g"].float()
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / n


def pack_conv(w, cin_pad=None, n_pad=None):
    """[N, C_in, K] -> [N_pad, ldw] with element (n, k*cin_pad + ci); zero padded; ldw % 4 == 0."""
    N, Cin, Kk = w.shape
    cin_pad = cin_pad or Cin
    n_pad = n_pad or N
    p = torch.zeros(n_pad, Kk, cin_pad, dtype=torch.float32, device=w.device)
    p[:N, :, :Cin] = w.float().permute(0, 2, 1)
    p = p.reshape(n_pad, Kk * cin_pad)
    ldw = _round4(p.shape[1])
    if ldw != p.shape[1]:
        p = torch.cat([p, torch.zeros(n_pad, ldw - p.shape[1], dtype=torch.float32, device=w.device)], dim=1)
    return p.contiguous()


def pad_vec(v, n_pad):
    out = torch.zeros(n_pad, dtype=torch.float32, device=v.device)
    out[:v.shape[0]] = v.float()
    return out


def pack_conv_transpose(w, bias, u, p, cin_pad=None, cout_pad=None):
    """ConvTranspose1d weight [C_in, C_out, K] -> (W [u*cout_pad, taps*cin_pad], bias [u*cout_pad], taps, pad).

    y[u*q + r, co] = sum_j sum_ci x[q - j, ci] * w[ci, co, u*j + r + p]; tap index k' = jmax - j so that the
    GEMM's row is q + k' - jmax (conv with ksize = taps, pad = jmax)."""
    Cin, Cout, Kk = w.shape
    cin_pad = cin_pad or Cin
    cout_pad = cout_pad or Cout
    js = [j for r in range(u) for j in range(-Kk, Kk + 1) if 0 <= u * j + r + p < Kk]
    jmin, jmax = min(js), max(js)
    taps = jmax - jmin + 1
    W = torch.zeros(u, cout_pad, taps, cin_pad, dtype=torch.float32, device=w.device)
    for r in range(u):
        for j in range(jmin, jmax + 1):
            kk = u * j + r + p
            if 0 <= kk < Kk:
                W[r, :Cout, jmax - j, :Cin] = w[:, :, kk].float().t()
    b = torch.zeros(u, cout_pad, dtype=torch.float32, device=w.device)
    b[:, :Cout] = bias.float()[None, :]
    return W.reshape(u * cout_pad, taps * cin_pad).contiguous(), b.reshape(-1).contiguous(), taps, jmax


class VitsWeights:
    """Packed ``SynthesizerInfer`` parameters on one device."""

    def __init__(self, sd, hp, device):
        from .vits import consts as K
        f = lambda t: t.float().to(device).contiguous()
        H, I = hp.vits.hidden_channels, hp.vits.inter_channels
        self.hp = hp
        self.H, self.I, self.half = H, I, I // 2
        self.n_heads = K.ENC_HEADS
        # ---- prior encoder
        self.pre_w, self.pre_b = f(pack_conv(sd["enc_p.pre.weight"])), f(sd["enc_p.pre.bias"])
        self.hub_w, self.hub_b = f(pack_conv(sd["enc_p.hub.weight"])), f(sd["enc_p.hub.bias"])
        self.pit_emb = f(sd["enc_p.pit.weight"])
        self.enc = []
        for i in range(K.ENC_LAYERS):
            a = f"enc_p.enc.attn_layers.{i}"
            qkv_w = torch.cat([sd[a + ".conv_q.weight"], sd[a + ".conv_k.weight"], sd[a + ".conv_v.weight"]], 0)
            qkv_b = torch.cat([sd[a + ".conv_q.bias"], sd[a + ".conv_k.bias"], sd[a + ".conv_v.bias"]], 0)
            self.enc.append(dict(
                qkv_w=f(pack_conv(qkv_w)), qkv_b=f(qkv_b),
                o_w=f(pack_conv(sd[a + ".conv_o.weight"])), o_b=f(sd[a + ".conv_o.bias"]),
                rel_k=f(sd[a + ".emb_rel_k"][0]), rel_v=f(sd[a + ".emb_rel_v"][0]),
                g1=f(sd[f"enc_p.enc.norm_layers_1.{i}.gamma"]), b1=f(sd[f"enc_p.enc.norm_layers_1.{i}.beta"]),
                f1_w=f(pack_conv(sd[f"enc_p.enc.ffn_layers.{i}.conv_1.weight"])), f1_b=f(sd[f"enc_p.enc.ffn_layers.{i}.conv_1.bias"]),
                f2_w=f(pack_conv(sd[f"enc_p.enc.ffn_layers.{i}.conv_2.weight"])), f2_b=f(sd[f"enc_p.enc.ffn_layers.{i}.conv_2.bias"]),
                g2=f(sd[f"enc_p.enc.norm_layers_2.{i}.gamma"]), b2=f(sd[f"enc_p.enc.norm_layers_2.{i}.beta"])))
        self.proj_w, self.proj_b = f(pack_conv(sd["enc_p.proj.weight"])), f(sd["enc_p.proj.bias"])
        # ---- flow, in execution order of reverse=True: (Flip, RCL3), (Flip, RCL2), (Flip, RCL1), (Flip, RCL0)
        # The physical tensor is never permuted; `flipped` tracks whether logical channel c sits at C-1-c.
        half = self.half
        self.flow = []
        flipped = False
        for fl in reversed(range(K.FLOW_N)):
            flipped = not flipped
            p = f"flow.flows.{2 * fl}"
            pre_w, post_w, post_b = sd[p + ".pre.weight"].float(), sd[p + ".post.weight"].float(), sd[p + ".post.bias"].float()
            snac_w, snac_b = sd[p + ".snac.weight"].float()[:, :, 0], sd[p + ".snac.bias"].float()
            m_w, v_w, m_b, v_b = snac_w[:half], snac_w[half:], snac_b[:half], snac_b[half:]
            if flipped:
                # logical x0[c] = phys[C-1-c] = upper half reversed; logical x1[c] = phys[half-1-c]
                pre_w = pre_w.flip(1)
                post_w, post_b = post_w.flip(0), post_b.flip(0)
                m_w, v_w, m_b, v_b = m_w.flip(0), v_w.flip(0), m_b.flip(0), v_b.flip(0)
            layer = dict(x0_off=half if flipped else 0, x1_off=0 if flipped else half,
                         # pre writes (h | skip) rows of 2H: the H zero output channels clear the skip accumulator
                         pre_w=f(pack_conv(pre_w, n_pad=2 * H)), pre_b=f(pad_vec(sd[p + ".pre.bias"], 2 * H)),
                         post_w=f(pack_conv(post_w)), post_b=f(post_b),
                         snac_w=f(pack_conv(torch.cat([m_w, v_w], 0).unsqueeze(-1))), snac_b=f(torch.cat([m_b, v_b], 0)),
                         wn=[])
            for l in range(K.FLOW_WN_LAYERS):
                layer["wn"].append(dict(
                    in_w=f(pack_conv(fold_weight_norm(sd, f"{p}.enc.in_layers.{l}"))), in_b=f(sd[f"{p}.enc.in_layers.{l}.bias"]),
                    rs_w=f(pack_conv(fold_weight_norm(sd, f"{p}.enc.res_skip_layers.{l}"))), rs_b=f(sd[f"{p}.enc.res_skip_layers.{l}.bias"])))
            self.flow.append(layer)
        assert not flipped, "an even number of Flip modules leaves the tensor un-permuted"
        # ---- generator
        g = hp.gen
        self.rates, self.up_k = list(g.upsample_rates), list(g.upsample_kernel_sizes)
        self.rk = list(g.resblock_kernel_sizes)
        self.rd = [list(d) for d in g.resblock_dilation_sizes]
        self.hop = int(math.prod(self.rates))
        self.ad_w = f(pack_conv(torch.cat([sd["dec.adapter.W_scale.weight"], sd["dec.adapter.W_bias.weight"]], 0).unsqueeze(-1)))
        self.ad_b = f(torch.cat([sd["dec.adapter.W_scale.bias"], sd["dec.adapter.W_bias.bias"]], 0))
        self.U = g.upsample_input
        self.pre_conv_w, self.pre_conv_b = f(pack_conv(sd["dec.conv_pre.weight"])), f(sd["dec.conv_pre.bias"])
        self.merge_w = f(sd["dec.m_source.merge_w"].view(-1))
        self.merge_b = float(sd["dec.m_source.merge_b"].view(-1)[0])
        self.filt = f(sd["dec.activation_post.upsample.filter"].view(-1))
        C0 = g.upsample_initial_channel
        self.stages = []
        for i, (u, k) in enumerate(zip(self.rates, self.up_k)):
            cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
            cin_p, cout_p = _round4(cin), _round4(cout)
            up_w, up_b, taps, upad = pack_conv_transpose(fold_weight_norm(sd, f"dec.ups.{i}"), sd[f"dec.ups.{i}.bias"],
                                                         u, (k - u) // 2, cin_p, cout_p)
            nw = sd[f"dec.noise_convs.{i}.weight"]
            s = int(math.prod(self.rates[i + 1:])) if i + 1 < len(self.rates) else 1
            st = dict(u=u, c=cout, cp=cout_p, up_w=f(up_w), up_b=f(up_b), up_taps=taps, up_pad=upad,
                      nz_w=f(pack_conv(nw, n_pad=cout_p)), nz_b=f(pad_vec(sd[f"dec.noise_convs.{i}.bias"], cout_p)),
                      nz_k=nw.shape[2], nz_stride=s, nz_pad=(s // 2 if i + 1 < len(self.rates) else 0), blocks=[])
            for j, kk in enumerate(self.rk):
                b = f"dec.resblocks.{i * len(self.rk) + j}"
                blk = dict(k=kk, d=self.rd[j], c1=[], c2=[], a1=[], a2=[])
                for q in range(3):
                    blk["c1"].append((f(pack_conv(fold_weight_norm(sd, f"{b}.convs1.{q}"), cout_p, cout_p)), f(pad_vec(sd[f"{b}.convs1.{q}.bias"], cout_p))))
                    blk["c2"].append((f(pack_conv(fold_weight_norm(sd, f"{b}.convs2.{q}"), cout_p, cout_p)), f(pad_vec(sd[f"{b}.convs2.{q}.bias"], cout_p))))
                    blk["a1"].append((f(pad_vec(sd[f"{b}.activations.{2 * q}.act.alpha"], cout_p)), f(pad_vec(sd[f"{b}.activations.{2 * q}.act.beta"], cout_p))))
                    blk["a2"].append((f(pad_vec(sd[f"{b}.activations.{2 * q + 1}.act.alpha"], cout_p)), f(pad_vec(sd[f"{b}.activations.{2 * q + 1}.act.beta"], cout_p))))
                st["blocks"].append(blk)
            self.stages.append(st)
        cl = self.stages[-1]["cp"]
        self.post_a = (f(pad_vec(sd["dec.activation_post.act.alpha"], cl)), f(pad_vec(sd["dec.activation_post.act.beta"], cl)))
        self.post_w = f(pack_conv(sd["dec.conv_post.weight"], cin_pad=cl))


class WhisperWeights:
    """Packed truncated Whisper audio encoder (first 3/4 of the blocks, whisper/inference.py:16-19)."""

    def __init__(self, ckpt, device):
        dims, sd = ckpt["dims"], ckpt["model_state_dict"]
        f = lambda t: t.float().to(device).contiguous()
        self.dims = dict(dims) if isinstance(dims, dict) else dict(vars(dims))
        S, L = self.dims["n_audio_state"], self.dims["n_audio_layer"]
        self.S, self.heads = S, self.dims["n_audio_head"]
        self.n_layers = L - L // 4
        self.n_mels = self.dims["n_mels"]
        self.conv1_w, self.conv1_b = f(pack_conv(sd["encoder.conv1.weight"])), f(sd["encoder.conv1.bias"])
        self.conv2_w, self.conv2_b = f(pack_conv(sd["encoder.conv2.weight"])), f(sd["encoder.conv2.bias"])
        self.pos = f(sd["encoder.positional_embedding"]) if "encoder.positional_embedding" in sd else None
        self.blocks = []
        zeros = torch.zeros(S)
        for i in range(self.n_layers):
            b = f"encoder.blocks.{i}"
            qkv_w = torch.cat([sd[b + ".attn.query.weight"], sd[b + ".attn.key.weight"], sd[b + ".attn.value.weight"]], 0)
            qkv_b = torch.cat([sd[b + ".attn.query.bias"].float().cpu(), zeros, sd[b + ".attn.value.bias"].float().cpu()], 0)
            self.blocks.append(dict(
                ln1_g=f(sd[b + ".attn_ln.weight"]), ln1_b=f(sd[b + ".attn_ln.bias"]),
                qkv_w=f(qkv_w), qkv_b=f(qkv_b),
                o_w=f(sd[b + ".attn.out.weight"]), o_b=f(sd[b + ".attn.out.bias"]),
                ln2_g=f(sd[b + ".mlp_ln.weight"]), ln2_b=f(sd[b + ".mlp_ln.bias"]),
                m1_w=f(sd[b + ".mlp.0.weight"]), m1_b=f(sd[b + ".mlp.0.bias"]),
                m2_w=f(sd[b + ".mlp.2.weight"]), m2_b=f(sd[b + ".mlp.2.bias"])))
        self.lnp_g, self.lnp_b = f(sd["encoder.ln_post.weight"]), f(sd["encoder.ln_post.bias"])


class HubertWeights:
    """Packed ``HubertSoft`` parameters (hubert/hubert_model.py state-dict keys) on one device; dimensions are read
    from the tensor shapes."""

    def __init__(self, sd, device):
        f = lambda t: t.float().to(device).contiguous()
        self.conv0_w = f(pack_conv(sd["feature_extractor.conv0.weight"]))
        self.norm0_g, self.norm0_b = f(sd["feature_extractor.norm0.weight"]), f(sd["feature_extractor.norm0.bias"])
        self.convs = []
        i = 1
        while f"feature_extractor.conv{i}.weight" in sd:
            w = sd[f"feature_extractor.conv{i}.weight"]
            self.convs.append((f(pack_conv(w)), int(w.shape[2])))
            i += 1
        self.fp_g, self.fp_b = f(sd["feature_projection.norm.weight"]), f(sd["feature_projection.norm.bias"])
        self.fp_w = f(pack_conv(sd["feature_projection.projection.weight"].unsqueeze(-1)))
        self.fp_b2 = f(sd["feature_projection.projection.bias"])
        # positional conv: weight_norm(dim=2) -> one norm per kernel tap over (out, in) (hubert_model.py:123-124)
        v, g = sd["positional_embedding.conv.weight_v"].float(), sd["positional_embedding.conv.weight_g"].float()
        w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
        self.E, self.pos_k = int(v.shape[0]), int(v.shape[2])
        self.G = self.E // int(v.shape[1])
        cg = self.E // self.G
        pb = sd["positional_embedding.conv.bias"].float()
        self.pos_w = [f(pack_conv(w[gi * cg:(gi + 1) * cg])) for gi in range(self.G)]
        self.pos_b = [f(pb[gi * cg:(gi + 1) * cg]) for gi in range(self.G)]
        self.norm_g, self.norm_b = f(sd["norm.weight"]), f(sd["norm.bias"])
        self.layers = []
        i = 0
        while f"encoder.layers.{i}.norm1.weight" in sd:
            p = f"encoder.layers.{i}."
            self.layers.append(dict(
                in_w=f(pack_conv(sd[p + "self_attn.in_proj_weight"].unsqueeze(-1))), in_b=f(sd[p + "self_attn.in_proj_bias"]),
                out_w=f(pack_conv(sd[p + "self_attn.out_proj.weight"].unsqueeze(-1))), out_b=f(sd[p + "self_attn.out_proj.bias"]),
                l1_w=f(pack_conv(sd[p + "linear1.weight"].unsqueeze(-1))), l1_b=f(sd[p + "linear1.bias"]),
                l2_w=f(pack_conv(sd[p + "linear2.weight"].unsqueeze(-1))), l2_b=f(sd[p + "linear2.bias"]),
                n1_g=f(sd[p + "norm1.weight"]), n1_b=f(sd[p + "norm1.bias"]),
                n2_g=f(sd[p + "norm2.weight"]), n2_b=f(sd[p + "norm2.bias"])))
            i += 1
        # heads: the reference fixes 12 x 64 (hubert_model.py:21); smaller test models keep head_dim 16
        self.heads = 12 if self.E == 768 else self.E // 16
        self.proj_w, self.proj_b = f(pack_conv(sd["proj.weight"].unsqueeze(-1))), f(sd["proj.bias"])


CREPE_FRAME_ROWS = 256          # rows out of the first CREPE layer for a 1024-sample frame (stride 4)
CREPE_DENSE_MAX_ROWS = 32       # layers with at most this many input rows run as one dense GEMM over the frames (CrepeWeights)


class CrepeWeights:
    """Packed ``crepe.Crepe`` parameters (crepe/model.py state-dict keys): Conv2d [out, in, k, 1] -> implicit-GEMM
    operand [out, k*in]; eval-mode BatchNorm2d (eps 0.0010000000474974513) -> per-channel scale / shift."""

    def __init__(self, sd, device, eps=0.0010000000474974513):
        f = lambda t: t.float().to(device).contiguous()
        self.layers = []
        i = 1
        while f"conv{i}.weight" in sd:
            w = sd[f"conv{i}.weight"].float()[:, :, :, 0]                       # [out, in, k]
            if i == 1:      # in == 1: k = 512 taps read as 128 taps x 4 "channels"; the flat order is already tap-major
                packed = w[:, 0, :].contiguous()
            else:
                packed = pack_conv(w)
            g, b = sd[f"conv{i}_BN.weight"].float(), sd[f"conv{i}_BN.bias"].float()
            mu, var = sd[f"conv{i}_BN.running_mean"].float(), sd[f"conv{i}_BN.running_var"].float()
            scale = g / torch.sqrt(var + eps)
            layer = dict(w=f(packed), b=f(sd[f"conv{i}.bias"]), scale=f(scale), shift=f(b - mu * scale))
            # Short layers as ONE dense GEMM over the frames.  A 1024-sample frame leaves t_in = 256 / 2^(i-1) rows for layer i; at
            # t_in <= 32 every output row sees every input row (64 taps, pad 31 / 32), so the layer is the linear map
            #     y[f, (t_out, n)] = sum_{t_in, c} x[f, (t_in, c)] * w[n, c, t_in - t_out + 31]
            # with the frames as GEMM rows: no 64-row tile padded from 8-32 time steps and no taps that only ever meet the zero
            # padding (64x / 16x / 4x less matrix work than the implicit convolution for t_in = 8 / 16 / 32).
            t_in = CREPE_FRAME_ROWS >> (i - 1)
            if i > 1 and t_in <= CREPE_DENSE_MAX_ROWS:
                n_out, c_in, taps = w.shape
                ti = torch.arange(t_in)
                k = ti[None, :] - ti[:, None] + (taps // 2 - 1)                       # [t_out, t_in] tap index, always inside 0..taps-1 here
                assert int(k.min()) >= 0 and int(k.max()) < taps
                dense = w[:, :, k].permute(2, 0, 3, 1).reshape(t_in * n_out, t_in * c_in)      # [(t_out, n), (t_in, c)]
                layer.update(dense_w=f(dense), dense_b=f(sd[f"conv{i}.bias"].float().repeat(t_in)), t_in=t_in)
            self.layers.append(layer)
            i += 1
        self.fc_w = f(pack_conv(sd["classifier.weight"].float().unsqueeze(-1)))
        self.fc_b = f(sd["classifier.bias"])
