"""Packed weights -> the C model structs of the stage-level entry points (include/svcmi.h: svcmi_whisper_model,
svcmi_synth_model).  The structs only carry device pointers and dimensions: the tensors stay owned by the ``VitsWeights`` /
``WhisperWeights`` object (and the 16-bit images by the per-tensor cache of ``Ops.lp_weight``), which the returned handle keeps
alive."""
from . import _lib
from ._lib import PREC_BF16, PREC_BF16X3, PREC_F16, PREC_F32
from .vits import consts as K


class CModel:
    """A filled C struct + everything its pointers refer to."""

    def __init__(self, struct, keep):
        self.struct, self.keep = struct, keep


def _weight(cw, ops, w, bias, prec, keep, a16=False):
    """``a16``: also pack the natural-order image of the 16-bit-activation kernels."""
    cw.w, cw.bias = w.data_ptr(), (0 if bias is None else bias.data_ptr())
    cw.n, cw.ldw = int(w.shape[0]), int(w.shape[1])
    cw.w16, cw.w16a, cw.ldw16, cw.prec16 = 0, 0, 0, 0
    if prec != PREC_F32 and w.dim() == 2 and w.is_contiguous():
        base = PREC_F16 if prec == _lib.PREC_F16W2 else prec               # f16w2: the launches without 16-bit activations are plain fp16
        img = ops.lp_weight(w, base)
        cw.w16, cw.ldw16, cw.prec16 = img.data_ptr(), img.shape[1] // (2 if prec == PREC_BF16X3 else 1), prec
        keep.append(img)
        if a16:
            a16_code = {PREC_BF16X3: _lib.PREC_BF16X3_A16, _lib.PREC_F16W2: _lib.PREC_F16W2_A16}.get(prec, prec + 2)
            img_a = ops.lp_weight(w, a16_code)          # PREC_BF16_A16 / PREC_F16_A16 / PREC_BF16X3_A16 / PREC_F16W2_A16 ([hi | lo] rows)
            cw.w16a = img_a.data_ptr()
            keep.append(img_a)
    keep.append(w)
    keep.append(bias)


def whisper_cmodel(w, ops, prec=PREC_F32):
    """``w``: svcmi.weights.WhisperWeights."""
    m, keep = _lib.WhisperModel(), [w]
    if w.n_layers > _lib.MAX_WHISPER_BLOCKS:
        raise _lib.SvcmiError(f"{w.n_layers} encoder blocks > {_lib.MAX_WHISPER_BLOCKS}")
    m.n_state, m.n_heads, m.n_layers, m.n_mels = w.S, w.heads, w.n_layers, w.n_mels
    m.n_ctx, m.precision = int(w.pos.shape[0]), prec
    x3 = prec == PREC_BF16X3         # split-bf16: only the QKV projection takes 16-bit (split) activation rows (host_stages.hip)
    _weight(m.conv1, ops, w.conv1_w, w.conv1_b, prec, keep)
    _weight(m.conv2, ops, w.conv2_w, w.conv2_b, prec, keep)
    m.pos, m.lnp_g, m.lnp_b = w.pos.data_ptr(), w.lnp_g.data_ptr(), w.lnp_b.data_ptr()
    for i, b in enumerate(w.blocks):
        cb = m.blocks[i]
        cb.ln1_g, cb.ln1_b, cb.ln2_g, cb.ln2_b = (b[k].data_ptr() for k in ("ln1_g", "ln1_b", "ln2_g", "ln2_b"))
        _weight(cb.qkv, ops, b["qkv_w"], b["qkv_b"], prec, keep, a16=True)
        _weight(cb.o, ops, b["o_w"], b["o_b"], prec, keep, a16=not x3)
        _weight(cb.m1, ops, b["m1_w"], b["m1_b"], prec, keep, a16=not x3)
        _weight(cb.m2, ops, b["m2_w"], b["m2_b"], prec, keep, a16=not x3)
    return CModel(m, keep)


def synth_cmodel(w, ops, prec=PREC_F32):
    """``w``: svcmi.weights.VitsWeights.  ``prec``: one mode for every GEMM (an ``enum svcmi_precision`` code) or the
    ``(PREC_MIXED, class modes)`` tuple of ``_lib.parse_precision``: every weight then gets the 16-bit images of ITS class's mode."""
    code, classes = prec if isinstance(prec, tuple) else (prec, None)
    m, keep = _lib.SynthModel(), [w]
    hp = w.hp
    if len(w.enc) > _lib.MAX_ENC_LAYERS or len(w.flow) > _lib.MAX_FLOWS or len(w.stages) > _lib.MAX_STAGES:
        raise _lib.SvcmiError("model exceeds the fixed capacities of svcmi_synth_model")
    m.hidden, m.inter, m.n_heads = w.H, w.I, w.n_heads
    m.enc_window, m.enc_ffn_kernel, m.flow_kernel = K.ENC_WINDOW, K.ENC_FFN_KERNEL, K.FLOW_KERNEL
    m.n_enc, m.n_flow, m.n_stages = len(w.enc), len(w.flow), len(w.stages)
    m.ppg_dim, m.vec_dim, m.spk_dim = hp.vits.ppg_dim, hp.vits.vec_dim, hp.vits.spk_dim
    m.upsample_input, m.hop = w.U, w.hop
    m.precision, m.lp_min_flops = code, 0.0
    for i in range(_lib.PREC_CLASSES):
        m.class_prec[i] = classes[i] if classes else 0
    m.sampling_rate, m.merge_b = float(hp.data.sampling_rate), float(w.merge_b)
    state = {"cls": _lib.CLASS_ENC}

    def W(cw, wt, b=None, a16=False):
        p = classes[state["cls"]] if classes else code
        _weight(cw, ops, wt, b, p, keep, a16=a16 and p != PREC_BF16X3)     # 16-bit activation rows: bf16 / f16 modes
    W(m.pre, w.pre_w, w.pre_b)
    W(m.hub, w.hub_w, w.hub_b)
    W(m.proj, w.proj_w, w.proj_b)
    m.pit_emb = w.pit_emb.data_ptr()
    for i, L in enumerate(w.enc):
        e = m.enc[i]
        W(e.qkv, L["qkv_w"], L["qkv_b"], a16=True)          # (16-bit activations from LayerNorm / attention / the ReLU epilogue in the bf16 / f16 modes)
        W(e.o, L["o_w"], L["o_b"], a16=True)
        W(e.f1, L["f1_w"], L["f1_b"], a16=True)
        W(e.f2, L["f2_w"], L["f2_b"], a16=True)
        e.rel_k, e.rel_v = L["rel_k"].data_ptr(), L["rel_v"].data_ptr()
        e.g1, e.b1, e.g2, e.b2 = (L[k].data_ptr() for k in ("g1", "b1", "g2", "b2"))
    state["cls"] = _lib.CLASS_FLOW
    for i, L in enumerate(w.flow):
        f = m.flow[i]
        if len(L["wn"]) > _lib.MAX_WN_LAYERS:
            raise _lib.SvcmiError("too many WN layers")
        f.x0_off, f.x1_off, f.n_wn = L["x0_off"], L["x1_off"], len(L["wn"])
        W(f.pre, L["pre_w"], L["pre_b"])
        W(f.post, L["post_w"], L["post_b"])
        W(f.snac, L["snac_w"], L["snac_b"])
        for l, Wl in enumerate(L["wn"]):
            W(f.wn[l].in_, Wl["in_w"], Wl["in_b"])
            W(f.wn[l].rs, Wl["rs_w"], Wl["rs_b"])
    state["cls"] = _lib.CLASS_UPS
    W(m.adapter, w.ad_w, w.ad_b)
    W(m.conv_pre, w.pre_conv_w, w.pre_conv_b)
    W(m.post, w.post_w, None)
    m.merge_w, m.filt = w.merge_w.data_ptr(), w.filt.data_ptr()
    m.post_alpha, m.post_beta = w.post_a[0].data_ptr(), w.post_a[1].data_ptr()
    for i, st in enumerate(w.stages):
        s = m.stages[i]
        if len(st["blocks"]) > _lib.MAX_AMP_BLOCKS:
            raise _lib.SvcmiError("more AMP blocks per stage than svcmi_gen_stage holds")
        s.u, s.c, s.cp, s.up_taps, s.up_pad = st["u"], st["c"], st["cp"], st["up_taps"], st["up_pad"]
        s.nz_k, s.nz_stride, s.nz_pad, s.n_blocks = st["nz_k"], st["nz_stride"], st["nz_pad"], len(st["blocks"])
        state["cls"] = _lib.CLASS_UPS
        W(s.up, st["up_w"], st["up_b"])
        W(s.nz, st["nz_w"], st["nz_b"])
        state["cls"] = _lib.CLASS_AMP0 + min(i, _lib.AMP_CLASSES - 1)        # a sixth stage shares amp4 (never the encoder-attention class at index 8)
        for j, blk in enumerate(st["blocks"]):
            b = s.blocks[j]
            if len(blk["d"]) > _lib.MAX_AMP_DILATIONS:
                raise _lib.SvcmiError("more dilations per AMP block than svcmi_amp_block holds")
            b.k, b.n_dil = blk["k"], len(blk["d"])
            for q, d in enumerate(blk["d"]):
                b.dil[q] = d
                W(b.c1[q], blk["c1"][q][0], blk["c1"][q][1], a16=st["cp"] % 8 == 0)      # wide stages: SnakeAlias hands 16-bit rows to the GEMM
                W(b.c2[q], blk["c2"][q][0], blk["c2"][q][1], a16=st["cp"] % 8 == 0)
                b.a1_alpha[q], b.a1_beta[q] = blk["a1"][q][0].data_ptr(), blk["a1"][q][1].data_ptr()
                b.a2_alpha[q], b.a2_beta[q] = blk["a2"][q][0].data_ptr(), blk["a2"][q][1].data_ptr()
    return CModel(m, keep)
