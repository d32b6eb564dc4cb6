"""ctypes binding of libsvcmi.so (include/svcmi.h).  No fallback: a missing library raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libsvcmi.so")

ACT_NONE, ACT_RELU, ACT_GELU, ACT_MISH, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4, 5
CONV_ACCUMULATE, CONV_MASK_IN, CONV_MASK_OUT, CONV_PARTIALS = 1, 2, 4, 8
ABI_VERSION = 22
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16, PREC_BF16_A16, PREC_F16_A16, PREC_BF16X3_A16 = 0, 1, 2, 3, 4, 5, 6
PRECISIONS = {None: 0, "f32": 0, "fp32": 0, "bf16x3": 1, "bf16": 2, "f16": 3, "fp16": 3, "f16w2": 8}
PREC_F16W2, PREC_F16W2_A16 = 8, 9          # fp16 activations x split fp16 weights: a mode like f16 whose 16-bit-activation launches take two MFMAs on (hi, lo) weight images
PREC_MIXED = 7                     # svcmi_synth_model.precision only: per-class modes in class_prec[]
PREC_CLASSES = 12
CLASS_ENC, CLASS_FLOW, CLASS_UPS, CLASS_AMP0 = 0, 1, 2, 3
AMP_CLASSES = 5                    # SVCMI_AMP_CLASSES
CLASS_NAMES = {"enc": 0, "flow": 1, "ups": 2, "amp0": 3, "amp1": 4, "amp2": 5, "amp3": 6, "amp4": 7, "encattn": 8}
# The default per-layer policy of the 16-bit synthesizer ("mixed"; scripts/precision_sensitivity.py ranks the classes on the CPU oracle,
# tests/test_gpu_precision.py measures the candidates on MI355X -- profiles/r04e_precision_report.json): the layers every sample passes
# through once with large fan-in -- conv_pre and the transposed convolutions (half of the fp16 waveform error for 2 % of the FLOPs) --
# and the prior encoder (LayerNorm gains / outlier channels land here) run in split-bf16 (bf16x3: fp32-class products), stage 0 of the
# generator too; the flow and the remaining AMP convolutions -- three quarters of the FLOPs -- run in fp16 on 16-bit activations.
# configs[2]: 3.6e-4 on the waveform against the fp32 oracle (plain f16 8.5e-4, bf16 7.5e-3) at 93 % of the f16 speed.  The prior
# encoder's attention stays on the fp32 matrix cores: with "encattn": "f16" (svcmi_attention16) the step is 4 % faster and as
# accurate on ordinary weights (3.2e-4), but on the outlier-stress weights (LayerNorm gains up to 30: logits of hundreds) the fp16
# q / k rounding costs 1.7e-2 against 1.9e-3.
# The two narrow stages (amp3 / amp4: 20 and 10 channels, fused SnakeAlias + convolution kernels) run their convolution on the fp16
# matrix cores with split weights (f16w2: only the activated input is rounded; svcmi_snake_conv_group_lp): 50 instead of 93 us per
# grouped half-step at 20 channels, 49 instead of 65 at 10 (profiles/r04p_amplp.log) for 3.9e-4 -> 4.3e-4 on configs[2]
# (plain "f16" there: 1.5 % faster, 4.7e-4).
MIXED_DEFAULT = {"enc": "bf16x3", "encattn": "f32", "ups": "bf16x3", "flow": "f16", "amp0": "bf16x3", "amp1": "f16", "amp2": "f16", "amp3": "f16w2", "amp4": "f16w2"}


def parse_precision(p):
    """Model-level precision spec -> (code, class tuple).  ``p``: None / "f32" / "bf16x3" / "bf16" / "f16" (one mode for every GEMM),
    "mixed" (MIXED_DEFAULT), "mixed:enc=bf16x3,flow=f16,..." (unnamed classes keep their MIXED_DEFAULT mode), a dict of the same, an
    int code, or an already parsed tuple."""
    if isinstance(p, tuple):
        return p
    if isinstance(p, int) and not isinstance(p, bool):
        if p == PREC_MIXED:
            raise SvcmiError("precision code SVCMI_PREC_MIXED needs its per-class modes: pass \"mixed\", \"mixed:...\" or a dict")
        if p not in set(PRECISIONS.values()):
            raise SvcmiError(f"unknown precision code {p}")
        return (p, None)
    if (p is None or isinstance(p, str)) and p in PRECISIONS:
        return (PRECISIONS[p], None)
    pol = dict(MIXED_DEFAULT)
    if isinstance(p, dict):
        pol.update(p)
    elif isinstance(p, str) and (p == "mixed" or p.startswith("mixed:")):
        for item in filter(None, p[6:].split(",")):
            k, _, v = item.partition("=")
            pol[k.strip()] = v.strip()
    else:
        raise SvcmiError(f"unknown precision {p!r}")
    bad = [k for k in pol if k not in CLASS_NAMES] + [v for v in pol.values() if v not in PRECISIONS]
    if bad:
        raise SvcmiError(f"mixed precision policy: unknown class / mode {bad}; classes {sorted(CLASS_NAMES)}, modes f32 / bf16x3 / bf16 / f16 / f16w2")
    cls = [0] * PREC_CLASSES
    for k, v in pol.items():
        cls[CLASS_NAMES[k]] = PRECISIONS[v]
    return (PREC_MIXED, tuple(cls))
CONV_TILE_64x128 = 9


class SnakeConvDesc(Structure):
    """svcmi_snake_conv_desc"""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("y", c_void_p),
                ("alpha_log", c_void_p), ("beta_log", c_void_p),
                ("ldw", c_int32), ("ksize", c_int32), ("dilation", c_int32), ("accumulate", c_int32), ("alpha", c_float)]


class ConvDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("y", c_void_p), ("lengths", c_void_p),
        ("x_bstride", c_int64), ("y_bstride", c_int64), ("res_bstride", c_int64),
        ("batch", c_int32), ("t_in", c_int32), ("t_out", c_int32), ("c_in", c_int32), ("ldx", c_int32),
        ("n_out", c_int32), ("ldw", c_int32), ("ldy", c_int32), ("ldr", c_int32),
        ("ksize", c_int32), ("stride", c_int32), ("dilation", c_int32), ("pad", c_int32), ("x_row_shift", c_int32),
        ("act", c_int32), ("flags", c_int32), ("alpha", c_float), ("split_k", c_int32),
        ("workspace", c_void_p), ("workspace_floats", c_int64),
        ("counters", c_void_p), ("counters_len", c_int64),
        ("y16", c_void_p), ("y16_bstride", c_int64), ("ldy16", c_int32), ("y16_format", c_int32),
    ]


# ---- stage-level entry points (include/svcmi.h "Stage-level entry points"): ctypes mirrors of the model / io structs
MAX_WHISPER_BLOCKS, MAX_ENC_LAYERS, MAX_FLOWS, MAX_WN_LAYERS, MAX_STAGES, MAX_AMP_BLOCKS, MAX_AMP_DILATIONS = 32, 8, 8, 8, 6, 3, 3
STOP_NONE, STOP_PRIOR, STOP_FLOW, STOP_GEN_PRE, STOP_STAGE0 = 0, 1, 2, 3, 4


class Weight(Structure):
    """svcmi_weight"""
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("w16", c_void_p), ("w16a", c_void_p), ("n", c_int32), ("ldw", c_int32),
                ("ldw16", c_int32), ("prec16", c_int32)]


class WhisperBlock(Structure):
    _fields_ = [("ln1_g", c_void_p), ("ln1_b", c_void_p), ("ln2_g", c_void_p), ("ln2_b", c_void_p),
                ("qkv", Weight), ("o", Weight), ("m1", Weight), ("m2", Weight)]


class WhisperModel(Structure):
    _fields_ = [("n_state", c_int32), ("n_heads", c_int32), ("n_layers", c_int32), ("n_mels", c_int32), ("n_ctx", c_int32),
                ("precision", c_int32), ("conv1", Weight), ("conv2", Weight), ("pos", c_void_p), ("lnp_g", c_void_p), ("lnp_b", c_void_p),
                ("blocks", WhisperBlock * MAX_WHISPER_BLOCKS),
                ("split_o", c_int32), ("split_mlp", c_int32), ("tile_qkv", c_int32), ("tile_o", c_int32), ("tile_mlp1", c_int32),
                ("tile_mlp2", c_int32), ("small_m_rows", c_int32), ("reserved", c_int32), ("lp_min_flops", c_float), ("reserved2", c_int32)]


class EncLayer(Structure):
    _fields_ = [("qkv", Weight), ("o", Weight), ("f1", Weight), ("f2", Weight), ("rel_k", c_void_p), ("rel_v", c_void_p),
                ("g1", c_void_p), ("b1", c_void_p), ("g2", c_void_p), ("b2", c_void_p)]


class WnLayer(Structure):
    _fields_ = [("in_", Weight), ("rs", Weight)]


class FlowLayer(Structure):
    _fields_ = [("x0_off", c_int32), ("x1_off", c_int32), ("n_wn", c_int32), ("reserved", c_int32),
                ("pre", Weight), ("post", Weight), ("snac", Weight), ("wn", WnLayer * MAX_WN_LAYERS)]


class AmpBlock(Structure):
    _fields_ = [("k", c_int32), ("n_dil", c_int32), ("dil", c_int32 * MAX_AMP_DILATIONS), ("reserved", c_int32),
                ("c1", Weight * MAX_AMP_DILATIONS), ("c2", Weight * MAX_AMP_DILATIONS),
                ("a1_alpha", c_void_p * MAX_AMP_DILATIONS), ("a1_beta", c_void_p * MAX_AMP_DILATIONS),
                ("a2_alpha", c_void_p * MAX_AMP_DILATIONS), ("a2_beta", c_void_p * MAX_AMP_DILATIONS)]


class GenStage(Structure):
    _fields_ = [("u", c_int32), ("c", c_int32), ("cp", c_int32), ("up_taps", c_int32), ("up_pad", c_int32), ("nz_k", c_int32),
                ("nz_stride", c_int32), ("nz_pad", c_int32), ("n_blocks", c_int32), ("reserved", c_int32),
                ("up", Weight), ("nz", Weight), ("blocks", AmpBlock * MAX_AMP_BLOCKS)]


class SynthModel(Structure):
    _fields_ = [("hidden", c_int32), ("inter", c_int32), ("n_heads", c_int32), ("enc_window", c_int32), ("enc_ffn_kernel", c_int32),
                ("flow_kernel", c_int32), ("n_enc", c_int32), ("n_flow", c_int32), ("n_stages", c_int32),
                ("ppg_dim", c_int32), ("vec_dim", c_int32), ("spk_dim", c_int32), ("upsample_input", c_int32), ("hop", c_int32),
                ("precision", c_int32), ("lp_min_flops", c_float), ("sampling_rate", c_float), ("merge_b", c_float),
                ("class_prec", c_int32 * PREC_CLASSES),
                ("pre", Weight), ("hub", Weight), ("proj", Weight), ("pit_emb", c_void_p),
                ("enc", EncLayer * MAX_ENC_LAYERS), ("flow", FlowLayer * MAX_FLOWS),
                ("adapter", Weight), ("conv_pre", Weight), ("post", Weight),
                ("merge_w", c_void_p), ("filt", c_void_p), ("post_alpha", c_void_p), ("post_beta", c_void_p),
                ("stages", GenStage * MAX_STAGES)]


class SynthIO(Structure):
    _fields_ = [("ppg", c_void_p), ("vec", c_void_p), ("pit", c_void_p), ("spk", c_void_p), ("lengths", c_void_p), ("source", c_void_p),
                ("noise", c_void_p), ("ppg_bstride", c_int64), ("ppg_row_shift", c_int32), ("batch", c_int32), ("t", c_int32),
                ("stream_frames", c_int32), ("stop_after", c_int32), ("reserved", c_int32),
                ("wave", c_void_p), ("z_p", c_void_p), ("z", c_void_p)]


class TraceRecord(Structure):
    _fields_ = [("op", c_int32), ("ms", c_float), ("flops", c_double), ("bytes", c_double)]


_P, _I, _L, _F = c_void_p, c_int32, c_int64, c_float
SIGNATURES = {
    "svcmi_abi_version": (c_int, []),
    "svcmi_build_info": (c_char_p, []),
    "svcmi_conv_gemm_f32": (c_int, [POINTER(ConvDesc), _P]),
    "svcmi_pack_weights_lp": (c_int, [_P, _I, _I, _I, _P, _I, _P]),
    "svcmi_conv_gemm_lp": (c_int, [POINTER(ConvDesc), _I, _P]),
    "svcmi_conv_gemm_group_lp": (c_int, [_P, _I, _I, _P]),
    "svcmi_layernorm_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _I, _P]),
    "svcmi_channel_norm_gelu_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "svcmi_splitk_layernorm_f32": (c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _I, _I, _P]),
    "svcmi_attention_f32": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P, _I, _L, _I, _P]),
    "svcmi_attention16": (c_int, [_P, _P, _P, _I, _L, _P, _I, _L, _P, _I, _L, _I, _I, _I, _I, _F, _P, _P, _I, _P, _I, _P]),
    "svcmi_snake_alias_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "svcmi_snake_conv_supported": (c_int, [_I, _I, _I, _I]),
    "svcmi_snake_conv_preferred": (c_int, [_I, _I, _I, _I]),
    "svcmi_snake_conv_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "svcmi_upsample_noise_supported": (c_int, [_I, _I, _I]),
    "svcmi_upsample_noise_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I, _I, _I, _P]),
    "svcmi_tune_set": (c_int, [c_char_p, _I]),
    "svcmi_tune_get": (c_int, [c_char_p, POINTER(c_int32)]),
    "svcmi_wn_gate_f32": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "svcmi_wn_update_f32": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "svcmi_coupling_pre_f32": (c_int, [_P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P]),
    "svcmi_coupling_post_f32": (c_int, [_P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _P]),
    "svcmi_embed_pitch_f32": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "svcmi_sample_prior_f32": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "svcmi_ncl_to_nlc_f32": (c_int, [_P, _P, _F, _P, _I, _I, _I, _I, _P]),
    "svcmi_nlc_to_ncl_f32": (c_int, [_P, _I, _P, _I, _I, _I, _P]),
    "svcmi_pitch_prefix_f64": (c_int, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "svcmi_pitch_source_f32": (c_int, [_P, _P, _P, _P, _F, _P, _I, _I, _I, _F, _P]),
    "svcmi_reflect_pad_f32": (c_int, [_P, _P, _I, _L, _I, _P]),
    "svcmi_power_spectrum_f32": (c_int, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "svcmi_logmel_finish_f32": (c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "svcmi_crepe_frames_f32": (c_int, [_P, _L, _I, _I, _I, _P, _I, _P]),
    "svcmi_bn_maxpool2_f32": (c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _P, _I, _I, _P]),
    "svcmi_viterbi_decode": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "svcmi_row_sqnorm_f32": (c_int, [_P, _I, _L, _I, _P, _P]),
    "svcmi_knn_blend_f32": (c_int, [_P, _I, _P, _I, _P, _L, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "svcmi_ivf_assign_f32": (c_int, [_P, _I, _P, _L, _P, _I, _I, _I, _P, _P, _P]),
    "svcmi_ivf_blend_f32": (c_int, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "svcmi_segment_mean_f32": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "svcmi_snake_post_supported": (c_int, [_I, _I, _I]),
    "svcmi_snake_post_f32": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "svcmi_conv_gemm_group_f32": (c_int, [_P, _I, _P]),
    "svcmi_snake_conv_group_f32": (c_int, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "svcmi_snake_conv_lp_supported": (c_int, [_I, _I, _I, _I, _I]),
    "svcmi_snake_conv_group_lp": (c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "svcmi_snake_alias_group_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "svcmi_block_mean_f32": (c_int, [_P, _I, _P, _L, _P]),
    "svcmi_source2wav_i16": (c_int, [_P, _P, _L, _P]),
    "svcmi_whisper_workspace_bytes": (c_int64, [POINTER(WhisperModel), _I, _I]),
    "svcmi_whisper_encoder_fwd": (c_int, [POINTER(WhisperModel), _P, _P, _F, _I, _I, _P, _P, _L, _P]),
    "svcmi_synth_workspace_bytes": (c_int64, [POINTER(SynthModel), _I, _I, _I]),
    "svcmi_pitch2source_fwd": (c_int, [POINTER(SynthModel), _P, _P, _P, _I, _I, _P, _P, _L, _P]),
    "svcmi_text_encoder_fwd": (c_int, [POINTER(SynthModel), POINTER(SynthIO), _P, _P, _L, _P]),
    "svcmi_flow_reverse_fwd": (c_int, [POINTER(SynthModel), POINTER(SynthIO), _P, _P, _L, _P]),
    "svcmi_generator_fwd": (c_int, [POINTER(SynthModel), POINTER(SynthIO), _P, _P, _L, _P]),
    "svcmi_synth_infer_fwd": (c_int, [POINTER(SynthModel), POINTER(SynthIO), _P, _L, _P]),
    "svcmi_copy2d_f32": (c_int, [_P, _L, _P, _L, _L, _L, _P]),
    "svcmi_trace_begin": (c_int, [_I]),
    "svcmi_trace_end": (c_int, [POINTER(TraceRecord), _I]),
    "svcmi_trace_op_name": (c_char_p, [_I]),
    "svcmi_struct_sizes": (c_int, [POINTER(c_int64), _I]),
    "svcmi_packed_model_info": (c_int, [_P, _L, POINTER(c_int32), POINTER(c_int64), POINTER(c_int64)]),
    "svcmi_packed_model_bind": (c_int, [_P, _L, _P, _P, _L]),
}


class SvcmiError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen the HIP library and declare every symbol of include/svcmi.h.  Raises if it is absent."""
    path = path or os.environ.get("SVCMI_LIB", DEFAULT_LIB)
    if not os.path.exists(path):
        raise SvcmiError(
            f"{path} not found: the HIP kernels are not built. Run `python whisper-vits-svc_amd/build.py` "
            "(hipcc --offload-arch=gfx950). svcmi has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SvcmiError(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.svcmi_abi_version() != ABI_VERSION:
        raise SvcmiError(f"ABI mismatch: library {lib.svcmi_abi_version()} vs binding {ABI_VERSION}")
    import ctypes as _c
    sizes = (c_int64 * 8)()
    n = lib.svcmi_struct_sizes(sizes, 8)
    mine = [_c.sizeof(t) for t in (Weight, WhisperModel, SynthModel, SynthIO, TraceRecord, ConvDesc, SnakeConvDesc)]
    if list(sizes[:n]) != mine:
        raise SvcmiError(f"struct layout mismatch: library {list(sizes[:n])} vs binding {mine}")
    return lib
