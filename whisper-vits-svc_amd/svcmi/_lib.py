"""ctypes binding of libsvcmi.so (include/svcmi.h).  No fallback: a missing library raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libsvcmi.so")

ACT_NONE, ACT_RELU, ACT_GELU, ACT_MISH, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4, 5
CONV_ACCUMULATE, CONV_MASK_IN, CONV_MASK_OUT, CONV_PARTIALS = 1, 2, 4, 8
ABI_VERSION = 16
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16 = 0, 1, 2, 3
PRECISIONS = {None: 0, "f32": 0, "fp32": 0, "bf16x3": 1, "bf16": 2, "f16": 3, "fp16": 3}
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
    ]


_P, _I, _L, _F = c_void_p, c_int32, c_int64, c_float
SIGNATURES = {
    "svcmi_abi_version": (c_int, []),
    "svcmi_build_info": (c_char_p, []),
    "svcmi_conv_gemm_f32": (c_int, [POINTER(ConvDesc), _P]),
    "svcmi_pack_weights_lp": (c_int, [_P, _I, _I, _I, _P, _I, _P]),
    "svcmi_conv_gemm_lp": (c_int, [POINTER(ConvDesc), _I, _P]),
    "svcmi_conv_gemm_group_lp": (c_int, [_P, _I, _I, _P]),
    "svcmi_layernorm_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "svcmi_channel_norm_gelu_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "svcmi_splitk_layernorm_f32": (c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "svcmi_attention_f32": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P]),
    "svcmi_snake_alias_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "svcmi_snake_conv_supported": (c_int, [_I, _I, _I, _I]),
    "svcmi_snake_conv_preferred": (c_int, [_I, _I, _I, _I]),
    "svcmi_snake_conv_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "svcmi_upsample_noise_supported": (c_int, [_I, _I, _I]),
    "svcmi_upsample_noise_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _I, _I, _I, _P]),
    "svcmi_tune_set": (c_int, [c_char_p, _I]),
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
    "svcmi_bn_maxpool2_f32": (c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _P]),
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
    "svcmi_snake_alias_group_f32": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "svcmi_block_mean_f32": (c_int, [_P, _I, _P, _L, _P]),
    "svcmi_source2wav_i16": (c_int, [_P, _P, _L, _P]),
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
    return lib
