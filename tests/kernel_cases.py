"""Per-kernel parity checks shared by the emulator tests (CPU, not gpu) and the GPU tests.

Each check draws seeded inputs on the CPU, evaluates a plain torch fp32 / oracle reference there,
runs the kernel through ``ops`` (C ABI) on ``device`` and compares.  Tolerances are fp32 round-off
class (different summation orders), stated per check.
"""
import math
import zlib

import numpy as np
import torch

from svcmi._lib import PRECISIONS
import torch.nn.functional as F

from oracle import svc_oracle as O
from workload import weights as W
from svcmi import weights as PW
from svcmi.ops import ACT_GELU, ACT_MISH, ACT_NONE, ACT_RELU, ACT_TANH, SPLIT16


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _close(got, want, tol, what=""):
    got = got.detach().cpu()
    err = (got - want).abs().max().item()
    scale = max(1.0, want.abs().max().item())
    assert err <= tol * scale, f"{what}: max-abs err {err:.3e} (scale {scale:.2f}) > {tol:.1e}"


_ACT = {ACT_NONE: lambda v: v, ACT_RELU: torch.relu, ACT_GELU: F.gelu,
        ACT_MISH: lambda v: v * torch.tanh(F.softplus(v)), ACT_TANH: torch.tanh}

# id, B, T, Cin, N, K, stride, dil, pad, act, res, alpha, accumulate, lengths, mask_in, mask_out, repeat, tile
CONV_CASES_SMALL = [
    dict(id="linear_1x1", B=1, T=37, cin=16, n=10, k=1),
    dict(id="k5_dil2_relu_res", B=2, T=37, cin=8, n=10, k=5, dil=2, pad=4, act=ACT_RELU, res=True),
    dict(id="k3_stride2_gelu", B=1, T=41, cin=32, n=40, k=3, stride=2, pad=1, act=ACT_GELU),
    dict(id="k5_masks", B=2, T=29, cin=12, n=24, k=5, pad=2, lengths=[29, 17], mask_in=True, mask_out=True),
    dict(id="k3_chunk_tap_multi_kstep", B=1, T=70, cin=64, n=70, k=3, pad=1, act=ACT_MISH),
    dict(id="repeat2_fused", B=1, T=20, cin=16, n=8, k=5, pad=2, repeat=True),
    dict(id="cin1_scalar_path", B=2, T=64, cin=1, n=6, k=8, stride=4, pad=2),
    dict(id="accumulate_alpha", B=1, T=33, cin=8, n=8, k=7, pad=3, res=True, alpha=1.0 / 3.0, accumulate=True),
    dict(id="n1_tanh_nobias", B=1, T=50, cin=12, n=1, k=7, pad=3, act=ACT_TANH, bias=False),
    dict(id="splitk3_chunk", B=2, T=40, cin=64, n=70, k=5, pad=2, act=ACT_GELU, res=True, split_k=3),
    dict(id="splitk_auto_vec_masks", B=2, T=29, cin=12, n=24, k=25, pad=12, lengths=[29, 17], mask_in=True, mask_out=True, split_k=0),
    dict(id="splitk2_accumulate", B=1, T=33, cin=8, n=8, k=31, pad=15, res=True, alpha=1.0 / 3.0, accumulate=True, split_k=2),
    dict(id="vec_cin20_k11_d5", B=1, T=90, cin=20, n=20, k=11, dil=5, pad=25, res=True),
    dict(id="tile_128x64", B=1, T=150, cin=16, n=70, k=3, pad=1, tile=2),
    dict(id="tile_128x128", B=1, T=150, cin=16, n=140, k=1, tile=3),
    # 16x16x4 policy (tiles 4..8; auto-selected for 65 <= n_out <= 80 and t_out >= 1024)
    dict(id="p16_n40_vec_k11_d5_res", B=1, T=300, cin=40, n=40, k=11, dil=5, pad=25, res=True, tile=5),
    dict(id="p16_n80_vec_k3_acc_auto", B=2, T=1030, cin=80, n=80, k=3, pad=1, res=True, alpha=1.0 / 3.0, accumulate=True),
    dict(id="p16_n160_chunk_k7_d3_splitk", B=1, T=200, cin=160, n=160, k=7, dil=3, pad=9, res=True, split_k=2, tile=8),
    dict(id="p16_n38_ragged_masks", B=2, T=150, cin=16, n=38, k=5, pad=2, lengths=[150, 111], mask_in=True, mask_out=True, tile=4),
    dict(id="p16_n70_128x80", B=1, T=200, cin=32, n=70, k=3, pad=1, act=ACT_GELU, tile=7),
    # eight-wave blocks (tile 10 = SVCMI_CONV_TILE_P16W8_128x80): ragged rows / columns, several N tiles, residual + accumulate, split-K, masks
    dict(id="w8_n200_chunk_k3_gelu", B=1, T=300, cin=64, n=200, k=3, pad=1, act=ACT_GELU, tile=10),
    dict(id="w8_n70_vec_k5_d2_res_acc", B=2, T=150, cin=20, n=70, k=5, dil=2, pad=4, res=True, alpha=1.0 / 3.0, accumulate=True, tile=10),
    dict(id="w8_n84_splitk3_masks", B=2, T=140, cin=32, n=84, k=7, pad=3, lengths=[140, 101], mask_in=True, mask_out=True, split_k=3, tile=10),
    # outlier-scale inputs: the exact-erf GELU / Mish far in both tails (|v| up to ~100)
    dict(id="gelu_large_arguments", B=1, T=70, cin=64, n=70, k=3, pad=1, act=ACT_GELU, xscale=30.0),
    dict(id="mish_large_arguments", B=1, T=70, cin=32, n=40, k=7, pad=3, act=ACT_MISH, xscale=30.0),
]
# reduced-precision operand modes (svcmi_conv_gemm_lp): every tile policy x gather mode, split-K, the K tail, masks
CONV_CASES_LP_SMALL = [
    dict(id="lp_bf16x3_64x64_chunk_k3", B=1, T=70, cin=64, n=70, k=3, pad=1, act=ACT_GELU, prec="bf16x3", tile=1),
    dict(id="lp_bf16_64x64_vec_k5_d2_res", B=2, T=37, cin=8, n=10, k=5, dil=2, pad=4, act=ACT_RELU, res=True, prec="bf16"),
    dict(id="lp_f16_64x128_chunk", B=1, T=90, cin=32, n=150, k=1, prec="f16", tile=9),
    dict(id="lp_bf16x3_64x128_vec_masks", B=2, T=29, cin=12, n=136, k=5, pad=2, lengths=[29, 17], mask_in=True, mask_out=True, prec="bf16x3", tile=9),
    dict(id="lp_bf16x3_128x128_ktail", B=1, T=150, cin=20, n=140, k=3, pad=1, prec="bf16x3", tile=3),
    dict(id="lp_bf16x3_repeat2_fused", B=1, T=40, cin=32, n=24, k=5, pad=2, repeat=True, prec="bf16x3"),
    dict(id="lp_bf16x3_splitk3_acc", B=2, T=40, cin=64, n=70, k=5, pad=2, res=True, alpha=1.0 / 3.0, accumulate=True, split_k=3, prec="bf16x3"),
    dict(id="lp_f16_splitk_auto", B=1, T=33, cin=16, n=8, k=63, pad=31, split_k=0, prec="f16"),
    dict(id="lp_bf16x3_p16_n40_k11_d5", B=1, T=300, cin=40, n=40, k=11, dil=5, pad=25, res=True, prec="bf16x3", tile=4),
    dict(id="lp_bf16_p16_n80_k3_acc", B=2, T=130, cin=80, n=80, k=3, pad=1, res=True, alpha=1.0 / 3.0, accumulate=True, prec="bf16", tile=6),
    dict(id="lp_f16_p16_n38_masks", B=2, T=150, cin=16, n=38, k=5, pad=2, lengths=[150, 111], mask_in=True, mask_out=True, prec="f16", tile=4),
    dict(id="lp_bf16x3_stride2", B=1, T=81, cin=32, n=40, k=3, stride=2, pad=1, act=ACT_GELU, prec="bf16x3"),
    # 16-bit ACTIVATIONS (SVCMI_PREC_*_A16: x arrives as a bf16 / fp16 tensor, natural-order weight image) + the 16-bit output copy
    dict(id="a16_f16_64x64_chunk_k1_out16", B=2, T=70, cin=64, n=72, k=1, prec="f16", a16=True, out16=True, act=ACT_GELU),
    dict(id="a16_bf16_64x128_chunk_k3", B=1, T=90, cin=32, n=150, k=3, pad=1, prec="bf16", a16=True, tile=9, res=True),
    dict(id="a16_f16_vec_c40_k11_d5_p16", B=1, T=300, cin=40, n=40, k=11, dil=5, pad=25, res=True, prec="f16", a16=True, tile=4),
    dict(id="a16_bf16_vec_c24_masks_ktail", B=2, T=37, cin=24, n=20, k=5, pad=2, lengths=[37, 20], mask_in=True, mask_out=True, prec="bf16", a16=True),
    dict(id="a16_f16_128x128_splitk_partials", B=1, T=150, cin=256, n=140, k=1, prec="f16", a16=True, tile=3),
    dict(id="a16_f16_stride2_k3", B=1, T=81, cin=32, n=40, k=3, stride=2, pad=1, prec="f16", a16=True, out16=True),
    dict(id="a16_x3_64x64_chunk_k1_out16", B=2, T=70, cin=128, n=72, k=1, prec="bf16x3", a16=True, out16=True, act=ACT_GELU),
    dict(id="a16_x3_64x128_chunk_k3", B=1, T=90, cin=64, n=150, k=3, pad=1, prec="bf16x3", a16=True, tile=9, res=True),
    dict(id="a16_x3_vec_c40_k11_d5_p16_out16", B=1, T=300, cin=40, n=40, k=11, dil=5, pad=25, res=True, prec="bf16x3", a16=True, tile=4, out16=True),
    dict(id="a16_x3_vec_c24_masks_ktail", B=2, T=37, cin=24, n=20, k=5, pad=2, lengths=[37, 20], mask_in=True, mask_out=True, prec="bf16x3", a16=True),
    dict(id="a16_x3_128x128_k1", B=1, T=150, cin=256, n=140, k=1, prec="bf16x3", a16=True, tile=3),
    dict(id="a16_x3_stride2_k3_splitk", B=1, T=81, cin=96, n=40, k=3, stride=2, pad=1, prec="bf16x3", a16=True, split_k=3),
    # fp16 activations x split fp16 weights (SVCMI_PREC_F16W2_A16: rows [hi | lo], two MFMAs; lo is mostly subnormal fp16)
    dict(id="a16_w2_64x64_chunk_k3", B=1, T=90, cin=64, n=70, k=3, pad=1, prec="f16w2", a16=True, res=True),
    dict(id="a16_w2_p16_c40_k11_d5_out16", B=1, T=300, cin=40, n=40, k=11, dil=5, pad=25, res=True, prec="f16w2", a16=True, tile=4, out16=True),
    dict(id="a16_w2_p16_c80_k7_d3", B=2, T=130, cin=80, n=80, k=7, dil=3, pad=9, prec="f16w2", a16=True, tile=6),
    dict(id="a16_w2_vec_c24_masks_ktail", B=2, T=37, cin=24, n=20, k=5, pad=2, lengths=[37, 20], mask_in=True, mask_out=True, prec="f16w2", a16=True),
    dict(id="w2_without_x16_is_plain_f16", B=1, T=70, cin=32, n=40, k=3, pad=1, prec="f16w2"),
    # split_k = 0 (library heuristic) on long-K, few-tile shapes WITH the 16-bit output copy: the copy only comes out of the non-split
    # epilogue, so the launchers must not split (y16 was left uninitialised)
    dict(id="lp_f16_splitk_auto_out16", B=1, T=33, cin=16, n=8, k=63, pad=31, split_k=0, prec="f16", out16=True),
    dict(id="a16_bf16_splitk_auto_out16", B=1, T=40, cin=64, n=16, k=21, pad=10, split_k=0, prec="bf16", a16=True, out16=True),
]
CONV_CASES_LP_LARGE = [
    dict(id="lp_whisper_qkv_bf16x3", B=1, T=750, cin=1280, n=3840, k=1, prec="bf16x3"),
    dict(id="lp_whisper_mlp2_res_f16", B=1, T=500, cin=5120, n=1280, k=1, res=True, prec="f16", split_k=4),
    dict(id="lp_encp_pre_repeat_bf16x3", B=1, T=1000, cin=1280, n=192, k=5, pad=2, repeat=True, prec="bf16x3", split_k=0),
    dict(id="lp_amp_k11_d5_bf16", B=1, T=5000, cin=160, n=160, k=11, dil=5, pad=25, res=True, prec="bf16"),
    dict(id="lp_auto_p16_big_bf16x3", B=2, T=20000, cin=80, n=80, k=7, pad=3, prec="bf16x3"),
    dict(id="lp_b16_wn_in_bf16x3", B=16, T=1000, cin=192, n=384, k=5, pad=2, prec="bf16x3"),
    dict(id="a16_whisper_qkv_f16", B=1, T=750, cin=1280, n=3840, k=1, prec="f16", a16=True),
    dict(id="a16_whisper_mlp1_gelu_out16_bf16", B=2, T=500, cin=1280, n=5120, k=1, prec="bf16", a16=True, out16=True, act=ACT_GELU),
    dict(id="a16_whisper_mlp2_f16_splitk4", B=1, T=500, cin=5120, n=1280, k=1, res=True, prec="f16", a16=True, split_k=4),
    dict(id="a16_whisper_qkv_x3", B=1, T=750, cin=1280, n=3840, k=1, prec="bf16x3", a16=True),
    dict(id="a16_whisper_mlp1_gelu_out16_x3", B=2, T=500, cin=1280, n=5120, k=1, prec="bf16x3", a16=True, out16=True, act=ACT_GELU),
    dict(id="a16_whisper_mlp2_x3_splitk4", B=1, T=500, cin=5120, n=1280, k=1, res=True, prec="bf16x3", a16=True, split_k=4),
]
CONV_CASES_LARGE = [
    dict(id="whisper_qkv", B=1, T=500, cin=1280, n=3840, k=1),
    dict(id="whisper_mlp2_res", B=1, T=500, cin=5120, n=1280, k=1, res=True),
    dict(id="encp_pre_repeat", B=1, T=1000, cin=1280, n=192, k=5, pad=2, repeat=True),
    dict(id="amp_k11_d5", B=1, T=5000, cin=160, n=160, k=11, dil=5, pad=25, res=True),
    dict(id="stage4_c12_k11", B=1, T=40000, cin=12, n=12, k=11, pad=5, res=True),
    dict(id="auto_tile_big", B=2, T=20000, cin=80, n=80, k=7, pad=3),
]


def check_conv(ops, c, device):
    B, T, cin, n, k = c["B"], c["T"], c["cin"], c["n"], c["k"]
    stride, dil, pad = c.get("stride", 1), c.get("dil", 1), c.get("pad", 0)
    act = c.get("act", ACT_NONE)
    g = _g(zlib.crc32(c["id"].encode()) % 10000)
    rep = c.get("repeat", False)
    t_phys = T // 2 if rep else T
    x = torch.randn(B, t_phys, cin, generator=g) * c.get("xscale", 1.0)
    w = torch.randn(n, cin, k, generator=g) / math.sqrt(cin * k)
    bias = torch.randn(n, generator=g) if c.get("bias", True) else None
    xl = x.repeat_interleave(2, dim=1) if rep else x                      # logical input
    Tl = xl.shape[1]
    lengths = torch.tensor(c["lengths"], dtype=torch.int32) if "lengths" in c else None
    mask = (torch.arange(Tl)[None, :] < lengths[:, None]).float().unsqueeze(-1) if lengths is not None else None
    xin = xl * mask if c.get("mask_in") else xl
    prec = c.get("prec")
    conv = lambda a, ww, bb=None: F.conv1d(a.transpose(1, 2), ww, bb, stride=stride, dilation=dil, padding=pad).transpose(1, 2)
    exact = conv(xin, w, bias)
    if prec is None:
        ref = exact
    else:       # the same rounding the kernel applies to both operands, products and sums in fp32
        rnd = (lambda t: t.half().float()) if prec in ("f16", "f16w2") else (lambda t: t.bfloat16().float())
        xh, wh = rnd(xin), rnd(w)
        ref = conv(xh, wh, bias)
        if prec == "bf16x3":
            ref = ref + conv(rnd(xin - xh), wh) + conv(xh, rnd(w - wh))
        if prec == "f16w2" and c.get("a16"):          # a * hi + a * lo: the weight exact to 2^-22, the activation rounded (without x16: plain fp16)
            ref = ref + conv(xh, rnd(w - wh))
        # and the mode's own error against the fp32 result stays in its class
        bound = {"bf16x3": 2e-5, "f16": 2e-3, "bf16": 1.5e-2, "f16w2": 2e-3}[prec]
        err = (ref - exact).abs().max().item() / max(1.0, exact.abs().max().item())
        assert err <= bound, f"{c['id']}: {prec} rounding error {err:.2e} > {bound:.0e}"
    t_out = ref.shape[1]
    ref = _ACT[act](ref)
    res = torch.randn(B, t_out, n, generator=g) if c.get("res") else None
    if res is not None:
        ref = ref + res
    ref = ref * c.get("alpha", 1.0)
    y0 = torch.randn(B, t_out, n, generator=g) if c.get("accumulate") else None
    if y0 is not None:
        ref = ref + y0
    if c.get("mask_out"):
        ref = ref * mask[:, :t_out]
    wp = PW.pack_conv(w).to(device)
    dev = lambda t: None if t is None else t.to(device)
    out = dev(y0.clone()) if y0 is not None else None
    xd = dev(x)
    if cin == 1:
        xd = xd.reshape(B, t_phys)      # a plain signal, ldx = 1
    launches, saved_min = ops.launches, ops.lp_min_flops
    ops.lp_min_flops = 0.0
    dt16 = {"f16": torch.float16, "bf16": torch.bfloat16, "f16w2": torch.float16}.get(prec)
    x16 = xd.to(dt16) if c.get("a16") and dt16 else None          # what a producing kernel's 16-bit output holds: the same values, rounded
    if c.get("a16") and prec == "bf16x3":                # ... or split into (hi, lo) bf16 planes: rows [hi: Cp | lo: Cp]
        x16 = _split16(xd)
        dt16 = SPLIT16
    with ops.use_precision(prec):
        y = ops.conv(xd, wp, dev(bias), ksize=k, stride=stride, dilation=dil, pad=pad, act=act, res=dev(res),
                     alpha=c.get("alpha", 1.0), accumulate=c.get("accumulate", False), lengths=dev(lengths),
                     mask_in=c.get("mask_in", False), mask_out=c.get("mask_out", False), out=out,
                     x_row_shift=1 if rep else 0, c_in=cin, ldx=cin, n_out=n, tile=c.get("tile", 0), split_k=c.get("split_k", 1),
                     t_in=Tl, x_bstride=t_phys * cin, x16=x16, out16=dt16 if c.get("out16") else None)
    ops.lp_min_flops = saved_min
    if prec is not None:
        assert getattr(wp, "_svcmi_lp", None), f"{c['id']}: the reduced-precision kernel did not run"
        if c.get("a16"):
            a16_code = {"bf16x3": 6, "f16w2": 9}.get(prec, PRECISIONS[prec] + 2)
            assert wp._svcmi_lp.get(a16_code) is not None, f"{c['id']}: the 16-bit-activation kernel did not run"
    if c.get("out16"):
        y, y16 = y
        want16 = _split16(y.cpu()) if dt16 == SPLIT16 else y.cpu().to(dt16)
        assert torch.equal(y16.cpu(), want16), f"{c['id']}: the 16-bit output copy is not the rounded fp32 output"
    assert y.shape == ref.shape
    _close(y, ref, 2e-5 if cin * k < 4096 else 1e-4, c["id"])



def check_conv_ring2(ops, device, tile, n, cin=64, k=5, T=150, B=2):
    """SVCMI_CONV_RING2 (2-deep operand ring of the 64-row fp32 tiles, for launches that share the chip): the same bits as the default
    3-deep ring on every tile that has the instantiation (64x64, P16 64x48, P16 64x80), K ranges of 1 / 2 / many K-steps included;
    ignored (not an error) where there is none."""
    g = _g(77 + tile + n)
    x = (torch.randn(B, T, cin, generator=g)).to(device)
    w = PW.pack_conv(torch.randn(n, cin, k, generator=g) / math.sqrt(cin * k)).to(device)
    bias = torch.randn(n, generator=g).to(device)
    res = torch.randn(B, T, n, generator=g).to(device)
    kw = dict(ksize=k, pad=(k - 1) // 2, act=ACT_GELU, res=res, split_k=1)
    import ctypes
    ring = ctypes.c_int32(0)
    y3 = ops.conv(x, w, bias, tile=tile, **kw)
    assert ops.lib.svcmi_tune_get(b"last_conv_ring", ctypes.byref(ring)) == 0 and ring.value == 3
    y2 = ops.conv(x, w, bias, tile=tile | 16, **kw)          # (tile << 8) | 0x1000 = SVCMI_CONV_RING2
    # the flag is IGNORED where no 2-deep instantiation exists (128-row four-wave tiles): a regression that stops dispatching it must not pass silently
    assert ops.lib.svcmi_tune_get(b"last_conv_ring", ctypes.byref(ring)) == 0 and ring.value == (2 if tile in (1, 4, 6, 10) else 3), (tile, ring.value)
    assert torch.equal(y2, y3), (tile, n, cin, k, float((y2 - y3).abs().max()))
    ref = F.gelu(F.conv1d(x.cpu().transpose(1, 2), w.cpu()[:, :cin * k].view(n, k, cin).permute(0, 2, 1).contiguous(), bias.cpu(), padding=(k - 1) // 2).transpose(1, 2)) + res.cpu()
    _close(y2, ref, 2e-5, f"ring2 tile {tile}")
    return True

def check_conv_w8(ops, device, n, cin=64, k=3, T=300, B=2, partials=False):
    """SVCMI_CONV_TILE_P16W8_128x80 (tile 10: the 64x80 wave tile on eight-wave blocks) against SVCMI_CONV_TILE_P16_64x80 (tile 6): the same
    wave tile and the same K order, so the same bits -- full launches (bias, GELU, residual) and raw split-K slabs, both ring depths."""
    g = _g(1000 + n + cin + k)
    x = (torch.randn(B, T, cin, generator=g)).to(device)
    w = PW.pack_conv(torch.randn(n, cin, k, generator=g) / math.sqrt(cin * k)).to(device)
    bias = torch.randn(n, generator=g).to(device)
    res = torch.randn(B, T, n, generator=g).to(device)
    for ring in (0, 16):
        if partials:
            y6 = ops.conv(x, w, None, ksize=k, pad=(k - 1) // 2, tile=6 | ring, split_k=2, partials=True)
            y10 = ops.conv(x, w, None, ksize=k, pad=(k - 1) // 2, tile=10 | ring, split_k=2, partials=True)
        else:
            y6 = ops.conv(x, w, bias, ksize=k, pad=(k - 1) // 2, act=ACT_GELU, res=res, split_k=1, tile=6 | ring)
            y10 = ops.conv(x, w, bias, ksize=k, pad=(k - 1) // 2, act=ACT_GELU, res=res, split_k=1, tile=10 | ring)
        assert torch.equal(y6, y10), (n, cin, k, ring, float((y6 - y10).abs().max()))
    return True


def _split16(x):
    """fp32 [..., C] -> the SPLIT16 layout [..., 2*Cp] (ops.SPLIT16): hi = bf16(x), lo = bf16(x - hi), zero pads."""
    c = x.shape[-1]
    cp = (c + 7) // 8 * 8
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    out = torch.zeros(*x.shape[:-1], 2 * cp, dtype=torch.bfloat16, device=x.device)
    out[..., :c], out[..., cp:cp + c] = hi, lo
    return out


def check_layernorm(ops, c, device):
    g = _g(c)
    B, T = 2, 9
    x, r = torch.randn(B, T, c, generator=g) * 2 + 0.5, torch.randn(B, T, c, generator=g)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    _close(ops.layernorm(x.to(device), gamma.to(device), beta.to(device), res=r.to(device)),
           F.layer_norm(x + r, (c,), gamma, beta, 1e-5), 2e-5, "ln+res")
    _close(ops.layernorm(x.to(device)), F.layer_norm(x, (c,), None, None, 1e-5), 2e-5, "ln plain")
    gb, bb = torch.randn(B, c, generator=g), torch.randn(B, c, generator=g)
    want = F.layer_norm(x, (c,), None, None, 1e-5) * gb[:, None] + bb[:, None]
    _close(ops.layernorm(x.to(device), gb.to(device), bb.to(device), per_batch_affine=True), want, 2e-5, "ln per-batch")


def check_viterbi(ops, device, frames=70, batch_frames=32, jumps=False):
    """Device Viterbi (fp32 softmax, fp64 DP, per decoding batch) vs the host restatement of librosa.sequence.viterbi."""
    from svcmi.pitch import inference as PI
    g = _g(31 + frames)
    centre = 120 + 40 * torch.sin(torch.arange(frames) / 7.0)
    if jumps:          # octave-like leaps far outside the 11-bin band, with sharp posteriors: the out-of-band predecessor wins
        centre = centre + 150.0 * ((torch.arange(frames) // 9) % 2)
    sharp = 40.0 if jumps else 1.0
    prob = torch.sigmoid(sharp * -((torch.arange(360)[None, :] - centre[:, None]) / 6.0) ** 2 + 0.8 * torch.randn(frames, 360, generator=g))
    lo, hi = PI._frequency_to_bins(50.0), PI._frequency_to_bins(1000.0, ceil=True)
    # with CREPE's own matrix a leap costs log(tiny) = -87 nats against <= 1 nat of evidence per frame and never pays; the jump case
    # uses a milder floor so that the out-of-band predecessor is actually taken
    lt = torch.from_numpy(np.log(PI._transition() + (0.5 if jumps else np.finfo(np.float32).tiny)))
    band = PI.transition_band(PI._transition())
    assert band == 11
    dense = ops.viterbi_decode(prob.to(device), lt.to(device), batch_frames, lo, hi).cpu().numpy()
    got = ops.viterbi_decode(prob.to(device), lt.to(device), batch_frames, lo, hi, band=band).cpu().numpy()
    assert np.array_equal(got, dense)                      # the banded DP is the dense one, bit for bit
    if jumps:
        assert np.abs(np.diff(got)).max() > 100            # the path really leaves the band
        return
    want = []
    for i in range(0, frames, batch_frames):
        p = prob[i:i + batch_frames].t().clone()
        p[:lo] = -float("inf")
        p[hi:] = -float("inf")
        want.append(PI.viterbi_path(torch.softmax(p, dim=0).numpy(), PI._transition()))
    want = np.concatenate(want)
    assert got.shape == want.shape and (got == want).mean() >= 0.99, float((got == want).mean())


def check_knn_blend(ops, device, t=37, n=301, d=64, k=3, ratio=0.5):
    """GEMM scores + knn_blend through the host index class vs the exhaustive-search restatement of faiss + the RVC
    weighting (oracle/retrieval_oracle.py).  fp32 round-off only: the winners' distances are re-measured exactly."""
    from oracle import retrieval_oracle as RO
    from svcmi.feature_retrieval import KnnFeatureIndex
    g = _g(97 + t + n + k)
    centres = torch.randn(8, d, generator=g) * 2.0
    bank = (centres[torch.randint(0, 8, (n,), generator=g)] + torch.randn(n, d, generator=g)).numpy()
    feats = (centres[torch.randint(0, 8, (t,), generator=g)] + torch.randn(t, d, generator=g)).numpy()
    index = KnnFeatureIndex(bank, ratio, k, device=device, ops=ops)
    if 128 < n < 10000:
        index._bank_block = 128          # exercise the multi-launch score GEMM over bank blocks
    _close(index.bank_sq, torch.from_numpy((bank.astype(np.float64) ** 2).sum(1)).float(), 2e-6, "row_sqnorm")
    want = torch.from_numpy(RO.retriv(feats, bank, ratio, k))
    got = index.retriv(feats)
    assert isinstance(got, np.ndarray) and got.dtype == np.float32
    _close(torch.from_numpy(got), want, 2e-5, f"knn_blend t={t} n={n} d={d} k={k}")
    got_t = index.retriv(torch.from_numpy(feats).to(device))      # tensors stay on their device
    assert got_t.device.type == torch.device(device).type
    _close(got_t, want, 2e-5, "knn_blend tensor")


def _blobs(g, n, d, centres=8, spread=1.0):
    c = torch.randn(centres, d, generator=g) * 4.0
    return (c[torch.randint(0, centres, (n,), generator=g)] + spread * torch.randn(n, d, generator=g)).numpy()


def check_ivf_index(ops, device, t=37, n=600, d=32, k=3, ratio=0.5, nlist=9, tmp_path=None):
    """The reference's index type (IVF-Flat, nprobe = 1): coarse assignment + list scan + RVC blend through the host index class vs
    the numpy restatement of faiss's algorithm (oracle/retrieval_oracle.py ivf_*), on an index whose lists are deliberately ragged
    (one empty cell, one with fewer than k vectors)."""
    from oracle import retrieval_oracle as RO
    from svcmi.ivf_index import IvfFlatFeatureIndex
    g = _g(7 + t + n + k)
    bank = _blobs(g, n, d)
    feats = _blobs(g, t, d)
    cent = bank[torch.randperm(n, generator=g)[:nlist].numpy()].copy()
    cent[nlist - 1] = 1.0e3                                     # a cell nothing falls into
    cell, _ = RO.coarse_assign(bank, cent)
    thin = int(np.bincount(cell, minlength=nlist)[:nlist - 1].argmin())
    keep = np.ones(n, bool)
    keep[np.flatnonzero(cell == thin)[max(k - 1, 1):]] = False   # leave k - 1 vectors in the thinnest cell
    bank, cell = bank[keep], cell[keep]
    ids = np.arange(len(bank), dtype=np.int64) * 3 + 1           # labels are not row numbers
    lists = [(bank[cell == c], ids[cell == c]) for c in range(nlist)]
    off = np.zeros(nlist + 1, np.int32)
    off[1:] = np.cumsum([len(i) for _, i in lists])
    index = IvfFlatFeatureIndex(cent, off, np.concatenate([v for v, _ in lists]), np.concatenate([i for _, i in lists]),
                                ratio, k, device=device, ops=ops)
    feats[0] = cent[thin] + 1e-3                                  # make sure the thin cell is probed
    want_cell, _ = RO.coarse_assign(feats, cent)
    dist_w, lab_w, rec_w = RO.ivf_search(feats, cent, lists, k)
    dist, lab, rec = index.search_and_reconstruct(feats, k)
    assert np.array_equal(lab, lab_w), "ivf labels"
    fin = np.isfinite(dist_w)
    assert np.array_equal(np.isfinite(dist), fin) and np.allclose(dist[fin], dist_w[fin], rtol=2e-5), "ivf distances"
    assert np.array_equal(np.isnan(rec), np.isnan(rec_w)) and np.array_equal(rec[~np.isnan(rec)], rec_w[~np.isnan(rec_w)]), "ivf vectors"
    assert (lab_w[0] >= 0).sum() == min(len(lists[thin][1]), k) <= max(k - 1, 1), "the thin cell was not probed"
    got = index.retriv(feats)
    _close(torch.from_numpy(got), torch.from_numpy(RO.ivf_retriv(feats, cent, lists, ratio, k)), 2e-5, f"ivf_blend t={t} n={n} d={d} k={k}")
    got_t = index.retriv(torch.from_numpy(feats).to(device))
    assert got_t.device.type == torch.device(device).type and np.array_equal(got_t.cpu().numpy(), got)
    if tmp_path is not None:                                      # faiss file layout: save -> load -> same answers
        f = tmp_path / "t.index"
        index.save(f)
        again = IvfFlatFeatureIndex.from_faiss(f, ratio, k, device=device, ops=ops)
        assert np.array_equal(again.retriv(feats), got) and again.ntotal == len(bank) and again.nlist == nlist
    return want_cell


def check_ivf_train(ops, device, n=900, d=16, blobs=6, n_ivf=None, exact=True):
    """faiss's k-means + add (svcmi.ivf_index.train_kmeans / IvfFlatFeatureIndex.train) vs the numpy restatement: same permutations
    (std::mt19937 replay); ONE Lloyd step from the same start must agree (assignment flips only at fp32 ties); the full 25
    iterations agree centroid by centroid when the clusters are separated (``exact``: n_ivf <= blobs), and otherwise -- k-means
    amplifies a single flipped boundary point -- in the quantisation error they reach."""
    from oracle import retrieval_oracle as RO
    from svcmi.ivf_index import IvfFlatFeatureIndex, faiss_rand_perm, ivf_list_count, train_kmeans
    g = _g(31 + n + d)
    x = _blobs(g, n, d, centres=blobs, spread=0.5)
    k = n_ivf or ivf_list_count(n)
    assert np.array_equal(faiss_rand_perm(50, 1235), RO.rand_perm(50, 1235))
    xd = torch.from_numpy(x).to(device)
    one = train_kmeans(xd, k, ops, niter=1).cpu().numpy()
    one_w = RO.kmeans_faiss(x, k, niter=1)
    close = np.isclose(one, one_w, rtol=1e-4, atol=1e-4).all(1)
    # |x|^2 + |c|^2 - 2 x.c carries ~1e-7 * |x|^2 of round-off (faiss's own BLAS path does too): at d = 1280 a few of 20 000 boundary
    # points change cells between two summation orders, each moving two centroids by (x - c) / count
    assert close.mean() >= 0.95, f"one Lloyd step: {int((~close).sum())} of {k} centroids differ"
    assert np.abs(one - one_w).max() <= 0.05 * np.abs(one_w).max(), float(np.abs(one - one_w).max())
    index = IvfFlatFeatureIndex.train(x, device=device, ops=ops, n_ivf=n_ivf)
    cent_w, lists_w = RO.ivf_build(x, n_ivf=n_ivf)
    assert index.nlist == k == len(lists_w)
    off = index.list_off.cpu().numpy()
    ids = index.ids.cpu().numpy()
    assert index.ntotal == n and np.array_equal(np.sort(ids), np.arange(n))
    assert np.array_equal(index.bank.cpu().numpy(), x[ids])
    assert all(np.all(np.diff(ids[off[c]:off[c + 1]]) > 0) for c in range(k)), "rows keep their insertion order inside a cell"
    if exact:
        _close(index.centroids, torch.from_numpy(cent_w), 2e-5, "k-means centroids")
        same = sum(np.array_equal(ids[off[c]:off[c + 1]], lists_w[c][1]) for c in range(k))
        assert same >= k - 1, f"{k - same} inverted lists differ"          # a boundary point may flip on round-off
    else:
        cent = index.centroids.cpu().numpy()
        err = lambda c: float(RO.coarse_assign(x, c)[1].min(1).mean())
        e, e_w = err(cent), err(cent_w)
        assert abs(e - e_w) <= 0.02 * e_w, f"quantisation error {e:.4f} vs restatement {e_w:.4f}"
    return index


def check_channel_norm_gelu(ops, device, B=2, T=700, c=32):
    g = _g(5 + c)
    x = torch.randn(B, T, c, generator=g) * 2.0 + 0.7
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    want = F.gelu(F.group_norm(x.transpose(1, 2), c, gamma, beta, 1e-5)).transpose(1, 2)
    got = ops.channel_norm_gelu(x.to(device), gamma.to(device), beta.to(device))
    _close(got, want, 2e-5, "channel_norm_gelu")


def check_splitk_layernorm(ops, device, B=2, S=3, T=7, c=1280):
    """conv(partials=True) slabs -> fused reduce + bias + residual + LayerNorm, vs the unsplit conv + torch layer_norm."""
    g = _g(100 + c + S)
    cin = 64 * S
    a = torch.randn(B, T, cin, generator=g)
    w = torch.randn(c, cin, 1, generator=g) / math.sqrt(cin)
    bias, x = torch.randn(c, generator=g), torch.randn(B, T, c, generator=g)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    x_ref = x + F.conv1d(a.transpose(1, 2), w, bias).transpose(1, 2)
    y_ref = F.layer_norm(x_ref, (c,), gamma, beta, 1e-5)
    dev = lambda t: t.to(device)
    xd = dev(x.clone())
    part = ops.conv(dev(a), PW.pack_conv(w).to(device), None, partials=True, split_k=S)
    assert tuple(part.shape) == (B, S, T, c)
    y = ops.splitk_layernorm(part, dev(bias), xd, dev(gamma), dev(beta))
    _close(xd, x_ref, 2e-5, "splitk_ln residual stream")
    _close(y, y_ref, 5e-5, "splitk_ln output")


ATTN_CASES_SMALL = [
    dict(id="d16_rel_w4_ragged", B=2, T=45, H=2, D=16, rel=True, W=4, lengths=[45, 31]),
    dict(id="d32_plain", B=1, T=70, H=3, D=32),
    dict(id="d64_plain_T130", B=1, T=130, H=2, D=64),
    dict(id="d96_rel_short_T3", B=1, T=3, H=2, D=96, rel=True, W=4),
    dict(id="d96_rel_T67", B=1, T=67, H=2, D=96, rel=True, W=4, lengths=[60]),
    dict(id="d32_rel_w2_T140_keysplit2", B=1, T=140, H=2, D=32, rel=True, W=2, lengths=[133]),
    dict(id="d96_rel_T260_keysplit4", B=2, T=260, H=2, D=96, rel=True, W=4, lengths=[260, 201]),
]
# band-free cases forced through the two-query-tile kernel (the heuristic only picks it for >= 512 (head, q-tile) blocks)
ATTN_CASES_Q32 = [
    dict(id="q32_d32_plain_T70", B=1, T=70, H=3, D=32, q32=True),
    dict(id="q32_d64_T130_ns2", B=1, T=130, H=2, D=64, q32=True, ns=2),
    dict(id="q32_d64_ragged_T77", B=2, T=77, H=2, D=64, lengths=[77, 40], q32=True),
    dict(id="q32_d16_T33_ns1", B=1, T=33, H=1, D=16, q32=True, ns=1),
    dict(id="q32_d64_T300_ns4", B=1, T=300, H=2, D=64, q32=True, ns=4),
]
# The LDS-staged kernel: K / V tiles of a head staged once per block and shared by QT query tiles x KS key ranges ("attn_lds" knob =
# 10 * QT + KS; 1 = the 4 x 2 / 4 x 1 default shape).  Every compiled shape, ragged lengths, ranges that lie entirely past T.
ATTN_CASES_LDS = [
    dict(id="lds_d64_T130_one_range", B=1, T=130, H=2, D=64, lds=1),
    dict(id="lds_d64_T300_two_ranges", B=1, T=300, H=2, D=64, lds=1),
    dict(id="lds_d64_ragged_T257", B=2, T=257, H=2, D=64, lengths=[257, 140], lds=1),
    dict(id="lds_d32_T70", B=1, T=70, H=3, D=32, lds=1),
    dict(id="lds_d16_T33", B=1, T=33, H=1, D=16, lds=1),
    dict(id="lds_d64_T288_short_second_range", B=1, T=288, H=1, D=64, lds=1),
    dict(id="lds21_d64_T77", B=1, T=77, H=2, D=64, lds=21),
    dict(id="lds22_d64_ragged_T131", B=2, T=131, H=2, D=64, lengths=[131, 64], lds=22),
    dict(id="lds24_d64_T200", B=1, T=200, H=2, D=64, lds=24),
    dict(id="lds24_d32_T40_empty_ranges", B=1, T=40, H=1, D=32, lds=24),
    dict(id="lds44_d64_T300", B=1, T=300, H=1, D=64, lds=44),
    dict(id="lds81_d64_T150", B=1, T=150, H=2, D=64, lds=81),
    dict(id="lds82_d64_ragged_T260", B=2, T=260, H=1, D=64, lengths=[200, 260], lds=82),
    # relative-position band through the LDS-staged kernel (round 4; head widths 32 / 96)
    dict(id="lds41_rel_d32_T70", B=1, T=70, H=2, D=32, rel=True, W=4, lds=41),
    dict(id="lds42_rel_d32_ragged_T150", B=2, T=150, H=2, D=32, rel=True, W=4, lengths=[150, 61], lds=42),
    dict(id="lds81_rel_d32_T200_w2", B=1, T=200, H=1, D=32, rel=True, W=2, lds=81),
    dict(id="lds82_rel_d96_T140", B=1, T=140, H=2, D=96, rel=True, W=4, lds=82),
    # the key-split kernel forced (no LDS staging) at every split of the head width Whisper uses
    dict(id="ks_d64_T130_ns1", B=1, T=130, H=2, D=64, lds=-1, ns=1),
    dict(id="ks_d64_ragged_T257_ns2", B=2, T=257, H=2, D=64, lengths=[257, 140], lds=-1, ns=2),
    dict(id="ks_d64_T300_ns4_short_last_range", B=1, T=300, H=1, D=64, lds=-1, ns=4),
    dict(id="ks_d64_T33_ns8_empty_ranges", B=1, T=33, H=1, D=64, lds=-1, ns=8),
]
# Round 6: the 64-key-step kernel (attention_wide_kernel: D = 64 band-free, D = 96 with the band; "attn_wide" knob 1 = forced, 0 = the
# 32-key-step kernel it replaced -- which short sequences still take, so both stay covered at the same shapes).
ATTN_CASES_WIDE = [
    dict(id="wide_d64_T33_ns8_empty_ranges", B=1, T=33, H=1, D=64, lds=-1, ns=8, wide=1),
    dict(id="wide_d64_ragged_T200_ns2", B=2, T=200, H=2, D=64, lengths=[200, 77], lds=-1, ns=2, wide=1),
    dict(id="wide_d64_T64_exact_step", B=1, T=64, H=1, D=64, lds=-1, ns=1, wide=1),
    dict(id="wide_d64_T65_one_key_tail", B=1, T=65, H=1, D=64, lds=-1, ns=1, wide=1),
    dict(id="wide_d96_rel_T140_ns2", B=1, T=140, H=2, D=96, rel=True, W=4, lds=-1, ns=2, wide=1),
    dict(id="wide_d96_rel_w2_T100_ragged", B=2, T=100, H=2, D=96, rel=True, W=2, lengths=[90, 100], lds=-1, ns=1, wide=1),
    dict(id="wide_d96_rel_T3", B=1, T=3, H=2, D=96, rel=True, W=4, lds=-1, ns=1, wide=1),
    dict(id="narrow_d64_T300_ns4", B=1, T=300, H=1, D=64, lds=-1, ns=4, wide=0),
    dict(id="narrow_d96_rel_T260_ns2", B=2, T=260, H=2, D=96, rel=True, W=4, lengths=[260, 201], lds=-1, ns=2, wide=0),
]
# the same kernel at the sizes it is selected for (GPU only)
ATTN_CASES_LDS_LARGE = [
    dict(id="lds42_whisper_T500", B=1, T=500, H=20, D=64, lds=42),
    dict(id="lds42_whisper_T750_B2", B=2, T=750, H=20, D=64, lds=42),
    dict(id="lds81_whisper_T500_B4", B=4, T=500, H=20, D=64, lds=81),
    dict(id="lds24_whisper_T500", B=1, T=500, H=20, D=64, lds=24),
    dict(id="lds82_T1500", B=1, T=1500, H=4, D=64, lds=82),
    dict(id="lds81_rel_encp_T1000_B4", B=4, T=1000, H=2, D=96, rel=True, W=4, lds=81),
    dict(id="lds82_rel_encp_ragged_B3", B=3, T=301, H=2, D=96, rel=True, W=4, lengths=[301, 250, 7], lds=82),
    dict(id="auto_rel_encp_T1000_B16", B=16, T=1000, H=2, D=96, rel=True, W=4),          # the heuristic picks the LDS-staged kernel (256 blocks)
]

ATTN_CASES_LARGE = [
    dict(id="whisper_T500", B=1, T=500, H=20, D=64),
    dict(id="whisper_T500_narrow", B=1, T=500, H=20, D=64, wide=0),
    dict(id="whisper_T500_wide_ns8", B=1, T=500, H=20, D=64, lds=-1, ns=8, wide=1),
    dict(id="encp_T1000_narrow", B=1, T=1000, H=2, D=96, rel=True, W=4, wide=0),
    dict(id="encp_T1000", B=1, T=1000, H=2, D=96, rel=True, W=4),
    dict(id="encp_ragged_B3", B=3, T=301, H=2, D=96, rel=True, W=4, lengths=[301, 250, 7]),
]


def attention_reference(qkv, H, scale, rel_k, rel_v, W, lengths):
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    D = Cc // H
    q, k, v = [t.view(B, T, H, D).permute(0, 2, 1, 3).double() for t in qkv.split(Cc, dim=-1)]
    s = q @ k.transpose(-1, -2)
    idx = torch.arange(T)
    rel = idx[None, :] - idx[:, None]
    if rel_k is not None:
        band = rel.abs() <= W
        relc = (rel + W).clamp(0, 2 * W)
        qe = q @ rel_k.double().t()
        s = s + torch.where(band, qe.gather(-1, relc.expand(B, H, T, T)), torch.zeros((), dtype=torch.double))
    s = s * scale
    if lengths is not None:
        m = (idx[None, :] < lengths[:, None])
        mm = (m[:, :, None] & m[:, None, :])[:, None]
        s = s.masked_fill(~mm, -1e4)
    p = torch.softmax(s, dim=-1)
    o = p @ v
    if rel_k is not None:
        pb = torch.where(band, p, torch.zeros((), dtype=torch.double))
        rw = torch.zeros(B, H, T, 2 * W + 1, dtype=torch.double)
        rw.scatter_add_(-1, relc.expand(B, H, T, T), pb)
        o = o + rw @ rel_v.double()
    return o.permute(0, 2, 1, 3).reshape(B, T, Cc).float()


def check_attention(ops, c, device):
    g = _g(7 + c["T"])
    B, T, H, D = c["B"], c["T"], c["H"], c["D"]
    qkv = torch.randn(B, T, 3 * H * D, generator=g)
    rel_k = torch.randn(2 * c["W"] + 1, D, generator=g) * D ** -0.5 if c.get("rel") else None
    rel_v = torch.randn(2 * c["W"] + 1, D, generator=g) * D ** -0.5 if c.get("rel") else None
    lengths = torch.tensor(c["lengths"], dtype=torch.int32) if "lengths" in c else None
    scale = D ** -0.5
    want = attention_reference(qkv, H, scale, rel_k, rel_v, c.get("W", 0), lengths)
    dev = lambda t: None if t is None else t.to(device)
    if c.get("q32"):
        assert ops.lib.svcmi_tune_set(b"attn_q32", 1) == 0 and ops.lib.svcmi_tune_set(b"attn_ns", c.get("ns", 0)) == 0
    if c.get("lds"):        # the LDS-staged kernel, forced to one of its compiled shapes
        assert ops.lib.svcmi_tune_set(b"attn_lds", int(c["lds"])) == 0
    if c.get("lds") == -1:
        assert ops.lib.svcmi_tune_set(b"attn_ns", c.get("ns", 0)) == 0
    if "wide" in c:
        assert ops.lib.svcmi_tune_set(b"attn_wide", int(c["wide"])) == 0
    try:
        got = ops.attention(dev(qkv), H, scale, rel_k=dev(rel_k), rel_v=dev(rel_v), window=c.get("W", 0), lengths=dev(lengths))
    finally:
        ops.lib.svcmi_tune_set(b"attn_q32", -1)
        ops.lib.svcmi_tune_set(b"attn_ns", 0)
        ops.lib.svcmi_tune_set(b"attn_lds", 0)
        ops.lib.svcmi_tune_set(b"attn_wide", -1)
    _close(got, want, 2e-5, c["id"])


def check_snake(ops, n, c, device, amp=1.5, alpha_mean=0.0):
    """``amp`` / ``alpha_mean`` >> the defaults: outlier-scale activations and SnakeBeta frequencies (sin^2 arguments of
    hundreds to thousands of radians -- the Cody-Waite range reduction, not the polynomial core, csrc/snake_math.h)."""
    g = _g(n * 100 + c)
    B = 2
    x = torch.randn(B, n, c, generator=g) * amp
    al, be = torch.randn(c, generator=g) * 0.3 + alpha_mean, torch.randn(c, generator=g) * 0.3
    filt = W.kaiser_sinc_filter().view(-1)
    want = O.snake_alias(x.transpose(1, 2), al, be, filt).transpose(1, 2)
    got = ops.snake_alias(x.to(device), al.to(device), be.to(device), filt.to(device))
    _close(got, want, 1e-5, f"snake n={n} c={c}")


UPNOISE_CASES = [
    dict(id="stage4_20to10", cin=20, cout=10, k=4, u=2, p=1, nz_k=1, nz_s=1, nz_p=0, T=37, B=2),
    dict(id="stage3_40to20", cin=40, cout=20, k=4, u=2, p=1, nz_k=4, nz_s=2, nz_p=1, T=50, B=1),
]


def check_upsample_noise(ops, c_, device):
    """ups[i] (ConvTranspose1d) + noise_convs[i] (strided Conv1d on the source) in one launch vs torch."""
    g = _g(zlib.crc32(c_["id"].encode()) % 10000)
    cin, cout, k, u, pp, T, B = c_["cin"], c_["cout"], c_["k"], c_["u"], c_["p"], c_["T"], c_["B"]
    cp = (cout + 3) // 4 * 4
    x = torch.randn(B, T, cin, generator=g)
    w = torch.randn(cin, cout, k, generator=g) / math.sqrt(cin * k / u)
    bias = torch.randn(cout, generator=g)
    L = T * u * c_["nz_s"]
    src = torch.randn(B, L, generator=g)
    nw = torch.randn(cout, 1, c_["nz_k"], generator=g)
    nb = torch.randn(cout, generator=g)
    ref = F.conv_transpose1d(x.transpose(1, 2), w, bias, stride=u, padding=pp)
    ref = ref + F.conv1d(src[:, None, :], nw, nb, stride=c_["nz_s"], padding=c_["nz_p"])
    up_w, up_b, taps, upad = PW.pack_conv_transpose(w, bias, u, pp, cin, cp)
    assert ops.upsample_noise_supported(u, cp, cin)
    dev = lambda t: t.to(device)
    y = ops.upsample_noise(dev(x), dev(up_w), dev(up_b), taps, upad, u, cp, dev(src), dev(PW.pack_conv(nw, n_pad=cp)),
                           dev(PW.pad_vec(nb, cp)), c_["nz_k"], c_["nz_s"], c_["nz_p"])
    assert tuple(y.shape) == (B, T * u, cp)
    _close(y[..., :cout], ref.transpose(1, 2), 2e-5, c_["id"])
    if cp > cout:
        assert float(y[..., cout:].abs().max()) == 0.0
    # noise-only mode on top of a GEMM-computed ups[i](x)
    y0 = ops.conv(dev(x), dev(up_w), dev(up_b), ksize=taps, pad=upad, t_out=T).view(B, T * u, cp)
    y2 = ops.upsample_noise(None, None, None, taps, upad, u, cp, dev(src), dev(PW.pack_conv(nw, n_pad=cp)),
                            dev(PW.pad_vec(nb, cp)), c_["nz_k"], c_["nz_s"], c_["nz_p"], y=y0)
    _close(y2[..., :cout], ref.transpose(1, 2), 2e-5, c_["id"] + " noise-only")


SNAKE_CONV_CASES = [
    dict(id="c10_k3_d1_short", B=2, n=37, c=10, ld=12, k=3, d=1),
    dict(id="c10_k11_d5_res_acc", B=1, n=1100, c=10, ld=12, k=11, d=5, res=True, alpha=1.0 / 3.0, accumulate=True),
    dict(id="c20_k7_d3_res", B=1, n=530, c=20, ld=20, k=7, d=3, res=True),
    dict(id="c20_k11_d1", B=2, n=70, c=20, ld=20, k=11, d=1),
    dict(id="c40_k11_d5_res", B=1, n=300, c=40, ld=40, k=11, d=5, res=True),
    dict(id="c40_k3_d1_tiny", B=1, n=5, c=40, ld=40, k=3, d=1, res=True),
]


def check_snake_conv(ops, c_, device):
    """Fused SnakeAlias -> conv half-step vs oracle SnakeAlias + torch conv1d."""
    g = _g(zlib.crc32(c_["id"].encode()) % 10000)
    B, n, c, ld, k, d = c_["B"], c_["n"], c_["c"], c_["ld"], c_["k"], c_["d"]
    x = torch.zeros(B, n, ld)
    x[..., :c] = torch.randn(B, n, c, generator=g) * 1.5
    al, be = torch.zeros(ld), torch.zeros(ld)
    al[:c], be[:c] = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3
    filt = W.kaiser_sinc_filter().view(-1)
    w = torch.randn(c, c, k, generator=g) / math.sqrt(c * k)
    bias = torch.randn(c, generator=g)
    s = O.snake_alias(x[..., :c].transpose(1, 2), al[:c], be[:c], filt)
    ref = F.conv1d(s, w, bias, dilation=d, padding=(k - 1) * d // 2).transpose(1, 2)
    res = None
    if c_.get("res"):
        res = torch.zeros(B, n, ld)
        res[..., :c] = torch.randn(B, n, c, generator=g)
        ref = ref + res[..., :c]
    alpha = c_.get("alpha", 1.0)
    ref = ref * alpha
    y0 = None
    if c_.get("accumulate"):
        y0 = torch.zeros(B, n, ld)
        y0[..., :c] = torch.randn(B, n, c, generator=g)
        ref = ref + y0[..., :c]
    dev = lambda t: None if t is None else t.to(device)
    wp = PW.pack_conv(w, ld, ld).to(device)
    assert ops.snake_conv_supported(c, ld, k, d)
    out = dev(y0.clone()) if y0 is not None else torch.full((B, n, ld), 7.0).to(device)
    got = ops.snake_conv(dev(x), dev(al), dev(be), dev(filt), wp, dev(PW.pad_vec(bias, ld)), c=c, ksize=k, dilation=d,
                         res=dev(res), alpha=alpha, accumulate=c_.get("accumulate", False), out=out)
    _close(got[..., :c], ref, 2e-5, c_["id"])
    assert float(got[..., c:].abs().max()) == 0.0 if ld > c else True


def check_grouped_launches(ops, device, B=2, n=333, c=40, ld=40, prec=None):
    """Grouped GEMM / SnakeAlias launches (3 problems per grid) are bit-identical to the single launches, and
    block_mean = ((a + b) + c) / 3."""
    g = _g(77 + n + c)
    filt = W.kaiser_sinc_filter().view(-1).to(device)
    xs = [torch.randn(B, n, ld, generator=g).to(device) for _ in range(3)]
    als = [(torch.randn(ld, generator=g) * 0.3).to(device) for _ in range(3)]
    bes = [(torch.randn(ld, generator=g) * 0.3).to(device) for _ in range(3)]
    want = [ops.snake_alias(x, a, b, filt) for x, a, b in zip(xs, als, bes)]
    got = ops.snake_alias_group(xs, als, bes, filt, [torch.empty_like(x) for x in xs])
    for w_, g_ in zip(want, got):
        assert torch.equal(w_, g_)
    if ld % 8 == 0:     # 16-bit-only outputs (the A operand of an _A16 GEMM): the fp32 result rounded / split
        for dt in (torch.float16, torch.bfloat16, SPLIT16):
            outs = [torch.empty(B, n, 2 * ld if dt == SPLIT16 else ld, dtype=torch.bfloat16 if dt == SPLIT16 else dt, device=device) for _ in xs]
            got16 = ops.snake_alias_group(xs, als, bes, filt, outs)
            for w_, g_ in zip(want, got16):
                assert torch.equal(g_.cpu(), _split16(w_.cpu()) if dt == SPLIT16 else w_.cpu().to(dt)), dt
    ks, ds = (3, 7, 11), (1, 3, 5)
    ws = [PW.pack_conv(torch.randn(c, c, k, generator=g) / math.sqrt(c * k), ld, ld).to(device) for k in ks]
    bs = [torch.randn(ld, generator=g).to(device) for _ in ks]
    rs = [torch.randn(B, n, ld, generator=g).to(device) for _ in ks]
    saved_min, ops.lp_min_flops = ops.lp_min_flops, 0.0
    for n_prob in (1, 2, 3):
        with ops.use_precision(prec):
            want = [ops.conv(xs[j], ws[j], bs[j], ksize=ks[j], dilation=ds[j], pad=(ks[j] - 1) * ds[j] // 2, res=rs[j], split_k=1,
                             tile=(6 if c == 80 else 4 if c == 40 else 1)) for j in range(n_prob)]
            got = ops.conv_group([dict(x=xs[j], w=ws[j], bias=bs[j], ksize=ks[j], dilation=ds[j], pad=(ks[j] - 1) * ds[j] // 2, res=rs[j],
                                       out=torch.full((B, n, ld), 7.0).to(device)) for j in range(n_prob)])
        for j in range(n_prob):
            assert torch.equal(want[j], got[j]), (n_prob, j, float((want[j] - got[j]).abs().max()))
        if prec is not None:        # the grouped reduced-precision launch really ran, and stays in its error class
            assert all(getattr(w_, "_svcmi_lp", None) for w_ in ws[:n_prob])
            exact = ops.conv(xs[0], ws[0], bs[0], ksize=ks[0], dilation=ds[0], pad=(ks[0] - 1) * ds[0] // 2, res=rs[0], split_k=1)
            err = float((exact - got[0]).abs().max()) / max(1.0, float(exact.abs().max()))
            assert 0 < err <= {"bf16x3": 2e-5, "f16": 2e-3, "bf16": 1.5e-2}[prec], err
    ops.lp_min_flops = saved_min
    m = ops.block_mean(want)
    assert torch.equal(m.cpu(), ((want[0].cpu() + want[1].cpu()) + want[2].cpu()) / 3.0)


def check_snake_conv_group(ops, device, c=20, ld=20, B=2, n=300):
    """Grouped fused half-steps (3 / 7 / 11 taps in one launch) equal the single launches bit for bit."""
    g = _g(500 + c + n)
    filt = W.kaiser_sinc_filter().view(-1).to(device)
    probs, want = [], []
    for k, d in ((3, 1), (11, 5), (7, 3)):
        x = torch.zeros(B, n, ld)
        x[..., :c] = torch.randn(B, n, c, generator=g)
        res = torch.zeros(B, n, ld)
        res[..., :c] = torch.randn(B, n, c, generator=g)
        al, be = torch.zeros(ld), torch.zeros(ld)
        al[:c], be[:c] = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3
        w = PW.pack_conv(torch.randn(c, c, k, generator=g) / math.sqrt(c * k), ld, ld).to(device)
        bias = PW.pad_vec(torch.randn(c, generator=g), ld).to(device)
        pr = dict(x=x.to(device), alpha_log=al.to(device), beta_log=be.to(device), w=w, bias=bias, ksize=k, dilation=d,
                  res=res.to(device), alpha=0.5)
        want.append(ops.snake_conv(pr["x"], pr["alpha_log"], pr["beta_log"], filt, w, bias, c=c, ksize=k, dilation=d, res=pr["res"], alpha=0.5))
        probs.append(dict(pr, out=torch.full((B, n, ld), 7.0).to(device)))
    assert ops.lib.svcmi_tune_set(b"amp_mfma", 0) == 0       # the vector-ALU kernels: the single launches' arithmetic, bit for bit
    try:
        for amp_u in ((-1, 1) if c <= 20 else (0,)):     # 1: the variant that keeps the up-sampled activation tile in LDS (snake_tile_u), -1: never; same bits
            assert ops.lib.svcmi_tune_set(b"amp_u", amp_u) == 0
            try:
                for n_prob in (1, 2, 3):
                    for pr in probs:
                        pr["out"].fill_(7.0)
                    got = ops.snake_conv_group(probs[:n_prob], filt, c=c)
                    for j in range(n_prob):
                        assert torch.equal(got[j], want[j]), (amp_u, n_prob, j, float((got[j] - want[j]).abs().max()))
            finally:
                ops.lib.svcmi_tune_set(b"amp_u", 0)
    finally:
        ops.lib.svcmi_tune_set(b"amp_mfma", 1)
    if c > 20:
        return
    # the product default: the same half-step with its convolution on the fp32 matrix cores (snake_convm_group_kernel) -- fp32 products
    # and sums in another order; both activation-phase forms give the same bits
    outs = {}
    for form in (2, 3):
        assert ops.lib.svcmi_tune_set(b"amp_mfma", form) == 0
        try:
            for n_prob in (1, 3):
                for pr in probs:
                    pr["out"].fill_(7.0)
                got = ops.snake_conv_group(probs[:n_prob], filt, c=c)
                for j in range(n_prob):
                    _close(got[j], want[j].cpu(), 1e-5, f"snake_conv_group on the fp32 matrix cores c={c} form={form} problem {j}")
                    assert not torch.equal(got[j], want[j])          # (it really is the other kernel)
                    if ld > c:
                        assert float(got[j][..., c:].abs().max()) == 0.0
                outs[(form, n_prob)] = [g_.clone() for g_ in got[:n_prob]]
        finally:
            ops.lib.svcmi_tune_set(b"amp_mfma", 1)
    for n_prob in (1, 3):
        for a, b in zip(outs[(2, n_prob)], outs[(3, n_prob)]):
            assert torch.equal(a, b)


def check_snake_conv_group_lp(ops, device, c=20, ld=20, B=2, n=300, precision="f16"):
    """The grouped half-step with its convolution on the fp16 matrix cores (svcmi_snake_conv_group_lp) against the SAME arithmetic in
    torch: S = SnakeAlias(x) rounded to fp16, weights rounded to fp16 ("f16") or kept as hi + lo fp16 ("f16w2"), products accumulated
    wide -- what is left is the fp32 accumulation order (1e-5) -- and, loosely, against the fp32 kernel (the error class of the mode).
    3 / 7 / 11 taps x dilations 1 / 5 / 3, residual, alpha, accumulate, sequence ends inside a tile, both activation-phase variants."""
    g = _g(900 + c + n)
    filt_c = W.kaiser_sinc_filter().view(-1)
    filt = filt_c.to(device)
    probs, want, want32 = [], [], []
    for i, (k, d) in enumerate(((3, 1), (11, 5), (7, 3))):
        x = torch.zeros(B, n, ld)
        x[..., :c] = torch.randn(B, n, c, generator=g) * 1.5
        res = torch.zeros(B, n, ld)
        res[..., :c] = torch.randn(B, n, c, generator=g)
        y0 = torch.zeros(B, n, ld)
        y0[..., :c] = torch.randn(B, n, c, generator=g)
        al, be = torch.zeros(ld), torch.zeros(ld)
        al[:c], be[:c] = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3
        w = torch.randn(c, c, k, generator=g) / math.sqrt(c * k)
        bias = torch.randn(c, generator=g)
        acc = i == 1
        s = O.snake_alias(x[..., :c].transpose(1, 2), al[:c], be[:c], filt_c)
        # the rounded operand from the library's own SnakeAlias (the fused tile runs the same operation sequence, csrc/snake_math.h): an
        # activation that differs from the oracle's in the last fp32 bit can land on the other side of an fp16 rounding boundary, and one
        # such flip is 2^-11 |s| |w| ~ 2e-4 -- the oracle's S (checked against the library's to 1e-5 by check_snake) only feeds the loose bound
        s_lib = ops.snake_alias(x.to(device), al.to(device), be.to(device), filt).cpu()[..., :c].transpose(1, 2)
        s16 = s_lib.half().double()
        w_hi = w.half()
        wq = w_hi.double() + ((w - w_hi.float()).half().double() if precision == "f16w2" else 0.0)
        ref = F.conv1d(s16, wq, bias.double(), dilation=d, padding=(k - 1) * d // 2).transpose(1, 2)
        ref = (ref + res[..., :c].double()) * 0.5 + (y0[..., :c].double() if acc else 0.0)
        want.append(ref.float())
        ref32 = F.conv1d(s, w, bias, dilation=d, padding=(k - 1) * d // 2).transpose(1, 2)
        want32.append((ref32 + res[..., :c]) * 0.5 + (y0[..., :c] if acc else 0.0))
        probs.append(dict(x=x.to(device), alpha_log=al.to(device), beta_log=be.to(device), w=PW.pack_conv(w, ld, ld).to(device),
                          bias=PW.pad_vec(bias, ld).to(device), ksize=k, dilation=d, res=res.to(device), alpha=0.5, accumulate=acc, y0=y0))
    code = PRECISIONS[precision]
    for k, d in ((3, 1), (7, 3), (11, 5)):
        assert ops.lib.svcmi_snake_conv_lp_supported(c, ld, k, d, code)
    assert not ops.lib.svcmi_snake_conv_lp_supported(40, 40, 3, 1, code) and not ops.lib.svcmi_snake_conv_lp_supported(c, ld, 3, 1, 2)
    for amp_u in (-1, 1):
        assert ops.lib.svcmi_tune_set(b"amp_u", amp_u) == 0
        try:
            for n_prob in (1, 3):
                for pr in probs:
                    pr["out"] = pr["y0"].clone().to(device) if pr["accumulate"] else torch.full((B, n, ld), 7.0).to(device)
                got = ops.snake_conv_group(probs[:n_prob], filt, c=c, precision=precision)
                for j in range(n_prob):
                    _close(got[j][..., :c], want[j], 2e-5, f"snake_conv_group_lp {precision} c={c} amp_u={amp_u} problem {j}")
                    err = float((got[j][..., :c].cpu() - want32[j]).abs().max())
                    assert 0 < err < (2e-3 if precision == "f16w2" else 4e-3) * max(1.0, float(want32[j].abs().max())), (precision, j, err)
                    if ld > c:
                        assert float(got[j][..., c:].abs().max()) == 0.0
        finally:
            ops.lib.svcmi_tune_set(b"amp_u", 0)



def check_snake_post(ops, device, B=2, n=700):
    """Fused output layer (SnakeAlias -> conv_post 10 -> 1, k = 7, no bias -> tanh) vs oracle SnakeAlias + torch conv1d."""
    g = _g(4242 + n)
    c, ld, k = 10, 12, 7
    x = torch.zeros(B, n, ld)
    x[..., :c] = torch.randn(B, n, c, generator=g) * 1.5
    al, be = torch.zeros(ld), torch.zeros(ld)
    al[:c], be[:c] = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3
    filt = W.kaiser_sinc_filter().view(-1)
    w = torch.randn(1, c, k, generator=g) / math.sqrt(c * k)
    s = O.snake_alias(x[..., :c].transpose(1, 2), al[:c], be[:c], filt)
    ref = torch.tanh(F.conv1d(s, w, None, padding=(k - 1) // 2))[:, 0]
    assert ops.snake_post_supported(c, ld, k) and not ops.snake_post_supported(2, 4, k)
    got = ops.snake_post(x.to(device), al.to(device), be.to(device), filt.to(device), PW.pack_conv(w, cin_pad=ld).to(device), c=c, ksize=k)
    _close(got, ref, 2e-5, f"snake_post n={n}")


def check_flow_glue(ops, device):
    g = _g(11)
    B, T, H = 2, 13, 8
    lengths = torch.tensor([13, 6], dtype=torch.int32)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).float().unsqueeze(-1)
    a = torch.randn(B, T, 2 * H, generator=g)
    _close(ops.wn_gate(a.to(device)), torch.tanh(a[..., :H]) * torch.sigmoid(a[..., H:]), 1e-6, "gate")
    slabs, gb = torch.randn(B, 3, T, 2 * H, generator=g), torch.randn(2 * H, generator=g)      # split-K slabs + bias
    v = (slabs[:, 0] + slabs[:, 1]) + slabs[:, 2] + gb
    _close(ops.wn_gate(slabs.to(device), bias=gb.to(device)), torch.tanh(v[..., :H]) * torch.sigmoid(v[..., H:]), 1e-6, "gate slabs")
    # update, not last
    rs, x, skip = torch.randn(B, T, 2 * H, generator=g), torch.randn(B, T, H, generator=g), torch.randn(B, T, H, generator=g)
    xd, sd = x.clone().to(device), skip.clone().to(device)
    ops.wn_update(rs.to(device), xd, sd, lengths.to(device), first=False, last=False)
    _close(xd, (x + rs[..., :H]) * mask, 1e-6, "wn x")
    _close(sd, skip + rs[..., H:], 1e-6, "wn skip")
    sd = torch.full((B, T, H), 7.0).to(device)
    ops.wn_update(rs.to(device), x.clone().to(device), sd, lengths.to(device), first=True, last=False)
    _close(sd, rs[..., H:], 1e-6, "wn skip first")
    rs1 = torch.randn(B, T, H, generator=g)
    sd = skip.clone().to(device)
    ops.wn_update(rs1.to(device), None, sd, lengths.to(device), first=False, last=True)
    _close(sd, (skip + rs1) * mask, 1e-6, "wn skip last")
    # coupling
    half = 6
    xx = torch.randn(B, T, 2 * half, generator=g)
    msvs = torch.randn(B, 2 * half, generator=g) * 0.3
    ms, vs = msvs[:, None, :half], msvs[:, None, half:]
    for x0_off in (0, half):
        got = ops.coupling_pre(xx.to(device), x0_off, msvs.to(device), lengths.to(device), half)
        _close(got, (xx[..., x0_off:x0_off + half] - ms) * torch.exp(-vs) * mask, 1e-6, "coupling pre")
        m = torch.randn(B, T, half, generator=g) * mask
        xd = xx.clone().to(device)
        x1_off = half - x0_off
        ops.coupling_post(xd, x1_off, m.to(device), msvs.to(device), lengths.to(device), half)
        want = xx.clone()
        x1 = (xx[..., x1_off:x1_off + half] - m) * mask
        want[..., x1_off:x1_off + half] = (ms + x1 * torch.exp(vs)) * mask
        _close(xd, want, 1e-6, "coupling post")


def check_prior_glue(ops, device):
    g = _g(12)
    B, T, Cc, I = 2, 37, 8, 12
    lengths = torch.tensor([37, 20], dtype=torch.int32)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, T, Cc, generator=g)
    pit = torch.cat([torch.zeros(B, 5), torch.rand(B, T - 5, generator=g) * 1200.0], dim=1).round()
    emb = torch.randn(256, Cc, generator=g)
    xd = x.clone().to(device)
    ops.embed_pitch(xd, pit.to(device), emb.to(device), lengths.to(device))
    _close(xd, (x + emb[O.f0_to_coarse(pit)]) * mask, 1e-6, "embed_pitch")
    stats = torch.randn(B, T, 2 * I, generator=g) * 0.5
    noise = torch.randn(B, I, T, generator=g)
    want = (stats[..., :I] + noise.transpose(1, 2) * torch.exp(stats[..., I:])) * mask
    _close(ops.sample_prior(stats.to(device), noise.to(device), lengths.to(device)), want, 1e-6, "sample_prior")


def check_bridges(ops, device):
    g = _g(13)
    B, Cc, T = 2, 80, 45
    x, n = torch.randn(B, Cc, T, generator=g), torch.randn(B, Cc, T, generator=g)
    _close(ops.ncl_to_nlc(x.to(device), n.to(device), 0.1), (x + 0.1 * n).transpose(1, 2), 1e-7, "ncl_to_nlc")
    y = ops.ncl_to_nlc(x[:, :6].contiguous().to(device), ld=8)
    _close(y[..., :6], x[:, :6].transpose(1, 2), 1e-7, "ncl_to_nlc padded")
    assert float(y[..., 6:].abs().max()) == 0.0
    z = torch.randn(B, T, 12, generator=g)
    _close(ops.nlc_to_ncl(z.to(device), c=10), z[..., :10].transpose(1, 2), 1e-7, "nlc_to_ncl")


def check_pitch2source(ops, T, B, device, hop=320):
    from workload import config as C
    from workload import inputs as I
    hp = C.base_hp()
    if hop != 320:
        hp = C.AttrDict({**C.BASE, "gen": {**C.BASE["gen"], "upsample_rates": [hop], "upsample_kernel_sizes": [2 * hop]}})
    g = _g(T)
    f0 = torch.stack([I.synth_f0(T, seed=3 + b, base=200.0 + 150 * b) for b in range(B)])
    rand_ini = torch.rand(B, 11, generator=g)
    noise = torch.randn(B, T * hop, 11, generator=g)
    sd = {"dec.m_source.merge_w": torch.tensor([C.NSF_MERGE_W]), "dec.m_source.merge_b": torch.tensor([C.NSF_MERGE_B])}
    want = O.pitch2source(sd, hp, f0, rand_ini, noise)[:, 0]
    got = ops.pitch2source(f0.to(device), rand_ini.to(device), noise.to(device),
                           sd["dec.m_source.merge_w"].view(-1).to(device), C.NSF_MERGE_B, hop, 32000.0)
    # the reference accumulates 3.2e5 fp32 adds; re-association is worth ~3e-6
    _close(got, want, 5e-5, "pitch2source")


def check_source2wav(ops, device):
    g = _g(14)
    x = torch.cat([torch.randn(1000, generator=g) * 0.6, torch.tensor([1.0, -1.0, 0.99999, 2.0, -2.0, 0.0])])
    got = ops.source2wav(x.to(device)).cpu()
    assert torch.equal(got, torch.from_numpy(O.source2wav(x)))


def check_outputs16(ops, device):
    """The optional 16-bit output copies of LayerNorm / split-K LayerNorm / attention (the A operands of the _A16 GEMMs): exactly the fp32
    output rounded to the requested type, or split into (hi, lo) bf16 planes (SPLIT16)."""
    g = _g(99)
    B, T, c, H = 2, 37, 64, 2
    x, r = torch.randn(B, T, c, generator=g), torch.randn(B, T, c, generator=g)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    d = lambda t: t.to(device)
    for dt in (torch.float16, torch.bfloat16, SPLIT16):
        want = (lambda y: _split16(y.cpu())) if dt == SPLIT16 else (lambda y: y.cpu().to(dt))
        y, y16 = ops.layernorm(d(x), d(gamma), d(beta), res=d(r), out16=dt)
        assert torch.equal(y16.cpu(), want(y))
        part = torch.randn(B, 3, T, c, generator=g)
        xs = d(x.clone())
        y, y16 = ops.splitk_layernorm(d(part), d(beta), xs, d(gamma), d(beta), out16=dt)
        assert torch.equal(y16.cpu(), want(y))
        qkv = torch.randn(B, 70, 3 * H * 32, generator=g)
        o, o16 = ops.attention(d(qkv), H, 32 ** -0.5, out16=dt)
        assert torch.equal(o16.cpu(), want(o))
        lens = torch.tensor([70, 41], dtype=torch.int32)
        o, o16 = ops.attention(d(qkv), H, 32 ** -0.5, lengths=d(lens), out16=dt)            # (another kernel shape: masked)
        assert torch.equal(o16.cpu(), want(o))
    y, y16 = ops.layernorm(d(x[..., :36].contiguous()), out16=SPLIT16)                      # a row length with pad columns (Cp = 40)
    assert y16.shape[-1] == 80 and torch.equal(y16.cpu(), _split16(y.cpu()))
    # CREPE's BatchNorm + max-pool (svcmi_bn_maxpool2_f32) with its 16-bit second output
    xb, sc, sh = torch.randn(3, 10, 24, generator=g), torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g)
    want = torch.maximum(xb[:, 0::2] * sc + sh, xb[:, 1::2] * sc + sh)
    for dt in (torch.float16, torch.bfloat16, SPLIT16):
        yb, yb16 = ops.bn_maxpool2(d(xb), d(sc), d(sh), out16=dt)
        _close(yb, want, 1e-6, "bn_maxpool2")
        assert torch.equal(yb16.cpu(), _split16(yb.cpu()) if dt == SPLIT16 else yb.cpu().to(dt)), dt


# attention on the 16-bit matrix cores (svcmi_attention16): every block shape, ragged lengths, key ranges past T, both formats
ATTN16_CASES = [
    dict(id="a16_f16_d64_T130_41", B=1, T=130, H=2, D=64, fmt="f16", shape=41),
    dict(id="a16_bf16_d64_T300_42", B=1, T=300, H=2, D=64, fmt="bf16", shape=42),
    dict(id="a16_f16_d64_ragged_T257_42", B=2, T=257, H=2, D=64, fmt="f16", shape=42, lengths=[257, 140]),
    dict(id="a16_f16_d32_T70_auto", B=1, T=70, H=3, D=32, fmt="f16", shape=0),
    dict(id="a16_bf16_d32_T40_44_empty_ranges", B=1, T=40, H=1, D=32, fmt="bf16", shape=44),
    dict(id="a16_f16_d64_T150_81", B=1, T=150, H=2, D=64, fmt="f16", shape=81),
    dict(id="a16_f16_d64_ragged_T260_82", B=2, T=260, H=1, D=64, fmt="f16", shape=82, lengths=[200, 260]),
    # with the relative-position band of the prior encoder (2 heads x 96 at base.yaml)
    dict(id="a16_f16_d96_rel_T67_21", B=1, T=67, H=2, D=96, fmt="f16", shape=21, rel=True, W=4, lengths=[60]),
    dict(id="a16_bf16_d96_rel_T260_24", B=2, T=260, H=2, D=96, fmt="bf16", shape=24, rel=True, W=4, lengths=[260, 201]),
    dict(id="a16_f16_d96_rel_T140_14", B=1, T=140, H=2, D=96, fmt="f16", shape=14, rel=True, W=4),
    dict(id="a16_f16_d32_rel_w2_T140_42", B=1, T=140, H=2, D=32, fmt="f16", shape=42, rel=True, W=2, lengths=[133]),
    dict(id="a16_f16_d96_rel_short_T3_auto", B=1, T=3, H=2, D=96, fmt="f16", shape=0, rel=True, W=4),
    dict(id="a16_f16_d32_rel_T300_44", B=1, T=300, H=1, D=32, fmt="f16", shape=44, rel=True, W=4),
]
ATTN16_CASES_LARGE = [
    dict(id="a16_f16_whisper_T500", B=1, T=500, H=20, D=64, fmt="f16", shape=0),
    dict(id="a16_bf16_whisper_T750_B2", B=2, T=750, H=20, D=64, fmt="bf16", shape=0),
    dict(id="a16_f16_whisper_T500_B16_81", B=16, T=500, H=20, D=64, fmt="f16", shape=81),
    dict(id="a16_f16_encp_T1000_rel", B=1, T=1000, H=2, D=96, fmt="f16", shape=0, rel=True, W=4),
    dict(id="a16_bf16_encp_ragged_B3_rel", B=3, T=301, H=2, D=96, fmt="bf16", shape=0, rel=True, W=4, lengths=[301, 250, 7]),
]


def check_attention16(ops, c, device):
    g = _g(31 + c["T"])
    B, T, H, D = c["B"], c["T"], c["H"], c["D"]
    dt = torch.float16 if c["fmt"] == "f16" else torch.bfloat16
    qkv16 = torch.randn(B, T, 3 * H * D, generator=g).to(dt)
    lengths = torch.tensor(c["lengths"], dtype=torch.int32) if "lengths" in c else None
    scale = D ** -0.5
    rel_k = torch.randn(2 * c["W"] + 1, D, generator=g) * D ** -0.5 if c.get("rel") else None
    rel_v = torch.randn(2 * c["W"] + 1, D, generator=g) * D ** -0.5 if c.get("rel") else None
    want = attention_reference(qkv16.float(), H, scale, rel_k, rel_v, c.get("W", 0), lengths)        # fp64 attention of the SAME 16-bit operands
    assert ops.lib.svcmi_tune_set(b"attn16", int(c["shape"])) == 0
    dev = lambda t: None if t is None else t.to(device)
    try:
        o, o16 = ops.attention16(qkv16.to(device), H, scale, rel_k=dev(rel_k), rel_v=dev(rel_v), window=c.get("W", 0), lengths=dev(lengths))
    finally:
        ops.lib.svcmi_tune_set(b"attn16", 0)
    assert torch.equal(o16.cpu(), o.cpu().to(dt)), c["id"]
    # the only rounding left is P -> 16 bits before the PV product: rel. 2^-11 (f16) / 2^-8 (bf16) per probability
    _close(o, want, 1.5e-3 if c["fmt"] == "f16" else 1.2e-2, c["id"])


def check_kernels_in_flight_beside_fp16_half_step(ops, victim="alias", replays=8, launches=30, culprit_launches=60):
    """Round 6: launches of one kernel (`victim`) on one stream while the fp16 fused half-step (v_mfma_f32_16x16x32_f16) runs on another must
    give the bits they give alone.  They did not: MI355X computes packed-fp32 instructions with the src1 half-select wrongly in lanes 48..63
    beside that matrix-core shape (tests/test_isa_packed_operand_select.py, scripts/probes/pkfma_mfma_corun.hip), which showed as SnakeAlias
    values of one clip off by 1e-2 with two 16-bit clips in flight.  Victims: the SnakeAlias stream kernel, the fp32 fused half-steps (vector and
    matrix-core forms), the stage entry (upsample + noise convolution) -- every kernel family that holds packed fp32 arithmetic."""
    dev = "cuda"
    g = _g(4242)
    filt = W.kaiser_sinc_filter().view(-1).to(dev)

    def graph(fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(); fn()
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            o = fn()
        return gr, o, s

    def amp_problems(c, ld, n):
        probs = []
        for k, dil in ((3, 1), (11, 5), (7, 3)):
            x = torch.zeros(1, n, ld); x[..., :c] = torch.randn(1, n, c, generator=g)
            res = torch.zeros(1, n, ld); res[..., :c] = torch.randn(1, n, c, generator=g)
            al, be = torch.zeros(ld), torch.zeros(ld)
            al[:c], be[:c] = torch.randn(c, generator=g) * 0.3, torch.randn(c, generator=g) * 0.3
            w = PW.pack_conv(torch.randn(c, c, k, generator=g) / math.sqrt(c * k), ld, ld).to(dev)
            bias = PW.pad_vec(torch.randn(c, generator=g), ld).to(dev)
            probs.append(dict(x=x.to(dev), alpha_log=al.to(dev), beta_log=be.to(dev), w=w, bias=bias, ksize=k, dilation=dil, res=res.to(dev), alpha=0.5))
        return probs

    cp = amp_problems(20, 20, 48000)
    couts = [torch.empty_like(p["x"]) for p in cp]

    def culprit():
        for _ in range(culprit_launches):
            ops.snake_conv_group([dict(p, out=o) for p, o in zip(cp, couts)], filt, c=20, precision="f16w2")
        return couts

    if victim == "alias":
        xs = [torch.randn(1, 24000, 40, generator=g).to(dev) for _ in range(3)]
        al = [(torch.randn(40, generator=g) * 0.3).to(dev) for _ in range(3)]
        be = [(torch.randn(40, generator=g) * 0.3).to(dev) for _ in range(3)]
        outs = [[torch.empty_like(x) for x in xs] for _ in range(launches)]

        def vfn():
            for r in range(launches):
                ops.snake_alias_group(xs, al, be, filt, outs[r])
            return [o for oo in outs for o in oo]
    elif victim in ("amp10", "amp20", "amp20_vector"):
        c, ld, n = (10, 12, 96000) if victim == "amp10" else (20, 20, 48000)
        vp = amp_problems(c, ld, n)
        outs = [[torch.empty_like(p["x"]) for p in vp] for _ in range(launches)]

        def vfn():
            for r in range(launches):
                ops.snake_conv_group([dict(p, out=o) for p, o in zip(vp, outs[r])], filt, c=c)
            return [o for oo in outs for o in oo]
    else:
        raise ValueError(victim)
    tuned = victim == "amp20_vector"
    if tuned:
        assert ops.lib.svcmi_tune_set(b"amp_mfma", 0) == 0
    try:
        A, B = graph(culprit), graph(vfn)
        torch.cuda.synchronize()
        with torch.cuda.stream(B[2]):
            B[0].replay()
        B[2].synchronize()
        ref = [o.clone() for o in B[1]]
        with torch.cuda.stream(A[2]):
            A[0].replay()
        A[2].synchronize()
        cref = [o.clone() for o in A[1]]
        for rep in range(replays):
            for gr, _, s in (A, B):
                with torch.cuda.stream(s):
                    gr.replay()
            torch.cuda.synchronize()
            for i, (o, r) in enumerate(zip(B[1], ref)):
                assert torch.equal(o, r), f"{victim}: replay {rep}, launch {i // 3}, problem {i % 3}: {int((o != r).sum())} values differ, max {float((o - r).abs().max()):.3e}"
            for o, r in zip(A[1], cref):
                assert torch.equal(o, r), f"the fp16 half-step itself differs beside {victim} (replay {rep})"
    finally:
        if tuned:
            assert ops.lib.svcmi_tune_set(b"amp_mfma", 1) == 0
