#!/usr/bin/env python
"""Kernel microbenchmarks on the GPU box (tuning aid, not the judged bench):
    python scripts/microbench.py gemm|snake|dec|attn [...]
Times single kernels through the C ABI on the launch stream with HIP events, prints TFLOP/s or GB/s."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from svcmi import Ops  # noqa: E402
from svcmi import weights as PW  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters      # us


def gemm(ops, tag, T, cin, n, k=1, dil=1, res=False, tiles=(0, 1, 2, 3), splits=(1, 0), B=1, prec=None, partials=False, a16=False):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, cin, generator=g).cuda()
    w = PW.pack_conv(torch.randn(n, cin, k, generator=g) / math.sqrt(cin * k)).cuda()
    bias = torch.randn(n, generator=g).cuda()
    r = torch.randn(B, T, n, generator=g).cuda() if res else None
    out = torch.empty(B, T, n, device="cuda")
    fl = 2.0 * B * T * n * k * cin
    x16 = None                                                                             # the producer's 16-bit copy: _A16 kernels
    if a16 and prec == "bf16x3":                                                           # ... split rows [hi | lo]
        hi = x.bfloat16()
        x16 = torch.cat([hi, (x - hi.float()).bfloat16()], dim=-1).contiguous()
    elif a16:
        x16 = x.to({"f16": torch.float16, "bf16": torch.bfloat16}[prec])
    for tile in tiles:
        for sk in splits:
            try:
                with ops.use_precision(prec):
                    if partials and sk >= 1:
                        us = timeit(lambda: ops.conv(x, w, None, ksize=k, dilation=dil, pad=(k - 1) * dil // 2, tile=tile, split_k=sk, partials=True, x16=x16))
                    else:
                        us = timeit(lambda: ops.conv(x, w, bias, ksize=k, dilation=dil, pad=(k - 1) * dil // 2, res=r, out=out, tile=tile, split_k=sk, n_out=n, x16=x16))
                print(f"gemm {tag:18s} {(prec or 'f32') + ('+a16' if a16 else ''):8s} B={B} T={T} cin={cin} n={n} k={k} d={dil} tile={tile} split={sk}{' partials' if partials else ''}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
            except Exception as e:      # noqa: BLE001
                print(f"gemm {tag} tile={tile} split={sk}: {e}")


def main():
    global timeit
    what = sys.argv[1:] or ["gemm", "snake", "dec", "attn"]
    ops = Ops()
    if "attnrel" in what:     # relative-position attention of the prior encoder (2 heads x 96): register-fed kernel vs the LDS-staged one
        H, D, W = 2, 96, 4
        for (B, T) in (((16, 1000),) if "pmc" in what else ((1, 1000), (4, 1000), (16, 1000), (1, 2510))):
            g = torch.Generator().manual_seed(B * T)
            qkv = torch.randn(B, T, 3 * H * D, generator=g).cuda()
            rk = (torch.randn(2 * W + 1, D, generator=g) * D ** -0.5).cuda()
            rv = (torch.randn(2 * W + 1, D, generator=g) * D ** -0.5).cuda()
            out = torch.empty(B, T, H * D, device="cuda")
            fl = 4.0 * B * T * T * H * D
            base = None
            for code in ((-1, 82) if "pmc" in what else (-1, 0, 41, 42, 81, 82)):
                ops.lib.svcmi_tune_set(b"attn_lds", code)
                try:
                    o = ops.attention(qkv, H, D ** -0.5, rel_k=rk, rel_v=rv, window=W, out=out).clone()
                    us = timeit(lambda: ops.attention(qkv, H, D ** -0.5, rel_k=rk, rel_v=rv, window=W, out=out))
                    if base is None:
                        base = o
                    print(f"attnrel B={B} T={T} attn_lds={code:3d}: {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s  max diff to the register-fed kernel {float((o - base).abs().max()):.1e}", flush=True)
                finally:
                    ops.lib.svcmi_tune_set(b"attn_lds", 0)
    if "gemm" in what:
        gemm(ops, "whisper_qkv", 500, 1280, 3840)
        gemm(ops, "whisper_o", 500, 1280, 1280, res=True, splits=(1, 0, 2, 4))
        gemm(ops, "whisper_mlp1", 500, 1280, 5120)
        gemm(ops, "whisper_mlp2", 500, 5120, 1280, res=True, splits=(1, 0, 4, 8))
    if "lp" in what:          # reduced-precision operand modes on the Whisper window shapes (M = 500 / 750) and a batch-16 flow shape
        for T in (500, 750):
            gemm(ops, "whisper_qkv", T, 1280, 3840, tiles=(1,), splits=(1,))
            gemm(ops, "whisper_mlp1", T, 1280, 5120, tiles=(6,), splits=(1,))
            gemm(ops, "whisper_o", T, 1280, 1280, tiles=(6,), splits=(2,), partials=True)
            gemm(ops, "whisper_mlp2", T, 5120, 1280, tiles=(6,), splits=(4,), partials=True)
            for prec in ("bf16x3", "bf16", "f16"):
                gemm(ops, "whisper_qkv", T, 1280, 3840, tiles=(1, 9, 3), splits=(1,), prec=prec)
                gemm(ops, "whisper_mlp1", T, 1280, 5120, tiles=(1, 9, 3), splits=(1,), prec=prec)
                gemm(ops, "whisper_o", T, 1280, 1280, tiles=(1, 9), splits=(1, 2, 3, 4), prec=prec, partials=True)
                gemm(ops, "whisper_mlp2", T, 5120, 1280, tiles=(1, 9), splits=(2, 4, 8), prec=prec, partials=True)
        for prec in (None, "bf16x3", "bf16"):
            gemm(ops, "square4096", 4096, 4096, 4096, tiles=(9, 3) if prec else (3,), splits=(1,), prec=prec)
            gemm(ops, "flow_in_B16", 1000, 192, 384, k=5, tiles=(0,), splits=(1,), B=16, prec=prec)
            gemm(ops, "encp_pre_B16", 1000, 1280, 192, k=5, tiles=(0,), splits=(1,), B=16, prec=prec)
            gemm(ops, "dec_pre_B16", 1000, 192, 320, k=7, tiles=(0,), splits=(1,), B=16, prec=prec)
            gemm(ops, "stage0_C160_B16", 5000, 160, 160, k=7, dil=3, res=True, tiles=(0,), splits=(1,), B=16, prec=prec)
            gemm(ops, "stage1_C80_B16", 20000, 80, 80, k=7, dil=3, res=True, tiles=(0,), splits=(1,), B=16, prec=prec)
            gemm(ops, "stage2_C40_B4", 80000, 40, 40, k=7, dil=3, res=True, tiles=(0,), splits=(1,), B=4, prec=prec)
    if "a16" in what:         # 16-bit activations (K-step 64) against the in-register rounding kernels on the Whisper window shapes
        for T in (500, 750, 1500):
            for a16 in (False, True):
                gemm(ops, "whisper_qkv", T, 1280, 3840, tiles=(1, 9, 3), splits=(1,), prec="f16", a16=a16)
                gemm(ops, "whisper_mlp1", T, 1280, 5120, tiles=(1, 9, 3), splits=(1,), prec="f16", a16=a16)
                gemm(ops, "whisper_o", T, 1280, 1280, tiles=(1, 9), splits=(1, 2, 4), prec="f16", partials=True, a16=a16)
                gemm(ops, "whisper_mlp2", T, 5120, 1280, tiles=(1, 9), splits=(1, 2, 4, 8), prec="f16", partials=True, a16=a16)
    if "biggemm" in what:     # chip-filling GEMMs, one tile policy per kernel family (for PMC runs: scripts/pmc_cmd.sh)
        gemm(ops, "square4096", 4096, 4096, 4096, tiles=(3,), splits=(1,))
        gemm(ops, "square4096", 4096, 4096, 4096, tiles=(3,), splits=(1,), prec="bf16x3")
        gemm(ops, "square4096", 4096, 4096, 4096, tiles=(3,), splits=(1,), prec="bf16")
        gemm(ops, "square4096", 4096, 4096, 4096, tiles=(3,), splits=(1,), prec="bf16", a16=True)
        gemm(ops, "crepe_l2_B512", 128, 1024, 128, k=64, tiles=(3,), splits=(1,), B=512, prec="bf16x3")
    if "x3a" in what:         # split-bf16 products: in-register split (K-step 32) vs split activation rows from the producer (K-step 64)
        for T in (500, 750):
            for a16 in (False, True):
                gemm(ops, "whisper_qkv", T, 1280, 3840, tiles=(1, 9, 3), splits=(1,), prec="bf16x3", a16=a16)
                gemm(ops, "whisper_mlp1", T, 1280, 5120, tiles=(1, 9, 3), splits=(1,), prec="bf16x3", a16=a16)
                gemm(ops, "whisper_o", T, 1280, 1280, tiles=(1, 9), splits=(2, 4), prec="bf16x3", partials=True, a16=a16)
                gemm(ops, "whisper_mlp2", T, 5120, 1280, tiles=(1, 9), splits=(2, 4, 8), prec="bf16x3", partials=True, a16=a16)
        for a16 in (False, True):
            gemm(ops, "square4096", 4096, 4096, 4096, tiles=(9, 3), splits=(1,), prec="bf16x3", a16=a16)
            gemm(ops, "crepe_l2_B512", 128, 1024, 128, k=64, tiles=(1, 9, 3), splits=(1,), B=512, prec="bf16x3", a16=a16)
            gemm(ops, "crepe_l5_B512", 16, 128, 256, k=64, tiles=(1, 9, 3), splits=(1,), B=512, prec="bf16x3", a16=a16)
    if "a16" in what:
        for a16 in (False, True):
            gemm(ops, "square4096", 4096, 4096, 4096, tiles=(9, 3), splits=(1,), prec="bf16", a16=a16)
            gemm(ops, "stage1_C80_B16", 20000, 80, 80, k=7, dil=3, res=True, tiles=(0,), splits=(1,), B=16, prec="bf16", a16=a16)
            gemm(ops, "stage0_C160_B16", 5000, 160, 160, k=7, dil=3, res=True, tiles=(0,), splits=(1,), B=16, prec="bf16", a16=a16)
    if "gemmpmc" in what:     # few launches, for counter collection
        _t = timeit
        timeit = lambda fn, iters=3, warm=1: _t(fn, iters, warm)
        gemm(ops, "whisper_mlp1", 500, 1280, 5120, tiles=(1, 2, 3), splits=(1,))
        gemm(ops, "whisper_mlp2", 500, 5120, 1280, res=True, tiles=(1, 3), splits=(8, 1))
        gemm(ops, "square4096", 4096, 4096, 4096, tiles=(1, 2, 3), splits=(1,))
    if "ktrace" in what:      # SVCMI_LIB = a -DSVCMI_PROBE_KTRACE=1 build: where the cycles of a K-step go, per wave (s_memtime sums)
        import ctypes
        import numpy as np
        lib = ctypes.CDLL(os.environ["SVCMI_LIB"])
        for (tag, T, cin, n, tile, blocks) in (("mlp1", 500, 1280, 5120, 6, 512), ("mlp1_K2560", 500, 2560, 5120, 6, 512), ("qkv", 500, 1280, 3840, 1, 480),
                                               ("mlp1_128x80", 500, 1280, 5120, 7, 256), ("sq4096_128x128", 4096, 4096, 4096, 3, 1024)):
            gemm(ops, tag, T, cin, n, tiles=(tile,), splits=(1,))
            torch.cuda.synchronize()
            cal = (ctypes.c_ulonglong * (16 * 8192 + 2))()
            assert lib.svcmi_probe_ktrace_read(cal, 16 * 8192 + 2) == 0
            print(f"  s_memtime {cal[16 * 8192]} ticks in {cal[16 * 8192 + 1] / 100.0:.1f} us of s_memrealtime -> {cal[16 * 8192] / max(cal[16 * 8192 + 1], 1) / 10.0:.3f} GHz")
            buf = (ctypes.c_ulonglong * (16 * blocks))()
            assert lib.svcmi_probe_ktrace_read(buf, 16 * blocks) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(blocks, 4, 4).astype(np.float64)
            nst = a[..., 3]
            ok = nst > 0
            step, vm, bar = a[..., 0][ok] / nst[ok], a[..., 1][ok] / (nst[ok] + 1), a[..., 2][ok] / (nst[ok] + 1)
            print(f"  ktrace {tag}: K-steps per wave {nst[ok].mean():.0f}; cycles per K-step {step.mean():.0f} (min {step.min():.0f}, max {step.max():.0f});"
                  f" in the DMA wait {vm.mean():.0f} (max over waves {vm.max():.0f}); in the barrier {bar.mean():.0f} (min {bar.min():.0f}, max {bar.max():.0f})", flush=True)
            for w in range(4):
                okw = ok[:, w]
                print(f"    wave {w}: step {(a[:, w, 0][okw] / nst[:, w][okw]).mean():.0f}  vmwait {(a[:, w, 1][okw] / (nst[:, w][okw] + 1)).mean():.0f}  barrier {(a[:, w, 2][okw] / (nst[:, w][okw] + 1)).mean():.0f}")
    if "w8" in what:          # the 64x80 wave tile on four-wave (tile 6) and eight-wave blocks (tile 10), both ring depths (| 16 = SVCMI_CONV_RING2)
        for T in (500, 750):
            gemm(ops, "whisper_mlp1", T, 1280, 5120, tiles=(6, 10, 6 | 16, 10 | 16), splits=(1,))
            gemm(ops, "whisper_qkv", T, 1280, 3840, tiles=(1, 6, 10), splits=(1,))
            gemm(ops, "whisper_o", T, 1280, 1280, tiles=(6, 10), splits=(2, 4), partials=True)
            gemm(ops, "whisper_mlp2", T, 5120, 1280, tiles=(6, 10), splits=(4, 8), partials=True)
        gemm(ops, "mlp1_M4096", 4096, 1280, 5120, tiles=(6, 10, 3), splits=(1,))
    if "kprobe" in what:      # K-loop timing probes (build_variant.sh -DSVCMI_PROBE_*): the same launches on every variant library
        for cin in (640, 1280, 2560):
            gemm(ops, "mlp1_K", 500, cin, 5120, tiles=(6, 7, 3), splits=(1,))
        for cin in (640, 1280, 2560):
            gemm(ops, "qkv_K", 500, cin, 3840, tiles=(1, 2, 3), splits=(1,))
        gemm(ops, "square4096", 4096, 4096, 4096, tiles=(3, 6), splits=(1,))
    if "kscale" in what:      # fixed overhead vs per-K-step cost of the Whisper tiles: time against K at constant M x N
        for cin in (320, 640, 1280, 2560, 5120):
            gemm(ops, "mlp1_K", 500, cin, 5120, tiles=(6,), splits=(1,))
        for cin in (320, 640, 1280, 2560, 5120):
            gemm(ops, "qkv_K", 500, cin, 3840, tiles=(1,), splits=(1,))
        for T in (256, 512, 1024):
            gemm(ops, "mlp1_M", T, 1280, 5120, tiles=(6,), splits=(1,))
    if "group" in what:       # the grouped AMP-stage GEMMs: tile policy x ring depth
        for (C, n, tls) in ((160, 5000, (0, 1, 7)), (80, 20000, (0, 7)), (40, 80000, (0, 5))):
            g = torch.Generator().manual_seed(0)
            xs = [torch.randn(1, n, C, generator=g).cuda() for _ in range(3)]
            rs = [torch.randn(1, n, C, generator=g).cuda() for _ in range(3)]
            outs = [torch.empty(1, n, C, device="cuda") for _ in range(3)]
            ws = [PW.pack_conv(torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).cuda() for k in (3, 7, 11)]
            bs = [torch.randn(C, generator=g).cuda() for _ in range(3)]
            fl = sum(2.0 * n * C * C * k for k in (3, 7, 11))
            for d in (1, 5):
                for tile in tls:
                    for nst in (3, 2):
                        assert ops.lib.svcmi_tune_set(b"group_nst", nst) == 0
                        probs = [dict(x=xs[j], w=ws[j], bias=bs[j], ksize=k, dilation=d, pad=(k - 1) * d // 2, res=rs[j], out=outs[j],
                                      tile=tile) for j, k in enumerate((3, 7, 11))]
                        us = timeit(lambda: ops.conv_group(probs))
                        print(f"group C={C} n={n} d={d} tile={tile} nst={nst}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
        ops.lib.svcmi_tune_set(b"group_nst", 0)
    if "lpgroup" in what:     # grouped AMP-stage GEMMs with 16-bit operands: 64-row vs 128-row right-sized tiles, batch 1 and 16
        for B in (1, 16):
            # (128-row tiles 7 / 5 were tried for the 16-bit kernels too and lose everywhere, e.g. bf16, B = 16: 598 vs 549 us at 80
            # channels, 1268 vs 875 us at 40 -- these launches move 0.9-1.8 GB of activations and sit at ~2 TB/s)
            for (C, n, tls) in ((160, 5000, (0,)), (80, 20000, (6,)), (40, 80000, (4,))):
                g = torch.Generator().manual_seed(0)
                xs = [torch.randn(B, n, C, generator=g).cuda() for _ in range(3)]
                rs = [torch.randn(B, n, C, generator=g).cuda() for _ in range(3)]
                outs = [torch.empty(B, n, C, device="cuda") for _ in range(3)]
                ws = [PW.pack_conv(torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).cuda() for k in (3, 7, 11)]
                bs = [torch.randn(C, generator=g).cuda() for _ in range(3)]
                fl = sum(2.0 * B * n * C * C * k for k in (3, 7, 11))
                for prec in (None, "bf16x3", "bf16"):
                    for tile in tls:
                        probs = [dict(x=xs[j], w=ws[j], bias=bs[j], ksize=k, dilation=3, pad=(k - 1) * 3 // 2, res=rs[j], out=outs[j],
                                      tile=tile) for j, k in enumerate((3, 7, 11))]
                        with ops.use_precision(prec):
                            us = timeit(lambda: ops.conv_group(probs), iters=10, warm=2)
                        print(f"lpgroup B={B} C={C} n={n} {prec or 'f32':6s} tile={tile}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
    if "ampgroup" in what:    # grouped fused SnakeAlias+conv half-steps of the narrow stages: time steps per thread
        filt = torch.tensor([0.00202896, 0.00938947, -0.02554346, -0.05765738, 0.12857258, 0.44320980, 0.44320980,
                             0.12857258, -0.05765738, -0.02554346, 0.00938947, 0.00202896], device="cuda")
        for (C, n) in ((20, 160000), (10, 320000)):
            ld = (C + 3) // 4 * 4
            probs = []
            for k in (3, 7, 11):
                probs.append(dict(x=torch.randn(1, n, ld, device="cuda"), alpha_log=torch.randn(ld, device="cuda") * 0.3,
                                  beta_log=torch.randn(ld, device="cuda") * 0.3, w=PW.pack_conv(torch.randn(C, C, k) / math.sqrt(C * k), ld, ld).cuda(),
                                  bias=torch.randn(ld, device="cuda"), ksize=k, dilation=1, res=torch.randn(1, n, ld, device="cuda"),
                                  out=torch.empty(1, n, ld, device="cuda")))
            fl = sum(2.0 * n * C * C * k for k in (3, 7, 11))
            for d in (1, 5):
                for pr in probs:
                    pr["dilation"] = d
                for tt in (1, 2, 4):
                    assert ops.lib.svcmi_tune_set(b"amp_tt", tt) == 0
                    us = timeit(lambda: ops.snake_conv_group(probs, filt, c=C))
                    print(f"ampgroup C={C} n={n} d={d} tt={tt}: {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s", flush=True)
                ops.lib.svcmi_tune_set(b"amp_tt", 0)
                assert ops.lib.svcmi_tune_set(b"amp_u", 1) == 0      # up-sampled activation tile in LDS (default tile of the width)
                for B in (1, 4):
                    pb = [dict(pr, x=pr["x"].expand(B, -1, -1).contiguous(), res=pr["res"].expand(B, -1, -1).contiguous(),
                               out=torch.empty(B, n, ld, device="cuda")) for pr in probs]
                    for u in (-1, 1):
                        ops.lib.svcmi_tune_set(b"amp_u", u)
                        us = timeit(lambda: ops.snake_conv_group(pb, filt, c=C))
                        print(f"ampgroup C={C} n={n} d={d} B={B} amp_u={u}: {us:8.1f} us  {B * fl / us / 1e6:6.1f} TF/s", flush=True)
                ops.lib.svcmi_tune_set(b"amp_u", 0)
        ops.lib.svcmi_tune_set(b"amp_tt", 0)
    if "amplp" in what:       # the grouped half-step: fp32 vector kernel vs the fp16 matrix-core variants, both activation-phase forms
        filt = torch.tensor([0.00202896, 0.00938947, -0.02554346, -0.05765738, 0.12857258, 0.44320980, 0.44320980,
                             0.12857258, -0.05765738, -0.02554346, 0.00938947, 0.00202896], device="cuda")
        for (C, n) in ((20, 160000), (10, 320000)):
            ld = (C + 3) // 4 * 4
            fl = sum(2.0 * n * C * C * k for k in (3, 7, 11))
            for B in (1, 4):
                probs = [dict(x=torch.randn(B, n, ld, device="cuda"), alpha_log=torch.randn(ld, device="cuda") * 0.3,
                              beta_log=torch.randn(ld, device="cuda") * 0.3, w=PW.pack_conv(torch.randn(C, C, k) / math.sqrt(C * k), ld, ld).cuda(),
                              bias=torch.randn(ld, device="cuda"), ksize=k, dilation=1, res=torch.randn(B, n, ld, device="cuda"),
                              out=torch.empty(B, n, ld, device="cuda")) for k in (3, 7, 11)]
                for d in (1, 5):
                    for pr in probs:
                        pr["dilation"] = d
                    for u in (-1, 1):
                        ops.lib.svcmi_tune_set(b"amp_u", u)
                        ref = None
                        for name, prec, mfma in (("fp32 VALU", None, 0), ("fp32 MFMA", None, 2 if u > 0 else 3), ("f16", "f16", 0), ("f16w2", "f16w2", 0)):
                            ops.lib.svcmi_tune_set(b"amp_mfma", mfma)
                            us = timeit(lambda: ops.snake_conv_group(probs, filt, c=C, precision=prec))
                            out = probs[2]["out"][..., :C].clone()
                            ref = out if ref is None else ref
                            print(f"amplp C={C} n={n} B={B} d={d} amp_u={u:2d} {name:9s}: {us:8.1f} us  {B * fl / us / 1e6:6.1f} TF/s  "
                                  f"max diff to fp32 VALU {float((out - ref).abs().max()):.1e}", flush=True)
                        ops.lib.svcmi_tune_set(b"amp_mfma", 1)
                ops.lib.svcmi_tune_set(b"amp_u", 0)
    if "small" in what:       # the short-K / few-tile GEMMs of the prior encoder, flow and widest decoder stage
        sp = (1, 0, 2, 3, 4)
        for k, d in ((3, 1), (7, 3), (11, 5)):
            gemm(ops, "stage0_C160", 5000, 160, 160, k=k, dil=d, res=True, tiles=(0,), splits=sp)
        for k, d in ((3, 1), (11, 5)):
            gemm(ops, "stage1_C80", 20000, 80, 80, k=k, dil=d, res=True, tiles=(0,), splits=(1, 0))
        gemm(ops, "flow_in", 1000, 192, 384, k=5, tiles=(0,), splits=sp)
        gemm(ops, "flow_rs", 1000, 192, 384, k=1, tiles=(0,), splits=sp)
        gemm(ops, "encp_ffn1", 1000, 192, 640, k=3, tiles=(0,), splits=sp)
        gemm(ops, "encp_ffn2", 1000, 640, 192, k=3, tiles=(0,), splits=sp)
        gemm(ops, "encp_qkv", 1000, 192, 576, k=1, tiles=(0,), splits=sp)
        gemm(ops, "encp_pre", 1000, 1280, 192, k=5, tiles=(0,), splits=(1, 0, 4, 8))
        gemm(ops, "dec_pre", 1000, 192, 320, k=7, tiles=(0,), splits=sp)
        gemm(ops, "up0", 1000, 320, 800, k=3, tiles=(0,), splits=sp)
    if "wtune" in what:       # Whisper window GEMMs as the encoder launches them (raw split-K slabs for the two N = n_state projections)
        for T in (500, 750):
            gemm(ops, "whisper_qkv", T, 1280, 3840, tiles=(1, 6, 2), splits=(1,))
            gemm(ops, "whisper_mlp1", T, 1280, 5120, tiles=(1, 6, 7, 2), splits=(1,))
            gemm(ops, "whisper_o", T, 1280, 1280, tiles=(1, 6), splits=(1, 2, 3, 4), partials=True)
            gemm(ops, "whisper_mlp2", T, 5120, 1280, tiles=(1, 6), splits=(2, 4, 6, 8), partials=True)
    if "wp16" in what:        # Whisper shapes on the 16x16x4 tiles (128x80 gives exactly 256 blocks for N = 5120)
        gemm(ops, "whisper_qkv", 500, 1280, 3840, tiles=(1, 6, 7), splits=(1,))
        gemm(ops, "whisper_o", 500, 1280, 1280, res=True, tiles=(1, 6, 7), splits=(4, 2, 3))
        gemm(ops, "whisper_mlp1", 500, 1280, 5120, tiles=(1, 6, 7), splits=(1,))
        gemm(ops, "whisper_mlp2", 500, 5120, 1280, res=True, tiles=(1, 6, 7), splits=(8, 4))
    if "p16" in what:         # 64-multiple tiles vs right-sized 16x16x4 tiles for the generator's mid stages
        for (C, n, tl) in ((160, 5000, (0, 1, 8)), (80, 20000, (1, 6, 7)), (40, 80000, (1, 4, 5))):
            for k, d in ((3, 1), (7, 3), (11, 5), (11, 1)):
                gemm(ops, f"amp_C{C}", n, C, C, k=k, dil=d, res=True, tiles=tl, splits=(0,))
    if "dec" in what:
        for (C, n) in ((160, 5000), (80, 20000), (40, 80000), (20, 160000), (10, 320000)):
            for k, d in ((3, 1), (7, 3), (11, 5), (11, 1)):
                gemm(ops, f"amp_C{C}", n, (C + 3) // 4 * 4, C, k=k, dil=d, res=True, tiles=(0,), splits=(0,))
    if "snake" in what:
        filt = torch.tensor([0.00202896, 0.00938947, -0.02554346, -0.05765738, 0.12857258, 0.44320980, 0.44320980,
                             0.12857258, -0.05765738, -0.02554346, 0.00938947, 0.00202896], device="cuda")
        for (C, n) in ((160, 5000), (80, 20000), (40, 80000), (20, 160000), (10, 320000)):
            ld = (C + 3) // 4 * 4
            x = torch.randn(1, n, ld, device="cuda")
            al, be = torch.randn(ld, device="cuda") * 0.3, torch.randn(ld, device="cuda") * 0.3
            y = torch.empty_like(x)
            us = timeit(lambda: ops.snake_alias(x, al, be, filt, out=y))
            print(f"snake C={C} n={n}: {us:8.1f} us  {8.0 * C * n / us / 1e3:7.1f} GB/s", flush=True)
    if "amp" in what:
        filt = torch.tensor([0.00202896, 0.00938947, -0.02554346, -0.05765738, 0.12857258, 0.44320980, 0.44320980,
                             0.12857258, -0.05765738, -0.02554346, 0.00938947, 0.00202896], device="cuda")
        for tt, (C, n) in [(t, cn) for t in (1, 2, 4) for cn in ((40, 80000), (20, 160000), (10, 320000))]:
            assert ops.lib.svcmi_tune_set(b"amp_tt", tt) == 0
            ld = (C + 3) // 4 * 4
            x = torch.randn(1, n, ld, device="cuda")
            r = torch.randn(1, n, ld, device="cuda")
            al, be = torch.randn(ld, device="cuda") * 0.3, torch.randn(ld, device="cuda") * 0.3
            y = torch.empty_like(x)
            for k, d in ((3, 1), (7, 3), (11, 5), (11, 1)):
                w = PW.pack_conv(torch.randn(C, C, k) / math.sqrt(C * k), ld, ld).cuda()
                bias = torch.randn(ld, device="cuda")
                us = timeit(lambda: ops.snake_conv(x, al, be, filt, w, bias, c=C, ksize=k, dilation=d, res=r, out=y))
                print(f"amp tt={tt} C={C} n={n} k={k} d={d}: {us:8.1f} us  conv {2.0 * C * C * k * n / us / 1e6:6.1f} TF/s", flush=True)
        ops.lib.svcmi_tune_set(b"amp_tt", 0)
    if "attnlds" in what:     # every compiled shape of the LDS-staged kernel (knob = 10 * QT + KS) against the heuristic's choice (0): alone and 4 streams
        for (B, T, H, D) in ((1, 500, 20, 64), (2, 500, 20, 64), (4, 500, 20, 64), (16, 500, 20, 64), (1, 750, 20, 64), (2, 750, 20, 64), (1, 1500, 20, 64)):
            qkv = torch.randn(B, T, 3 * H * D, device="cuda")
            out = torch.empty(B, T, H * D, device="cuda")
            ref = None
            for code in (-1, 0, 21, 22, 24, 41, 42, 44, 81, 82):      # -1: the register-fed kernels, 0: the library heuristic
                assert ops.lib.svcmi_tune_set(b"attn_lds", code) == 0
                us = timeit(lambda: ops.attention(qkv, H, D ** -0.5, out=out))
                ref = out.clone() if ref is None else ref
                print(f"attnlds B={B} T={T} code={code:2d}: {us:8.1f} us  {4.0 * B * T * T * H * D / us / 1e6:7.1f} TF/s  "
                      f"maxdiff {float((out - ref).abs().max()):.1e}", flush=True)
        # four independent T = 500 windows on four streams (the clips-in-flight regime of bench.py)
        streams = [torch.cuda.Stream() for _ in range(4)]
        qs = [torch.randn(1, 500, 3 * 20 * 64, device="cuda") for _ in range(4)]
        os_ = [torch.empty(1, 500, 20 * 64, device="cuda") for _ in range(4)]
        for code in (-1, 22, 24, 41, 42, 44, 81, 82):
            assert ops.lib.svcmi_tune_set(b"attn_lds", code) == 0
            def run4():
                for i, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        for _ in range(20):
                            ops.attention(qs[i], 20, 0.125, out=os_[i])
            run4()
            torch.cuda.synchronize()
            import time as _t
            t0 = _t.perf_counter()
            run4()
            torch.cuda.synchronize()
            us = (_t.perf_counter() - t0) * 1e6 / 80
            print(f"attnlds 4 streams x T=500 code={code:2d}: {us:8.1f} us per launch  {4.0 * 500 * 500 * 1280 / us / 1e6:7.1f} TF/s", flush=True)
        ops.lib.svcmi_tune_set(b"attn_lds", 0)
    if "attn16" in what:      # attention on the 16-bit matrix cores: block shapes (10 * QT + KS) against the fp32 kernels' choice
        for (B, T, H, D) in ((1, 500, 20, 64), (2, 500, 20, 64), (4, 500, 20, 64), (16, 500, 20, 64), (1, 750, 20, 64), (2, 750, 20, 64), (1, 1500, 20, 64)):
            qkv = torch.randn(B, T, 3 * H * D, device="cuda")
            q16 = qkv.half()
            out = torch.empty(B, T, H * D, device="cuda")
            us = timeit(lambda: ops.attention(qkv, H, D ** -0.5, out=out))
            print(f"attn16 B={B} T={T} fp32 kernels: {us:8.1f} us  {4.0 * B * T * T * H * D / us / 1e6:7.1f} TF/s", flush=True)
            for code in (0, 41, 42, 44, 81, 82):
                assert ops.lib.svcmi_tune_set(b"attn16", code) == 0
                us = timeit(lambda: ops.attention16(q16, H, D ** -0.5))
                o, _ = ops.attention16(q16, H, D ** -0.5)
                print(f"attn16 B={B} T={T} f16 shape={code:2d}: {us:8.1f} us  {4.0 * B * T * T * H * D / us / 1e6:7.1f} TF/s  maxdiff vs fp32 {float((o - out).abs().max()):.1e}", flush=True)
        for (B, T, H, D) in ((1, 1000, 2, 96), (16, 1000, 2, 96), (1, 2510, 2, 96), (1, 510, 2, 96)):       # the prior encoder's rel-pos attention
            qkv = torch.randn(B, T, 3 * H * D, device="cuda")
            q16 = qkv.half()
            rk, rv = torch.randn(9, D, device="cuda") * 0.1, torch.randn(9, D, device="cuda") * 0.1
            out = torch.empty(B, T, H * D, device="cuda")
            us = timeit(lambda: ops.attention(qkv, H, D ** -0.5, rel_k=rk, rel_v=rv, window=4, out=out))
            print(f"attn16 rel B={B} T={T} fp32 kernel: {us:8.1f} us  {4.0 * B * T * T * H * D / us / 1e6:7.1f} TF/s", flush=True)
            for code in (0, 14, 21, 24, 42, 44):
                assert ops.lib.svcmi_tune_set(b"attn16", code) == 0
                us = timeit(lambda: ops.attention16(q16, H, D ** -0.5, rel_k=rk, rel_v=rv, window=4))
                o, _ = ops.attention16(q16, H, D ** -0.5, rel_k=rk, rel_v=rv, window=4)
                print(f"attn16 rel B={B} T={T} f16 shape={code:2d}: {us:8.1f} us  {4.0 * B * T * T * H * D / us / 1e6:7.1f} TF/s  maxdiff vs fp32 {float((o - out).abs().max()):.1e}", flush=True)
        ops.lib.svcmi_tune_set(b"attn16", 0)
    if "attn" in what:
        for (T, H, D, rel) in ((500, 20, 64, False), (1000, 2, 96, True), (1500, 20, 64, False), (2520, 2, 96, True)):
            qkv = torch.randn(1, T, 3 * H * D, device="cuda")
            rk = torch.randn(9, D, device="cuda") * 0.1 if rel else None
            rv = torch.randn(9, D, device="cuda") * 0.1 if rel else None
            out = torch.empty(1, T, H * D, device="cuda")
            for (q32, wide) in (((0, 0), (0, 1), (1, 0)) if not rel else ((0, 0), (0, 1))):      # wide: the 64-key-step kernel of round 6
                for ns in (0, 2, 4, 8):
                    assert ops.lib.svcmi_tune_set(b"attn_ns", ns) == 0 and ops.lib.svcmi_tune_set(b"attn_q32", q32) == 0 and ops.lib.svcmi_tune_set(b"attn_wide", wide) == 0
                    us = timeit(lambda: ops.attention(qkv, H, D ** -0.5, rel_k=rk, rel_v=rv, window=4 if rel else 0, out=out))
                    print(f"attn T={T} H={H} D={D} rel={rel} q32={q32} wide={wide} ns={ns}: {us:8.1f} us  {4.0 * T * T * H * D / us / 1e6:7.1f} TF/s", flush=True)
        ops.lib.svcmi_tune_set(b"attn_ns", 0)
        ops.lib.svcmi_tune_set(b"attn_q32", -1)
        ops.lib.svcmi_tune_set(b"attn_wide", -1)
        # the opt-in LDS-staged kernel (K / V tiles shared by 4 query tiles x 2 key ranges per block) against the heuristic's choice,
        # one window alone and chip-filling batches
        for (B, T, H, D) in ((1, 500, 20, 64), (1, 750, 20, 64), (2, 750, 20, 64), (4, 500, 20, 64), (16, 500, 20, 64), (1, 1500, 20, 64)):
            qkv = torch.randn(B, T, 3 * H * D, device="cuda")
            out = torch.empty(B, T, H * D, device="cuda")
            ref = None
            for lds in (0, 1):
                assert ops.lib.svcmi_tune_set(b"attn_lds", lds) == 0
                us = timeit(lambda: ops.attention(qkv, H, D ** -0.5, out=out))
                ref = out.clone() if ref is None else ref
                print(f"attn B={B} T={T} H={H} D={D} lds={lds}: {us:8.1f} us  {4.0 * B * T * T * H * D / us / 1e6:7.1f} TF/s  "
                      f"maxdiff vs heuristic kernel {float((out - ref).abs().max()):.1e}", flush=True)
        ops.lib.svcmi_tune_set(b"attn_lds", 0)


if __name__ == "__main__":
    main()
