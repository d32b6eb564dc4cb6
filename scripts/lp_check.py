#!/usr/bin/env python
"""Mid-barrier K loop of the 16-bit-ACTIVATION GEMM kernels (conv_gemm_body.h, SVCMI_GEMM_MIDBAR16) against the library it replaces:
    python scripts/lp_check.py <other libsvcmi.so>          (on a CPU box: <other emu lib> <new emu lib>, tiny shapes, no timing)
Loads BOTH libraries in one process, runs the same 16-bit convolutions (bf16 / f16 / split-bf16 / f16 with split weights, every tile of
the 16-bit kernels, split-K) through each and reports whether the results are the same bits and what the Whisper window shapes cost."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from svcmi import _lib  # noqa: E402
from svcmi import weights as PW  # noqa: E402
from svcmi.ops import Ops  # noqa: E402

DEV = "cuda" if torch.cuda.is_available() else "cpu"


def timeit(fn, iters=40, warm=6):
    if DEV == "cpu":
        fn()
        return 1.0
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def split16(x):
    c = x.shape[-1]
    cp = (c + 7) // 8 * 8
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    out = torch.zeros(*x.shape[:-1], 2 * cp, dtype=torch.bfloat16, device=x.device)
    out[..., :c], out[..., cp:cp + c] = hi, lo
    return out


def x16_of(x, prec):
    if prec == "bf16x3":
        return split16(x)
    return x.to(torch.float16 if prec in ("f16", "f16w2") else torch.bfloat16)


def main():
    new = Ops() if DEV == "cuda" else Ops(_lib.load_library(sys.argv[2]))
    old = Ops(_lib.load_library(sys.argv[1]))
    new.lp_min_flops = old.lp_min_flops = 0.0
    small = (lambda t: min(t, 70)) if DEV == "cpu" else (lambda t: t)
    g = torch.Generator().manual_seed(13)
    bad = n = 0
    shapes = ((2, 150, 64, 80, 5, 1), (1, 2000, 40, 40, 7, 3), (1, 513, 32, 200, 1, 1), (1, 700, 96, 48, 3, 1), (1, 1300, 64, 160, 1, 1),
              (1, 500, 1280, 640, 1, 1), (3, 333, 192, 384, 5, 1))
    if DEV == "cpu":
        shapes = shapes[:5]
    for prec in ("f16", "bf16", "bf16x3", "f16w2"):
        for tile in (0, 1, 3, 4, 6, 9):
            for (B, T, cin, nn, k, dil) in shapes:
                T = small(T)
                x = torch.randn(B, T, cin, generator=g).to(DEV)
                w = PW.pack_conv(torch.randn(nn, cin, k, generator=g) / math.sqrt(cin * k)).to(DEV)
                b = torch.randn(nn, generator=g).to(DEV)
                r = torch.randn(B, T, nn, generator=g).to(DEV)
                x16 = x16_of(x, prec) if cin % 8 == 0 else None
                for sk in (1, 3):
                    kw = dict(ksize=k, dilation=dil, pad=(k - 1) * dil // 2, res=r, tile=tile, split_k=sk, n_out=nn, x16=x16)
                    try:
                        with new.use_precision(prec):
                            y1 = new.conv(x, w, b, **kw)
                        with old.use_precision(prec):
                            y0 = old.conv(x, w, b, **kw)
                    except Exception:       # noqa: BLE001  (a tile / split the mode does not support: both libraries refuse alike)
                        continue
                    n += 1
                    if not torch.equal(y1, y0):
                        bad += 1
                        print("DIFF", prec, tile, sk, (B, T, cin, nn, k, dil), float((y1 - y0).abs().max()), flush=True)
    print(f"bit comparison: {n} launches, {bad} differ", flush=True)
    rows = (("whisper_qkv", 500, 1280, 3840, 1, False), ("whisper_mlp1", 500, 1280, 5120, 1, False), ("whisper_o", 500, 1280, 1280, 2, True),
            ("whisper_mlp2", 500, 5120, 1280, 4, True), ("square4096", 4096, 4096, 4096, 1, False))
    for prec in (("f16", "bf16x3") if DEV == "cuda" else ("f16",)):
        for (tag, T, cin, nn, sk, partials) in rows:
            T, cin, nn = small(T), (cin if DEV == "cuda" else 64), (nn if DEV == "cuda" else 96)
            x = torch.randn(1, T, cin, generator=g).to(DEV)
            w = PW.pack_conv(torch.randn(nn, cin, 1, generator=g) / math.sqrt(cin)).to(DEV)
            b = torch.randn(nn, generator=g).to(DEV)
            out = torch.empty(1, T, nn, device=DEV)
            x16 = x16_of(x, prec)
            fl = 2.0 * T * nn * cin
            res = {}
            for rep in range(2):
                for name, ops in (("new", new), ("old", old)):
                    def fn(ops=ops):
                        with ops.use_precision(prec):
                            if partials:
                                return ops.conv(x, w, None, ksize=1, split_k=sk, partials=True, x16=x16)
                            return ops.conv(x, w, b, ksize=1, out=out, split_k=sk, n_out=nn, x16=x16)
                    try:
                        res.setdefault(name, []).append(timeit(fn))
                    except Exception as e:       # noqa: BLE001
                        print(f"gemm {tag} {prec}: {name} refused ({e})", flush=True)
            if len(res.get("new", [])) < 1 or len(res.get("old", [])) < 1:
                continue
            t1, t0 = min(res["new"]), min(res["old"])
            print(f"gemm {tag:13s} {prec:6s}+a16 T={T} cin={cin} n={nn} split={sk}: new {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF/s)  old {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF/s)  {t0 / t1:.3f}x", flush=True)


if __name__ == "__main__":
    main()
