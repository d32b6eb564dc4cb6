#!/usr/bin/env python
"""Pinned K loop (conv_gemm_body.h, SVCMI_GEMM_SPREAD) against the library it replaces, on the GPU box:
    python scripts/spread_check.py <other libsvcmi.so>
Loads BOTH libraries in one process, runs the same fp32 convolutions through each (every tile policy, gather mode, ring depth, split-K,
grouped launches) and reports (a) whether the results are the same bits and (b) the event-timed duration of the judged path's GEMM
shapes under each.  A tuning / validation aid, not the judged bench."""
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


DEV = "cuda" if torch.cuda.is_available() else "cpu"        # cpu: a dry run of this script on the emulator builds (tests/emu), tiny shapes, no timing


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


def main():
    other = sys.argv[1]
    new = Ops() if DEV == "cuda" else Ops(_lib.load_library(sys.argv[2]))
    old = Ops(_lib.load_library(other))
    small = (lambda t: min(t, 70)) if DEV == "cpu" else (lambda t: t)
    g = torch.Generator().manual_seed(11)
    bad = n = 0
    # (a) bits: tiles 0..8 (auto, 64x64, 128x64, 128x128, P16 64x48 / 128x48 / 64x80 / 128x80 / 64x160) x shapes x ring depth x split-K
    shapes = ((2, 150, 64, 80, 5, 1), (1, 2000, 40, 40, 7, 3), (1, 513, 32, 200, 1, 1), (1, 700, 96, 48, 3, 1), (1, 1300, 64, 160, 1, 1),
              (1, 500, 1280, 640, 1, 1), (3, 333, 192, 384, 5, 1), (1, 1030, 4, 32, 7, 1), (1, 4000, 1, 16, 3, 1))
    for tile in range(9):
        for (B, T, cin, nn, k, dil) in shapes:
            T = small(T)
            x = torch.randn(B, T, cin, generator=g).to(DEV)
            w = PW.pack_conv(torch.randn(nn, cin, k, generator=g) / math.sqrt(cin * k)).to(DEV)
            b = torch.randn(nn, generator=g).to(DEV)
            r = torch.randn(B, T, nn, generator=g).to(DEV)
            xx = x.reshape(B, T) if cin == 1 else x
            for extra in (0, 16):
                for sk in (1, 3):
                    kw = dict(ksize=k, dilation=dil, pad=(k - 1) * dil // 2, res=r, tile=tile | extra, split_k=sk, n_out=nn, c_in=cin, ldx=cin)
                    try:
                        y1 = new.conv(xx, w, b, **kw)
                        y0 = old.conv(xx, w, b, **kw)
                    except Exception:       # noqa: BLE001  (a tile the shape does not support: both libraries refuse alike)
                        continue
                    n += 1
                    if not torch.equal(y1, y0):
                        bad += 1
                        print("DIFF", tile, extra, sk, (B, T, cin, nn, k, dil), float((y1 - y0).abs().max()), flush=True)
    for (C, T) in ((160, 5000), (80, 20000), (40, 80000)):
        T = small(T)
        xs = [torch.randn(1, T, C, generator=g).to(DEV) for _ in range(3)]
        rs = [torch.randn(1, T, C, generator=g).to(DEV) for _ in range(3)]
        ws = [PW.pack_conv(torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).to(DEV) for k in (3, 7, 11)]
        bs = [torch.randn(C, generator=g).to(DEV) for _ in range(3)]
        outs = {}
        for name, ops in (("new", new), ("old", old)):
            o = [torch.empty(1, T, C, device=DEV) for _ in range(3)]
            probs = [dict(x=xs[j], w=ws[j], bias=bs[j], ksize=k, dilation=3, pad=(k - 1) * 3 // 2, res=rs[j], out=o[j]) for j, k in enumerate((3, 7, 11))]
            ops.conv_group(probs)
            outs[name] = (o, probs)
        n += 1
        same = all(torch.equal(a, b_) for a, b_ in zip(outs["new"][0], outs["old"][0]))
        bad += 0 if same else 1
        fl = sum(2.0 * T * C * C * k for k in (3, 7, 11))
        t1, t0 = timeit(lambda: new.conv_group(outs["new"][1]), 20, 4), timeit(lambda: old.conv_group(outs["old"][1]), 20, 4)
        print(f"group C={C} n={T}: same bits {same}; new {t1:7.1f} us ({fl / t1 / 1e6:5.1f} TF/s)  old {t0:7.1f} us ({fl / t0 / 1e6:5.1f} TF/s)  {t0 / t1:.3f}x", flush=True)
    print(f"bit comparison: {n} launches, {bad} differ", flush=True)
    # (b) the judged path's Whisper window GEMMs and a chip-filling square, single launches alone on the GPU
    rows = (("whisper_qkv", 500, 1280, 3840, 1, 1, False), ("whisper_mlp1", 500, 1280, 5120, 6, 1, False), ("whisper_o", 500, 1280, 1280, 6, 2, True),
            ("whisper_mlp2", 500, 5120, 1280, 6, 4, True), ("prior_ffn", 1000, 192, 768, 0, 1, False), ("flow_in_k5", 1000, 192, 384, 0, 0, False),
            ("square4096", 4096, 4096, 4096, 3, 1, False), ("mlp1_M4096", 4096, 1280, 5120, 6, 1, False), ("mlp1_M4096_t3", 4096, 1280, 5120, 3, 1, False))
    for (tag, T, cin, nn, tile, sk, partials) in rows:
        T, cin, nn = small(T), (cin if DEV == "cuda" else min(cin, 128)), (nn if DEV == "cuda" else min(nn, 96))
        k = 5 if tag == "flow_in_k5" else 1
        x = torch.randn(1, T, cin, generator=g).to(DEV)
        w = PW.pack_conv(torch.randn(nn, cin, k, generator=g) / math.sqrt(cin * k)).to(DEV)
        b = torch.randn(nn, generator=g).to(DEV)
        out = torch.empty(1, T, nn, device=DEV)
        fl = 2.0 * T * nn * cin * k
        res = {}
        for rep in range(2):
            for name, ops in (("new", new), ("old", old)):
                if partials:
                    fn = lambda ops=ops: ops.conv(x, w, None, ksize=k, pad=(k - 1) // 2, tile=tile, split_k=sk, partials=True)
                else:
                    fn = lambda ops=ops: ops.conv(x, w, b, ksize=k, pad=(k - 1) // 2, out=out, tile=tile, split_k=sk, n_out=nn)
                res.setdefault(name, []).append(timeit(fn))
        t1, t0 = min(res["new"]), min(res["old"])
        print(f"gemm {tag:14s} T={T} cin={cin} n={nn} k={k} tile={tile} split={sk}: new {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF/s)  old {t0:7.1f} us ({fl / t0 / 1e6:6.1f} TF/s)  {t0 / t1:.3f}x", flush=True)


if __name__ == "__main__":
    main()
