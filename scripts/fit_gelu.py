#!/usr/bin/env python
"""Coefficients of svcmi_gelu (csrc/svcmi_rt.h): log2(erfc(t)) ~ t * q(t), q of degree 7 on [0, 4.2], weighted least squares with Lawson
re-weighting towards the minimax fit of erf itself (the weight erfc(t) turns an error of the exponent into an error of erf), then the fp32
evaluation (fused multiply-adds emulated exactly in fp64) against the fp64 function and against torch's fp32 GELU."""
import numpy as np
import torch
from scipy.special import erf, erfc


def fit(deg=8, tmax=4.2, n=40001, iters=60):
    t = np.linspace(1e-6, tmax, n)
    r, w = np.log2(erfc(t)), erfc(t)
    A = np.stack([t ** (k + 1) for k in range(deg)], 1)
    ww, best = w.copy(), None
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(A * ww[:, None], r * ww, rcond=None)
        err = np.abs(1 - np.exp2(A @ c) - erf(t))
        if best is None or err.max() < best[0]:
            best = (err.max(), c.copy())
        ww = ww * (err / err.mean()) ** 0.5
        ww /= ww.max()
    return best


def gelu32(v, c32):
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    v = v.astype(np.float32)
    t = np.minimum(np.abs(v) * np.float32(0.70710678118654752440), np.float32(4.2)).astype(np.float32)
    q = np.full_like(t, c32[-1])
    for k in range(len(c32) - 2, -1, -1):
        q = fma(q, t, np.full_like(t, c32[k]))
    e = np.exp2((q.astype(np.float64) * t.astype(np.float64)).astype(np.float32).astype(np.float64)).astype(np.float32)
    h = (np.float32(0.5) * v * e).astype(np.float32)
    return np.where(v > 0, (v - h).astype(np.float32), h)


if __name__ == "__main__":
    m, c = fit()
    c32 = c.astype(np.float32)
    print("max |erf error| of the fit", m)
    print("coefficients (t^1 .. t^8):", [float(x) for x in c32])
    v = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).normal(size=1000000) * 3]).astype(np.float32)
    ref = 0.5 * v.astype(np.float64) * (1 + erf(v.astype(np.float64) / np.sqrt(2)))
    print("svcmi_gelu vs fp64: max abs", np.abs(gelu32(v, c32) - ref).max())
    print("torch fp32 gelu vs fp64: max abs", np.abs(torch.nn.functional.gelu(torch.from_numpy(v)).numpy() - ref).max())
