"""GPU time of one attention launch, measured inside a captured graph (N launches per replay: no Python / launch-rate floor).

    python scripts/attn_probe.py [whisper|encp|b16 ...]          (SVCMI_LIB=<variant .so> selects a probe build)

Prints, per shape and knob setting (attn_wide / attn_ns), the microseconds per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "whisper-vits-svc_amd"))
from svcmi import Ops  # noqa: E402


def graph_us(fn, n=40, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        s.synchronize()
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            s.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    what = sys.argv[1:] or ["whisper", "encp"]
    ops = Ops()
    tag = os.path.basename(os.environ.get("SVCMI_LIB", "libsvcmi.so"))
    shapes = {"whisper": (1, 500, 20, 64, False), "encp": (1, 1000, 2, 96, True), "b16": (16, 500, 20, 64, False), "b4": (4, 500, 20, 64, False)}
    for name in what:
        B, T, H, D, rel = shapes[name]
        qkv = torch.randn(B, T, 3 * H * D, device="cuda")
        rk = torch.randn(9, D, device="cuda") * 0.1 if rel else None
        rv = torch.randn(9, D, device="cuda") * 0.1 if rel else None
        out = torch.empty(B, T, H * D, device="cuda")
        for wide in (0, 1):
            for ns in (0, 4, 8):
                assert ops.lib.svcmi_tune_set(b"attn_wide", wide) == 0 and ops.lib.svcmi_tune_set(b"attn_ns", ns) == 0
                us = graph_us(lambda: ops.attention(qkv, H, D ** -0.5, rel_k=rk, rel_v=rv, window=4 if rel else 0, out=out))
                print(f"[{tag}] {name} B={B} T={T} H={H} D={D} wide={wide} ns={ns}: {us:7.2f} us  {4.0 * B * T * T * H * D / us / 1e6:6.1f} TF/s", flush=True)
        ops.lib.svcmi_tune_set(b"attn_wide", -1)
        ops.lib.svcmi_tune_set(b"attn_ns", 0)


if __name__ == "__main__":
    main()
