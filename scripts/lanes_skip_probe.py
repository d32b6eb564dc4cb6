#!/usr/bin/env python
"""What does a kernel class cost under clips in flight?  Encoder-only lanes (1 / 2 / 4) with the attention or the split-K LayerNorm
launches skipped (outputs left stale: timing only).  Tuning aid of round 2: python scripts/lanes_skip_probe.py
HISTORICAL: it patched the Python launch wrappers of the encoder; since round 3 the encoder is composed by the C++ stage host
(svcmi_whisper_encoder_fwd), so this script no longer skips anything -- kept for the record of profiles/r02z_*."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

from svcmi import Ops, weights as PW  # noqa: E402
from svcmi.lanes import GraphLanes  # noqa: E402
from svcmi.whisper.inference import WhisperEncoderModel  # noqa: E402
from workload import config as C, inputs as I, weights as W  # noqa: E402


def rate(lanes, clips=32):
    for _ in range(2 * len(lanes)):
        lanes.launch()
    lanes.synchronize()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(clips):
        lanes.launch()
    lanes.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / clips * 1e3


def main():
    dev = torch.device("cuda")
    ops = Ops()
    hp = C.base_hp()
    whisper = WhisperEncoderModel(None, dev, ops=ops, packed=PW.WhisperWeights(W.make_whisper_state(C.WHISPER_LARGE_V2), dev))

    def enc_fn(i):
        mel = I.synth_clip(T=1000, hp=hp, seed=100 + i, B=1, ppg=False)["mel"].to(dev)
        return lambda: whisper.encoder(mel, torch.randn_like(mel), 0.1)[:, :500]

    real_attn, real_ln = ops.attention, ops.splitk_layernorm
    cache = {}

    def fake_attn(qkv, heads, scale, **kw):
        key = ("a", tuple(qkv.shape), torch.cuda.current_stream().cuda_stream)
        if key not in cache:
            cache[key] = torch.zeros(qkv.shape[0], qkv.shape[1], qkv.shape[2] // 3, device=qkv.device)
        return cache[key]

    def fake_ln(p, bias, x, g, b, out=None):
        return out if out is not None else x

    for name, a, l in (("all kernels", real_attn, real_ln), ("no attention", fake_attn, real_ln), ("no splitk_layernorm", real_attn, fake_ln),
                       ("neither", fake_attn, fake_ln)):
        ops.attention, ops.splitk_layernorm = a, l
        r = [rate(GraphLanes([enc_fn(i) for i in range(n)])) for n in (1, 2, 4)]
        print(f"encoder, {name:22s}: 1 / 2 / 4 lanes {r[0]:.3f} / {r[1]:.3f} / {r[2]:.3f} ms per clip", flush=True)


def knobs():
    """Attention launch shapes (key-split waves per block, two query tiles per wave) under 1 / 4 lanes."""
    dev = torch.device("cuda")
    ops = Ops()
    hp = C.base_hp()
    whisper = WhisperEncoderModel(None, dev, ops=ops, packed=PW.WhisperWeights(W.make_whisper_state(C.WHISPER_LARGE_V2), dev))

    def enc_fn(i):
        mel = I.synth_clip(T=1000, hp=hp, seed=100 + i, B=1, ppg=False)["mel"].to(dev)
        return lambda: whisper.encoder(mel, torch.randn_like(mel), 0.1)[:, :500]

    for q32 in (0, 1):
        for ns in (0, 1, 2, 4, 8):
            assert ops.lib.svcmi_tune_set(b"attn_q32", q32) == 0 and ops.lib.svcmi_tune_set(b"attn_ns", ns) == 0
            r = [rate(GraphLanes([enc_fn(i) for i in range(n)])) for n in (1, 4)]
            print(f"encoder, attn_q32 {q32} attn_ns {ns}: 1 / 4 lanes {r[0]:.3f} / {r[1]:.3f} ms per clip", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "knobs":
        knobs()
        sys.exit(0)
    main()
