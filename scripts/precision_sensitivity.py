"""CPU study for the per-layer mixed-precision policy: which GEMM groups of the synthesizer carry the
16-bit operand-rounding error of the waveform?

The 16-bit kernels round the two MULTIPLICANDS of a GEMM (bf16 / f16) and accumulate in fp32; products of two 16-bit values are
exact in fp32.  That is emulated here on the oracle (TEST INFRASTRUCTURE, torch CPU): F.conv1d / F.conv_transpose1d / F.linear are
wrapped, the weight tensor is looked up by name in the state dict (folded weight-norm tensors are tagged by fold_weight_norm), and
the operands of the groups a policy selects are rounded before the fp32 op.  Output: waveform max-abs error vs the fp32 oracle
for (a) everything in the mode, (b) all-but-one group, (c) only one group.

    python scripts/precision_sensitivity.py [--T 400] [--mode f16] [--stress]
"""
import argparse
import os
import re
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import svc_oracle as O                      # noqa: E402
from workload import config as C, inputs as I, weights as W      # noqa: E402

GROUPS = [
    ("enc.pre_hub", r"^enc_p\.(pre|hub)\."),
    ("enc.attn_proj", r"^enc_p\.enc\.attn_layers\.\d+\.conv_[qkvo]\."),
    ("enc.ffn", r"^enc_p\.enc\.ffn_layers\."),
    ("enc.proj", r"^enc_p\.proj\."),
    ("flow.pre", r"^flow\.flows\.\d+\.pre\."),
    ("flow.in", r"^flow\.flows\.\d+\.enc\.in_layers\."),
    ("flow.rs", r"^flow\.flows\.\d+\.enc\.res_skip_layers\."),
    ("flow.post", r"^flow\.flows\.\d+\.post\."),
    ("dec.conv_pre", r"^dec\.conv_pre\."),
    ("dec.ups0", r"^dec\.ups\.0$"), ("dec.ups1", r"^dec\.ups\.1$"), ("dec.ups2", r"^dec\.ups\.2$"),
    ("dec.amp0", r"^dec\.resblocks\.[012]\."), ("dec.amp1", r"^dec\.resblocks\.[345]\."), ("dec.amp2", r"^dec\.resblocks\.[678]\."),
]

STATE = {"names": {}, "policy": None, "mode": "f16", "hits": {}, "operands": "aw"}
FINE = [
    ("amp0.c1", r"^dec\.resblocks\.[012]\.convs1"), ("amp0.c2", r"^dec\.resblocks\.[012]\.convs2"),
    ("amp1.c1", r"^dec\.resblocks\.[345]\.convs1"), ("amp1.c2", r"^dec\.resblocks\.[345]\.convs2"),
    ("amp2.c1", r"^dec\.resblocks\.[678]\.convs1"), ("amp2.c2", r"^dec\.resblocks\.[678]\.convs2"),
    ("flow", r"^flow\."), ("enc", r"^enc_p\."), ("pre_ups", r"^dec\.(conv_pre|ups)"),
]


def rnd(x, mode):
    if mode == "f16":
        return x.to(torch.float16).to(torch.float32)
    if mode == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    return x


def group_of(name):
    for g, pat in GROUPS:
        if re.search(pat, name):
            return g
    return None


def wrap(fn):
    def inner(x, w, *a, **kw):
        pol = STATE["policy"]
        if pol is not None:
            name = STATE["names"].get(w.data_ptr())
            g = group_of(name) if name else None
            if g is not None and g in pol:
                STATE["hits"][g] = STATE["hits"].get(g, 0) + 1
                ops = STATE["operands"]
                return fn(rnd(x, STATE["mode"]) if "a" in ops else x, rnd(w, STATE["mode"]) if "w" in ops else w, *a, **kw)
        return fn(x, w, *a, **kw)
    return inner


def install(sd):
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() >= 2:
            STATE["names"][v.data_ptr()] = k
    orig_fold = O.fold_weight_norm
    cache = {}

    def fold(sd_, name):
        if name not in cache:
            cache[name] = orig_fold(sd_, name)
            STATE["names"][cache[name].data_ptr()] = name
        return cache[name]
    O.fold_weight_norm = fold

    class FX:
        pass
    fx = FX()
    for k in dir(F):
        setattr(fx, k, getattr(F, k))
    fx.conv1d = wrap(F.conv1d)
    fx.conv_transpose1d = wrap(F.conv_transpose1d)
    fx.linear = wrap(F.linear)
    O.F = fx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--mode", default="f16")
    ap.add_argument("--stress", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fine", action="store_true", help="AMP stages split into convs1 / convs2; flow / enc / pre+ups as one group each")
    ap.add_argument("--operands", default="aw", help="which multiplicands are rounded: aw | a | w")
    ap.add_argument("--policy", default=None, help="comma-separated groups to round (just this one run)")
    a = ap.parse_args()
    torch.set_num_threads(16)
    hp = C.base_hp()
    sd = W.make_vits_state(hp, seed=1234)
    if a.stress:
        sd = W.stress_vits_state(sd, hp)
    install(sd)
    STATE["mode"], STATE["operands"] = a.mode, a.operands
    if a.fine:
        GROUPS[:] = FINE
    d = I.synth_clip(T=a.T, hp=hp, seed=a.seed, B=1)

    def run(policy):
        STATE["policy"], STATE["hits"] = policy, {}
        with torch.no_grad():
            src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
            return O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, d["enc_noise"])

    ref = run(None)
    rms = float(ref.pow(2).mean().sqrt())
    names = [g for g, _ in GROUPS]
    if a.policy:
        pol = set(a.policy.split(","))
        assert pol <= set(names), names
        print(f"mode {a.mode} operands {a.operands} T {a.T} stress {a.stress} rms {rms:.3f} policy {sorted(pol)}: err {float((run(pol) - ref).abs().max()):.3e}")
        return
    full = run(set(names))
    print(f"mode {a.mode}  T {a.T}  stress {a.stress}  wave rms {rms:.3f}  ALL groups: err {float((full - ref).abs().max()):.3e}   hits {STATE['hits']}")
    rows = []
    for g in names:
        only = float((run({g}) - ref).abs().max())
        but = float((run(set(names) - {g}) - ref).abs().max())
        rows.append((g, only, but))
        print(f"  {g:14s} only-this {only:.3e}   all-but-this {but:.3e}", flush=True)
    rows.sort(key=lambda r: -r[1])
    print("ranked by own contribution:", ", ".join(f"{g} {e:.1e}" for g, e, _ in rows))


if __name__ == "__main__":
    main()
