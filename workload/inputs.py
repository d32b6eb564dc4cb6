"""Seeded synthetic inputs for the benchmark configurations (workload definition; no model arithmetic).

SURVEY.md section 8d, config 2: mel ~ clamp-normalised log-mel range, vec ~ N(0,1), integer-Hz F0 with
~20 % unvoiced runs, the three stochastic draws of the path supplied explicitly.
"""
import numpy as np
import torch


def synth_f0(T, seed=3, base=220.0):
    """Integer Hz contour (pitch CSVs store ints, pitch/inference.py:110,118) with unvoiced runs."""
    rng = np.random.RandomState(seed)
    t = np.arange(T) / 100.0
    f0 = np.round(base * 2.0 ** (0.5 * np.sin(2 * np.pi * t / 2.0)))
    i = 0
    target = int(0.2 * T)
    zeroed = 0
    while zeroed < target:
        start = rng.randint(0, max(T - 30, 1))
        run = rng.randint(5, 31)
        f0[start:start + run] = 0
        zeroed = int((f0 == 0).sum())
        i += 1
        if i > 1000:
            break
    return torch.from_numpy(f0.astype(np.float32))


def synth_spk(dim=256, seed=7):
    """Stand-in for configs/singers/singer0001.npy (256-d, ||.|| ~ 0.82)."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(dim, generator=g)
    return (v / v.norm() * 0.82).float()


def synth_clip(T=1000, hp=None, seed=0, B=1, ppg=True):
    """All inputs + noise draws for ``B`` clips of ``T`` frames (10 ms each)."""
    from workload import config as C
    hp = hp or C.base_hp()
    g = torch.Generator().manual_seed(1000 + seed)
    hop = int(np.prod(list(hp.gen.upsample_rates)))
    L = T * hop
    n_mel = T  # 100 mel frames/s at 16 kHz hop 160 == 100 fps; encoder output is 50 fps
    d = {
        "mel": (torch.randn(B, 80, n_mel, generator=g) * 0.5).clamp(-1.0, 1.5),
        "mel_noise": torch.randn(B, 80, n_mel, generator=g),
        "vec": torch.randn(B, T, hp.vits.vec_dim, generator=g),
        "pit": torch.stack([synth_f0(T, seed=3 + seed + b, base=180.0 + 40 * b) for b in range(B)]),
        "spk": torch.stack([synth_spk(hp.vits.spk_dim, seed=7 + b) for b in range(B)]),
        "enc_noise": torch.randn(B, hp.vits.inter_channels, T, generator=g),
        "rand_ini": torch.rand(B, 11, generator=g),
        "src_noise": torch.randn(B, L, 11, generator=g),
        "lengths": torch.full((B,), T, dtype=torch.long),
    }
    if ppg:
        d["ppg"] = torch.randn(B, T, hp.vits.ppg_dim, generator=g)
    return d
