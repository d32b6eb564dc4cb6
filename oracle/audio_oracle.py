"""CPU oracle of the Whisper log-mel front-end (TEST INFRASTRUCTURE; row N2 of SURVEY.md 8f).

``log_mel_spectrogram`` restates whisper/audio.py:68-100 (torch.stft on the CPU).  ``slaney_mel_filterbank`` restates
``librosa.filters.mel(sr=16000, n_fft=400, n_mels=80)`` (whisper/audio.py:65) from librosa's published algorithm --
librosa is an un-vendored dependency that is not installed here, so this one matrix cannot be compared with librosa's own
output; since round 5 it is pinned against an INDEPENDENT implementation of the same published construction,
``transformers.audio_utils.mel_filter_bank(201, 80, 0, 8000, 16000, norm="slaney", mel_scale="slaney")`` (9.2e-10 max-abs,
tests/test_independent_pins.py).  Everything downstream of it is pinned against the reference function by oracle/make_golden.py
(tests/golden/logmel_*.npz).
"""
import math

import numpy as np
import torch

SR, N_FFT, HOP, N_MELS = 16000, 400, 160, 80


def slaney_mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS):
    """Slaney mel scale (linear below 1 kHz, log above), triangular filters, area ('slaney') normalisation."""
    f_sp, brk = 200.0 / 3.0, 1000.0
    brk_mel, step = brk / f_sp, math.log(6.4) / 27.0
    hz2mel = lambda f: brk_mel + math.log(f / brk) / step if f >= brk else f / f_sp
    mel2hz = lambda m: brk * math.exp(step * (m - brk_mel)) if m >= brk_mel else f_sp * m
    lo, hi = hz2mel(0.0), hz2mel(sr / 2.0)
    edges = [mel2hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    freqs = [sr / 2.0 * j / (n_fft // 2) for j in range(n_fft // 2 + 1)]
    fb = np.zeros((n_mels, len(freqs)), dtype=np.float64)
    for i in range(n_mels):
        l, c, r = edges[i], edges[i + 1], edges[i + 2]
        for j, f in enumerate(freqs):
            fb[i, j] = max(0.0, min((f - l) / (c - l), (r - f) / (r - c))) * 2.0 / (r - l)
    return fb.astype(np.float32)


def log_mel_spectrogram(audio, filterbank=None):
    """whisper/audio.py:86-100.  audio: float32 tensor [n] at 16 kHz -> [80, n // 160]."""
    fb = torch.from_numpy(slaney_mel_filterbank()) if filterbank is None else filterbank
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP, window=window, return_complex=True)
    mag = stft[..., :-1].abs() ** 2
    spec = torch.clamp(fb @ mag, min=1e-10).log10()
    spec = torch.maximum(spec, spec.max() - 8.0)
    return (spec + 4.0) / 4.0


def synth_audio(n, seed):
    """Seeded test signal in [-1, 1]: a vibrato tone with harmonics, a burst of noise and a silent gap (so that both
    the clamp at max-8 and ordinary bins are exercised)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / SR
    f0 = 220.0 * 2.0 ** (0.3 * torch.sin(2 * math.pi * 3.0 * t))
    ph = 2 * math.pi * torch.cumsum(f0, 0) / SR
    x = sum((0.5 / (h + 1)) * torch.sin((h + 1) * ph) for h in range(6))
    x = x + 0.05 * torch.randn(n, generator=g, dtype=torch.float64)
    x[n // 3: n // 3 + n // 10] *= 1e-4
    return (0.6 * x).clamp(-1, 1).float()
