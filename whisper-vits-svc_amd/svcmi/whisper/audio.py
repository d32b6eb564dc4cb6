"""Drop-in for the reference's whisper/audio.py (row N2 of SURVEY.md 8f): 16 kHz loader and the Whisper log-mel
front-end, with the arithmetic on the GPU -- the windowed DFT and the mel projection are launches of the
implicit-GEMM kernel, the rest is csrc/audio_frontend.hip.

    log_mel_spectrogram(audio)  ->  [80, n // 160] float32 device tensor      (whisper/audio.py:68-100)

Pinned vs the reference: STFT framing (n_fft 400, hop 160, periodic hann, center=True / reflect, last frame
dropped), |.|^2, log10 / (max - 8) / (x + 4) / 4 -- checked against the reference function itself
(oracle/make_golden.py, tests/golden/logmel_*.npz).  The filterbank (``librosa.filters.mel``, an un-vendored dependency
that is not installed here; restated below from its published algorithm: Slaney mel scale, Slaney area normalisation) is
pinned against an independent implementation instead (transformers.audio_utils.mel_filter_bank, 9.2e-10:
tests/test_independent_pins.py).  A DOCUMENTED DIVERGENCE remains for non-16 kHz files: ``librosa.load`` resamples with soxr,
``load_audio`` with scipy's polyphase filter -- both band-limited resamplers, not the same taps; the test file checks ours
against analytically sampled tones (16 kHz PCM input, what the reference's own preprocessing writes, is bit-exact).
"""
import math
from functools import lru_cache

import numpy as np
import torch

SAMPLE_RATE = 16000
N_FFT = 400
N_MELS = 80
HOP_LENGTH = 160
NBINS = N_FFT // 2 + 1          # 201
HALF = 204                      # re / im blocks of the DFT output, padded to a multiple of 4


def slaney_mel_filterbank(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=N_MELS):
    """librosa.filters.mel(sr, n_fft, n_mels) defaults (fmin=0, fmax=sr/2, htk=False, norm='slaney') -> float32 [n_mels, 1+n_fft/2]."""
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


@lru_cache(maxsize=None)
def _operands(device):
    """Packed GEMM operands: DFT basis [2*HALF, 400] (rows: hann*cos | -hann*sin, zero rows as padding) and the
    filterbank [80, HALF]."""
    k = np.arange(N_FFT, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / N_FFT)                       # torch.hann_window(400), periodic
    ang = 2.0 * np.pi * np.outer(np.arange(NBINS, dtype=np.float64), k) / N_FFT
    basis = np.zeros((2 * HALF, N_FFT), dtype=np.float32)
    basis[:NBINS] = (win * np.cos(ang)).astype(np.float32)
    basis[HALF:HALF + NBINS] = (-win * np.sin(ang)).astype(np.float32)
    fb = np.zeros((N_MELS, HALF), dtype=np.float32)
    fb[:, :NBINS] = slaney_mel_filterbank()
    return torch.from_numpy(basis).to(device), torch.from_numpy(fb).to(device)


def mel_filters(device, n_mels=N_MELS):
    """whisper/audio.py:53-65."""
    assert n_mels == 80, f"Unsupported n_mels: {n_mels}"
    return torch.from_numpy(slaney_mel_filterbank()).to(device)


def load_audio(file, sr=SAMPLE_RATE):
    """whisper/audio.py:24-26 (librosa.load): mono float32 at ``sr``.  16 kHz PCM wav files are exact; other rates go
    through scipy's polyphase resampler (librosa uses soxr -- not reproducible here)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    rate, x = wavfile.read(file)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim > 1:
        x = x.mean(axis=1)
    if rate != sr:
        g = math.gcd(int(rate), int(sr))
        x = resample_poly(x, sr // g, rate // g).astype(np.float32)
    return x


@torch.no_grad()
def log_mel_spectrogram(audio, n_mels=N_MELS, ops=None, device=None):
    """whisper/audio.py:68-100 on the GPU.  ``audio``: path, numpy array or tensor [n] (or [B, n]) at 16 kHz.
    Returns [80, n // 160] (or [B, 80, n // 160]) on the device."""
    from ..ops import Ops
    assert n_mels == N_MELS
    if isinstance(audio, str):
        audio = load_audio(audio)
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.asarray(audio, dtype=np.float32))
    ops = ops if ops is not None else Ops()
    dev = torch.device(device) if device is not None else (audio.device if audio.is_cuda else torch.device("cuda" if ops.on_gpu else "cpu"))
    x = audio.to(dev, torch.float32)
    batched = x.dim() == 2
    x = (x if batched else x.unsqueeze(0)).contiguous()
    B, n = x.shape
    frames = n // HOP_LENGTH                      # torch.stft gives 1 + n // 160 frames; the last one is dropped (:92)
    if frames < 1:
        raise ValueError("audio shorter than one hop")
    basis, fb = _operands(dev)
    xp = ops.reflect_pad(x, N_FFT // 2)                                                            # [B, n + 400]
    ri = ops.conv(xp, basis, None, ksize=N_FFT, stride=HOP_LENGTH, pad=0, c_in=1, ldx=1, t_in=xp.shape[1],
                  t_out=frames, x_bstride=xp.stride(0))                                            # [B, frames, 408]
    p = ops.power_spectrum(ri, NBINS, HALF)                                                        # [B, frames, 204]
    mel = ops.conv(p, fb, None)                                                                    # [B, frames, 80]
    out = ops.logmel_finish(mel)                                                                   # [B, 80, frames]
    return out if batched else out[0]
