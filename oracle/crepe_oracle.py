"""CPU oracle of the CREPE F0 extractor as the SVC pipeline uses it (TEST INFRASTRUCTURE; row N3 of SURVEY.md 8f).

Restates, with every random draw as an explicit argument:
  * crepe/core.py:626-703  preprocess  (zero pad 512/512, 1024-sample frames at `hop`, per-frame mean / unbiased-std norm)
  * crepe/model.py:102-134 the network (6 x [pad, conv, ReLU, eval BatchNorm, max-pool 2], flatten, Linear, sigmoid)
  * crepe/core.py:567-623  postprocess (bins outside [fmin, fmax] -> -inf) and crepe/decode.py (argmax, viterbi)
  * crepe/convert.py       bins -> cents (+ dither) -> Hz;  crepe/filter.py:10-57 the nan-aware mean filter
  * pitch/inference.py:74-99 compute_f0_sing (noise 1e-3, hop 320, batches of 512 frames, repeat x2, mean-5)
Pinned against the reference package by oracle/make_golden.py, EXCEPT the Viterbi routine: the reference calls
``librosa.sequence.viterbi`` (un-vendored, not installed) -- restated below from its documented algorithm and injected
into the reference when the fixture is made, i.e. not comparable with librosa's own output.  Since round 5 the dynamic programme is
pinned against GROUND TRUTH instead: every path of small trellises enumerated (S <= 5 states, T <= 6 frames, random and banded
transitions, exact ties), tests/test_independent_pins.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

CENTS_PER_BIN, PITCH_BINS, SAMPLE_RATE, WINDOW_SIZE, MAX_FMAX = 20, 360, 16000, 1024, 2006.0
BN_EPS = 0.0010000000474974513


def preprocess(audio, hop):
    """audio [1, n] @16 kHz -> frames [1 + n // hop, 1024], normalised (core.py:664-703)."""
    total = 1 + int(audio.size(1) // hop)
    x = F.pad(audio, (WINDOW_SIZE // 2, WINDOW_SIZE // 2))
    frames = F.unfold(x[:, None, None, :], kernel_size=(1, WINDOW_SIZE), stride=(1, hop))
    frames = frames.transpose(1, 2).reshape(-1, WINDOW_SIZE)[:total].clone()
    frames -= frames.mean(dim=1, keepdim=True)
    frames /= torch.max(torch.tensor(1e-10), frames.std(dim=1, keepdim=True))
    return frames


def network(sd, frames):
    """frames [F, 1024] -> per-bin probabilities [F, 360] (model.py:102-134)."""
    x = frames[:, None, :, None]
    i = 1
    while f"conv{i}.weight" in sd:
        pad = (0, 0, 254, 254) if i == 1 else (0, 0, 31, 32)
        x = F.conv2d(F.pad(x, pad), sd[f"conv{i}.weight"], sd[f"conv{i}.bias"], stride=(4, 1) if i == 1 else (1, 1))
        x = F.relu(x)
        x = F.batch_norm(x, sd[f"conv{i}_BN.running_mean"], sd[f"conv{i}_BN.running_var"], sd[f"conv{i}_BN.weight"],
                         sd[f"conv{i}_BN.bias"], False, 0.0, BN_EPS)
        x = F.max_pool2d(x, (2, 1), (2, 1))
        i += 1
    x = x.permute(0, 2, 1, 3).reshape(x.shape[0], -1)
    return torch.sigmoid(F.linear(x, sd["classifier.weight"], sd["classifier.bias"]))


def frequency_to_bins(f, ceil=False):
    b = (1200.0 * np.log2(np.float32(f) / np.float32(10.0)) - 1997.3794084376191) / CENTS_PER_BIN
    return int(np.ceil(b)) if ceil else int(np.floor(b))


def viterbi_path(prob, transition):
    """``librosa.sequence.viterbi(prob, transition)`` restated (uniform initial distribution, log domain, ties -> lowest
    state): prob [S, T] column-stochastic observation likelihoods, transition [S, S] row-stochastic."""
    S, T = prob.shape
    eps = np.finfo(prob.dtype).tiny
    lp, lt = np.log(prob + eps), np.log(transition + eps)
    val = np.zeros((T, S))
    ptr = np.zeros((T, S), dtype=np.int64)
    val[0] = lp[:, 0] + np.log(np.full(S, 1.0 / S) + eps)
    for t in range(1, T):
        tr = val[t - 1][:, None] + lt                      # tr[j, k]: from j to k
        ptr[t] = np.argmax(tr, axis=0)
        val[t] = lp[:, t] + tr[ptr[t], np.arange(S)]
    out = np.zeros(T, dtype=np.int64)
    out[-1] = np.argmax(val[-1])
    for t in range(T - 2, -1, -1):
        out[t] = ptr[t + 1][out[t + 1]]
    return out


def transition_matrix():
    xx, yy = np.meshgrid(range(360), range(360))
    tr = np.maximum(12 - abs(xx - yy), 0)
    return tr / tr.sum(axis=1, keepdims=True)


def decode(prob, fmin, fmax, decoder, dither):
    """prob [F, 360] of ONE batch -> Hz [F] (core.py:592-603, decode.py, convert.py).  dither [F] cents (triangular)."""
    p = prob.t().clone()[None]                              # [1, 360, F]
    p[:, :frequency_to_bins(fmin)] = -float("inf")
    p[:, frequency_to_bins(fmax, ceil=True):] = -float("inf")
    if decoder == "argmax":
        bins = p.argmax(dim=1)[0].numpy()
    else:
        seq = torch.softmax(p, dim=1)[0].numpy()            # decode.py:62-63: softmax over the (sigmoid) outputs
        bins = viterbi_path(seq, transition_matrix())
    cents = CENTS_PER_BIN * torch.from_numpy(bins) + 1997.3794084376191
    cents = cents + cents.new_tensor(dither)
    return 10 * 2 ** (cents / 1200)


def mean_filter(signals, win):
    """crepe/filter.py:10-57 (nan-aware moving average; exact zeros become NaN).  signals [1, T]."""
    x = signals.unsqueeze(1)
    mask = ~torch.isnan(x)
    mx = torch.where(mask, x, torch.zeros_like(x))
    ones = torch.ones(1, 1, win)
    s = F.conv1d(mx, ones, padding=win // 2)
    c = F.conv1d(mask.float(), ones, padding=win // 2).clamp(min=1)
    out = s / c
    out[out == 0] = float("nan")
    return out.squeeze(1)


def compute_f0_sing(sd, audio, audio_noise, dither, decoder="viterbi", batch_size=512):
    """pitch/inference.py:74-99 from the loaded 16 kHz waveform [n]: -> Hz [2 * (1 + n // 320)]."""
    a = (audio + audio_noise * 0.001)[None]
    frames = preprocess(a, 320)
    prob = network(sd, frames)
    out = []
    for i in range(0, prob.shape[0], batch_size):           # core.py:683-686: one postprocess (decode) per batch
        out.append(decode(prob[i:i + batch_size], 50.0, 1000.0, decoder, dither[i:i + batch_size]))
    pitch = torch.cat(out)[None].float()
    pitch = torch.from_numpy(np.repeat(pitch.numpy(), 2, -1))
    return mean_filter(pitch, 5).squeeze(0)
