"""Drop-in for the reference's pitch/inference.py (``compute_f0_sing``, the CSV pitch format) with the CREPE network on
the svcmi kernels (row N3 of SURVEY.md 8f: the extractor that produces the ``pit`` input of the synthesiser; at 705 GMAC
per 10 s it is the heaviest stage of the reference CLI).

    crepe = load_crepe("crepe/assets/full.pth", "cuda")          # crepe/load.py:23-36 checkpoint (state dict)
    f0 = compute_f0_sing("in.wav", "cuda", model=crepe)          # np.float32 [2 * (1 + n // 320)] Hz, NaN-free
    save_csv_pitch(f0, "in.pit.csv"); load_csv_pitch("in.pit.csv")

On the GPU: framing + per-frame normalisation, six [pad, conv, ReLU, BatchNorm, max-pool] layers (the convolutions are
implicit GEMMs with the frames as the batch; the 512-tap stride-4 first layer runs as a 128-tap stride-1 convolution over
rows of 4 samples), classifier + sigmoid.  On the host, as in the reference (numpy + librosa there): the Viterbi / argmax
decoding of the 360-bin posteriorgram per batch of 512 frames, dithering, the x2 repeat and the mean filter.
The Viterbi routine restates ``librosa.sequence.viterbi`` (not installed here; parity unpinned for that one function).
"""
import numpy as np
import torch

from .. import weights as PW
from .._lib import PREC_BF16X3
from ..ops import ACT_RELU, ACT_SIGMOID, Ops

CENTS_PER_BIN, PITCH_BINS, SAMPLE_RATE, WINDOW_SIZE = 20, 360, 16000, 1024
FRAME_LD = 1536          # 254 + 1024 + 258 floats: 384 rows of 4 samples
NET_BATCH = 512          # frames per pass through the network in compute_f0_sing


class Crepe:
    """crepe.Crepe in eval mode (crepe/model.py:14-134): ``probabilities(audio) -> [frames, 360]``."""

    def __init__(self, state_dict, device, ops=None):
        self.ops = ops if ops is not None else Ops()
        self.device = torch.device(device)
        self.w = PW.CrepeWeights(state_dict, self.device)
        self.precision = None        # GEMM operand precision (Ops.use_precision): None = fp32

    @torch.no_grad()
    def probabilities(self, audio, hop=320, batch_size=512):
        """audio [n] float @16 kHz -> sigmoid outputs [1 + n // hop, 360] (device tensor), batches of ``batch_size`` frames."""
        with self.ops.use_precision(self.precision):
            return self._probabilities(audio, hop, batch_size)

    def _probabilities(self, audio, hop, batch_size):
        w, ops = self.w, self.ops
        audio = audio.to(self.device, torch.float32).contiguous().view(-1)
        total = 1 + audio.numel() // hop
        out = torch.empty(total, PITCH_BINS, dtype=torch.float32, device=self.device)
        for f0 in range(0, total, batch_size):
            nf = min(batch_size, total - f0)
            x = ops.crepe_frames(audio, hop, f0, nf, FRAME_LD).view(nf, FRAME_LD // 4, 4)
            # bf16 / f16 modes: the pooling kernel also writes the 16-bit rows the next layer's GEMM takes as its A operand (the _A16
            # kernels: nothing rounded in the GEMM's registers, K-steps of 64).  Not in bf16x3: on these long-K layers split rows + the
            # _BF16X3_A16 kernel are slower than the in-register split (layer 2: 3.6 vs 3.4 ms, profiles/r03p_microbench_x3a.log)
            fmt16, x16 = (None if ops.precision == PREC_BF16X3 else ops.act16_format()), None
            for i, L in enumerate(w.layers):
                if i == 0:      # 512 taps, stride 4, pad 254/254 == 128 taps, stride 1 over rows of 4 samples
                    x = ops.conv(x, L["w"], L["b"], ksize=128, pad=0, t_out=256, act=ACT_RELU)
                elif "dense_w" in L and x.shape[1] == L["t_in"]:     # short layers: one dense GEMM with the frames as rows (weights.CrepeWeights)
                    t_in, c_in = x.shape[1], x.shape[2]
                    x = ops.conv(x.view(1, nf, t_in * c_in), L["dense_w"], L["dense_b"], act=ACT_RELU,
                                 x16=None if x16 is None else x16.view(1, nf, t_in * c_in)).view(nf, t_in, -1)
                else:           # 64 taps, pad 31/32 (the right pad is the kernel's out-of-range zero fill)
                    x = ops.conv(x, L["w"], L["b"], ksize=64, pad=31, t_out=x.shape[1], act=ACT_RELU, x16=x16)
                if fmt16 is not None and i + 1 < len(w.layers):
                    x, x16 = ops.bn_maxpool2(x, L["scale"], L["shift"], out16=fmt16)
                else:
                    x = ops.bn_maxpool2(x, L["scale"], L["shift"])
            feat = x.reshape(1, nf, -1)                     # [F, 4, 512] -> 2048 = h * 512 + c, model.py:112
            ops.conv(feat, w.fc_w, w.fc_b, act=ACT_SIGMOID, out=out[f0:f0 + nf].view(1, nf, PITCH_BINS))
        return out


def load_crepe(path, device, ops=None):
    sd = torch.load(path, map_location="cpu") if isinstance(path, str) else path
    return Crepe(sd, device, ops=ops)


# ------------------------------------------------------------------------------------------ host-side decoding
def _frequency_to_bins(f, ceil=False):
    b = (1200.0 * np.log2(np.float32(f) / np.float32(10.0)) - 1997.3794084376191) / CENTS_PER_BIN     # crepe/convert.py:29-50
    return int(np.ceil(b)) if ceil else int(np.floor(b))


def viterbi_path(prob, transition):
    """Most likely state path, ``librosa.sequence.viterbi(prob, transition)`` semantics: prob [S, T] observation
    likelihoods, row-stochastic transition [S, S], uniform initial distribution, log domain."""
    S, T = prob.shape
    tiny = np.finfo(prob.dtype).tiny
    lp, lt = np.log(prob + tiny), np.log(transition + tiny)
    val = lp[:, 0] + np.log(1.0 / S + tiny)
    ptr = np.zeros((T, S), dtype=np.int64)
    cols = np.arange(S)
    for t in range(1, T):
        tr = val[:, None] + lt
        ptr[t] = np.argmax(tr, axis=0)
        val = lp[:, t] + tr[ptr[t], cols]
    path = np.zeros(T, dtype=np.int64)
    path[-1] = int(np.argmax(val))
    for t in range(T - 2, -1, -1):
        path[t] = ptr[t + 1][path[t + 1]]
    return path


_TRANSITION = None


def _transition():
    global _TRANSITION
    if _TRANSITION is None:             # crepe/decode.py:55-60
        xx, yy = np.meshgrid(range(PITCH_BINS), range(PITCH_BINS))
        tr = np.maximum(12 - abs(xx - yy), 0)
        _TRANSITION = tr / tr.sum(axis=1, keepdims=True)
    return _TRANSITION


def transition_band(transition):
    """Half-width of a banded transition matrix (largest |i - j| with a non-zero entry), or 0 when the entries outside that
    band are not all equal -- what ``svcmi_viterbi_decode`` needs to know to take its banded path."""
    tr = np.asarray(transition)
    ii, jj = np.nonzero(tr)
    band = int(np.abs(ii - jj).max())
    off = np.abs(np.subtract.outer(np.arange(tr.shape[0]), np.arange(tr.shape[1]))) > band
    return band if (2 * band + 1 < tr.shape[0] and np.all(tr[off] == tr[0, -1])) else 0


def bins_to_hz(bins, dither=None):
    """crepe/convert.py:13-34,58-64: bins -> cents (+ triangular dither) -> Hz."""
    bins = torch.as_tensor(bins).cpu().long()
    if dither is None:      # convert.py:58-64 draws scipy.stats.triang(c=0.5, loc=-C, scale=2C): the symmetric triangular law on [-C, C]
        dither = np.random.triangular(-CENTS_PER_BIN, 0.0, CENTS_PER_BIN, size=tuple(bins.shape))
    cents = CENTS_PER_BIN * bins + 1997.3794084376191
    cents = cents + cents.new_tensor(np.asarray(dither))
    return 10 * 2 ** (cents / 1200)


def decode(prob, fmin=50.0, fmax=1000.0, decoder="viterbi", dither=None):
    """One batch of posteriors [F, 360] (CPU) -> Hz [F]: crepe/core.py:592-603, decode.py, convert.py.
    ``dither`` [F] cents; None draws the reference's triangular noise (scipy.stats.triang, convert.py:58-64)."""
    p = prob.t().clone()
    p[:_frequency_to_bins(fmin)] = -float("inf")
    p[_frequency_to_bins(fmax, ceil=True):] = -float("inf")
    if decoder == "argmax":
        bins = p.argmax(dim=0).numpy()
    elif decoder == "viterbi":
        bins = viterbi_path(torch.softmax(p, dim=0).numpy(), _transition())
    else:
        raise ValueError(decoder)
    if dither is None:
        dither = np.random.triangular(-CENTS_PER_BIN, 0.0, CENTS_PER_BIN, size=bins.shape)
    cents = CENTS_PER_BIN * torch.from_numpy(bins) + 1997.3794084376191
    cents = cents + cents.new_tensor(np.asarray(dither))
    return 10 * 2 ** (cents / 1200)


@torch.no_grad()
def compute_f0_sing(filename, device, model=None, noise=None, dither=None, decoder="viterbi"):
    """pitch/inference.py:74-99.  ``filename``: wav path or a 16 kHz float waveform [n]; ``model``: a ``Crepe`` (the
    reference loads crepe/assets/full.pth on first use).  ``noise`` [n] ~ N(0,1) pins the 1e-3 input noise (:77),
    ``dither`` [frames] pins convert.py:58-64.  Returns np.float32 Hz [2 * (1 + n // 320)]."""
    return compute_f0_sing_begin(filename, device, model=model, noise=noise, decoder=decoder)(dither)


@torch.no_grad()
def compute_f0_sing_begin(filename, device, model=None, noise=None, decoder="viterbi"):
    """The device half of ``compute_f0_sing``: everything up to the decoded bins is ENQUEUED on the current stream and nothing waits for
    it; returns ``finish(dither=None) -> np.float32 Hz`` which does the device -> host copy and the host-side tail.  Lets a caller put
    other work (the other two extractors, svc_inference.extract_features) in flight before it blocks on the F0 track."""
    if model is None:
        raise ValueError("pass model=load_crepe(<crepe full.pth>, device)")
    if isinstance(filename, str):
        from ..whisper.audio import load_audio
        audio = torch.from_numpy(load_audio(filename))
    else:
        audio = torch.as_tensor(filename, dtype=torch.float32)
    # the 1e-3 input noise (:77) is drawn / added on the model's device: two elementwise passes over the waveform on the host cost
    # more than the whole network on a many-core box (torch CPU ops wake a thread pool per call)
    audio = audio.to(model.device)
    nz = torch.randn_like(audio) if noise is None else torch.as_tensor(noise, dtype=torch.float32).to(model.device)
    audio = audio + nz * 0.001
    # (network batch: frames are independent, so its size only shapes the launches -- the reference's 512 is a memory bound; the Viterbi
    #  decoder below restarts every 512 frames like crepe/core.py:683-686 whatever this is)
    prob = model.probabilities(audio, hop=320, batch_size=NET_BATCH)
    on_device = decoder == "viterbi" and model.ops.on_gpu
    if on_device:      # the DP on the device (one block per 512-frame decoding batch)
        lt, band = _viterbi_constants(prob.device)
        bins = model.ops.viterbi_decode(prob, lt, 512, _frequency_to_bins(50.0), _frequency_to_bins(1000.0, ceil=True), band=band)

    def finish(dither=None):
        if on_device:
            pitch = bins_to_hz(bins, dither)[None].float()
        else:
            p, out = prob.cpu(), []
            for i in range(0, p.shape[0], 512):         # crepe/core.py:683-686: decoding restarts with every batch
                d = None if dither is None else np.asarray(dither)[i:i + 512]
                out.append(decode(p[i:i + 512], 50.0, 1000.0, decoder, d))
            pitch = torch.cat(out)[None].float()
        pitch = np.repeat(pitch.numpy(), 2, -1)                           # 320 -> 160 * 2 (:95)
        return _mean_filter_np(pitch[0], 5)
    return finish


_VITERBI_CONSTANTS = {}


def _viterbi_constants(device):
    """log(transition + tiny) on the device and the band of the matrix, made once per device."""
    key = str(device)
    if key not in _VITERBI_CONSTANTS:
        tr = _transition()
        _VITERBI_CONSTANTS[key] = (torch.from_numpy(np.log(tr + np.finfo(np.float32).tiny)).to(device), transition_band(tr))
    return _VITERBI_CONSTANTS[key]


def _mean_filter_np(x, win_length):
    """crepe/filter.py:10-57 (NaN-aware moving average; exact zeros become NaN) for one fp32 track in numpy (the track is a few thousand values: torch's CPU convolution spends its time
    waking a thread pool on a 128-core host)."""
    x = np.asarray(x, dtype=np.float32)
    mask = ~np.isnan(x)
    ones = np.ones(win_length, dtype=np.float32)
    # 'full' + slice keeps len(x) outputs also when len(x) < win_length (mode="same" returns max(len(x), win_length) values;
    # the reference's F.conv1d(padding=win_length // 2) keeps the length, crepe/filter.py:10-37)
    lo = win_length // 2
    s = np.convolve(np.where(mask, x, np.float32(0)), ones, mode="full")[lo:lo + x.shape[0]]
    c = np.maximum(np.convolve(mask.astype(np.float32), ones, mode="full")[lo:lo + x.shape[0]], np.float32(1))
    out = (s / c).astype(np.float32)
    out[out == 0] = np.nan
    return out


def save_csv_pitch(pitch, path):
    """pitch/inference.py:102-110: one line per 10 ms frame, ``<m>m <s>s <ms>,<int Hz>``."""
    with open(path, "w", encoding="utf-8") as f:
        for i in range(len(pitch)):
            t = i * 10
            minute = t // 60000
            seconds = (t - minute * 60000) // 1000
            millisecond = t % 1000
            print(f"{minute}m {seconds}s {millisecond:3d},{int(pitch[i])}", file=f)


def quantize_pitch_like_csv(pitch):
    """What a ``save_csv_pitch`` -> ``load_csv_pitch`` round trip does to an F0 track, without the file: ``int(p)`` truncation
    per frame (and the reference's ``ValueError`` on NaN, pitch/inference.py:110).  The synthesizer only ever sees F0 that went
    through the CSV (svc_inference.py:150-154,183), so every in-process path must apply this."""
    return [int(p) for p in pitch]


def load_csv_pitch(path):
    """pitch/inference.py:113-119."""
    pitch = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f.readlines():
            pitch.append(int(line.strip().split(",")[-1]))
    return pitch
