"""Drop-in for the reference's whisper/inference.py: ``load_model`` / ``pred_ppg`` on svcmi kernels.

``load_model(path, device)`` consumes the OpenAI checkpoint format ``{"dims", "model_state_dict"}`` and,
like whisper/inference.py:11-29, keeps only the audio encoder truncated to the first 3/4 of its blocks
(24 of 32 for large-v2).  The returned object exposes ``.encoder(mel[B,80,n]) -> [B, ceil(n/2), state]``.
Arithmetic is fp32 (the reference runs fp16 on CUDA, fp32 on CPU -- :22-23; fp32 is the parity default,
SURVEY.md section 0).
"""
import os
import threading

import numpy as np
import torch

from .. import _lib, cmodel
from .. import weights as PW
from ..ops import Ops
from ..vits import consts as K


class AudioEncoder:
    """whisper/model.py:132-163 (conv stem + pre-LN attention blocks + ln_post), time-major fp32.  The forward pass is composed by
    the C++ host inside libsvcmi.so (csrc/host_stages.hip: svcmi_whisper_encoder_fwd); this object owns the weights, the tuning
    table and makes ONE library call."""

    def __init__(self, w, ops):
        self.w, self.ops = w, ops
        # Tuning table for the M = 500-row window GEMMs (0 = the library's own table, measured with scripts/microbench.py gemm / wp16
        # / lp on MI355X: 2 / 4 K slices for the two N = n_state projections, the 64x80 tile of the 16x16x4 policy where it balances
        # the 256 CUs better than 64x64; -1 = the tile heuristic of svcmi_conv_gemm_f32).  Fields of svcmi_whisper_model.
        self.split_o = self.split_mlp = 0
        self.tile_qkv = self.tile_o = self.tile_mlp1 = self.tile_mlp2 = 0
        self.small_m_rows = 0                    # above this many GEMM rows (0 = 1024) the library's own tile heuristic takes over
        # tuning runs: SVCMI_WHISPER_TUNE="split_o=1,tile_mlp1=1,..." overrides the fields above for this process (A/B runs of bench.py)
        for item in filter(None, os.environ.get("SVCMI_WHISPER_TUNE", "").split(",")):
            k, _, v = item.partition("=")
            if k.strip() not in ("split_o", "split_mlp", "tile_qkv", "tile_o", "tile_mlp1", "tile_mlp2", "small_m_rows"):
                raise ValueError(f"SVCMI_WHISPER_TUNE: unknown field {k!r}")
            setattr(self, k.strip(), int(v))
        # GEMM operand precision of this encoder: None = fp32 (parity default); "bf16x3" / "bf16" / "f16" (the reference's
        # own accelerator path is fp16: whisper/inference.py:22-23,43-44) route the linear layers through
        # svcmi_conv_gemm_lp.  LayerNorm, softmax, GELU, residual stream and accumulation stay fp32 in every mode.
        self.precision = None
        self._cm = {}
        self._lock = threading.Lock()

    def _cmodel(self):
        prec = _lib.PRECISIONS.get(self.precision, self.precision)
        cm = self._cm.get(prec)
        if cm is None:
            with self._lock:
                cm = self._cm.get(prec)
                if cm is None:
                    cm = self._cm[prec] = cmodel.whisper_cmodel(self.w, self.ops, prec)
        m = cm.struct
        m.split_o, m.split_mlp, m.small_m_rows = self.split_o, self.split_mlp, self.small_m_rows
        m.tile_qkv, m.tile_o, m.tile_mlp1, m.tile_mlp2 = self.tile_qkv, self.tile_o, self.tile_mlp1, self.tile_mlp2
        return cm

    @torch.no_grad()
    def __call__(self, mel, noise=None, noise_scale=0.1):
        dev = self.w.lnp_g.device
        mel = mel.to(dev, torch.float32).contiguous()
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
        return self.ops.whisper_encoder_fwd(self._cmodel(), mel, noise, noise_scale)


class WhisperEncoderModel:
    """What ``load_model`` returns: the reference returns a ``Whisper`` whose decoder was deleted; callers
    only touch ``.encoder`` (whisper/inference.py:47,59)."""

    def __init__(self, ckpt, device, ops=None, packed=None):
        """``packed``: an already packed ``svcmi.weights.WhisperWeights`` on ``device`` (``svcmi.dist.broadcast_packed``);
        ``ckpt`` is then ignored."""
        self.ops = ops if ops is not None else Ops()
        self.weights = packed if packed is not None else PW.WhisperWeights(ckpt, device)
        self.dims = self.weights.dims
        self.encoder = AudioEncoder(self.weights, self.ops)
        self.device = torch.device(device)

    def eval(self):
        return self

    def embed_audio(self, mel):
        return self.encoder(mel)


def load_model(path, device, ops=None):
    """whisper/inference.py:11-29.  ``path`` may also be an already-loaded checkpoint dict."""
    ckpt = torch.load(path, map_location="cpu") if isinstance(path, str) else path
    return WhisperEncoderModel(ckpt, device, ops=ops)


@torch.no_grad()
def pred_ppg_from_mel(whisper, mels, kept_frames, mel_noises=None, max_batch=8):
    """The encoder half of whisper/inference.py:32-62 starting at the mel tensor (the hot-path contract
    starts there): per 15 s window ``mel + 0.1*randn`` (:46,58) -> encoder -> first
    ``len//320`` frames (:40,48).  Returns a device tensor [T50, state].  The reference runs the windows one after the other; they are
    independent, so consecutive windows of equal length (all but the remainder window) go through the encoder as ONE batch of up to
    ``max_batch`` -- M = B * 750 rows fill the chip where a single window's GEMMs are launch-bound (bench.py configs[4] times exactly
    this)."""
    out, i, dev = [None] * len(mels), 0, whisper.device
    while i < len(mels):
        j = i + 1
        while j < len(mels) and j - i < max_batch and mels[j].shape == mels[i].shape:
            j += 1
        mel = torch.stack([m.to(dev, torch.float32) for m in mels[i:j]])
        nz = torch.randn_like(mel) if mel_noises is None else torch.stack([z.to(dev, torch.float32) for z in mel_noises[i:j]])
        ppg = whisper.encoder(mel, nz, 0.1)
        for k in range(i, j):
            out[k] = ppg[k - i, :kept_frames[k]]
        i = j
    return torch.cat(out, 0)


def window_plan(n_samples, sr=16000, window_s=K.WHISPER_WINDOW_S):
    """(start, stop, kept_frames) per window, whisper/inference.py:37-61: full 15 s windows while
    ``idx + 15 s < len`` and one remainder window; kept frames = samples // 320."""
    plan, idx = [], 0
    step = window_s * sr
    while idx + step < n_samples:
        plan.append((idx, idx + step, step // 320))
        idx += step
    if idx < n_samples:
        plan.append((idx, n_samples, (n_samples - idx) // 320))
    return plan


@torch.no_grad()
def ppg_from_audio(whisper, wav):
    """whisper/inference.py:32-62 on a 16 kHz float waveform (numpy [n]): log-mel front-end on the GPU, 15 s windows, encoder, kept
    frames -> device tensor [T50, n_state].  Nothing here waits for the device."""
    from . import audio
    plan = window_plan(wav.shape[0])
    mels = [audio.log_mel_spectrogram(torch.from_numpy(wav[s:e]), ops=whisper.ops, device=whisper.device) for (s, e, _) in plan]
    return pred_ppg_from_mel(whisper, mels, [k for (_, _, k) in plan])


def pred_ppg(whisper, wavPath, ppgPath, device):
    """whisper/inference.py:32-62 with the 16 kHz loader + GPU log-mel front-end of svcmi.whisper.audio (row N2 of
    SURVEY.md 8f); writes the same float32 [T50, 1280] .npy."""
    from . import audio
    np.save(ppgPath, ppg_from_audio(whisper, audio.load_audio(wavPath)).cpu().numpy(), allow_pickle=False)
