"""Drop-in for the reference's whisper/inference.py: ``load_model`` / ``pred_ppg`` on svcmi kernels.

``load_model(path, device)`` consumes the OpenAI checkpoint format ``{"dims", "model_state_dict"}`` and,
like whisper/inference.py:11-29, keeps only the audio encoder truncated to the first 3/4 of its blocks
(24 of 32 for large-v2).  The returned object exposes ``.encoder(mel[B,80,n]) -> [B, ceil(n/2), state]``.
Arithmetic is fp32 (the reference runs fp16 on CUDA, fp32 on CPU -- :22-23; fp32 is the parity default,
SURVEY.md section 0).
"""
import numpy as np
import torch

from .. import weights as PW
from ..ops import ACT_GELU, Ops
from ..vits import consts as K


class AudioEncoder:
    """whisper/model.py:132-163 (conv stem + pre-LN attention blocks + ln_post), time-major fp32."""

    def __init__(self, w, ops):
        self.w, self.ops = w, ops
        # Tuning table for the M = 500-row window GEMMs (scripts/microbench.py gemm / wp16, MI355X): K slices of the two
        # N = n_state projections, and the 64x80 tile of the 16x16x4 policy where it balances the 256 CUs better than
        # 64x64 (N = 5120 -> 8 x 64 = 512 blocks; N = 1280 with 2-4 K slices); the QKV projection stays on 64x64.
        self.split_o, self.split_mlp = 2, 4
        self.tile_qkv = 0                        # 0 = library heuristic (64x64 at M = 500)
        self.tile_o = self.tile_mlp1 = self.tile_mlp2 = 6          # SVCMI_CONV_TILE_P16_64x80 >> 8
        # GEMM operand precision of this encoder: None = fp32 (parity default); "bf16x3" / "bf16" / "f16" (the reference's
        # own accelerator path is fp16: whisper/inference.py:22-23,43-44) route the linear layers through
        # svcmi_conv_gemm_lp.  LayerNorm, softmax, GELU, residual stream and accumulation stay fp32 in every mode.
        self.precision = None
        self.small_m_rows = 1024                 # above this many GEMM rows the library's own tile heuristic takes over
        self.lp_split_o, self.lp_split_mlp = 2, 4
        # measured (scripts/microbench.py lp, profiles/r02a_microbench_lp.log): the two N = n_state projections run best on 64x128
        # tiles with 2 / 4 K slices (17 vs 22 us, 37 vs 38 us at Tw = 500); QKV / MLP-up follow the library heuristic
        self.lp_tile_qkv = self.lp_tile_mlp1 = 0      # 0 = library heuristic
        self.lp_tile_o = self.lp_tile_mlp2 = 9        # SVCMI_CONV_TILE_64x128 >> 8

    @torch.no_grad()
    def __call__(self, mel, noise=None, noise_scale=0.1):
        with self.ops.use_precision(self.precision):
            return self._forward(mel, noise, noise_scale)

    def _forward(self, mel, noise, noise_scale):
        w, ops = self.w, self.ops
        lp = ops.precision != 0
        split_o, split_mlp = (self.lp_split_o, self.lp_split_mlp) if lp else (self.split_o, self.split_mlp)
        tile_qkv, tile_o, tile_m1, tile_m2 = (self.lp_tile_qkv, self.lp_tile_o, self.lp_tile_mlp1, self.lp_tile_mlp2) if lp else \
            (self.tile_qkv, self.tile_o, self.tile_mlp1, self.tile_mlp2)
        if mel.shape[0] * ((mel.shape[2] + 1) // 2) > self.small_m_rows:
            # batched windows (BASELINE.json configs[3] / [4]): M = B * Tw rows fill the chip with large tiles; the K slices and
            # the 64x80 tiles above are a single-window (M = 500 .. 750) tuning
            split_o = split_mlp = 1
            tile_qkv = tile_o = tile_m1 = tile_m2 = 0
        dev = w.lnp_g.device
        mel = mel.to(dev, torch.float32).contiguous()
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
        x = ops.ncl_to_nlc(mel, noise, noise_scale if noise is not None else 0.0)          # [B, n, 80]
        x = ops.conv(x, w.conv1_w, w.conv1_b, ksize=3, pad=1, act=ACT_GELU)               # model.py:150
        n = x.shape[1]
        tw = (n + 2 - 3) // 2 + 1
        if tw > w.pos.shape[0]:
            raise AssertionError("incorrect audio shape")                                   # model.py:156
        x = ops.conv(x, w.conv2_w, w.conv2_b, ksize=3, stride=2, pad=1, act=ACT_GELU, res=w.pos[:tw])  # :151-158
        scale = float(w.S // w.heads) ** -0.5          # (d^-0.25 on q) * (d^-0.25 on k), model.py:90-92
        # model.py:118-129.  The two residual projections (attention out, MLP down: N = 1280 leaves most CUs idle) run
        # split-K; their slabs are summed by the kernel that also applies the residual add and the NEXT LayerNorm
        # (attn_ln -> mlp_ln -> next block's attn_ln -> ... -> ln_post), so a block is 6 launches.
        nb = len(w.blocks)
        h = ops.layernorm(x, w.blocks[0]["ln1_g"], w.blocks[0]["ln1_b"]) if nb else None
        for i, blk in enumerate(w.blocks):
            qkv = ops.conv(h, blk["qkv_w"], blk["qkv_b"], tile=tile_qkv)
            a = ops.attention(qkv, w.heads, scale)
            p = ops.conv(a, blk["o_w"], None, partials=True, split_k=max(1, min(split_o, blk["o_w"].shape[1] // 128)), tile=tile_o)
            h = ops.splitk_layernorm(p, blk["o_b"], x, blk["ln2_g"], blk["ln2_b"], out=h)
            m = ops.conv(h, blk["m1_w"], blk["m1_b"], act=ACT_GELU, tile=tile_m1, split_k=1)
            p = ops.conv(m, blk["m2_w"], None, partials=True, split_k=max(1, min(split_mlp, blk["m2_w"].shape[1] // 128)), tile=tile_m2)
            g, b = (w.blocks[i + 1]["ln1_g"], w.blocks[i + 1]["ln1_b"]) if i + 1 < nb else (w.lnp_g, w.lnp_b)
            h = ops.splitk_layernorm(p, blk["m2_b"], x, g, b, out=h)
        return h if nb else ops.layernorm(x, w.lnp_g, w.lnp_b)


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


def pred_ppg(whisper, wavPath, ppgPath, device):
    """whisper/inference.py:32-62 with the 16 kHz loader + GPU log-mel front-end of svcmi.whisper.audio (row N2 of
    SURVEY.md 8f); writes the same float32 [T50, 1280] .npy."""
    from . import audio
    wav = audio.load_audio(wavPath)
    plan = window_plan(wav.shape[0])
    mels = [audio.log_mel_spectrogram(torch.from_numpy(wav[s:e]), ops=whisper.ops, device=whisper.device) for (s, e, _) in plan]
    ppg = pred_ppg_from_mel(whisper, mels, [k for (_, _, k) in plan])
    np.save(ppgPath, ppg.cpu().numpy(), allow_pickle=False)
