"""Drop-in for the reference's hubert/inference.py + ``HubertSoft.units`` (hubert/hubert_model.py:64-72) on the svcmi
kernels (row N3 of SURVEY.md 8f: the extractor that produces the ``vec`` input of the synthesiser).

    model = load_model("hubert_pretrain/hubert-soft-0d54a1f4.pt", "cuda")     # hubert/inference.py:17-23
    vec = model.units(wav[1, 1, n])                                           # [1, T, 256], T = (n + 80 - 400) // 320 + 1
    pred_vec(model, "in.wav", "out.vec.npy", "cuda")                          # 20 s windows, float32 [T, 256] (:25-50)

Everything is the existing kernels: the strided feature-extractor convolutions, the 16-group positional convolution
(one implicit-GEMM launch per group on column views), the q/k/v / MLP projections (conv_gemm), attention (12 heads x
64), post-LN LayerNorms with the residual add fused; the only new kernel is the per-channel norm over time + GELU of the
first layer (GroupNorm(512, 512)).  fp32 throughout (the reference runs fp16 on CUDA, fp32 on CPU -- :20-21).
"""
import numpy as np
import torch

from .. import weights as PW
from ..ops import ACT_GELU, Ops

WINDOW_S = 20


class HubertSoft:
    def __init__(self, state_dict, device, ops=None):
        self.ops = ops if ops is not None else Ops()
        self.device = torch.device(device)
        self.w = PW.HubertWeights(state_dict, self.device)
        self.precision = None        # GEMM operand precision (Ops.use_precision): None = fp32; the reference's CUDA path is fp16 (:20-21)

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise NotImplementedError("construct the model on its device (load_model(path, device))")
        return self

    @torch.no_grad()
    def units(self, wav):
        """wav [B, 1, n] (or [n]) float at 16 kHz -> soft units [B, T, proj] on the device."""
        with self.ops.use_precision(self.precision):
            return self._units(wav)

    def _units(self, wav):
        w, ops = self.w, self.ops
        x = wav.to(self.device, torch.float32)
        if x.dim() == 1:
            x = x.view(1, 1, -1)
        B, _, n = x.shape
        x = x.reshape(B, n).contiguous()
        # FeatureExtractor (:75-96); the F.pad(wav, (40, 40)) of :70 is the zero padding of the first convolution
        t0 = (n + 80 - 10) // 5 + 1
        h = ops.conv(x, w.conv0_w, None, ksize=10, stride=5, pad=40, c_in=1, ldx=1, t_in=n, t_out=t0, x_bstride=x.stride(0))
        h = ops.channel_norm_gelu(h, w.norm0_g, w.norm0_b, out=h)
        for (cw, k) in w.convs:
            h = ops.conv(h, cw, None, ksize=k, stride=2, act=ACT_GELU)
        T = h.shape[1]
        # FeatureProjection (:99-110)
        h = ops.layernorm(h, w.fp_g, w.fp_b)
        x = ops.conv(h, w.fp_w, w.fp_b2)
        # PositionalConvEmbedding (:113-131): grouped, weight-normed per tap, last frame dropped; x = x + gelu(pos(x))
        y = torch.empty_like(x)
        cg = w.E // w.G
        for gi in range(w.G):
            xs = x[:, :, gi * cg:(gi + 1) * cg]
            ops.conv(xs, w.pos_w[gi], w.pos_b[gi], ksize=w.pos_k, pad=w.pos_k // 2, t_out=T, act=ACT_GELU, res=xs,
                     out=y[:, :, gi * cg:(gi + 1) * cg], c_in=cg, ldx=x.stride(1), t_in=T, x_bstride=x.stride(0))
        x = ops.layernorm(y, w.norm_g, w.norm_b)
        # 12 post-LN encoder layers (:20-25): x = LN1(x + SA(x)); x = LN2(x + W2 gelu(W1 x))
        scale = float(w.E // w.heads) ** -0.5
        for L in w.layers:
            qkv = ops.conv(x, L["in_w"], L["in_b"])
            a = ops.attention(qkv, w.heads, scale)
            o = ops.conv(a, L["out_w"], L["out_b"])
            x = ops.layernorm(x, L["n1_g"], L["n1_b"], res=o)
            m = ops.conv(x, L["l1_w"], L["l1_b"], act=ACT_GELU)
            o = ops.conv(m, L["l2_w"], L["l2_b"])
            x = ops.layernorm(x, L["n2_g"], L["n2_b"], res=o)
        return ops.conv(x, w.proj_w, w.proj_b)                                       # :72


def load_model(path, device, ops=None):
    """hubert/inference.py:17-23 + hubert_model.hubert_soft (:209-222).  ``path`` may be an already loaded state dict."""
    sd = torch.load(path, map_location="cpu") if isinstance(path, str) else path
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}     # consume_prefix (:219)
    return HubertSoft(sd, device, ops=ops)


def window_plan(n_samples, sr=16000, window_s=WINDOW_S):
    """hubert/inference.py:29-49: full 20 s windows while ``idx + 20 s < len``, then one remainder window."""
    plan, idx, step = [], 0, window_s * sr
    while idx + step < n_samples:
        plan.append((idx, idx + step))
        idx += step
    if idx < n_samples:
        plan.append((idx, n_samples))
    return plan


@torch.no_grad()
def pred_vec(model, wavPath, vecPath, device):
    """hubert/inference.py:25-50: float32 [T, 256] .npy (hop 320)."""
    from ..whisper.audio import load_audio
    audio = load_audio(wavPath)
    np.save(vecPath, units_windowed(model, audio).cpu().numpy(), allow_pickle=False)


@torch.no_grad()
def units_windowed(model, audio, max_batch=8):
    """The reference's 20 s window loop (hubert/inference.py:31-48); consecutive windows of equal length run as one batch."""
    plan = window_plan(audio.shape[0])
    out, i = [], 0
    while i < len(plan):
        j = i + 1
        while j < len(plan) and j - i < max_batch and plan[j][1] - plan[j][0] == plan[i][1] - plan[i][0]:
            j += 1
        # (numpy, not torch, on the host: a torch CPU op on a many-core box wakes its whole intra-op thread pool -- milliseconds per call)
        wav = torch.from_numpy(np.ascontiguousarray(np.stack([audio[s:e] for (s, e) in plan[i:j]])[:, None, :]))
        out.extend(model.units(wav))
        i = j
    return torch.cat(out, 0)
