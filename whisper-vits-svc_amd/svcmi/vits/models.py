"""Drop-in ``SynthesizerInfer`` (reference: vits/models.py:211-256) running on the svcmi HIP kernels.

Same constructor, ``state_dict``/``load_state_dict`` key names, ``eval``/``to``, ``pitch2source``,
``source2wav`` and ``inference`` signatures and tensor layouts as the reference class, so
``svc_inference.py``-style callers need no change.  Differences, all additive:
  * the three stochastic draws of the path (vits/models.py:51, vits_decoder/nsf.py:232-235,311) can be
    passed explicitly (``noise=...``) so runs are reproducible / comparable; when omitted they are drawn
    with torch's generator on the device, as the reference does;
  * there is no autograd and no CPU path: without the gfx950 library construction of ``Ops`` raises.
Internally everything is time-major ``[B, T, C]`` fp32.  The forward pass itself -- prior encoder, reverse flow,
generator -- is composed by the C++ host inside libsvcmi.so (csrc/host_stages.hip: svcmi_synth_infer_fwd, svcmi_pitch2source_fwd):
this class converts layouts at the API edge, owns the weights and makes ONE library call per stage.
"""
import threading
from collections import OrderedDict

import torch

from .. import _lib, cmodel
from .. import weights as PW
from ..ops import Ops
from . import consts as K
from .spec import default_state_dict, param_shapes


class SynthesizerInfer:
    def __init__(self, spec_channels, segment_size, hp, ops=None):
        self.spec_channels, self.segment_size, self.hp = spec_channels, segment_size, hp
        self._ops = ops
        self._sd = None          # CPU state dict (reference key names)
        self._w = None           # packed device weights
        self._device = None
        self.training = False
        self._stop_after = None          # tuning aid (scripts/stage_times.py), never set in production
        self._cm = {}                    # C model structs (svcmi_synth_model), one per GEMM operand precision
        self._lock = threading.RLock()   # lazy weight packing / struct building may be reached from several worker threads
        # GEMM operand precision of prior encoder / flow / generator: None = fp32 (parity default), "bf16x3" / "bf16" /
        # "f16" (one mode for every GEMM), or "mixed" / "mixed:enc=bf16x3,flow=f16,..." = a mode per layer class
        # (_lib.parse_precision, _lib.MIXED_DEFAULT).  Element-wise kernels, softmax, LayerNorm, SnakeAlias and accumulation stay fp32.
        self.precision = None
        # Streaming decoder (BASELINE.json configs[4]): None = the generator sees a whole synthesis chunk;
        # N = it runs over time tiles of N frames plus a STREAM_HALO-frame halo on each side that is computed and discarded.
        # The generator's exact receptive field is < 31 frames (see STREAM_HALO) and every kernel's arithmetic for an output row is
        # independent of the row's position in a launch, so the kept samples are BIT-IDENTICAL for every N in fp32 (split-K is pinned
        # off in this mode: its slice count would otherwise follow the problem size; in the 16-bit modes a launch's eligibility for the
        # 16-bit kernels follows its size too, so tilings there agree to the mode's rounding, 3e-4 under "mixed").  The reference's own chunk seams
        # (svc_inference.py:94-131) are untouched: tiling happens inside a chunk.
        self.stream_frames = None
        # svc_infer: synthesis chunks of one clip in flight on this many HIP streams (1 = one after the other, the reference's order);
        # results do not depend on it (bit-identical, tests/test_gpu_engine.py); a 3-minute song: 59.3 -> 48.4 ms with 3 streams
        # (profiles/r02w_chunk_streams.log).
        self.chunk_streams = 3

    # ------------------------------------------------------------------ nn.Module-like surface
    @property
    def ops(self):
        if self._ops is None:
            self._ops = Ops()    # raises if libsvcmi.so / GPU is missing: no fallback
        return self._ops

    def state_dict(self):
        if self._sd is None:
            self._sd = default_state_dict(self.hp)
        return OrderedDict(self._sd)

    def load_state_dict(self, sd, strict=True):
        want = param_shapes(self.hp)
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        new = default_state_dict(self.hp) if missing else OrderedDict()
        for k, shape in want.items():
            if k in sd:
                t = sd[k].detach().float().cpu()
                if tuple(t.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(shape)}")
                new[k] = t
        self._sd = OrderedDict((k, new[k]) for k in want)
        self._w, self._cm = None, {}
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("svcmi is an inference engine")
        return self

    def to(self, device):
        self._device = torch.device(device)
        self._w, self._cm = None, {}
        return self

    def remove_weight_norm(self):
        """Weight-norm is folded at pack time; kept for API parity (the reference's own method raises,
        vits/models.py:96-98)."""
        return None

    def parameters(self):
        return iter(self.state_dict().values())

    def load_packed(self, weights, device):
        """Adopt kernel-ready weights (a ``svcmi.weights.VitsWeights`` whose tensors live on ``device``, e.g. the views of
        the arena ``svcmi.dist.broadcast_packed`` delivered): no folding / packing happens on this rank."""
        self._w, self._device, self._cm = weights, torch.device(device), {}
        return self

    def _weights(self):
        if self._w is None:
            with self._lock:
                if self._w is None:
                    dev = self._device or torch.device("cuda" if self.ops.on_gpu else "cpu")
                    w = PW.VitsWeights(self.state_dict(), self.hp, dev)
                    if dev.type == "cuda":
                        torch.cuda.synchronize(dev)      # uploaded on the packing thread's stream; every stream may use them from here on
                    self._device, self._w = dev, w
        return self._w

    def _cmodel(self, precision=None):
        """The svcmi_synth_model struct for ``precision`` (None = self.precision): pointers into the packed weights (+ their 16-bit
        images, packed on first use)."""
        p = self.precision if precision is None else precision
        prec = _lib.parse_precision(p)            # (code, None) or (PREC_MIXED, per-class modes): hashable, one struct per distinct policy
        prec = prec[0] if prec[1] is None else prec
        cm = self._cm.get(prec)
        if cm is None:
            with self._lock:
                cm = self._cm.get(prec)
                if cm is None:
                    cm = self._cm[prec] = cmodel.synth_cmodel(self._weights(), self.ops, prec)
        return cm

    def warm(self):
        """Pack the weights and build the C model now (on the calling thread) instead of on first use: call before handing the model to
        worker threads / lanes."""
        self._cmodel()
        return self

    # ------------------------------------------------------------------ reference methods
    STREAM_HALO = 32      # frames of halo per side of a streaming-decoder tile (csrc/host_stages.hip: the FIR chain's exact support is 30.9)

    def pitch2source(self, f0, noise=None):
        """f0 [B,T] Hz -> harmonic source [B,1,320*T] (vits_decoder/generator.py:160-165).
        ``noise`` = (rand_ini [B,11], randn [B,L,11]) to pin the draws of nsf.py:232-235,311."""
        w, ops = self._weights(), self.ops
        f0 = f0.to(self._device, torch.float32).contiguous()
        B, T = f0.shape
        L = T * w.hop
        if noise is None:
            rand_ini = torch.rand(B, K.NSF_HARMONICS, device=f0.device)
            nz = torch.randn(B, L, K.NSF_HARMONICS, device=f0.device)
        else:
            rand_ini, nz = (t.to(self._device, torch.float32).contiguous() for t in noise)
        return ops.pitch2source_fwd(self._cmodel(), f0, rand_ini, nz).view(B, 1, L)

    def source2wav(self, source):
        """-> int16 numpy (vits_decoder/generator.py:167-173)."""
        return self.ops.source2wav(source.to(self._device, torch.float32).squeeze()).cpu().numpy()

    def _stop_code(self):
        s = self._stop_after
        if s is None:
            return _lib.STOP_NONE
        if isinstance(s, tuple):
            return _lib.STOP_STAGE0 + int(s[1])
        return {"prior": _lib.STOP_PRIOR, "flow": _lib.STOP_FLOW, "gen_pre": _lib.STOP_GEN_PRE}[s]

    @torch.no_grad()
    def inference(self, ppg, vec, pit, spk, ppg_l, source, noise=None, return_parts=False):
        """vits/models.py:251-256.  ppg [B,T,ppg_dim], vec [B,T,vec_dim], pit [B,T] Hz, spk [B,spk_dim],
        ppg_l int64 [B], source [B,1,hop*T] -> waveform [B,1,hop*T] (device tensor).
        ``noise``: the randn_like(m) of vits/models.py:51 in the reference layout [B,inter,T]."""
        w, ops, dev = self._weights(), self.ops, self._device
        ppg = ppg.to(dev, torch.float32).contiguous()
        vec = vec.to(dev, torch.float32).contiguous()
        pit = pit.to(dev, torch.float32).contiguous()
        spk = spk.to(dev, torch.float32).contiguous()
        lengths = ppg_l.to(dev, torch.int32).contiguous()
        B, T, _ = ppg.shape
        source = source.to(dev, torch.float32).contiguous().view(B, T * w.hop)
        if noise is None:
            noise = torch.randn(B, w.I, T, device=dev)
        noise = noise.to(dev, torch.float32).contiguous()
        out = ops.synth_infer_fwd(self._cmodel(), ppg, vec, pit, spk, lengths, source, noise, stream_frames=self.stream_frames or 0,
                                  want_parts=return_parts)
        if return_parts:
            o, (z_p, z) = out
            return o, {"z_p": ops.nlc_to_ncl(z_p), "z": ops.nlc_to_ncl(z)}
        return out

    __call__ = inference

    @torch.no_grad()
    def inference_ppg50(self, ppg50, vec, pit, spk, lengths, source, noise=None):
        """Same as ``inference`` but takes the Whisper PPG at its native 50 fps ([B, ceil(T/2), ppg_dim], may be a batch-strided view of
        the encoder output) and fuses the ``np.repeat(ppg, 2, 0)`` of svc_inference.py:175-177 into the first conv's loads.  All
        arguments must already be device tensors (fp32 / int32 lengths); nothing here synchronises, so the call can be captured in a
        HIP graph."""
        w, ops = self._weights(), self.ops
        B, T = pit.shape
        if noise is None:
            noise = torch.randn(B, w.I, T, device=pit.device)
        return ops.synth_infer_fwd(self._cmodel(), ppg50, vec, pit, spk, lengths, source.reshape(B, T * w.hop), noise, ppg_row_shift=1,
                                   stream_frames=self.stream_frames or 0, stop_after=self._stop_code())
