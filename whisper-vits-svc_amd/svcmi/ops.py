"""Thin tensor-level wrappers over the C ABI (include/svcmi.h).

PyTorch is used for device memory and streams only.  Activations are time-major fp32 tensors
``[B, T, C]`` (contiguous, ``C % 4 == 0``); see DESIGN.md for the layout rationale.  Every method
launches one kernel on the current stream and returns its output tensor.
"""
import contextlib
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_MISH, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, CONV_ACCUMULATE, CONV_MASK_IN,  # noqa: F401
                   CONV_MASK_OUT, CONV_PARTIALS, PREC_BF16, PREC_BF16X3, PREC_F16, PREC_F32, PRECISIONS, ConvDesc, SnakeConvDesc, SvcmiError,
                   SynthIO, TraceRecord)


SPLIT16 = "bf16x3"      # ``out16`` value: rows [hi: Cp bf16 | lo: Cp bf16], hi = bf16(v), lo = bf16(v - hi), Cp = C rounded up to 8


def _fmt16(dtype):
    """torch.bfloat16 / torch.float16 / SPLIT16 -> the svcmi_precision code of a 16-bit output copy."""
    return {torch.bfloat16: PREC_BF16, torch.float16: PREC_F16, SPLIT16: PREC_BF16X3}[dtype]


def _alloc16(lead, c, out16, device):
    """The 16-bit second output of a producer: [*lead, c] in the 16-bit dtype, or the split layout [*lead, 2*Cp] (zero-filled pads)."""
    if out16 == SPLIT16:
        cp = (c + 7) // 8 * 8
        return (torch.zeros if cp != c else torch.empty)(*lead, 2 * cp, dtype=torch.bfloat16, device=device)
    return torch.empty(*lead, c, dtype=out16, device=device)


def split16_to_f32(x16, c):
    """hi + lo of a SPLIT16 tensor as fp32 [..., c] (tests, debugging)."""
    cp = x16.shape[-1] // 2
    return x16[..., :c].float() + x16[..., cp:cp + c].float()


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class Ops:
    """Bound to one loaded library.  ``Ops()`` loads the hipcc-built libsvcmi.so and requires CUDA(HIP)
    tensors; there is no CPU path in the product (tests/emu builds a separate emulation library from the
    same kernel source and passes it in explicitly)."""

    def __init__(self, lib=None):
        # streams that should overlap (clips / chunks in flight, lanes.py) each need their own hardware queue; the runtime reads
        # this when HIP initialises, so it only helps if no HIP call was made yet (the CLIs set it first thing)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        self.lib = lib if lib is not None else _lib.load_library()
        self.build = self.lib.svcmi_build_info().decode()
        self.on_gpu = self.build.startswith("hip")
        if self.on_gpu and not torch.cuda.is_available():
            raise SvcmiError("libsvcmi.so is the gfx950 build but no GPU is visible; svcmi has no CPU fallback")
        self.launches = 0
        self.workspaces = {}     # split-K scratch, one per (device, stream): launches on parallel streams must not share it
        self.workspace_floats = 16 * 1024 * 1024
        self._retired = []
        self._graph_pinned = set()        # workspace keys a HIP graph capture has recorded pointers into: their outgrown buffers are kept
        # split-K slices combined inside the GEMM launch by the last-arriving block of each tile.  Correct (bit-identical to
        # the two-kernel reduction, tests/test_gpu_kernels.py) but SLOWER on MI355X at these slab sizes (64-128 KB per tile):
        # 10 s step 15.8 ms (write-through slabs) / 16.1 ms (release fence per block) vs 12.6 ms -> off by default.
        self.inlaunch_reduce = False
        self.timeline = None     # set to a list to record (kernel, work, start_event, end_event) per launch
        # GEMM operand precision (include/svcmi.h: enum svcmi_precision).  fp32 is the parity default; "bf16x3" / "bf16" /
        # "f16" route every eligible convolution through svcmi_conv_gemm_lp with a 16-bit weight image packed on first use
        # (cached on the fp32 weight tensor).  Launches below `lp_min_flops` stay fp32: a GEMM of < ~1.5 GFLOP is latency-bound
        # on this chip (8-17 us whatever the operand type; measured: the 75 small prior / flow GEMMs of a 10 s clip cost the same
        # 1.35 ms through either kernel), so rounding its operands buys nothing and costs accuracy.  At batch 16 the same
        # layers are 16x the work and cross the threshold.
        self._tls = threading.local()          # `precision` is per calling thread (worker threads of the folder driver share one Ops)
        self._lock = threading.RLock()         # guards the lazily built per-tensor 16-bit weight images; re-entrant: ClipLanes.capture holds it around a capture whose warm-up may build them
        self.lp_min_flops = 1.5e9
        # development knobs of the library (svcmi_tune_set: tile variants, fused / unfused paths).  Tile / split / launch-shape knobs
        # leave every result bit-identical; two select another KERNEL FORM of the narrow generator stages and change the fp32
        # summation order (<= 5e-6 on the waveform) or the operand precision: amp_mfma (fp32 matrix-core half-step at 20 channels and
        # batch <= 2 -- so fp32 bits are reproducible per batch class, <= 2 clips vs more), amp_lp (fp16 operands).
        # SVCMI_TUNE="amp_mfma=0,ring2=15" applies them to this process (A/B runs of bench.py on the GPU box)
        self.tune = {}                         # what SVCMI_TUNE set for this process (a scoped change restores to these, not to the defaults)
        for item in filter(None, os.environ.get("SVCMI_TUNE", "").split(",")):
            k, _, v = item.partition("=")
            if self.lib.svcmi_tune_set(k.strip().encode(), int(v)) != 0:
                raise SvcmiError(f"SVCMI_TUNE: unknown knob or value {item!r}")
            self.tune[k.strip()] = int(v)

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return torch.cuda.current_stream().cuda_stream if self.on_gpu else 0

    def _chk(self, *tensors):
        for t in tensors:
            if t is None:
                continue
            if t.is_cuda != self.on_gpu:
                raise SvcmiError(f"tensor on {t.device} but library build is {self.build}")

    def _call(self, name, *args, work=None):
        if self.timeline is not None and self.on_gpu:
            # HIP events on the SAME stream the kernel is launched on (bench.py roofline leg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = getattr(self.lib, name)(*args)
            e1.record()
            self.timeline.append((name, work or {}, e0, e1))
        else:
            rc = getattr(self.lib, name)(*args)
        self.launches += 1
        if rc != 0:
            raise SvcmiError(f"{name} failed with code {rc}")

    # ------------------------------------------------------------------ reduced precision
    @property
    def precision(self):
        return getattr(self._tls, "precision", PREC_F32)

    @precision.setter
    def precision(self, value):
        self._tls.precision = value

    def set_precision(self, precision):
        if precision not in PRECISIONS and precision not in PRECISIONS.values():
            raise SvcmiError(f"unknown precision {precision!r}; one of {sorted(k for k in PRECISIONS if k)}")
        self.precision = PRECISIONS.get(precision, precision)

    @contextlib.contextmanager
    def use_precision(self, precision):
        """``with ops.use_precision("bf16x3"): ...`` -- None leaves the current setting untouched."""
        old = self.precision
        if precision is not None:
            self.set_precision(precision)
        try:
            yield self
        finally:
            self.precision = old

    def lp_weight(self, w, prec):
        """16-bit image of the packed fp32 operand ``w`` [n, ldw] (svcmi_pack_weights_lp), made once per (tensor, precision)."""
        cache = w.__dict__.setdefault("_svcmi_lp", {})
        img = cache.get(prec)
        if img is None:
            with self._lock:
                img = cache.get(prec)
                if img is None:
                    n, ldw = w.shape
                    ldw16 = (ldw + 31) // 32 * 32
                    img = torch.empty(n, (2 if prec in (PREC_BF16X3, _lib.PREC_BF16X3_A16, _lib.PREC_F16W2_A16) else 1) * ldw16, dtype=torch.int16, device=w.device)
                    self._call("svcmi_pack_weights_lp", _ptr(w), n, ldw, prec, _ptr(img), ldw16, self._stream())
                    if self.on_gpu:            # packed on THIS thread's stream; other streams may read it right away
                        torch.cuda.current_stream().synchronize()
                    cache[prec] = img
        return img

    def _lp_eligible(self, d, w, work):
        """The reduced-precision kernels take 16-byte gathers only (c_in, ldx, batch stride % 4, aligned x)."""
        return (self.precision != PREC_F32 and work["flops"] >= self.lp_min_flops and w.dim() == 2 and w.is_contiguous()
                and d.c_in % 4 == 0 and d.ldx % 4 == 0 and d.x_bstride % 4 == 0 and (d.x or 0) % 16 == 0
                and (not d.x_row_shift or d.c_in % 32 == 0) and (d.flags >> 8) in (0, 1, 3, 4, 6, 9))

    def _to_lp(self, d, w, work):
        prec = PREC_F16 if self.precision == _lib.PREC_F16W2 else self.precision        # f16w2 without 16-bit activations = plain fp16
        img = self.lp_weight(w, prec)
        d.w, d.ldw = img.data_ptr(), img.shape[1] // (2 if prec == PREC_BF16X3 else 1)
        work["bytes"] += d.n_out * d.ksize * d.c_in * (4.0 if prec == PREC_BF16X3 else 2.0) - 4.0 * d.n_out * d.ksize * d.c_in
        return prec

    def empty(self, *shape, like=None, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=like.device if like is not None else ("cuda" if self.on_gpu else "cpu"))


    # ------------------------------------------------------------------ stage-level entry points (csrc/host_stages.hip)
    def stage_workspace(self, nbytes, device):
        """The caller-owned workspace of the stage entry points: one buffer per (device, stream), grown on demand.  Stages issued on one
        stream run one after the other, so they can share it; lanes / chunk streams each get their own."""
        key = (device, self._stream(), "stage")
        ws = self.workspaces.get(key)
        capturing = self.on_gpu and torch.cuda.is_current_stream_capturing()
        if ws is None or ws.numel() < nbytes + 256:
            grow = int(nbytes) + 512
            if ws is not None:
                # geometric growth: a run whose clip lengths keep setting new maxima re-allocates O(log) times, not per clip.  The
                # outgrown buffer is kept only if a HIP graph captured on this stream recorded pointers into it; otherwise it goes
                # back to torch's allocator, which is stream-ordered: kernels already queued on this stream finish before any reuse.
                grow = max(grow, ws.numel() + ws.numel() // 2)
                if key in self._graph_pinned or capturing:
                    self._retired.append(ws)
            ws = self.workspaces[key] = torch.empty(grow, dtype=torch.uint8, device=device)
        if capturing:
            self._graph_pinned.add(key)
        base = (ws.data_ptr() + 255) & ~255            # the entry points want 256-byte alignment (CPU tensors of the emulator tests: 64)
        return base, ws.numel() - (base - ws.data_ptr())

    def release_retired(self):
        """Drop the outgrown stage workspaces kept for captured graphs (call after the graphs that may point into them are gone)."""
        self._retired.clear()

    def _stage_call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        self.launches += 1
        if rc != 0:
            raise SvcmiError(f"{name} failed with code {rc}")

    def whisper_encoder_fwd(self, cm, mel, noise, noise_scale):
        """AudioEncoder.forward (whisper/model.py:147-163) as ONE call: mel [B, n_mels, n] (+ noise_scale * noise) -> [B, tw, n_state]."""
        self._chk(mel, noise)
        m = cm.struct
        m.lp_min_flops = max(float(self.lp_min_flops), 1e-30)
        B, _, n = mel.shape
        need = self.lib.svcmi_whisper_workspace_bytes(ctypes.byref(m), B, n)
        if need < 0:
            if (n - 1) // 2 + 1 > m.n_ctx:
                raise AssertionError("incorrect audio shape")                           # whisper/model.py:156
            raise SvcmiError(f"svcmi_whisper_workspace_bytes failed with code {need}")
        ws, ws_bytes = self.stage_workspace(need, mel.device)
        out = torch.empty(B, (n - 1) // 2 + 1, m.n_state, dtype=torch.float32, device=mel.device)
        self._stage_call("svcmi_whisper_encoder_fwd", ctypes.byref(m), _ptr(mel), _ptr(noise), float(noise_scale), B, n, _ptr(out), ws,
                         ws_bytes, self._stream())
        return out

    def pitch2source_fwd(self, cm, f0, rand_ini, noise):
        """Generator.pitch2source (vits_decoder/generator.py:160-165): f0 [B,T], rand_ini [B,11], noise [B,T*hop,11] -> [B, T*hop]."""
        self._chk(f0, rand_ini, noise)
        m = cm.struct
        B, T = f0.shape
        ws, ws_bytes = self.stage_workspace(B * T * 11 * 8 + 256, f0.device)
        out = torch.empty(B, T * m.hop, dtype=torch.float32, device=f0.device)
        self._stage_call("svcmi_pitch2source_fwd", ctypes.byref(m), _ptr(f0), _ptr(rand_ini), _ptr(noise), B, T, _ptr(out), ws,
                         ws_bytes, self._stream())
        return out

    def synth_infer_fwd(self, cm, ppg, vec, pit, spk, lengths, source, noise, *, ppg_row_shift=0, stream_frames=0, stop_after=0,
                        want_parts=False):
        """SynthesizerInfer.inference (vits/models.py:251-256) as ONE call.  ppg [B, T >> shift, ppg_dim] (a batch-strided view is fine),
        vec [B,T,vec_dim], pit [B,T], spk [B,spk_dim], lengths int32 [B], source [B, T*hop], noise [B, inter, T] -> wave [B, 1, T*hop]
        (+ time-major z_p / z [B,T,inter] with ``want_parts``)."""
        self._chk(ppg, vec, pit, spk, lengths, source, noise)
        m = cm.struct
        m.lp_min_flops = max(float(self.lp_min_flops), 1e-30)
        B, T = pit.shape
        if ppg.stride(2) != 1 or ppg.stride(1) != ppg.shape[2] or ppg.shape[1] != (T + (1 << ppg_row_shift) - 1) >> ppg_row_shift:
            raise SvcmiError(f"ppg {tuple(ppg.shape)} / strides {ppg.stride()} do not fit {T} frames with row shift {ppg_row_shift}")
        for t_, shape in ((vec, (B, T, m.vec_dim)), (spk, (B, m.spk_dim)), (source, (B, T * m.hop)), (noise, (B, m.inter, T))):
            if tuple(t_.shape) != shape or not t_.is_contiguous():
                raise SvcmiError(f"expected a contiguous tensor of shape {shape}, got {tuple(t_.shape)}")
        need = self.lib.svcmi_synth_workspace_bytes(ctypes.byref(m), B, T, int(stream_frames))
        if need < 0:
            raise SvcmiError(f"svcmi_synth_workspace_bytes failed with code {need}")
        ws, ws_bytes = self.stage_workspace(need, pit.device)
        io = SynthIO()
        io.ppg, io.vec, io.pit, io.spk, io.lengths, io.source, io.noise = (_ptr(t_) for t_ in (ppg, vec, pit, spk, lengths, source, noise))
        io.ppg_bstride, io.ppg_row_shift, io.batch, io.t = ppg.stride(0), ppg_row_shift, B, T
        io.stream_frames, io.stop_after = int(stream_frames), int(stop_after)
        wave = torch.empty(B, 1, T * m.hop, dtype=torch.float32, device=pit.device)
        parts = None
        if want_parts:
            parts = (torch.empty(B, T, m.inter, dtype=torch.float32, device=pit.device), torch.empty(B, T, m.inter, dtype=torch.float32, device=pit.device))
            io.z_p, io.z = _ptr(parts[0]), _ptr(parts[1])
        io.wave = _ptr(wave)
        self._stage_call("svcmi_synth_infer_fwd", ctypes.byref(m), ctypes.byref(io), ws, ws_bytes, self._stream())
        return (wave, parts) if want_parts else wave

    def synth_stages_fwd(self, cm, ppg, vec, pit, spk, lengths, source, noise, *, ppg_row_shift=0, stream_frames=0):
        """The same forward pass as ``synth_infer_fwd`` through the three per-stage entry points (svcmi_text_encoder_fwd ->
        svcmi_flow_reverse_fwd -> svcmi_generator_fwd), for callers that want the intermediate tensors.  Returns (wave, z_p, z);
        z_p / z time-major [B, T, inter]."""
        self._chk(ppg, vec, pit, spk, lengths, source, noise)
        m = cm.struct
        m.lp_min_flops = max(float(self.lp_min_flops), 1e-30)
        B, T = pit.shape
        need = self.lib.svcmi_synth_workspace_bytes(ctypes.byref(m), B, T, int(stream_frames))
        if need < 0:
            raise SvcmiError(f"svcmi_synth_workspace_bytes failed with code {need}")
        ws, ws_bytes = self.stage_workspace(need, pit.device)
        io = SynthIO()
        io.ppg, io.vec, io.pit, io.spk, io.lengths, io.source, io.noise = (_ptr(t_) for t_ in (ppg, vec, pit, spk, lengths, source, noise))
        io.ppg_bstride, io.ppg_row_shift, io.batch, io.t, io.stream_frames = ppg.stride(0), ppg_row_shift, B, T, int(stream_frames)
        wave = torch.empty(B, 1, T * m.hop, dtype=torch.float32, device=pit.device)
        io.wave = _ptr(wave)
        z_p = torch.empty(B, T, m.inter, dtype=torch.float32, device=pit.device)
        self._stage_call("svcmi_text_encoder_fwd", ctypes.byref(m), ctypes.byref(io), _ptr(z_p), ws, ws_bytes, self._stream())
        z = z_p.clone()
        self._stage_call("svcmi_flow_reverse_fwd", ctypes.byref(m), ctypes.byref(io), _ptr(z), ws, ws_bytes, self._stream())
        self._stage_call("svcmi_generator_fwd", ctypes.byref(m), ctypes.byref(io), _ptr(z), ws, ws_bytes, self._stream())
        return wave, z_p, z

    def trace_begin(self, max_records=8192):
        """Per-launch timing of everything launched from here on (stage entry points: HIP events inside the C host; single-op calls of
        this class: torch events): bench.py's roofline leg."""
        self.timeline = []
        if self.lib.svcmi_trace_begin(max_records) != 0:
            raise SvcmiError("svcmi_trace_begin failed")

    def trace_end(self):
        """-> {entry point: {launches, ms, flops, bytes}} since trace_begin (waits for the traced launches)."""
        recs = (TraceRecord * 8192)()
        n = min(self.lib.svcmi_trace_end(recs, 8192), 8192)
        tl, self.timeline = self.timeline or [], None
        agg = {}

        def add(name, ms, flops, nbytes, a16=False, prec=0):
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "a16_launches": 0, "precisions": []})
            if prec and prec not in a["precisions"]:
                a["precisions"].append(prec)                   # enum svcmi_precision codes the reduced-precision launches of this op ran in
            a["launches"] += 1
            a["ms"] += ms
            a["flops"] += flops
            a["bytes"] += nbytes
            a["a16_launches"] += int(a16)         # GEMM launches that took 16-bit activations (SVCMI_PREC_*_A16)
        for i in range(n):
            add(self.lib.svcmi_trace_op_name(recs[i].op).decode(), recs[i].ms, recs[i].flops, recs[i].bytes, (recs[i].op >> 8) >= _lib.PREC_BF16_A16,
                recs[i].op >> 8)
        if tl and self.on_gpu:
            torch.cuda.synchronize()
        for name, work, e0, e1 in tl:
            add(name, e0.elapsed_time(e1) if self.on_gpu else 0.0, work.get("flops", 0.0), work.get("bytes", 0.0))
        return agg

    # ------------------------------------------------------------------ conv / linear
    def _conv_desc(self, x, w, bias=None, *, ksize=1, stride=1, dilation=1, pad=0, t_out=None, act=ACT_NONE,
             res=None, alpha=1.0, accumulate=False, lengths=None, mask_in=False, mask_out=False, out=None,
             x_row_shift=0, c_in=None, ldx=None, t_in=None, n_out=None, x_bstride=None, tile=0, split_k=0, partials=False):
        """y[b,t,n] = epilogue(sum_k sum_ci x[b, t*stride + k*dilation - pad, ci] * w[n, k*c_in + ci]).
        ``x`` is [B, T, C]; ``w`` is [N, ldw] packed (weights.pack_conv)."""
        self._chk(x, w, bias, res, out, lengths)
        B = x.shape[0]
        if t_in is None:
            t_in = x.shape[1] << x_row_shift
        if ldx is None:
            ldx = x.shape[2] if x.dim() == 3 else 1
        if c_in is None:
            c_in = ldx
        if x_bstride is None:
            x_bstride = x.stride(0)
        N = w.shape[0] if n_out is None else n_out
        if t_out is None:
            t_out = (t_in + 2 * pad - dilation * (ksize - 1) - 1) // stride + 1
        if out is None and not partials:
            out = torch.empty(B, t_out, N, dtype=torch.float32, device=x.device)
        d = ConvDesc()
        d.x, d.w, d.bias, d.res, d.y, d.lengths = _ptr(x), _ptr(w), _ptr(bias), _ptr(res), _ptr(out if out is not None else x), _ptr(lengths)
        d.x_bstride, d.y_bstride = x_bstride, (out.stride(0) if out is not None else 0)
        d.res_bstride = res.stride(0) if (res is not None and res.dim() == 3) else 0   # 2-D res is shared by the batch
        d.batch, d.t_in, d.t_out, d.c_in, d.ldx = B, t_in, t_out, c_in, ldx
        d.n_out, d.ldw, d.ldy = N, w.shape[1], (out.stride(1) if out is not None else N)
        d.ldr = res.stride(-2) if res is not None else 0
        d.ksize, d.stride, d.dilation, d.pad, d.x_row_shift = ksize, stride, dilation, pad, x_row_shift
        d.act = act
        d.flags = (CONV_ACCUMULATE if accumulate else 0) | (CONV_MASK_IN if mask_in else 0) | (CONV_MASK_OUT if mask_out else 0) | \
                  (CONV_PARTIALS if partials else 0) | (tile << 8)
        d.alpha = alpha
        if split_k != 1 or partials:
            key = (x.device, self._stream())
            ws = self.workspaces.get(key)
            need = max(self.workspace_floats, B * max(split_k, 1) * t_out * N if partials else 0)
            if ws is None or ws[0].numel() < need:     # slabs + zeroed per-tile arrival counters (the library keeps them zero)
                ws = self.workspaces[key] = (torch.empty(need, dtype=torch.float32, device=x.device),
                                             torch.zeros(65536, dtype=torch.int32, device=x.device))
            d.split_k, d.workspace, d.workspace_floats = split_k, ws[0].data_ptr(), ws[0].numel()
            if self.inlaunch_reduce and not partials:
                d.counters, d.counters_len = ws[1].data_ptr(), ws[1].numel()
        else:
            d.split_k, d.workspace, d.workspace_floats = 1, 0, 0
        # algorithmic work: every operand element once (input rows, the weight matrix, the output (+ residual / accumulate))
        nbytes = 4.0 * (B * (t_in >> x_row_shift) * c_in + N * ksize * c_in + B * t_out * N * (1 + (res is not None) + bool(accumulate)))
        return d, out, (ws if (split_k != 1 or partials) else None), B, t_out, N, {"flops": 2.0 * B * t_out * N * ksize * c_in, "bytes": nbytes}

    def conv(self, x, w, bias=None, *, x16=None, out16=None, **kw):
        """y[b,t,n] = epilogue(sum_k sum_ci x[b, t*stride + k*dilation - pad, ci] * w[n, k*c_in + ci]).
        ``x`` is [B, T, C]; ``w`` is [N, ldw] packed (weights.pack_conv).  Keywords: see ``_conv_desc``.
        ``x16``: the same activations as a bf16 / float16 tensor written by the producing kernel -- in the matching 16-bit mode the launch
        then takes the _A16 kernel (no in-register rounding).  ``out16`` = torch.bfloat16 | torch.float16: also return that copy of y."""
        d, out, ws, B, t_out, N, work = self._conv_desc(x, w, bias, **kw)
        y16 = None
        if out16 is not None:
            y16 = _alloc16(out.shape[:-1], N, out16, out.device)
            d.y16, d.y16_bstride, d.ldy16, d.y16_format = y16.data_ptr(), y16.stride(0), y16.stride(1), _fmt16(out16)
        if self._lp_eligible(d, w, work):
            prec = self._to_lp(d, w, work)
            # x16 in this mode's format (bf16x3: a SPLIT16 tensor, recognised by its doubled rows)
            if x16 is not None and ((prec == PREC_BF16X3 and x16.dtype == torch.bfloat16 and x16.shape[-1] == 2 * ((d.c_in + 7) // 8 * 8))
                                    or (prec != PREC_BF16X3 and prec == _fmt16(x16.dtype))):
                self._chk(x16)
                prec = {PREC_BF16: _lib.PREC_BF16_A16, PREC_F16: _lib.PREC_F16_A16, PREC_BF16X3: _lib.PREC_BF16X3_A16}[prec]   # natural-order weight image
                if self.precision == _lib.PREC_F16W2:
                    prec = _lib.PREC_F16W2_A16                                     # ... of (hi, lo) fp16 pairs
                img = self.lp_weight(w, prec)
                d.x, d.w, d.ldx, d.x_bstride = x16.data_ptr(), img.data_ptr(), x16.stride(1), x16.stride(0)
            self._call("svcmi_conv_gemm_lp", ctypes.byref(d), prec, self._stream(), work=work)
        else:
            if (d.flags >> 8) & 15 == 9:          # SVCMI_CONV_TILE_64x128 is a reduced-precision tile: on the fp32 kernel the library chooses
                d.flags &= ~0xF00
            self._call("svcmi_conv_gemm_f32", ctypes.byref(d), self._stream(), work=work)
        if kw.get("partials"):      # [B, split, t_out, N] view of this stream's workspace; valid until the next split-K launch on it
            return ws[0][:B * d.split_k * t_out * N].view(B, d.split_k, t_out, N)
        return out if out16 is None else (out, y16)

    def conv_group(self, problems):
        """Up to 3 convolutions of one geometry in one launch (svcmi_conv_gemm_group_f32).  ``problems``: dicts of ``conv``
        arguments (x, w, bias + keywords; ``out`` required to differ).  Returns the outputs."""
        descs = (ConvDesc * len(problems))()
        outs, works, ws = [], [], []
        for i, pr in enumerate(problems):
            pr = dict(pr)
            w = pr.pop("w")
            d, out, _, _, _, _, wk = self._conv_desc(pr.pop("x"), w, pr.pop("bias", None), split_k=1, **pr)
            descs[i] = d
            outs.append(out)
            works.append(wk)
            ws.append(w)
        total = {"flops": sum(wk["flops"] for wk in works), "bytes": 0.0}
        lp = all(self._lp_eligible(descs[i], ws[i], total) for i in range(len(problems))) and (descs[0].flags >> 8) in (0, 1, 4, 6)
        prec = PREC_F32
        if lp:
            for i in range(len(problems)):
                prec = self._to_lp(descs[i], ws[i], works[i])
        total["bytes"] = sum(wk["bytes"] for wk in works)
        if lp:
            self._call("svcmi_conv_gemm_group_lp", descs, len(problems), prec, self._stream(), work=total)
        else:
            self._call("svcmi_conv_gemm_group_f32", descs, len(problems), self._stream(), work=total)
        return outs

    # ------------------------------------------------------------------ norm / attention
    def layernorm(self, x, gamma=None, beta=None, *, res=None, eps=1e-5, per_batch_affine=False, out=None, out16=None):
        """``out16`` = torch.bfloat16 | torch.float16 | SPLIT16: also return the rows rounded to that format (a following GEMM's 16-bit A operand)."""
        self._chk(x, gamma, beta, res, out)
        B, T, Cc = x.shape
        if out is None:
            out = torch.empty_like(x)
        y16 = _alloc16((B, T), Cc, out16, x.device) if out16 is not None else None
        self._call("svcmi_layernorm_f32", _ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(out), B, T, Cc,
                   x.stride(1), res.stride(1) if res is not None else 0, out.stride(1),
                   (gamma.stride(0) if gamma is not None else beta.stride(0)) if per_batch_affine else 0, eps,
                   _ptr(y16), y16.stride(1) if y16 is not None else 0, _fmt16(out16) if out16 is not None else 0, self._stream())
        return out if out16 is None else (out, y16)

    def channel_norm_gelu(self, x, gamma, beta, eps=1e-5, out=None):
        """GroupNorm(C, C) over time + GELU on time-major x [B, T, C] (hubert/hubert_model.py:78,88)."""
        self._chk(x, gamma, beta, out)
        B, T, Cc = x.shape
        if out is None:
            out = torch.empty_like(x)
        scratch = torch.empty(B * 129 * Cc, dtype=torch.float64, device=x.device)
        self._call("svcmi_channel_norm_gelu_f32", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(scratch), B, T, Cc,
                   x.stride(1), out.stride(1), eps, self._stream())
        return out

    def splitk_layernorm(self, partials, bias, x, gamma, beta, *, eps=1e-5, out=None, out16=None):
        """x += bias + sum_s partials[:, s]  (in place);  returns LayerNorm(x) * gamma + beta.  partials: [B, S, T, C]."""
        self._chk(partials, bias, x, gamma, beta, out)
        B, S, T, Cc = partials.shape
        if out is None:
            out = torch.empty_like(x)
        y16 = _alloc16((B, T), Cc, out16, x.device) if out16 is not None else None
        self._call("svcmi_splitk_layernorm_f32", _ptr(partials), S, _ptr(bias), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(out),
                   B, T, Cc, x.stride(1), out.stride(1), eps, _ptr(y16), y16.stride(1) if y16 is not None else 0,
                   _fmt16(out16) if out16 is not None else 0, self._stream())
        return out if out16 is None else (out, y16)

    def attention(self, qkv, heads, scale, *, rel_k=None, rel_v=None, window=0, lengths=None, out=None, out16=None):
        """qkv: [B, T, 3*C] fused projection (q | k | v).  Returns [B, T, C] (+ its bf16 / float16 copy with ``out16``)."""
        self._chk(qkv, rel_k, rel_v, lengths, out)
        B, T, C3 = qkv.shape
        Cc = C3 // 3
        if out is None:
            out = torch.empty(B, T, Cc, dtype=torch.float32, device=qkv.device)
        base = qkv.data_ptr()
        bs = qkv.stride(0)
        o16 = _alloc16((B, T), Cc, out16, qkv.device) if out16 is not None else None
        self._call("svcmi_attention_f32", base, base + 4 * Cc, base + 8 * Cc, _ptr(out), C3, C3, C3, out.stride(1),
                   bs, bs, bs, out.stride(0), B, T, heads, Cc // heads, scale, _ptr(rel_k), _ptr(rel_v), window,
                   _ptr(lengths), _ptr(o16), o16.stride(1) if o16 is not None else 0, o16.stride(0) if o16 is not None else 0,
                   _fmt16(out16) if out16 is not None else 0, self._stream(),
                   work={"flops": 4.0 * B * T * T * Cc})
        return out if out16 is None else (out, o16)

    def attention16(self, qkv16, heads, scale, *, rel_k=None, rel_v=None, window=0, lengths=None, want_f32=True):
        """qkv16: [B, T, 3*C] bf16 / float16 fused projection (the 16-bit output copy of the QKV GEMM).  Attention on the 16-bit matrix
        cores (svcmi_attention16); returns (o fp32 [B,T,C] or None, o16 [B,T,C] in qkv16's dtype)."""
        self._chk(qkv16, lengths, rel_k, rel_v)
        B, T, C3 = qkv16.shape
        Cc = C3 // 3
        o = torch.empty(B, T, Cc, dtype=torch.float32, device=qkv16.device) if want_f32 else None
        o16 = torch.empty(B, T, Cc, dtype=qkv16.dtype, device=qkv16.device)
        base, esz = qkv16.data_ptr(), 2
        self._call("svcmi_attention16", base, base + esz * Cc, base + 2 * esz * Cc, C3, qkv16.stride(0), _ptr(o), Cc, T * Cc, _ptr(o16), Cc, T * Cc,
                   B, T, heads, Cc // heads, scale, _ptr(rel_k), _ptr(rel_v), window, _ptr(lengths), _fmt16(qkv16.dtype), self._stream(),
                   work={"flops": 4.0 * B * T * T * Cc})
        return o, o16

    # ------------------------------------------------------------------ generator pieces
    def snake_alias(self, x, alpha_log, beta_log, filt, out=None):
        self._chk(x, alpha_log, beta_log, filt, out)
        B, L, Cc = x.shape
        if out is None:
            out = torch.empty_like(x)
        self._call("svcmi_snake_alias_f32", _ptr(x), _ptr(out), _ptr(alpha_log), _ptr(beta_log), _ptr(filt),
                   B, L, Cc, x.stride(1), self._stream(), work={"bytes": 8.0 * B * L * Cc})
        return out

    def snake_alias_group(self, xs, alpha_logs, beta_logs, filt, outs):
        """SnakeAlias of up to 3 same-shape tensors (own alpha / beta each) in one launch."""
        self._chk(*xs, *alpha_logs, *beta_logs, filt, *outs)
        n = len(xs)
        B, L, Cc = xs[0].shape
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        if outs[0].dtype != torch.float32:      # bf16 / float16 outputs: the 16-bit rows a following _A16 GEMM reads
            split = outs[0].dtype == torch.bfloat16 and outs[0].shape[-1] == 2 * xs[0].stride(1)      # SPLIT16 rows [hi: ld | lo: ld]
            self._call("svcmi_snake_alias_group_f32", arr(xs), None, arr(alpha_logs), arr(beta_logs), _ptr(filt), n,
                       B, L, Cc, xs[0].stride(1), arr(outs), _fmt16(SPLIT16 if split else outs[0].dtype), self._stream(),
                       work={"bytes": (8.0 if split else 6.0) * n * B * L * Cc})
            return outs
        self._call("svcmi_snake_alias_group_f32", arr(xs), arr(outs), arr(alpha_logs), arr(beta_logs), _ptr(filt), n,
                   B, L, Cc, xs[0].stride(1), None, 0, self._stream(), work={"bytes": 8.0 * n * B * L * Cc})
        return outs

    def block_mean(self, xs, out=None):
        """((xs[0] + xs[1]) + xs[2]) / len(xs) for contiguous same-shape tensors."""
        self._chk(*xs, out)
        if out is None:
            out = torch.empty_like(xs[0])
        arr = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
        self._call("svcmi_block_mean_f32", arr, len(xs), _ptr(out), xs[0].numel(), self._stream(),
                   work={"bytes": 4.0 * (len(xs) + 1) * xs[0].numel()})
        return out

    def upsample_noise_supported(self, u, cp, cin):
        return bool(self.lib.svcmi_upsample_noise_supported(u, cp, cin))

    def upsample_noise(self, x, up_w, up_b, taps, pad, u, cp, src, nz_w, nz_b, nz_k, nz_stride, nz_pad, y=None):
        """ups[i](x) + noise_convs[i](src) for the narrow stages: x [B, t_in, cin], src [B, L] -> [B, t_in*u, cp].
        With ``y`` given (= ups[i](x) already computed, [B, t_in*u, cp]) only the noise convolution is added."""
        self._chk(x, up_w, up_b, src, nz_w, nz_b, y)
        noise_only = y is not None
        if noise_only:
            B, t_in, cin = y.shape[0], y.shape[1] // u, 4
        else:
            B, t_in, cin = x.shape
            y = torch.empty(B, t_in * u, cp, dtype=torch.float32, device=x.device)
        self._call("svcmi_upsample_noise_f32", 0 if noise_only else _ptr(x), _ptr(up_w), _ptr(up_b), _ptr(src), _ptr(nz_w),
                   _ptr(nz_b), _ptr(y), B, t_in, cin, up_w.shape[1] if up_w is not None else 0, taps, pad, u, cp, src.shape[1],
                   nz_k, nz_stride, nz_pad, nz_w.shape[1], self._stream(), work={"bytes": 4.0 * B * t_in * (cin + u * cp)})
        return y

    def snake_conv_supported(self, c, ld, ksize, dilation):
        return bool(self.lib.svcmi_snake_conv_supported(c, ld, ksize, dilation))

    def snake_conv_preferred(self, c, ld, ksize, dilation):
        return bool(self.lib.svcmi_snake_conv_preferred(c, ld, ksize, dilation))

    def snake_conv(self, x, alpha_log, beta_log, filt, w, bias, *, c, ksize, dilation=1, res=None, alpha=1.0,
                   accumulate=False, out=None):
        """Fused SnakeAlias -> 'same' Conv1d (+ bias + res) * alpha (+= out) for the narrow stages; x [B, L, ld]."""
        self._chk(x, alpha_log, beta_log, filt, w, bias, res, out)
        B, L, ld = x.shape
        if out is None:
            out = torch.empty_like(x)
        self._call("svcmi_snake_conv_f32", _ptr(x), _ptr(w), _ptr(bias), _ptr(res), _ptr(out), _ptr(alpha_log),
                   _ptr(beta_log), _ptr(filt), B, L, c, ld, w.shape[1], ksize, dilation, float(alpha), int(accumulate),
                   self._stream(), work={"flops": 2.0 * B * L * c * c * ksize, "bytes": 8.0 * B * L * c})
        return out

    def snake_post_supported(self, c, ld, ksize):
        return bool(self.lib.svcmi_snake_post_supported(c, ld, ksize))

    def snake_post(self, x, alpha_log, beta_log, filt, w, *, c, ksize):
        """Output layer: tanh(conv_post(SnakeAlias(x))) with one output channel; x [B, L, ld], w [1, >= ksize*ld] -> [B, L]."""
        self._chk(x, alpha_log, beta_log, filt, w)
        B, L, ld = x.shape
        out = torch.empty(B, L, dtype=torch.float32, device=x.device)
        self._call("svcmi_snake_post_f32", _ptr(x), _ptr(w), _ptr(out), _ptr(alpha_log), _ptr(beta_log), _ptr(filt), B, L, c, ld,
                   ksize, self._stream(), work={"flops": 2.0 * B * L * c * ksize, "bytes": 4.0 * B * L * (c + 1)})
        return out

    def snake_conv_group(self, problems, filt, *, c, precision=None):
        """The fused half-step for up to 3 AMP blocks in one launch.  ``problems``: dicts with x, alpha_log, beta_log, w, bias,
        ksize, dilation (1), res (None), alpha (1.0), accumulate (False), out.  ``precision`` "f16" / "f16w2": the convolution on the
        fp16 matrix cores (svcmi_snake_conv_group_lp; same fp32 tensors in and out)."""
        descs = (SnakeConvDesc * len(problems))()
        B, L, ld = problems[0]["x"].shape
        flops = 0.0
        for i, pr in enumerate(problems):
            self._chk(pr["x"], pr["alpha_log"], pr["beta_log"], pr["w"], pr.get("bias"), pr.get("res"), pr["out"])
            d = descs[i]
            d.x, d.w, d.bias, d.res, d.y = _ptr(pr["x"]), _ptr(pr["w"]), _ptr(pr.get("bias")), _ptr(pr.get("res")), _ptr(pr["out"])
            d.alpha_log, d.beta_log = _ptr(pr["alpha_log"]), _ptr(pr["beta_log"])
            d.ldw, d.ksize, d.dilation = pr["w"].shape[1], pr["ksize"], pr.get("dilation", 1)
            d.accumulate, d.alpha = int(pr.get("accumulate", False)), float(pr.get("alpha", 1.0))
            flops += 2.0 * B * L * c * c * pr["ksize"]
        if precision is not None:
            self._call("svcmi_snake_conv_group_lp", descs, len(problems), _ptr(filt), B, L, c, ld, PRECISIONS[precision], self._stream(),
                       work={"flops": flops, "bytes": 8.0 * len(problems) * B * L * c})
        else:
            self._call("svcmi_snake_conv_group_f32", descs, len(problems), _ptr(filt), B, L, c, ld, self._stream(),
                       work={"flops": flops, "bytes": 8.0 * len(problems) * B * L * c})
        return [pr["out"] for pr in problems]

    def pitch2source(self, f0, rand_ini, noise, merge_w, merge_b, hop, sr):
        """f0 [B,T], rand_ini [B,11], noise [B,T*hop,11] -> source [B, T*hop]."""
        self._chk(f0, rand_ini, noise, merge_w)
        B, T = f0.shape
        prefix = torch.empty(B, T, 11, dtype=torch.float64, device=f0.device)
        self._call("svcmi_pitch_prefix_f64", _ptr(f0), _ptr(rand_ini), _ptr(prefix), B, T, hop, float(sr), self._stream())
        out = torch.empty(B, T * hop, dtype=torch.float32, device=f0.device)
        self._call("svcmi_pitch_source_f32", _ptr(f0), _ptr(prefix), _ptr(noise), _ptr(merge_w), float(merge_b),
                   _ptr(out), B, T, hop, float(sr), self._stream())
        return out

    # ------------------------------------------------------------------ whisper log-mel front-end
    def reflect_pad(self, x, pad):
        self._chk(x)
        B, n = x.shape
        y = torch.empty(B, n + 2 * pad, dtype=torch.float32, device=x.device)
        self._call("svcmi_reflect_pad_f32", _ptr(x), _ptr(y), B, n, pad, self._stream())
        return y

    def power_spectrum(self, ri, nbins, half):
        self._chk(ri)
        B, T, ld = ri.shape
        p = torch.empty(B, T, half, dtype=torch.float32, device=ri.device)
        self._call("svcmi_power_spectrum_f32", _ptr(ri), _ptr(p), B * T, nbins, half, ld, half, self._stream())
        return p

    def logmel_finish(self, mel_power):
        self._chk(mel_power)
        B, T, Cc = mel_power.shape
        scratch = torch.empty(B * 64, dtype=torch.float32, device=mel_power.device)
        out = torch.empty(B, Cc, T, dtype=torch.float32, device=mel_power.device)
        self._call("svcmi_logmel_finish_f32", _ptr(mel_power), _ptr(scratch), _ptr(out), B, T, Cc, self._stream())
        return out

    # ------------------------------------------------------------------ CREPE glue
    def crepe_frames(self, audio, hop, frame0, frames, ld=1536):
        """audio [n] -> normalised, first-layer-padded frames [frames, ld] (crepe/core.py:664-703)."""
        self._chk(audio)
        out = torch.empty(frames, ld, dtype=torch.float32, device=audio.device)
        self._call("svcmi_crepe_frames_f32", _ptr(audio), audio.numel(), hop, frame0, frames, _ptr(out), ld, self._stream())
        return out

    def bn_maxpool2(self, x, scale, shift, out16=None):
        """x [B, T, C] (T even) -> max over row pairs of x*scale + shift: [B, T/2, C] (crepe/model.py:128-134).  ``out16``: also return
        the 16-bit copy (torch.bfloat16 | torch.float16 | SPLIT16) the next layer's GEMM reads."""
        self._chk(x, scale, shift)
        B, T, Cc = x.shape
        y = torch.empty(B, T // 2, Cc, dtype=torch.float32, device=x.device)
        y16 = _alloc16((B, T // 2), Cc, out16, x.device) if out16 is not None else None
        self._call("svcmi_bn_maxpool2_f32", _ptr(x), _ptr(scale), _ptr(shift), _ptr(y), B * (T // 2), Cc, x.stride(1), y.stride(1),
                   _ptr(y16), y16.stride(1) if y16 is not None else 0, _fmt16(out16) if out16 is not None else 0,
                   self._stream(), work={"bytes": 6.0 * B * T * Cc})
        return y if out16 is None else (y, y16)

    def act16_format(self):
        """The 16-bit activation format of the current precision mode (``out16`` of the producers), None in fp32 (f16w2 = fp16 rows, as
        fmt16_of() on the C side)."""
        return {PREC_BF16: torch.bfloat16, PREC_F16: torch.float16, _lib.PREC_F16W2: torch.float16, PREC_BF16X3: SPLIT16}.get(self.precision)

    def viterbi_decode(self, prob, log_trans, batch_frames, minidx, maxidx, band=0):
        """prob [frames, 360] (sigmoid outputs), log_trans [360, 360] float64 -> decoded bins int32 [frames].
        ``band`` > 0: entries with |i - j| > band are one constant (see ``transition_band``)."""
        self._chk(prob, log_trans)
        Fr = prob.shape[0]
        lp = torch.empty(Fr * 360, dtype=torch.float32, device=prob.device)
        ptr = torch.empty(Fr * 360, dtype=torch.int16, device=prob.device)
        path = torch.empty(Fr, dtype=torch.int32, device=prob.device)
        self._call("svcmi_viterbi_decode", _ptr(prob), _ptr(log_trans), _ptr(lp), _ptr(ptr), _ptr(path), Fr, batch_frames,
                   minidx, maxidx, band, self._stream())
        return path

    # ------------------------------------------------------------------ feature retrieval
    def row_sqnorm(self, x):
        """x [rows, d] -> [rows] squared L2 norms."""
        self._chk(x)
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        self._call("svcmi_row_sqnorm_f32", _ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(out), self._stream())
        return out

    def knn_blend(self, x, bank, dots, bank_sq, k, ratio, out=None):
        """x [t, d], bank [n, d], dots [t, >= n] = x @ bank.T -> (1 - ratio) * x + ratio * weighted k-NN mean."""
        self._chk(x, bank, dots, bank_sq, out)
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        self._call("svcmi_knn_blend_f32", _ptr(x), x.stride(0), _ptr(bank), bank.stride(0), _ptr(dots), dots.stride(0),
                   _ptr(bank_sq), _ptr(out), out.stride(0), x.shape[0], bank.shape[0], x.shape[1], k, ratio, self._stream())
        return out

    def ivf_assign(self, x, dots, cent_sq, want_dist=False):
        """x [t, d], dots [t, >= nlist] = x @ centroids.T -> int32 [t] nearest centroid (+ its BLAS-form distance)."""
        self._chk(x, dots, cent_sq)
        t = x.shape[0]
        assign = torch.empty(t, dtype=torch.int32, device=x.device)
        dist = torch.empty(t, dtype=torch.float32, device=x.device) if want_dist else None
        self._call("svcmi_ivf_assign_f32", _ptr(x), x.stride(0), _ptr(dots), dots.stride(0), _ptr(cent_sq), t, cent_sq.shape[0],
                   x.shape[1], _ptr(assign), _ptr(dist) if want_dist else 0, self._stream())
        return (assign, dist) if want_dist else assign

    def ivf_blend(self, x, assign, list_off, bank, k, ratio, out=None, want_neighbours=False):
        """x [t, d], assign int32 [t], list_off int32 [nlist + 1], bank [n, d] grouped by cell -> blended [t, d]
        (+ int32 [t, k] bank rows, float [t, k] distances)."""
        self._chk(x, bank, out)
        t = x.shape[0]
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        idx = torch.empty(t, k, dtype=torch.int32, device=x.device) if want_neighbours else None
        dist = torch.empty(t, k, dtype=torch.float32, device=x.device) if want_neighbours else None
        self._call("svcmi_ivf_blend_f32", _ptr(x), x.stride(0), _ptr(assign), _ptr(list_off), _ptr(bank), bank.stride(0), _ptr(out),
                   out.stride(0), t, x.shape[1], k, ratio, _ptr(idx) if want_neighbours else 0, _ptr(dist) if want_neighbours else 0,
                   self._stream())
        return (out, idx, dist) if want_neighbours else out

    def segment_mean(self, x, order, seg_off, out):
        """out[c] = mean of x[order[seg_off[c]:seg_off[c+1]]] (empty segments untouched)."""
        self._chk(x, out)
        self._call("svcmi_segment_mean_f32", _ptr(x), x.stride(0), _ptr(order), _ptr(seg_off), _ptr(out), out.stride(0),
                   seg_off.shape[0] - 1, x.shape[1], self._stream())
        return out

    def source2wav(self, x):
        self._chk(x)
        x = x.contiguous()
        out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
        self._call("svcmi_source2wav_i16", _ptr(x), _ptr(out), x.numel(), self._stream())
        return out

    # ------------------------------------------------------------------ flow / prior glue
    def wn_gate(self, a, out=None, bias=None):
        """a [B, T, 2H] -> tanh(a[..., :H]) * sigmoid(a[..., H:]); or a [B, S, T, 2H] = split-K slabs of the in_layer
        convolution (``conv(partials=True)``) summed here together with ``bias``."""
        self._chk(a, out, bias)
        if a.dim() == 4:
            B, S, T, H2 = a.shape
            assert a.is_contiguous()
        else:
            (B, T, H2), S = a.shape, 1
            assert a.stride(0) == T * a.stride(1)
        H = H2 // 2
        if out is None:
            out = torch.empty(B, T, H, dtype=torch.float32, device=a.device)
        self._call("svcmi_wn_gate_f32", _ptr(a), _ptr(bias), _ptr(out), B, T, H, a.stride(-2), out.stride(1), S, self._stream())
        return out

    def wn_update(self, rs, x, skip, lengths, first, last):
        self._chk(rs, x, skip, lengths)
        B, T, H = skip.shape
        self._call("svcmi_wn_update_f32", _ptr(rs), _ptr(x), _ptr(skip), _ptr(lengths), B, T, H, rs.stride(1),
                   int(first), int(last), self._stream())

    def coupling_pre(self, x, x0_off, ms_vs, lengths, half):
        self._chk(x, ms_vs, lengths)
        B, T, _ = x.shape
        out = torch.empty(B, T, half, dtype=torch.float32, device=x.device)
        self._call("svcmi_coupling_pre_f32", _ptr(x), x.stride(1), x0_off, _ptr(ms_vs), _ptr(out), half, _ptr(lengths),
                   B, T, half, self._stream())
        return out

    def coupling_post(self, x, x1_off, m, ms_vs, lengths, half):
        self._chk(x, m, ms_vs, lengths)
        B, T, _ = x.shape
        self._call("svcmi_coupling_post_f32", _ptr(x), x.stride(1), x1_off, _ptr(m), m.stride(1), _ptr(ms_vs),
                   _ptr(lengths), B, T, half, self._stream())

    def embed_pitch(self, x, pit, emb, lengths):
        self._chk(x, pit, emb, lengths)
        B, T, Cc = x.shape
        self._call("svcmi_embed_pitch_f32", _ptr(x), x.stride(1), _ptr(pit), _ptr(emb), _ptr(lengths), B, T, Cc, self._stream())

    def sample_prior(self, stats, noise_ncl, lengths):
        self._chk(stats, noise_ncl, lengths)
        B, T, I2 = stats.shape
        I = I2 // 2
        z = torch.empty(B, T, I, dtype=torch.float32, device=stats.device)
        self._call("svcmi_sample_prior_f32", _ptr(stats), stats.stride(1), _ptr(noise_ncl), _ptr(lengths), _ptr(z), I,
                   B, T, I, self._stream())
        return z

    def ncl_to_nlc(self, x, add=None, add_scale=0.0, ld=None):
        self._chk(x, add)
        B, Cc, T = x.shape
        ld = ld or Cc
        y = torch.empty(B, T, ld, dtype=torch.float32, device=x.device) if ld == Cc else \
            torch.zeros(B, T, ld, dtype=torch.float32, device=x.device)
        self._call("svcmi_ncl_to_nlc_f32", _ptr(x), _ptr(add), float(add_scale), _ptr(y), B, Cc, T, ld, self._stream())
        return y

    def nlc_to_ncl(self, x, c=None):
        self._chk(x)
        B, T, ld = x.shape
        c = c or ld
        y = torch.empty(B, c, T, dtype=torch.float32, device=x.device)
        self._call("svcmi_nlc_to_ncl_f32", _ptr(x), x.stride(1), _ptr(y), B, c, T, self._stream())
        return y
