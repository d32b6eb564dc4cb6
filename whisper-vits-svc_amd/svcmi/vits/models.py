"""Drop-in ``SynthesizerInfer`` (reference: vits/models.py:211-256) running on the svcmi HIP kernels.

Same constructor, ``state_dict``/``load_state_dict`` key names, ``eval``/``to``, ``pitch2source``,
``source2wav`` and ``inference`` signatures and tensor layouts as the reference class, so
``svc_inference.py``-style callers need no change.  Differences, all additive:
  * the three stochastic draws of the path (vits/models.py:51, vits_decoder/nsf.py:232-235,311) can be
    passed explicitly (``noise=...``) so runs are reproducible / comparable; when omitted they are drawn
    with torch's generator on the device, as the reference does;
  * there is no autograd and no CPU path: without the gfx950 library construction of ``Ops`` raises.
Internally everything is time-major ``[B, T, C]`` fp32.
"""
import math
from collections import OrderedDict

import torch

from .. import weights as PW
from ..ops import ACT_MISH, ACT_RELU, ACT_TANH, Ops
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
        self.parallel_blocks = False     # fallback for stages the grouped scheme does not fit: AMP blocks on forked HIP streams
        self.grouped_blocks = True       # the AMP blocks of a stage advance in lock-step through grouped launches
        self._stop_after = None          # tuning aid (scripts/stage_times.py), never set in production
        # GEMM operand precision of prior encoder / flow / generator: None = fp32 (parity default), "bf16x3" / "bf16" /
        # "f16" (Ops.use_precision).  Element-wise kernels, softmax, LayerNorm, SnakeAlias and accumulation stay fp32.
        self.precision = None
        # Streaming decoder (BASELINE.json configs[4]): None = the generator sees a whole synthesis chunk;
        # N = it runs over time tiles of N frames plus a STREAM_HALO-frame halo on each side that is computed and discarded.
        # The generator's exact receptive field is < 31 frames (see STREAM_HALO) and every kernel's arithmetic for an output row is
        # independent of the row's position in a launch, so the kept samples are BIT-IDENTICAL for every N (split-K is pinned
        # off in this mode: its slice count would otherwise follow the problem size).  The reference's own chunk seams
        # (svc_inference.py:94-131) are untouched: tiling happens inside a chunk.
        self.stream_frames = None
        # svc_infer: synthesis chunks of one clip in flight on this many HIP streams (1 = one after the other, the reference's order);
        # results do not depend on it (bit-identical, tests/test_gpu_engine.py); a 3-minute song: 59.3 -> 48.4 ms with 3 streams
        # (profiles/r02w_chunk_streams.log).
        self.chunk_streams = 3
        self._streams, self._streams_dev = None, None

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
        self._w = None
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
        self._w = None
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
        self._w, self._device = weights, torch.device(device)
        return self

    def _weights(self):
        if self._w is None:
            dev = self._device or torch.device("cuda" if self.ops.on_gpu else "cpu")
            self._w = PW.VitsWeights(self.state_dict(), self.hp, dev)
            self._device = dev
        return self._w

    # ------------------------------------------------------------------ reference methods
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
        src = ops.pitch2source(f0, rand_ini, nz, w.merge_w, w.merge_b, w.hop, float(self.hp.data.sampling_rate))
        return src.view(B, 1, L)

    def source2wav(self, source):
        """-> int16 numpy (vits_decoder/generator.py:167-173)."""
        return self.ops.source2wav(source.to(self._device, torch.float32).squeeze()).cpu().numpy()

    @torch.no_grad()
    def inference(self, ppg, vec, pit, spk, ppg_l, source, noise=None, return_parts=False):
        """vits/models.py:251-256.  ppg [B,T,ppg_dim], vec [B,T,vec_dim], pit [B,T] Hz, spk [B,spk_dim],
        ppg_l int64 [B], source [B,1,hop*T] -> waveform [B,1,hop*T] (device tensor).
        ``noise``: the randn_like(m) of vits/models.py:51 in the reference layout [B,inter,T]."""
        w, ops, dev = self._weights(), self.ops, None
        dev = self._device
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
        with ops.use_precision(self.precision):
            z_p = self._prior_encoder(w, ops, ppg, vec, pit, lengths, noise)
            z = self._flow_reverse(w, ops, z_p.clone() if return_parts else z_p, spk, lengths)
            o = self._generator(w, ops, z, spk, source)
        if return_parts:
            return o, {"z_p": ops.nlc_to_ncl(z_p), "z": ops.nlc_to_ncl(z)}
        return o

    __call__ = inference

    @torch.no_grad()
    def inference_ppg50(self, ppg50, vec, pit, spk, lengths, source, noise=None):
        """Same as ``inference`` but takes the Whisper PPG at its native 50 fps ([B, T/2, ppg_dim]) and fuses
        the ``np.repeat(ppg, 2, 0)`` of svc_inference.py:175-177 into the first conv's loads.  All arguments
        must already be device tensors (fp32 / int32 lengths); nothing here synchronises, so the call can be
        captured in a HIP graph."""
        w, ops = self._weights(), self.ops
        B, T = pit.shape
        if noise is None:
            noise = torch.randn(B, w.I, T, device=pit.device)
        with ops.use_precision(self.precision):
            z_p = self._prior_encoder(w, ops, ppg50, vec, pit, lengths, noise, ppg_row_shift=1)
            if self._stop_after == "prior":      # scripts/stage_times.py: truncated pipelines for in-situ stage timing
                return z_p
            z = self._flow_reverse(w, ops, z_p, spk, lengths)
            if self._stop_after == "flow":
                return z
            return self._generator(w, ops, z, spk, source.view(B, T * w.hop))

    # ------------------------------------------------------------------ stages (time-major)
    def _prior_encoder(self, w, ops, ppg, vec, pit, lengths, noise, ppg_row_shift=0):
        """TextEncoder.forward, vits/models.py:39-52 + attentions.Encoder.forward, attentions.py:60-72."""
        x = ops.conv(ppg, w.pre_w, w.pre_b, ksize=5, pad=2, lengths=lengths, mask_out=True, x_row_shift=ppg_row_shift)
        ops.conv(vec, w.hub_w, w.hub_b, ksize=5, pad=2, res=x, lengths=lengths, mask_out=True, out=x)
        ops.embed_pitch(x, pit, w.pit_emb, lengths)
        scale = 1.0 / math.sqrt(w.H // w.n_heads)
        Bq, Tq = x.shape[0], x.shape[1]
        f2_nk = (w.enc[0]["f2_w"].shape[1] + 31) // 32      # K-steps of the second FFN convolution
        f2_blocks = Bq * ((Tq + 63) // 64) * ((w.H + 63) // 64)
        f2_split = 1 if f2_blocks >= 256 else max(1, min(f2_nk // 10, (288 + f2_blocks // 2) // f2_blocks))
        for L in w.enc:
            qkv = ops.conv(x, L["qkv_w"], L["qkv_b"])
            a = ops.attention(qkv, w.n_heads, scale, rel_k=L["rel_k"], rel_v=L["rel_v"], window=K.ENC_WINDOW, lengths=lengths)
            y = ops.conv(a, L["o_w"], L["o_b"])
            x = ops.layernorm(x, L["g1"], L["b1"], res=y)
            pl = (K.ENC_FFN_KERNEL - 1) // 2
            h = ops.conv(x, L["f1_w"], L["f1_b"], ksize=K.ENC_FFN_KERNEL, pad=pl, act=ACT_RELU, lengths=lengths, mask_in=True, mask_out=True)
            # second FFN convolution: raw split-K slabs -> one launch that sums them with the bias and the residual and
            # applies norm_layers_2 (the `* x_mask` of attentions.py:209 only affects rows past the length, which no valid
            # row ever reads: keys are masked in the attention, inputs in the convolutions)
            p = ops.conv(h, L["f2_w"], None, ksize=K.ENC_FFN_KERNEL, pad=pl, partials=True, split_k=f2_split)
            x = ops.splitk_layernorm(p, L["f2_b"], x, L["g2"], L["b2"])
        stats = ops.conv(x, w.proj_w, w.proj_b, lengths=lengths, mask_in=True, mask_out=True)
        return ops.sample_prior(stats, noise, lengths)

    def _flow_reverse(self, w, ops, x, spk, lengths):
        """ResidualCouplingBlock.forward(reverse=True), vits/models.py:89-94; layers vits/modules.py:288-321,178-203.
        ``x`` [B,T,I] is updated in place."""
        B, T, _ = x.shape
        spk3 = spk.view(B, 1, -1)
        half, H = w.half, w.H
        # WN state as rows of (h | skip): the res_skip convolution then does `h = (h + rs[:, :H]) * mask; skip += rs[:, H:]`
        # (modules.py:196-203) in its own epilogue (ACCUMULATE | MASK_OUT into the 2H-wide row), and the in_layer convolution
        # hands its raw split-K slabs to the gate kernel, which adds them, the bias, and applies tanh * sigmoid: 3 launches
        # per WN layer.  Masking `skip` every layer instead of once at the end only touches rows past the length.
        hs = torch.empty(B, T, 2 * H, dtype=torch.float32, device=x.device)
        h, skip = hs[:, :, :H], hs[:, :, H:]
        nk = (K.FLOW_KERNEL * H + 31) // 32
        blocks = B * ((T + 63) // 64) * ((2 * H + 63) // 64)
        split = 1 if blocks >= 256 else max(1, min(nk // 8, (288 + blocks // 2) // blocks))
        for Lr in w.flow:
            msvs = ops.conv(spk3, Lr["snac_w"], Lr["snac_b"]).view(B, 2 * half)
            x0n = ops.coupling_pre(x, Lr["x0_off"], msvs, lengths, half)
            ops.conv(x0n, Lr["pre_w"], Lr["pre_b"], lengths=lengths, mask_out=True, out=hs)
            n = len(Lr["wn"])
            for l, Wl in enumerate(Lr["wn"]):
                a = ops.conv(h, Wl["in_w"], None, ksize=K.FLOW_KERNEL, pad=(K.FLOW_KERNEL - 1) // 2, ldx=2 * H, c_in=H,
                             partials=True, split_k=split)
                acts = ops.wn_gate(a, bias=Wl["in_b"])
                ops.conv(acts, Wl["rs_w"], Wl["rs_b"], lengths=lengths, mask_out=True, accumulate=True,
                         out=skip if l == n - 1 else hs)
            m = ops.conv(skip, Lr["post_w"], Lr["post_b"], lengths=lengths, mask_out=True, ldx=2 * H, c_in=H)
            ops.coupling_post(x, Lr["x1_off"], m, msvs, lengths, half)
        return x

    def _block_streams(self, n):
        dev = self._device
        if self._streams is None or len(self._streams) < n or self._streams_dev != dev:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
            self._streams_dev = dev
        return self._streams[:n]

    def _amp_block(self, w, ops, st, blk, y, acc, bufs, j, nb, done=None):
        """AMPBlock.forward (vits_decoder/bigv.py:50-58) + the (sum of blocks)/nb of generator.py:188-194:
        for d in dilations: x = x + conv2(act(conv1_d(act(x)))); the last iteration lands in ``acc``.
        A generator: it yields once, right before the final convolution -- everything before that point is independent
        of the other blocks of the stage; the final `acc (+)= ...` waits for ``done[-1]`` (the previous block's event)."""
        xj, tmp, tmp2 = bufs
        k = blk["k"]
        xc = y
        # narrow stages: SnakeAlias + conv as one kernel (csrc/amp_fused.hip); wide stages: two kernels
        fused = all(ops.snake_conv_preferred(st["c"], st["cp"], k, d) for d in blk["d"])
        for q, d in enumerate(blk["d"]):
            last = q == len(blk["d"]) - 1
            if fused:
                b = ops.snake_conv(xc, blk["a1"][q][0], blk["a1"][q][1], w.filt, blk["c1"][q][0], blk["c1"][q][1],
                                   c=st["c"], ksize=k, dilation=d, out=tmp)
                a2 = None
            else:
                a = ops.snake_alias(xc, blk["a1"][q][0], blk["a1"][q][1], w.filt, out=tmp)
                b = ops.conv(a, blk["c1"][q][0], blk["c1"][q][1], ksize=k, dilation=d, pad=(k * d - d) // 2, out=tmp2)
                a2 = ops.snake_alias(b, blk["a2"][q][0], blk["a2"][q][1], w.filt, out=a)
            if last:
                yield
                if done:
                    torch.cuda.current_stream().wait_event(done[-1])       # acc += ... in block order
            out, alpha, accum = (acc, 1.0 / nb, j > 0) if last else (xj, 1.0, False)
            if fused:
                ops.snake_conv(b, blk["a2"][q][0], blk["a2"][q][1], w.filt, blk["c2"][q][0], blk["c2"][q][1],
                               c=st["c"], ksize=k, res=xc, alpha=alpha, accumulate=accum, out=out)
            else:
                ops.conv(a2, blk["c2"][q][0], blk["c2"][q][1], ksize=k, pad=(k - 1) // 2, res=xc,
                         alpha=alpha, accumulate=accum, out=out)
            xc = xj

    def _amp_stage_grouped(self, w, ops, st, y, acc):
        """The nb AMP blocks of a wide stage in lock-step: at every step the blocks' activations go out as ONE grouped
        SnakeAlias launch and their convolutions as ONE grouped GEMM launch (3x the blocks per grid, longest K first), then
        `acc = ((o_0 + o_1) + o_2) / nb` exactly as generator.py:188-194 sums them.  12 + 1 launches per stage instead of
        36-48, and no reliance on multi-stream concurrency.  Returns False when the stage does not fit the scheme."""
        blocks = st["blocks"]
        nb = len(blocks)
        nd = len(blocks[0]["d"])
        if not (2 <= nb <= 3) or any(len(b["d"]) != nd for b in blocks) or y.shape[2] % 4:
            return False
        fused = [ops.snake_conv_preferred(st["c"], st["cp"], b["k"], d) for b in blocks for d in b["d"]]
        if any(fused) and not all(fused):
            return False
        xj = [torch.empty_like(y) for _ in range(nb)]
        t1 = [torch.empty_like(y) for _ in range(nb)]
        t2 = [torch.empty_like(y) for _ in range(nb)]
        xc = [y] * nb
        for q in range(nd if all(fused) else 0):      # narrow stages: activation + convolution are one VALU kernel (csrc/amp_fused.hip)
            ops.snake_conv_group([dict(x=xc[j], alpha_log=b["a1"][q][0], beta_log=b["a1"][q][1], w=b["c1"][q][0], bias=b["c1"][q][1],
                                       ksize=b["k"], dilation=b["d"][q], out=t1[j]) for j, b in enumerate(blocks)], w.filt, c=st["c"])
            outs = t2 if q == nd - 1 else xj
            ops.snake_conv_group([dict(x=t1[j], alpha_log=b["a2"][q][0], beta_log=b["a2"][q][1], w=b["c2"][q][0], bias=b["c2"][q][1],
                                       ksize=b["k"], res=xc[j], out=outs[j]) for j, b in enumerate(blocks)], w.filt, c=st["c"])
            xc = outs
        for q in range(0 if all(fused) else nd):
            ops.snake_alias_group(xc, [b["a1"][q][0] for b in blocks], [b["a1"][q][1] for b in blocks], w.filt, t1)
            ops.conv_group([dict(x=t1[j], w=b["c1"][q][0], bias=b["c1"][q][1], ksize=b["k"], dilation=b["d"][q],
                                 pad=(b["k"] * b["d"][q] - b["d"][q]) // 2, out=t2[j]) for j, b in enumerate(blocks)])
            ops.snake_alias_group(t2, [b["a2"][q][0] for b in blocks], [b["a2"][q][1] for b in blocks], w.filt, t1)
            outs = t2 if q == nd - 1 else xj          # t2 is free again once the second activation has read it
            ops.conv_group([dict(x=t1[j], w=b["c2"][q][0], bias=b["c2"][q][1], ksize=b["k"], pad=(b["k"] - 1) // 2,
                                 res=xc[j], out=outs[j]) for j, b in enumerate(blocks)])
            xc = outs
        ops.block_mean(xc, out=acc)
        return True

    # Halo of the streaming decoder, in frames.  SURVEY.md A.4 measured an EFFECTIVE receptive field of -23.3 .. +23.7 frames
    # (fp64 perturbation test); the exact support of the FIR chain is wider, because the Kaiser tails it ignores are ~1e-3:
    # per side, in output samples: conv_pre 3 frames = 960; ups 480 + 64 + 16 + 4 + 2; an AMP block with k = 11 adds
    # (5 + 15 + 25) + 3 * 5 + 6 * 6 = 96 samples at its stage's rate = 96 * (64 + 16 + 4 + 2 + 1) = 8352; output layer 9:
    # 9887 samples = 30.9 frames.  24 frames left a 9e-8 leak (measured, MI355X); 32 makes tiling bit-exact.
    STREAM_HALO = 32

    def _generator(self, w, ops, z, spk, source):
        """The generator over a whole chunk, or -- with ``stream_frames`` -- over time tiles with a discarded halo."""
        S, H = self.stream_frames, self.STREAM_HALO
        if not S:
            return self._generator_tile(w, ops, z, spk, source, 0)
        B, T, _ = z.shape
        hop = w.hop
        out = torch.empty(B, 1, T * hop, dtype=torch.float32, device=z.device)
        for t0 in range(0, T, S):
            a, b, n = max(0, t0 - H), min(T, t0 + S + H), min(S, T - t0)
            o = self._generator_tile(w, ops, z[:, a:b].contiguous(), spk, source[:, a * hop:b * hop].contiguous(), 1)
            out[:, :, t0 * hop:(t0 + n) * hop] = o[:, :, (t0 - a) * hop:(t0 - a + n) * hop]
        return out

    def _generator_tile(self, w, ops, z, spk, source, split_k):
        """Generator.inference, vits_decoder/generator.py:175-200 (+ SpeakerAdapter :36-47, AMPBlock bigv.py:50-58).
        z [B,T,U] time-major, source [B, hop*T] -> [B,1,hop*T].  ``split_k``: 0 = library heuristic, 1 = off (streaming)."""
        B, T, U = z.shape
        sb = ops.conv(spk.view(B, 1, -1), w.ad_w, w.ad_b).view(B, 2 * U)
        x = ops.layernorm(z, sb[:, :U], sb[:, U:], per_batch_affine=True)
        x = ops.conv(x, w.pre_conv_w, w.pre_conv_b, ksize=7, pad=3, act=ACT_MISH, split_k=split_k)
        if self._stop_after == "gen_pre":
            return x
        for st in w.stages:
            t_in = x.shape[1]
            if ops.upsample_noise_supported(st["u"], st["cp"], x.shape[2]):
                # narrowest stages: the source convolution (and at 10 channels the transposed convolution too) is a pure
                # stream -- one VALU kernel instead of padded GEMM launches (105 / 63 us for < 0.1 GFLOP)
                fuse_up = st["cp"] <= 12
                y = None if fuse_up else ops.conv(x, st["up_w"], st["up_b"], ksize=st["up_taps"], pad=st["up_pad"],
                                                  t_out=t_in, split_k=split_k).view(B, t_in * st["u"], st["cp"])
                y = ops.upsample_noise(x, st["up_w"], st["up_b"], st["up_taps"], st["up_pad"], st["u"], st["cp"], source,
                                       st["nz_w"], st["nz_b"], st["nz_k"], st["nz_stride"], st["nz_pad"], y=y)
            else:
                y = ops.conv(x, st["up_w"], st["up_b"], ksize=st["up_taps"], pad=st["up_pad"], t_out=t_in, split_k=split_k)
                y = y.view(B, t_in * st["u"], st["cp"])
                ops.conv(source, st["nz_w"], st["nz_b"], ksize=st["nz_k"], stride=st["nz_stride"], pad=st["nz_pad"],
                         c_in=1, ldx=1, t_in=source.shape[1], t_out=y.shape[1], accumulate=True, out=y,
                         x_bstride=source.stride(0), split_k=split_k)
            acc = torch.empty_like(y)
            nb = len(st["blocks"])
            # The nb AMP blocks of a stage (generator.py:188-194) only share their input; each runs its 3 iterations as
            # an independent chain.  On the GPU they go to separate HIP streams (forked from / joined to the current
            # one, also under graph capture) so that the many medium-sized launches of a stage overlap; the
            # `acc (+)= (conv + x)/nb` of block j waits for block j-1's, which keeps the summation order fixed.
            if self.grouped_blocks and self._amp_stage_grouped(w, ops, st, y, acc):
                x = acc
                if self._stop_after == ("stage", w.stages.index(st)):
                    return x
                continue
            streams = self._block_streams(nb) if (self.parallel_blocks and ops.on_gpu) else None
            bufs = [tuple(torch.empty_like(y) for _ in range(3)) for _ in range(nb if streams else 1)]
            main = torch.cuda.current_stream() if streams else None
            done = []
            if not streams:
                for j, blk in enumerate(st["blocks"]):
                    for _ in self._amp_block(w, ops, st, blk, y, acc, bufs[0], j, nb):
                        pass
            else:
                # heads (all but the final convolution) first, then the tails in block order, which fixes the summation
                # order of `acc`
                chains = {}
                for j in range(nb):
                    streams[j].wait_stream(main)
                    with torch.cuda.stream(streams[j]):
                        chains[j] = self._amp_block(w, ops, st, st["blocks"][j], y, acc, bufs[j], j, nb, done=done)
                        next(chains[j])
                for j in range(nb):
                    with torch.cuda.stream(streams[j]):
                        for _ in chains[j]:
                            pass
                        ev = torch.cuda.Event()
                        ev.record(streams[j])
                        done.append(ev)
            if streams:
                for sj in streams:
                    main.wait_stream(sj)
            x = acc
            if self._stop_after == ("stage", w.stages.index(st)):
                return x
        c_last = w.stages[-1]["c"]
        if ops.snake_post_supported(c_last, x.shape[2], 7) and w.post_w.shape[1] >= 7 * x.shape[2]:
            return ops.snake_post(x, w.post_a[0], w.post_a[1], w.filt, w.post_w, c=c_last, ksize=7).view(B, 1, -1)
        a = ops.snake_alias(x, w.post_a[0], w.post_a[1], w.filt)
        o = ops.conv(a, w.post_w, None, ksize=7, pad=3, act=ACT_TANH, n_out=1)
        return o.view(B, 1, -1)
