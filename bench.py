#!/usr/bin/env python
"""bench.py -- end-to-end SVC throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM.  The default (the judged line) is
BASELINE.json configs[1]:
    mel [1,80,1000] --(+0.1*randn)--> Whisper-24L encoder --> PPG 50 fps --(x2 repeat fused)-->
    prior encoder -> reverse flow -> NSF-BigVGAN (incl. pitch2source) --> 32 kHz waveform [1,1,320000] in HBM,
fp32, random-init weights of the named architecture (no checkpoints / network), the path's stochastic draws made on the
device inside the step.  `--config` selects the other BASELINE.json configurations (same JSON schema, `config.workload` names it):
    2  1 GPU: batch 16 x 10 s clips, flow + decoder only (pre-extracted PPG / F0), bf16 GEMM operands
    3  512 x 10 s utterances sharded over the ranks (64 per GPU at 8), batches of 16, full pipeline incl. Whisper; a step is
       one pass over this rank's shard, so the total work is fixed: "scaling": "strong"
    4  30 s clips per rank through the reference schedule: two 15 s Whisper windows (Tw = 750 each), T = 3000 frames -> synthesis
       chunks [0,2510) / [2490,3000) with the halo trim of svc_inference.py:101-131; fp16 GEMM operands + 16-bit activations + fp16
       attention in both networks, like the reference's .half() accelerator path
Clips in flight (`--inflight`, default 4 for config 1): a step is still ONE batch-1 clip through the whole path, but the K timed steps are
replayed round-robin on 4 lanes (HIP stream + captured HIP graph + own static buffers each, svcmi/lanes.py), so that one clip's
latency-bound launches run beside another clip's Whisper GEMMs; `ms_per_step` = wall time / K (the throughput figure), and
`config.single_stream` carries the same K steps on ONE lane (= the latency of a clip, the round-1 way of running this line).
With N GPUs every rank runs its own clips (weak scaling; config 3: strong); rank 0 makes and PACKS the weights once and all
ranks receive the packed arena through one RCCL broadcast per model; the hot loop has no collective.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (the implicit-GEMM conv/linear kernel):
algorithmic FLOPs of its launches / their HIP-event durations, measured on the launch stream in an instrumented pass of the
same step (`frac`), beside the same ratio from the newest committed rocprofv3 summary (`frac_rocprof`).  `cpu_baseline`
times the oracle (CPU port of the reference path) on one clip of the workload on this host's cores.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

if not os.environ.get("GPU_MAX_HW_QUEUES", "").isdigit() or int(os.environ["GPU_MAX_HW_QUEUES"]) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"         # one hardware queue per lane of clips in flight (svcmi/lanes.py); read at HIP init

import torch  # noqa: E402

from workload.stamp import csrc_sha  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak (no TF32/xf32 on gfx950)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 / fp16 matrix peak (the 5 PF headline figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ workloads
def synth_features(T, hp, seeds, device):
    """mel / vec / F0 / speaker of len(seeds) clips of T frames, resident on ``device`` (the noise draws of the path are made
    inside the step).  Same recipe as workload.inputs.synth_clip, without its 14 MB-per-clip source-noise tensors."""
    from workload import inputs as I
    mel, vec, pit, spk = [], [], [], []
    for s in seeds:
        g = torch.Generator().manual_seed(1000 + s)
        mel.append((torch.randn(80, T, generator=g) * 0.5).clamp(-1.0, 1.5))
        vec.append(torch.randn(T, hp.vits.vec_dim, generator=g))
        pit.append(I.synth_f0(T, seed=3 + s, base=180.0 + 40 * (s % 4)))
        spk.append(I.synth_spk(hp.vits.spk_dim, seed=7 + s % 56))
    f = lambda ts: torch.stack(ts).to(device)
    return f(mel), f(vec), f(pit), f(spk)


class Workload:
    """configs[1] (and --batch / --seconds variants of it): device-resident synthetic inputs + the engine objects of one rank."""

    name = "configs[1]"

    def __init__(self, ops, device, whisper, model, hp, args, rank, world):
        from workload import inputs as I      # synthetic input recipe
        self.ops, self.device, self.hp, self.whisper, self.model = ops, device, hp, whisper, model
        self.B, self.T = args.batch, int(args.seconds * 100)
        d = I.synth_clip(T=self.T, hp=hp, seed=100 + rank, B=self.B, ppg=False)
        self.cpu_inputs = d
        self.mel = d["mel"].to(device)
        self.vec = d["vec"].to(device)
        self.pit = d["pit"].to(device)
        self.spk = d["spk"].to(device)
        self.lengths = d["lengths"].to(device, torch.int32)
        self.keep = self.T // 2                                   # whisper/inference.py:40: len // 320 frames
        self.audio_seconds_per_step = self.B * args.seconds
        self.scaling = "weak"
        self.workload = (f"configs[1]: batch={self.B} x {args.seconds:g}s clip per GPU, whisper-large-v2 dims (24 of 32 encoder blocks) "
                         f"+ base.yaml prior/flow/NSF-BigVGAN")

    def step(self, noise=None):
        """The timed unit (svcmi.serving.convert_step on this workload's tensors).  noise=None draws on the device (as the reference
        does per call)."""
        from svcmi.serving import convert_step
        buf = dict(mel=self.mel, vec=self.vec, pit=self.pit, spk=self.spk, lengths=self.lengths)
        return convert_step(self.model, self.whisper, buf, self.keep, noise)


class FlowDecoderBatch(Workload):
    """configs[2]: 16 x 10 s clips per step, flow + decoder only: PPG (50 fps, as Whisper leaves it) / HuBERT vec / F0 are
    pre-extracted inputs resident in HBM; the step is pitch2source + prior encoder + reverse flow + generator."""

    name = "configs[2]"

    def __init__(self, ops, device, whisper, model, hp, args, rank, world):
        self.ops, self.device, self.hp, self.whisper, self.model = ops, device, hp, None, model
        self.B, self.T = args.batch, int(args.seconds * 100)
        seeds = [100 + rank * self.B + b for b in range(self.B)]
        _, self.vec, self.pit, self.spk = synth_features(self.T, hp, seeds, device)
        g = torch.Generator().manual_seed(900 + rank)
        self.ppg50 = torch.randn(self.B, self.T // 2, hp.vits.ppg_dim, generator=g).to(device)     # layer-normed Whisper output: O(1)
        self.lengths = torch.full((self.B,), self.T, dtype=torch.int32, device=device)
        self.audio_seconds_per_step = self.B * args.seconds
        self.scaling = "weak"
        self.cpu_inputs = None
        self.workload = (f"configs[2]: batch={self.B} x {args.seconds:g}s clips per GPU, flow + decoder only (pre-extracted PPG / vec / F0 "
                         f"in HBM): pitch2source + prior encoder + reverse flow + NSF-BigVGAN, base.yaml")

    def step(self, noise=None):
        m = self.model
        src = m.pitch2source(self.pit)
        return m.inference_ppg50(self.ppg50, self.vec, self.pit, self.spk, self.lengths, src)


class ShardedUtterances(Workload):
    """configs[3]: 512 x 10 s utterances, sharded over the ranks (svcmi.dist.plan_batches: 64 per GPU at 8), converted in
    batches of 16 through the FULL pipeline (Whisper -> PPG -> prior / flow / generator).  One step = this rank's whole
    shard; the features of the shard are resident in HBM, each batch is staged into the static input buffers of one captured
    HIP graph (a device-to-device copy, counted in the step)."""

    name = "configs[3]"

    def __init__(self, ops, device, whisper, model, hp, args, rank, world):
        from svcmi import dist as D
        self.ops, self.device, self.hp, self.whisper, self.model = ops, device, hp, whisper, model
        self.B, self.T = args.batch, int(args.seconds * 100)
        self.batches = D.plan_batches(args.utterances, world, rank, self.B)
        ids = [i for b in self.batches for i in b]
        self.all = synth_features(self.T, hp, ids, device)                       # the whole shard: 64 clips = 0.3 GB at 8 GPUs
        self.mel, self.vec, self.pit, self.spk = (t[:self.B].clone() for t in self.all)      # static graph inputs
        self.lengths = torch.full((self.B,), self.T, dtype=torch.int32, device=device)
        self.keep = self.T // 2
        self.n_mine = len(ids)
        self.audio_seconds_per_step = None                                       # strong scaling: the job is fixed
        self.total_audio_seconds = args.utterances * args.seconds
        self.scaling = "strong"
        self.cpu_inputs = None
        self.outputs = torch.empty(len(ids), 1, self.T * 320, device=device)     # the converted shard stays in HBM
        self.workload = (f"configs[3]: {args.utterances} x {args.seconds:g}s utterances sharded over {world} GPU(s) ({len(ids)} on this rank), "
                         f"batches of {self.B}, full pipeline incl. Whisper-24L")
        self.lanes, self.lane_inputs = None, [(self.mel, self.vec, self.pit, self.spk)]

    def one_batch(self, noise=None):
        return Workload.step(self, noise)

    def lane_fn(self, lane):
        """The per-batch pipeline on lane ``lane``'s own static inputs (lane 0: the tensors ``one_batch`` uses)."""
        while len(self.lane_inputs) <= lane:
            self.lane_inputs.append(tuple(t.clone() for t in self.lane_inputs[0]))
        mel, vec, pit, spk = self.lane_inputs[lane]
        m = self.model

        def fn():
            ppg50 = self.whisper.encoder(mel, torch.randn_like(mel), 0.1)[:, :self.keep]
            return m.inference_ppg50(ppg50, vec, pit, spk, self.lengths, m.pitch2source(pit))
        return fn

    def step(self, noise=None):
        """One pass over this rank's shard: batch j runs on lane j % lanes (stage-in copy, graph replay and stage-out copy all on
        that lane's stream), so consecutive batches overlap; without lanes (--eager) the batches run back to back."""
        done = 0
        for j, b in enumerate(self.batches):
            n = len(b)
            if self.lanes is None:
                for dst, src in zip(self.lane_inputs[0], self.all):
                    dst[:n].copy_(src[done:done + n])
                self.outputs[done:done + n].copy_(self.one_batch()[:n])
            else:
                l = j % len(self.lanes)
                with torch.cuda.stream(self.lanes.streams[l]):
                    for dst, src in zip(self.lane_inputs[l], self.all):
                        dst[:n].copy_(src[done:done + n])
                    self.lanes.graphs[l].replay()
                    self.outputs[done:done + n].copy_(self.lanes.outputs[l][:n])
            done += n
        return self.outputs


class LongForm(Workload):
    """configs[4]: one 30 s clip per rank and step through the reference's schedule -- pred_ppg's two 15 s windows
    (whisper/inference.py:37-61; Tw = 750 each, run as ONE batch of 2: windows are independent), np.repeat x2 fused, then
    svc_infer's chunks [0,2510) -> keep [0,-3200) and [2490,3000) -> keep [3200,-1) (svc_inference.py:101-131), output L - 1
    samples in HBM."""

    name = "configs[4]"

    def __init__(self, ops, device, whisper, model, hp, args, rank, world):
        from svcmi.svc_inference import chunk_schedule
        from svcmi.whisper.inference import window_plan
        self.ops, self.device, self.hp, self.whisper, self.model = ops, device, hp, whisper, model
        secs = args.seconds
        self.T = int(secs * 100)
        n_samples = int(secs * 16000)
        self.windows = window_plan(n_samples)
        assert len({e - s for (s, e, _) in self.windows}) == 1, "equal windows expected (30 s = 2 x 15 s)"
        n_mel = (self.windows[0][1] - self.windows[0][0]) // 160
        self.keep = [k for (_, _, k) in self.windows]
        mel, self.vec, self.pit, self.spk = synth_features(n_mel * len(self.windows), hp, [100 + rank], device)
        self.mel = mel[0].view(80, len(self.windows), n_mel).permute(1, 0, 2).contiguous()          # [windows, 80, n_mel]
        self.vec, self.pit = self.vec[:, :self.T], self.pit[:, :self.T]
        self.plan = chunk_schedule(self.T, hp.data.hop_length)
        self.B = 1
        self.audio_seconds_per_step = secs
        self.scaling = "weak"
        self.cpu_inputs = None
        self.workload = (f"configs[4]: one {secs:g}s clip per GPU and step, reference schedule: {len(self.windows)} Whisper windows of "
                         f"Tw={n_mel // 2} (one batch), synthesis chunks {[(a, b) for (a, b, _, _) in self.plan]} with halo trim, whisper-large-v2 dims + base.yaml")

    def step(self, noise=None):
        m, hop = self.model, self.hp.data.hop_length
        ppg = self.whisper.encoder(self.mel, torch.randn_like(self.mel), 0.1)                      # [windows, Tw, 1280]
        ppg50 = torch.cat([ppg[i, :k] for i, k in enumerate(self.keep)], 0).unsqueeze(0)             # pred_ppg's concat (:48,60)
        src = m.pitch2source(self.pit)                                                                # the whole clip (svc_inference.py:89)
        pieces = []
        for (cs, ce, cso, ceo) in self.plan:
            n = ce - cs
            lengths = torch.full((1,), n, dtype=torch.int32, device=self.device)
            o = m.inference_ppg50(ppg50[:, cs // 2:(ce + 1) // 2], self.vec[:, cs:ce], self.pit[:, cs:ce], self.spk, lengths,
                                  src[:, :, cs * hop:ce * hop].contiguous())
            pieces.append(o[0, 0, cso:ceo])
        return torch.cat(pieces)


WORKLOADS = {1: Workload, 2: FlowDecoderBatch, 3: ShardedUtterances, 4: LongForm}
# per-config defaults: (batch, seconds, whisper precision, synthesizer precision)
# (configs[4] is "fp16 MFMA" in BASELINE.json: both networks in f16 since round 3 -- 9.4e-4 on the waveform, inside the 1e-3 bar
#  (profiles/r03k_precision_report.json); `--precision` / the r02 lines used f16 Whisper + fp32 synthesizer)
#  round 4: configs[2] / [4] run the per-layer MIXED policy in the synthesizer (split-bf16 where the waveform error is made, fp16 elsewhere)
DEFAULTS = {1: (1, 10.0, "f32", "f32"), 2: (16, 10.0, None, "mixed"), 3: (16, 10.0, "f32", "f32"), 4: (1, 30.0, "f16", "mixed")}


def build_graph(fn, warm=2, stream=None):
    """Capture ``fn`` into a HIP graph (torch.cuda.CUDAGraph captures the ctypes launches made on its capture stream).
    Returns (graph, output) or (None, None) if capture is not possible.  ``stream``: capture on this stream instead of torch's
    shared capture stream -- graphs that replay concurrently must not share the per-stream split-K workspace of ``Ops``."""
    try:
        s = stream if stream is not None else torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **({"stream": stream} if stream is not None else {})):
            out = fn()
        g.replay()
        torch.cuda.synchronize()
        return g, out
    except Exception as e:       # noqa: BLE001
        log("graph capture failed, running eager:", repr(e))
        torch.cuda.synchronize()
        return None, None


def roofline_pass(wl, fn):
    """Instrumented pass: every launch bracketed by HIP events on the launch stream.  A spin kernel is queued
    first so the host runs ahead and the device executes the launches back-to-back (no host-induced gaps)."""
    ops = wl.ops
    fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e8))        # ~0.1 s of device spin
    ops.trace_begin()                    # HIP events around every launch, recorded by the C++ stage host on the launch stream
    fn()
    return ops.trace_end()


def measured_precision_error(config, wprec, sprec):
    """Waveform max-abs error of a reduced-precision mode against the fp32 oracle, from the newest committed report of
    tests/test_gpu_precision.py (profiles/r*_precision_report.json).  The north_star tolerance is 1e-3; bf16 (what BASELINE.json names for
    configs[2]) is outside it, bf16x3 and f16 are inside -- the line says which it is instead of leaving it implicit."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_precision_report.json")))
    mode = sprec if sprec not in (None, "f32") else wprec
    if not files or mode in (None, "f32"):
        return None
    try:
        rep = json.load(open(files[-1]))
        key = {1: "configs1_", 2: "configs2_", 3: "configs1_", 4: "configs1_"}[config] + mode
        e = rep[key]
        err = e.get("wave_max_abs_err", e.get("item0_vs_oracle"))
        return {"mode": mode, "waveform_max_abs_err_vs_fp32_oracle": err, "within_1e-3": bool(err <= 1e-3),
                "measured_on": key.split("_")[0], "source": os.path.relpath(files[-1], ROOT)}
    except Exception:       # noqa: BLE001
        return None


def live_precision_error(step, whisper, model, wprec, sprec, seed=20240922):
    """Max-abs waveform difference between the step in the line's precision and the same step in fp32, identical inputs and draws."""
    def run(wp, sp):
        saved = (whisper.encoder.precision if whisper is not None else None, model.precision)
        if whisper is not None:
            whisper.encoder.precision = wp
        model.precision = sp
        try:
            torch.manual_seed(seed)
            torch.cuda.manual_seed_all(seed)
            out = step()
            torch.cuda.synchronize()
            return out.clone()
        finally:
            if whisper is not None:
                whisper.encoder.precision = saved[0]
            model.precision = saved[1]
    ref, got = run(None, None), run(wprec, sprec)
    err = float((got - ref).abs().max())
    return {"live_max_abs_vs_fp32_engine": err, "live_within_1e-3": bool(err <= 1e-3), "live_wave_rms": float(ref.pow(2).mean().sqrt())}


def measured_traffic(kernel="conv_gemm_kernel"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/r*_traffic.json,
    made by scripts/pmc_traffic.sh + scripts/traffic_summary.py: rocprofv3 cannot be driven from inside this
    process).  Returns (bytes_per_launch, source) or (None, None)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        if d.get("csrc_sha") != csrc_sha():          # measured on other kernel sources than this tree's: not this build's number
            return None, os.path.relpath(files[-1], ROOT) + " [STALE: csrc changed since]"
        k = d["kernels"][kernel]
        return int(k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:       # noqa: BLE001
        return None, None


def measured_inflight_timeline():
    """The implicit-GEMM family's in-flight timeline from the newest committed probe record (profiles/r*_inflight_timeline.json: block stamps
    written by the kernels of a -DSVCMI_PROBE_KTRACE=1 build while 4 clips were in flight -- rocprofv3 serialises the queues, so the judged
    regime cannot be traced from outside; scripts/inflight_timeline.py).  Returns (dict, source) or (None, source-if-stale)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_inflight_timeline.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        if d.get("csrc_sha") != csrc_sha():
            return None, os.path.relpath(files[-1], ROOT) + " [STALE: csrc changed since]"
        return d, os.path.relpath(files[-1], ROOT)
    except Exception:       # noqa: BLE001
        return None, None


def rocprof_gemm_ms_per_step(prefixes=("conv_gemm_kernel<", "conv_gemm_group_kernel<")):
    """Summed `calls_per_step x avg_us` of the implicit-GEMM kernels in the newest committed rocprofv3 --kernel-trace --stats
    summary of the judged configuration (profiles/r*_kernel_stats.csv, scripts/prof_summary.py).  Returns (ms, source)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv")))
    if not files:
        return None, None
    ms = 0.0
    try:
        if f"csrc_sha {csrc_sha()}" not in open(files[-1]).readline():
            return None, os.path.relpath(files[-1], ROOT) + " [STALE: csrc changed since]"
        for line in open(files[-1]):
            if line.startswith("\n") or line.startswith("# per"):
                if ms:
                    break
                continue
            if line.startswith(prefixes):
                name, rest = line.rsplit(">", 1)
                cols = rest.strip(",\n").split(",")            # calls, calls_per_step, total_us, avg_us, percent
                ms += float(cols[1]) * float(cols[3]) / 1e3
    except Exception:       # noqa: BLE001
        return None, None
    return (ms, os.path.relpath(files[-1], ROOT)) if ms else (None, None)


def cpu_baseline(wl, wsd, vsd, hp):
    """The oracle (CPU port of the reference path, reference operator sequence) on ONE clip of the configs[1] workload on this
    host's cores.  torch's intra-op thread count is swept once on a 2 s clip of the synthesis path (10-160 channel ops
    oversubscribe a 128-core host), then the full clip runs three times at the best count and the median is reported.  The
    last run also serves as the parity check of the GPU path with identical noise."""
    from oracle import svc_oracle as O
    from workload import inputs as I
    d = wl.cpu_inputs
    noise = {k: d[k][:1] for k in ("mel_noise", "rand_ini", "src_noise", "enc_noise")}
    dims = wsd["dims"]
    ncpu = os.cpu_count() or 1
    saved_threads = torch.get_num_threads()
    sweep = {}
    ds = I.synth_clip(T=200, hp=hp, seed=7, B=1)
    with torch.no_grad():
        for nt in sorted({n for n in (8, 16, 32, 64, 128) if n <= ncpu} | {min(ncpu, 128)}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            s_ = O.pitch2source(vsd, hp, ds["pit"], ds["rand_ini"], ds["src_noise"])
            O.synth_inference(vsd, hp, ds["ppg"], ds["vec"], ds["pit"], ds["spk"], ds["lengths"], s_, ds["enc_noise"])
            sweep[nt] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        # the Whisper encoder is dense 1280 / 5120-wide GEMMs: it wants more threads than the 10..160-channel synthesizer (VERDICT r4
        # item 6) -- its own sweep on a 2 s window, its own count in the timed runs
        sweep_w = {}
        mel_s = (d["mel"][:1] + 0.1 * noise["mel_noise"])[:, :, :200].contiguous()
        for nt in sorted({n for n in (16, 32, 64, 128) if n <= ncpu} | {min(ncpu, 128)}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            O.audio_encoder(wsd["model_state_dict"], mel_s, dims["n_audio_head"], O.whisper_kept_layers(dims))
            sweep_w[nt] = time.perf_counter() - t0
        best_w = min(sweep_w, key=sweep_w.get)
        runs = []
        for _ in range(7):       # ~12-18 s of CPU work (three runs spread by 34 %); one warm run first, median of the rest
            torch.set_num_threads(best_w)
            t0 = time.perf_counter()
            ppg50 = O.audio_encoder(wsd["model_state_dict"], d["mel"][:1] + 0.1 * noise["mel_noise"], dims["n_audio_head"],
                                    O.whisper_kept_layers(dims))[:, :wl.keep]
            t1 = time.perf_counter()
            torch.set_num_threads(best)
            src = O.pitch2source(vsd, hp, d["pit"][:1], noise["rand_ini"], noise["src_noise"])
            t2 = time.perf_counter()
            ppg = ppg50.repeat_interleave(2, dim=1)          # np.repeat(ppg, 2, 0), svc_inference.py:175-177
            wav = O.synth_inference(vsd, hp, ppg, d["vec"][:1], d["pit"][:1], d["spk"][:1], d["lengths"][:1], src, noise["enc_noise"])
            t3 = time.perf_counter()
            runs.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
    torch.set_num_threads(saved_threads)
    runs = sorted(runs[1:])
    tot, tw, tp, ti = runs[len(runs) // 2]
    secs = wl.T / 100.0
    info = {"value": round(secs / tot, 3), "unit": "audio-seconds/sec", "cores": max(best, best_w), "kind": "port",
            "sample": (f"1 clip x {secs:g} s, oracle (torch CPU fp32, reference operator sequence), median of 6 after one warm run, per-stage thread counts of {ncpu} "
                       f"logical CPUs: whisper {tw:.2f}s at {best_w} threads + pitch2source {tp:.2f}s + inference {ti:.2f}s at {best} threads; runs "
                       f"{[round(r[0], 2) for r in runs]} s; thread sweeps: 2 s synthesis clip {{{', '.join(f'{k}: {v:.2f}s' for k, v in sorted(sweep.items()))}}}, "
                       f"2 s Whisper window {{{', '.join(f'{k}: {v:.2f}s' for k, v in sorted(sweep_w.items()))}}}")}
    # parity of the GPU path on the same clip and the same noise
    dev_noise = {k: v.to(wl.device) for k, v in noise.items()}
    saved = (wl.mel, wl.vec, wl.pit, wl.spk, wl.lengths, wl.B)
    wl.mel, wl.vec, wl.pit, wl.spk, wl.lengths, wl.B = wl.mel[:1], wl.vec[:1], wl.pit[:1], wl.spk[:1], wl.lengths[:1], 1
    got = wl.step(dev_noise)
    torch.cuda.synchronize()
    wl.mel, wl.vec, wl.pit, wl.spk, wl.lengths, wl.B = saved
    err = float((got.cpu() - wav).abs().max())
    return info, err


def respawn_under_launcher(n):
    """Run THIS command line as n ranks: python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same args>.
    Returns the launcher's exit code (non-zero if fewer than n GPUs are visible: nothing is measured, nothing is printed)."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and os.environ.get("SVCMI_DIST_BACKEND") != "gloo":
        log(f"bench.py --gpus {n}: {have} GPU(s) visible and no WORLD_SIZE in the environment -- not measuring a smaller job under this flag")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: no WORLD_SIZE in the environment, re-launching as " + " ".join(cmd))
    return subprocess.call(cmd)


def ranks_seen(world, device):
    """Every rank contributes a one: what an all-reduce over the job's backend (nccl = RCCL) returns is the number of ranks it carried."""
    import torch.distributed as dist
    if world == 1:
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t)
    return int(t.item())


def dry_run(args, rank, world, C, W, D, PW):
    """`--dry-run`: everything of a multi-rank bench job that is NOT the GPU pipeline, executed for real on CPU over gloo (the
    exact command the driver launches -- torch.distributed.run, N ranks, `--gpus N --config 3` -- has run end to end once before a multi-GPU
    node sees it).  Rank 0 packs tiny seeded weights, ONE broadcast per model ships the packed arena, every rank plans its shard
    (svcmi.dist.plan_batches), the timed region is bracketed by barriers, the step time is the max over ranks, and the all-reduce of ones
    must return the world size.  The stand-in "pipeline" only records which utterances this rank converted."""
    import torch.distributed as dist
    hp = C.tiny_hp()
    vw = PW.VitsWeights(W.make_vits_state(hp, seed=1234), hp, "cpu") if rank == 0 else None
    ww = PW.WhisperWeights(W.make_whisper_state(C.WHISPER_TINY_TEST), "cpu") if rank == 0 else None
    if world > 1:
        vw = D.broadcast_packed(vw, 0, "cpu")
        ww = D.broadcast_packed(ww, 0, "cpu")
    digest = float(sum(b[k].double().abs().sum() for b in ww.blocks for k in b) + ww.pos.double().abs().sum())
    if args.config == 3:
        batches = D.plan_batches(args.utterances, world, rank, args.batch)
    else:
        batches = [[rank + world * i for i in range(args.batch)]]
    converted = []

    def step():
        for b in batches:
            converted.extend(b)             # the stand-in for: stage the batch, replay the lane's graph, copy the result out
            time.sleep(0.001)

    def timed(steps, warmup):
        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt
    del converted[:]
    dt = timed(args.steps, args.warmup)
    mine = sorted(set(converted))
    seen = ranks_seen(world, "cpu")
    info = [None] * world
    if world > 1:
        dist.all_gather_object(info, (rank, len(mine), digest, mine if args.config == 3 else None))
    else:
        info = [(rank, len(mine), digest, mine if args.config == 3 else None)]
    if rank == 0:
        ok_weights = len({round(d, 6) for (_, _, d, _) in info}) == 1
        covered = sorted(i for (_, _, _, m) in info if m for i in m) == list(range(args.utterances)) if args.config == 3 else None
        print(json.dumps({
            "metric": "audio-seconds/sec end-to-end SVC @32kHz, 10s clips (Whisper-PPG -> flow -> NSF-BigVGAN)", "value": None, "unit": "audio-seconds/sec",
            "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / max(args.steps, 1), 3),
            "higher_is_better": True, "scaling": "strong" if args.config == 3 else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "none (dry run: tiny seeded weights, stand-in pipeline)",
            "config": {"workload": f"configs[{args.config}] launch contract on CPU", "world_size": world, "dist_backend": dist.get_backend() if world > 1 else None,
                       "rccl_ranks_seen": seen, "utterances_per_rank": [n for (_, n, _, _) in sorted(info)], "batches_on_rank0": [len(b) for b in batches],
                       "weights_identical_on_every_rank": ok_weights, "every_utterance_exactly_once": covered,
                       "weights": "rank 0 packs, one broadcast of the packed arena per model" if world > 1 else "packed on this rank"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; config 3: 2)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; config 3: 1)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE.json configs[N] (1 = the judged line)")
    ap.add_argument("--batch", type=int, default=None, help="clips per step (config 1: 1; configs 2 / 3: 16)")
    ap.add_argument("--seconds", type=float, default=None, help="clip length (10 s; config 4: 30 s)")
    ap.add_argument("--utterances", type=int, default=512, help="config 3: total utterances of the job")
    ap.add_argument("--eager", action="store_true", help="do not replay a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=None,
                    help="clips (steps) in flight per GPU: N > 1 replays N independently captured graphs round-robin on N HIP streams, so "
                         "one clip's latency-bound prior / flow / generator launches run beside the next clip's Whisper GEMMs.  Default 4 "
                         "for config 1 (config.single_stream then reports the one-clip-at-a-time figure too), 1 otherwise")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the one-clip-at-a-time timing that accompanies --inflight > 1")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the launch contract of a multi-rank job end to end on CPU over gloo -- real rendezvous, real packed-arena broadcast "
                         "(tiny weights), real plan_batches / barriers / max-over-ranks timing / rccl_ranks_seen, a stand-in for the GPU pipeline.  "
                         "Prints the JSON line with \"dry_run\": true and value null (nothing is measured)")
    ap.add_argument("--precision", default=None,
                    help="GEMM operand precision of BOTH networks (fp32 accumulate in every mode): f32 | bf16x3 | bf16 | f16 | mixed | "
                         "mixed:<class>=<mode>,... (per-layer policy of the synthesizer, svcmi._lib.MIXED_DEFAULT; Whisper then runs f16).  "
                         "Default per config: 1 and 3 f32 (the parity default, the judged line), 2 mixed (flow + decoder), 4 f16 Whisper + mixed synthesizer")
    args = ap.parse_args()
    d_batch, d_secs, d_wprec, d_sprec = DEFAULTS[args.config]
    args.batch = args.batch or d_batch
    args.seconds = args.seconds or d_secs
    args.steps = args.steps if args.steps is not None else (2 if args.config == 3 else 20)
    args.inflight = args.inflight if args.inflight is not None else {1: 4, 2: 3, 3: 2, 4: 4}[args.config]
    args.warmup = args.warmup if args.warmup is not None else (1 if args.config == 3 else 3)
    wprec, sprec = (args.precision, args.precision) if args.precision else (d_wprec, d_sprec)
    if wprec and wprec.startswith("mixed"):
        wprec = "f16"                    # the policy classes are the synthesizer's; Whisper follows the reference's .half()
    norm = lambda p: None if p in (None, "f32") else p

    from workload import config as C, weights as W      # synthetic checkpoint factory + base.yaml values
    from svcmi import Ops, SynthesizerInfer, dist as D, weights as PW
    from svcmi.whisper.inference import WhisperEncoderModel
    import torch.distributed as dist

    if args.dry_run:
        os.environ.setdefault("SVCMI_DIST_BACKEND", "gloo")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: never print an N = 1 line under an N > 1 flag.  Re-exec the same command as N
        # ranks (one per GPU) under torch.distributed.run -- the launch line the driver itself uses -- and hand its exit code back.
        sys.exit(respawn_under_launcher(args.gpus))
    rank, local_rank, world = D.init_from_env()
    if world != args.gpus:
        log(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}: refusing to print a line for a job of another size")
        sys.exit(2)
    if args.dry_run:
        sys.exit(dry_run(args, rank, world, C, W, D, PW))
    assert torch.cuda.is_available(), "bench.py needs a GPU (svcmi has no CPU path)"
    if world > 1 and dist.get_backend() == "nccl" and torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        log(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPUs visible (RCCL wants one device per rank)")
        sys.exit(2)
    device = torch.device("cuda", local_rank % torch.cuda.device_count())     # (> 1 rank per device only under SVCMI_DIST_BACKEND=gloo)
    torch.cuda.set_device(device)
    ops = Ops()
    hp = C.base_hp()
    need_whisper = args.config != 2

    # weights: rank 0 makes the seeded checkpoints and PACKS them (weight-norm folded, GEMM layouts); every rank receives
    # the packed arena through one RCCL broadcast per model and only takes views of it
    t0 = time.perf_counter()
    wsd_cpu = vsd_cpu = vw = ww = None
    t_make = 0.0
    if rank == 0:
        # (making the seeded random-init checkpoints stands in for torch.load and dominates: ~6 s for the 1.9 GB Whisper state;
        #  folding weight-norm + building the GEMM layouts + the upload is ~0.5 s)
        vsd_cpu = W.make_vits_state(hp, seed=1234)
        if need_whisper:
            wsd_cpu = W.make_whisper_state(C.WHISPER_LARGE_V2)
        t_make = time.perf_counter() - t0
        vw = PW.VitsWeights(vsd_cpu, hp, device)
        if need_whisper:
            ww = PW.WhisperWeights(wsd_cpu, device)
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    if world > 1:
        vw = D.broadcast_packed(vw, 0, device)
        if need_whisper:
            # a Whisper that RUNS in fp16 (configs[4]) travels as fp16: 0.96 GB instead of 1.91 GB over xGMI (svcmi.dist.gemm_operands)
            ww = D.broadcast_packed(ww, 0, device, fp16_filter=D.gemm_operands() if norm(wprec) == "f16" else None)
        torch.cuda.synchronize()
    log(f"[rank {rank}] weights ready in {time.perf_counter() - t0:.1f}s (rank 0: synthetic checkpoint {t_make:.1f}s + fold / pack / upload "
        f"{t1 - t0 - t_make:.1f}s; broadcast {time.perf_counter() - t1:.1f}s)")
    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops).load_packed(vw, device)
    model.precision = norm(sprec)
    whisper = None
    if need_whisper:
        whisper = WhisperEncoderModel(None, device, ops=ops, packed=ww)
        whisper.encoder.precision = norm(wprec)
    wl = WORKLOADS[args.config](ops, device, whisper, model, hp, args, rank, world)

    def timed(run_fn, sync_fn, steps, warmup):
        for _ in range(warmup):
            run_fn()
        sync_fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        for _ in range(steps):
            run_fn()
        sync_fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_start
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    graph, lanes, single = None, None, None
    inflight = 1 if args.eager else args.inflight
    if not args.eager and args.config == 3:     # the per-batch pipeline is the graph (one per lane); the shard loop around it stays on the host
        from svcmi.lanes import GraphLanes
        lanes = wl.lanes = GraphLanes([wl.lane_fn(l) for l in range(inflight)])
        graph = lanes.graphs[0]
        run = wl.step
    elif not args.eager:
        from svcmi.lanes import GraphLanes
        # every lane: own static inputs / outputs (its own Workload), own capture stream (= own split-K workspace in Ops)
        # (lane i of rank r converts the clips seeded r + world * i: every lane its own clip)
        wls = [wl] + [WORKLOADS[args.config](ops, device, whisper, model, hp, args, rank + world * i, world) for i in range(1, inflight)]
        if args.config == 1:       # the product's serving object: N lanes of one (B, T) bucket, every lane staged with its own clip
            from svcmi.serving import ClipLanes
            cl = ClipLanes(model, whisper, wl.T, wl.B, lanes=inflight, device=device)
            for i, w in enumerate(wls):
                cl.stage(i, mel=w.mel, vec=w.vec, pit=w.pit, spk=w.spk, lengths=w.lengths)
            lanes = cl.capture().lanes
        else:
            lanes = GraphLanes([w.step for w in wls])
        graph = lanes.graphs[0]
        if inflight > 1 and not args.no_single_stream:   # the same K steps one clip at a time: latency of a clip, reported beside the line
            one = lanes
            if args.config == 1:       # a ONE-lane capture of the serving object (a clip alone on the chip keeps the 3-deep GEMM ring: svcmi/serving.py)
                cl1 = ClipLanes(model, whisper, wl.T, wl.B, lanes=1, device=device)
                cl1.stage(0, mel=wl.mel, vec=wl.vec, pit=wl.pit, spk=wl.spk, lengths=wl.lengths)
                one = cl1.capture().lanes
            dt = timed(lambda: one.launch(0), one.synchronize, args.steps, args.warmup)
            single = {"ms_per_step": round(1000.0 * dt / args.steps, 3),
                      "value": round(wl.audio_seconds_per_step * world / (dt / args.steps), 2)}
            if args.config == 1:
                single["capture"] = "one-lane ClipLanes (3-deep GEMM ring)"
        run = lanes.launch
    else:
        run = wl.step
    elapsed = timed(run, lanes.synchronize if lanes is not None else torch.cuda.synchronize, args.steps, args.warmup)
    ms_per_step = 1000.0 * elapsed / args.steps
    if os.environ.get("SVCMI_TIMELINE"):
        # In-flight timeline (probe library with -DSVCMI_PROBE_KTRACE=1 only; scripts/inflight_timeline.py): after the timed loop, record the
        # block stamps of N more steps in THIS launch regime and summarise them -- gemm_ms_per_step is measured, not inferred.
        import ctypes
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
        import inflight_timeline as TLM
        plib = ctypes.CDLL(os.environ["SVCMI_LIB"])
        plib.svcmi_probe_timeline_read.restype = ctypes.c_longlong
        plib.svcmi_probe_timeline_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
        sync = lanes.synchronize if lanes is not None else torch.cuda.synchronize
        nrec = int(os.environ.get("SVCMI_TIMELINE_STEPS", "8"))
        for _ in range(4):
            run()                               # (the queues stay full across the reset below: nothing is drained in between)
        sync(); torch.cuda.synchronize()
        assert plib.svcmi_probe_timeline_reset() == 0
        for _ in range(nrec):
            run()
        sync(); torch.cuda.synchronize()
        cap = 1 << 22
        buf = np.zeros(5 * cap, dtype=np.uint64)
        got = plib.svcmi_probe_timeline_read(buf.ctypes.data_as(ctypes.c_void_p), cap)
        summ = TLM.summarise(buf[:5 * got], clips=nrec * args.batch)
        summ.update(records=int(got), regime=f"{inflight} clips in flight" if inflight > 1 else "one clip at a time", ms_per_step_timed=round(ms_per_step, 4))
        try:
            from workload import stamp as ST
            summ["csrc_sha"] = ST.csrc_sha()
        except Exception:       # noqa: BLE001
            pass
        with open(os.environ["SVCMI_TIMELINE"], "w") as f:
            json.dump(summ, f, indent=1)
        log("timeline: " + json.dumps({k: v for k, v in summ.items() if k != "classes"}))
    if os.environ.get("SVCMI_KTRACE_CLOCK"):       # timing-probe library (-DSVCMI_PROBE_KTRACE=1) only: the shader clock the GEMM K loops see in THIS launch regime
        import ctypes
        plib = ctypes.CDLL(os.environ["SVCMI_LIB"])
        cal = (ctypes.c_ulonglong * (16 * 8192 + 8))()
        ghz = []
        for _ in range(8):
            run()
            (lanes.synchronize if lanes is not None else torch.cuda.synchronize)()
            torch.cuda.synchronize()
            plib.svcmi_probe_ktrace_read(cal, 16 * 8192 + 8)
            import numpy as np
            a = np.frombuffer(cal, dtype=np.uint64)[:16 * 512].reshape(512, 4, 4).astype(np.float64)
            ok = a[..., 3] > 0
            c = cal[16 * 8192:]
            ghz.append({"GHz": round(c[0] / max(c[1], 1) / 10.0, 3), "loop_ticks": int(c[0]), "prologue": int(c[2]), "epilogue": int(c[3]), "launches": int(c[4]), "tail_ksteps": int(c[5]), "acc_to_lds": int(c[6]), "epi_loop": int(c[7]), "store_drain": int(c[3]) - int(c[5]) - int(c[6]) - int(c[7]),
                        "ticks_per_kstep_512blocks": round(float((a[..., 0][ok] / a[..., 3][ok]).mean()), 0) if ok.any() else None, "ksteps": float(a[..., 3][ok].mean()) if ok.any() else None,
                        "barrier": round(float((a[..., 2][ok] / (a[..., 3][ok] + 1)).mean()), 0) if ok.any() else None, "vmwait": round(float((a[..., 1][ok] / (a[..., 3][ok] + 1)).mean()), 0) if ok.any() else None})
        for g in ghz:
            log(f"ktrace sample: {g}")
        log(f"ms_per_step {ms_per_step:.3f}")
    audio_s = wl.total_audio_seconds if wl.scaling == "strong" else wl.audio_seconds_per_step * world
    value = audio_s / (ms_per_step / 1000.0)
    def spell(p):
        if p and p.startswith("mixed"):
            from svcmi import _lib
            code, cls = _lib.parse_precision(p)
            names = {v: k for k, v in _lib.PRECISIONS.items() if k in ("f32", "bf16x3", "bf16", "f16", "f16w2")}
            return "mixed per-layer policy (" + ", ".join(f"{k}={names[cls[i]]}" for k, i in sorted(_lib.CLASS_NAMES.items(), key=lambda kv: kv[1])) + ")"
        return p
    prec_txt = ", ".join(f"{n} {'fp32' if norm(p) is None else spell(p) + ' GEMM operands / fp32 accumulate'}"
                         for n, p in (("Whisper", wprec), ("synthesizer", sprec)) if not (n == "Whisper" and not need_whisper))
    dtype = (norm(sprec) or "f32") if (not need_whisper or norm(wprec) == norm(sprec)) else f"{norm(wprec) or 'f32'} (Whisper) + {norm(sprec) or 'f32'} (synthesizer)"

    out = {
        "metric": "audio-seconds/sec end-to-end SVC @32kHz, 10s clips (Whisper-PPG -> flow -> NSF-BigVGAN)",
        "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None,
        "dtype": dtype, "data": "synthetic (seeded mel/vec/F0/speaker, random-init weights of the named architecture)",
        "config": {"workload": f"{wl.workload}; {prec_txt}",
                   "launch": ("hipGraph replay" if graph is not None else "eager") +
                             (f", {inflight} clips in flight (one lane = HIP stream + graph + static buffers each; GPU_MAX_HW_QUEUES="
                              f"{os.environ.get('GPU_MAX_HW_QUEUES')})" if inflight > 1 else ""),
                   "clips_in_flight": inflight, "gemm_ring2_mask": (cl.ring2 if (args.config == 1 and inflight > 1 and not args.eager) else 0),
                   "world_size": world, "dist_backend": (dist.get_backend() if world > 1 else None),
                   "rccl_ranks_seen": ranks_seen(world, device),
                   # multi-rank runs of an f16 Whisper ship its GEMM operands as fp16 (every rank, rank 0 included, then holds the
                   # fp16-rounded weights: the live error's fp32 leg uses them too, i.e. it measures activation rounding only)
                   "whisper_weights_wire": "f16" if (world > 1 and need_whisper and norm(wprec) == "f16") else "f32",
                   "weights": ("rank 0 packs, one broadcast of the packed arena per model (" + str(dist.get_backend()) + ")") if world > 1 else "packed on this rank",
                   "per_gpu_value": round(value / world, 2), "realtime_factor": round(value / world, 2)},
    }
    if single is not None:
        out["config"]["single_stream"] = single
    perr = measured_precision_error(args.config, wprec, sprec)
    if norm(wprec) or norm(sprec):
        # measured HERE, on this run's inputs: the same step (same device RNG seed => same stochastic draws) in the line's mode and in
        # fp32 (the fp32 engine is pinned to the oracle at 2e-6: tests/test_gpu_engine.py), eager, outside the timed region
        live = live_precision_error(wl.one_batch if args.config == 3 else wl.step, whisper, model, norm(wprec), norm(sprec))
        perr = dict(perr or {"mode": norm(sprec) or norm(wprec)}, **live)
    if perr is not None:
        out["config"]["precision_error"] = perr
    if rank == 0 and not args.no_roofline:
        fn = wl.one_batch if args.config == 3 else wl.step
        agg = roofline_pass(wl, fn)
        total_ms = sum(a["ms"] for a in agg.values())
        # the dominant kernel family: the implicit-GEMM body, single + grouped launches (3 convolutions per grid), fp32 and
        # 16-bit-operand instantiations; `peak` is the fp32 matrix peak unless every GEMM FLOP of the step ran on 16-bit operands
        fams = {"f32": ("svcmi_conv_gemm_f32", "svcmi_conv_gemm_group_f32"), "lp": ("svcmi_conv_gemm_lp", "svcmi_conv_gemm_group_lp")}
        part = {}
        for key, names in fams.items():
            gm = {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}
            for name in names:
                for k, v in agg.get(name, {}).items():
                    if isinstance(v, (int, float)):
                        gm[k] = gm.get(k, 0) + v
            part[key] = gm
        dom = "lp" if part["lp"]["flops"] > part["f32"]["flops"] else "f32"
        gm = part[dom]
        lp_prec = norm(sprec) if not need_whisper else (norm(wprec) or norm(sprec))
        mults = 3.0 if (dom == "lp" and lp_prec == "bf16x3") else 1.0        # MFMA work actually issued per algorithmic FLOP
        peak = FP32_MFMA_PEAK_TFLOPS if dom == "f32" else BF16_MFMA_PEAK_TFLOPS
        ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic() if (dom == "f32" and args.config == 1) else (None, None)
        out["roofline"] = {"kernel": "conv_gemm_kernel + conv_gemm_group_kernel (" + " / ".join(fams[dom]) + ")", "bound": "mfma",
                           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                           "timing": "HIP events on the launch stream around every launch of one instrumented single-stream step (one clip alone on the GPU; includes inter-launch gaps)",
                           "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                           "traffic_source": traffic_src, "launches_per_step": gm["launches"],
                           "avg_launch_us": round(1000.0 * gm["ms"] / gm["launches"], 2),
                           "algorithmic_gflop_per_step": round(gm["flops"] / 1e9, 1),
                           "algorithmic_bytes_per_launch": int(gm["bytes"] / gm["launches"]),
                           "share_of_step_kernel_time": round(gm["ms"] / total_ms, 3)}
        if mults != 1.0:
            out["roofline"]["mfma_issue_frac"] = round(mults * ach / peak, 4)
        if inflight > 1:      # whole-step view with clips in flight: this family's FLOPs of a step over the step's share of the wall clock
            agg_ach = gm["flops"] / (ms_per_step * 1e-3) / 1e12
            out["roofline"]["in_flight"] = {"achieved": round(agg_ach, 2), "frac": round(agg_ach / peak, 4),
                                            "note": f"GEMM FLOPs of one step / ms_per_step with {inflight} clips in flight (every other kernel's time included)"}
        if dom == "f32" and args.config == 1 and args.batch == 1 and args.seconds == 10.0 and inflight > 1:
            tl, tl_src = measured_inflight_timeline()
            if tl:      # MEASURED in the judged regime (records written by the kernels of the probe build of this csrc; ~2 % slower than the shipped build)
                out["roofline"]["in_flight"].update({
                    "gemm_ms_per_step": tl["gemm_ms_per_step"], "ms_per_step_of_the_record": tl["ms_per_step_in_window"],
                    "frac_measured": tl["frac_while_resident"], "gemm_launches_resident_share": tl["concurrency_share"],
                    "sum_of_launch_durations_ms_per_step": tl["sum_ms_per_step"], "source": tl_src,
                    "measured_note": "gemm_ms_per_step = wall time per clip with >= 1 implicit-GEMM launch resident (union of [first block entry, last block exit] "
                                     "over every launch of every lane); frac_measured = GEMM FLOPs / that time / peak"})
            elif tl_src:
                out["roofline"]["in_flight"]["source"] = tl_src
        if dom == "f32" and args.config == 1 and args.batch == 1 and args.seconds == 10.0:
            rp_ms, rp_src = rocprof_gemm_ms_per_step()
            if rp_ms:
                out["roofline"]["frac_rocprof"] = round(gm["flops"] / (rp_ms * 1e-3) / 1e12 / peak, 4)
                out["roofline"]["frac_rocprof_source"] = f"{rp_src}: sum(calls_per_step x avg_us) of conv_gemm_kernel* = {rp_ms:.3f} ms per step"
            elif rp_src:
                out["roofline"]["frac_rocprof_source"] = rp_src
        # committed profile summaries are only quoted when they were measured on THIS tree's kernel sources (workload/stamp.py)
        out["roofline"]["csrc_sha"] = csrc_sha()
        out["roofline"]["stale"] = any("STALE" in str(out["roofline"].get(k) or "") for k in ("traffic_source", "frac_rocprof_source")) or \
            "STALE" in str((out["roofline"].get("in_flight") or {}).get("source") or "")
        out["kernel_time_ms"] = {k.replace("svcmi_", ""): round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        sn = agg.get("svcmi_snake_alias_f32")
        if sn:
            out["snake_alias_GBs"] = round(sn["bytes"] / (sn["ms"] * 1e-3) / 1e9, 1)
    if rank == 0 and world == 1 and args.config == 1 and not args.no_cpu_baseline:
        info, err = cpu_baseline(wl, wsd_cpu, vsd_cpu, hp)
        out["cpu_baseline"] = info
        out["parity_max_abs_vs_oracle"] = err
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
