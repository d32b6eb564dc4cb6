#!/usr/bin/env python
"""bench.py -- end-to-end SVC throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
    mel [B,80,1000] --(+0.1*randn)--> Whisper-24L encoder --> PPG 50 fps --(x2 repeat fused)-->
    prior encoder -> reverse flow -> NSF-BigVGAN (incl. pitch2source) --> 32 kHz waveform [B,1,320000] in HBM.
The workload is BASELINE.json configs[1] (1 GPU, batch 1, 10 s clip, whisper-large-v2 dims + base.yaml decoder),
fp32, random-init weights of that architecture (no checkpoints/network), the path's stochastic draws made on the
device inside the step.  With N GPUs every rank runs the same per-GPU work on its own clip (weak scaling);
weights are generated on rank 0 and broadcast once over RCCL; the hot loop has no collective.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fp32-MFMA conv/linear GEMM):
algorithmic FLOPs of its launches / their HIP-event durations, measured on the launch stream in an instrumented
pass of the same step.  `cpu_baseline` times the oracle (CPU port of the reference path) on one clip.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "whisper-vits-svc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak (no TF32/xf32 on gfx950)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 / fp16 matrix peak (the 5 PF headline figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Workload:
    """Device-resident synthetic inputs + the engine objects for one rank."""

    def __init__(self, ops, device, batch, seconds, wsd, vsd, hp, seed, precision=None):
        from workload import inputs as I      # synthetic input recipe
        from svcmi import SynthesizerInfer
        from svcmi.whisper.inference import load_model
        self.ops, self.device, self.hp = ops, device, hp
        self.B, self.T = batch, int(seconds * 100)
        self.whisper = load_model(wsd, device, ops=ops)
        self.model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops)
        self.model.load_state_dict(vsd)
        self.model.eval()
        self.model.to(device)
        self.model._weights()
        self.whisper.encoder.precision = self.model.precision = precision     # None = fp32 (the judged line)
        d = I.synth_clip(T=self.T, hp=hp, seed=seed, B=batch, ppg=False)
        self.cpu_inputs = d
        self.mel = d["mel"].to(device)
        self.vec = d["vec"].to(device)
        self.pit = d["pit"].to(device)
        self.spk = d["spk"].to(device)
        self.lengths = d["lengths"].to(device, torch.int32)
        self.keep = self.T // 2                                   # whisper/inference.py:40: len // 320 frames

    def step(self, noise=None):
        """The timed unit.  noise=None draws on the device (as the reference does per call)."""
        m, B, T = self.model, self.B, self.T
        mel_noise = torch.randn_like(self.mel) if noise is None else noise["mel_noise"]
        ppg50 = self.whisper.encoder(self.mel, mel_noise, 0.1)[:, :self.keep]
        src = m.pitch2source(self.pit, noise=None if noise is None else (noise["rand_ini"], noise["src_noise"]))
        return m.inference_ppg50(ppg50, self.vec, self.pit, self.spk, self.lengths, src,
                                 noise=None if noise is None else noise["enc_noise"])


def build_graph(wl, warm=2):
    """Capture one step into a HIP graph (torch.cuda.CUDAGraph captures the ctypes launches made on its
    capture stream).  Returns (graph, output) or (None, None) if capture is not possible."""
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm):
                wl.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = wl.step()
        g.replay()
        torch.cuda.synchronize()
        return g, out
    except Exception as e:       # noqa: BLE001
        log("graph capture failed, running eager:", repr(e))
        torch.cuda.synchronize()
        return None, None


def roofline_pass(wl):
    """Instrumented pass: every launch bracketed by HIP events on the launch stream.  A spin kernel is queued
    first so the host runs ahead and the device executes the launches back-to-back (no host-induced gaps)."""
    ops = wl.ops
    wl.step()
    torch.cuda.synchronize()
    ops.timeline = []
    torch.cuda._sleep(int(2.0e8))        # ~0.1 s of device spin
    wl.step()
    torch.cuda.synchronize()
    tl, ops.timeline = ops.timeline, None
    agg = {}
    for name, work, e0, e1 in tl:
        a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        a["launches"] += 1
        a["ms"] += e0.elapsed_time(e1)
        a["flops"] += work.get("flops", 0.0)
        a["bytes"] += work.get("bytes", 0.0)
    return agg


def measured_traffic(kernel="conv_gemm_kernel"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/r*_traffic.json,
    made by scripts/pmc_traffic.sh + scripts/traffic_summary.py: rocprofv3 cannot be driven from inside this
    process).  Returns (bytes_per_launch, source) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"][kernel]
        return int(k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:       # noqa: BLE001
        return None, None


def cpu_baseline(wl, wsd, vsd, hp, gpu_step):
    """Oracle (CPU port of the reference path) on ONE clip of the same workload, on this host's cores; also the
    parity check of the GPU path against it with identical noise."""
    from oracle import svc_oracle as O
    d = wl.cpu_inputs
    noise = {k: d[k][:1] for k in ("mel_noise", "rand_ini", "src_noise", "enc_noise")}
    dims = wsd["dims"]
    t0 = time.perf_counter()
    with torch.no_grad():
        ppg50 = O.audio_encoder(wsd["model_state_dict"], d["mel"][:1] + 0.1 * noise["mel_noise"], dims["n_audio_head"],
                                O.whisper_kept_layers(dims))[:, :wl.keep]
        t1 = time.perf_counter()
        src = O.pitch2source(vsd, hp, d["pit"][:1], noise["rand_ini"], noise["src_noise"])
        t2 = time.perf_counter()
        ppg = ppg50.repeat_interleave(2, dim=1)          # np.repeat(ppg, 2, 0), svc_inference.py:175-177
        wav = O.synth_inference(vsd, hp, ppg, d["vec"][:1], d["pit"][:1], d["spk"][:1], d["lengths"][:1], src, noise["enc_noise"])
        t3 = time.perf_counter()
    secs = wl.T / 100.0
    info = {"value": round(secs / (t3 - t0), 3), "unit": "audio-seconds/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"1 clip x {secs:g} s, oracle (torch CPU fp32): whisper {t1 - t0:.2f}s + pitch2source {t2 - t1:.2f}s + inference {t3 - t2:.2f}s"}
    # parity of the GPU path on the same clip and the same noise
    dev_noise = {k: v.to(wl.device) for k, v in noise.items()}
    saved = (wl.mel, wl.vec, wl.pit, wl.spk, wl.lengths, wl.B)
    wl.mel, wl.vec, wl.pit, wl.spk, wl.lengths, wl.B = wl.mel[:1], wl.vec[:1], wl.pit[:1], wl.spk[:1], wl.lengths[:1], 1
    got = wl.step(dev_noise)
    torch.cuda.synchronize()
    wl.mel, wl.vec, wl.pit, wl.spk, wl.lengths, wl.B = saved
    err = float((got.cpu() - wav).abs().max())
    return info, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="clips per step per GPU (configs[1] = 1)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--eager", action="store_true", help="do not replay a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16", "f16"],
                    help="GEMM operand precision (fp32 accumulate in every mode); f32 is the parity default and the judged line")
    args = ap.parse_args()

    from workload import config as C, weights as W      # synthetic checkpoint factory + base.yaml values
    from svcmi import Ops, dist as D
    import torch.distributed as dist

    rank, local_rank, world = D.init_from_env()
    assert world == args.gpus or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (svcmi has no CPU path)"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    ops = Ops()
    hp = C.base_hp()

    # weights: rank 0 makes the seeded checkpoints, everyone receives them through one RCCL broadcast
    t0 = time.perf_counter()
    if rank == 0:
        wck = W.make_whisper_state(C.WHISPER_LARGE_V2)
        vsd = W.make_vits_state(hp, seed=1234)
        wsd_cpu, vsd_cpu = wck, vsd
    else:
        wck = {"dims": dict(C.WHISPER_LARGE_V2), "model_state_dict": None}
        vsd = None
    if world > 1:
        wck = {"dims": wck["dims"], "model_state_dict": D.broadcast_state_dict(wck["model_state_dict"], 0, device)}
        vsd = D.broadcast_state_dict(vsd, 0, device)
        torch.cuda.synchronize()
    log(f"[rank {rank}] weights ready in {time.perf_counter() - t0:.1f}s")
    prec = None if args.precision == "f32" else args.precision
    wl = Workload(ops, device, args.batch, args.seconds, wck, vsd, hp, seed=100 + rank, precision=prec)

    graph, gout = (None, None) if args.eager else build_graph(wl)
    run = (lambda: graph.replay()) if graph is not None else (lambda: wl.step())
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1000.0 * elapsed / args.steps
    audio_s = args.batch * args.seconds * world
    value = audio_s / (ms_per_step / 1000.0)

    out = {
        "metric": "audio-seconds/sec end-to-end SVC @32kHz, 10s clips (Whisper-PPG -> flow -> NSF-BigVGAN)",
        "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic (seeded mel/vec/F0/speaker, random-init weights of the named architecture)",
        "config": {"workload": f"configs[1]: batch={args.batch} x {args.seconds:g}s clip per GPU, whisper-large-v2 dims "
                               f"(24 of 32 encoder blocks) + base.yaml prior/flow/NSF-BigVGAN, {'fp32' if prec is None else prec + ' GEMM operands / fp32 accumulate'}",
                   "launch": "hipGraph replay" if graph is not None else "eager",
                   "per_gpu_value": round(value / world, 2), "realtime_factor": round(value / world, 2)},
    }
    if rank == 0 and not args.no_roofline:
        agg = roofline_pass(wl)
        total_ms = sum(a["ms"] for a in agg.values())
        # the dominant kernel family: the implicit-GEMM body, single + grouped launches (3 convolutions per grid); in a
        # reduced-precision run its 16-bit instantiations (the few GEMMs that stay fp32 there are reported in kernel_time_ms)
        fam = ("svcmi_conv_gemm_f32", "svcmi_conv_gemm_group_f32") if prec is None else ("svcmi_conv_gemm_lp", "svcmi_conv_gemm_group_lp")
        gm = {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}
        for name in fam:
            for k, v in agg.get(name, {}).items():
                gm[k] += v
        mults = {None: 1.0, "bf16x3": 3.0}.get(prec, 1.0)        # MFMA work actually issued per algorithmic FLOP
        peak = FP32_MFMA_PEAK_TFLOPS if prec is None else BF16_MFMA_PEAK_TFLOPS
        ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic() if prec is None else (None, None)
        out["roofline"] = {"kernel": "conv_gemm_kernel + conv_gemm_group_kernel (" + " / ".join(fam) + ")", "bound": "mfma", "achieved": round(ach, 2),
                           "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                           "mfma_issue_frac": round(mults * ach / peak, 4),
                           "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                           "traffic_source": traffic_src, "launches_per_step": gm["launches"],
                           "avg_launch_us": round(1000.0 * gm["ms"] / gm["launches"], 2),
                           "algorithmic_gflop_per_step": round(gm["flops"] / 1e9, 1),
                           "algorithmic_bytes_per_launch": int(gm["bytes"] / gm["launches"]),
                           "share_of_step_kernel_time": round(gm["ms"] / total_ms, 3)}
        out["kernel_time_ms"] = {k.replace("svcmi_", ""): round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        sn = agg.get("svcmi_snake_alias_f32")
        if sn:
            out["snake_alias_GBs"] = round(sn["bytes"] / (sn["ms"] * 1e-3) / 1e9, 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        info, err = cpu_baseline(wl, wsd_cpu, vsd_cpu, hp, run)
        out["cpu_baseline"] = info
        out["parity_max_abs_vs_oracle"] = err
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
