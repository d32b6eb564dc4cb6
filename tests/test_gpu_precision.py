"""Reduced-precision GEMM operand modes (svcmi_conv_gemm_lp: bf16x3 / bf16 / f16) against the fp32 CPU oracle.

The reference itself runs its accelerator path in fp16 (whisper/inference.py:22-23,43-44 `.half()`); fp32 is this
library's parity default and the 16-bit modes are opt-in.  What is asserted, on the shapes BASELINE.json names:
  * bf16x3 (split-bf16, three MFMAs, fp32 accumulate) meets the north_star bar -- <= 1e-3 max-abs on the waveform --
    on configs[1] (B=1, 10 s, Whisper-large-v2 dims + base.yaml decoder, end to end), on configs[2] (B=16 x 10 s,
    flow + decoder from pre-extracted PPG/F0) and on a 15 s Whisper window (Tw = 750);
  * plain bf16 / f16 errors are MEASURED, printed and bounded (they exceed 1e-3, as SURVEY.md section 0 predicted:
    random-init weights amplify an 8/11-bit operand rounding through 24 + 6 + 16 + 90 layers).
Error bounds of the plain modes are ~4x what was measured on MI355X (profiles/r02b_precision_report.json holds the numbers).
"""
import json
import os

import pytest
import torch

from oracle import svc_oracle as O
from tests import engine_cases as E
from workload import config as C
from workload import inputs as I
from workload import weights as W

pytestmark = pytest.mark.gpu

MODES = ("bf16x3", "bf16", "f16")
# max-abs bounds: waveform in [-1, 1] (rms ~0.1 with these weights); PPG relative to its max |value|
# measured on MI355X (profiles/r02b_precision_report.json): waveform bf16x3 1.3e-5 / 1.7e-5, f16 0.99e-3 / 1.0e-3, bf16 7.3e-3 / 7.9e-3
# (configs[1] / configs[2]); PPG relative error bf16x3 9e-6, f16 6.9e-4, bf16 5.8e-3
WAVE_BOUND = {"bf16x3": E.WAVE_TOL, "bf16": 3e-2, "f16": 4e-3}
PPG_REL_BOUND = {"bf16x3": 1e-4, "bf16": 2e-2, "f16": 3e-3}
REPORT = {}


@pytest.fixture(scope="module")
def ops():
    from svcmi import Ops
    o = Ops()
    assert o.build == "hip:gfx950" and o.on_gpu
    o.lp_min_flops = 2.0e7      # EVERY eligible GEMM of both networks in the mode under test (the product leaves launches below 1.5 GFLOP in fp32)
    return o


@pytest.fixture(scope="module", autouse=True)
def write_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if REPORT and os.path.isdir(out):
        with open(os.path.join(out, "precision_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def whisper(ops):
    from svcmi.whisper.inference import load_model
    ck = W.make_whisper_state(C.WHISPER_LARGE_V2)
    return ck, load_model(ck, "cuda", ops=ops)


@pytest.fixture(scope="module")
def clip10(ops, whisper):
    """configs[1] inputs and the oracle's intermediate / final results for them (one CPU run shared by all modes)."""
    ck, _ = whisper
    hp = C.base_hp()
    m, sd = E.make_model(hp, ops, "cuda")
    d = I.synth_clip(T=1000, hp=hp, seed=0, B=1, ppg=False)
    with torch.no_grad():
        ppg50 = O.audio_encoder(ck["model_state_dict"], d["mel"] + 0.1 * d["mel_noise"], 20, 24)[:, :500]
        src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        wav = O.synth_inference(sd, hp, ppg50.repeat_interleave(2, dim=1), d["vec"], d["pit"], d["spk"], d["lengths"], src, d["enc_noise"])
    return dict(hp=hp, m=m, sd=sd, d=d, ppg50=ppg50, src=src, wav=wav)


def _count_lp(ops, fn):
    """Run fn and return (result, number of reduced-precision GEMM launches it made)."""
    ops.trace_begin()
    out = fn()
    trace = ops.trace_end()
    return out, sum(v["launches"] for name, v in trace.items() if name.endswith("_lp") and "pack" not in name)


@pytest.mark.parametrize("mode", MODES)
def test_configs1_end_to_end(ops, whisper, clip10, mode):
    """mel -> Whisper-24L -> PPG -> prior/flow/generator, every GEMM of both networks in `mode`, vs the fp32 oracle."""
    _, wm = whisper
    c, d, m = clip10, clip10["d"], clip10["m"]
    wm.encoder.precision = m.precision = mode
    try:
        def run():
            ppg50 = wm.encoder(d["mel"], d["mel_noise"], 0.1)[:, :500]
            src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
            wav = m.inference_ppg50(ppg50, d["vec"].cuda(), d["pit"].cuda(), d["spk"].cuda(), d["lengths"].to("cuda", torch.int32), src,
                                    noise=d["enc_noise"].cuda())
            return ppg50, wav
        (ppg50, wav), n_lp = _count_lp(ops, run)
    finally:
        wm.encoder.precision = m.precision = None
    e_ppg = E.maxerr(ppg50, c["ppg50"]) / float(c["ppg50"].abs().max())
    e_wav = E.maxerr(wav, c["wav"])
    rms = float(c["wav"].pow(2).mean().sqrt())
    REPORT[f"configs1_{mode}"] = dict(ppg_rel_err=e_ppg, wave_max_abs_err=e_wav, wave_rms=rms, lp_launches=n_lp)
    print(f"configs[1] {mode}: ppg rel err {e_ppg:.2e}, waveform max-abs err {e_wav:.2e} (rms {rms:.3f}), {n_lp} lp launches")
    assert n_lp >= 24 * 4 + 40, n_lp                      # the Whisper linears and the VITS GEMMs really ran in `mode`
    assert bool(torch.isfinite(wav).all())
    assert e_ppg <= PPG_REL_BOUND[mode] and e_wav <= WAVE_BOUND[mode]


def test_configs1_f16_whisper_mixed_synthesizer(ops, whisper, clip10):
    """configs[1] / [4] style end to end: Whisper in fp16 (the reference's .half()), the synthesizer under the default mixed policy."""
    _, wm = whisper
    c, d, m = clip10, clip10["d"], clip10["m"]
    wm.encoder.precision, m.precision = "f16", "mixed"
    try:
        ppg50 = wm.encoder(d["mel"], d["mel_noise"], 0.1)[:, :500]
        src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
        wav = m.inference_ppg50(ppg50, d["vec"].cuda(), d["pit"].cuda(), d["spk"].cuda(), d["lengths"].to("cuda", torch.int32), src,
                                noise=d["enc_noise"].cuda())
    finally:
        wm.encoder.precision = m.precision = None
    e_wav = E.maxerr(wav, c["wav"])
    REPORT["configs1_mixed"] = dict(wave_max_abs_err=e_wav, ppg_rel_err=E.maxerr(ppg50, c["ppg50"]) / float(c["ppg50"].abs().max()))
    print(f"configs[1] f16 Whisper + mixed synthesizer: waveform max-abs err {e_wav:.2e}")
    assert e_wav <= 1e-3


@pytest.mark.parametrize("mode", MODES)
def test_configs2_batch16_flow_decoder(ops, clip10, mode):
    """configs[2]: 16 x 10 s clips, flow + decoder only (pre-extracted PPG / F0).  Item 0 is the configs[1] clip (its PPG
    = the oracle's Whisper output) and is compared with the oracle waveform; all 16 items are compared with the fp32
    engine, which test_gpu_engine.py pins to the oracle (solo == batch, 10 s clip vs oracle)."""
    c, m, hp = clip10, clip10["m"], clip10["hp"]
    B = 16
    items = [c["d"]] + [I.synth_clip(T=1000, hp=hp, seed=s, B=1, ppg=False) for s in range(1, B)]
    g = torch.Generator().manual_seed(5)
    ppgs = [c["ppg50"]] + [torch.randn(1, 500, hp.vits.ppg_dim, generator=g) * float(c["ppg50"].std()) for _ in range(1, B)]
    d = {k: torch.cat([it[k] for it in items], 0) for k in ("vec", "pit", "spk", "enc_noise", "rand_ini", "src_noise", "lengths")}
    ppg50 = torch.cat(ppgs, 0).cuda()
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    args = (ppg50, d["vec"].cuda(), d["pit"].cuda(), d["spk"].cuda(), d["lengths"].to("cuda", torch.int32), src)
    ref32 = m.inference_ppg50(*args, noise=d["enc_noise"].cuda()).clone()
    m.precision = mode
    try:
        wav, n_lp = _count_lp(ops, lambda: m.inference_ppg50(*args, noise=d["enc_noise"].cuda()))
    finally:
        m.precision = None
    e_item0 = E.maxerr(wav[:1], c["wav"])
    e_all = E.maxerr(wav, ref32)
    REPORT[f"configs2_{mode}"] = dict(item0_vs_oracle=e_item0, all_vs_fp32_engine=e_all, lp_launches=n_lp)
    print(f"configs[2] {mode}: item 0 vs oracle {e_item0:.2e}, 16 items vs fp32 engine {e_all:.2e}, {n_lp} lp launches")
    assert n_lp >= 40 and wav.shape == (B, 1, 320000) and bool(torch.isfinite(wav).all())
    assert E.maxerr(ref32[:1], c["wav"]) <= E.WAVE_TOL
    assert e_item0 <= WAVE_BOUND[mode] and e_all <= 2 * WAVE_BOUND[mode]


# Per-layer mixed precision: a mode per layer class (svcmi_synth_model.class_prec).  scripts/precision_sensitivity.py
# (16-bit operand rounding emulated on the CPU oracle) ranks the classes: conv_pre + the transposed convolutions carry half of the
# fp16 waveform error for 2 % of the FLOPs, the prior encoder a quarter (and all of it on outlier weights), the AMP convolutions and
# the flow -- 90 % of the FLOPs -- the rest in equal small parts.  The sweep below measures candidate policies on configs[2]
# (error of item 0 against the fp32 ORACLE, all 16 items against the fp32 engine, time per step) and on the outlier-stress weights.
MIXED_POLICIES = ["mixed", "mixed:amp3=f16,amp4=f16", "mixed:amp3=f32,amp4=f32", "mixed:amp0=f16w2", "mixed:amp0=f16w2,amp1=f16w2,amp2=f16w2", "mixed:amp1=f16w2,amp2=f16w2", "mixed:encattn=f16", "mixed:amp0=f16", "mixed:amp1=bf16x3", "mixed:amp1=bf16x3,amp2=bf16x3", "mixed:flow=bf16x3",
                  "mixed:flow=bf16x3,amp1=bf16x3", "mixed:enc=f16,ups=f16,amp0=f16"]
MIXED_BOUND = 5e-4          # waveform max-abs of the DEFAULT policy vs the fp32 oracle on configs[2] (north_star bar: 1e-3)


def _time_ms(fn, n=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def test_mixed_precision_policies_configs2(ops, clip10):
    """configs[2] (B = 16 x 10 s, flow + decoder) under the per-class policies; the default one is asserted inside MIXED_BOUND."""
    c, m, hp = clip10, clip10["m"], clip10["hp"]
    B = 16
    items = [c["d"]] + [I.synth_clip(T=1000, hp=hp, seed=s, B=1, ppg=False) for s in range(1, B)]
    g = torch.Generator().manual_seed(5)
    ppgs = [c["ppg50"]] + [torch.randn(1, 500, hp.vits.ppg_dim, generator=g) * float(c["ppg50"].std()) for _ in range(1, B)]
    d = {k: torch.cat([it[k] for it in items], 0) for k in ("vec", "pit", "spk", "enc_noise", "rand_ini", "src_noise", "lengths")}
    ppg50 = torch.cat(ppgs, 0).cuda()
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    args = (ppg50, d["vec"].cuda(), d["pit"].cuda(), d["spk"].cuda(), d["lengths"].to("cuda", torch.int32), src)
    nz = d["enc_noise"].cuda()
    saved = ops.lp_min_flops
    ops.lp_min_flops = 1.5e9          # the product's own threshold (the fixture forces every launch into the mode): this is the configs[2] line as bench.py runs it
    try:
        ref32 = m.inference_ppg50(*args, noise=nz).clone()
        t32 = _time_ms(lambda: m.inference_ppg50(*args, noise=nz))
        rows = {}
        for pol in ["f16", "f16w2", "bf16x3"] + MIXED_POLICIES:
            m.precision = pol
            try:
                wav, n_lp = _count_lp(ops, lambda: m.inference_ppg50(*args, noise=nz))
                ms = _time_ms(lambda: m.inference_ppg50(*args, noise=nz))
            finally:
                m.precision = None
            rows[pol] = dict(item0_vs_oracle=E.maxerr(wav[:1], c["wav"]), all_vs_fp32_engine=E.maxerr(wav, ref32), ms_per_step=ms, lp_launches=n_lp)
            print(f"configs[2] {pol}: item 0 vs oracle {rows[pol]['item0_vs_oracle']:.2e}, 16 items vs fp32 engine {rows[pol]['all_vs_fp32_engine']:.2e}, "
                  f"{ms:.2f} ms / step (fp32 {t32:.2f}), {n_lp} lp launches")
            assert bool(torch.isfinite(wav).all())
    finally:
        ops.lp_min_flops = saved
    REPORT["configs2_mixed_policies"] = dict(fp32_ms_per_step=t32, policies=rows)
    REPORT["configs2_mixed"] = rows["mixed"]
    assert rows["mixed"]["item0_vs_oracle"] <= MIXED_BOUND and rows["mixed"]["all_vs_fp32_engine"] <= 1.5 * MIXED_BOUND, rows["mixed"]
    assert rows["mixed"]["lp_launches"] >= 40


def test_mixed_precision_on_outlier_stress_weights(ops):
    """The same policies on workload.weights.stress_vits_state (x50 channels at conv_pre, LayerNorm gains up to 30, SnakeBeta frequencies
    up to e): plain fp16 loses the prior encoder there (emulated: 5e-2); the default mixed policy keeps it in split-bf16."""
    hp = C.base_hp()
    m, sd = E.make_model(hp, ops, "cuda", stress=True)
    d = I.synth_clip(T=1000, hp=hp, seed=2, B=1)
    with torch.no_grad():
        src_o = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        wav_o = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src_o, d["enc_noise"])
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    run = lambda: m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"])
    rms = float(wav_o.pow(2).mean().sqrt())
    rows = {"f32": E.maxerr(run(), wav_o)}
    for pol in ["f16", "bf16x3", "mixed", "mixed:amp0=f16w2,amp1=f16w2,amp2=f16w2", "mixed:encattn=f16", "mixed:flow=bf16x3,amp1=bf16x3", "mixed:amp1=bf16x3,amp2=bf16x3"]:
        m.precision = pol
        try:
            rows[pol] = E.maxerr(run(), wav_o)
        finally:
            m.precision = None
    REPORT["stress_mixed_policies"] = dict(wave_rms=rms, wave_max_abs_err=rows)
    print("stress weights (rms %.3f): " % rms + ", ".join(f"{k} {v:.2e}" for k, v in rows.items()))
    assert rows["f32"] <= E.WAVE_TOL
    assert rows["mixed"] <= 4e-3 and rows["mixed"] < rows["f16"]        # (measured numbers: profiles/r04*_precision_report.json)


def test_whisper_15s_window_modes(ops, whisper):
    """The 15 s window every clip > 15 s uses (n = 1500 mel frames, Tw = 750: 12 M-tiles) in fp32 and in each mode."""
    ck, wm = whisper
    g = torch.Generator().manual_seed(15)
    mel = (torch.randn(1, 80, 1500, generator=g) * 0.5).clamp(-1, 1.5)
    nz = torch.randn(1, 80, 1500, generator=g)
    with torch.no_grad():
        ref = O.audio_encoder(ck["model_state_dict"], mel + 0.1 * nz, 20, 24)
    scale = float(ref.abs().max())
    out = wm.encoder(mel, nz, 0.1)
    assert out.shape == (1, 750, 1280)
    e32 = E.maxerr(out, ref) / scale
    REPORT["whisper15s_f32"] = dict(ppg_rel_err=e32)
    print(f"whisper 15 s fp32: rel err {e32:.2e}")
    assert e32 <= 1e-4
    for mode in MODES:
        wm.encoder.precision = mode
        try:
            out, n_lp = _count_lp(ops, lambda: wm.encoder(mel, nz, 0.1))
        finally:
            wm.encoder.precision = None
        e = E.maxerr(out, ref) / scale
        REPORT[f"whisper15s_{mode}"] = dict(ppg_rel_err=e, lp_launches=n_lp)
        print(f"whisper 15 s {mode}: rel err {e:.2e}, {n_lp} lp launches")
        assert n_lp >= 24 * 4 and e <= PPG_REL_BOUND[mode]


def test_extractors_in_bf16x3_match_the_reference_goldens(ops):
    """Rows N3 with split-bf16 GEMM operands: HuBERT-Soft units and CREPE posteriors against the REFERENCE's own outputs
    (tests/golden/hubert_soft_1s.npz, crepe_full_1s.npz) -- an fp32-class result (tolerances 5x the fp32 ones)."""
    e_h = E.check_hubert_golden(ops, "cuda", tol=5e-4, precision="bf16x3")
    e_c = E.check_crepe_golden(ops, "cuda", tol=1e-4, precision="bf16x3")
    REPORT["extractors_bf16x3"] = dict(hubert_units_max_abs=e_h, crepe_posterior_max_abs=e_c)
    print(f"bf16x3 extractors: hubert units err {e_h:.2e}, crepe posterior err {e_c:.2e}")


@pytest.mark.parametrize("mode,tol", [("f16", 2e-2), ("bf16", 1e-1)])
def test_crepe_full_16bit_activation_chain(ops, mode, tol):
    """CREPE `full` on 3 s: layers 2-6 take the pooling kernel's 16-bit rows through the _A16 GEMM kernels; posterior error against
    the fp32 oracle inside the mode's class."""
    err, same = E.check_crepe_precision(ops, "cuda", "full", 48000, mode, tol)
    REPORT.setdefault("crepe_a16", {})[mode] = dict(posterior_max_abs=err, f0_frames_equal_to_fp32_oracle=same)
    print(f"crepe full {mode} (16-bit activation chain): posterior err {err:.2e}, decoded F0 equal to the fp32 oracle's on {same:.4f} of the frames")
    assert same >= (0.97 if mode == "f16" else 0.90)

