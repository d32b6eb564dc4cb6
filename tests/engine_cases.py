"""End-to-end parity checks of the svcmi facade against the golden vectors of the real reference and
against the oracle; shared by the emulator tests (tiny shapes, CPU) and the GPU tests."""
import os

import numpy as np
import pytest
import torch

from workload import config as C
from workload import inputs as I
from oracle import svc_oracle as O
from workload import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WAVE_TOL = 1e-3     # north_star: max-abs on the waveform vs the reference CPU path
TIGHT = 1e-4        # what fp32 kernels are expected to reach (reported, asserted where stable)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


def make_model(hp, ops, device, seed=1234, stress=False):
    from svcmi import SynthesizerInfer
    m = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp, ops=ops)
    sd = W.make_vits_state(hp, seed=seed)
    if stress:
        sd = W.stress_vits_state(sd, hp)
    m.load_state_dict(sd)
    m.eval()
    m.to(device)
    return m, sd


def check_vits_golden(ops, device, tag, hp, tol=TIGHT):
    g = golden(tag)
    m, _ = make_model(hp, ops, device)
    d = I.synth_clip(T=int(g["T"]), hp=hp, seed=int(g["seed"]), B=int(g["B"]))
    lens = _t(g["lengths"])
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    e_src = maxerr(src, _t(g["source"]))
    # feed the GOLDEN source so the waveform comparison isolates inference()
    wav, parts = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, _t(g["source"]), noise=d["enc_noise"], return_parts=True)
    errs = dict(source=e_src, z_p=maxerr(parts["z_p"], _t(g["z_p"])), z=maxerr(parts["z"], _t(g["z"])), wave=maxerr(wav, _t(g["wave"])))
    pit16 = m.source2wav(_t(g["source"])[:1])
    assert np.abs(pit16.astype(np.int32) - g["pitwav"].astype(np.int32)).max() <= 1
    assert errs["source"] <= 5e-5, errs
    assert errs["z_p"] <= tol and errs["z"] <= tol, errs
    assert errs["wave"] <= min(tol * 5, WAVE_TOL), errs
    return errs


def check_generator_widths_against_oracle(ops, device, T=5, B=2, tol=TIGHT, precision=None, stress=False):
    """The base.yaml generator widths (320 -> 160, 80, 40, 20, 10 channels: 64x64 / 64x80 / 64x48 grouped GEMM tiles, the grouped
    fused VALU kernels at 20 / 10 channels, the streaming ups+noise kernels and the fused output layer) on a short ragged batch
    with the small prior encoder / flow of the tiny config, against the oracle."""
    hp = C.tiny_hp()
    hp["gen"] = dict(hp["gen"], upsample_initial_channel=320)
    m, sd = make_model(hp, ops, device, stress=stress)
    d = I.synth_clip(T=T, hp=hp, seed=21, B=B)
    lens = d["lengths"].clone()
    if B > 1:
        lens[-1] = max(1, T - 2)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    m.precision = precision
    saved = ops.lp_min_flops
    if precision is not None:
        ops.lp_min_flops = 0.0           # tiny shapes: force every eligible GEMM through the reduced-precision kernels
    ops.trace_begin()
    wav = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"])
    trace = ops.trace_end()
    ops.lp_min_flops = saved
    with torch.no_grad():
        o_src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        o_wav = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], lens, o_src, d["enc_noise"])
    errs = dict(source=maxerr(src, o_src), wave=maxerr(wav, o_wav))
    if stress:
        errs.update(stress_floor(sd, hp, d, lens, o_src, o_wav, wav))
        assert errs["source"] <= 5e-5 and errs["wave_vs_fp64"] <= WAVE_TOL and errs["wave"] <= WAVE_TOL, errs
        return errs
    if precision is not None:
        assert trace.get("svcmi_conv_gemm_lp", {}).get("launches", 0) >= 20 and trace.get("svcmi_conv_gemm_group_lp", {}).get("launches", 0) >= 6, \
            f"reduced-precision kernels did not run: { {k: v['launches'] for k, v in trace.items()} }"
        assert errs["wave"] > 0.0
        if precision in ("f16", "bf16", "f16w2"):      # the wide stages' grouped GEMMs read SnakeAlias's 16-bit rows
            if precision == "f16w2":           # ... and take the split-weight kernel there
                from svcmi import _lib
                assert _lib.PREC_F16W2_A16 in trace["svcmi_conv_gemm_group_lp"]["precisions"], trace["svcmi_conv_gemm_group_lp"]
            assert trace["svcmi_conv_gemm_group_lp"]["a16_launches"] >= 6, trace["svcmi_conv_gemm_group_lp"]
    assert errs["source"] <= 5e-5 and errs["wave"] <= min(tol * 5, WAVE_TOL), errs
    return errs


def check_mixed_precision_policy(ops, device, T=3, B=1):
    """The per-class precision switch of the stage host (svcmi_synth_model.class_prec): a policy that names ONE mode for every class must
    reproduce that mode's waveform bit for bit (same kernels, same images per launch); the default policy uses split-bf16 AND fp16
    launches in one pass and lands between fp32 and plain fp16; class names / modes are validated."""
    from svcmi import _lib
    hp = C.tiny_hp()
    hp["gen"] = dict(hp["gen"], upsample_initial_channel=320)
    m, sd = make_model(hp, ops, device)
    d = I.synth_clip(T=T, hp=hp, seed=23, B=B)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    saved = ops.lp_min_flops
    ops.lp_min_flops = 0.0
    outs, traces = {}, {}
    try:
        for pol in (None, "f16", "mixed:" + ",".join(f"{k}=f16" for k in _lib.CLASS_NAMES), "bf16x3",
                    "mixed:" + ",".join(f"{k}=bf16x3" for k in _lib.CLASS_NAMES), "mixed", "mixed:" + ",".join(f"{k}=f32" for k in _lib.CLASS_NAMES)):
            m.precision = pol
            ops.trace_begin()
            outs[pol] = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"]).clone()
            traces[pol] = ops.trace_end()
    finally:
        m.precision, ops.lp_min_flops = None, saved
    keys = list(outs)
    assert torch.equal(outs[keys[1]], outs[keys[2]]) and torch.equal(outs[keys[3]], outs[keys[4]])
    assert torch.equal(outs[None], outs[keys[6]])                            # every class fp32 == the fp32 model
    with torch.no_grad():
        o_src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        o_wav = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], o_src, d["enc_noise"])
    errs = {str(k): maxerr(v, o_wav) for k, v in outs.items()}
    precs = lambda tr: sorted({r for name in ("svcmi_conv_gemm_lp", "svcmi_conv_gemm_group_lp") for r in tr.get(name, {}).get("precisions", [])})
    assert errs["None"] <= TIGHT * 5 and errs["bf16x3"] <= 2e-4 and errs["mixed"] <= 4e-3 and errs["f16"] <= 4e-3, errs
    assert not torch.equal(outs["mixed"], outs["f16"]) and not torch.equal(outs["mixed"], outs["bf16x3"])
    ran = precs(traces["mixed"])
    assert _lib.PREC_BF16X3 in ran and (_lib.PREC_F16 in ran or _lib.PREC_F16_A16 in ran), ran      # both families in ONE pass
    # the narrow stages (classes amp3 / amp4 = f16w2 by default): their half-steps ran on the fp16 matrix cores, and only there
    assert traces["mixed"].get("svcmi_snake_conv_group_lp", {}).get("launches", 0) >= 12, {k: v["launches"] for k, v in traces["mixed"].items()}
    assert "svcmi_snake_conv_group_lp" not in traces[None] and "svcmi_snake_conv_group_lp" not in traces["bf16x3"]
    for bad in ("mixed:decoder=f16", "mixed:enc=int8", "fp8"):
        with pytest.raises(_lib.SvcmiError):
            _lib.parse_precision(bad)
    return errs, {str(k): precs(v) for k, v in traces.items()}


def check_whisper_golden(ops, device, tag, dims, tol=TIGHT, precision=None):
    from svcmi.whisper.inference import load_model
    g = golden(tag)
    ck = W.make_whisper_state(dims)
    wm = load_model(ck, device, ops=ops)
    wm.encoder.precision = precision
    saved = ops.lp_min_flops
    if precision is not None:
        ops.lp_min_flops = 0.0
    ops.trace_begin()
    out = wm.encoder(_t(g["mel"]), _t(g["mel_noise"]), 0.1)
    trace = ops.trace_end()
    ops.lp_min_flops = saved
    if precision is not None:
        assert trace.get("svcmi_conv_gemm_lp", {}).get("launches", 0) >= 4 * len(wm.weights.blocks), "reduced-precision kernels did not run"
        # ... and the block GEMMs took their A operand as 16-bit rows from the producer (LayerNorm / attention / GELU epilogue);
        # bf16x3: the QKV projection only (split rows)
        per_block = 1 if precision == "bf16x3" else 4
        assert trace["svcmi_conv_gemm_lp"]["a16_launches"] >= per_block * len(wm.weights.blocks), trace["svcmi_conv_gemm_lp"]
    err = maxerr(out, _t(g["ppg"]))
    assert err <= tol * max(1.0, float(np.abs(g["ppg"]).max())), err
    return err


def check_svc_infer_golden(ops, device, tol=TIGHT):
    from svcmi import DummyRetrieval, svc_infer
    g = golden("svc_infer_tiny_2chunks")
    hp = C.tiny_hp()
    T = int(g["T"])
    m, _ = make_model(hp, ops, device)
    d = I.synth_clip(T=T, hp=hp, seed=2, B=1)
    gen = torch.Generator().manual_seed(77)
    enc_noises = [torch.randn(1, hp.vits.inter_channels, ce - cs, generator=gen) for (cs, ce, _, _) in O.chunk_schedule(T, 320)]
    wav = svc_infer(m, DummyRetrieval(), d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0], hp, device,
                    noise={"rand_ini": d["rand_ini"], "src_noise": d["src_noise"], "enc_noises": enc_noises},
                    write_pit_wav=False)
    assert wav.dtype == np.float32 and wav.shape[0] == int(g["length"]) == T * 320 - 1
    seam = C.CHUNK_FRAMES * 320
    errs = dict(sub=float(np.abs(wav[::97] - g["wave_sub"]).max()),
                seam=float(np.abs(wav[seam - 3000:seam + 3000] - g["wave_seam"]).max()),
                tail=float(np.abs(wav[-2000:] - g["wave_tail"]).max()))
    assert max(errs.values()) <= min(tol * 5, WAVE_TOL), errs
    return errs


def check_svc_infer_retrieval(ops, device, T=90, tol=2e-4, check_changed=True):
    """svc_infer with the on-device kNN blend (svcmi.feature_retrieval) vs the same engine fed through a CPU IRetrieval
    hook that runs the retrieval oracle -- checks the wiring of row N4 into the chunk loop (features stay on the device,
    same chunk slices), the blend itself being covered by kernel_cases.check_knn_blend."""
    from oracle import retrieval_oracle as RO
    from svcmi import IRetrieval, svc_infer
    from svcmi.feature_retrieval import KnnFeatureIndex, KnnIndexRetrieval
    hp = C.tiny_hp()
    m, _ = make_model(hp, ops, device)
    d = I.synth_clip(T=T, hp=hp, seed=5, B=1)
    gen = torch.Generator().manual_seed(9)
    banks = {"ppg": torch.randn(400, hp.vits.ppg_dim, generator=gen).numpy(), "vec": torch.randn(333, hp.vits.vec_dim, generator=gen).numpy()}
    enc_noises = [torch.randn(1, hp.vits.inter_channels, ce - cs, generator=gen) for (cs, ce, _, _) in O.chunk_schedule(T, 320)]
    noise = {"rand_ini": d["rand_ini"], "src_noise": d["src_noise"], "enc_noises": enc_noises}

    class OracleHook(IRetrieval):
        def retriv_whisper(self, vec):
            assert vec.device.type == "cpu"
            return torch.from_numpy(RO.retriv(vec.numpy(), banks["ppg"], 0.5, 3))

        def retriv_hubert(self, vec):
            return torch.from_numpy(RO.retriv(vec.numpy(), banks["vec"], 0.5, 3))

    knn = KnnIndexRetrieval(hubert_index=KnnFeatureIndex(banks["vec"], 0.5, 3, device=device, ops=ops),
                            whisper_index=KnnFeatureIndex(banks["ppg"], 0.5, 3, device=device, ops=ops))
    args = (d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0], hp, device)
    got = svc_infer(m, knn, *args, noise=noise, write_pit_wav=False)
    want = svc_infer(m, OracleHook(), *args, noise=noise, write_pit_wav=False)
    err = float(np.abs(got - want).max())
    assert err <= tol, err
    if check_changed:      # the blend actually changed the features
        plain = svc_infer(m, None, *args, noise=noise, write_pit_wav=False)
        assert float(np.abs(got - plain).max()) > 100 * tol
    return err


def check_logmel_golden(ops, device, tol=2e-4):
    """GPU log-mel front-end (svcmi.whisper.audio) vs the reference's own log_mel_spectrogram (golden fixture)."""
    from oracle import audio_oracle as A
    from svcmi.whisper import audio as PA
    g = golden("logmel_2p5s")
    x = A.synth_audio(int(g["n"]), int(g["seed"]))
    got = PA.log_mel_spectrogram(x, ops=ops, device=device)
    assert tuple(got.shape) == tuple(g["logmel"].shape)
    err = maxerr(got, _t(g["logmel"]))
    # log10 of a bin at the (max - 8) floor amplifies fp32 round-off of the 400-point transform; the reference's own
    # fp32 FFT has the same noise there, so the tolerance is on the log-mel value, not on the power
    assert err <= tol, err
    return err


def check_hubert_against_oracle(ops, device, dims, n, heads, tol=TIGHT):
    """HuBERT-Soft units (svcmi.hubert) vs the oracle on seeded weights of ``dims`` and a seeded waveform."""
    from oracle import hubert_oracle as H
    from svcmi.hubert import load_model
    sd = W.make_hubert_state(dims)
    m = load_model(sd, device, ops=ops)
    g = torch.Generator().manual_seed(3)
    wav = torch.randn(2, 1, n, generator=g) * 0.3
    got = m.units(wav)
    with torch.no_grad():
        want = H.units(sd, wav, heads)
    assert tuple(got.shape) == tuple(want.shape) == (2, (n + 80 - 400) // 320 + 1, dims["proj"])
    err = maxerr(got, want)
    assert err <= tol * max(1.0, float(want.abs().max())), err
    return err


def check_hubert_golden(ops, device, tol=TIGHT, precision=None):
    """svcmi.hubert at the reference dimensions vs the reference HubertSoft.units itself (golden fixture)."""
    from svcmi.hubert import load_model
    g = golden("hubert_soft_1s")
    sd = W.make_hubert_state()
    m = load_model(sd, device, ops=ops)
    m.precision = precision
    gen = torch.Generator().manual_seed(int(g["seed"]))
    wav = torch.randn(1, 1, int(g["n"]), generator=gen) * 0.3
    err = maxerr(m.units(wav), _t(g["units"]))
    assert err <= tol * max(1.0, float(np.abs(g["units"]).max())), err
    return err


def crepe_test_audio(n, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    return (0.4 * torch.sin(2 * np.pi * (180 + 60 * torch.sin(2 * np.pi * 1.5 * t)) * t) + 0.02 * torch.randn(n, generator=g)).float()


def check_crepe_against_oracle(ops, device, capacity, n, tol=2e-5):
    """CREPE posteriors on the GPU kernels vs the oracle, then compute_f0_sing (Viterbi, injected noise/dither) end to end."""
    from oracle import crepe_oracle as CO
    from svcmi.pitch import compute_f0_sing, load_crepe
    sd = W.make_crepe_state(capacity)
    m = load_crepe(sd, device, ops=ops)
    audio = crepe_test_audio(n, 5)
    got = m.probabilities(audio, hop=320)
    with torch.no_grad():
        want = CO.network(sd, CO.preprocess(audio[None], 320))
    assert tuple(got.shape) == tuple(want.shape) == (1 + n // 320, 360)
    err = maxerr(got, want)
    assert err <= tol, err
    g = torch.Generator().manual_seed(9)
    noise = torch.randn(n, generator=g)
    dither = (torch.rand(1 + n // 320, generator=g) * 2 - 1).numpy() * 20.0
    f0 = compute_f0_sing(audio, device, model=m, noise=noise, dither=dither)
    with torch.no_grad():
        f0_ref = CO.compute_f0_sing(sd, audio, noise, dither).numpy()
    assert f0.shape == f0_ref.shape == (2 * (1 + n // 320),)
    # the decoded path is discrete: posteriors that agree to 1e-5 give the same bins except at exact ties
    same = np.isclose(f0, f0_ref, rtol=1e-5, atol=1e-3, equal_nan=True)
    assert same.mean() >= 0.98, float(same.mean())
    return err, float(same.mean())


def check_crepe_precision(ops, device, capacity, n, precision, tol):
    """CREPE posteriors in a reduced-precision mode vs the fp32 oracle: layers 2-6 read the 16-bit rows the pooling kernel wrote
    (the _A16 GEMM kernels; bf16x3: split rows), error inside the mode's class."""
    from oracle import crepe_oracle as CO
    from svcmi._lib import PRECISIONS
    from svcmi.pitch import load_crepe
    sd = W.make_crepe_state(capacity)
    m = load_crepe(sd, device, ops=ops)
    m.precision = precision
    audio = crepe_test_audio(n, 5)
    saved, ops.lp_min_flops = ops.lp_min_flops, 0.0
    got = m.probabilities(audio, hop=320)
    ops.lp_min_flops = saved
    with torch.no_grad():
        want = CO.network(sd, CO.preprocess(audio[None], 320))
    err = maxerr(got, want)
    code = PRECISIONS[precision] + 2
    for L in m.w.layers[1:]:          # (the short layers run as dense GEMMs over the frames: weights.CrepeWeights)
        assert getattr(L.get("dense_w", L["w"]), "_svcmi_lp", {}).get(code) is not None, "the 16-bit-activation kernel did not run"
    assert 0.0 < err <= tol, err
    # what the path consumes is the decoded F0 track (Viterbi over the 360 bins): the share of frames whose Hz value is the oracle's
    from svcmi.pitch import compute_f0_sing
    g = torch.Generator().manual_seed(9)
    noise = torch.randn(n, generator=g)
    dither = (torch.rand(1 + n // 320, generator=g) * 2 - 1).numpy() * 20.0
    ops.lp_min_flops = 0.0
    f0 = compute_f0_sing(audio, device, model=m, noise=noise, dither=dither)
    ops.lp_min_flops = saved
    with torch.no_grad():
        f0_ref = CO.compute_f0_sing(sd, audio, noise, dither).numpy()
    same = float(np.isclose(f0, f0_ref, rtol=1e-5, atol=1e-3, equal_nan=True).mean())
    return err, same


def check_crepe_golden(ops, device, tol=2e-5, precision=None):
    """svcmi.pitch.Crepe at the reference's `full` capacity vs the reference crepe package itself (golden fixture)."""
    from svcmi.pitch import decode, load_crepe
    g = golden("crepe_full_1s")
    m = load_crepe(W.make_crepe_state("full"), device, ops=ops)
    m.precision = precision
    audio = crepe_test_audio(int(g["n"]), int(g["seed"]))
    prob = m.probabilities(audio, hop=320).cpu()
    err = maxerr(prob, _t(g["prob"]))
    assert err <= tol, err
    f0 = decode(prob, 50.0, 1000.0, "argmax", np.zeros(prob.shape[0])).numpy()
    assert np.isclose(f0, g["f0_argmax"], rtol=1e-5).mean() >= 0.98
    return err


def config0_noise(g, T, hp):
    """The draws make_golden.config0_fixture injected into the reference (one generator, fixed order)."""
    gen = torch.Generator().manual_seed(int(g["noise_seed"]))
    mel_noise = torch.randn(80, g["logmel"].shape[1], generator=gen)
    rand_ini = torch.rand(1, 11, generator=gen)
    src_noise = torch.randn(1, T * hp.data.hop_length, 11, generator=gen)
    enc_noises = [torch.randn(1, hp.vits.inter_channels, T, generator=gen)]
    return mel_noise, rand_ini, src_noise, enc_noises


def check_config0_wav_to_wav(ops, device, tol=TIGHT):
    """BASELINE.json configs[0] on the reference's own fixture: tests/golden/035.wav (3.99 s) + singer0001.npy through the
    engine's file-level path -- 16 kHz loader -> log-mel -> Whisper (large-v2 dims) -> PPG; HuBERT-soft -> vec; np.repeat x2;
    svc_infer -- against what the REFERENCE's pred_ppg / pred_vec / svc_infer produced for the same file, weights and noise
    (oracle/make_golden.py config0_fixture).  F0 comes from the fixture the way `--pit` passes a CSV."""
    from svcmi import DummyRetrieval, svc_infer
    from svcmi.hubert import inference as hubert_inf
    from svcmi.whisper import audio as A
    from svcmi.whisper.inference import load_model, pred_ppg_from_mel, window_plan
    g = golden("config0_035")
    hp = C.base_hp()
    T = int(g["T"])
    wav_path, spk_path = os.path.join(GOLDEN, "035.wav"), os.path.join(GOLDEN, "singer0001.npy")
    mel_noise, rand_ini, src_noise, enc_noises = config0_noise(g, T, hp)
    audio = A.load_audio(wav_path)
    assert audio.shape == (63902,) and audio.dtype == np.float32
    ck = W.make_whisper_state(C.WHISPER_LARGE_V2)
    wm = load_model(ck, device, ops=ops)
    plan = window_plan(audio.shape[0])
    assert plan == [(0, 63902, 199)]
    mels = [A.log_mel_spectrogram(torch.from_numpy(audio[s:e]), ops=ops, device=device) for (s, e, _) in plan]
    ppg = pred_ppg_from_mel(wm, mels, [k for (_, _, k) in plan], mel_noises=[mel_noise.to(device)])
    hub = hubert_inf.load_model(W.make_hubert_state(), device, ops=ops)
    vec = torch.cat([hub.units(torch.from_numpy(audio[s:e]).view(1, 1, -1))[0] for (s, e) in hubert_inf.window_plan(audio.shape[0])], 0)
    errs = dict(logmel=maxerr(mels[0], _t(g["logmel"])), ppg=maxerr(ppg, _t(g["ppg"])), vec=maxerr(vec, _t(g["vec"])))
    m, _ = make_model(hp, ops, device)
    spk = torch.FloatTensor(np.load(spk_path))
    ppg2 = torch.FloatTensor(np.repeat(ppg.cpu().numpy(), 2, 0))            # svc_inference.py:175-182
    vec2 = torch.FloatTensor(np.repeat(vec.cpu().numpy(), 2, 0))
    wav = svc_infer(m, DummyRetrieval(), spk, _t(g["pit"]), ppg2, vec2, hp, device,
                    noise={"rand_ini": rand_ini, "src_noise": src_noise, "enc_noises": enc_noises}, write_pit_wav=False)
    assert wav.dtype == np.float32 and wav.shape == g["wave"].shape == (T * hp.data.hop_length - 1,)
    errs["wave"] = float(np.abs(wav - g["wave"]).max())
    assert errs["logmel"] <= 1e-4 and errs["ppg"] <= tol * max(1.0, float(np.abs(g["ppg"]).max())) and errs["vec"] <= tol * 10, errs
    assert errs["wave"] <= min(tol * 5, WAVE_TOL), errs
    return errs


def check_whisper_stress(ops, device, dims, n, tol=TIGHT):
    """Outlier-stress weights (workload.weights.stress_whisper_state: massive residual channels, LayerNorm gains up to 30,
    saturated GELU inputs, peaked softmax) through the encoder vs the oracle; also reports how large the residual stream got."""
    from svcmi.whisper.inference import load_model
    ck = W.stress_whisper_state(W.make_whisper_state(dims))
    wm = load_model(ck, device, ops=ops)
    g = torch.Generator().manual_seed(n)
    mel = (torch.randn(1, 80, n, generator=g) * 0.5).clamp(-1, 1.5)
    nz = torch.randn(1, 80, n, generator=g)
    out = wm.encoder(mel, nz, 0.1)
    with torch.no_grad():
        ref = O.audio_encoder(ck["model_state_dict"], mel + 0.1 * nz, dims["n_audio_head"], O.whisper_kept_layers(dims))
    scale = float(ref.abs().max())
    err = maxerr(out, ref)
    assert scale > 20.0, scale                      # the LayerNorm gains really produced outlier-scale outputs
    assert err <= tol * scale, (err, scale)
    return dict(err=err, out_max=scale)


def check_whisper_batched_rows_flattened(ops, device, dims, B=3, n=90, small_m_rows=48, tol=TIGHT):
    """Batched windows above ``small_m_rows`` GEMM rows: the four k = 1 projections of a block see ONE matrix of B * tw rows (M tiles span
    batch items, host_stages.hip: whisper_fwd).  Every item equals its solo run on the library's own tiles (bit for bit) and the oracle."""
    from svcmi.whisper.inference import load_model
    ck = W.make_whisper_state(dims)
    wm = load_model(ck, device, ops=ops)
    wm.encoder.small_m_rows = small_m_rows
    g = torch.Generator().manual_seed(n + B)
    mel = (torch.randn(B, 80, n, generator=g) * 0.5).clamp(-1, 1.5)
    nz = torch.randn(B, 80, n, generator=g)
    tw = (n - 1) // 2 + 1
    assert B * tw > small_m_rows >= tw, "batched rows above the threshold, one item below it"
    out = wm.encoder(mel, nz, 0.1)
    wm.encoder.small_m_rows = 1                 # solo runs on the same (library-chosen) tiles: a lone item is never flattened (B = 1)
    for i in range(B):
        solo = wm.encoder(mel[i:i + 1], nz[i:i + 1], 0.1)
        assert torch.equal(out[i:i + 1], solo), f"item {i}: max diff {float((out[i:i + 1] - solo).abs().max()):.3e}"
    with torch.no_grad():
        ref = O.audio_encoder(ck["model_state_dict"], mel + 0.1 * nz, dims["n_audio_head"], O.whisper_kept_layers(dims))
    err = maxerr(out, ref)
    assert err <= tol * max(1.0, float(ref.abs().max())), err
    return err


def stress_floor(sd, hp, d, lens, o_src, o_wav, wav):
    """Noise floor of an outlier-stress run: the oracle in fp64 vs itself in fp32, and the engine vs the fp64 result.  The
    stress sets are tuned to stay well-conditioned (asserted: the fp32 oracle stays within 3e-4 of fp64 -- with LayerNorm
    gains of 30 both fp32 implementations sit 1e-5..3e-4 from fp64, in either order depending on the clip), so a failure of
    the 1e-3 bar cannot be blamed on chaos."""
    with torch.no_grad():
        sd64 = {k: v.double() for k, v in sd.items()}
        o64 = O.synth_inference(sd64, hp, d["ppg"].double(), d["vec"].double(), d["pit"].double(), d["spk"].double(), lens,
                                o_src.double(), d["enc_noise"].double())
    floor = float((o_wav.double() - o64).abs().max())
    assert floor <= 3e-4, f"stress set is ill-conditioned: fp32 vs fp64 oracle {floor:.2e}"
    return dict(oracle_fp32_vs_fp64=floor, wave_vs_fp64=float((wav.detach().cpu().double() - o64).abs().max()))


def check_streaming_decoder(ops, device, hp, T, tiles, B=1, seed=41, precision=None):
    """In-chunk time tiling of the generator (SynthesizerInfer.stream_frames, BASELINE.json configs[4]): every tile size
    gives the SAME BITS (halo 32 frames >= the exact receptive field of 30.9 frames, position-independent kernel arithmetic, split-K pinned off), a
    single tile covering the chunk is the untiled network, and the result stays at fp32 round-off from the default path
    (which lets the library pick split-K).  Returns the max difference to the default path."""
    m, sd = make_model(hp, ops, device)
    d = I.synth_clip(T=T, hp=hp, seed=seed, B=B)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    run = lambda: m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"]).clone()
    m.precision = precision                  # (a reduced-precision mode: the same bit-identity of the tiles; the kernel choice of every
    base = run()                             # launch must then not depend on the tile length either, e.g. the narrow stages' matrix-core forms)
    m.stream_frames = 10 * T                 # one tile: the untiled generator with split-K off
    whole = run()
    outs = []
    for S in tiles:
        m.stream_frames = S
        outs.append(run())
    m.stream_frames = None
    m.precision = None
    if precision is not None:
        # 16-bit modes: which launches take the 16-bit kernels follows their size (Ops.lp_min_flops: a short tile's small GEMMs stay fp32), so
        # tiles differ from the whole chunk by roundings of that mode -- bounded, not zero; the fp32 bit-identity is the contract
        return max(maxerr(o, whole) for o in outs + [base])
    for S, o in zip(tiles, outs):
        assert torch.equal(o, whole), (S, float((o - whole).abs().max()))
    err = maxerr(whole, base)
    assert err <= 1e-5, err
    return err


def check_clips_in_flight(ops, device, lanes=3, rounds=3, T=300, layers=4):
    """svcmi.lanes.GraphLanes: ``lanes`` clips (own inputs, own noise) captured on their own streams and replayed concurrently, against
    every clip alone on the current stream -- bit-identical, i.e. lanes share nothing but the read-only weights (the split-K
    workspaces, which the Whisper attention-out / MLP-down and the flow layers write, are per stream).  Full-width Whisper (fewer
    blocks) + base.yaml synthesizer, so the launches are the split-K / grouped shapes of the judged line."""
    from svcmi.lanes import GraphLanes
    from svcmi.whisper.inference import load_model
    hp = C.base_hp()
    m, _ = make_model(hp, ops, device)
    wm = load_model(W.make_whisper_state(dict(C.WHISPER_LARGE_V2, n_audio_layer=layers)), device, ops=ops)
    fns, want = [], []
    for i in range(lanes):
        d = {k: v.to(device) for k, v in I.synth_clip(T=T, hp=hp, seed=50 + i, B=1, ppg=False).items()}
        lens = d["lengths"].to(torch.int32)

        def fn(d=d, lens=lens):
            ppg50 = wm.encoder(d["mel"], d["mel_noise"], 0.1)[:, :T // 2]
            src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
            return m.inference_ppg50(ppg50, d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"])
        fns.append(fn)
        want.append(fn().clone())
    assert not torch.equal(want[0], want[1])
    L = GraphLanes(fns)
    for _ in range(rounds * lanes):
        L.launch()
    L.synchronize()
    for i in range(lanes):
        assert torch.equal(L.outputs[i], want[i]), f"lane {i}: max diff {float((L.outputs[i] - want[i]).abs().max()):.3e}"
    i = L.launch()
    assert torch.equal(L.wait(i), want[i]) and len(L) == lanes


def check_chunk_streams(ops, device, T=5300, streams=3):
    """svc_infer with the synthesis chunks of one clip in flight on ``streams`` HIP streams vs one after the other: same launches, pinned
    noise -> bit-identical waveform (3 chunks: full, full, remainder)."""
    from svcmi import DummyRetrieval, svc_infer
    hp = C.tiny_hp()
    m, _ = make_model(hp, ops, device)
    d = I.synth_clip(T=T, hp=hp, seed=4, B=1)
    gen = torch.Generator().manual_seed(78)
    plan = O.chunk_schedule(T, 320)
    assert len(plan) >= 3
    enc_noises = [torch.randn(1, hp.vits.inter_channels, ce - cs, generator=gen) for (cs, ce, _, _) in plan]
    args = (m, DummyRetrieval(), d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0], hp, device)
    kw = dict(noise={"rand_ini": d["rand_ini"], "src_noise": d["src_noise"], "enc_noises": enc_noises}, write_pit_wav=False)
    m.chunk_streams = 1
    want = svc_infer(*args, **kw)
    m.chunk_streams = streams
    for _ in range(3):
        got = svc_infer(*args, **kw)
        assert np.array_equal(got, want), float(np.abs(got - want).max())
    pools = m.__dict__.get("_svcmi_chunk_streams", {})
    assert sum(len(v) for v in pools.values()) == (streams if str(device) != "cpu" else 0)


def check_hubert_windows_batched(ops, device, dims, seconds=45.0):
    """pred_vec's 20 s window loop with equal windows as one batch vs one window at a time: same units (fp32 round-off)."""
    from svcmi.hubert import load_model
    from svcmi.hubert.inference import units_windowed, window_plan
    m = load_model(W.make_hubert_state(dims), device, ops=ops)
    audio = (torch.randn(int(seconds * 16000), generator=torch.Generator().manual_seed(8)) * 0.3).numpy()
    plan = window_plan(audio.shape[0])
    assert len(plan) == 3 and plan[0][1] - plan[0][0] == plan[1][1] - plan[1][0] != plan[2][1] - plan[2][0]
    got = units_windowed(m, audio)
    solo = units_windowed(m, audio, max_batch=1)
    assert got.shape == solo.shape
    err = maxerr(got, solo.cpu())
    assert err <= 2e-5 * max(1.0, float(solo.abs().max())), err
    return err


def check_clip_lanes(ops, device, lanes=3, requests=7, T=300, layers=4, precision=None, capture_first=False):
    """svcmi.serving.ClipLanes: ``requests`` different clips (own inputs, pinned noise, one of them ragged) submitted through ``lanes`` lanes,
    collected in order, each against the same conversion run eagerly on the current stream -- bit-identical.
    ``precision`` + ``capture_first``: the graph capture is the FIRST thing that runs the models in a reduced-precision mode, i.e. the
    lazily built 16-bit weight images are packed inside ClipLanes.capture (round 6: that deadlocked on a non-reentrant Ops lock)."""
    from svcmi.serving import ClipLanes, convert_step
    from svcmi.whisper.inference import load_model
    hp = C.base_hp()
    m, _ = make_model(hp, ops, device)
    wm = load_model(W.make_whisper_state(dict(C.WHISPER_LARGE_V2, n_audio_layer=layers)), device, ops=ops)
    if precision:
        m.precision = precision
        wm.encoder.precision = "f16" if precision.startswith("mixed") else precision
    cl = ClipLanes(m, wm, T, B=1, lanes=lanes, device=device, pinned_noise=True)
    if capture_first:
        d0 = {k: v.to(device) for k, v in I.synth_clip(T=T, hp=hp, seed=69, B=1, ppg=False).items()}
        for lane in range(lanes):
            cl.stage(lane, noise={k: d0[k] for k in ("mel_noise", "rand_ini", "src_noise", "enc_noise")},
                     lengths=torch.tensor([T], dtype=torch.int32), **{k: d0[k] for k in ("mel", "vec", "pit", "spk")})
        cl.capture()
    reqs, want = [], []
    for i in range(requests):
        d = {k: v.to(device) for k, v in I.synth_clip(T=T, hp=hp, seed=70 + i, B=1, ppg=False).items()}
        lens = torch.tensor([T if i != 2 else T - 37], dtype=torch.int32, device=device)
        noise = {k: d[k] for k in ("mel_noise", "rand_ini", "src_noise", "enc_noise")}
        buf = dict(mel=d["mel"], vec=d["vec"], pit=d["pit"], spk=d["spk"], lengths=lens)
        reqs.append((buf, noise))
        want.append(convert_step(m, wm, buf, T // 2, noise).clone())
    pending, got = [], []
    for buf, noise in reqs:
        if len(pending) == lanes:                         # all lanes busy: collect the oldest first
            got.append(cl.result(pending.pop(0)))
        pending.append(cl.submit(noise=noise, lengths=buf["lengths"], **{k: buf[k] for k in ("mel", "vec", "pit", "spk")}))
    got += [cl.result(t) for t in pending]
    assert len(got) == requests
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g, w), f"request {i}: max diff {float((g - w).abs().max()):.3e}"
    assert not torch.equal(want[0], want[1])
