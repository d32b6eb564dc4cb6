"""End-to-end parity on a real MI355X: the svcmi facade against the golden vectors of the real
reference (tests/golden, made by oracle/make_golden.py) and against the oracle at the full 10 s
configuration (BASELINE.json configs[1]).  Tolerance: north_star's 1e-3 max-abs on the waveform;
the fp32 kernels are expected ~1e-5 and the tests print what they reach."""
import os

import numpy as np
import pytest
import torch

from workload import config as C
from workload import inputs as I
from oracle import svc_oracle as O
from workload import weights as W
from tests import engine_cases as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from svcmi import Ops
    o = Ops()
    assert o.build == "hip:gfx950" and o.on_gpu
    return o


def test_vits_tiny_ragged_golden(ops):
    print(E.check_vits_golden(ops, "cuda", "vits_tiny_ragged", C.tiny_hp()))


def test_vits_base_T60_golden(ops):
    print(E.check_vits_golden(ops, "cuda", "vits_base_T60", C.base_hp()))


def test_whisper_tiny_golden(ops):
    print(E.check_whisper_golden(ops, "cuda", "whisper_tiny", C.WHISPER_TINY_TEST))


def test_whisper_large_v2_golden(ops):
    print(E.check_whisper_golden(ops, "cuda", "whisper_large_v2_n200", C.WHISPER_LARGE_V2))


def test_logmel_frontend_golden(ops):
    print(E.check_logmel_golden(ops, "cuda"))


def test_logmel_15s_window_against_oracle(ops):
    """One full 15 s Whisper window (whisper/inference.py:37): wav -> log-mel on the GPU vs the CPU oracle."""
    from oracle import audio_oracle as A
    from svcmi.whisper import audio as PA
    x = A.synth_audio(15 * 16000, 5)
    got = PA.log_mel_spectrogram(x, ops=ops, device="cuda")
    ref = A.log_mel_spectrogram(x)
    assert got.shape == (80, 1500)
    assert E.maxerr(got, ref) <= 2e-4


def test_hubert_soft_golden(ops):
    print(E.check_hubert_golden(ops, "cuda"))


def test_hubert_soft_10s_against_oracle(ops):
    """HuBERT-Soft at the reference dimensions on a 10 s window (T = 500 frames) vs the CPU oracle."""
    print(E.check_hubert_against_oracle(ops, "cuda", C.HUBERT_SOFT, n=160000, heads=12))


def test_hubert_equal_windows_run_as_one_batch(ops):
    E.check_hubert_windows_batched(ops, "cuda", C.HUBERT_SOFT)


def test_hubert_tiny_against_oracle(ops):
    print(E.check_hubert_against_oracle(ops, "cuda", C.HUBERT_TINY_TEST, n=4000, heads=4))


def test_crepe_full_golden(ops):
    print(E.check_crepe_golden(ops, "cuda"))


def test_crepe_full_10s_against_oracle(ops):
    """compute_f0_sing at the reference's `full` capacity on 3 s (151 frames; the CPU oracle needs ~1.4 GMAC per frame)."""
    print(E.check_crepe_against_oracle(ops, "cuda", "full", n=48000))


def test_crepe_tiny_against_oracle(ops):
    print(E.check_crepe_against_oracle(ops, "cuda", "tiny", n=16000 * 12))     # 601 frames: two decoding batches


def test_svc_infer_two_chunks_golden(ops):
    print(E.check_svc_infer_golden(ops, "cuda"))


def test_svc_infer_with_knn_retrieval(ops):
    print(E.check_svc_infer_retrieval(ops, "cuda", T=300))


def test_generator_base_widths_match_oracle(ops):
    print(E.check_generator_widths_against_oracle(ops, "cuda", T=37, B=3))


def test_full_10s_clip_against_oracle(ops):
    """configs[1]: B=1, 10 s, base.yaml decoder; pitch2source + inference vs the CPU oracle, same noise."""
    hp = C.base_hp()
    m, sd = E.make_model(hp, ops, "cuda")
    d = I.synth_clip(T=1000, hp=hp, seed=0, B=1)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    wav = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"])
    torch.cuda.synchronize()
    with torch.no_grad():
        o_src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        o_wav = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], o_src, d["enc_noise"])
    e_src, e_wav = E.maxerr(src, o_src), E.maxerr(wav, o_wav)
    print(f"10 s clip: source err {e_src:.2e}, waveform err {e_wav:.2e}, rms {float(o_wav.pow(2).mean().sqrt()):.3f}")
    assert wav.shape == (1, 1, 320000)
    assert e_src <= 5e-5 and e_wav <= E.WAVE_TOL


def test_equal_length_batch_reproduces_solo_runs(ops):
    """Size-independent property: items of an equal-length batch equal their solo runs."""
    hp = C.base_hp()
    m, _ = E.make_model(hp, ops, "cuda")
    d = I.synth_clip(T=200, hp=hp, seed=4, B=3)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    wav = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"])
    for b in range(3):
        s1 = m.pitch2source(d["pit"][b:b + 1], noise=(d["rand_ini"][b:b + 1], d["src_noise"][b:b + 1]))
        w1 = m.inference(d["ppg"][b:b + 1], d["vec"][b:b + 1], d["pit"][b:b + 1], d["spk"][b:b + 1],
                         d["lengths"][b:b + 1], s1, noise=d["enc_noise"][b:b + 1])
        assert E.maxerr(s1, src[b:b + 1]) == 0.0
        assert E.maxerr(w1, wav[b:b + 1]) <= 1e-5
    assert wav.shape[-1] == 320 * 200                      # out_len == hop * T


def test_batch16_x_10s_items_equal_solo_runs(ops):
    """BASELINE.json configs[2] shape (16 x 10 s clips, flow + decoder, pre-extracted PPG/F0) in fp32: every item of the
    batch must reproduce its solo run -- the size-independent property that stands in for a 16-clip CPU oracle run --
    and item 0 is the clip test_full_10s_clip_against_oracle checks against the oracle."""
    hp = C.base_hp()
    m, _ = E.make_model(hp, ops, "cuda")
    B = 16
    items = [I.synth_clip(T=1000, hp=hp, seed=s, B=1) for s in range(B)]
    d = {k: torch.cat([it[k] for it in items], 0) for k in items[0]}
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    wav = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"])
    assert wav.shape == (B, 1, 320000) and bool(torch.isfinite(wav).all())
    for b in (0, 7, 15):
        it = items[b]
        s1 = m.pitch2source(it["pit"], noise=(it["rand_ini"], it["src_noise"]))
        w1 = m.inference(it["ppg"], it["vec"], it["pit"], it["spk"], it["lengths"], s1, noise=it["enc_noise"])
        assert E.maxerr(s1, src[b:b + 1]) == 0.0
        assert E.maxerr(w1, wav[b:b + 1]) <= 1e-5


def test_30s_clip_second_chunk_against_oracle(ops):
    """BASELINE.json configs[4] shape: a 30 s clip = synth chunks [0,2510) and [2490,3000) (svc_inference.py:101-131).
    The whole clip runs through svc_infer on the GPU; the oracle recomputes the second chunk (5.1 s, so the CPU side
    stays bounded) and the kept samples [2500*320, L-1) must match, as must the total length L-1."""
    from svcmi import DummyRetrieval, svc_infer
    hp = C.base_hp()
    m, sd = E.make_model(hp, ops, "cuda")
    T, hop = 3000, 320
    d = I.synth_clip(T=T, hp=hp, seed=31, B=1)
    plan = O.chunk_schedule(T, hop)
    assert [(a, b) for (a, b, _, _) in plan] == [(0, 2510), (2490, 3000)]
    gen = torch.Generator().manual_seed(3)
    enc_noises = [torch.randn(1, hp.vits.inter_channels, ce - cs, generator=gen) for (cs, ce, _, _) in plan]
    wav = svc_infer(m, DummyRetrieval(), d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0], hp, "cuda",
                    noise={"rand_ini": d["rand_ini"], "src_noise": d["src_noise"], "enc_noises": enc_noises}, write_pit_wav=False)
    assert wav.shape[0] == T * hop - 1
    cs, ce, cso, ceo = plan[1]
    with torch.no_grad():
        o_src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        o = O.synth_inference(sd, hp, d["ppg"][:, cs:ce], d["vec"][:, cs:ce], d["pit"][:, cs:ce], d["spk"],
                              torch.tensor([ce - cs]), o_src[:, :, cs * hop:ce * hop], enc_noises[1])
    ref = o[0, 0, cso:ceo].numpy()
    got = wav[2500 * hop:]
    assert got.shape == ref.shape
    err = float(np.abs(got - ref).max())
    print(f"30 s clip, chunk 2: err {err:.2e}")
    assert err <= E.WAVE_TOL


def test_streaming_decoder_tiles_with_the_matrix_core_half_steps(ops):
    """The time-tiled generator with the narrow stages' matrix-core half-steps in the path.  fp32, B = 1 (the fp32 matrix-core form at 20
    channels, chosen by the batch size alone): tiles reproduce the untiled generator bit for bit.  Mixed policy (configs[4] as bench.py
    runs it): the eligibility of a launch for the 16-bit kernels follows its size, so tiles differ from the whole chunk by roundings of
    the mode -- inside the mode's error class, measured here (round 4: 3.4e-4)."""
    hp = C.base_hp()
    print("fp32 B = 1, tiled vs default path:", E.check_streaming_decoder(ops, "cuda", hp, T=700, tiles=(256, 97), B=1))
    d = E.check_streaming_decoder(ops, "cuda", hp, T=700, tiles=(256, 97), B=1, precision="mixed")
    print("mixed: largest difference between a tiling and the untiled chunk %.2e" % d)
    assert d <= 1e-3


def test_streaming_decoder_is_bit_identical_and_matches_oracle_on_30s_chunk(ops):
    """configs[4]: the time-tiled ("streaming") generator inside the reference chunks of a 30 s clip.  Tiles of 256 / 500 / 1000
    frames reproduce the untiled generator bit for bit at base.yaml widths (B = 2, T = 1300), and the second chunk of the 30 s
    clip run through svc_infer WITH tiling matches the oracle like the untiled path does."""
    from svcmi import DummyRetrieval, svc_infer
    hp = C.base_hp()
    print("tiled vs default path:", E.check_streaming_decoder(ops, "cuda", hp, T=1300, tiles=(256, 500, 1000, 17), B=2))      # 17: tiles of 49-81 frames, below every row-count threshold of the tile heuristics
    m, sd = E.make_model(hp, ops, "cuda")
    T, hop = 3000, 320
    d = I.synth_clip(T=T, hp=hp, seed=31, B=1)
    plan = O.chunk_schedule(T, hop)
    gen = torch.Generator().manual_seed(3)
    enc_noises = [torch.randn(1, hp.vits.inter_channels, ce - cs, generator=gen) for (cs, ce, _, _) in plan]
    m.stream_frames = 512
    wav = svc_infer(m, DummyRetrieval(), d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0], hp, "cuda",
                    noise={"rand_ini": d["rand_ini"], "src_noise": d["src_noise"], "enc_noises": enc_noises}, write_pit_wav=False)
    m.stream_frames = None
    cs, ce, cso, ceo = plan[1]
    with torch.no_grad():
        o_src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        o = O.synth_inference(sd, hp, d["ppg"][:, cs:ce], d["vec"][:, cs:ce], d["pit"][:, cs:ce], d["spk"],
                              torch.tensor([ce - cs]), o_src[:, :, cs * hop:ce * hop], enc_noises[1])
    err = float(np.abs(wav[2500 * hop:] - o[0, 0, cso:ceo].numpy()).max())
    print(f"30 s clip, streaming decoder (512-frame tiles), chunk 2 vs oracle: err {err:.2e}")
    assert wav.shape[0] == T * hop - 1 and err <= E.WAVE_TOL


def test_run_to_run_bit_equality(ops):
    """No atomics in any reduction: identical launches give identical bits (cheap race detector)."""
    hp = C.tiny_hp()
    m, _ = E.make_model(hp, ops, "cuda")
    d = I.synth_clip(T=64, hp=hp, seed=9, B=2)
    outs = []
    for _ in range(3):
        src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
        outs.append(m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"]).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_clips_in_flight_are_bit_identical_to_single_stream(ops):
    E.check_clips_in_flight(ops, "cuda")


def test_clip_lanes_serving_object_matches_eager_conversions(ops):
    E.check_clip_lanes(ops, "cuda")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("precision", ["f16", "mixed"])
def test_clip_lanes_capture_packs_the_16bit_weight_images(ops, precision):
    """bench.py --precision f16 | bf16x3 | mixed on configs[1]: the capture is the first run of the models in that mode."""
    E.check_clip_lanes(ops, "cuda", lanes=2, requests=3, precision=precision, capture_first=True)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("precision", ["f16", "mixed", "bf16x3"])
def test_four_lanes_in_a_reduced_precision_mode_equal_their_eager_runs(ops, precision):
    """Round 6: with the library of round 5 every one of these differed by ~1e-2 (profiles/r06ze_lanes_soak_before.log) -- SnakeAlias / vector
    half-step waves of one clip beside the 16x16x32 matrix-core instructions of another; fp32 was never affected."""
    E.check_clip_lanes(ops, "cuda", lanes=4, requests=12, precision=precision)


def test_svc_infer_chunks_in_flight_are_bit_identical(ops):
    E.check_chunk_streams(ops, "cuda")


def test_whisper_10s_against_oracle(ops):
    """Whisper-24L at the 10 s shape: mel [1,80,1000] -> [1,500,1280] vs the oracle."""
    from svcmi.whisper.inference import load_model
    ck = W.make_whisper_state(C.WHISPER_LARGE_V2)
    wm = load_model(ck, "cuda", ops=ops)
    g = torch.Generator().manual_seed(11)
    mel = (torch.randn(1, 80, 1000, generator=g) * 0.5).clamp(-1, 1.5)
    nz = torch.randn(1, 80, 1000, generator=g)
    out = wm.encoder(mel, nz, 0.1)
    with torch.no_grad():
        ref = O.audio_encoder(ck["model_state_dict"], mel + 0.1 * nz, 20, 24)
    err = E.maxerr(out, ref)
    print(f"whisper 10 s: err {err:.2e} (|ppg|max {float(ref.abs().max()):.2f})")
    assert out.shape == (1, 500, 1280) and err <= 1e-3


def test_config0_bundled_wav_and_speaker_through_the_engine(ops):
    """BASELINE.json configs[0]: the reference's own 035.wav + singer0001.npy, wav in -> wav out, vs the reference's outputs."""
    print(E.check_config0_wav_to_wav(ops, "cuda"))


def test_outlier_stress_weights_against_oracle(ops):
    """Outlier-stress weight sets (massive residual channels, LayerNorm gains up to 30, saturated GELU, SnakeBeta frequencies
    up to e^2.5 on x50 channels) at full size: Whisper-large-v2 dims on a 10 s window and the base.yaml synthesizer on a
    3 s clip, vs the oracle."""
    print("whisper stress:", E.check_whisper_stress(ops, "cuda", C.WHISPER_LARGE_V2, n=1000))
    hp = C.base_hp()
    m, sd = E.make_model(hp, ops, "cuda", stress=True)
    d = I.synth_clip(T=300, hp=hp, seed=12, B=1)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    wav = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"])
    with torch.no_grad():
        o_src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        o_wav = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], o_src, d["enc_noise"])
    err = E.maxerr(wav, o_wav)
    fl = E.stress_floor(sd, hp, d, d["lengths"], o_src, o_wav, wav)
    print(f"synth stress: waveform err {err:.2e} (vs fp64 oracle {fl['wave_vs_fp64']:.2e}; fp32 oracle vs fp64 {fl['oracle_fp32_vs_fp64']:.2e}), "
          f"rms {float(o_wav.pow(2).mean().sqrt()):.3f}")
    assert err <= E.WAVE_TOL and fl["wave_vs_fp64"] <= E.WAVE_TOL


def test_pred_ppg_two_windows_against_oracle(ops):
    """Row a2 (whisper/inference.py:32-62): a 33.2 s clip = two full 15 s windows (n = 1500 mel frames, Tw = 750, all kept; the engine
    runs them as one batch) + a ragged remainder window (odd mel length, kept = samples // 320 < Tw) through ``pred_ppg_from_mel``
    with explicit noise, against ``oracle.pred_ppg_from_mel`` (window by window, as the reference does): values, window seams and
    total length; and the batched windows against the same windows run solo."""
    from svcmi.whisper.inference import load_model, pred_ppg_from_mel, window_plan
    ck = W.make_whisper_state(C.WHISPER_LARGE_V2)
    wm = load_model(ck, "cuda", ops=ops)
    n_samples = 2 * 15 * 16000 + 51733
    plan = window_plan(n_samples)
    assert [(e - s_) for (s_, e, _) in plan] == [240000, 240000, 51733] and [k for (_, _, k) in plan] == [750, 750, 161]
    g = torch.Generator().manual_seed(23)
    mels = [(torch.randn(80, (e - s_) // 160, generator=g) * 0.5).clamp(-1, 1.5) for (s_, e, _) in plan]      # audio.py:87: n // 160 frames
    noises = [torch.randn(m_.shape, generator=g) for m_ in mels]
    keep = [k for (_, _, k) in plan]
    assert mels[2].shape[1] == 323 and (mels[2].shape[1] + 1) // 2 == 162 > keep[2]                            # the trim really drops a frame
    got = pred_ppg_from_mel(wm, mels, keep, mel_noises=[z.cuda() for z in noises])
    solo = pred_ppg_from_mel(wm, mels, keep, mel_noises=[z.cuda() for z in noises], max_batch=1)
    with torch.no_grad():
        ref = O.pred_ppg_from_mel(ck["model_state_dict"], C.WHISPER_LARGE_V2, mels, noises, keep)
    assert got.shape == ref.shape == (1500 + 161, 1280)
    scale = float(ref.abs().max())
    errs = [E.maxerr(got[a:b], ref[a:b]) for a, b in ((0, 750), (750, 1500), (1500, 1661))]
    err_solo = E.maxerr(solo, ref)
    print(f"pred_ppg 3 windows: err windows 1 + 2 (Tw=750, one batch) {errs[0]:.2e} {errs[1]:.2e}, window 3 (Tw=162, kept 161) {errs[2]:.2e}, "
          f"solo windows {err_solo:.2e}, |ppg|max {scale:.2f}")
    assert max(errs + [err_solo]) <= 1e-4 * max(1.0, scale)
    assert torch.equal(got[1500:], solo[1500:])                 # the remainder window is the same launch either way


def test_cli_main_wav_to_wav(ops, tmp_path, monkeypatch):
    """The reference's CLI flow (svc_inference.py:137-203) in one process: wav -> PPG / vec / F0 files -> svc_out.wav,
    with seeded checkpoints in the reference's formats; checks the file formats and that the result equals running the
    stages by hand on the same intermediate files."""
    import json
    import yaml
    from scipy.io import wavfile
    from oracle import audio_oracle as A
    from svcmi import svc_inference as SI
    from svcmi.pitch import load_csv_pitch
    monkeypatch.chdir(tmp_path)
    hp = C.tiny_hp()
    audio = (A.synth_audio(16000 * 2, 8) * 0.5).numpy()
    wavfile.write("in.wav", 16000, (audio * 32767).astype(np.int16))
    torch.save({"model_g": W.make_vits_state(hp, seed=1234)}, "svc.pth")
    torch.save(W.make_whisper_state({**C.WHISPER_TINY_TEST, "n_audio_state": hp.vits.ppg_dim, "n_audio_head": 4}), "whisper.pt")
    hub = dict(C.HUBERT_TINY_TEST, proj=hp.vits.vec_dim)
    torch.save(W.make_hubert_state(hub), "hubert.pt")
    torch.save(W.make_crepe_state("tiny"), "crepe.pth")
    np.save("spk.npy", I.synth_spk(hp.vits.spk_dim, seed=7).numpy())
    with open("cfg.yaml", "w") as f:
        yaml.safe_dump(json.loads(json.dumps(hp)), f)
    args = SI.build_parser().parse_args(["--config", "cfg.yaml", "--model", "svc.pth", "--wave", "in.wav", "--spk", "spk.npy",
                                         "--whisper", "whisper.pt", "--hubert", "hubert.pt", "--crepe", "crepe.pth", "--shift", "2"])
    torch.manual_seed(0)
    out = SI.main(args)
    ppg, vec, pit = np.load("svc_tmp.ppg.npy"), np.load("svc_tmp.vec.npy"), load_csv_pitch("svc_tmp.pit.csv")
    assert ppg.dtype == np.float32 and ppg.shape == (100, hp.vits.ppg_dim)          # 2 s -> 100 frames @50 fps
    assert vec.dtype == np.float32 and vec.shape == (100, hp.vits.vec_dim)          # (32000 + 80 - 400) // 320 + 1
    assert len(pit) == 202 and all(isinstance(v, int) for v in pit)                 # 2 * (1 + 32000 // 320)
    T = min(len(pit), 2 * vec.shape[0], 2 * ppg.shape[0])
    assert out.dtype == np.float32 and out.shape == (T * hp.data.hop_length - 1,) and np.isfinite(out).all()
    sr, written = wavfile.read("svc_out.wav")
    assert sr == hp.data.sampling_rate and np.array_equal(written, out)
    assert wavfile.read("svc_out_pit.wav")[1].dtype == np.int16


def test_batch_folder_driver_single_rank(ops, tmp_path, monkeypatch):
    """svc_inference_batch.py:15-52 as one process with resident models: every .wav of a folder -> _svc_out/<file>, equal to
    converting the file through the single-file CLI path with the same seeds."""
    import json
    import yaml
    from scipy.io import wavfile
    from oracle import audio_oracle as A
    from svcmi import svc_inference_batch as SB
    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    hp = C.tiny_hp()
    os.makedirs("waves")
    for i, secs in enumerate((1.0, 1.5)):
        audio = (A.synth_audio(int(16000 * secs), 8 + i) * 0.5).numpy()
        wavfile.write(f"waves/u{i}.wav", 16000, (audio * 32767).astype(np.int16))
    torch.save({"model_g": W.make_vits_state(hp, seed=1234)}, "svc.pth")
    torch.save(W.make_whisper_state({**C.WHISPER_TINY_TEST, "n_audio_state": hp.vits.ppg_dim, "n_audio_head": 4}), "whisper.pt")
    torch.save(W.make_hubert_state(dict(C.HUBERT_TINY_TEST, proj=hp.vits.vec_dim)), "hubert.pt")
    torch.save(W.make_crepe_state("tiny"), "crepe.pth")
    np.save("spk.npy", I.synth_spk(hp.vits.spk_dim, seed=7).numpy())
    with open("cfg.yaml", "w") as f:
        yaml.safe_dump(json.loads(json.dumps(hp)), f)
    args = SB.build_parser().parse_args(["--config", "cfg.yaml", "--model", "svc.pth", "--wave", "waves", "--spk", "spk.npy",
                                         "--whisper", "whisper.pt", "--hubert", "hubert.pt", "--crepe", "crepe.pth"])
    mine = SB.run_batch(args)
    assert mine == ["u0.wav", "u1.wav"]
    for f, secs in (("u0.wav", 1.0), ("u1.wav", 1.5)):
        sr, x = wavfile.read(os.path.join("_svc_out", f))
        assert sr == hp.data.sampling_rate and x.dtype == np.float32 and np.isfinite(x).all()
        assert abs(len(x) - int(secs * hp.data.sampling_rate)) <= 2 * hp.data.hop_length
    assert not [f for f in os.listdir("_svc_out") if f.startswith(".rank")]          # intermediates removed


def test_ungrouped_generator_paths_agree(ops):
    """The fallback structure of the C++ stage host (one launch per AMP block and step, for stages the grouped scheme does not fit)
    against the grouped launches."""
    hp = C.base_hp()
    m, _ = E.make_model(hp, ops, "cuda")
    d = I.synth_clip(T=60, hp=hp, seed=11, B=2)
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    run = lambda: m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src, noise=d["enc_noise"])
    want = run()
    assert ops.lib.svcmi_tune_set(b"amp_grouped", 0) == 0
    try:
        got = run()
    finally:
        ops.lib.svcmi_tune_set(b"amp_grouped", 1)
    torch.cuda.synchronize()
    assert E.maxerr(got, want) <= 2e-5


def test_cpp_host_without_python_matches_the_facade(ops, tmp_path):
    """The drop-in boundary for a non-Python host (VERDICT r2 missing #3): `python -m svcmi.tools pack` writes the packed synthesizer,
    examples/stage_host (C++, built by build.py, linked against libsvcmi.so) loads it with svcmi_packed_model_bind and runs
    svcmi_pitch2source_fwd + svcmi_synth_infer_fwd -- its waveform equals the Python facade's bit for bit (same kernels, same launch
    sequence: the facade makes the same two calls)."""
    import os
    import struct
    import subprocess
    import numpy as np
    from svcmi import packed
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "whisper-vits-svc_amd", "examples", "stage_host")
    assert os.path.exists(exe), "examples/stage_host is built by whisper-vits-svc_amd/build.py"
    hp = C.base_hp()
    m, _ = E.make_model(hp, ops, "cuda")
    B, T = 2, 120
    d = I.synth_clip(T=T, hp=hp, seed=13, B=B)
    lens = d["lengths"].clone()
    lens[-1] = T - 7
    src = m.pitch2source(d["pit"], noise=(d["rand_ini"], d["src_noise"]))
    want = m.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src, noise=d["enc_noise"]).cpu().numpy().reshape(B, -1)
    (tmp_path / "model.svcmi").write_bytes(packed.pack_model(m._weights()))
    f32 = lambda t: t.contiguous().float().numpy().tobytes()
    with open(tmp_path / "inputs.bin", "wb") as f:
        f.write(struct.pack("<ii", B, T))
        for t in (d["ppg"], d["vec"], d["pit"], d["spk"]):
            f.write(f32(t))
        f.write(lens.to(torch.int32).numpy().tobytes())
        for t in (d["rand_ini"], d["src_noise"], d["enc_noise"]):
            f.write(f32(t))
    r = subprocess.run([exe, str(tmp_path / "model.svcmi"), str(tmp_path / "inputs.bin"), str(tmp_path / "wave.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    print(r.stdout.strip())
    got = np.fromfile(tmp_path / "wave.bin", dtype=np.float32).reshape(B, -1)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_cpp_host_runs_the_whisper_stage(ops, tmp_path):
    """The same C++ host on a packed Whisper encoder (kind 2): svcmi_whisper_encoder_fwd on mel + 0.1 * noise, two 7 s windows of the
    large-v2 architecture (4 blocks kept to keep the file small) -- the PPG equals the Python facade's bit for bit."""
    import os
    import struct
    import subprocess
    import numpy as np
    from svcmi import packed
    from svcmi.whisper.inference import load_model
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "whisper-vits-svc_amd", "examples", "stage_host")
    assert os.path.exists(exe), "examples/stage_host is built by whisper-vits-svc_amd/build.py"
    dims = dict(C.WHISPER_LARGE_V2)
    dims["n_audio_layer"] = 4
    wm = load_model(W.make_whisper_state(dims), "cuda", ops=ops)
    g = torch.Generator().manual_seed(5)
    B, n = 2, 700
    mel = (torch.randn(B, 80, n, generator=g) * 0.5).clamp(-1, 1.5)
    nz = torch.randn(B, 80, n, generator=g)
    want = wm.encoder(mel, nz, 0.1).cpu().numpy()
    (tmp_path / "whisper.svcmi").write_bytes(packed.pack_model(wm.weights))
    with open(tmp_path / "mel.bin", "wb") as f:
        f.write(struct.pack("<ii", B, n))
        f.write(mel.contiguous().numpy().tobytes())
        f.write(nz.contiguous().numpy().tobytes())
    r = subprocess.run([exe, str(tmp_path / "whisper.svcmi"), str(tmp_path / "mel.bin"), str(tmp_path / "ppg.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    print(r.stdout.strip())
    got = np.fromfile(tmp_path / "ppg.bin", dtype=np.float32).reshape(want.shape)
    assert np.array_equal(got, want)

