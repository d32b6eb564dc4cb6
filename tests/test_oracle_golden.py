"""The oracle restatement against the golden vectors produced by the real reference
(oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from workload import config as C
from workload import inputs as I
from oracle import svc_oracle as O
from workload import weights as W
from oracle.make_golden import checksum

TOL = 5e-5   # fp32 CPU vs fp32 CPU (different BLAS blocking across hosts); waveform in [-1,1]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_filter_taps_match_reference_constants():
    # taps quoted in SURVEY.md a28 (computed from vits_decoder/alias/filter.py:28-57)
    f = W.kaiser_sinc_filter().view(-1).numpy()
    want = [0.00202896, 0.00938947, -0.02554346, -0.05765738, 0.12857258, 0.44320980]
    assert np.allclose(f[:6], want, atol=2e-8) and np.allclose(f[6:], want[::-1], atol=2e-8)


@pytest.mark.parametrize("tag,hp", [("vits_tiny_ragged", C.tiny_hp()), ("vits_base_T60", C.base_hp())])
def test_vits_path_against_reference_golden(golden_dir, tag, hp):
    g = _load(golden_dir, tag)
    sd = W.make_vits_state(hp, seed=1234)
    assert checksum(sd) == pytest.approx(float(g["weights_checksum"]), rel=1e-9), "weight RNG drifted"
    d = I.synth_clip(T=int(g["T"]), hp=hp, seed=int(g["seed"]), B=int(g["B"]))
    assert checksum(d) == pytest.approx(float(g["inputs_checksum"]), rel=1e-9), "input RNG drifted"
    lens = _t(g["lengths"])
    with torch.no_grad():
        src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        wav, parts = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], lens, src,
                                       d["enc_noise"], return_parts=True)
    assert (O.f0_to_coarse(d["pit"]).numpy() == g["f0_coarse"]).all()
    assert (src - _t(g["source"])).abs().max() <= 1e-6
    assert (parts["z_p"] - _t(g["z_p"])).abs().max() <= TOL
    assert (parts["z"] - _t(g["z"])).abs().max() <= TOL
    assert (wav - _t(g["wave"])).abs().max() <= TOL
    assert np.abs(O.source2wav(src[:1]).astype(np.int32) - g["pitwav"].astype(np.int32)).max() <= 1


def test_tiny_fixture_carries_its_inputs(golden_dir):
    g = _load(golden_dir, "vits_tiny_ragged")
    d = I.synth_clip(T=int(g["T"]), hp=C.tiny_hp(), seed=int(g["seed"]), B=int(g["B"]))
    for k, v in d.items():
        assert np.array_equal(v.numpy(), g["in_" + k]), k


@pytest.mark.parametrize("tag,dims", [("whisper_tiny", C.WHISPER_TINY_TEST)])
def test_whisper_encoder_against_reference_golden(golden_dir, tag, dims):
    g = _load(golden_dir, tag)
    ck = W.make_whisper_state(dims)
    assert checksum(ck["model_state_dict"]) == pytest.approx(float(g["weights_checksum"]), rel=1e-9)
    with torch.no_grad():
        out = O.audio_encoder(ck["model_state_dict"], _t(g["mel"]) + 0.1 * _t(g["mel_noise"]),
                              dims["n_audio_head"], O.whisper_kept_layers(dims))
    assert (out - _t(g["ppg"])).abs().max() <= TOL


def test_chunk_schedule_matches_reference_plan(golden_dir):
    g = _load(golden_dir, "svc_infer_tiny_2chunks")
    plan = O.chunk_schedule(int(g["T"]), 320)
    assert np.array_equal(np.array(plan), g["plan"])
    # 30 s clip -> [0,2510) keep [0,-3200) ; [2490,3000) keep [3200,-1)
    assert O.chunk_schedule(3000, 320) == [(0, 2510, 0, -3200), (2490, 3000, 3200, -1)]
    assert O.chunk_schedule(1000, 320) == [(0, 1000, 0, -1)]


def test_svc_infer_two_chunks_against_reference_golden(golden_dir):
    g = _load(golden_dir, "svc_infer_tiny_2chunks")
    hp = C.tiny_hp()
    T = int(g["T"])
    sd = W.make_vits_state(hp, seed=1234)
    d = I.synth_clip(T=T, hp=hp, seed=2, B=1)
    gen = torch.Generator().manual_seed(77)
    enc_noises = [torch.randn(1, hp.vits.inter_channels, ce - cs, generator=gen)
                  for (cs, ce, _, _) in O.chunk_schedule(T, 320)]
    with torch.no_grad():
        wav, _ = O.svc_infer(sd, hp, d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0],
                             d["rand_ini"], d["src_noise"], enc_noises)
    assert wav.shape[0] == int(g["length"]) == T * 320 - 1
    seam = C.CHUNK_FRAMES * 320
    assert np.abs(wav[::97] - g["wave_sub"]).max() <= TOL
    assert np.abs(wav[seam - 3000:seam + 3000] - g["wave_seam"]).max() <= TOL
    assert np.abs(wav[-2000:] - g["wave_tail"]).max() <= TOL


@pytest.mark.needs_reference
def test_oracle_equals_live_reference_on_fresh_seed():
    """Not just the stored vectors: a different seed/shape through the imported reference."""
    from oracle import ref_import as R
    hp = C.tiny_hp()
    sd = W.make_vits_state(hp, seed=99)
    ref = R.ref_synthesizer(hp, sd)
    d = I.synth_clip(T=31, hp=hp, seed=5, B=1)
    with torch.no_grad():
        with R.injected_noise([d["src_noise"]], [d["rand_ini"]]):
            src = ref.pitch2source(d["pit"])
        with R.injected_noise([d["enc_noise"]]):
            wav = ref.inference(d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"], src)
        o = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], d["lengths"],
                              O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"]), d["enc_noise"])
    assert (o - wav).abs().max() <= TOL


def test_logmel_oracle_matches_reference_golden(golden_dir):
    from oracle import audio_oracle as A
    g = _load(golden_dir, "logmel_2p5s")
    x = A.synth_audio(int(g["n"]), int(g["seed"]))
    assert abs(checksum([x]) - float(g["audio_checksum"])) <= 1e-6 * abs(float(g["audio_checksum"]))
    assert abs(checksum([torch.from_numpy(A.slaney_mel_filterbank())]) - float(g["filterbank_checksum"])) <= 1e-9
    got = A.log_mel_spectrogram(x)
    assert float((got - _t(g["logmel"])).abs().max()) <= 1e-5


def test_slaney_filterbank_properties():
    """The one matrix that cannot be pinned against librosa here: structural checks of the published construction."""
    from oracle import audio_oracle as A
    fb = A.slaney_mel_filterbank()
    assert fb.shape == (80, 201) and (fb >= 0).all()
    peaks = fb.argmax(1)
    assert (np.diff(peaks) >= 0).all() and peaks[0] >= 1 and peaks[-1] <= 199      # centres increase with the mel index
    freqs = np.linspace(0, 8000, 201)
    centres = (fb * freqs).sum(1) / fb.sum(1)
    assert abs(centres[0] - 200.0 / 3.0) < 30.0                                    # first centre = 1 mel step = 66.7 Hz
    # Slaney normalisation: each triangle has unit area in Hz
    assert np.allclose(fb.sum(1) * 40.0, 1.0, atol=0.08)


def test_hubert_oracle_matches_reference_golden(golden_dir):
    from oracle import hubert_oracle as H
    g = _load(golden_dir, "hubert_soft_1s")
    sd = W.make_hubert_state()
    assert abs(checksum(sd) - float(g["weights_checksum"])) <= 1e-6 * abs(float(g["weights_checksum"]))
    gen = torch.Generator().manual_seed(int(g["seed"]))
    wav = torch.randn(1, 1, int(g["n"]), generator=gen) * 0.3
    with torch.no_grad():
        got = H.units(sd, wav, 12)
    assert float((got - _t(g["units"])).abs().max()) <= TOL


def test_crepe_oracle_matches_reference_golden(golden_dir):
    from oracle import crepe_oracle as CO
    from tests.engine_cases import crepe_test_audio
    g = _load(golden_dir, "crepe_full_1s")
    sd = W.make_crepe_state("full")
    audio = crepe_test_audio(int(g["n"]), int(g["seed"]))
    with torch.no_grad():
        prob = CO.network(sd, CO.preprocess(audio[None], 320))
    assert float((prob - _t(g["prob"])).abs().max()) <= 1e-5
    z = np.zeros(prob.shape[0])
    assert np.allclose(CO.decode(prob, 50., 1000., "argmax", z).numpy(), g["f0_argmax"], rtol=1e-5)
    assert np.allclose(CO.decode(prob, 50., 1000., "viterbi", z).numpy(), g["f0_viterbi"], rtol=1e-5)


def test_viterbi_restatement_on_a_known_case():
    """Hand-checkable 3-state case: sticky transitions keep the path on state 0 through one ambiguous frame."""
    from oracle import crepe_oracle as CO
    prob = np.array([[0.8, 0.45, 0.8], [0.1, 0.5, 0.1], [0.1, 0.05, 0.1]])
    tr = np.array([[0.9, 0.05, 0.05], [0.05, 0.9, 0.05], [0.05, 0.05, 0.9]])
    assert CO.viterbi_path(prob, tr).tolist() == [0, 0, 0]
    assert CO.viterbi_path(prob, np.full((3, 3), 1 / 3)).tolist() == [0, 1, 0]


@pytest.mark.needs_reference
def test_crepe_mean_filter_oracle_matches_the_reference_filter():
    """the tail of compute_f0_sing (np.repeat x2 + crepe.filter.mean(5), pitch/inference.py:95-97) was only
    compared oracle-to-engine.  Here the oracle's mean_filter and the engine's numpy form run against the reference's own
    crepe/filter.py:10-57 (imported from /root/reference) on tracks with NaNs, exact zeros and lengths below the window."""
    import importlib.util
    import numpy as np
    from oracle import crepe_oracle as CO
    from svcmi.pitch import inference as PI
    spec = importlib.util.spec_from_file_location("ref_crepe_filter", "/root/reference/crepe/filter.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(5)
    for T in (1, 2, 3, 4, 5, 6, 9, 64, 1001):
        for win in (3, 5):
            x = 50.0 + 900.0 * torch.rand(1, T, generator=g)
            x[torch.rand(1, T, generator=g) < 0.2] = float("nan")
            x[torch.rand(1, T, generator=g) < 0.1] = 0.0
            x = torch.from_numpy(np.repeat(x.numpy(), 2, -1))              # pitch/inference.py:95
            want = ref.mean(x.clone(), win)
            assert torch.allclose(CO.mean_filter(x.clone(), win), want, rtol=0, atol=0, equal_nan=True)
            got = PI._mean_filter_np(x[0].numpy(), win)
            assert got.shape == (2 * T,) and np.allclose(got, want[0].numpy(), rtol=1e-6, atol=0, equal_nan=True)


def test_config0_fixture_oracle_reproduces_the_reference(golden_dir):
    """BASELINE.json configs[0] (plumbing on the CPU): the bundled 035.wav + singer0001.npy.  The stored PPG / vec / waveform are
    what the REFERENCE's pred_ppg, pred_vec and svc_infer produced (make_golden.config0_fixture); the oracle must reproduce
    the log-mel from the wav file and the waveform from the stored features with the same injected noise."""
    import scipy.io.wavfile
    from oracle import audio_oracle as A
    from tests.engine_cases import config0_noise
    g = _load(golden_dir, "config0_035")
    hp = C.base_hp()
    sr, pcm = scipy.io.wavfile.read(os.path.join(golden_dir, "035.wav"))
    assert sr == 16000 and pcm.shape == (63902,)
    audio = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
    assert checksum([audio]) == pytest.approx(float(g["audio_checksum"]), rel=1e-12)
    assert (A.log_mel_spectrogram(audio) - _t(g["logmel"])).abs().max() <= 1e-6
    T = int(g["T"])
    assert g["ppg"].shape == (63902 // 320, 1280) and 2 * g["ppg"].shape[0] == T == 398
    spk = torch.FloatTensor(np.load(os.path.join(golden_dir, "singer0001.npy")))
    assert spk.shape == (256,) and abs(float(spk.norm()) - 0.82) < 0.05
    sd = W.make_vits_state(hp, seed=1234)
    assert checksum(sd) == pytest.approx(float(g["vits_checksum"]), rel=1e-9)
    _, rand_ini, src_noise, enc_noises = config0_noise(g, T, hp)
    ppg2 = torch.FloatTensor(np.repeat(g["ppg"], 2, 0))
    vec2 = torch.FloatTensor(np.repeat(g["vec"], 2, 0))
    with torch.no_grad():
        wav, _ = O.svc_infer(sd, hp, spk, _t(g["pit"]), ppg2, vec2, rand_ini, src_noise, enc_noises)
    assert wav.shape == g["wave"].shape == (T * 320 - 1,)
    assert np.abs(wav - g["wave"]).max() <= TOL
