"""Generate tests/golden/*.npz by executing the REAL reference (build container only).

    python -m oracle.make_golden            # from the repo root; needs /root/reference

The reference holds no tests or golden vectors; these fixtures are the pin:
outputs of the reference's own modules (``SynthesizerInfer``, ``Whisper.encoder``, ``svc_infer``) on
seeded weights/inputs with the stochastic draws injected (oracle/ref_import.py).  The script also
asserts that the oracle restatement agrees with the reference on every fixture before writing it.
Weights are never stored: they are regenerated from their seed (oracle/weights.py); a checksum of
the regenerated tensors is stored so RNG drift is detected instead of mis-reported as a parity bug.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

from workload import config as C
from workload import inputs as I
from . import ref_import as R
from . import svc_oracle as O
from workload import weights as W

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TOL = 5e-5


def checksum(tensors):
    """Order-independent fp64 digest of a dict/list of tensors."""
    if isinstance(tensors, dict):
        tensors = [tensors[k] for k in sorted(tensors)]
    return float(sum(t.double().abs().sum().item() + 0.5 * t.double().sum().item() for t in tensors))


def _agree(name, a, b, tol=TOL):
    err = (a - b).abs().max().item()
    print(f"  oracle vs reference [{name}]: max-abs {err:.3e}")
    assert err <= tol, (name, err)


def vits_fixture(tag, hp, T, B, lengths=None, store_inputs=False, seed=1):
    print(f"[{tag}] T={T} B={B}")
    sd = W.make_vits_state(hp, seed=1234)
    ref = R.ref_synthesizer(hp, sd)
    d = I.synth_clip(T=T, hp=hp, seed=seed, B=B)
    lens = d["lengths"] if lengths is None else torch.tensor(lengths, dtype=torch.long)
    from vits.utils import f0_to_coarse
    with torch.no_grad():
        with R.injected_noise([d["src_noise"]], [d["rand_ini"]]):
            src = ref.pitch2source(d["pit"])
        with R.injected_noise([d["enc_noise"]]):
            z_p, m_p, logs_p, mask, _ = ref.enc_p(d["ppg"], lens, d["vec"], f0=f0_to_coarse(d["pit"]))
        z, _ = ref.flow(z_p, mask, g=d["spk"], reverse=True)
        with R.injected_noise([d["enc_noise"]]):
            wav = ref.inference(d["ppg"], d["vec"], d["pit"], d["spk"], lens, src)
        pitwav = ref.source2wav(src[:1])
        o_src = O.pitch2source(sd, hp, d["pit"], d["rand_ini"], d["src_noise"])
        o_wav, parts = O.synth_inference(sd, hp, d["ppg"], d["vec"], d["pit"], d["spk"], lens, o_src,
                                         d["enc_noise"], return_parts=True)
    _agree("source", o_src, src, 1e-6)
    _agree("z_p", parts["z_p"], z_p)
    _agree("z", parts["z"], z)
    _agree("wave", o_wav, wav)
    out = {
        "T": T, "B": B, "seed": seed, "lengths": lens.numpy(),
        "weights_checksum": checksum(sd), "inputs_checksum": checksum({k: v for k, v in d.items()}),
        "source": src.numpy(), "z_p": z_p.numpy(), "z": z.numpy(), "wave": wav.numpy(),
        "pitwav": pitwav, "f0_coarse": f0_to_coarse(d["pit"]).numpy(),
    }
    if store_inputs:
        for k, v in d.items():
            out["in_" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)


def whisper_fixture(tag, dims, n, store_all=True):
    print(f"[{tag}] n_mel_frames={n}")
    ck = W.make_whisper_state(dims)
    ref = R.ref_whisper_encoder(ck)
    g = torch.Generator().manual_seed(5)
    mel = (torch.randn(1, 80, n, generator=g) * 0.5).clamp(-1.0, 1.5)
    nz = torch.randn(1, 80, n, generator=g)
    with torch.no_grad():
        # whisper/inference.py:46-47: mel + randn_like(mel)*0.1 then encoder
        with R.injected_noise([nz[0]]):
            m = mel[0] + torch.randn_like(mel[0]) * 0.1
        ppg = ref.encoder(m.unsqueeze(0))
        o = O.audio_encoder(ck["model_state_dict"], (mel + 0.1 * nz), dims["n_audio_head"], O.whisper_kept_layers(dims))
    _agree("ppg", o, ppg)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), n=n, mel=mel.numpy(), mel_noise=nz.numpy(),
                        ppg=ppg.numpy(), weights_checksum=checksum(ck["model_state_dict"]))


def _import_ref_driver():
    """Import the reference's svc_inference.py with its unavailable third-party imports stubbed
    (omegaconf, faiss, the crepe-based pitch package); the code under test -- svc_infer, lines
    77-134 -- touches none of them."""
    R._prepare()
    for name, attrs in (("omegaconf", {"OmegaConf": object}), ("faiss", {"IndexIVF": object, "Index": object}),
                        ("pitch", {"load_csv_pitch": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    import svc_inference
    return svc_inference


def svc_infer_fixture(tag, hp, T):
    """Two-chunk run through the reference's own host loop (chunk 2500 + halo 10, svc_inference.py:94-131)."""
    print(f"[{tag}] T={T}")
    drv = _import_ref_driver()
    from feature_retrieval import DummyRetrieval
    sd = W.make_vits_state(hp, seed=1234)
    ref = R.ref_synthesizer(hp, sd)
    d = I.synth_clip(T=T, hp=hp, seed=2, B=1)
    plan = O.chunk_schedule(T, hp.data.hop_length)
    g = torch.Generator().manual_seed(77)
    enc_noises = [torch.randn(1, hp.vits.inter_channels, ce - cs, generator=g) for (cs, ce, _, _) in plan]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            with R.injected_noise([d["src_noise"]] + enc_noises, [d["rand_ini"]]):
                wav = drv.svc_infer(ref, DummyRetrieval(), d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0], hp, "cpu")
        finally:
            os.chdir(cwd)
    with torch.no_grad():
        o_wav, o_pit = O.svc_infer(sd, hp, d["spk"][0], d["pit"][0], d["ppg"][0], d["vec"][0],
                                   d["rand_ini"], d["src_noise"], enc_noises)
    _agree("svc_infer wave", torch.from_numpy(o_wav), torch.from_numpy(wav))
    L = T * hp.data.hop_length
    assert wav.shape[0] == L - 1, wav.shape            # cut_e_out = -1 drops the last sample (:112,129)
    seam = C.CHUNK_FRAMES * hp.data.hop_length
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), T=T, length=wav.shape[0], plan=np.array(plan),
                        wave_sub=wav[::97].copy(), wave_seam=wav[seam - 3000:seam + 3000].copy(),
                        wave_tail=wav[-2000:].copy(),
                        weights_checksum=checksum(sd), inputs_checksum=checksum(d))


def logmel_fixture(tag, n, seed):
    """The reference's own log_mel_spectrogram (whisper/audio.py:68-100) on a seeded signal.  Its filterbank comes
    from librosa (absent): the restated Slaney matrix is injected in its place, so the fixture pins everything
    EXCEPT that matrix."""
    print(f"[{tag}] n={n}")
    from . import audio_oracle as A
    R._prepare()
    import whisper.audio as ref_audio
    fb = A.slaney_mel_filterbank()
    ref_audio.librosa_mel_fn = lambda **kw: fb
    ref_audio.mel_filters.cache_clear()
    x = A.synth_audio(n, seed)
    with torch.no_grad():
        ref = ref_audio.log_mel_spectrogram(x.numpy())
        o = A.log_mel_spectrogram(x)
    _agree("logmel", o, ref, 1e-6)
    assert tuple(ref.shape) == (80, n // 160)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), n=n, seed=seed, logmel=ref.numpy(),
                        filterbank_checksum=checksum([torch.from_numpy(fb)]), audio_checksum=checksum([x]))


def hubert_fixture(tag, n, seed):
    """The reference HubertSoft.units (hubert/hubert_model.py:64-72) at its own dimensions on a seeded waveform."""
    print(f"[{tag}] n={n}")
    from . import hubert_oracle as H
    R._prepare()
    from hubert.hubert_model import HubertSoft
    sd = W.make_hubert_state()
    ref = HubertSoft()
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    g = torch.Generator().manual_seed(seed)
    wav = torch.randn(1, 1, n, generator=g) * 0.3
    with torch.no_grad():
        u = ref.units(wav)
        o = H.units(sd, wav, 12)
    _agree("hubert units", o, u)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), n=n, seed=seed, units=u.numpy(), weights_checksum=checksum(sd))


def crepe_fixture(tag, capacity, n, seed):
    """The reference's vendored crepe package (crepe/core.py preprocess + infer + postprocess) on seeded weights.
    resampy is stubbed (16 kHz input needs no resampling); librosa.sequence.viterbi -- absent -- is replaced by the
    oracle's restatement, so the Viterbi routine itself is the one unpinned piece."""
    print(f"[{tag}] n={n}")
    import importlib.machinery
    from . import crepe_oracle as CO
    R._prepare()
    for name in ("resampy", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)
                sys.modules[name] = m
    import librosa
    seq = types.ModuleType("librosa.sequence")
    seq.viterbi = lambda p, t: CO.viterbi_path(p, t)
    librosa.sequence = seq
    sys.modules["librosa.sequence"] = seq
    import crepe
    import scipy.stats
    sd = W.make_crepe_state(capacity)
    model = crepe.Crepe(capacity)
    model.load_state_dict(sd, strict=True)
    model.eval()
    crepe.infer.model, crepe.infer.capacity = model, capacity
    gen = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    audio = (0.4 * torch.sin(2 * np.pi * (180 + 60 * torch.sin(2 * np.pi * 1.5 * t)) * t) + 0.02 * torch.randn(n, generator=gen)).float()
    with torch.no_grad():
        frames = next(crepe.preprocess(audio[None].clone(), 16000, 320, None, "cpu", True))
        prob = crepe.infer(frames, capacity)
        o_prob = CO.network(sd, CO.preprocess(audio[None], 320))
    _agree("crepe posteriors", o_prob, prob, 1e-6)
    orig = scipy.stats.triang.rvs
    scipy.stats.triang.rvs = lambda c, loc, scale, size: np.zeros(tuple(size))       # dither off for the fixture
    try:
        f0a = crepe.postprocess(prob.reshape(1, -1, 360).transpose(1, 2).clone(), 50., 1000., crepe.decode.argmax)[0]
        f0v = crepe.postprocess(prob.reshape(1, -1, 360).transpose(1, 2).clone(), 50., 1000., crepe.decode.viterbi)[0]
    finally:
        scipy.stats.triang.rvs = orig
    z = np.zeros(prob.shape[0])
    _agree("argmax f0", CO.decode(o_prob, 50., 1000., "argmax", z), f0a, 1e-3)
    _agree("viterbi f0", CO.decode(o_prob, 50., 1000., "viterbi", z), f0v, 1e-3)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), n=n, seed=seed, prob=prob.numpy(), f0_argmax=f0a.numpy(),
                        f0_viterbi=f0v.numpy(), weights_checksum=checksum({k: v.float() for k, v in sd.items()}))


def config0_fixture(tag="config0_035"):
    """BASELINE.json configs[0]: the reference's own bundled sample (configs/singers_sample/22-wave-girl/035.wav, 63 902
    samples @16 kHz = 3.99 s) and speaker (configs/singers/singer0001.npy) through the reference's OWN functions in the order
    svc_inference.py:137-203 runs them: whisper/inference.py pred_ppg (large-v2 dims, 24 blocks) -> hubert/inference.py
    pred_vec -> np.repeat x2 -> svc_infer.  Weights are the seeded random-init ones (no checkpoints offline); librosa is
    absent, so `librosa.load` is replaced by the int16 / 32768 conversion it performs on a 16 kHz file and the Slaney
    filterbank by the restated one (as in logmel_fixture); the F0 track is a synthetic integer-Hz contour passed the way
    `--pit` passes a CSV (CREPE's Viterbi decoding needs librosa.sequence).  The wav and the speaker file are copied next
    to the fixture so the GPU box can read them."""
    import shutil
    import scipy.io.wavfile
    from . import audio_oracle as A
    from . import hubert_oracle as H
    print(f"[{tag}]")
    R._prepare()
    wav_src = os.path.join(R.REF, "configs", "singers_sample", "22-wave-girl", "035.wav")
    spk_src = os.path.join(R.REF, "configs", "singers", "singer0001.npy")
    shutil.copyfile(wav_src, os.path.join(OUT, "035.wav"))
    shutil.copyfile(spk_src, os.path.join(OUT, "singer0001.npy"))
    os.chmod(os.path.join(OUT, "035.wav"), 0o644)
    os.chmod(os.path.join(OUT, "singer0001.npy"), 0o644)
    sr, pcm = scipy.io.wavfile.read(wav_src)
    assert sr == 16000 and pcm.dtype == np.int16 and pcm.shape == (63902,)
    audio = pcm.astype(np.float32) / 32768.0
    import librosa
    librosa.load = lambda f, sr=16000: (audio.copy(), sr)
    hp = C.base_hp()
    # ---- PPG: the reference's pred_ppg
    import whisper.audio as ref_audio
    import whisper.inference as ref_winf
    fb = A.slaney_mel_filterbank()
    ref_audio.librosa_mel_fn = lambda **kw: fb
    ref_audio.mel_filters.cache_clear()
    ck = W.make_whisper_state(C.WHISPER_LARGE_V2)
    wm = R.ref_whisper_encoder(ck)
    g = torch.Generator().manual_seed(35)
    n_mel = audio.shape[0] // 160
    mel_noise = torch.randn(80, n_mel, generator=g)
    with tempfile.TemporaryDirectory() as tmp:
        with torch.no_grad(), R.injected_noise([mel_noise]):
            ref_winf.pred_ppg(wm, wav_src, os.path.join(tmp, "ppg.npy"), "cpu")
        ppg = np.load(os.path.join(tmp, "ppg.npy"))
        # ---- vec: the reference's pred_vec
        import hubert.inference as ref_hinf
        from hubert.hubert_model import HubertSoft
        hsd = W.make_hubert_state()
        hub = HubertSoft()
        hub.load_state_dict(hsd, strict=True)
        hub.eval()
        ref_hinf.pred_vec(hub, wav_src, os.path.join(tmp, "vec.npy"), "cpu")
        vec = np.load(os.path.join(tmp, "vec.npy"))
    assert ppg.shape == (audio.shape[0] // 320, 1280) and vec.shape[1] == 256, (ppg.shape, vec.shape)
    with torch.no_grad():
        mel = ref_audio.log_mel_spectrogram(audio)
        o_mel = A.log_mel_spectrogram(torch.from_numpy(audio))
        o_ppg = O.pred_ppg_from_mel(ck["model_state_dict"], C.WHISPER_LARGE_V2, [o_mel], [mel_noise], [audio.shape[0] // 320])
        o_vec = H.units(hsd, torch.from_numpy(audio)[None, None], 12)[0]
    _agree("logmel", o_mel, mel, 1e-6)
    _agree("ppg", o_ppg, torch.from_numpy(ppg))
    _agree("vec", o_vec, torch.from_numpy(vec))
    # ---- svc_inference.py:172-203, then the reference's own svc_infer
    drv = _import_ref_driver()
    from feature_retrieval import DummyRetrieval
    spk = torch.FloatTensor(np.load(spk_src))
    ppg2 = torch.FloatTensor(np.repeat(ppg, 2, 0))
    vec2 = torch.FloatTensor(np.repeat(vec, 2, 0))
    T = min(ppg2.shape[0], vec2.shape[0])
    pit = torch.FloatTensor([int(v) for v in I.synth_f0(T, seed=35).tolist()])
    sd = W.make_vits_state(hp, seed=1234)
    ref = R.ref_synthesizer(hp, sd)
    L = T * hp.data.hop_length
    rand_ini = torch.rand(1, 11, generator=g)
    src_noise = torch.randn(1, L, 11, generator=g)
    plan = O.chunk_schedule(T, hp.data.hop_length)
    assert len(plan) == 1
    enc_noises = [torch.randn(1, hp.vits.inter_channels, T, generator=g)]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            with R.injected_noise([src_noise] + enc_noises, [rand_ini]):
                wav = drv.svc_infer(ref, DummyRetrieval(), spk, pit, ppg2, vec2, hp, "cpu")
        finally:
            os.chdir(cwd)
    with torch.no_grad():
        o_wav, _ = O.svc_infer(sd, hp, spk, pit, ppg2, vec2, rand_ini, src_noise, enc_noises)
    _agree("svc_infer wave", torch.from_numpy(o_wav), torch.from_numpy(wav))
    assert wav.shape[0] == L - 1
    print(f"  T={T} frames, {wav.shape[0]} samples, rms {float(np.sqrt((wav ** 2).mean())):.3f}")
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), T=T, noise_seed=35, logmel=mel.numpy(), ppg=ppg, vec=vec, pit=pit.numpy(),
                        wave=wav, whisper_checksum=checksum(ck["model_state_dict"]), hubert_checksum=checksum(hsd),
                        vits_checksum=checksum(sd), audio_checksum=checksum([torch.from_numpy(audio)]))


def main():
    assert R.available(), "needs /root/reference"
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    vits_fixture("vits_tiny_ragged", C.tiny_hp(), T=24, B=2, lengths=[24, 17], store_inputs=True)
    vits_fixture("vits_base_T60", C.base_hp(), T=60, B=1)
    whisper_fixture("whisper_tiny", C.WHISPER_TINY_TEST, n=301)
    whisper_fixture("whisper_large_v2_n200", C.WHISPER_LARGE_V2, n=200)
    svc_infer_fixture("svc_infer_tiny_2chunks", C.tiny_hp(), T=2600)
    logmel_fixture("logmel_2p5s", n=40000, seed=21)
    hubert_fixture("hubert_soft_1s", n=16000, seed=3)
    crepe_fixture("crepe_full_1s", "full", n=16000, seed=5)
    config0_fixture()
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "config0":
        assert R.available(), "needs /root/reference"
        config0_fixture()
    else:
        main()
