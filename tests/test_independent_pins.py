"""Pins for the three third-party steps the reference delegates to packages that are NOT installed here (librosa, faiss):
the oracle's restatements are checked against INDEPENDENT implementations / ground truth available in this image.

  * librosa.filters.mel  (whisper/audio.py:53-65)       <- transformers.audio_utils.mel_filter_bank (slaney scale + slaney norm)
  * librosa.sequence.viterbi (crepe/decode.py:55)       <- exhaustive enumeration of every path of small trellises (ground truth)
  * faiss search_and_reconstruct (feature_retrieval/index.py:57-94) <- scikit-learn's brute-force NearestNeighbors

None of these is the package the reference calls, so the rows stay "restated"; what the tests establish is that the restated
ALGORITHM is the published one (two independent implementations agree to rounding) and that the dynamic programme is exact.
"""
import itertools

import numpy as np
import pytest


def test_slaney_filterbank_against_the_transformers_implementation():
    """an independent implementation of librosa.filters.mel(sr=16000, n_fft=400, n_mels=80) -- HTK off, Slaney
    area normalisation -- agrees with the oracle's matrix to 1e-9 (float32 storage of the oracle: 6e-10 measured)."""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    from oracle import audio_oracle as A
    theirs = audio_utils.mel_filter_bank(201, 80, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T      # [80, 201]
    ours = A.slaney_mel_filterbank()
    assert theirs.shape == ours.shape == (80, 201)
    assert float(np.abs(theirs - ours).max()) <= 2e-9
    # the same for another geometry (the restatement is not fitted to one shape)
    theirs = audio_utils.mel_filter_bank(257, 40, 0.0, 11025.0, 22050, norm="slaney", mel_scale="slaney").T
    assert float(np.abs(theirs - A.slaney_mel_filterbank(22050, 512, 40)).max()) <= 2e-9


def _brute_force_best(prob, trans):
    """Every state sequence of the trellis: max of log p0 + sum log obs + sum log trans (uniform initial distribution)."""
    S, T = prob.shape
    tiny = np.finfo(prob.dtype).tiny
    lp, lt = np.log(prob + tiny), np.log(trans + tiny)
    best, arg = -np.inf, []
    for path in itertools.product(range(S), repeat=T):
        s = np.log(1.0 / S + tiny) + lp[path[0], 0]
        for t in range(1, T):
            s += lt[path[t - 1], path[t]] + lp[path[t], t]
        if s > best + 1e-12:
            best, arg = s, [path]
        elif abs(s - best) <= 1e-12:
            arg.append(path)
    return best, arg


def _score(path, prob, trans):
    tiny = np.finfo(prob.dtype).tiny
    lp, lt = np.log(prob + tiny), np.log(trans + tiny)
    s = np.log(1.0 / prob.shape[0] + tiny) + lp[path[0], 0]
    for t in range(1, len(path)):
        s += lt[path[t - 1], path[t]] + lp[path[t], t]
    return s


@pytest.mark.parametrize("S,T", [(2, 1), (2, 6), (3, 5), (4, 6), (5, 6)])
def test_viterbi_restatement_is_the_exact_dynamic_programme(S, T):
    """the decoded path of ``oracle.crepe_oracle.viterbi_path`` (and of the product's host form) is THE optimum over
    all S^T paths on random trellises -- column-stochastic observations, row-stochastic transitions incl. CREPE's triangular band."""
    from oracle import crepe_oracle as CO
    rng = np.random.default_rng(100 * S + T)
    for trial in range(40):
        prob = rng.random((S, T)) + 1e-3
        prob /= prob.sum(0, keepdims=True)
        if trial % 2:
            xx, yy = np.meshgrid(range(S), range(S))
            trans = np.maximum(2 - abs(xx - yy), 0).astype(np.float64)          # banded like decode.py:58-61 (zeros outside the band)
        else:
            trans = rng.random((S, S)) + 1e-3
        trans /= trans.sum(1, keepdims=True)
        best, arg = _brute_force_best(prob, trans)
        got = tuple(int(v) for v in CO.viterbi_path(prob, trans))
        assert abs(_score(got, prob, trans) - best) <= 1e-9, (S, T, trial)
        if len(arg) == 1:
            assert got == arg[0]


def test_viterbi_ties_resolve_to_an_optimal_path_with_the_lowest_final_state():
    """Exact ties (equal columns, uniform transitions): every path is optimal; the restatement returns argmax's first index at
    every step -- the convention of numpy's argmax that librosa's implementation is built on."""
    from oracle import crepe_oracle as CO
    prob = np.full((3, 4), 1.0 / 3)
    trans = np.full((3, 3), 1.0 / 3)
    assert CO.viterbi_path(prob, trans).tolist() == [0, 0, 0, 0]
    prob = np.array([[0.4, 0.2, 0.4], [0.4, 0.6, 0.4], [0.2, 0.2, 0.2]])          # states 0 and 1 tie at the ends
    best, arg = _brute_force_best(prob, trans)
    got = tuple(CO.viterbi_path(prob, trans).tolist())
    assert got in arg and got[-1] == min(p[-1] for p in arg)


def test_product_host_viterbi_equals_the_oracle_on_small_trellises():
    """svcmi.pitch's host-side Viterbi (the CPU form the GPU kernel is tested against) on the same brute-forced cases."""
    PI = pytest.importorskip("svcmi.pitch.inference")
    fn = PI.viterbi_path
    from oracle import crepe_oracle as CO
    rng = np.random.default_rng(7)
    for _ in range(20):
        prob = rng.random((4, 6)) + 1e-3
        prob /= prob.sum(0, keepdims=True)
        trans = rng.random((4, 4)) + 1e-3
        trans /= trans.sum(1, keepdims=True)
        assert list(fn(prob, trans)) == CO.viterbi_path(prob, trans).tolist()


def test_exhaustive_knn_restatement_against_scikit_learn():
    """faiss's exact search (what nprobe = nlist computes) against scikit-learn's brute-force kNN: same neighbours, same squared
    distances, ascending order."""
    nn = pytest.importorskip("sklearn.neighbors")
    from oracle import retrieval_oracle as R
    rng = np.random.default_rng(3)
    bank = rng.standard_normal((700, 24)).astype(np.float32)
    x = rng.standard_normal((60, 24)).astype(np.float32)
    for k in (1, 4, 8):
        scores, ids = R.knn_search(x, bank, k)
        dist, idx = nn.NearestNeighbors(n_neighbors=k, algorithm="brute", metric="sqeuclidean").fit(bank.astype(np.float64)).kneighbors(x.astype(np.float64))
        assert np.array_equal(ids, idx)
        assert np.allclose(scores, dist, rtol=1e-5, atol=1e-6)
        # the blend on top of it (index.py:57-62, 75-94), written out with scikit-learn's neighbours
        w = 1.0 / np.square(dist)
        w /= w.sum(1, keepdims=True)
        want = 0.5 * x + 0.5 * (bank[idx] * w[..., None]).sum(1)
        assert np.allclose(R.retriv(x, bank, 0.5, k), want, rtol=1e-5, atol=1e-6)


def test_ivf_nprobe1_restatement_against_scikit_learn_per_cell():
    """IVF-Flat with nprobe = 1 = exact kNN inside the cell of the nearest centroid: both stages against scikit-learn."""
    nn = pytest.importorskip("sklearn.neighbors")
    from oracle import retrieval_oracle as R
    rng = np.random.default_rng(4)
    d, nlist, k = 16, 9, 4
    cent = rng.standard_normal((nlist, d)).astype(np.float32) * 2
    bank = (cent[rng.integers(0, nlist, 900)] + 0.5 * rng.standard_normal((900, d))).astype(np.float32)
    assign = nn.NearestNeighbors(n_neighbors=1, algorithm="brute").fit(cent.astype(np.float64)).kneighbors(bank.astype(np.float64))[1][:, 0]
    lists = [(bank[assign == c], np.nonzero(assign == c)[0]) for c in range(nlist)]
    x = (cent[rng.integers(0, nlist, 50)] + 0.5 * rng.standard_normal((50, d))).astype(np.float32)
    cell = nn.NearestNeighbors(n_neighbors=1, algorithm="brute").fit(cent.astype(np.float64)).kneighbors(x.astype(np.float64))[1][:, 0]
    got_cell, _ = R.coarse_assign(x, cent)
    assert np.array_equal(got_cell, cell)
    dist, labels, recons = R.ivf_search(x, cent, lists, k)
    for i in range(x.shape[0]):
        vec, ids = lists[int(cell[i])]
        m = min(k, len(ids))
        dd, ii = nn.NearestNeighbors(n_neighbors=m, algorithm="brute", metric="sqeuclidean").fit(vec.astype(np.float64)).kneighbors(x[i:i + 1].astype(np.float64))
        assert np.array_equal(labels[i, :m], ids[ii[0]])
        assert np.allclose(dist[i, :m], dd[0], rtol=1e-5, atol=1e-6)
        assert np.array_equal(recons[i, :m], vec[ii[0]])


@pytest.mark.parametrize("rate", [32000, 44100, 48000, 22050])
def test_loader_resampler_against_analytically_sampled_tones(tmp_path, rate):
    """``load_audio`` on a non-16 kHz file (librosa.load's job, whisper/audio.py:24-26; librosa uses soxr, we use scipy's polyphase
    filter: a documented divergence, so no parity claim) must at least BE a band-limited resampler: tones below 6 kHz written at
    ``rate`` come back at 16 kHz as the same tones sampled at 16 kHz (interior samples, 2e-3 of full scale), int16 PCM in."""
    from scipy.io import wavfile
    from svcmi.whisper.audio import load_audio
    dur = 0.5
    t_in = np.arange(int(rate * dur)) / rate
    freqs, amps = (220.0, 1730.0, 5900.0), (0.4, 0.25, 0.1)
    x = sum(a * np.sin(2 * np.pi * f * t_in) for f, a in zip(freqs, amps))
    path = tmp_path / f"tones_{rate}.wav"
    wavfile.write(path, rate, np.round(x * 32767.0).astype(np.int16))
    y = load_audio(str(path))
    assert y.dtype == np.float32 and abs(len(y) - int(round(16000 * dur))) <= 1
    t_out = np.arange(len(y)) / 16000.0
    want = sum(a * np.sin(2 * np.pi * f * t_out) for f, a in zip(freqs, amps))
    edge = 400                                                   # filter transients at both ends
    assert float(np.abs(y[edge:-edge] - want[edge:-edge]).max()) <= 2e-3
    # and a 16 kHz file is passed through exactly (int16 / 32768)
    wavfile.write(tmp_path / "same.wav", 16000, np.round(want * 32767.0).astype(np.int16))
    z = load_audio(str(tmp_path / "same.wav"))
    assert np.array_equal(z, np.round(want * 32767.0).astype(np.int16).astype(np.float32) / 32768.0)
