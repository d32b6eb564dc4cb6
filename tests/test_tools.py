"""Row N4 host tooling: checkpoint export / merge / speaker mix (formats the engine loads), the retrieval host logic,
and -- in the build container -- the same functions of the reference run side by side."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from svcmi import SynthesizerInfer, load_svc_model, tools
from workload import config as C
from workload import weights as W


def _train_ckpt(tmp_path, seed=0):
    hp = C.tiny_hp()
    sd = W.make_vits_state(hp, seed=seed)
    full = dict(sd)
    full["enc_q.pre.weight"] = torch.randn(4, 4, 1)             # posterior encoder: training only
    dropped = next(k for k in sd if k.startswith("dec.") and k.endswith("bias"))
    del full[dropped]                                            # a key the checkpoint lacks keeps its init value
    path = tmp_path / f"train_{seed}.pt"
    torch.save({"model_g": full, "model_d": {"d.weight": torch.ones(2)}, "optim_g": {}, "optim_d": {}, "step": 7, "epoch": 1,
                "hp_str": "x"}, path)
    return hp, sd, path, dropped


def test_export_checkpoint_format(tmp_path):
    hp, sd, path, dropped = _train_ckpt(tmp_path)
    out = tmp_path / "sovits5.0.pth"
    tools.export_checkpoint(hp, str(path), str(out))
    saved = torch.load(out, map_location="cpu")
    assert list(saved.keys()) == ["model_g"]
    model = SynthesizerInfer(hp.data.filter_length // 2 + 1, hp.data.segment_size // hp.data.hop_length, hp)
    assert list(saved["model_g"].keys()) == list(model.state_dict().keys())        # no enc_q, reference key order
    for k, v in saved["model_g"].items():
        if k != dropped:
            assert torch.equal(v, sd[k]), k
    assert torch.equal(saved["model_g"][dropped], model.state_dict()[dropped])
    load_svc_model(str(out), model)                                                 # and the engine's loader takes it


def test_save_pretrain(tmp_path):
    _, _, path, _ = _train_ckpt(tmp_path)
    out = tmp_path / "pre.pth"
    tools.save_pretrain(str(path), str(out))
    assert sorted(torch.load(out, map_location="cpu").keys()) == ["model_d", "model_g"]


def test_merge_and_average(tmp_path):
    hp = C.tiny_hp()
    a, b = W.make_vits_state(hp, seed=1), W.make_vits_state(hp, seed=2)
    m = tools.merge_model(a, b, 0.3)
    k = "flow.flows.0.enc.in_layers.0.weight_v" if "flow.flows.0.enc.in_layers.0.weight_v" in a else next(iter(a))
    assert torch.equal(m[k], 0.3 * a[k] + (1 - 0.3) * b[k])
    avg = tools.average_model([a, b])
    assert torch.allclose(avg[k], (a[k] + b[k]) / 2)
    with pytest.raises(AssertionError):
        tools.merge_model(a, b, 1.0)
    tools.save_model_g(m, tmp_path / "m.pth")
    assert torch.equal(tools.load_model_g(str(tmp_path / "m.pth"))[k], m[k])


def test_mix_speakers(tmp_path):
    rng = np.random.default_rng(0)
    p = []
    for i in range(3):
        p.append(str(tmp_path / f"s{i}.npy"))
        np.save(p[-1], rng.standard_normal(256).astype(np.float32))
    conf = {p[0]: 0, p[1]: 0.5, p[2]: 0.5}
    eva = tools.mix_speakers(conf, str(tmp_path / "eva.spk.npy"))
    assert eva.dtype == np.float64 and eva.shape == (256,)
    want = np.zeros(256)
    for path, v in conf.items():      # svc_eva.py:15-19: float32 product, float64 running sum
        want = want + np.load(path) * v
    assert np.array_equal(np.load(tmp_path / "eva.spk.npy"), want)


@pytest.mark.needs_reference
def test_merge_matches_reference(tmp_path):
    spec = importlib.util.spec_from_file_location("ref_svc_merge", "/root/reference/svc_merge.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    hp = C.tiny_hp()
    a, b = W.make_vits_state(hp, seed=3), W.make_vits_state(hp, seed=4)
    want, got = ref.merge_model(a, b, 0.7), tools.merge_model(a, b, 0.7)
    assert list(want.keys()) == list(got.keys()) and all(torch.equal(want[k], got[k]) for k in want)
    want, got = ref.average_model([a, b, a]), tools.average_model([a, b, a])
    assert all(torch.equal(want[k], got[k]) for k in want)


@pytest.mark.needs_reference
def test_export_matches_reference(tmp_path):
    """The reference's svc_export.load_model / save_model on ITS SynthesizerInfer vs export_checkpoint."""
    from oracle import ref_import as R
    hp, sd, path, dropped = _train_ckpt(tmp_path, seed=5)
    ref_model = R.ref_synthesizer(hp, sd)                       # reference nn.Module with the seeded weights
    saved = torch.load(path, map_location="cpu")["model_g"]
    want = {k: (saved[k] if k in saved else v) for k, v in ref_model.state_dict().items()}     # svc_export.py:18-23
    got = tools.export_checkpoint(hp, str(path), str(tmp_path / "o.pth"))
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if k != dropped:
            assert torch.equal(got[k], want[k]), k


# ---------------------------------------------------------------------------------------------- retrieval host logic
def test_retrieval_oracle_weighting():
    """k = 1 returns the stored vector itself at ratio 1; a duplicated neighbour set gives equal weights."""
    from oracle import retrieval_oracle as RO
    rng = np.random.default_rng(1)
    bank = rng.standard_normal((50, 8)).astype(np.float32)
    x = rng.standard_normal((6, 8)).astype(np.float32)
    out = RO.retriv(x, bank, 1.0, 1)
    ids = np.argmin(((x[:, None] - bank[None]) ** 2).sum(-1), axis=1)
    assert np.allclose(out, bank[ids], atol=1e-6)
    assert np.allclose(RO.retriv(x, bank, 0.0, 3), x)
    w = RO.weight_nearest_vectors(np.stack([bank[:2]] * 6), np.full((6, 2), 4.0, np.float32))
    assert np.allclose(w, bank[:2].mean(0), atol=1e-6)


def test_retrieval_cli_paths(tmp_path, monkeypatch):
    from svcmi import feature_retrieval as FR
    from svcmi.svc_inference import DummyRetrieval, build_parser
    assert FR.get_speaker_name_from_path("configs/singers/singer0001.npy") == "singer0001"
    args = build_parser().parse_args(["--config", "c", "--model", "m", "--wave", "w", "--spk", "s.spk.npy"])
    assert isinstance(FR.create_retrival(args), DummyRetrieval)
    assert args.retrieval_ratio == 0.5 and args.n_retrieval_vectors == 3 and args.retrieval_index_prefix == ""
    with pytest.raises(ValueError):
        FR.KnnFeatureIndex(np.zeros((4, 8), np.float32), 0.5, 0, ops=object())
    with pytest.raises(ValueError):
        FR.KnnFeatureIndex(np.zeros((4, 8), np.float32), 1.5, 1, ops=object())
    d = tmp_path / "feat" / "spk"
    os.makedirs(d)
    np.save(d / "a.npy", np.ones((3, 8), np.float32))
    np.save(d / "b.npy", np.zeros((2, 8), np.float32))
    bank = FR.build_index_bank(tmp_path / "feat", tmp_path / "bank.npy")
    assert bank.shape == (5, 8) and np.array_equal(np.load(tmp_path / "bank.npy"), bank)


def test_retrieval_oracle_search_matches_sklearn_brute_force():
    """faiss is absent, so the search half of the retrieval oracle is cross-checked against an independent exhaustive
    implementation (scikit-learn, squared-euclidean metric -- what faiss METRIC_L2 reports)."""
    from sklearn.neighbors import NearestNeighbors
    from oracle import retrieval_oracle as RO
    rng = np.random.default_rng(3)
    bank = rng.standard_normal((500, 24)).astype(np.float32)
    x = rng.standard_normal((40, 24)).astype(np.float32)
    nn = NearestNeighbors(n_neighbors=3, algorithm="brute", metric="sqeuclidean").fit(bank.astype(np.float64))
    dist, ids = nn.kneighbors(x.astype(np.float64))
    scores, got = RO.knn_search(x, bank, 3)
    assert np.array_equal(got, ids)
    assert np.allclose(scores, dist, rtol=1e-5)


def test_mt19937_replay_matches_std_mt19937():
    """faiss's RandomGenerator is std::mt19937; the C++ standard fixes its 10000th output for the default seed 5489 at 4123659995
    (first output 3499211612) -- numpy's legacy RandomState reproduces the stream, which is what rand_perm / split_clusters replay."""
    from oracle import retrieval_oracle as RO
    from svcmi import ivf_index as IV
    draws = IV._mt_draws(5489, 10000)
    assert draws[0] == 3499211612 and draws[9999] == 4123659995
    assert np.array_equal(draws, RO._mt19937(5489, 10000))
    p = IV.faiss_rand_perm(1000, 1234)
    assert np.array_equal(np.sort(p), np.arange(1000)) and np.array_equal(p, RO.rand_perm(1000, 1234))
    assert p[0] == 0 + int(IV._mt_draws(1234, 1)[0]) % 1000          # first swap partner, rand_perm's definition


def test_faiss_ivf_flat_file_layout(tmp_path):
    """Byte-for-byte layout of faiss 1.7.4 ``write_index(IndexIVFFlat)`` (impl/index_write.cpp: write_ivf_header, write_index_header,
    write_direct_map, write_InvertedLists) built by hand here, against the writer; then the reader on both a 'full' and a 'sprs' file."""
    import struct
    from svcmi import ivf_index as IV
    d, nlist = 4, 3
    cent = np.arange(nlist * d, dtype=np.float32).reshape(nlist, d)
    vecs = [np.full((2, d), 1.5, np.float32), np.zeros((0, d), np.float32), np.full((1, d), -2.0, np.float32)]
    ids = [np.array([7, 9], np.int64), np.zeros(0, np.int64), np.array([4], np.int64)]
    hdr = lambda n: struct.pack("<i", d) + struct.pack("<q", n) + struct.pack("<qq", 1 << 20, 1 << 20) + b"\x01" + struct.pack("<i", 1)
    want = (b"IwFl" + hdr(3) + struct.pack("<QQ", nlist, 1)
            + b"IxF2" + hdr(nlist) + struct.pack("<Q", nlist * d) + cent.tobytes()
            + b"\x00" + struct.pack("<Q", 0)
            + b"ilar" + struct.pack("<QQ", nlist, 4 * d) + b"full" + struct.pack("<Q", nlist) + struct.pack("<QQQ", 2, 0, 1)
            + vecs[0].tobytes() + ids[0].tobytes() + vecs[2].tobytes() + ids[2].tobytes())
    f = tmp_path / "a.index"
    IV.write_faiss_ivf_flat(f, cent, vecs, ids)
    assert f.read_bytes() == want
    r = IV.read_faiss_ivf_flat(f)
    assert (r["d"], r["ntotal"], r["nlist"], r["nprobe"], r["metric"], r["is_trained"]) == (d, 3, nlist, 1, 1, True)
    assert np.array_equal(r["centroids"], cent)
    for (v, i), v0, i0 in zip(r["lists"], vecs, ids):
        assert np.array_equal(v, v0) and np.array_equal(i, i0)
    # sparse size table: at most half of the lists non-empty
    g = tmp_path / "b.index"
    IV.write_faiss_ivf_flat(g, cent, [vecs[1], vecs[1], vecs[2]], [ids[1], ids[1], ids[2]])
    raw = g.read_bytes()
    assert b"sprs" + struct.pack("<Q", 2) + struct.pack("<QQ", 2, 1) in raw
    r = IV.read_faiss_ivf_flat(g)
    assert [len(i) for _, i in r["lists"]] == [0, 0, 1] and np.array_equal(r["lists"][2][1], ids[2])
    (tmp_path / "c.index").write_bytes(b"IxF2" + raw[4:])
    with pytest.raises(ValueError):
        IV.read_faiss_ivf_flat(tmp_path / "c.index")


def test_ivf_oracle_degenerates_to_exhaustive_search():
    """One cell = every vector probed: the IVF restatement must then equal the exhaustive one (and sklearn's, by the test above);
    with more cells every returned neighbour lies in the query's own cell and is the nearest there."""
    from oracle import retrieval_oracle as RO
    rng = np.random.default_rng(5)
    bank = rng.standard_normal((400, 16)).astype(np.float32)
    x = rng.standard_normal((30, 16)).astype(np.float32)
    ids = np.arange(400, dtype=np.int64)
    dist, lab, rec = RO.ivf_search(x, bank[:1] * 0, [(bank, ids)], 4)
    want_d, want_i = RO.knn_search(x, bank, 4)
    assert np.array_equal(lab, want_i) and np.allclose(dist, want_d, rtol=1e-6) and np.array_equal(rec, bank[want_i])
    assert np.allclose(RO.ivf_retriv(x, bank[:1] * 0, [(bank, ids)], 0.5, 4), RO.retriv(x, bank, 0.5, 4), atol=1e-6)
    cent, lists = RO.ivf_build(bank, n_ivf=8, niter=5)
    cell, _ = RO.coarse_assign(x, cent)
    dist, lab, _ = RO.ivf_search(x, cent, lists, 2)
    for i in range(len(x)):
        own = lists[int(cell[i])][1]
        assert set(lab[i][lab[i] >= 0]) <= set(own.tolist())
        d2 = ((x[i] - bank[own]) ** 2).sum(1)
        assert np.isclose(dist[i, 0], d2.min(), rtol=1e-5)
    assert sum(len(i) for _, i in lists) == 400
