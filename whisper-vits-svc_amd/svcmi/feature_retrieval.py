"""Feature retrieval on the GPU (row N4): the reference's optional faiss kNN blend of the content features
(feature_retrieval/index.py:57-94, retrieval.py:31-44, svc_inference.py:25-58).

The reference keeps one faiss IVF-Flat index (nprobe = 1, squared-L2 metric) per feature kind (whisper PPG, HuBERT vec) and, per
synthesis chunk, replaces each frame by ``(1 - ratio) * x + ratio * sum_q w_q * nn_q`` with ``w = (1/d2)^2`` normalised over the k
nearest stored vectors.  faiss is not part of this stack.  Two index kinds answer ``retriv``:

* ``IvfFlatFeatureIndex`` (ivf_index.py) -- the reference's own semantics: one probed cell per frame, exact scan of its list.  Reads and
  writes the reference's ``.index`` files (faiss IndexIVFFlat layout) and trains like ``svc_train_retrieval.py``;
* ``KnnFeatureIndex`` (here) -- an exact brute-force search over a ``.npy`` [n, d] bank (what IVF gives with nprobe = nlist): the stored
  vectors live in HBM as one matrix, the candidate scores are one svcmi_conv_gemm_f32 launch (X * Bank^T) and ``svcmi_knn_blend_f32``
  does selection, exact re-measurement and the blend.  The two agree whenever IVF's probed cell holds the true neighbours.
"""
import glob
import os
from pathlib import Path

import numpy as np
import torch

from .ivf_index import IvfFlatFeatureIndex
from .ops import Ops
from .svc_inference import IRetrieval

MAX_SCORE_FLOATS = 1 << 28        # 1 GiB of fp32 scores per search tile
MAX_NEAREST = 8                   # candidates svcmi_knn_blend_f32 keeps per frame
MAX_GEMM_FLOATS = (1 << 27) - 1   # one svcmi_conv_gemm_f32 operand addresses < 2^27 floats


class KnnFeatureIndex:
    """``FaissRVCRetrievableFeatureIndex`` (index.py:29-94) over a device-resident bank."""

    def __init__(self, bank, ratio, n_nearest_vectors, device="cuda", ops=None):
        if 1 > n_nearest_vectors:
            raise ValueError("n-retrieval-vectors must be gte 1")
        if n_nearest_vectors > MAX_NEAREST:
            raise ValueError(f"n-retrieval-vectors must be lte {MAX_NEAREST} (svcmi_knn_blend_f32 keeps {MAX_NEAREST} candidates per frame)")
        if not 0 <= ratio <= 1:
            raise ValueError(f"{ratio=} must be in rage (0, 1)")
        bank = torch.as_tensor(bank, dtype=torch.float32)
        if bank.dim() != 2 or bank.shape[0] < n_nearest_vectors or bank.shape[1] % 4:
            raise ValueError(f"bank must be [n >= {n_nearest_vectors}, d % 4 == 0], got {tuple(bank.shape)}")
        self.ops = ops if ops is not None else Ops()
        self.bank = bank.to(device).contiguous()
        self.bank_sq = self.ops.row_sqnorm(self.bank)
        self._ratio = float(ratio)
        self._n_nearest = int(n_nearest_vectors)
        self._bank_block = max(64, ((1 << 27) - 1) // self.bank.shape[1] // 64 * 64)

    @property
    def ntotal(self):
        return self.bank.shape[0]

    def save(self, filepath, rewrite=False):
        filepath = Path(filepath)
        if filepath.exists() and not rewrite:
            raise FileExistsError(f"index already exists by path {filepath}")
        with open(filepath, "wb") as f:
            np.save(f, self.bank.cpu().numpy(), allow_pickle=False)

    def retriv(self, features):
        """[t, d] (numpy like the reference, or a tensor on any device) -> same kind, blended."""
        as_numpy = isinstance(features, np.ndarray)
        x = torch.as_tensor(features, dtype=torch.float32)
        src_device = x.device
        x = x.to(self.bank.device).contiguous()
        t, d = x.shape
        if d != self.bank.shape[1]:
            raise ValueError(f"feature dim {d} != index dim {self.bank.shape[1]}")
        out = torch.empty_like(x)
        n = self.bank.shape[0]
        ldd = (n + 3) // 4 * 4
        rows = max(1, min(t, MAX_SCORE_FLOATS // ldd, MAX_GEMM_FLOATS // d))      # score tile and the GEMM's x operand both bounded
        dots = torch.empty(rows, ldd, dtype=torch.float32, device=x.device)
        for s in range(0, t, rows):
            e = min(t, s + rows)
            for b0 in range(0, n, self._bank_block):      # one GEMM launch addresses < 2^27 weight elements
                b1 = min(n, b0 + self._bank_block)
                self.ops.conv(x[None, s:e], self.bank[b0:b1], out=dots[None, :e - s, b0:b1], n_out=b1 - b0)
            self.ops.knn_blend(x[s:e], self.bank, dots[:e - s], self.bank_sq, self._n_nearest, self._ratio, out=out[s:e])
        return out.cpu().numpy() if as_numpy else out.to(src_device)


def load_retrieve_index(filepath, ratio, n_nearest_vectors, device="cuda", ops=None):
    """index.py:163-166: a faiss IVF-Flat ``.index`` file (the reference's format) -> ``IvfFlatFeatureIndex`` (nprobe = 1 semantics);
    a ``.npy`` [n, d] feature bank -> ``KnnFeatureIndex`` (exact search)."""
    filepath = str(filepath)
    if not os.path.exists(filepath):
        raise FileNotFoundError(f"retrieval index {filepath} not found (python -m svcmi.svc_train_retrieval builds .index files, "
                                f"build_index_bank .npy banks)")
    if filepath.endswith(".npy"):
        return KnnFeatureIndex(np.load(filepath), ratio, n_nearest_vectors, device=device, ops=ops)
    return IvfFlatFeatureIndex.from_faiss(filepath, ratio, n_nearest_vectors, device=device, ops=ops)


def build_index_bank(feature_dir, out_path=None):
    """Stack every ``*.npy`` [T, d] feature file under ``feature_dir`` (data_svc/whisper/<spk> or data_svc/hubert/<spk>,
    feature_retrieval/train.py) into the [n, d] bank an index holds; optionally save it."""
    files = sorted(glob.glob(os.path.join(str(feature_dir), "**", "*.npy"), recursive=True))
    if not files:
        raise FileNotFoundError(f"no .npy feature files under {feature_dir}")
    bank = np.concatenate([np.load(f).astype(np.float32) for f in files], axis=0)
    if out_path is not None:
        with open(out_path, "wb") as f:
            np.save(f, bank, allow_pickle=False)
    return bank


class KnnIndexRetrieval(IRetrieval):
    """``FaissIndexRetrieval`` (retrieval.py:31-44); chunks stay on the device."""

    on_device = True

    def __init__(self, hubert_index, whisper_index):
        self._hubert_index = hubert_index
        self._whisper_index = whisper_index

    def retriv_whisper(self, vec):
        return self._whisper_index.retriv(vec)

    def retriv_hubert(self, vec):
        return self._hubert_index.retriv(vec)


def get_speaker_name_from_path(speaker_path):
    """svc_inference.py:19-22 (str.rstrip with the suffix CHARACTERS, as the reference does)."""
    speaker_path = Path(speaker_path)
    return speaker_path.name.rstrip("".join(speaker_path.suffixes))


def create_retrival(cli_args, device="cuda"):
    """svc_inference.py:25-58: data_svc/indexes/<speaker>/<prefix>{hubert,whisper}.index unless ``--hubert-index-path /
    --whisper-index-path`` are given (a path ending in .npy selects the exact-search bank)."""
    from .svc_inference import DummyRetrieval
    if not cli_args.enable_retrieval:
        return DummyRetrieval()
    base_path = Path(".").absolute() / "data_svc" / "indexes" / get_speaker_name_from_path(cli_args.spk)
    prefix = cli_args.retrieval_index_prefix

    def default(kind):
        ivf = base_path / f"{prefix}{kind}.index"
        bank = base_path / f"{prefix}{kind}.index.npy"
        return bank if (bank.exists() and not ivf.exists()) else ivf

    hubert_path = cli_args.hubert_index_path or default("hubert")
    whisper_path = cli_args.whisper_index_path or default("whisper")
    ops = Ops()
    kw = dict(ratio=cli_args.retrieval_ratio, n_nearest_vectors=cli_args.n_retrieval_vectors, device=device, ops=ops)
    return KnnIndexRetrieval(hubert_index=load_retrieve_index(hubert_path, **kw),
                             whisper_index=load_retrieve_index(whisper_path, **kw))
