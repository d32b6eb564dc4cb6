"""IVF-Flat feature index (row N4): the index type the reference builds, trains, saves and searches
(feature_retrieval/index.py:97-166, svc_train_retrieval.py) -- ``faiss.index_factory(d, "IVF{n},Flat", METRIC_L2)`` with
``nprobe = 1`` -- on the GPU, without faiss.

faiss-cpu 1.7.4 is pinned by the reference's requirements.txt and is neither vendored there nor installed here, so its published
algorithm is restated and **parity is unpinned** (no faiss run to compare with):

* search (IndexIVF::search_preassigned + IVFFlatScanner, nprobe = 1): coarse quantizer ``IndexFlatL2`` over the centroids picks ONE
  cell per query (``svcmi_ivf_assign_f32`` on X * C^T scores), the cell's inverted list is scanned with exact squared distances and the
  k nearest kept (``svcmi_ivf_blend_f32``, which also applies the reference's RVC weighting, index.py:75-94);
* train (Clustering.cpp, ClusteringParameters defaults: niter 25, seed 1234, max_points_per_centroid 256): sub-sample by a seeded
  permutation, initial centroids = the first k points of a second permutation, Lloyd iterations with the quantizer's assignment,
  empty clusters re-seeded by ``split_clusters`` -- the permutations and the split draws replay ``std::mt19937`` exactly (numpy's
  legacy ``RandomState`` is the same generator with the same seeding);
* file format (impl/index_write.cpp / index_read.cpp): ``IwFl`` = IVF header + ``IxF2`` quantizer + direct map + ``ilar`` inverted
  lists (``full`` / ``sprs`` size tables), little endian -- files written here are laid out for ``faiss.read_index`` and files the
  reference wrote with ``faiss.write_index`` are read by ``read_faiss_ivf_flat``.
"""
import math
import struct
from pathlib import Path

import numpy as np
import torch

from .ops import Ops

MAX_SCORE_FLOATS = 1 << 28        # 1 GiB of fp32 scores per tile
MAX_GEMM_FLOATS = (1 << 27) - 1   # one svcmi_conv_gemm_f32 operand addresses < 2^27 floats
MAX_NEAREST = 8                   # svcmi_ivf_blend_f32 keeps 8 candidates
METRIC_L2 = 1                     # faiss::METRIC_L2
KMEANS_NITER = 25                 # ClusteringParameters defaults (Clustering.h)
KMEANS_SEED = 1234
KMEANS_MAX_POINTS_PER_CENTROID = 256
SPLIT_EPS = 1.0 / 1024.0          # Clustering.cpp EPS


# ----------------------------------------------------------------------------------------------------------- faiss random numbers
def _mt_draws(seed, count):
    """``count`` raw outputs of std::mt19937(seed) (faiss RandomGenerator, utils/random.cpp)."""
    return np.random.RandomState(int(seed) & 0xFFFFFFFF).randint(0, 1 << 32, size=count, dtype=np.uint64).astype(np.int64)


def faiss_rand_perm(n, seed):
    """utils/random.cpp rand_perm: Fisher-Yates with ``i2 = i + mt() % (n - i)``."""
    perm = np.arange(n, dtype=np.int64)
    if n < 2:
        return perm
    draws = _mt_draws(seed, n - 1)
    i2 = np.arange(n - 1, dtype=np.int64) + draws % (n - np.arange(n - 1, dtype=np.int64))
    p = perm.tolist()
    for i, j in enumerate(i2.tolist()):
        p[i], p[j] = p[j], p[i]
    return np.asarray(p, dtype=np.int64)


# ----------------------------------------------------------------------------------------------------------------- file format
def _fourcc(s):
    return struct.unpack("<I", s.encode("ascii"))[0]


class _Reader:
    def __init__(self, data):
        self.b, self.o = memoryview(data), 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def array(self, dtype, count):
        a = np.frombuffer(self.b, dtype=dtype, count=count, offset=self.o)
        self.o += a.nbytes
        return a


def _read_index_header(r):
    d, ntotal, _, _, is_trained, metric = r.take("iqqqBi")       # write_index_header: d, ntotal, 2 dummies, is_trained, metric_type
    if metric > 1:
        r.take("f")                                              # metric_arg
    return d, ntotal, bool(is_trained), metric


def read_faiss_ivf_flat(path):
    """-> dict(d, ntotal, nlist, nprobe, metric, centroids [nlist, d], lists = [(vectors [m, d], ids [m])] * nlist)."""
    r = _Reader(Path(path).read_bytes())
    h = r.take("I")
    if h != _fourcc("IwFl"):
        raise ValueError(f"{path}: not a faiss IndexIVFFlat file (fourcc {struct.pack('<I', h)!r}); the reference writes 'IwFl'")
    d, ntotal, is_trained, metric = _read_index_header(r)
    nlist, nprobe = r.take("QQ")
    qh = r.take("I")
    if qh not in (_fourcc("IxF2"), _fourcc("IxFI"), _fourcc("IxFl")):
        raise ValueError(f"{path}: coarse quantizer {struct.pack('<I', qh)!r} is not a flat index")
    qd, qn, _, _ = _read_index_header(r)
    nfloat = r.take("Q")                                         # xb vector: number of floats
    if qd != d or qn != nlist or nfloat != nlist * d:
        raise ValueError(f"{path}: quantizer shape {qn} x {qd} ({nfloat} floats) does not match nlist {nlist} x d {d}")
    centroids = r.array("<f4", nlist * d).reshape(nlist, d).copy()
    dm_type = r.take("b")                                        # direct map: type, array, (hashtable)
    dm_n = r.take("Q")
    r.array("<i8", dm_n)
    if dm_type == 2:
        r.array("<i8", 2 * r.take("Q"))
    ih = r.take("I")
    if ih != _fourcc("ilar"):
        raise ValueError(f"{path}: inverted lists {struct.pack('<I', ih)!r} are not array lists ('ilar')")
    il_nlist, code_size, list_type = r.take("QQI")
    if il_nlist != nlist or code_size != 4 * d:
        raise ValueError(f"{path}: inverted lists {il_nlist} x code {code_size} B do not match nlist {nlist}, d {d}")
    sizes = np.zeros(nlist, dtype=np.int64)
    nsz = r.take("Q")
    tab = r.array("<u8", nsz).astype(np.int64)
    if list_type == _fourcc("full"):
        sizes[:] = tab
    elif list_type == _fourcc("sprs"):
        sizes[tab[0::2]] = tab[1::2]
    else:
        raise ValueError(f"{path}: unknown list table {struct.pack('<I', list_type)!r}")
    lists = []
    for m in sizes.tolist():
        if m:
            vec = r.array("<f4", m * d).reshape(m, d)
            ids = r.array("<i8", m)
        else:
            vec, ids = np.zeros((0, d), np.float32), np.zeros(0, np.int64)
        lists.append((vec, ids))
    return dict(d=d, ntotal=ntotal, nlist=nlist, nprobe=nprobe, metric=metric, is_trained=is_trained, centroids=centroids, lists=lists)


def write_faiss_ivf_flat(path, centroids, list_vectors, list_ids, nprobe=1):
    """Inverse of ``read_faiss_ivf_flat``: ``list_vectors[i]`` [m_i, d] float32, ``list_ids[i]`` [m_i] int64."""
    centroids = np.ascontiguousarray(centroids, dtype="<f4")
    nlist, d = centroids.shape
    sizes = [len(i) for i in list_ids]
    ntotal = int(sum(sizes))
    dummy = 1 << 20

    def header(n):
        return struct.pack("<iqqqBi", d, n, dummy, dummy, 1, METRIC_L2)

    out = [struct.pack("<I", _fourcc("IwFl")), header(ntotal), struct.pack("<QQ", nlist, nprobe),
           struct.pack("<I", _fourcc("IxF2")), header(nlist), struct.pack("<Q", nlist * d), centroids.tobytes(),
           struct.pack("<bQ", 0, 0),                                                      # DirectMap::NoMap, empty array
           struct.pack("<IQQ", _fourcc("ilar"), nlist, 4 * d)]
    non0 = sum(1 for m in sizes if m)
    if non0 > nlist // 2:
        out += [struct.pack("<IQ", _fourcc("full"), nlist), np.asarray(sizes, dtype="<u8").tobytes()]
    else:
        tab = [v for i, m in enumerate(sizes) if m for v in (i, m)]
        out += [struct.pack("<IQ", _fourcc("sprs"), len(tab)), np.asarray(tab, dtype="<u8").tobytes()]
    for vec, ids in zip(list_vectors, list_ids):
        if len(ids):
            out += [np.ascontiguousarray(vec, dtype="<f4").tobytes(), np.ascontiguousarray(ids, dtype="<i8").tobytes()]
    Path(path).write_bytes(b"".join(out))


# ------------------------------------------------------------------------------------------------------------------ the index
def _tile_rows(t, ldd, d):
    return max(1, min(t, MAX_SCORE_FLOATS // ldd, MAX_GEMM_FLOATS // d))


def _assign(ops, x, centroids, cent_sq, want_dist=False):
    """Coarse quantizer over all rows of x (device tensors): int32 [t] (and the distances)."""
    t, d = x.shape
    nlist = centroids.shape[0]
    ldd = (nlist + 3) // 4 * 4
    rows = _tile_rows(t, ldd, d)
    cblock = max(64, MAX_GEMM_FLOATS // d // 64 * 64)
    dots = torch.empty(rows, ldd, dtype=torch.float32, device=x.device)
    assign = torch.empty(t, dtype=torch.int32, device=x.device)
    dist = torch.empty(t, dtype=torch.float32, device=x.device) if want_dist else None
    for s in range(0, t, rows):
        e = min(t, s + rows)
        for c0 in range(0, nlist, cblock):
            c1 = min(nlist, c0 + cblock)
            ops.conv(x[None, s:e], centroids[c0:c1], out=dots[None, :e - s, c0:c1], n_out=c1 - c0)
        r = ops.ivf_assign(x[s:e], dots[:e - s], cent_sq, want_dist=want_dist)
        if want_dist:
            assign[s:e], dist[s:e] = r
        else:
            assign[s:e] = r
    return (assign, dist) if want_dist else assign


def _group(assign, nlist):
    """Stable grouping of rows by cell: (order int32 [n] with the rows of cell 0 first, offsets int32 [nlist + 1])."""
    order = torch.sort(assign.to(torch.int64), stable=True).indices.to(torch.int32)
    counts = torch.bincount(assign.to(torch.int64), minlength=nlist)
    off = torch.zeros(nlist + 1, dtype=torch.int32, device=assign.device)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order, off, counts


def split_clusters(counts, centroids, n):
    """Clustering.cpp split_clusters: every empty cluster takes a copy of a cluster drawn with probability ~ its size, both
    perturbed by +-EPS on alternating coordinates.  counts: float64 numpy [k] (edited in place), centroids: device [k, d]."""
    k, d = centroids.shape
    empty = np.flatnonzero(counts == 0)
    if len(empty) == 0:
        return 0
    state = np.random.RandomState(1234)                         # RandomGenerator rng(1234), rand_float = mt() / float(mt.max())
    sign = torch.ones(d, dtype=torch.float32, device=centroids.device)
    sign[1::2] = -1.0
    for ci in empty.tolist():
        cj = 0
        while True:
            p = np.float32((counts[cj] - 1.0) / float(n - k))
            r = np.float32(state.randint(0, 1 << 32, dtype=np.uint64)) / np.float32(4294967295.0)
            if r < p:
                break
            cj = (cj + 1) % k
        src = centroids[cj].clone()
        centroids[ci] = src * (1.0 + SPLIT_EPS * sign)
        centroids[cj] = src * (1.0 - SPLIT_EPS * sign)
        counts[ci] = counts[cj] / 2
        counts[cj] -= counts[ci]
    return len(empty)


def train_kmeans(x, k, ops, niter=KMEANS_NITER, seed=KMEANS_SEED, max_points_per_centroid=KMEANS_MAX_POINTS_PER_CENTROID):
    """faiss Clustering::train on device rows x [n, d] -> centroids [k, d] (device)."""
    n, d = x.shape
    if n < k:
        raise ValueError(f"Number of training points ({n}) should be at least as large as number of clusters ({k})")
    if n > k * max_points_per_centroid:                        # subsample_training_set
        keep = faiss_rand_perm(n, seed)[:k * max_points_per_centroid]
        x = x[torch.from_numpy(keep).to(x.device)].contiguous()
        n = x.shape[0]
    if n == k:
        return x.clone()
    init = faiss_rand_perm(n, seed + 1)[:k]
    centroids = x[torch.from_numpy(init).to(x.device)].contiguous()
    for _ in range(niter):
        assign = _assign(ops, x, centroids, ops.row_sqnorm(centroids))
        order, off, counts = _group(assign, k)
        ops.segment_mean(x, order, off, centroids)
        split_clusters(counts.cpu().numpy().astype(np.float64), centroids, n)
    return centroids


class IvfFlatFeatureIndex:
    """``FaissRVCRetrievableFeatureIndex`` over ``IndexIVFFlat`` (index.py:29-94, 145-151): centroids, the stored vectors grouped by
    cell, their ids; ``retriv`` blends every frame with the k nearest vectors of its ONE probed cell."""

    def __init__(self, centroids, list_off, bank, ids, ratio=0.5, n_nearest_vectors=1, device="cuda", ops=None):
        if 1 > n_nearest_vectors:
            raise ValueError("n-retrieval-vectors must be gte 1")
        if n_nearest_vectors > MAX_NEAREST:
            raise ValueError(f"n-retrieval-vectors must be lte {MAX_NEAREST} (svcmi_ivf_blend_f32 keeps {MAX_NEAREST} candidates per frame)")
        if not 0 <= ratio <= 1:
            raise ValueError(f"{ratio=} must be in rage (0, 1)")
        self.ops = ops if ops is not None else Ops()
        f = lambda a, dt: torch.as_tensor(a, dtype=dt).to(device).contiguous()
        self.centroids, self.bank = f(centroids, torch.float32), f(bank, torch.float32)
        self.list_off, self.ids = f(list_off, torch.int32), f(ids, torch.int64)
        if self.centroids.shape[1] % 4 or self.bank.shape[1] != self.centroids.shape[1]:
            raise ValueError(f"centroids {tuple(self.centroids.shape)} / bank {tuple(self.bank.shape)}: d must match and be a multiple of 4")
        self.cent_sq = self.ops.row_sqnorm(self.centroids)
        self._ratio, self._n_nearest = float(ratio), int(n_nearest_vectors)

    nprobe = 1
    metric_type = METRIC_L2

    @property
    def ntotal(self):
        return self.bank.shape[0]

    @property
    def nlist(self):
        return self.centroids.shape[0]

    # ---- construction
    @classmethod
    def from_faiss(cls, filepath, ratio=0.5, n_nearest_vectors=1, device="cuda", ops=None):
        """``faiss.read_index`` for the reference's IVF-Flat files (index.py:163-166)."""
        f = read_faiss_ivf_flat(filepath)
        if f["metric"] != METRIC_L2:
            raise ValueError(f"index metric type index.metric_type={f['metric']} is unsupported self.supported_distance={METRIC_L2}")
        sizes = [len(i) for _, i in f["lists"]]
        off = np.zeros(f["nlist"] + 1, dtype=np.int32)
        off[1:] = np.cumsum(sizes)
        d = f["d"]
        bank = np.concatenate([v for v, _ in f["lists"]], 0) if f["ntotal"] else np.zeros((0, d), np.float32)
        ids = np.concatenate([i for _, i in f["lists"]], 0) if f["ntotal"] else np.zeros(0, np.int64)
        return cls(f["centroids"], off, bank, ids, ratio, n_nearest_vectors, device, ops)

    @classmethod
    def train(cls, feature_matrix, ratio=0.5, n_nearest_vectors=1, device="cuda", ops=None, n_ivf=None, batch_size=8192):
        """``FaissIVFFlatTrainableFeatureIndexBuilder.build`` + ``add_with_train`` (index.py:119-151): nlist =
        min(int(16 * sqrt(n)), n // 39), k-means on the features, then every feature is added to the cell of its nearest centroid
        (ids = row numbers; rows keep their order inside a cell, as faiss appends them)."""
        ops = ops if ops is not None else Ops()
        x = torch.as_tensor(feature_matrix, dtype=torch.float32).to(device).contiguous()
        n = x.shape[0]
        if n_ivf is None:
            n_ivf = min(int(16 * np.sqrt(n)), n // 39)
        if n_ivf < 1:
            raise ValueError(f"{n} feature vectors are too few for an IVF index (index.py:146 needs >= 39)")
        centroids = train_kmeans(x, n_ivf, ops)
        assign = _assign(ops, x, centroids, ops.row_sqnorm(centroids))
        order, off, _ = _group(assign, n_ivf)
        return cls(centroids, off, x[order.long()], order.long(), ratio, n_nearest_vectors, device, ops)

    def save(self, filepath, rewrite=False):
        """index.py:26-29 ``faiss.write_index``."""
        filepath = Path(filepath)
        if filepath.exists() and not rewrite:
            raise FileExistsError(f"index already exists by path {filepath}")
        off = self.list_off.cpu().numpy()
        bank, ids = self.bank.cpu().numpy(), self.ids.cpu().numpy()
        write_faiss_ivf_flat(filepath, self.centroids.cpu().numpy(), [bank[a:b] for a, b in zip(off[:-1], off[1:])],
                             [ids[a:b] for a, b in zip(off[:-1], off[1:])], nprobe=1)

    # ---- search
    def _run(self, features, k, ratio, want_neighbours):
        x = torch.as_tensor(features, dtype=torch.float32).to(self.bank.device).contiguous()
        if x.dim() != 2 or x.shape[1] != self.bank.shape[1]:
            raise ValueError(f"feature dim {tuple(x.shape)} != index dim {self.bank.shape[1]}")
        assign = _assign(self.ops, x, self.centroids, self.cent_sq)
        return self.ops.ivf_blend(x, assign, self.list_off, self.bank, k, ratio, want_neighbours=want_neighbours)

    def search_and_reconstruct(self, features, k):
        """faiss ``IndexIVF.search_and_reconstruct``: (squared distances [t, k], labels [t, k] with -1 padding, vectors [t, k, d])."""
        _, idx, dist = self._run(features, k, 0.0, True)
        rows = idx.long().clamp_min(0)
        labels = torch.where(idx >= 0, self.ids[rows], torch.full_like(rows, -1))
        vec = self.bank[rows]
        vec[idx < 0] = float("nan")
        return dist.cpu().numpy(), labels.cpu().numpy(), vec.cpu().numpy()

    def retriv(self, features):
        """index.py:57-62: [t, d] (numpy like the reference, or a tensor on any device) -> same kind, blended."""
        as_numpy = isinstance(features, np.ndarray)
        src_device = None if as_numpy else torch.as_tensor(features).device
        out = self._run(features, self._n_nearest, self._ratio, False)
        return out.cpu().numpy() if as_numpy else out.to(src_device)


def ivf_list_count(num_vectors):
    """index.py:146."""
    return min(int(16 * math.sqrt(num_vectors)), num_vectors // 39)
