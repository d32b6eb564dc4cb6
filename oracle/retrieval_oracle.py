"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's feature retrieval blend (row N4).

Follows feature_retrieval/index.py:57-62 (``retriv``) and :75-94 (``_weight_nearest_vectors``).  The search itself is
faiss (faiss-cpu 1.7.4, requirements.txt:17, NOT installed here): ``search_and_reconstruct`` on a
METRIC_L2 index returns the k smallest SQUARED L2 distances in ascending order with the stored vectors.  This restatement
does the exhaustive search (IVF with nprobe = nlist); the reference's nprobe = 1 (index.py:150) approximates it.
No faiss here, so there is no reference output for this row (parity unpinned against faiss itself); the weighting lines are
numpy and are restated operation by operation, and since round 5 the search semantics (exact kNN; nearest-centroid cell + exact kNN inside
it for nprobe = 1) are cross-checked against scikit-learn's brute-force NearestNeighbors (tests/test_independent_pins.py).

``ivf_*`` / ``kmeans_faiss`` restate what the reference actually runs -- faiss-cpu 1.7.4 (requirements.txt:17) IndexIVFFlat with
nprobe = 1 (index.py:145-151) and its k-means trainer (Clustering.cpp) -- from faiss's published sources, equally unpinned.
"""
import numpy as np


def knn_search(features, bank, k):
    """Exhaustive squared-L2 search in float64 -> (scores [t, k] float32 ascending, ids [t, k])."""
    x = features.astype(np.float64)
    b = bank.astype(np.float64)
    d2 = (x * x).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * x @ b.T
    ids = np.argsort(d2, axis=1, kind="stable")[:, :k]
    exact = ((x[:, None, :] - b[ids]) ** 2).sum(-1)
    return exact.astype(np.float32), ids


def weight_nearest_vectors(nearest_vectors, scores):
    """index.py:75-94."""
    weight = np.square(1 / scores)
    weight /= weight.sum(axis=1, keepdims=True)
    weight = np.expand_dims(weight, axis=2)
    return np.sum(nearest_vectors * weight, axis=1)


def retriv(features, bank, ratio, k):
    """index.py:57-62."""
    features = features.astype(np.float32)
    scores, ids = knn_search(features, bank, k)
    nearest = bank.astype(np.float32)[ids]
    return (1 - ratio) * features + ratio * weight_nearest_vectors(nearest, scores)


# ------------------------------------------------------------------------------------- IVF-Flat, nprobe = 1 (index.py:145-151)
def coarse_assign(features, centroids):
    """IndexFlatL2 search with k = 1 as faiss's BLAS path computes it (utils/distances.cpp exhaustive_L2sqr_blas):
    |x|^2 + |c|^2 - 2 x.c in float32, negative values clamped to 0, first minimum wins."""
    x = features.astype(np.float32)
    c = centroids.astype(np.float32)
    dis = ((x * x).sum(1, dtype=np.float32)[:, None] + (c * c).sum(1, dtype=np.float32)[None, :]) - np.float32(2.0) * (x @ c.T)
    dis = np.maximum(dis, np.float32(0.0))
    return np.argmin(dis, axis=1), dis


def ivf_search(features, centroids, lists, k):
    """``search_and_reconstruct`` of an IVF-Flat index with nprobe = 1: ``lists[c] = (vectors [m, d], ids [m])``.
    -> (squared distances [t, k] ascending, +inf padded; labels [t, k], -1 padded; vectors [t, k, d], NaN padded)."""
    x = features.astype(np.float32)
    t, d = x.shape
    cell, _ = coarse_assign(x, centroids)
    dist = np.full((t, k), np.inf, np.float32)
    labels = np.full((t, k), -1, np.int64)
    recons = np.full((t, k, d), np.nan, np.float32)
    for i in range(t):
        vec, ids = lists[int(cell[i])]
        if len(ids) == 0:
            continue
        d2 = ((x[i].astype(np.float64)[None, :] - vec.astype(np.float64)) ** 2).sum(1)      # IVFFlatScanner: fvec_L2sqr per stored vector
        order = np.argsort(d2, kind="stable")[:k]
        m = len(order)
        dist[i, :m], labels[i, :m], recons[i, :m] = d2[order], ids[order], vec[order]
    return dist, labels, recons


def ivf_retriv(features, centroids, lists, ratio, k):
    """index.py:57-62 on that search.  Cells holding fewer than k vectors: the reference multiplies faiss's NaN padding by a zero
    weight and emits a NaN frame; the engine documents a different choice (use what the cell has; empty cell = frame unchanged)
    and this restatement follows the engine there."""
    features = features.astype(np.float32)
    dist, labels, recons = ivf_search(features, centroids, lists, k)
    out = features.copy()
    for i in range(features.shape[0]):
        m = int((labels[i] >= 0).sum())
        if m:
            out[i] = (1 - ratio) * features[i] + ratio * weight_nearest_vectors(recons[i:i + 1, :m], dist[i:i + 1, :m])[0]
    return out


def _mt19937(seed, count):
    return np.random.RandomState(int(seed) & 0xFFFFFFFF).randint(0, 1 << 32, size=count, dtype=np.uint32).astype(np.int64)


def rand_perm(n, seed):
    """faiss utils/random.cpp: std::mt19937(seed), ``i2 = i + mt() % (n - i)``, swap."""
    perm = list(range(n))
    for i, r in enumerate(_mt19937(seed, max(n - 1, 0)).tolist()):
        j = i + r % (n - i)
        perm[i], perm[j] = perm[j], perm[i]
    return np.asarray(perm, dtype=np.int64)


def kmeans_faiss(x, k, niter=25, seed=1234, max_points_per_centroid=256):
    """Clustering::train (faiss 1.7.4, default ClusteringParameters, nredo = 1, no weights) -> centroids [k, d] float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    if n > k * max_points_per_centroid:
        x = x[rand_perm(n, seed)[:k * max_points_per_centroid]]
        n = x.shape[0]
    if n == k:
        return x.copy()
    cent = x[rand_perm(n, seed + 1)[:k]].copy()
    eps = np.float32(1.0 / 1024.0)
    for _ in range(niter):
        assign, _ = coarse_assign(x, cent)
        hassign = np.bincount(assign, minlength=k).astype(np.float64)
        new = np.zeros_like(cent)                       # compute_centroids: per-cluster sums of the assigned rows
        order = np.argsort(assign, kind="stable")
        nz = hassign > 0
        starts = np.concatenate([[0], np.cumsum(hassign.astype(np.int64))[:-1]])
        new[nz] = np.add.reduceat(x[order], starts[nz], axis=0)
        new[nz] *= (np.float32(1.0) / hassign[nz].astype(np.float32))[:, None]
        cent = new
        rng = np.random.RandomState(1234)               # split_clusters: RandomGenerator rng(1234)
        for ci in np.flatnonzero(hassign == 0).tolist():
            cj = 0
            while True:
                p = np.float32((hassign[cj] - 1.0) / float(n - k))
                r = np.float32(rng.randint(0, 1 << 32, dtype=np.uint32)) / np.float32(4294967295.0)
                if r < p:
                    break
                cj = (cj + 1) % k
            cent[ci] = cent[cj]
            sign = np.where(np.arange(d) % 2 == 0, 1.0, -1.0).astype(np.float32)
            cent[ci] *= 1 + eps * sign
            cent[cj] *= 1 - eps * sign
            hassign[ci] = hassign[cj] / 2
            hassign[cj] -= hassign[ci]
    return cent


def ivf_build(features, n_ivf=None, **kmeans_args):
    """``FaissIVFFlatTrainableFeatureIndexBuilder`` + ``add_with_train`` (index.py:119-151) -> (centroids, lists)."""
    x = np.ascontiguousarray(features, dtype=np.float32)
    n = x.shape[0]
    if n_ivf is None:
        n_ivf = min(int(16 * np.sqrt(n)), n // 39)
    cent = kmeans_faiss(x, n_ivf, **kmeans_args)
    cell, _ = coarse_assign(x, cent)
    lists = []
    for c in range(n_ivf):
        ids = np.flatnonzero(cell == c).astype(np.int64)
        lists.append((x[ids], ids))
    return cent, lists
