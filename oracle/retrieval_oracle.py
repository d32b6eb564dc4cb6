"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's feature retrieval blend (row N4).

Follows feature_retrieval/index.py:57-62 (``retriv``) and :75-94 (``_weight_nearest_vectors``).  The search itself is
faiss (faiss-cpu, unpinned in the reference's requirements.txt, NOT installed here): ``search_and_reconstruct`` on a
METRIC_L2 index returns the k smallest SQUARED L2 distances in ascending order with the stored vectors.  This restatement
does the exhaustive search (IVF with nprobe = nlist); the reference's nprobe = 1 (index.py:150) approximates it.
Parity unpinned: no faiss here, so there is no reference output for this row; the weighting lines are numpy and are
restated operation by operation.
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
