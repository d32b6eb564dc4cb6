"""``svc_train_retrieval.py`` of the reference (its lines 1-114 + feature_retrieval/train.py, transform.py): one IVF-Flat index per
speaker and feature kind under ``data_svc/indexes/<speaker>/<prefix>{hubert,whisper}.index``, trained and filled on the GPU
(svcmi.ivf_index) and written in faiss's IndexIVFFlat layout, so the reference's ``faiss.read_index`` and this engine's
``load_retrieve_index`` read the same files.

    python -m svcmi.svc_train_retrieval [--prefix P] [--speakers A B ...] [--compress-features-after N] [--n-clusters K] [--n-parallel J]
"""
import argparse
import logging
import multiprocessing
from pathlib import Path

import numpy as np

from .ivf_index import IvfFlatFeatureIndex

logger = logging.getLogger(__name__)


def get_feature_matrix(features_dir_path):
    """feature_retrieval/train.py:36-39."""
    matrices = [np.load(str(p)) for p in Path(features_dir_path).rglob("*.npy")]
    return np.concatenate(matrices, axis=0)


def minibatch_kmeans_transform(matrix, n_clusters, n_parallel):
    """feature_retrieval/transform.py:29-52 -- the reference compresses > 200 000 features to n_clusters centroids with
    scikit-learn's MiniBatchKMeans (a dependency of the reference that is present here); same constructor arguments."""
    from sklearn.cluster import MiniBatchKMeans
    cluster = MiniBatchKMeans(n_clusters=n_clusters, verbose=True, batch_size=n_parallel * 256, compute_labels=False, init="k-means++")
    return cluster.fit(matrix).cluster_centers_


def train_index(features_path, index_save_filepath, compress_features_after, n_clusters, n_parallel, device="cuda", ops=None,
                rewrite=False):
    """feature_retrieval/train.py:11-33."""
    logger.info("start getting feature vectors from %s", Path(features_path).absolute())
    feature_matrix = get_feature_matrix(features_path)
    logger.debug("fetched %s features", feature_matrix.shape[0])
    if feature_matrix.shape[0] > compress_features_after:
        logger.info("pass condition. Transform by rule MinibatchKmeansFeatureTransform")
        feature_matrix = minibatch_kmeans_transform(feature_matrix, n_clusters, n_parallel)
    else:
        logger.info("condition is not passed. Transform by rule DummyFeatureTransform")
    logger.info("adding features to index with training")
    index = IvfFlatFeatureIndex.train(np.ascontiguousarray(feature_matrix, dtype=np.float32), device=device, ops=ops)
    index.save(index_save_filepath, rewrite=rewrite)
    logger.info("index saved to %s", Path(index_save_filepath).absolute())
    return index


def get_speaker_list(base_path):
    speakers_path = Path(base_path) / "waves-16k"
    if not speakers_path.exists():
        raise FileNotFoundError(f"path {speakers_path} does not exists")
    return [d.name for d in speakers_path.iterdir() if d.is_dir()]


def create_index(feature_name, prefix, speaker, base_path, indexes_path, compress_features_after, n_clusters, n_parallel, **kw):
    """svc_train_retrieval.py:33-62."""
    features_path = Path(base_path) / feature_name / speaker
    if not features_path.exists():
        raise ValueError(f"features not found by path {features_path}")
    index_path = Path(indexes_path) / speaker
    index_path.mkdir(exist_ok=True)
    index_filepath = index_path / f"{prefix}{feature_name}.index"
    logger.debug("index will be save to %s", index_filepath)
    return train_index(features_path, index_filepath, compress_features_after, n_clusters, n_parallel, **kw)


def build_parser():
    p = argparse.ArgumentParser("crate faiss indexes for feature retrieval")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--prefix", default="", help="add prefix to index filename")
    p.add_argument("--speakers", nargs="+", help="speaker names to create an index. By default all speakers are from data_svc")
    p.add_argument("--compress-features-after", type=int, default=200_000,
                   help="If the number of features is greater than the value compress feature vectors using MiniBatchKMeans.")
    p.add_argument("--n-clusters", type=int, default=10_000, help="Number of centroids to which features will be compressed")
    p.add_argument("--n-parallel", type=int, default=multiprocessing.cpu_count() - 1,
                   help="Nuber of parallel job of MinibatchKmeans. Default is cpus-1")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.INFO)
    base_path = Path(".").absolute() / "data_svc"
    speakers = args.speakers if args.speakers else get_speaker_list(base_path)
    logger.info("got %s speakers: %s", len(speakers), speakers)
    indexes_path = base_path / "indexes"
    logger.info("create indexes folder %s", indexes_path)
    indexes_path.mkdir(exist_ok=True)
    for speaker in speakers:
        for feature_name in ("hubert", "whisper"):
            logger.info("create %s index for speaker %s", feature_name, speaker)
            create_index(feature_name, args.prefix, speaker, base_path, indexes_path, args.compress_features_after, args.n_clusters,
                         args.n_parallel)
    logger.info("done!")


if __name__ == "__main__":
    main()
