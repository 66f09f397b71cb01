"""Quantisation + histogram half of the reference's ``BOVW`` transformer
(backend/bag_of_visual_words.py:50-117), the caller of the k = 1 assignment on the scoped hot
path ("next" row f-3).

The reference loops over images: ``clusterer.transform(X)`` (one k = 1 search per image) and
``np.histogram(ids, bins=n_clusters)``.  Here the keypoint descriptors of many images are
assigned in ONE batch (the MFMA-bound assignment kernel) and all histograms come from one
kernel launch, one block per image (``ise_bovw_histogram_device``).

Quirk kept: ``np.histogram`` is called without a range, so the bins of an image span that
image's own [min label, max label], not [0, n_clusters) -- the kernel follows numpy's uniform-bin
arithmetic operation for operation and the GPU tests compare with ``np.histogram`` itself.

The keypoint descriptors (corners, SIFT/ORB ...) are out of scope: pass any describer the
``Describer`` of ``descriptors.py`` accepts, or call ``create_visual_word_histogram`` with
precomputed per-image descriptor arrays.
"""
from __future__ import annotations

import numpy as np

from . import _native as _n
from .kmeans_faiss import FaissKMeans

ROWS_PER_BATCH = 1 << 22  # keypoint rows assigned per upload (2 GiB of float32 at d = 128)


def create_visual_word_histogram(images_descriptions, clusterer: FaissKMeans, n_clusters: int) -> np.ndarray:
    """(n_images, n_clusters) float64 histogram of visual words, row i ==
    ``np.histogram(clusterer.transform(images_descriptions[i]), bins=n_clusters)[0]``."""
    import torch

    if clusterer.index is None:
        raise RuntimeError("the clusterer has no centroid index: fit it or pass index=")
    index = clusterer.index
    K = int(n_clusters)
    n_img = len(images_descriptions)
    out = np.zeros((n_img, K))
    if n_img == 0:
        return out
    lens = np.array([np.asarray(X).reshape(-1, index.d).shape[0] if np.asarray(X).size else 0
                     for X in images_descriptions], dtype=np.int64)
    dev = torch.device("cuda", index.device)
    i0 = 0
    while i0 < n_img:
        # as many whole images as fit one batch (at least one)
        i1, rows = i0, 0
        while i1 < n_img and (i1 == i0 or rows + lens[i1] <= ROWS_PER_BATCH):
            rows += int(lens[i1])
            i1 += 1
        offsets = np.zeros(i1 - i0 + 1, dtype=np.int64)
        np.cumsum(lens[i0:i1], out=offsets[1:])
        hist = torch.empty((i1 - i0, K), dtype=torch.float64, device=dev)
        if rows:
            x = np.concatenate([np.asarray(images_descriptions[i], dtype=np.float32).reshape(-1, index.d)
                                for i in range(i0, i1) if lens[i]])
            xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            _, lab = index.assign_torch(xd) if index._assign_applies(rows, 1) else index.search_torch(xd, 1)
            lab = lab.view(-1).contiguous()
        else:
            lab = torch.empty(0, dtype=torch.int64, device=dev)
        off_d = torch.from_numpy(offsets).to(dev)
        _n.check(_n.lib.ise_bovw_histogram_device(lab.data_ptr(), off_d.data_ptr(), i1 - i0, K, hist.data_ptr(),
                                                  index.device, torch.cuda.current_stream(dev).cuda_stream))
        out[i0:i1] = hist.cpu().numpy()
        i0 = i1
    return out


def run_clustering(descriptions, n_clusters: int) -> FaissKMeans:
    """backend/bag_of_visual_words.py:123-134: one k-means over every keypoint descriptor."""
    clusterer = FaissKMeans(n_clusters)
    clusterer.fit(np.concatenate(descriptions, axis=0))
    return clusterer


class BOVW:
    """fit = describe + cluster, transform = describe + quantise + histogram
    (backend/bag_of_visual_words.py:50-117).  Kept: ``transform`` reuses the descriptions of
    ``fit`` when they exist and ignores its argument then (backend/bag_of_visual_words.py:89-92)."""

    def __init__(self, describer, n_clusters: int):
        self.describer = describer
        self.n_clusters = n_clusters
        self.clusterer = None
        self.descriptions = None

    def fit(self, X, y=None):
        from .descriptors import describe_dataset

        self.descriptions = describe_dataset(self.describer, X)
        self.clusterer = run_clustering(self.descriptions, self.n_clusters)
        return self

    def transform(self, X, y=None) -> np.ndarray:
        from .descriptors import describe_dataset

        descriptions = self.descriptions
        if descriptions is None:
            descriptions = describe_dataset(self.describer, X, prediction=True)
        return create_visual_word_histogram(descriptions, self.clusterer, self.n_clusters)

    def fit_transform(self, X, y=None) -> np.ndarray:
        self.fit(X)
        return self.transform(X)
