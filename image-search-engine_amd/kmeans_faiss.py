"""Nearest-centroid quantiser with the surface of the reference's ``FaissKMeans``
(backend/kmeans_faiss.py:5-50): constructor arguments, ``fit``, ``transform``, and the
attributes ``index``, ``cluster_centers_``, ``inertia_``, ``kmeans``.

What runs where
  * ``transform`` -- the k = 1 search ``self.index.search(X, 1)`` the BoVW histogram loop calls
    per image (backend/bag_of_visual_words.py:98-106) -- is on the scoped hot path (SURVEY.md
    a11).  Batches of >= 2048 rows go through the MFMA-bound assignment kernel, smaller ones
    through the general scan; both return int64 ids of shape (n, 1).
  * ``fit`` trains through ``faiss_compat.Kmeans`` with the reference's settings (seed 42,
    spherical, ``nredo = n_init``, ``niter = max_iter``): Lloyd iterations on the GPU, a "next"
    row.
  * a saved codebook comes back the reference's way, without training:
    ``FaissKMeans(n_clusters, index=faiss.read_index(path))``
    (backend/bag_of_visual_words.py:207-216).
"""
from __future__ import annotations

import numpy as np

from . import faiss_compat as faiss

_SEED = 42  # backend/kmeans_faiss.py:30


class FaissKMeans:
    def __init__(self, n_clusters=8, n_init=3, max_iter=25, init_centroids=None, index=None):
        self.n_clusters = n_clusters
        self.n_init = n_init
        self.max_iter = max_iter
        self.init_centroids = init_centroids
        self.index = index
        self.kmeans = None
        self.cluster_centers_ = None
        self.inertia_ = None

    # -- training ("next" row)
    def fit(self, X: np.ndarray, y=None) -> None:
        rows = self._rows(X)
        self.kmeans = faiss.Kmeans(d=rows.shape[1], k=int(self.n_clusters), niter=int(self.max_iter),
                                   nredo=int(self.n_init), seed=_SEED, spherical=True, verbose=False)
        self.kmeans.train(rows, init_centroids=self.init_centroids)
        self.index = self.kmeans.index
        self.cluster_centers_ = self.kmeans.centroids
        self.inertia_ = self.kmeans.obj[-1]

    # -- assignment (hot path)
    def transform(self, X: np.ndarray) -> np.ndarray:
        """Id of the nearest centroid of every row of X: int64, shape (n, 1)."""
        if self.index is None:
            raise RuntimeError("FaissKMeans has no centroid index: call fit() or pass index=")
        _, nearest = self.index.search(self._rows(X), 1)
        return nearest

    def fit_transform(self, X: np.ndarray, y=None) -> np.ndarray:
        self.fit(X)
        return self.transform(X)

    @staticmethod
    def _rows(X) -> np.ndarray:
        rows = np.asarray(X)
        if rows.ndim != 2:
            raise ValueError("expected a 2-D array of descriptors (n, d)")
        return rows.astype(np.float32)
