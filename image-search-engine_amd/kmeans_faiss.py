"""``FaissKMeans`` with the reference's surface (backend/kmeans_faiss.py:5-50).
``transform`` (nearest-centroid assignment, ``self.index.search(X, 1)``) is on the
scoped hot path and runs on the GPU as one MFMA-bound kernel; ``fit``
drives ``faiss_compat.Kmeans`` (GPU Lloyd iterations, a "next" row).  A saved
codebook is reloaded the reference's way: ``FaissKMeans(n_clusters, index=read_index(path))``."""
from __future__ import annotations

import numpy as np

from . import faiss_compat as faiss


class FaissKMeans:
    def __init__(self, n_clusters=8, n_init=3, max_iter=25, init_centroids=None, index=None):
        self.n_clusters = n_clusters
        self.n_init = n_init
        self.max_iter = max_iter
        self.inertia_ = None
        self.cluster_centers_ = None
        self.kmeans = None
        self.init_centroids = init_centroids
        self.index = index

    def fit(self, X: np.ndarray, y=None) -> None:
        self.kmeans = faiss.Kmeans(seed=42, d=int(X.shape[1]), k=int(self.n_clusters), niter=self.max_iter,
                                   nredo=self.n_init, spherical=True, verbose=False)
        self.kmeans.train(np.asarray(X).astype(np.float32), init_centroids=self.init_centroids)
        self.index = self.kmeans.index
        self.cluster_centers_ = self.kmeans.centroids
        self.inertia_ = self.kmeans.obj[-1]

    def transform(self, X: np.ndarray) -> np.ndarray:
        """I: the nearest centroid for each row of X, int64 (n, 1)."""
        _, I = self.index.search(np.asarray(X).astype(np.float32), 1)
        return I
