"""``query_index`` with the reference's surface (backend/siamese/test_index.py:49-71).

``index_type == "faiss"``: normalise the embedding in place, ``index.search`` (inner product
over unit rows = cosine).  ``index_type == "dict"``: the reference's pure-numpy brute force over
a float64 unit-norm matrix (built at backend/siamese/siamese_tf/create_index.py:62-85):
normalise, per-row Euclidean distance, ascending argsort, take n -- note NON-squared L2 and the
``(indices, distances)`` return order.  Here the dict branch runs on the GPU through an
``IndexFlatL2`` over the same rows (cached per matrix) and takes the square root at the end."""
from __future__ import annotations

import numpy as np

from . import faiss_compat as faiss

_dict_cache: dict[int, tuple] = {}


def _index_for(matrix: np.ndarray):
    key = id(matrix)
    hit = _dict_cache.get(key)
    if hit is None or hit[0] is not matrix:
        idx = faiss.IndexFlatL2(matrix.shape[1])
        idx.add(np.ascontiguousarray(matrix, dtype=np.float32))
        _dict_cache.clear()  # one matrix at a time, like the reference's module-level index
        _dict_cache[key] = (matrix, idx)
        return idx
    return hit[1]


def query_index(embedding, index, index_type, n_results):
    if index_type == "faiss":
        faiss.normalize_L2(embedding)
        distances, indices = index.search(embedding, n_results)
        indices = indices.ravel().tolist()
        distances = distances.ravel().tolist()
    elif index_type == "dict":
        embedding = embedding / np.linalg.norm(embedding)
        q = np.ascontiguousarray(np.asarray(embedding, dtype=np.float32).reshape(1, -1))
        d2, ids = _index_for(index).search(q, n_results)
        indices = ids.ravel()
        distances = np.sqrt(d2.ravel().astype(np.float64))
    else:
        raise ValueError(f"unknown index_type {index_type!r}")
    return indices, distances
