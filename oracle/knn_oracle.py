"""CPU oracle for the brute-force kNN hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package never does: it fails loudly
when the HIP extension is missing instead of falling back to anything here.

PARITY UNPINNED.  The reference (ManuelZ/image-search-engine @ 2024_10_08)
ships no tests, golden vectors or fixtures for this path,
and the arithmetic lives in a third-party dependency that is absent from
/root/reference and from this image: PyPI ``faiss-cpu``, unpinned
(backend/siamese/requirements.txt:2; contemporaneous releases 1.8.0 / 1.9.0).
What is restated here is therefore

  * the reference's own call shapes and dtypes
      - backend/engine.py:46-57        run_image_query -> index.search(x(1,d), k)
      - backend/utils.py:293-330       create_search_index (l2 / cosine + add)
      - backend/kmeans_faiss.py:46-50  FaissKMeans.transform -> index.search(X, 1)
  * the only in-repo restatement of brute-force kNN semantics
      - backend/siamese/test_index.py:58-69   normalise, per-row Euclidean
        distance, ascending argsort, take n (non-squared L2 there; Faiss
        IndexFlatL2 reports SQUARED L2, which is what this oracle reports)
      - backend/siamese/siamese_tf/create_index.py:62-85  float64 unit-norm rows
  * Faiss's published IndexFlat semantics [upstream-faiss, restated from the
    published algorithm, not from a source tree]:
      - L2 = squared distance, ascending; IP = inner product, descending
      - a row enters the result only if it is STRICTLY better than the current
        k-th best, which starts at the heap's neutral value (+FLT_MAX for L2,
        -FLT_MAX for IP); so rows at distance >= FLT_MAX (inf, NaN) never
        enter, and unfilled slots come back as id -1 / dist +-FLT_MAX
      - ties are ordered by ascending id
      - normalize_L2: x_i <- x_i * (float)(1.0 / sqrtf(|x_i|^2)), rows with
        zero norm untouched.

Ground truth is computed in float64 (exact to ~1e-13 relative on the data
used), distances are rounded to float32 once at the end.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = np.float32(np.finfo(np.float32).max)

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def _as_f32_2d(x) -> np.ndarray:
    """Coerce like the Faiss python wrapper does (np.matrix accepted: SURVEY 5.9-11)."""
    x = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
    if x.ndim != 2:
        raise ValueError("expected a 2-D array")
    return x


def pairwise_f64(xq: np.ndarray, xb: np.ndarray, metric: int) -> np.ndarray:
    """(nq, N) float64 scores: squared L2 or inner product.

    Squared L2 is evaluated in the direct-difference form per block, which has
    no cancellation (reference semantics: backend/siamese/test_index.py:62-64).
    """
    q = xq.astype(np.float64)
    b = xb.astype(np.float64)
    if metric == METRIC_INNER_PRODUCT:
        return q @ b.T
    out = np.empty((q.shape[0], b.shape[0]), dtype=np.float64)
    # block over index rows to bound the (nq, B, d) temporary
    step = max(1, int(2**24 // max(1, q.shape[0] * q.shape[1])))
    for s in range(0, b.shape[0], step):
        diff = q[:, None, :] - b[None, s : s + step, :]
        out[:, s : s + step] = np.einsum("qbd,qbd->qb", diff, diff)
    return out


def _select_sorted(scores: np.ndarray, ids: np.ndarray, k: int, metric: int):
    """k best of one query under the total order (score, id); strict FLT_MAX gate."""
    if metric == METRIC_L2:
        ok = scores < float(FLT_MAX)  # NaN compares False -> never enters
        key = scores
    else:
        ok = scores > -float(FLT_MAX)
        key = -scores
    ids_ok = ids[ok]
    key_ok = key[ok]
    order = np.lexsort((ids_ok, key_ok))[:k]
    return key_ok[order], ids_ok[order]


def knn_exact(xb, xq, k: int, metric: int = METRIC_L2, id_offset: int = 0):
    """Exact kNN: returns (D float32 (nq,k), I int64 (nq,k)).

    Mirrors IndexFlatL2/IP.search as the reference calls it
    (backend/engine.py:55, backend/siamese/test_index.py:54,
    backend/kmeans_faiss.py:49).
    """
    xb = _as_f32_2d(xb) if np.size(xb) else np.zeros((0, np.shape(xq)[1]), np.float32)
    xq = _as_f32_2d(xq)
    nq = xq.shape[0]
    n = xb.shape[0]
    pad_d = FLT_MAX if metric == METRIC_L2 else -FLT_MAX
    D = np.full((nq, k), pad_d, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0 or k == 0:
        return D, I
    ids = np.arange(n, dtype=np.int64) + id_offset
    # block over queries so the (nq, N) float64 matrix stays bounded
    qstep = max(1, int(2**27 // max(1, n)))
    for q0 in range(0, nq, qstep):
        S = _pairwise_blocked(xq[q0 : q0 + qstep], xb, metric)
        for i in range(S.shape[0]):
            key, sel = _select_sorted(S[i], ids, k, metric)
            m = len(sel)
            val = key if metric == METRIC_L2 else -key
            D[q0 + i, :m] = val.astype(np.float32)
            I[q0 + i, :m] = sel
    return D, I


def _pairwise_blocked(xq, xb, metric):
    """float64 scores; for big N use the expanded form in float64 (error ~1e-13,
    far below float32 resolution) so the oracle finishes in seconds."""
    if xb.shape[0] * xq.shape[0] * xb.shape[1] <= 2**26:
        return pairwise_f64(xq, xb, metric)
    q = xq.astype(np.float64)
    out = np.empty((q.shape[0], xb.shape[0]), dtype=np.float64)
    step = 1 << 17
    # L2 is translation invariant: taking it around the column mean keeps the expanded form's
    # float64 error (~1e-16 of the squared norms) far below float32 resolution of the distances even
    # when the rows sit far from the origin or in far-apart clusters
    centre = np.zeros(q.shape[1])
    if metric != METRIC_INNER_PRODUCT:
        for s in range(0, xb.shape[0], step):
            centre += xb[s : s + step].astype(np.float64).sum(0)
        centre /= xb.shape[0]
        q = q - centre
    qn = np.einsum("ij,ij->i", q, q)
    for s in range(0, xb.shape[0], step):
        b = xb[s : s + step].astype(np.float64) - centre
        ip = q @ b.T
        if metric == METRIC_INNER_PRODUCT:
            out[:, s : s + step] = ip
        else:
            bn = np.einsum("ij,ij->i", b, b)
            out[:, s : s + step] = np.maximum(qn[:, None] + bn[None, :] - 2.0 * ip, 0.0)
    return out


def kth_gap(xb, xq, k: int, metric: int = METRIC_L2) -> np.ndarray:
    """Per query: smallest |score| gap between consecutive ranks 1..k+1 (float64).

    Fixtures store this so a test can tell a genuine id mismatch from a
    float32 near-tie."""
    xb = _as_f32_2d(xb)
    xq = _as_f32_2d(xq)
    S = _pairwise_blocked(xq, xb, metric)
    key = S if metric == METRIC_L2 else -S
    kk = min(k + 1, key.shape[1])
    part = np.sort(np.partition(key, kk - 1, axis=1)[:, :kk], axis=1)
    if kk < 2:
        return np.full(key.shape[0], np.inf)
    return np.min(np.diff(part, axis=1), axis=1)


def normalize_rows(x: np.ndarray) -> np.ndarray:
    """Out-of-place restatement of faiss.normalize_L2 [upstream-faiss fvec_renorm_L2]
    as called at backend/utils.py:303, backend/engine.py:53,
    backend/siamese/test_index.py:53.  float32 in, float32 out; the squared
    norm is accumulated in float64 here (the oracle is the exact value), the
    scale is (float)(1.0 / sqrt(nr)); zero rows are left untouched."""
    x = _as_f32_2d(x).copy()
    nr = np.einsum("ij,ij->i", x.astype(np.float64), x.astype(np.float64))
    nz = nr > 0
    inv = np.ones_like(nr)
    inv[nz] = 1.0 / np.sqrt(nr[nz].astype(np.float32).astype(np.float64))
    x[nz] = (x[nz] * inv[nz, None].astype(np.float32)).astype(np.float32)
    return x


def assign_nearest(X, centroids, metric: int = METRIC_INNER_PRODUCT) -> np.ndarray:
    """FaissKMeans.transform (backend/kmeans_faiss.py:46-50): index.search(X, 1)
    over the centroid index -> I int64 (n, 1).  spherical=True at
    backend/kmeans_faiss.py:36 makes that index an inner-product one over
    unit-norm centroids [upstream-faiss]; metric=METRIC_L2 gives the argmin-L2
    form BASELINE config 4 names."""
    _, I = knn_exact(centroids, X, 1, metric)
    return I


def lloyd_reference(x, c0, niter: int, spherical: bool = False):
    """float64 restatement of the k-means iteration Faiss's ``Clustering::train`` runs from given initial
    centroids [upstream-faiss Clustering.cpp; the reference reaches it through ``faiss.Kmeans(...).train``,
    backend/kmeans_faiss.py:29-44]: per iteration assign every row to its nearest centroid (spherical: largest
    inner product with unit centroids), record the objective (``Kmeans.obj``: summed squared distances, or
    summed inner products), replace every centroid by the mean of its rows (spherical: renormalised).  Faiss
    re-seeds EMPTY clusters by splitting large ones with its own RNG -- not restated: an empty cluster raises.
    Returns (centroids float64 (k, d), objective per iteration, labels of the last assignment)."""
    x64 = np.asarray(x, dtype=np.float64)
    c = np.asarray(c0, dtype=np.float64).copy()
    k = c.shape[0]
    obj, lab = [], None

    def unit(a):
        nrm = np.linalg.norm(a, axis=1, keepdims=True)
        return np.where(nrm > 0, a / np.where(nrm > 0, nrm, 1.0), a)

    for _ in range(int(niter)):
        if spherical:
            c = unit(c)
            s = x64 @ c.T
            lab = np.argmax(s, axis=1)          # ties: lowest centroid id, as the index search returns
            obj.append(float(s[np.arange(len(lab)), lab].sum()))
        else:
            d2 = (x64 * x64).sum(1, keepdims=True) + (c * c).sum(1)[None, :] - 2.0 * (x64 @ c.T)
            lab = np.argmin(d2, axis=1)
            obj.append(float(np.maximum(d2[np.arange(len(lab)), lab], 0.0).sum()))
        cnt = np.bincount(lab, minlength=k)
        if (cnt == 0).any():
            raise ValueError("empty cluster: Faiss's re-seeding is not restated")
        sums = np.zeros_like(c)
        np.add.at(sums, lab, x64)
        c = sums / cnt[:, None]
    if spherical:
        c = unit(c)
    return c, obj, lab


def merge_shards(D_parts, I_parts, k: int, metric: int = METRIC_L2):
    """Merge per-shard (D, I) lists (ids already global) under (score, id) order.
    Restates what one all-gather + merge must produce."""
    D_all = np.concatenate(D_parts, axis=1)
    I_all = np.concatenate(I_parts, axis=1)
    nq = D_all.shape[0]
    pad_d = FLT_MAX if metric == METRIC_L2 else -FLT_MAX
    D = np.full((nq, k), pad_d, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for i in range(nq):
        ok = I_all[i] >= 0
        d = D_all[i][ok]
        ids = I_all[i][ok]
        key = d if metric == METRIC_L2 else -d
        order = np.lexsort((ids, key))[:k]
        D[i, : len(order)] = d[order]
        I[i, : len(order)] = ids[order]
    return D, I


def knn_blas_f32(xb, xq, k: int, metric: int = METRIC_L2, bs_x: int = 4096, bs_y: int = 1024):
    """float32 restatement of Faiss's large-batch path (nq >= 20)
    [upstream-faiss exhaustive_L2sqr_blas]: blocks of 4096 queries x 1024 index
    rows, ip by SGEMM, dis = |x|^2 + |y|^2 - 2 ip clamped at 0, running top-k.
    Used as the nq >= 20 leg of bench.py's cpu_baseline (kind "port")."""
    xb = _as_f32_2d(xb)
    xq = _as_f32_2d(xq)
    nq, n = xq.shape[0], xb.shape[0]
    kk = min(k, n)
    D = np.full((nq, k), FLT_MAX if metric == METRIC_L2 else -FLT_MAX, np.float32)
    I = np.full((nq, k), -1, np.int64)
    xn = np.einsum("ij,ij->i", xq, xq)
    yn = np.einsum("ij,ij->i", xb, xb)
    for i0 in range(0, nq, bs_x):
        q = xq[i0 : i0 + bs_x]
        best_d = np.empty((q.shape[0], 0), np.float32)
        best_i = np.empty((q.shape[0], 0), np.int64)
        # several 1024-row database blocks are fused per GEMM call; the
        # arithmetic per element is unchanged
        step = bs_y * 64
        for j0 in range(0, n, step):
            ip = q @ xb[j0 : j0 + step].T
            if metric == METRIC_L2:
                dis = xn[i0 : i0 + bs_x, None] + yn[None, j0 : j0 + step] - 2 * ip
                np.maximum(dis, 0, out=dis)
                key = dis
            else:
                key = -ip
            m = min(kk, key.shape[1])
            part = np.argpartition(key, m - 1, axis=1)[:, :m]
            cd = np.take_along_axis(key, part, axis=1)
            ci = part.astype(np.int64) + j0
            best_d = np.concatenate([best_d, cd], axis=1)
            best_i = np.concatenate([best_i, ci], axis=1)
            if best_d.shape[1] > 4 * kk:
                sel = np.argpartition(best_d, kk - 1, axis=1)[:, :kk]
                best_d = np.take_along_axis(best_d, sel, axis=1)
                best_i = np.take_along_axis(best_i, sel, axis=1)
        for r in range(q.shape[0]):
            order = np.lexsort((best_i[r], best_d[r]))[:kk]
            val = best_d[r][order]
            D[i0 + r, :kk] = val if metric == METRIC_L2 else -val
            I[i0 + r, :kk] = best_i[r][order]
    return D, I
