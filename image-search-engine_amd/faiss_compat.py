"""Faiss-shaped surface over the gfx950 kNN library.

Exactly the names the reference uses from ``faiss`` on its hot path
(SURVEY.md 8b), so a maintainer can write
``import image_search_engine_amd.faiss_compat as faiss`` in
backend/utils.py:12, backend/engine.py:13, backend/indexer.py:9,
backend/kmeans_faiss.py:1 and backend/siamese/test_index.py:

    IndexFlatL2(d) / IndexFlatIP(d)      backend/utils.py:302,306
    index.add(x) / .ntotal / .d          backend/utils.py:327-328, backend/engine.py:117
    index.search(x, k) -> (D, I)         backend/engine.py:55, backend/kmeans_faiss.py:49
    normalize_L2(x)                      backend/utils.py:303, backend/engine.py:53
    write_index / read_index             backend/indexer.py:59, backend/engine.py:116
    Kmeans(...).index / .centroids       backend/kmeans_faiss.py:29-44 (assignment only)

All arithmetic runs on the MI355X through ``include/ise_knn.h``; there is no
CPU path here.  Without the HIP library or without a GPU the constructors and
``normalize_L2`` raise.
"""
from __future__ import annotations

import ctypes
import struct
import threading

import numpy as np

from . import _native as _n

METRIC_INNER_PRODUCT = _n.METRIC_INNER_PRODUCT
METRIC_L2 = _n.METRIC_L2

_FLT_MAX = float(np.finfo(np.float32).max)


def _as_rows(x, d=None) -> np.ndarray:
    """Coerce like the Faiss SWIG wrapper: C-contiguous float32 (n, d).
    ``np.matrix`` (what ``.todense()`` yields, backend/engine.py:96) is accepted."""
    x = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
    assert x.ndim == 2, "expected a 2-D array"
    if d is not None:
        assert x.shape[1] == d, f"dimension mismatch: got {x.shape[1]}, index has d={d}"
    return x


def _default_device() -> int:
    import torch

    return torch.cuda.current_device() if torch.cuda.is_available() else 0


class IndexFlat:
    """Exhaustive-search index owning a device copy of its rows.

    Mirrors faiss.IndexFlat as the reference touches it: ``d``, ``ntotal``,
    ``is_trained``, ``metric_type``, ``add``, ``search``, ``reset``,
    ``reconstruct_n``.  Thread-safe for concurrent ``search`` callers (Flask
    request threads, backend/engine.py:137; joblib threads,
    backend/descriptors.py:125); the GIL is released while the device works.
    """

    def __init__(self, d: int, metric: int = METRIC_L2, device: int | None = None, storage: str = "f32"):
        """``storage="bf16"`` (extension, not a Faiss IndexFlat feature) keeps rows as bf16 and
        rounds queries to bf16 too: approximate results at half the HBM traffic (BASELINE config 5)."""
        self.d = int(d)
        self.metric_type = int(metric)
        self.is_trained = True
        self.storage = storage
        self.device = _default_device() if device is None else int(device)
        self._h = ctypes.c_void_p()
        self._lock = threading.Lock()
        store = {"f32": _n.STORE_F32, "bf16": _n.STORE_BF16}[storage]
        _n.check(_n.lib.ise_index_create_ex(ctypes.byref(self._h), self.d, self.metric_type, self.device, store))

    # -- lifetime
    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _n.lib.ise_index_destroy(h)
            except Exception:  # interpreter shutdown: module globals may already be gone
                pass
            h.value = None

    @property
    def ntotal(self) -> int:
        n = ctypes.c_int64(0)
        _n.check(_n.lib.ise_index_info(self._h, None, None, ctypes.byref(n), None))
        return int(n.value)

    def reset(self) -> None:
        _n.check(_n.lib.ise_index_reset(self._h))

    def exact_stats(self) -> dict:
        """Counters of the exact float32 L2 path (include/ise_knn.h, ise_index_stats): queries
        re-ranked, queries the certificate sent to the exact direct-difference scan, refreshes of the
        shift vector."""
        out = (ctypes.c_uint64 * 4)()
        _n.check(_n.lib.ise_index_stats(self._h, out))
        return {"reranked": int(out[0]), "exact_scan": int(out[1]), "shift_updates": int(out[2]),
                "gemm_chunks": int(out[3])}

    def host_stats(self) -> dict:
        """Combining of concurrent ``search`` calls (include/ise_knn.h, ise_index_host_stats): how many
        shared batches ran and how many calls they served; and how many queries the direct small-batch
        scan answered (float32 L2, at most 4 queries, k <= 32)."""
        out = (ctypes.c_uint64 * 3)()
        _n.check(_n.lib.ise_index_host_stats(self._h, out))
        return {"combined_batches": int(out[0]), "combined_calls": int(out[1]), "direct_queries": int(out[2])}

    def short_stats(self) -> dict:
        """Batches scanned by the short-index kernel (include/ise_knn.h, ise_index_short_stats)."""
        out = (ctypes.c_uint64 * 1)()
        _n.check(_n.lib.ise_index_short_stats(self._h, out))
        return {"short_batches": int(out[0])}

    def reserve(self, nq: int, k: int) -> None:
        """Size every internal workspace for batches of ``nq`` queries / ``k`` results now, so that the
        first search of that shape allocates nothing (serving loops, bench.py)."""
        _n.check(_n.lib.ise_index_reserve_workspaces(self._h, int(nq), int(k)))

    # float32 L2 indexes filter around a shift vector (see include/ise_knn.h); results do not depend
    # on it
    def get_shift(self) -> np.ndarray:
        mu = np.zeros(self.d, dtype=np.float32)
        _n.check(_n.lib.ise_index_get_shift(self._h, mu.ctypes.data))
        return mu

    def set_shift(self, mu) -> None:
        mu = np.ascontiguousarray(mu, dtype=np.float32)
        assert mu.shape == (self.d,)
        _n.check(_n.lib.ise_index_set_shift(self._h, mu.ctypes.data))

    # -- build side
    def add(self, x) -> None:
        """Append rows (copied; the caller may mutate or free ``x`` afterwards)."""
        x = _as_rows(x, self.d)
        _n.check(_n.lib.ise_index_add_host(self._h, x.ctypes.data, x.shape[0]))

    def add_torch(self, x) -> None:
        """Append rows from a CUDA float32 tensor on this index's device (no host hop)."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == self.d
        x = x.contiguous()
        st = torch.cuda.current_stream(x.device).cuda_stream
        _n.check(_n.lib.ise_index_add_device(self._h, x.data_ptr(), x.shape[0], st))
        torch.cuda.current_stream(x.device).synchronize()  # x may be freed by the caller

    def reconstruct_n(self, i0: int = 0, n: int | None = None) -> np.ndarray:
        n = self.ntotal - i0 if n is None else n
        out = np.empty((n, self.d), dtype=np.float32)
        _n.check(_n.lib.ise_index_reconstruct_host(self._h, i0, n, out.ctypes.data))
        return out

    # -- query side
    ASSIGN_MIN_NQ = 2048  # k = 1 searches with at least this many rows use the assignment kernel

    def _assign_applies(self, nq: int, k: int) -> bool:
        return (k == 1 and nq >= self.ASSIGN_MIN_NQ and self.storage == "f32" and self.d <= 512
                and 0 < self.ntotal <= 65536)

    def assign_torch(self, x):
        """Nearest index row of every row of ``x`` (CUDA float32 (n, d)): the k = 1 search of
        FaissKMeans.transform (backend/kmeans_faiss.py:49) as one MFMA-bound kernel.
        Returns CUDA (D float32 (n, 1), I int64 (n, 1))."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == self.d
        x = x.contiguous()
        n = x.shape[0]
        D = torch.empty((n, 1), dtype=torch.float32, device=x.device)
        I = torch.empty((n, 1), dtype=torch.int64, device=x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        _n.check(_n.lib.ise_index_assign_device(self._h, x.data_ptr(), n, D.data_ptr(), I.data_ptr(), st))
        return D, I

    def search(self, x, k: int):
        """(D float32 (nq,k), I int64 (nq,k)), fresh arrays.  L2: squared distance
        ascending; IP: descending; unfilled slots -1 / +-FLT_MAX."""
        x = _as_rows(x, self.d)
        k = int(k)
        assert k > 0
        nq = x.shape[0]
        if self._assign_applies(nq, k):
            import torch

            D = np.empty((nq, 1), dtype=np.float32)
            I = np.empty((nq, 1), dtype=np.int64)
            dev = torch.device("cuda", self.device)
            step = max(1, (1 << 28) // (4 * self.d))  # 256 MiB of rows per upload
            for i0 in range(0, nq, step):
                d_, i_ = self.assign_torch(torch.from_numpy(x[i0:i0 + step]).to(dev))
                D[i0:i0 + step] = d_.cpu().numpy()
                I[i0:i0 + step] = i_.cpu().numpy()
            return D, I
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _n.check(_n.lib.ise_index_search_host(self._h, x.ctypes.data, nq, k, D.ctypes.data, I.ctypes.data))
        return D, I

    def search_torch(self, xq, k: int):
        """Device-resident search: CUDA float32 (nq,d) in, CUDA (D, I) out, enqueued on
        the current torch stream (no host synchronisation)."""
        import torch

        assert xq.is_cuda and xq.dtype == torch.float32 and xq.dim() == 2 and xq.shape[1] == self.d
        xq = xq.contiguous()
        nq = xq.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=xq.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=xq.device)
        st = torch.cuda.current_stream(xq.device).cuda_stream
        with self._lock:
            _n.check(_n.lib.ise_index_search_device(self._h, xq.data_ptr(), nq, int(k), D.data_ptr(),
                                                    I.data_ptr(), st))
        return D, I

    # -- allocation-free forms for latency-critical loops (bench.py, sharded search): the caller owns
    # every buffer and names the stream; nothing is checked beyond what the C ABI checks
    def search_into(self, xq, k: int, D, I, stream: int) -> None:
        with self._lock:
            _n.check(_n.lib.ise_index_search_device(self._h, xq.data_ptr(), xq.shape[0], int(k), D.data_ptr(),
                                                    I.data_ptr(), stream))

    def search_keys_into(self, xq, k: int, id_base: int, keys, stream: int) -> None:
        with self._lock:
            _n.check(_n.lib.ise_index_search_keys_device(self._h, xq.data_ptr(), xq.shape[0], int(k), int(id_base),
                                                         keys.data_ptr(), stream))

    def search_keys_torch(self, xq, k: int, id_base: int = 0):
        """Shard-local search for the multi-GPU path: packed uint64 candidates
        (as an int64 tensor (nq,k)), see include/ise_knn.h."""
        import torch

        assert xq.is_cuda and xq.dtype == torch.float32 and xq.dim() == 2 and xq.shape[1] == self.d
        xq = xq.contiguous()
        nq = xq.shape[0]
        keys = torch.empty((nq, k), dtype=torch.int64, device=xq.device)
        st = torch.cuda.current_stream(xq.device).cuda_stream
        with self._lock:
            _n.check(_n.lib.ise_index_search_keys_device(self._h, xq.data_ptr(), nq, int(k), int(id_base),
                                                         keys.data_ptr(), st))
        return keys

    def search_timed_torch(self, xq, k: int, iters: int):
        """bench.py hook: (D, I, scan_ms_avg, merge_ms_avg) with HIP events on the stream
        the kernels run on."""
        import torch

        xq = xq.contiguous()
        nq = xq.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=xq.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=xq.device)
        st = torch.cuda.current_stream(xq.device).cuda_stream
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        with self._lock:
            _n.check(_n.lib.ise_index_search_timed_device(self._h, xq.data_ptr(), nq, int(k), D.data_ptr(),
                                                          I.data_ptr(), st, int(iters), ctypes.byref(a),
                                                          ctypes.byref(b)))
        return D, I, float(a.value), float(b.value)


class IndexFlatL2(IndexFlat):
    def __init__(self, d: int, device: int | None = None, storage: str = "f32"):
        super().__init__(d, METRIC_L2, device, storage)


class IndexFlatIP(IndexFlat):
    def __init__(self, d: int, device: int | None = None, storage: str = "f32"):
        super().__init__(d, METRIC_INNER_PRODUCT, device, storage)


def merge_keys_torch(keys, metric: int):
    """Merge all-gathered candidate lists: ``keys`` int64 CUDA (n_lists, nq, k) ->
    (D float32 (nq,k), I int64 (nq,k)) on the same device."""
    import torch

    assert keys.is_cuda and keys.dtype == torch.int64 and keys.dim() == 3
    keys = keys.contiguous()
    n_lists, nq, k = keys.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=keys.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=keys.device)
    st = torch.cuda.current_stream(keys.device).cuda_stream
    _n.check(_n.lib.ise_merge_keys_device(keys.data_ptr(), n_lists, nq, k, int(metric), D.data_ptr(), I.data_ptr(),
                                          keys.device.index, st))
    return D, I


def merge_keys_into(keys, metric: int, D, I, stream: int) -> None:
    """Allocation-free form of ``merge_keys_torch``: keys int64 CUDA (n_lists, nq, k) contiguous."""
    n_lists, nq, k = keys.shape
    _n.check(_n.lib.ise_merge_keys_device(keys.data_ptr(), n_lists, nq, k, int(metric), D.data_ptr(), I.data_ptr(),
                                          keys.device.index, stream))


def normalize_L2(x) -> None:
    """In-place row L2 normalisation (faiss.normalize_L2): float32 C-contiguous (n, d)
    required, returns None, zero rows untouched.  Runs on the GPU."""
    import torch

    if isinstance(x, torch.Tensor):
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
        st = torch.cuda.current_stream(x.device).cuda_stream
        _n.check(_n.lib.ise_normalize_rows_device(x.data_ptr(), x.shape[0], x.shape[1], x.device.index, st))
        return None
    assert isinstance(x, np.ndarray) and x.dtype == np.float32 and x.ndim == 2 and x.flags.c_contiguous, \
        "normalize_L2 needs a C-contiguous float32 (n, d) array"
    _n.check(_n.lib.ise_normalize_rows_host(x.ctypes.data, x.shape[0], x.shape[1], _default_device()))
    return None


# ---------------------------------------------------------------- persistence
# Faiss on-disk IndexFlat layout [upstream-faiss index_write.cpp, restated from
# the published format; no sample .faiss file exists in the reference, so byte
# compatibility with real Faiss files is UNPINNED]:
#   fourcc "IxF2" (L2) / "IxFI" (IP); int32 d; int64 ntotal; int64 1<<20;
#   int64 1<<20; uint8 is_trained; int32 metric_type; uint64 count (= N*d
#   float32 words); count float32.
_FOURCC = {METRIC_L2: b"IxF2", METRIC_INNER_PRODUCT: b"IxFI"}
_HDR = struct.Struct("<4siqqqBi")


def serialize_flat(d: int, metric: int, xb: np.ndarray) -> bytes:
    xb = np.ascontiguousarray(xb, dtype="<f4")
    n = xb.shape[0] if xb.size else 0
    head = _HDR.pack(_FOURCC[metric], d, n, 1 << 20, 1 << 20, 1, metric)
    return head + struct.pack("<Q", n * d) + xb.tobytes()


def parse_flat(buf: bytes):
    """-> (d, metric, xb float32 (n, d)); raises RuntimeError on a foreign file."""
    if len(buf) < _HDR.size + 8:
        raise RuntimeError("truncated index file")
    fourcc, d, n, _, _, _, metric = _HDR.unpack_from(buf, 0)
    if fourcc not in (b"IxF2", b"IxFI", b"IxFl"):
        raise RuntimeError(f"unsupported index type {fourcc!r}: only flat indexes are readable")
    if fourcc == b"IxF2":
        metric = METRIC_L2
    elif fourcc == b"IxFI":
        metric = METRIC_INNER_PRODUCT
    (count,) = struct.unpack_from("<Q", buf, _HDR.size)
    if count != n * d or len(buf) < _HDR.size + 8 + 4 * count:
        raise RuntimeError("corrupt flat index payload")
    xb = np.frombuffer(buf, dtype="<f4", count=count, offset=_HDR.size + 8).reshape(n, d).astype(np.float32)
    return d, metric, xb


def write_index(index: IndexFlat, path) -> None:
    with open(str(path), "wb") as f:
        f.write(serialize_flat(index.d, index.metric_type, index.reconstruct_n(0, index.ntotal)))


def read_index(path, device: int | None = None) -> IndexFlat:
    with open(str(path), "rb") as f:
        d, metric, xb = parse_flat(f.read())
    index = IndexFlat(d, metric, device)
    if xb.shape[0]:
        index.add(xb)
    return index


class IndexIVFPQ:  # backend/utils.py:323 ("cell-probe"): approximate, out of scope
    def __init__(self, *a, **kw):
        raise NotImplementedError("IndexIVFPQ ('cell-probe') is outside the exact brute-force hot path")


class Kmeans:
    """``faiss.Kmeans`` as the reference drives it (backend/kmeans_faiss.py:29-44):
    ``Kmeans(d=, k=, niter=25, nredo=3, seed=42, spherical=True)``, ``.train(x, init_centroids=)``,
    then ``.index`` (the assignment index: inner product over unit centroids when spherical,
    L2 otherwise [upstream-faiss]), ``.centroids`` and ``.obj``.

    Assignment (``.index.search(X, 1)``) is the scoped hot path.  ``train`` is a
    "next" row and is provided as plain Lloyd iterations on the GPU: the assignment
    step is the MFMA assignment kernel, the centroid update a device scatter-add.  Faiss seeds its
    initial centroids from its own RNG, so trained centroids are not comparable run-for-run with
    Faiss's (parity unpinned).  ``init_centroids`` starts the iterations from a given codebook, as in
    Faiss; the reference reloads a saved model without training, by handing the index to
    ``FaissKMeans(index=...)`` (backend/bag_of_visual_words.py:207-216)."""

    def __init__(self, d, k, niter=25, nredo=1, seed=1234, spherical=False, verbose=False, **_):
        self.d, self.k = int(d), int(k)
        self.niter, self.nredo, self.seed = int(niter), int(nredo), int(seed)
        self.spherical, self.verbose = bool(spherical), bool(verbose)
        self.centroids = None
        self.index = None
        self.obj = np.zeros(0, dtype=np.float32)

    def _make_index(self, c: np.ndarray):
        index = IndexFlatIP(self.d) if self.spherical else IndexFlatL2(self.d)
        index.add(c)
        return index

    def _lloyd(self, x_dev, c0: np.ndarray, niter: int):
        import torch

        c = torch.from_numpy(c0).to(x_dev.device)
        obj = []
        n = x_dev.shape[0]
        # one assignment index for all iterations (its stream and buffers are created once): the centroids are
        # swapped in with reset() + add
        index = IndexFlat(self.d, METRIC_INNER_PRODUCT if self.spherical else METRIC_L2, x_dev.device.index)
        for _ in range(niter):
            if self.spherical:
                normalize_L2(c)
            index.reset()
            index.add_torch(c)
            D, I = index.assign_torch(x_dev) if index._assign_applies(n, 1) else index.search_torch(x_dev, 1)
            lab = I.view(-1).clamp_(min=0)
            obj.append(float(D.sum()))
            sums = torch.zeros_like(c).index_add_(0, lab, x_dev)
            cnt = torch.bincount(lab, minlength=self.k).to(c.dtype)
            empty = cnt == 0
            c = torch.where(empty[:, None], c, sums / cnt.clamp(min=1)[:, None])
            if bool(empty.any()):  # re-seed empty clusters from random rows
                g = torch.Generator(device="cpu").manual_seed(self.seed + len(obj))
                pick = torch.randint(0, n, (int(empty.sum()),), generator=g).to(x_dev.device)
                c[empty] = x_dev[pick]
        if self.spherical:
            normalize_L2(c)
        return c.cpu().numpy(), obj

    def train(self, x, init_centroids=None):
        import torch

        x = _as_rows(x, self.d)
        dev = torch.device("cuda", _default_device())
        best = None
        for redo in range(1 if init_centroids is not None else max(1, self.nredo)):
            if init_centroids is not None:
                c0 = _as_rows(init_centroids, self.d).copy()
                assert c0.shape[0] == self.k
            else:
                rs = np.random.RandomState(self.seed + redo)
                c0 = x[rs.choice(x.shape[0], self.k, replace=x.shape[0] < self.k)].copy()
            if self.niter > 0 and x.shape[0] > 0:
                c, obj = self._lloyd(torch.from_numpy(x).to(dev), c0, self.niter)
            else:
                c, obj = c0, [0.0]
                if self.spherical:
                    normalize_L2(c)
            # spherical k-means maximises the summed inner product, plain k-means minimises distance
            score = obj[-1] if self.spherical else -obj[-1]
            if best is None or score > best[0]:
                best = (score, c, obj)
        _, self.centroids, obj = best
        self.obj = np.asarray(obj, dtype=np.float32)
        self.index = self._make_index(self.centroids)
        return float(self.obj[-1])
