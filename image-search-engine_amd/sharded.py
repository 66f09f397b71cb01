"""Row-sharded exact kNN across the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).
Rank r owns a contiguous block of index rows; queries are replicated.  A
search is: shard-local scan -> sorted packed candidates (nq x k uint64, global
row ids in the low word) -> ONE all-gather (8*nq*k bytes per rank, latency
bound) -> k-way merge on every rank.  There is no other collective on the data
path; ``add`` is row-parallel.

The reference itself never shards (single in-RAM IndexFlat, backend/utils.py:327);
what it pins is only the result: identical to one unsharded index.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class HipShardBackend:
    """Shard-local compute on the MI355X through include/ise_knn.h."""

    def __init__(self, d: int, metric: int, device: int | None = None, storage: str = "f32"):
        from . import faiss_compat as fc

        self._fc = fc
        self.index = fc.IndexFlat(d, metric, device, storage)
        self.metric = metric
        self.device = torch.device("cuda", self.index.device)

    @property
    def ntotal(self) -> int:
        return self.index.ntotal

    def add(self, x) -> None:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            self.index.add_torch(x)
        else:
            self.index.add(x.numpy() if isinstance(x, torch.Tensor) else x)

    # the L2 shift vector (include/ise_knn.h): shard 0 defines it, the others adopt it
    def get_shift(self) -> torch.Tensor:
        return torch.from_numpy(self.index.get_shift())

    def set_shift(self, mu: torch.Tensor) -> None:
        self.index.set_shift(mu.cpu().numpy())

    def local_search_keys(self, xq: torch.Tensor, k: int, id_base: int) -> torch.Tensor:
        return self.index.search_keys_torch(xq, k, id_base)

    def merge(self, keys_all: torch.Tensor):
        return self._fc.merge_keys_torch(keys_all, self.metric)

    # allocation-free forms (caller-owned buffers, work enqueued on the current stream)
    def local_search_keys_into(self, xq: torch.Tensor, k: int, id_base: int, keys: torch.Tensor) -> None:
        self.index.search_keys_into(xq, k, id_base, keys, torch.cuda.current_stream(self.device).cuda_stream)

    def merge_into(self, keys_all: torch.Tensor, D: torch.Tensor, I: torch.Tensor) -> None:
        self._fc.merge_keys_into(keys_all, self.metric, D, I, torch.cuda.current_stream(self.device).cuda_stream)


class ShardedIndexFlat:
    """IndexFlat whose rows are split over the ranks of a process group."""

    def __init__(self, d: int, metric: int, group=None, backend=None, storage: str = "f32"):
        self.d, self.metric_type, self.group = int(d), int(metric), group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend if backend is not None else HipShardBackend(d, metric, storage=storage)
        self.id_base = 0
        self._counts = [0] * self.world

    @staticmethod
    def shard_bounds(n: int, world: int, rank: int):
        """Contiguous row block of rank r: [r*n//G, (r+1)*n//G)."""
        return n * rank // world, n * (rank + 1) // world

    @property
    def ntotal(self) -> int:
        return int(sum(self._counts))

    def add_global(self, x) -> None:
        """Every rank passes the same (N, d) rows; each keeps only its block."""
        if self.ntotal:
            raise RuntimeError("add_global on a non-empty sharded index would interleave id ranges")
        lo, hi = self.shard_bounds(x.shape[0], self.world, self.rank)
        self.add_local(x[lo:hi])

    def add_local(self, x_local) -> None:
        """Append this rank's rows; ids follow rank order (one tiny all-gather of counts)."""
        if self.ntotal:
            raise RuntimeError("sharded index is append-once: ids must stay contiguous per rank")
        dev = getattr(self.backend, "device", torch.device("cpu"))
        if hasattr(self.backend, "get_shift"):
            # every shard must measure distances around the same shift vector as an unsharded
            # index would: rank 0 (which holds the first rows) fixes it by adding first
            mu = torch.zeros(self.d, dtype=torch.float32, device=dev)
            if self.rank == 0:
                self.backend.add(x_local)
                mu.copy_(self.backend.get_shift())
            dist.broadcast(mu, src=0, group=self.group)
            if self.rank != 0:
                self.backend.set_shift(mu)
                self.backend.add(x_local)
        else:
            self.backend.add(x_local)
        mine = torch.tensor([x_local.shape[0]], dtype=torch.int64, device=dev)
        allc = torch.empty(self.world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        self._counts = [int(c) for c in allc.cpu().tolist()]
        self.id_base = int(sum(self._counts[: self.rank]))
        if self.ntotal >= 2 ** 32:
            raise RuntimeError("a sharded index holds fewer than 2^32 rows")

    # -- query side
    def make_buffers(self, nq: int, k: int):
        """Caller-owned buffers for the allocation-free ``search_begin/search_end`` form: one set per
        batch in flight."""
        dev = getattr(self.backend, "device", torch.device("cpu"))
        return {"keys": torch.empty((nq, k), dtype=torch.int64, device=dev),
                "gathered": torch.empty((self.world, nq, k), dtype=torch.int64, device=dev),
                "D": torch.empty((nq, k), dtype=torch.float32, device=dev),
                "I": torch.empty((nq, k), dtype=torch.int64, device=dev)}

    def search_begin(self, xq: torch.Tensor, k: int, bufs=None):
        """Enqueue the shard-local scan and start the all-gather; returns a ticket for
        ``search_end``.  Issuing the next ``search_begin`` before ``search_end`` overlaps
        the collective of batch i with the scan of batch i+1.  With ``bufs`` (``make_buffers``)
        nothing is allocated on the way."""
        if bufs is not None and hasattr(self.backend, "local_search_keys_into"):
            keys, gathered = bufs["keys"], bufs["gathered"]
            self.backend.local_search_keys_into(xq, k, self.id_base, keys)
        else:
            keys = self.backend.local_search_keys(xq, k, self.id_base)
            gathered = torch.empty((self.world,) + tuple(keys.shape), dtype=keys.dtype, device=keys.device)
        work = dist.all_gather_into_tensor(gathered.view(-1), keys.view(-1), group=self.group, async_op=True)
        return work, gathered, bufs

    def search_end(self, ticket):
        work, gathered, bufs = ticket
        work.wait()
        if bufs is not None and hasattr(self.backend, "merge_into"):
            self.backend.merge_into(gathered, bufs["D"], bufs["I"])
            return bufs["D"], bufs["I"]
        return self.backend.merge(gathered)

    def search(self, xq: torch.Tensor, k: int):
        """(D, I) identical on every rank and identical to an unsharded IndexFlat."""
        return self.search_end(self.search_begin(xq, k))
