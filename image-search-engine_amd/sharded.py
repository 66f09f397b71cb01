"""Row-sharded exact kNN across the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).
Rank r owns a contiguous block of index rows; queries are replicated.  A
search is: shard-local scan -> sorted packed candidates (nq x k uint64, global
row ids in the low word) -> ONE all-gather (8*nq*k bytes per rank, latency
bound) -> k-way merge on every rank.  There is no other collective on the data
path; ``add`` is row-parallel.

``ShardedIndexFlat.search`` is the one-batch form.  A server that has several
query batches in flight uses ``SearchPipeline``: the shard scans of ``depth``
consecutive batches share ONE all-gather and one merge (a bucketed collective,
each bucket on its own HIP stream), so the exchange costs one launch per bucket
instead of one per batch -- at 8 GPUs a 1M x 512 shard scan is 40 us, less than
the host cost of issuing a collective.

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

    def local_search_keys(self, xq: torch.Tensor, k: int, id_base: int) -> torch.Tensor:
        return self.index.search_keys_torch(xq, k, id_base)

    def merge(self, keys_all: torch.Tensor):
        return self._fc.merge_keys_torch(keys_all, self.metric)

    # allocation-free forms (caller-owned buffers, work enqueued on the current stream)
    # (``stream``: a raw hipStream_t handle; default = torch's current stream)
    def local_search_keys_into(self, xq: torch.Tensor, k: int, id_base: int, keys: torch.Tensor, stream=None) -> None:
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.index.search_keys_into(xq, k, id_base, keys, stream)

    def merge_into(self, keys_all: torch.Tensor, D: torch.Tensor, I: torch.Tensor, stream=None) -> None:
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self._fc.merge_keys_into(keys_all, self.metric, D, I, stream)


class RcclComm:
    """The library's own RCCL communicator (include/ise_knn.h, ise_comm_*): the all-gather of the
    packed candidates is enqueued by one C call on the stream the shard scans run on, instead of
    going through torch.distributed (47 us of host time per collective, more than a 125k-row shard
    scan).  Built once per process group: rank 0 draws the id, one broadcast hands it round."""

    def __init__(self, group, device: torch.device):
        import ctypes

        from . import _native as _n

        self._n = _n
        self.device = device
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_gpu = dist.get_backend(group) == "nccl"
        # Local preconditions first (librccl resolves, the device is usable), agreed on across the ranks
        # BEFORE anyone enters ncclCommInitRank: a rank that cannot get there -- a failed dlopen, a bad
        # device index -- would leave the others blocked in the rendezvous for good.  Every rank raises
        # (or none does), so the caller's own agreement step stays in lockstep.
        pre_err = None
        try:
            _n.check(_n.lib.ise_comm_precheck(device.index if device.index is not None else 0))
        except Exception as e:  # noqa: BLE001 -- agreed on below
            pre_err = e
        ok = torch.tensor([0 if pre_err else 1], dtype=torch.int32, device=device if on_gpu else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise RuntimeError(f"RCCL communicator preconditions failed on some rank (this rank: {pre_err!r})")
        buf = ctypes.create_string_buffer(128)
        id_err = None
        if rank == 0:
            try:
                _n.check(_n.lib.ise_comm_unique_id(buf))
            except Exception as e:  # noqa: BLE001 -- the broadcast below must still happen: the peers wait in it
                id_err, buf = e, ctypes.create_string_buffer(128)
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if on_gpu:
            t = t.to(device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(t.cpu().numpy().tobytes())
        if id_err is not None or not any(raw):  # an all-zero id = rank 0 could not draw one
            raise RuntimeError(f"no RCCL unique id from rank 0 ({id_err!r})")
        self._h = ctypes.c_void_p()
        _n.check(_n.lib.ise_comm_create(ctypes.byref(self._h), raw, world, rank, device.index))

    def self_test(self, group) -> list:
        """One tiny all-gather checked on the host (every rank sends its own rank): a communicator that
        cannot reach its peers shows here, at start-up, not as a wrong merge later.  Returns the ranks seen."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        send = torch.full((4,), rank, dtype=torch.int64, device=self.device)
        recv = torch.full((4 * world,), -1, dtype=torch.int64, device=self.device)
        torch.cuda.synchronize(self.device)
        self.all_gather_into(recv, send, torch.cuda.current_stream(self.device).cuda_stream)
        torch.cuda.synchronize(self.device)
        want = torch.arange(world, dtype=torch.int64).repeat_interleave(4)
        if not torch.equal(recv.cpu(), want):
            raise RuntimeError(f"RCCL self-test: rank {rank} gathered {recv.cpu().tolist()}")
        return recv.cpu()[::4].tolist()

    def all_gather_into(self, gathered: torch.Tensor, keys: torch.Tensor, stream: int) -> None:
        """gathered (world * n int64, contiguous) <- every rank's keys (n int64), on ``stream``."""
        self._n.check(self._n.lib.ise_comm_allgather_keys(self._h, keys.data_ptr(), gathered.data_ptr(),
                                                          keys.numel(), stream))

    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._n.lib.ise_comm_destroy(h)
            h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedIndexFlat:
    """IndexFlat whose rows are split over the ranks of a process group.

    ``collective``: "rccl" = the library's own communicator (``RcclComm``; needs the HIP backend and
    one GPU per rank), "torch" = ``torch.distributed`` on the group's backend (gloo on CPU, or
    several ranks sharing a GPU in rehearsals), "auto" (default) = "rccl" when the group's backend
    is nccl and the shard backend is on a GPU, else "torch"."""

    def __init__(self, d: int, metric: int, group=None, backend=None, storage: str = "f32", collective: str = "auto"):
        self.d, self.metric_type, self.group = int(d), int(metric), group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend if backend is not None else HipShardBackend(d, metric, storage=storage)
        self.id_base = 0
        self._counts = [0] * self.world
        dev = getattr(self.backend, "device", torch.device("cpu"))
        auto = collective == "auto"
        if auto:
            collective = "rccl" if (dev.type == "cuda" and dist.get_backend(group) == "nccl") else "torch"
        if collective not in ("rccl", "torch"):
            raise ValueError("collective must be 'auto', 'rccl' or 'torch'")
        self.comm = None
        self.ranks_seen = None  # ranks the library's communicator reached in its start-up self-test (collective "rccl")
        # every data-path collective this rank has issued, in order: (kind, elements per rank).  All-gathers pair up
        # across ranks by ORDER (several may be in flight on different streams over one communicator), so every
        # rank must issue the same sequence: ``check_collective_order`` compares the logs after a run
        self.collective_log: list = []
        if collective == "rccl":
            # every rank reports whether its communicator came up and passed the self-test; "auto"
            # then moves ALL ranks to the process group's own all-gather (still RCCL, issued by
            # torch.distributed) if any of them failed; an explicit "rccl" raises instead
            err = None
            try:
                self.comm = RcclComm(group, dev)
                self.ranks_seen = self.comm.self_test(group)
            except Exception as e:  # noqa: BLE001 -- reported below, on every rank
                err = e
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                if self.comm is not None:
                    self.comm.close()
                    self.comm = None
                if not auto:
                    raise RuntimeError(f"collective='rccl': the library's communicator failed on some rank"
                                       f" (this rank: {err!r})")
                import sys

                print(f"[image_search_engine_amd] rank {self.rank}: library RCCL communicator unavailable ({err!r});"
                      " using torch.distributed's all-gather", file=sys.stderr, flush=True)
                collective = "torch"
        self.collective = collective

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
        # no exchange on the build side: every shard's search is exact on its own (float32 L2 shards
        # re-rank by the direct difference, csrc/ise_exact.hpp), so shards share nothing but the id space
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
        self.collective_log.append(("search", int(keys.numel())))
        if self.comm is not None:  # stream-ordered behind the scan, nothing to wait for on the host
            self.comm.all_gather_into(gathered.view(-1), keys.view(-1),
                                      torch.cuda.current_stream(keys.device).cuda_stream)
            return None, gathered, bufs
        work = dist.all_gather_into_tensor(gathered.view(-1), keys.view(-1), group=self.group, async_op=True)
        return work, gathered, bufs

    def search_end(self, ticket):
        work, gathered, bufs = ticket
        if work is not None:
            work.wait()
        if bufs is not None and hasattr(self.backend, "merge_into"):
            self.backend.merge_into(gathered, bufs["D"], bufs["I"])
            return bufs["D"], bufs["I"]
        return self.backend.merge(gathered)

    def search(self, xq: torch.Tensor, k: int):
        """(D, I) identical on every rank and identical to an unsharded IndexFlat."""
        return self.search_end(self.search_begin(xq, k))

    def check_collective_order(self) -> dict:
        """Collective (call it on every rank, outside any timed region, with nothing in flight): every rank's
        log of issued all-gathers must be the same sequence -- the invariant the exchange relies on.  Returns
        ``{"collectives": n, "ranks_agree": True}`` or raises ``RuntimeError`` naming the first difference.
        The comparison itself goes through ``torch.distributed`` (one all-gather of a digest), never through
        the communicator under test."""
        import hashlib

        blob = repr(self.collective_log).encode()
        mine = torch.tensor([len(self.collective_log)] + list(hashlib.sha256(blob).digest()[:15]), dtype=torch.int64)
        dev = getattr(self.backend, "device", torch.device("cpu"))
        on_gpu = dev.type == "cuda" and dist.get_backend(self.group) == "nccl"
        if on_gpu:
            mine = mine.to(dev)
        allv = torch.empty(self.world * mine.numel(), dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(allv, mine, group=self.group)
        rows = allv.view(self.world, -1).cpu().tolist()
        for r, row in enumerate(rows):
            if row != rows[0]:
                raise RuntimeError(f"collective order differs between rank 0 ({rows[0][0]} collectives) and rank {r} "
                                   f"({row[0]} collectives); this rank's last entries: {self.collective_log[-4:]}")
        return {"collectives": len(self.collective_log), "ranks_agree": True}


class SearchPipeline:
    """Batches in flight over a ``ShardedIndexFlat``, ``depth`` batches per all-gather.

    ``submit(xq)`` enqueues the shard-local scan of one batch of exactly ``nq`` queries; every
    ``depth``-th submit closes the bucket: ONE all-gather of the bucket's packed candidates
    ([world][depth*nq][k] keys) and one merge.  A bucket lives on one HIP stream from its first scan
    to its merge, and ``buckets`` buckets (a ring of buffer sets, one stream each) are in flight, so
    the exchange of one bucket overlaps the scans of the next ones without any cross-stream event
    on the per-batch path.  Results come back one bucket late (so that waiting for them never
    stalls the scans being issued) as a list of ``(D, I)`` per batch, in submit order; ``flush()``
    closes a partial bucket and returns everything outstanding.  Returned tensors are views into
    the ring and stay valid until ``buckets - 1`` further buckets have been closed; they are
    ordered after their producer on the caller's current stream.

    ``submit(xq, xq_ready=True)`` skips ordering the scan after the caller's current stream: for
    query tensors that are already materialised (e.g. resident inputs, or produced before a
    synchronize).

    Every rank must submit the same sequence of batches (the all-gathers pair up by order).
    On a CPU process group (the gloo tests) the same code runs synchronously without streams.
    """

    def __init__(self, index: "ShardedIndexFlat", nq: int, k: int, depth: int = 4, buckets: int = 4):
        if depth < 1 or buckets < 2:
            raise ValueError("SearchPipeline needs depth >= 1 and buckets >= 2")
        self.index, self.nq, self.k, self.depth, self.nbuckets = index, int(nq), int(k), int(depth), int(buckets)
        self.dev = getattr(index.backend, "device", torch.device("cpu"))
        self.cuda = self.dev.type == "cuda"
        self.into = hasattr(index.backend, "local_search_keys_into")
        world = index.world
        self.sets = []
        per = self.nq * self.k
        for _ in range(self.nbuckets):
            keys = torch.empty((self.depth, self.nq, self.k), dtype=torch.int64, device=self.dev)
            gathered = torch.empty(world * self.depth * per, dtype=torch.int64, device=self.dev)
            D = torch.empty((self.depth * self.nq, self.k), dtype=torch.float32, device=self.dev)
            I = torch.empty((self.depth * self.nq, self.k), dtype=torch.int64, device=self.dev)
            # every view the hot loop needs is made once here (a tensor view costs microseconds)
            b = {"keys": keys, "gathered": gathered, "D": D, "I": I, "filled": 0, "merged": 0,
                 "keys_g": [keys[g] for g in range(self.depth)],
                 "full": (keys.view(-1), gathered, gathered.view(world, self.depth * self.nq, self.k), D, I),
                 "out": [(D[g * self.nq:(g + 1) * self.nq], I[g * self.nq:(g + 1) * self.nq])
                         for g in range(self.depth)]}
            if self.cuda:
                b["stream"] = torch.cuda.Stream(device=self.dev)
                b["handle"] = b["stream"].cuda_stream
                b["ready"] = torch.cuda.Event()
                b["done"] = torch.cuda.Event()
            self.sets.append(b)
        self.cur = 0          # set being filled
        self.undelivered = []  # closed buckets whose results have not been handed back, oldest first

    def submit(self, xq: torch.Tensor, xq_ready: bool = False):
        if xq.shape[0] != self.nq or xq.shape[1] != self.index.d:
            raise ValueError(f"SearchPipeline was built for batches of shape ({self.nq}, {self.index.d})")
        b = self.sets[self.cur]
        g = b["filled"]
        be = self.index.backend
        if self.cuda:
            if not xq_ready:  # order the scan after whatever produced xq
                b["ready"].record(torch.cuda.current_stream(self.dev))
                b["stream"].wait_event(b["ready"])
            be.local_search_keys_into(xq, self.k, self.index.id_base, b["keys_g"][g], b["handle"])
            # the scan reads xq on the bucket's stream: the caching allocator must not hand xq's memory
            # to a later allocation on the caller's stream while that read is pending
            xq.record_stream(b["stream"])
        elif self.into:
            be.local_search_keys_into(xq, self.k, self.index.id_base, b["keys_g"][g])
        else:
            b["keys_g"][g].copy_(be.local_search_keys(xq, self.k, self.index.id_base))
        b["filled"] = g + 1
        if b["filled"] == self.depth:
            return self._close()
        return []

    def flush(self):
        out = self._close() if self.sets[self.cur]["filled"] else []
        return out + self._deliver(keep=0)

    # -- one bucket: all-gather + merge, on the bucket's stream behind its scans
    def _close(self):
        b = self.sets[self.cur]
        m = b["filled"]
        world, per = self.index.world, self.nq * self.k
        if m == self.depth:
            keys, gathered, g3, D, I = b["full"]
        else:  # partial bucket (flush): prefixes of the same buffers
            keys = b["keys"].view(-1)[: m * per]
            gathered = b["gathered"][: world * m * per]
            g3 = gathered.view(world, m * self.nq, self.k)
            D, I = b["D"][: m * self.nq], b["I"][: m * self.nq]
        be = self.index.backend
        self.index.collective_log.append(("bucket", self.cur, m, int(keys.numel())))
        if self.cuda and getattr(self.index, "comm", None) is not None:  # one C call: ncclAllGather on the bucket's stream
            self.index.comm.all_gather_into(gathered, keys, b["handle"])
            be.merge_into(g3, D, I, b["handle"])
            b["done"].record(b["stream"])
        elif self.cuda:
            prev = torch.cuda.current_stream(self.dev)
            torch.cuda.set_stream(b["stream"])  # the process group orders the collective after this stream
            try:
                dist.all_gather_into_tensor(gathered, keys, group=self.index.group)
                be.merge_into(g3, D, I, b["handle"])
            finally:
                torch.cuda.set_stream(prev)
            b["done"].record(b["stream"])
        else:
            dist.all_gather_into_tensor(gathered, keys, group=self.index.group)
            if hasattr(be, "merge_into"):
                be.merge_into(g3, D, I)
            else:
                Dm, Im = be.merge(g3)
                D.copy_(Dm)
                I.copy_(Im)
        b["merged"], b["filled"] = m, 0
        self.undelivered.append(self.cur)
        self.cur = (self.cur + 1) % self.nbuckets
        return self._deliver(keep=1)

    def _deliver(self, keep: int):
        out = []
        while len(self.undelivered) > keep:
            b = self.sets[self.undelivered.pop(0)]
            if self.cuda:
                torch.cuda.current_stream(self.dev).wait_event(b["done"])
            out += b["out"][: b["merged"]]
        return out
