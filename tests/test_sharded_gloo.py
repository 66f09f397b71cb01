"""CPU tests of the N > 1 path: world_size-2 `gloo` process groups drive the real
ShardedIndexFlat host logic (row partition, id bases, packed-key all-gather,
begin/end pipelining).  The shard-local compute is a TEST DOUBLE built on the
oracle (the product backend is HIP-only and needs a GPU); the packed-key format
it speaks is the one include/ise_knn.h defines."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardBackend:
    """Test double for HipShardBackend: same interface, oracle arithmetic."""

    def __init__(self, d, metric, storage="f32"):
        from oracle import knn_oracle as ko

        self.ko, self.d, self.metric = ko, d, metric
        self.xb = np.zeros((0, d), np.float32)
        self.device = torch.device("cpu")
        # bf16 storage (BASELINE config 5): rows and queries are rounded to bf16, candidates are
        # float32 scores of the rounded values
        self.rnd = (lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16)
                    .to(torch.float32).numpy()) if storage == "bf16" else (lambda a: a)

    @property
    def ntotal(self):
        return self.xb.shape[0]

    def add(self, x):
        x = x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        self.xb = np.concatenate([self.xb, self.rnd(x.astype(np.float32))])

    def local_search_keys(self, xq, k, id_base):
        from tests import keycodec as kc

        D, I = self.ko.knn_exact(self.xb, self.rnd(xq.numpy()), k, self.metric, id_offset=id_base)
        return torch.from_numpy(kc.encode(D, I, self.metric).view(np.int64))

    def merge(self, keys_all):
        from tests import keycodec as kc

        g, nq, k = keys_all.shape
        keys = keys_all.numpy().view(np.uint64).transpose(1, 0, 2).reshape(nq, g * k)
        D, I = kc.decode(np.sort(keys, axis=1)[:, :k], self.metric)
        return torch.from_numpy(D), torch.from_numpy(I)


def _worker(rank, world, port, metric, n, q, storage="f32"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_search_engine_amd.sharded import ShardedIndexFlat
        from oracle import knn_oracle as ko

        rng = np.random.default_rng(7)
        d, k = 24, 6
        xb = rng.random((n, d), dtype=np.float32)
        if n > 40:
            xb[n - 3] = xb[2]  # duplicate across shards: tie must go to the lower global id
        xq = np.concatenate([rng.random((4, d), dtype=np.float32), xb[2:3]]) if n > 2 else \
            rng.random((3, d), dtype=np.float32)
        idx = ShardedIndexFlat(d, metric, backend=OracleShardBackend(d, metric, storage))
        assert idx.collective == "torch" and idx.comm is None  # a CPU group never takes the RCCL path
        idx.add_global(xb)
        if storage == "bf16":  # what the answers are compared with: the oracle on the rounded values
            xb, xq = idx.backend.rnd(xb), idx.backend.rnd(xq)
        lo, hi = ShardedIndexFlat.shard_bounds(n, world, rank)
        assert idx.backend.ntotal == hi - lo and idx.id_base == lo and idx.ntotal == n
        tq = torch.from_numpy(xq)
        D, I = idx.search(tq, k)
        Dr, Ir = ko.knn_exact(xb, xq, k, metric)
        assert np.array_equal(I.numpy(), Ir), (rank, I, Ir)
        assert np.array_equal(D.numpy(), Dr)
        # pipelined form: two batches in flight, results unchanged and in order
        t1 = idx.search_begin(tq, k)
        t2 = idx.search_begin(tq[:2], k)
        D1, I1 = idx.search_end(t1)
        D2, I2 = idx.search_end(t2)
        assert np.array_equal(I1.numpy(), Ir) and np.array_equal(I2.numpy(), Ir[:2])
        # bucketed pipeline: 3 batches per all-gather, 2 buffer sets, 8 distinct batches
        # (two full buckets + a partial one closed by flush); results in submit order
        from image_search_engine_amd.sharded import SearchPipeline

        nqp = xq.shape[0]
        batches = [np.ascontiguousarray(np.roll(xq, s, axis=0) * np.float32(1 + 0.125 * (s % 3))) for s in range(8)]
        pipe = SearchPipeline(idx, nqp, k, depth=3, buckets=2)
        got, when = [], []
        for i, bq in enumerate(batches):
            out = pipe.submit(torch.from_numpy(bq))
            got += [(Dg.clone(), Ig.clone()) for Dg, Ig in out]
            when.append(len(got))
        assert when == [0, 0, 0, 0, 0, 3, 3, 3], when  # a bucket is handed back when the next one closes
        got += [(Dg.clone(), Ig.clone()) for Dg, Ig in pipe.flush()]
        assert len(got) == len(batches)
        for bq, (Dg, Ig) in zip(batches, got):
            Db, Ib = ko.knn_exact(xb, idx.backend.rnd(bq), k, metric)
            assert np.array_equal(Ig.numpy(), Ib) and np.array_equal(Dg.numpy(), Db)
        assert pipe.flush() == []
        with pytest.raises(ValueError):
            pipe.submit(tq[:1])
        # the invariant the exchange relies on: every rank issued the same sequence of all-gathers
        # (3 one-batch searches + 3 buckets, the last one partial)
        rep = idx.check_collective_order()
        assert rep == {"collectives": 6, "ranks_agree": True}, rep
        assert idx.collective_log[3:] == [("bucket", 0, 3, 3 * nqp * k), ("bucket", 1, 3, 3 * nqp * k),
                                          ("bucket", 0, 2, 2 * nqp * k)], idx.collective_log
        if world > 1:
            if rank == 1:   # a rank that issued one more is found out (the check is collective: every rank runs it)
                idx.collective_log.append(("search", 1))
            try:
                idx.check_collective_order()
                raise AssertionError("a diverging collective log went unnoticed")
            except RuntimeError as e:
                assert "collective order differs" in str(e)
            if rank == 1:
                idx.collective_log.pop()
        # every rank holds the same answer
        allI = [torch.empty_like(I) for _ in range(world)]
        dist.all_gather(allI, I)
        assert all(torch.equal(allI[0], t) for t in allI)
        with pytest.raises(RuntimeError):
            idx.add_global(xb)  # append-once: ids stay contiguous per rank
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("metric,n,storage", [(1, 1001, "f32"), (1, 3, "f32"), (0, 1001, "f32"), (0, 3, "f32"),
                                              (0, 1001, "bf16")])
def test_world2_gloo_sharded_search_equals_unsharded(metric, n, storage):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, metric, n, q, storage)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_world1_gloo_pipeline_depth1():
    """A world of one, and a pipeline that closes a bucket on every batch (depth 1): the same
    worker, so the same checks -- plus the constructor's argument checks."""
    from image_search_engine_amd.sharded import SearchPipeline

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), 1, 1001, q))
    p.start()
    rank, msg = q.get(timeout=120)
    p.join(timeout=60)
    assert msg == "ok", msg

    class _Idx:  # enough of an index for the constructor
        world, d, backend = 1, 4, type("B", (), {"device": torch.device("cpu")})()

    with pytest.raises(ValueError):
        SearchPipeline(_Idx(), 4, 2, depth=0)
    with pytest.raises(ValueError):
        SearchPipeline(_Idx(), 4, 2, depth=2, buckets=1)
    pipe = SearchPipeline(_Idx(), 4, 2, depth=1, buckets=2)
    assert pipe.flush() == []
