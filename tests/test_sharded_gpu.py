"""GPU tests of the sharded path with the product backend (HipShardBackend):
  * two ranks sharing the one GPU of the test box over `gloo` (RCCL refuses two ranks on one
    device): real shard scans, real packed-key merge, host logic of a world of two;
  * a world of one over RCCL ("nccl"): the collective calls, streams and events exactly as the
    8-GPU run issues them.
Both compare ShardedIndexFlat.search and the bucketed SearchPipeline with the CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, backend, q, metric=1, storage="f32", collective="auto"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        import image_search_engine_amd.faiss_compat as faiss
        from image_search_engine_amd.sharded import SearchPipeline, ShardedIndexFlat
        from oracle import flat_oracle as fo
        from tests.knn_checks import assert_knn_matches

        rng = np.random.default_rng(11)
        n, d, nq, k = 30011, 96, 16, 10
        xb = rng.random((n, d), dtype=np.float32) - np.float32(0.5 if metric == 0 else 0.0)
        xb[n - 5] = xb[7]  # duplicate rows in different shards: the lower global id wins
        if collective == "auto-broken":  # the library's communicator fails its self-test on this rank
            import image_search_engine_amd.sharded as sh

            def _fail(self, group):
                raise RuntimeError("injected self-test failure")

            sh.RcclComm.self_test = _fail
            with pytest.raises(RuntimeError, match="communicator failed"):
                ShardedIndexFlat(d, metric, storage=storage, collective="rccl")
            idx = ShardedIndexFlat(d, metric, storage=storage, collective="auto")
            assert idx.collective == "torch" and idx.comm is None  # every rank moved to the process group's all-gather
        else:
            idx = ShardedIndexFlat(d, metric, storage=storage, collective=collective)
            assert idx.collective == ("rccl" if (backend == "nccl" and collective != "torch") else "torch")
        idx.add_global(torch.from_numpy(xb).to(dev))
        lo, hi = ShardedIndexFlat.shard_bounds(n, world, rank)
        assert idx.backend.ntotal == hi - lo and idx.id_base == lo and idx.ntotal == n

        batches = [rng.random((nq, d), dtype=np.float32) - np.float32(0.5 if metric == 0 else 0.0) for _ in range(11)]
        batches[3][0] = xb[7]
        if storage == "bf16":  # the oracle sees the values the bf16 index holds (queries are rounded too)
            rnd = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()  # noqa: E731
            xb_o = rnd(xb)
        else:
            rnd = lambda a: a  # noqa: E731
            xb_o = xb
        D, I = idx.search(torch.from_numpy(batches[3]).to(dev), k)
        Dr, Ir, _ = fo.knn_flat(xb_o, rnd(batches[3]), k, metric, 4)
        assert_knn_matches(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, xb_o, rnd(batches[3]), metric)
        if metric == 1:
            assert I[0, 0].item() == 7 and I[0, 1].item() == n - 5

        # 11 batches through buckets of 4 on a ring of 2 buffer sets: two full buckets and a
        # partial one; results are copied out as they are handed back (the ring is reused)
        pipe = SearchPipeline(idx, nq, k, depth=4, buckets=2)
        dq = [torch.from_numpy(b).to(dev) for b in batches]
        got = []
        for t in dq:
            got += [(Dg.clone(), Ig.clone()) for Dg, Ig in pipe.submit(t)]
        got += [(Dg.clone(), Ig.clone()) for Dg, Ig in pipe.flush()]
        assert len(got) == len(batches)
        for b, (Dg, Ig) in zip(batches, got):
            Db, Ib, _ = fo.knn_flat(xb_o, rnd(b), k, metric, 4)
            assert_knn_matches(Dg.cpu().numpy(), Ig.cpu().numpy(), Db, Ib, xb_o, rnd(b), metric)
        # the pipeline and the one-batch form agree bit for bit
        D1, I1 = idx.search(dq[10], k)
        assert torch.equal(I1, got[10][1]) and torch.equal(D1, got[10][0])
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("backend,world,metric,storage,collective", [
    ("gloo", 2, 1, "f32", "auto"),     # two ranks share the card: torch.distributed over gloo
    ("nccl", 1, 1, "f32", "auto"),     # the library's own RCCL communicator (ise_comm_*), as the 8-GPU run uses it
    ("nccl", 1, 1, "f32", "torch"),    # the same through torch.distributed
    ("gloo", 2, 0, "f32", "auto"),     # inner product
    ("gloo", 2, 0, "bf16", "auto"),    # BASELINE config 5: bf16 rows, inner product
    ("nccl", 1, 0, "bf16", "auto"),
    ("nccl", 1, 1, "f32", "auto-broken"),  # start-up self-test fails: "rccl" raises, "auto" falls back to torch's
])
def test_sharded_hip_backend(backend, world, metric, storage, collective):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q, metric, storage, collective))
             for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in procs]
    [p.join(timeout=60) for p in procs]
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
