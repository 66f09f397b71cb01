"""Host-side cost of the sharded step (dev aid, one GPU): world of one over RCCL, shard of
125k rows = the per-rank work of the 8-GPU 1M x 512 run.  Prints the host microseconds of the
pipeline's primitives and of submit / bucket close, and the pipelined step time per depth."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss  # noqa: E402
from image_search_engine_amd.sharded import SearchPipeline, ShardedIndexFlat  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)

n, d, nq, k = int(os.environ.get("N", 125000)), 512, 16, 10
rng = np.random.default_rng(1)
xb = torch.from_numpy(rng.random((n, d), dtype=np.float32)).to(dev)
xq = torch.from_numpy(rng.random((nq, d), dtype=np.float32)).to(dev)
index = ShardedIndexFlat(d, faiss.METRIC_L2)
index.add_local(xb)


def per_call(fn, reps=3000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return t


s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ev = torch.cuda.Event()
keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
g = torch.empty((4 * nq * k,), dtype=torch.int64, device=dev)
k4 = torch.empty((4 * nq * k,), dtype=torch.int64, device=dev)
D = torch.empty((4 * nq, k), dtype=torch.float32, device=dev)
I = torch.empty((4 * nq, k), dtype=torch.int64, device=dev)
print(f"current_stream        {per_call(lambda: torch.cuda.current_stream(dev)):.2f} us")
print(f"event.record(stream)  {per_call(lambda: ev.record(s1)):.2f} us")
print(f"stream.wait_event     {per_call(lambda: s2.wait_event(ev)):.2f} us")
print(f"set_stream            {per_call(lambda: torch.cuda.set_stream(s1)):.2f} us")
torch.cuda.set_stream(torch.cuda.default_stream(dev))
h1 = s1.cuda_stream
print(f"search_keys_into      {per_call(lambda: index.backend.local_search_keys_into(xq, k, 0, keys, h1), 1000):.2f} us")
print(f"all_gather (sync op)  {per_call(lambda: dist.all_gather_into_tensor(g, k4), 1000):.2f} us")
print(f"merge_into            {per_call(lambda: index.backend.merge_into(g.view(1, 4 * nq, k), D, I, h1), 1000):.2f} us")

READY = os.environ.get("READY", "1") == "1"
variant = os.environ.get("VARIANT", "")
if variant == "nocoll":      # no collective: the merge reads the bucket's own keys
    dist.all_gather_into_tensor = lambda out, inp, group=None: None
elif variant == "copycoll":  # a plain copy on the collective stream instead of RCCL
    dist.all_gather_into_tensor = lambda out, inp, group=None: out.copy_(inp)
print("variant:", variant or "rccl", " GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for depth in (2, 4, 8):
    pipe = SearchPipeline(index, nq, k, depth=depth, buckets=int(os.environ.get("BUCKETS", 4)))
    for _ in range(100):
        pipe.submit(xq, xq_ready=READY)
    pipe.flush()
    torch.cuda.synchronize()
    steps = 4000
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(xq, xq_ready=READY)
    pipe.flush()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"depth={depth}: step {el / steps * 1e6:.1f} us (host issue loop {host / steps * 1e6:.1f} us)", flush=True)
dist.destroy_process_group()
