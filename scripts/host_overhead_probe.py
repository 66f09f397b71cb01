"""Host-side enqueue cost per step (dev aid): time to ISSUE steps, before any synchronisation."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k, nq, n = 512, 10, 16, 1_000_000
xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
streams = [torch.cuda.Stream() for _ in range(4)]
for name, fn in (("search_torch (scan+merge)", lambda: index.search_torch(xq, k)),
                 ("keys + merge (3 kernels)", lambda: faiss.merge_keys_torch(index.search_keys_torch(xq, k, 0).unsqueeze(0), 1))):
    for i in range(20):
        with torch.cuda.stream(streams[i % 4]): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(100):   # 100 steps at ~330 us each = 33 ms of GPU work queued
        with torch.cuda.stream(streams[i % 4]): fn()
    host = (time.perf_counter() - t) / 100
    torch.cuda.synchronize()
    print(f"{name:28s} host enqueue {host*1e6:6.1f} us/step")
