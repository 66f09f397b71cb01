"""Step time vs number of streams at shard sizes (dev aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k, nq = 512, 10, 16
for n in (125_000, 250_000, 500_000, 1_000_000):
    xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    row = []
    for ns in (1, 2, 3, 4, 6):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        for i in range(20):
            with torch.cuda.stream(streams[i % ns]): index.search_torch(xq, k)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(300):
            with torch.cuda.stream(streams[i % ns]):
                keys = index.search_keys_torch(xq, k, 0)
                D, I = faiss.merge_keys_torch(keys.unsqueeze(0), 1)   # stands in for the post-all-gather merge
        torch.cuda.synchronize(); row.append((time.perf_counter() - t) / 300 * 1e6)
    print(f"n={n:8d}  step us by #streams 1,2,3,4,6: " + "  ".join(f"{v:7.1f}" for v in row))
    del index, xb
