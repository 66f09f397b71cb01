"""Scan-kernel times (HIP events inside the library) over index sizes and batch sizes: a regression table."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k = 512, 10
g = torch.Generator(device="cuda").manual_seed(1)
for n in (125_000, 1_000_000):
    xb = torch.rand((n, d), generator=g, device="cuda")
    for storage, metric in (("f32", 1), ("f32", 0), ("bf16", 0)):
        index = faiss.IndexFlat(d, metric, storage=storage); index.add_torch(xb)
        row = []
        for nq in (1, 16, 32, 48, 64):
            xq = torch.rand((nq, d), generator=g, device="cuda")
            for _ in range(100): index.search_torch(xq, k)
            torch.cuda.synchronize()
            t = min(index.search_timed_torch(xq, k, 50)[2] for _ in range(3)) * 1e3
            row.append(f"nq={nq}: {t:.1f}")
        print(f"n={n} {storage} metric={metric}  scan us  " + "  ".join(row), flush=True)
        del index
    del xb
