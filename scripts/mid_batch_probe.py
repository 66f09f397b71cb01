"""Batches of 32 / 48 / 64 queries (the shared passes of concurrent one-query callers) against small and
mid-size indexes of short rows: per-batch time on one stream, seeded inputs, result checksums."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
k = 10
torch.manual_seed(5)
for rows, d in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((1_000, 512), (10_000, 512), (30_000, 512), (100_000, 512), (10_000, 128), (100_000, 128)):
    xb = torch.rand((rows, d), device="cuda")
    for name, cls in (("L2", faiss.IndexFlatL2), ("IP", faiss.IndexFlatIP)):
        index = cls(d); index.add_torch(xb)
        for nq in (32, 48, 64):
            xq = torch.rand((nq, d), device="cuda")
            for _ in range(20): index.search_torch(xq, k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 300
            for _ in range(reps): out = index.search_torch(xq, k)
            torch.cuda.synchronize()
            print(f"{rows}x{d} {name} nq={nq}: {(time.perf_counter() - t0) / reps * 1e6:7.1f} us  {int(out[1].sum())} {float(out[0].double().sum()):.6f}", flush=True)
