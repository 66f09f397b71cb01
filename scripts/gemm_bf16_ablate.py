"""dev build: what bounds the bf16 GEMM-shaped pass?  ISE_GEMM_ABLATE bits: 1 no epilogue, 2 no query staging, 4 no MFMA."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
n, d, k, nq = 1_000_000, 512, 10, 1024
g = torch.Generator(device="cuda").manual_seed(1)
xb = torch.rand((n, d), generator=g, device="cuda"); xq = torch.rand((nq, d), generator=g, device="cuda")
index = faiss.IndexFlat(d, 0, storage="bf16"); index.add_torch(xb)
for _ in range(3): index.search_torch(xq, k)
for abl in (0, 1, 2, 3, 4, 5, 6, 7, 0):
    os.environ["ISE_GEMM_ABLATE"] = str(abl)
    for _ in range(2): index.search_torch(xq, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): index.search_torch(xq, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"ablate {abl} ({'no-epilogue ' if abl & 1 else ''}{'no-staging ' if abl & 2 else ''}{'no-mfma' if abl & 4 else ''}): {dt*1e3:.3f} ms per 1024 queries")
