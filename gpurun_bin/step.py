import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import image_search_engine_amd.faiss_compat as faiss
torch.manual_seed(5)
for rows in (5000, 6000, 6500, 7000, 8000):
    xb = torch.rand((rows, 512), device="cuda")
    index = faiss.IndexFlatL2(512); index.add_torch(xb)
    xq = torch.rand((1, 512), device="cuda")
    for _ in range(20): index.search_torch(xq, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): index.search_torch(xq, 10)
    torch.cuda.synchronize()
    print(rows, f"{(time.perf_counter()-t0)/300*1e6:.1f} us", index.exact_stats(), index.short_stats(), flush=True)
