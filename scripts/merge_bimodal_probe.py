import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import image_search_engine_amd.faiss_compat as faiss
torch.manual_seed(5)
warm = faiss.IndexFlatL2(512); warm.add_torch(torch.rand((100_000, 512), device="cuda"))
xw = torch.rand((16, 512), device="cuda")
for _ in range(20000): warm.search_torch(xw, 10)
torch.cuda.synchronize()
keep = []
for i in range(10):
    xb = torch.rand((6000, 512), device="cuda")
    index = faiss.IndexFlatL2(512); index.add_torch(xb)
    xq = torch.rand((1, 512), device="cuda")
    for _ in range(50): index.search_torch(xq, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): index.search_torch(xq, 10)
    torch.cuda.synchronize()
    print(i, f"{(time.perf_counter()-t0)/300*1e6:.1f} us  xb {xb.data_ptr():#x}", flush=True)
    keep.append((xb, index))
