import os, sys
import torch
sys.path.insert(0, "/root/repo")
import image_search_engine_amd.faiss_compat as faiss
n, d, nq, k = 1_000_000, 512, 16, 10
g = torch.Generator(device="cuda").manual_seed(1)
xb = torch.rand((n, d), generator=g, device="cuda"); xq = torch.rand((nq, d), generator=g, device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
for _ in range(300): index.search_torch(xq, k)
torch.cuda.synchronize()
for rnd in range(3):
    for nm, abl in (("full", 0), ("no shift sub / mus reads", 1024), ("no lds reads at all (B stale)", 512 | 1024)):
        os.environ["ISE_ABLATE"] = str(abl)
        for _ in range(5): index.search_torch(xq, k)
        torch.cuda.synchronize()
        _, _, a, b = index.search_timed_torch(xq, k, 60)
        print(f"round {rnd} {nm:32s}: scan {a*1e3:.1f} us", flush=True)
