"""dev build: block shapes of the one-tile scan kernel (ISE_PLAN=waves,blocks_per_cu; read per plan) at 1M x 512."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
n, d, nq, k = int(os.environ.get("N", "1000000")), 512, 16, 10
g = torch.Generator(device="cuda").manual_seed(1)
xb = torch.rand((n, d), generator=g, device="cuda"); xq = torch.rand((nq, d), generator=g, device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
for _ in range(300): index.search_torch(xq, k)
torch.cuda.synchronize()
for rnd in range(2):
    for plan in os.environ.get("PLANS", "8,2;4,3;4,4").split(";"):
        os.environ["ISE_PLAN"] = plan
        index.reserve(nq, k)
        for _ in range(20): index.search_torch(xq, k)
        torch.cuda.synchronize()
        _, _, a, b = index.search_timed_torch(xq, k, 100)
        print(f"round {rnd} plan {plan}: scan {a*1e3:.1f} us, rest {b*1e3:.1f} us")
