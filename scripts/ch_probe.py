"""Scan time vs chunk size for nq=16 (dev aid; ablate build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k, nq = 512, 10, 16
for n in (125_000, 1_000_000):
    xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for ch in ("8", "4", "2"):
        for plan in ("8,2", "8,1"):
            os.environ["ISE_CH"] = ch; os.environ["ISE_PLAN"] = plan
            index.search_torch(xq, k)
            res = [index.search_timed_torch(xq, k, 30)[2] for _ in range(3)]
            print(f"n={n:8d} CH={ch} plan={plan}  scan {min(res)*1e3:7.1f} us (runs: {' '.join('%.1f' % (r*1e3) for r in res)})")
    del index, xb
