"""Scan time vs waves per block / blocks per CU (dev aid; ablate build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k, nq = 512, 10, 16
for n in (125_000, 1_000_000):
    xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for plan in ("8,2", "8,1", "4,1", "4,2", "4,3", "4,4"):
        os.environ["ISE_PLAN"] = plan
        index.search_torch(xq, k)
        _, _, scan_ms, merge_ms = index.search_timed_torch(xq, k, 30)
        print(f"n={n:8d} waves,blocks/CU={plan}  scan {scan_ms*1e3:7.1f} us  merge {merge_ms*1e3:5.1f} us")
    del index, xb
