"""dev build: phase stamps inside merge_kernel<true> (merge + exact re-rank) behind a 1M x 512 scan.
run: ISE_KNN_LIB=.../libise_knn_ablate.so ISE_DEBUG_STAMPS=1 python scripts/merge_stamp_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
n, d, nq, k = int(os.environ.get("N", "1000000")), 512, 16, 10
g = torch.Generator(device="cuda").manual_seed(1)
xb = torch.rand((n, d), generator=g, device="cuda"); xq = torch.rand((nq, d), generator=g, device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
for _ in range(5):
    for _ in range(20):
        index.search_torch(xq, k)
    torch.cuda.synchronize()
    index.exact_stats()   # prints the stamps of the last launch
