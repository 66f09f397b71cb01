"""In-kernel phase stamps of the scan kernel on a SMALL index of LONG rows, inner product (the reference's default
"cosine" index at its own size: ~1000 x 2048, one query) -- dev aid; ablate build + ISE_STAMPS."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
k = 20
for n, d, nq in ((1_000, 2048, 1), (1_000, 2048, 16), (10_000, 2048, 1)):
    xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatIP(d); index.add_torch(xb)
    for _ in range(5): index.search_torch(xq, k)
    st = torch.zeros((1024 * 8 * 16,), dtype=torch.int64, device="cuda")
    os.environ["ISE_STAMPS"] = str(st.data_ptr())
    index.search_torch(xq, k); torch.cuda.synchronize()
    os.environ.pop("ISE_STAMPS")
    s = st.cpu().numpy().reshape(1024, 8, 16).astype(np.float64)
    used = s[:, :, 0] > 0
    t0 = s[:, :, 0][used].min()
    names = ["entry", "staged", "boot in", "boot out", "loop end", "final barrier", "exit"]
    print(f"n={n} d={d} nq={nq}: blocks with stamps {int(used.any(axis=1).sum())}, waves {int(used.sum())}")
    for i, nm in enumerate(names):
        v = (s[:, :, i][used] - t0) / 100.0
        v = v[v >= 0]
        if v.size: print(f"  {nm:14s} min {v.min():8.2f}  median {np.median(v):8.2f}  max {v.max():8.2f} us")
    del index, xb
