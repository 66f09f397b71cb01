"""Details of an id mismatch between the GPU search and the float32 C oracle on bench data."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
from bench import make_inputs
from oracle import flat_oracle as fo, knn_oracle as ko
n = int(os.environ.get("N", "125000")); d, nq, k = 512, 16, 10
xb, xq = make_inputs(n, d, nq, 0, n)
index = faiss.IndexFlatL2(d); index.add(xb)
D, I = index.search(xq, k)
Dc, Ic, _ = fo.knn_flat(xb, xq, k, 1, 16)
D64, I64 = ko.knn_exact(xb, xq, k, 1)
print("gpu==c_oracle", np.array_equal(I, Ic), "gpu==f64", np.array_equal(I, I64), "c==f64", np.array_equal(Ic, I64))
for q, r in np.argwhere(I != Ic):
    ids = sorted(set(I[q]) ^ set(Ic[q]))
    tr = lambda i: float(((xb[i].astype(np.float64) - xq[q].astype(np.float64)) ** 2).sum())
    print("q", q, "rank", r, "gpu", I[q, r], D[q, r], "c", Ic[q, r], Dc[q, r], "f64:", I64[q, r], "true", tr(I[q, r]), tr(Ic[q, r]), "symdiff", ids)
print(index.exact_stats())
