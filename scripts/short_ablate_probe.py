"""Short-index kernel time with phases skipped / ring depths / grid shapes (dev aid; `make -C csrc ablate`,
ISE_KNN_LIB=.../libise_knn_ablate.so).  Results are wrong by construction when a phase is skipped."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k = 512, 10
n = int(os.environ.get("N", "100000"))
nq = int(os.environ.get("NQ", "16"))
names = {0: "full", 4: "no select", 64 | 4: "stream only (no mfma/lds/select)",
         128 | 4: "mfma+lds only (no index loads)", 16 | 4: "staging only"}
xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
for shape in ("0", "1", "2"):   # 0: one 16-wave block per CU, 1: three 8-wave blocks, 2: two 8-wave blocks
    for ring in ("2", "3") if shape != "1" else ("0",):
        os.environ["ISE_SHORT_SHAPE"] = shape
        os.environ["ISE_SHORT_RING"] = ring
        for abl, nm in names.items():
            os.environ["ISE_ABLATE"] = str(abl)
            index.search_torch(xq, k)
            _, _, scan_ms, merge_ms = index.search_timed_torch(xq, k, 100)
            print(f"n={n} nq={nq} shape={shape} ring={ring} ablate={abl:3d} {nm:40s} kernel {scan_ms*1e3:7.1f} us  behind {merge_ms*1e3:5.1f} us", flush=True)
os.environ.pop("ISE_ABLATE")
