"""Scan-kernel time with phases skipped (dev aid; needs `make -C csrc ablate` and
ISE_KNN_LIB=.../libise_knn_ablate.so).  Results are wrong by construction."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k = 512, 10
nq = int(os.environ.get("NQ", "16"))
names = {0: "full", 128 | 1 | 8 | 4: "mfma+lds only, no A loads", 128 | 64 | 1 | 8 | 4: "loop skeleton only",
         128: "no A loads, rest kept", 1 | 8 | 4 | 64: "stream only (no mfma/lds)", 64: "no mfma/lds, rest kept", 1: "no query loads", 2: "no inserts", 4: "no flush/merge", 8: "no epilogue",
         2 | 4: "no inserts+flush", 8 | 4: "no epilogue+flush", 1 | 8 | 4: "stream+mfma only", 16 | 4 | 1: "launch only",
         16: "no main loop", 16 | 4: "staging only", 32: "no HBM stream (L1-hot A)", 32 | 2 | 4: "L1-hot A, no inserts/final",
         32 | 8 | 4: "L1-hot A, mfma only"}
for n in (1_000_000,):
    xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for abl, nm in names.items():
        os.environ["ISE_ABLATE"] = str(abl)
        index.search_torch(xq, k)
        _, _, scan_ms, merge_ms = index.search_timed_torch(xq, k, 30)
        print(f"n={n:8d} ablate={abl:2d} {nm:22s} scan {scan_ms*1e3:7.1f} us  merge {merge_ms*1e3:5.1f} us")
    del index, xb
