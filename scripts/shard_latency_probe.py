"""Per-step cost of one shard-sized search (dev aid): eager vs hipGraph replay."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss

d, k = 512, 10
for n, nq in ((125_000, 16), (250_000, 16), (500_000, 16), (1_000_000, 16), (125_000, 64), (1_000_000, 64)):
    xb = torch.rand((n, d), device="cuda")
    xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for _ in range(20): index.search_torch(xq, k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200): index.search_torch(xq, k)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 200
    t = time.perf_counter()
    for _ in range(200): keys = index.search_keys_torch(xq, k, 0); D, I = faiss.merge_keys_torch(keys.unsqueeze(0), 1)
    torch.cuda.synchronize(); eager2 = (time.perf_counter() - t) / 200
    _, _, scan_ms, merge_ms = index.search_timed_torch(xq, k, 50)
    # graph
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        index.search_torch(xq, k)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            D, I = index.search_torch(xq, k)
    torch.cuda.synchronize()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t) / 200
    print(f"n={n:8d} nq={nq:3d}  scan {scan_ms*1e3:7.1f} us  merge {merge_ms*1e3:6.1f} us | step eager {eager*1e6:7.1f} us"
          f"  keys+merge eager {eager2*1e6:7.1f} us  graph {graph*1e6:7.1f} us")
    del index, xb
