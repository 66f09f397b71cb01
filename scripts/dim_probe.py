"""Scan bandwidth vs row length at a fixed 2 GB index (dev aid): the reference's real widths."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
k, nq = 10, 16
for d in (32, 64, 128, 200, 512, 1024, 2048):
    n = int(2.048e9 / (4 * d))
    xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    index.search_torch(xq, k)
    _, _, scan_ms, merge_ms = index.search_timed_torch(xq, k, 20)
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(100):
        with torch.cuda.stream(streams[i % 4]): index.search_torch(xq, k)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / 100
    print(f"d={d:5d} n={n:9d}  scan {scan_ms*1e3:7.1f} us  {4.0*n*d/scan_ms/1e6:6.0f} GB/s alg  step(4 streams) {el*1e6:7.1f} us  QPS {nq/el:8.0f}")
    del index, xb
