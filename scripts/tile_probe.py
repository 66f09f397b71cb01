"""Scan time / QPS vs queries per pass (dev aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k = 512, 10
for n in (125_000, 1_000_000):
    xb = torch.rand((n, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for nq in (1, 16, 32, 48, 64, 96, 1024):
        xq = torch.rand((nq, d), device="cuda")
        index.search_torch(xq, k)
        _, _, scan_ms, merge_ms = index.search_timed_torch(xq, k, 20)
        streams = [torch.cuda.Stream() for _ in range(2)]
        torch.cuda.synchronize()
        steps = 100 if nq <= 96 else 20
        t = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % 2]):
                index.search_torch(xq, k)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t) / steps
        print(f"n={n:8d} nq={nq:5d}  scan {scan_ms*1e3:8.1f} us  merge {merge_ms*1e3:6.1f} us  "
              f"step(2 streams) {el*1e6:8.1f} us  QPS {nq/el:10.0f}  GB/s(alg, scan) {4.0*n*d/scan_ms/1e6:7.0f}")
    del index, xb
