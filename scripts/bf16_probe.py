"""bf16-storage scan: time, QPS and recall vs the fp32 index (dev aid / config 5 numbers)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k = 512, 10
n = int(os.environ.get("N", "1000000"))
xb = torch.nn.functional.normalize(torch.randn((n, d), device="cuda"), dim=1)
exact = faiss.IndexFlatIP(d); exact.add_torch(xb)
approx = faiss.IndexFlatIP(d, storage="bf16"); approx.add_torch(xb)
for nq in (16, 32, 48, 64, 96, 128, 1024):
    xq = torch.nn.functional.normalize(torch.randn((nq, d), device="cuda"), dim=1)
    D0, I0 = exact.search_torch(xq, k)
    D1, I1 = approx.search_torch(xq, k)
    torch.cuda.synchronize()
    rec = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(I0.cpu(), I1.cpu())) / (nq * k)
    _, _, scan_ms, merge_ms = approx.search_timed_torch(xq, k, 20)
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize(); steps = 100 if nq <= 96 else 20
    t = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % 2]):
            approx.search_torch(xq, k)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / steps
    print(f"bf16 n={n} nq={nq:5d}  scan {scan_ms*1e3:8.1f} us  step {el*1e6:8.1f} us  QPS {nq/el:10.0f}  "
          f"GB/s(alg) {2.0*n*d/scan_ms/1e6:6.0f}  recall@10 vs fp32 exact {rec:.4f}  max|dD| {float((D0-D1).abs().max()):.2e}")
