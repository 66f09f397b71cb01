"""Small float32 L2 batches: the direct-difference scan on its own (default for nq <= 4, k <= 32) against the
filtered path ($ISE_NO_DIRECT=1, read per call): same bits, time per batch (HIP events around 100 batches)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
n, d = int(os.environ.get("N", "1000000")), int(os.environ.get("D", "512"))
g = torch.Generator(device="cuda").manual_seed(1)
xb = torch.rand((n, d), generator=g, device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
for _ in range(200): index.search_torch(xb[:16], 10)
torch.cuda.synchronize()

def timed(xq, k, iters=100):
    for _ in range(10): index.search_torch(xq, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): D, I = index.search_torch(xq, k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, D, I

for k in [int(v) for v in os.environ.get("KS", "10").split(",")]:
    for nq in [int(v) for v in os.environ.get("NQS", "1,2,4").split(",")]:
        xq = torch.rand((nq, d), generator=g, device="cuda")
        xq[0] = xb[min(12345, n - 1)]
        os.environ["ISE_NO_DIRECT"] = "0"
        t_dir, D1, I1 = timed(xq, k)
        os.environ["ISE_NO_DIRECT"] = "1"
        t_fil, D2, I2 = timed(xq, k)
        os.environ["ISE_NO_DIRECT"] = "0"
        same = bool(torch.equal(I1, I2) and torch.equal(D1, D2))
        print(f"k={k:2d} nq={nq}: direct {t_dir:6.1f} us   filtered {t_fil:6.1f} us   same bits: {same}   self hit: {int(I1[0,0])==min(12345, n - 1) and float(D1[0,0])==0.0}", flush=True)
print(index.host_stats())
