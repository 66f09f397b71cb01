"""Why does a 20-step timed region run slower per step than a 1000-step one?  Times K-step regions
(the bench loop: 16 streams round-robin) after different amounts of preceding GPU work."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss

n, d, nq, k = 1_000_000, 512, 16, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(1)
xb = torch.rand((n, d), generator=g, device=dev)
xq = torch.rand((nq, d), generator=g, device=dev)
index = faiss.IndexFlatL2(d); index.add_torch(xb); index.reserve(nq, k)
NS = int(os.environ.get("NS", "16"))
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev)) for _ in range(NS)]
hs = [s.cuda_stream for s in streams]
def run(steps):
    for i in range(steps):
        j = i % NS
        index.search_into(xq, k, outs[j][0], outs[j][1], hs[j])
def timed(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6
torch.cuda.synchronize()
print("cold: warm 5 then 20 steps: %.1f us/step" % (run(5), timed(20))[1])
for K in (20, 20, 40, 100, 20, 1000, 20, 20):
    print("K=%d: %.1f us/step" % (K, timed(K)))
time.sleep(0.5)
print("after 0.5 s idle, K=20: %.1f" % timed(20))
print("again K=20: %.1f" % timed(20))
# host issue time alone
t0 = time.perf_counter(); run(20); t1 = time.perf_counter(); torch.cuda.synchronize()
print("host issue of 20 steps: %.1f us/step" % ((t1 - t0) / 20 * 1e6))
