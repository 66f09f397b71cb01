"""float32 inner-product large batches: GEMM-shaped pass against the streaming passes (ISE_NO_GEMM=1), 1M x 512."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
n, d, k = 1_000_000, 512, 10
xb = torch.nn.functional.normalize(torch.randn((n, d), device="cuda"), dim=1)
index = faiss.IndexFlatIP(d); index.add_torch(xb)
for nq in (128, 256, 512, 1024, 4096):
    xq = torch.nn.functional.normalize(torch.randn((nq, d), device="cuda"), dim=1)
    for _ in range(3): index.search_torch(xq, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps): index.search_torch(xq, k)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
    print(f"nq={nq:5d} no_gemm={os.environ.get('ISE_NO_GEMM','0')}: {el*1e3:8.3f} ms per batch = {nq/el/1e3:7.1f} k QPS = {2*nq*n*d/el/1e12:6.1f} TFLOP/s  gemm_chunks={index.exact_stats()['gemm_chunks']}", flush=True)
