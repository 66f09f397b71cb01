"""Config 4 shape at reduced n: SIFT-valued rows vs 4096 unit centroids, k = 1 (dev aid / numbers)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, K = 128, 4096
n = int(os.environ.get("N", "4000000"))
cent = torch.nn.functional.normalize(torch.randn((K, d), device="cuda"), dim=1)
X = torch.randint(0, 256, (n, d), device="cuda").float()
for metric, name in ((faiss.METRIC_INNER_PRODUCT, "IP"), (faiss.METRIC_L2, "L2")):
    index = faiss.IndexFlat(d, metric); index.add_torch(cent)
    index.assign_torch(X[:100000]); torch.cuda.synchronize()
    t = time.perf_counter(); D, I = index.assign_torch(X); torch.cuda.synchronize(); el = time.perf_counter() - t
    flops = 2.0 * n * K * d
    print(f"assign {name}: n={n} K={K} d={d}  {el*1e3:8.2f} ms  {flops/el/1e12:6.1f} TFLOP/s fp32 "
          f"({flops/el/1e12/157.3*100:4.1f}% of 157.3)  {n/el/1e6:7.1f} M rows/s  -> 50M rows in {50e6/(n/el):.2f} s")
    # spot check against the general scan path
    Ds, Is = index.search_torch(X[:64], 1); torch.cuda.synchronize()
    print("   agrees with scan path on 64 rows:", bool((Is == I[:64]).all()))
