"""A/B of the exact path's spare candidate slots (ISE_EXACT_EXTRA, read once per process): scan and
post-scan times at 1M x 512, nq = 16."""
import os, sys, subprocess, json
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import image_search_engine_amd.faiss_compat as faiss
    n, d, nq, k = int(os.environ.get("N", "1000000")), 512, 16, 10
    g = torch.Generator(device="cuda").manual_seed(1)
    xb = torch.rand((n, d), generator=g, device="cuda"); xq = torch.rand((nq, d), generator=g, device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for _ in range(300): index.search_torch(xq, k)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        _, _, a, b = index.search_timed_torch(xq, k, 100)
        out.append((round(a * 1e3, 1), round(b * 1e3, 1)))
    print(os.environ.get("ISE_EXACT_EXTRA"), out, index.exact_stats())
else:
    for extra in ("6", "4", "2", "6", "4"):
        env = dict(os.environ, ISE_EXACT_EXTRA=extra)
        subprocess.run([sys.executable, __file__, "child"], env=env)
