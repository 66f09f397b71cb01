"""Bandwidth of the build-side helpers: normalize_L2 (device), add (pad/copy + shift + norms)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
n, d = 1_000_000, 512
x = torch.randn((n, d), device="cuda")
faiss.normalize_L2(x); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): faiss.normalize_L2(x)
torch.cuda.synchronize(); el = (time.perf_counter() - t) / 10
print(f"normalize_L2 device {n}x{d}: {el*1e3:.3f} ms  {8.0*n*d/el/1e9:.0f} GB/s (read+write)")
assert float((x.norm(dim=1) - 1).abs().max()) < 1e-5
for storage in ("f32", "bf16"):
    idx = faiss.IndexFlatL2(d, storage=storage)
    torch.cuda.synchronize(); t = time.perf_counter()
    idx.add_torch(x); torch.cuda.synchronize(); el = time.perf_counter() - t
    print(f"add_torch {storage} {n}x{d}: {el*1e3:.2f} ms")
    del idx
