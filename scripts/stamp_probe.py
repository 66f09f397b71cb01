"""In-kernel phase stamps of the scan kernel (dev aid; ablate build + ISE_STAMPS)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k = 512, 10
for n, nq in ((125_000, 16), (1_000_000, 16), (1_000_000, 32), (1_000_000, 48)):
    xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
    index = faiss.IndexFlatL2(d); index.add_torch(xb)
    for _ in range(5): index.search_torch(xq, k)
    st = torch.zeros((1024 * 8 * 16,), dtype=torch.int64, device="cuda")
    os.environ["ISE_STAMPS"] = str(st.data_ptr())
    index.search_torch(xq, k); torch.cuda.synchronize()
    os.environ.pop("ISE_STAMPS")
    s = st.cpu().numpy().reshape(1024, 8, 16).astype(np.float64)
    used = s[:, :, 0].max(axis=1) > 0
    s = s[used]                      # [blocks][waves][stamps]
    t0 = s[:, :, 0].min()
    us = (s - t0) / 100.0            # 100 MHz -> us
    names = ["entry", "staged", "boot in", "boot out", "loop end", "final barrier", "exit"]
    clk = (s[:, :, 9] - s[:, :, 8]) / np.maximum(s[:, :, 4] - s[:, :, 1], 1) * 100.0  # MHz
    print(f"n={n} nq={nq}: blocks={s.shape[0]}  in-kernel clock over the main loop: median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})")
    for i, nm in enumerate(names):
        v = us[:, :, i]
        print(f"  {nm:14s} min {v.min():8.2f}  median {np.median(v):8.2f}  max {v.max():8.2f} us")
    print(f"  per-wave final phase (exit - final barrier): median {np.median(us[:,:,6]-us[:,:,5]):.2f} max {(us[:,:,6]-us[:,:,5]).max():.2f}")
    print(f"  per-wave boot (out - in): median {np.median(us[:,:,3]-us[:,:,2]):.2f} max {(us[:,:,3]-us[:,:,2]).max():.2f}")
    print(f"  boot: wait for block (barrier1 - in): median {np.median(us[:,:,7]-us[:,:,2]):.2f} max {(us[:,:,7]-us[:,:,2]).max():.2f};"
          f" select+barrier2 (out - barrier1): median {np.median(us[:,:,3]-us[:,:,7]):.2f} max {(us[:,:,3]-us[:,:,7]).max():.2f}")
    print(f"  barrier wait (final barrier - loop end): median {np.median(us[:,:,5]-us[:,:,4]):.2f} max {(us[:,:,5]-us[:,:,4]).max():.2f}")
    del index, xb
