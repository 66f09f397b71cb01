"""dev build: per-block end-of-loop times of the one-tile scan kernel at 1M x 512, grouped by blockIdx % 8
(blocks b and b + 8 share an XCD) -- where does the cross-block tail come from?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
n, d, nq, k = 1_000_000, 512, 16, 10
xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
for _ in range(50): index.search_torch(xq, k)
for rep in range(3):
    st = torch.zeros((1024 * 8 * 16,), dtype=torch.int64, device="cuda")
    os.environ["ISE_STAMPS"] = str(st.data_ptr())
    index.search_torch(xq, k); torch.cuda.synchronize()
    os.environ.pop("ISE_STAMPS")
    s = st.cpu().numpy().reshape(1024, 8, 16).astype(np.float64)
    used = s[:, :, 0].max(axis=1) > 0
    nb = int(used.sum())
    s = s[:nb]
    t0 = s[:, :, 0].min()
    loop_end = (s[:, :, 4].max(axis=1) - t0) / 100.0     # per block: its slowest wave
    boot_in = (s[:, :, 2].min(axis=1) - t0) / 100.0
    full = np.arange(nb) < nb - 1
    print(f"rep {rep}: blocks {nb}; loop end median {np.median(loop_end[full]):.1f} p90 {np.percentile(loop_end[full], 90):.1f} max {loop_end[full].max():.1f}")
    for r in range(8):
        sel = full & (np.arange(nb) % 8 == r)
        print(f"   b%8={r}: loop-end median {np.median(loop_end[sel]):7.1f}  min {loop_end[sel].min():7.1f} max {loop_end[sel].max():7.1f}   first-tile median {np.median(boot_in[sel]):5.1f}")
    # first half of the grid vs second half (which block of a CU's pair started first?)
    print("   by block index quartile:", [round(float(np.median(loop_end[full & (np.arange(nb) // (nb // 4 + 1) == qd)])), 1) for qd in range(4)])
