"""Who finishes late?  Block finish time of the scan kernel by blockIdx % 8 (XCD group) and by
position (dev aid; ablate build + ISE_STAMPS)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k, nq, n = 512, 10, 16, 1_000_000
xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
index = faiss.IndexFlatL2(d); index.add_torch(xb)
for _ in range(5): index.search_torch(xq, k)
for rep in range(3):
    st = torch.zeros((1024 * 8 * 16,), dtype=torch.int64, device="cuda")
    os.environ["ISE_STAMPS"] = str(st.data_ptr())
    index.search_torch(xq, k); torch.cuda.synchronize()
    os.environ.pop("ISE_STAMPS")
    s = st.cpu().numpy().reshape(1024, 8, 16).astype(np.float64)
    used = s[:, :, 0].max(axis=1) > 0
    nb = int(used.sum()); s = s[:nb]
    t0 = s[:, :, 0].min()
    end = (s[:, :, 4] - t0) / 100.0          # loop end per wave, us
    blk = end.max(axis=1)                    # block = its slowest wave
    print(f"rep {rep}: blocks={nb} loop-end per block: median {np.median(blk):.1f} p90 {np.percentile(blk,90):.1f} max {blk.max():.1f} us")
    print("   by blockIdx%8 (median / max): " + "  ".join(f"{np.median(blk[g::8]):.0f}/{blk[g::8].max():.0f}" for g in range(8)))
    print("   by block quartile of index (median): " + "  ".join(f"{np.median(blk[i*nb//4:(i+1)*nb//4]):.0f}" for i in range(4)))
    wv = end[:-1]                            # ignore the short last block
    print(f"   per wave: median {np.median(wv):.1f}  p10 {np.percentile(wv,10):.1f}  p90 {np.percentile(wv,90):.1f}  max {wv.max():.1f}")
