"""In-kernel phase stamps of the short-index kernel (dev aid; `make ablate` build + ISE_STAMPS).
usage: ISE_KNN_LIB=.../libise_knn_ablate.so python scripts/short_stamp_probe.py [n ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
d, k = 512, 10
sizes = [int(a) for a in sys.argv[1:]] or [100_000, 125_000]
names = ["entry", "staged", "loop end", "barrier", "selected"]
for n in sizes:
    for nq in (16, 1):
        xb = torch.rand((n, d), device="cuda"); xq = torch.rand((nq, d), device="cuda")
        index = faiss.IndexFlatL2(d); index.add_torch(xb)
        for _ in range(20): index.search_torch(xq, k)
        st = torch.zeros((1024 * 8 * 16,), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        os.environ["ISE_STAMPS"] = str(st.data_ptr())
        index.search_torch(xq, k); torch.cuda.synchronize()
        os.environ.pop("ISE_STAMPS")
        s = st.cpu().numpy().reshape(-1, 1, 16).astype(np.float64)   # one row per wave (8 or 16 waves per block)
        used = s[:, 0, 0] > 0
        s = s[used]
        t0 = s[:, :, 0][s[:, :, 0] > 0].min()
        us = np.where(s > 0, (s - t0) / 100.0, np.nan)
        print(f"n={n} nq={nq}: waves={s.shape[0]} short={index.short_stats()}")
        for i, nm in enumerate(names):
            v = us[:, :, i]
            if np.isnan(v).all():
                continue
            print(f"  {nm:10s} min {np.nanmin(v):8.2f}  p10 {np.nanpercentile(v,10):8.2f} median {np.nanmedian(v):8.2f}  p90 {np.nanpercentile(v,90):8.2f} max {np.nanmax(v):8.2f} us   (waves stamped: {np.count_nonzero(~np.isnan(v))})")
        ft = us[:, :, 8]
        if not np.isnan(ft).all():
            print(f"  first tile done: median {np.nanmedian(ft):.2f} p90 {np.nanpercentile(ft,90):.2f} max {np.nanmax(ft):.2f}; first tile time (done - staged): median {np.nanmedian(ft-us[:,:,1]):.2f} p90 {np.nanpercentile(ft-us[:,:,1],90):.2f}")
        if not np.isnan(us[:, :, 9]).all():
            seg = [("load", 9, 3), ("threshold", 10, 9), ("compact", 11, 10), ("rank+write", 12, 11), ("second query", 4, 12)]
            print("  select, first query of a wave: " + "; ".join(f"{nm} {np.nanmedian(us[:,:,a]-us[:,:,b]):.2f} (p90 {np.nanpercentile(us[:,:,a]-us[:,:,b],90):.2f})" for nm, a, b in seg))
        print(f"  stream (loop end - staged): median {np.nanmedian(us[:,:,2]-us[:,:,1]):.2f} max {np.nanmax(us[:,:,2]-us[:,:,1]):.2f};"
              f" select (selected - barrier): median {np.nanmedian(us[:,:,4]-us[:,:,3]):.2f} max {np.nanmax(us[:,:,4]-us[:,:,3]):.2f}")
        del index, xb
