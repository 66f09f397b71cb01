"""BASELINE config 2: 100k x 512 embeddings of a seeded random-init ResNet-50 on seeded synthetic images ->
L2 index -> searches.  How often does the exact path's certificate fail on such data (post-ReLU features
with a large common component), what do the searches cost, and are the ids the oracle's?"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_engine_amd.descriptors import CNNDescriptor
from image_search_engine_amd.utils import create_search_index
from oracle import flat_oracle as fo
from tests.knn_checks import assert_knn_matches

n, nqs, d, k = int(os.environ.get("N", "100000")), 1024, 512, 10
desc = CNNDescriptor(out_dim=d, seed=0)
g = torch.Generator(device="cuda").manual_seed(1234)
feats = []
t0 = time.time()
for i0 in range(0, n + nqs, 500):
    imgs = torch.randint(0, 256, (min(500, n + nqs - i0), 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    feats.append(desc.extract_features_tensor(imgs))
feats = torch.cat(feats)
torch.cuda.synchronize()
t_desc = time.time() - t0
xb, xq = feats[:n].cpu().numpy(), feats[n:].cpu().numpy()
print(f"described {n + nqs} images in {t_desc:.1f} s ({(n + nqs) / t_desc:.0f} img/s); |y|^2 ~ {float((xb ** 2).sum(1).mean()):.3g}, "
      f"spread |y - mean|^2 ~ {float(((xb - xb.mean(0)) ** 2).sum(1).mean()):.3g}")
index = create_search_index(xb.copy(), index_type="l2")
res = {}
for nq in (1, 16, 256, 1024):
    q = torch.from_numpy(xq[:nq]).cuda()
    s0 = index.exact_stats()
    for _ in range(3): index.search_torch(q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): D, I = index.search_torch(q, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    s1 = index.exact_stats()
    frac = (s1["exact_scan"] - s0["exact_scan"]) / max(1, s1["reranked"] - s0["reranked"])
    Dc, Ic, _ = fo.knn_flat(xb, xq[:nq], k, 1, 16)
    mism = assert_knn_matches(D.cpu().numpy(), I.cpu().numpy(), Dc, Ic, xb, xq[:nq], 1)
    res[nq] = {"us_per_batch": round(dt * 1e6, 1), "qps": round(nq / dt), "exact_scan_fraction": round(frac, 4), "id_mismatches_at_ties": mism}
    print(nq, res[nq])
json.dump({"n": n, "d": d, "k": k, "describe_images_per_s": round((n + nqs) / t_desc), "search": res},
          open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "config2_probe.json"), "w"), indent=1)
