#!/usr/bin/env python3
"""BASELINE.json configs 2, 4, 5 and the nq sweep of config 3 on ONE MI355X (numbers for
BASELINE.md section 7).  Writes gpurun_out/configs.json.  Parity gates use the CPU oracle
on bounded samples."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_search_engine_amd.faiss_compat as faiss
from image_search_engine_amd.descriptors import CNNDescriptor
from oracle import knn_oracle as ko

out = {}
dev = torch.device("cuda", 0)

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

def two_stream_qps(index, xq, k, steps):
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % 2]): index.search_torch(xq, k)
    torch.cuda.synchronize()
    return xq.shape[0] * steps / (time.perf_counter() - t)

# ---- config 2: 100k x 512 CNN embeddings (random-init ResNet-50, synthetic uint8 images) + HIP L2 kNN k=10
n_img = int(os.environ.get("CFG2_IMAGES", "100000")); B = 256
res = {}
for name, dt in (("fp32", torch.float32), ("bf16 autocast", torch.bfloat16)):
    cnn = CNNDescriptor(out_dim=512, dtype=dt)
    g = torch.Generator(device=dev).manual_seed(7)
    imgs = torch.randint(0, 256, (B, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
    el = timed(lambda: cnn.extract_features_tensor(imgs), 5)
    res[name] = {"images_per_s": B / el, "batch": B}
    if dt == torch.float32:
        feats = torch.empty((n_img, 512), device=dev)
        t = time.perf_counter()
        for i0 in range(0, n_img, B):
            m = min(B, n_img - i0)
            imgs = torch.randint(0, 256, (m, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
            feats[i0:i0 + m] = cnn.extract_features_tensor(imgs)
        torch.cuda.synchronize(); res[name]["describe_100k_s"] = time.perf_counter() - t
index = faiss.IndexFlatL2(512); index.add_torch(feats)
xq = feats[torch.randperm(n_img, device=dev)[:16]] + 0.01 * torch.randn((16, 512), device=dev)
D, I = index.search_torch(xq, 10); torch.cuda.synchronize()
Dr, Ir = ko.knn_exact(feats.cpu().numpy(), xq.cpu().numpy(), 10, ko.METRIC_L2)
_, _, scan_ms, merge_ms = index.search_timed_torch(xq, 10, 50)
out["config2"] = {"cnn": res, "n": n_img, "d": 512, "k": 10, "nq": 16, "scan_us": scan_ms * 1e3,
                  "qps_2streams": two_stream_qps(index, xq, 10, 200),
                  "alg_GBps": 4.0 * n_img * 512 / scan_ms / 1e6,
                  "ids_identical": bool(np.array_equal(I.cpu().numpy(), Ir)),
                  "max_rel_dist_err": float(np.max(np.abs(D.cpu().numpy() - Dr) / np.maximum(1, Dr)))}
print("config2", json.dumps(out["config2"]), flush=True)
del index, feats

# ---- config 4: 50M x 128 SIFT-valued rows vs 4096 unit centroids, k = 1 (assignment kernel)
n4 = int(os.environ.get("CFG4_ROWS", "50000000")); K, d4 = 4096, 128
cent = torch.nn.functional.normalize(torch.randn((K, d4), device=dev, generator=torch.Generator(device=dev).manual_seed(1)), dim=1)
km = faiss.IndexFlatIP(d4); km.add_torch(cent)   # spherical k-means index (backend/kmeans_faiss.py:36,42)
chunk = 5_000_000; total = 0.0; hist = torch.zeros(K, device=dev, dtype=torch.int64)
g = torch.Generator(device=dev).manual_seed(2)
first = None
for i0 in range(0, n4, chunk):
    m = min(chunk, n4 - i0)
    X = torch.randint(0, 256, (m, d4), generator=g, device=dev).float()
    torch.cuda.synchronize(); t = time.perf_counter()
    Dk, Ik = km.assign_torch(X); torch.cuda.synchronize(); total += time.perf_counter() - t
    hist += torch.bincount(Ik.view(-1), minlength=K)
    if first is None: first = (X[:2000].cpu().numpy(), Ik[:2000].cpu().numpy())
_, Ir = ko.knn_exact(cent.cpu().numpy(), first[0], 1, ko.METRIC_INNER_PRODUCT)
flops = 2.0 * n4 * K * d4
out["config4"] = {"n": n4, "K": K, "d": d4, "seconds": total, "TFLOPs_fp32": flops / total / 1e12,
                  "frac_of_157.3": flops / total / 1e12 / 157.3, "rows_per_s": n4 / total,
                  "labels_sum": int(hist.sum()), "sample_ids_identical": bool(np.array_equal(first[1], Ir))}
print("config4", json.dumps(out["config4"]), flush=True)
del km, X

# ---- config 5 (one GPU's shard of 8): 1.25M x 512 unit rows, bf16 storage, IP, k = 10
n5 = 1_250_000
xb = torch.nn.functional.normalize(torch.randn((n5, 512), device=dev, generator=torch.Generator(device=dev).manual_seed(3)), dim=1)
exact = faiss.IndexFlatIP(512); exact.add_torch(xb)
bf = faiss.IndexFlatIP(512, storage="bf16"); bf.add_torch(xb)
xq = torch.nn.functional.normalize(torch.randn((1000, 512), device=dev), dim=1)
D0, I0 = exact.search_torch(xq, 10); D1, I1 = bf.search_torch(xq, 10); torch.cuda.synchronize()
rec = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(I0.cpu(), I1.cpu())) / 10000.0
_, _, scan_ms, _ = bf.search_timed_torch(xq[:48].contiguous(), 10, 30)
out["config5_shard"] = {"rows": n5, "d": 512, "storage": "bf16", "recall_at_10_vs_fp32_exact_1000q": rec,
                        "scan_us_nq48": scan_ms * 1e3, "alg_GBps_nq48": 2.0 * n5 * 512 / scan_ms / 1e6,
                        "qps_nq48_2streams": two_stream_qps(bf, xq[:48].contiguous(), 10, 100),
                        "qps_nq1000": two_stream_qps(bf, xq, 10, 10)}
print("config5", json.dumps(out["config5_shard"]), flush=True)
del exact, bf, xb

# ---- config 3 sweep on one GPU: 1M x 512 fp32 L2 k=10
rng = np.random.default_rng(1234)
xb = torch.from_numpy(rng.random((1_000_000, 512), dtype=np.float32)).to(dev)
index = faiss.IndexFlatL2(512); index.add_torch(xb)
sweep = []
for nq in (1, 16, 32, 48, 1024):
    xq = torch.from_numpy(np.random.default_rng(4321).random((nq, 512), dtype=np.float32)).to(dev)
    _, _, scan_ms, merge_ms = index.search_timed_torch(xq, 10, 20)
    sweep.append({"nq": nq, "scan_us": scan_ms * 1e3, "merge_us": merge_ms * 1e3,
                  "qps_2streams": two_stream_qps(index, xq, 10, 100 if nq < 1024 else 10),
                  "alg_GBps": (4.0 * 1e6 * 512 + 4 * nq * 512 + 12 * nq * 10) / scan_ms / 1e6})
out["config3_sweep_1gpu"] = sweep
print("config3", json.dumps(sweep), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
