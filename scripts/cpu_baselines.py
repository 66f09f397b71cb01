#!/usr/bin/env python3
"""CPU baselines beside the GPU numbers: the Faiss-equivalent CPU restatement on the
GPU box's own host cores at nq = 1, 16 (oracle/flat_oracle.c: per-pair scan, OpenMP) and nq = 1024
(oracle/knn_oracle.knn_blas_f32: blocked SGEMM through numpy's BLAS), 1M x 512 fp32, k = 10.
Bounded: the scans run on the first 250k rows and are scaled by 4; the SGEMM leg runs 256 of the
1024 queries against all rows and is scaled by 4.  Baseline, not target."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import host_cores  # noqa: E402
from oracle import flat_oracle as fo, knn_oracle as ko  # noqa: E402

cores = host_cores()
try:
    from threadpoolctl import threadpool_limits
    limit = threadpool_limits(limits=cores)
except Exception:  # pragma: no cover
    limit = None
rng = np.random.default_rng(1234)
n, d, k = 1_000_000, 512, 10
xb = rng.random((n, d), dtype=np.float32)
out = {"host_cores": cores, "cpu_count": os.cpu_count(), "n": n, "d": d, "k": k}
for nq in (1, 16):
    xq = np.random.default_rng(4321).random((nq, d), dtype=np.float32)
    sub = xb[:250_000]
    fo.knn_flat(sub[:1000], xq, k, 1, cores)
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        fo.knn_flat(sub, xq, k, 1, cores)
    el = (time.perf_counter() - t) / reps * 4.0
    out[f"nq{nq}"] = {"qps": nq / el, "ms_per_batch_1M": el * 1e3, "threads": cores, "path": "per-pair scan (flat_oracle.c)"}
    print(f"nq={nq}: {nq / el:.1f} QPS ({el * 1e3:.1f} ms per batch at 1M rows, {cores} threads)", flush=True)
xq = np.random.default_rng(4321).random((256, d), dtype=np.float32)
ko.knn_blas_f32(xb[:65536], xq, k)
t = time.perf_counter()
ko.knn_blas_f32(xb, xq, k)
el = (time.perf_counter() - t) * 4.0
out["nq1024"] = {"qps": 1024 / el, "ms_per_batch_1M": el * 1e3, "threads": cores, "path": "blocked SGEMM + partial top-k (knn_blas_f32, numpy BLAS)"}
print(f"nq=1024: {1024 / el:.1f} QPS ({el * 1e3:.0f} ms per batch, {cores} BLAS threads)", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cpu_baselines.json"), "w"), indent=1)
