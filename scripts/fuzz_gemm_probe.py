"""Randomised differential run of the LARGE-BATCH paths (nq >= 128 against >= 128k rows: GEMM-shaped passes, threshold
sample, candidate buffers, re-rank, exact-scan fallback) against the C oracle on the GPU box's cores.  float32 L2,
float32 inner product and bf16 inner product; uniform / offset / clustered / duplicated rows; queries near rows, far from all rows, duplicated."""
import os, sys, time, traceback
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
from oracle import flat_oracle as fo
from tests.knn_checks import assert_knn_matches

fo.build()
budget = float(os.environ.get("SECONDS", "200")); seed0 = int(os.environ.get("SEED", "0"))
t_end = time.time() + budget
fails = runs = case = 0
stats = {"gemm_chunks": 0, "exact_scan": 0}
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng(seed0 * 104729 + case)
    kind = rng.random()
    bf16 = kind < 0.3
    f32ip = 0.3 <= kind < 0.55
    metric = 0 if (bf16 or f32ip) else 1
    d = int(rng.choice([128, 256, 384, 512]))
    n = int(rng.choice([131072, 140001, 200000, 262144 + 17]))
    nq = int(rng.choice([128, 256, 257, 300, 512, 1024, 1100]))
    k = int(rng.choice([1, 5, 10, 20, 28, 32]))
    shape = str(rng.choice(["uniform", "offset", "clusters", "dups"]))
    if shape == "uniform":
        xb = rng.random((n, d), dtype=np.float32)
    elif shape == "offset":
        xb = rng.random((n, d), dtype=np.float32) * np.float32(0.1) + np.float32(30.0)
    elif shape == "clusters":
        c = rng.standard_normal((8, d)).astype(np.float32) * np.float32(rng.choice([1.0, 30.0]))
        xb = c[np.sort(rng.integers(0, 8, n))] + rng.standard_normal((n, d)).astype(np.float32) * np.float32(0.1)
    else:
        base = rng.random((n // 50, d), dtype=np.float32)
        xb = base[rng.integers(0, base.shape[0], n)]
    if bf16 or f32ip:
        xb = xb - xb.mean(0, keepdims=True)
        xb = xb / np.maximum(np.linalg.norm(xb, axis=1, keepdims=True), 1e-20)
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    qk = str(rng.choice(["near", "far", "same"]))
    if qk == "near":
        xq = xb[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.01)
    elif qk == "far":
        xq = rng.standard_normal((nq, d)).astype(np.float32) * np.float32(3.0)
    else:
        xq = np.repeat(xb[rng.integers(0, n, 1)], nq, axis=0)
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    desc = f"case {case}: {'bf16 IP' if bf16 else ('f32 IP' if f32ip else 'f32 L2')} n={n} d={d} nq={nq} k={k} rows={shape} queries={qk}"
    try:
        index = faiss.IndexFlat(d, metric, storage="bf16" if bf16 else "f32")
        index.add(xb)
        D, I = index.search(xq, k)
        st = index.exact_stats()
        stats["gemm_chunks"] += st["gemm_chunks"]; stats["exact_scan"] += st["exact_scan"]
        if bf16:
            rnd = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
            xb_o, xq_o = rnd(xb), rnd(xq)
        else:
            xb_o, xq_o = xb, xq
        Dr, Ir, _ = fo.knn_flat(xb_o, xq_o, k, metric, 16)
        scale = float(np.linalg.norm(xq_o, axis=1).max() * np.linalg.norm(xb_o, axis=1).max())
        assert_knn_matches(D, I, Dr, Ir, xb_o, xq_o, metric, rtol=max(1e-4, 4e-7 * scale) if metric == 0 else 1e-4)
        runs += 1
        del index
    except Exception as e:
        fails += 1
        print("FAIL", desc, "->", repr(e)[:300], flush=True)
        if fails <= 3:
            traceback.print_exc()
print(f"fuzz_gemm: {runs} cases passed, {fails} failed, seed {seed0}; {stats}")
sys.exit(1 if fails else 0)
