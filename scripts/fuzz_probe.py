"""Randomised differential run of the C-ABI search against the CPU oracle (dev aid on the GPU box; the oracle is the
checker, as in tests/): random n, d, k, nq, metric, storage, data shape (uniform / offset clusters / duplicates /
tiny n), adds in several pieces, host and device entry points, shard-style id bases.  Prints one line per failure and
a summary; exits 1 if anything failed."""
import os, sys, time, traceback
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
from oracle import knn_oracle as ko
from tests.knn_checks import assert_knn_matches

budget = float(os.environ.get("SECONDS", "150"))
seed0 = int(os.environ.get("SEED", "0"))
t_end = time.time() + budget
fails, runs, case = 0, 0, 0
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng(seed0 * 100003 + case)
    metric = int(rng.integers(0, 2))
    storage = "bf16" if rng.random() < 0.25 else "f32"
    d = int(rng.choice([1, 3, 7, 16, 20, 31, 32, 33, 64, 96, 100, 128, 130, 200, 256, 384, 512, 520, 640]))
    n = int(rng.choice([1, 2, 5, 15, 16, 17, 100, 1000, 4095, 4096, 4097, 20000, 60000]))
    if n * d > 16_000_000:
        n = 16_000_000 // d
    nq = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 48, 49, 64, 70]))
    k = int(rng.choice([1, 2, 5, 10, 16, 20, 28, 29, 32, 33, 40, 64, 100]))
    shape = str(rng.choice(["uniform", "offset", "clusters", "dups", "small_int"]))
    if shape == "uniform":
        xb = rng.random((n, d), dtype=np.float32)
    elif shape == "offset":
        xb = (rng.random((n, d), dtype=np.float32) * np.float32(0.1) + np.float32(rng.choice([10.0, 300.0])))
    elif shape == "clusters":
        c = rng.standard_normal((4, d)).astype(np.float32) * np.float32(rng.choice([1.0, 100.0]))
        xb = np.sort(rng.integers(0, 4, n))[:, None] * 0 + c[np.sort(rng.integers(0, 4, n))] + rng.standard_normal((n, d)).astype(np.float32) * np.float32(0.05)
    elif shape == "dups":
        base = rng.random((max(1, n // 7), d), dtype=np.float32)
        xb = base[rng.integers(0, base.shape[0], n)]
    else:
        xb = rng.integers(0, 4, (n, d)).astype(np.float32)      # many exact ties
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    xq = xb[rng.integers(0, n, nq)] + (rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.01) if rng.random() < 0.7 else 0)
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    desc = f"case {case}: metric={metric} storage={storage} n={n} d={d} nq={nq} k={k} data={shape}"
    try:
        index = faiss.IndexFlat(d, metric, storage=storage)
        pieces = int(rng.integers(1, 4))
        cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, pieces - 1)]))
        for a, b in zip(cuts, cuts[1:]):
            if rng.random() < 0.5:
                index.add(xb[a:b])
            else:
                index.add_torch(torch.from_numpy(xb[a:b]).cuda())
        assert index.ntotal == n
        if storage == "bf16":
            rnd = lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()
            xb_o, xq_o = rnd(xb), rnd(xq)
        else:
            xb_o, xq_o = xb, xq
        Dr, Ir = ko.knn_exact(xb_o, xq_o, k, metric)
        if rng.random() < 0.5:
            D, I = index.search(xq, k)
        else:
            Dt, It = index.search_torch(torch.from_numpy(xq).cuda(), k)
            D, I = Dt.cpu().numpy(), It.cpu().numpy()
        if storage == "bf16" and metric == 1:
            # bf16 L2 is the clamped expanded form: approximate by construction -> recall only
            ok = np.mean([len(set(I[q][I[q] >= 0]) & set(Ir[q][Ir[q] >= 0])) / max(1, (Ir[q] >= 0).sum()) for q in range(nq)])
            assert ok >= 0.9 or shape in ("dups", "small_int", "offset", "clusters"), f"recall {ok}"
        else:
            # float32 accumulation against the float64 oracle: the error scales with |x||y| (an inner product
            # of far-apart clusters can cancel to ~0), so the tolerance floor does too
            scale = float(np.linalg.norm(xq_o, axis=1).max() * np.linalg.norm(xb_o, axis=1).max())
            try:
                assert_knn_matches(D, I, Dr, Ir, xb_o, xq_o, metric, rtol=max(1e-4, 4e-7 * scale))
            except AssertionError as e:
                if "near-ties" not in str(e):
                    raise
                # the checker calls two ranks a tie when they differ by 2e-6 |score|; an inner product that
                # cancels is only good to a multiple of 2^-24 |x||y|: a swap inside that band is not an error either
                from tests.knn_checks import true_scores
                ts = true_scores(xb_o, xq_o, I, metric)
                ref = Dr.astype(np.float64) if metric == 1 else Dr.astype(np.float64)
                # a float32 chain of d products: ~sqrt(d) 2^-24 |x||y| (worst case d 2^-24 |x||y|)
                band = 4.0 * np.sqrt(d) * 2.0 ** -24 * np.linalg.norm(xq_o, axis=1)[:, None] * np.linalg.norm(xb_o, axis=1).max()
                mism = (I != Ir) & (Ir >= 0)
                assert (np.abs(ts - ref)[mism] <= band.repeat(I.shape[1], 1)[mism]).all(), str(e)
        runs += 1
        del index
    except Exception as e:
        fails += 1
        print("FAIL", desc, "->", repr(e)[:300], flush=True)
        if fails <= 3:
            traceback.print_exc()
print(f"fuzz: {runs} cases passed, {fails} failed, seed {seed0}")
sys.exit(1 if fails else 0)
