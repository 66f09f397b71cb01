"""Randomised differential run of the other C-ABI entry points against the CPU oracle (dev aid on the GPU box):
  * shard-style search: rows split into 1-5 shards with id bases, packed keys, merge == unsharded oracle;
  * k = 1 assignment kernel (FaissKMeans.transform's search) == oracle argmin / argmax;
  * normalize_L2 in place (host and device) == oracle, zero rows untouched;
  * write_index / read_index / reconstruct_n round trips, reset + re-add."""
import os, sys, tempfile, time, traceback
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_search_engine_amd.faiss_compat as faiss
from oracle import knn_oracle as ko
from tests.knn_checks import assert_knn_matches

budget = float(os.environ.get("SECONDS", "150"))
seed0 = int(os.environ.get("SEED", "0"))
t_end = time.time() + budget
fails, runs, case = 0, {"shards": 0, "assign": 0, "normalize": 0, "persist": 0}, 0
tmp = tempfile.mkdtemp(prefix="ise_fuzz_")
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng(seed0 * 7919 + case)
    kind = str(rng.choice(["shards", "assign", "normalize", "persist"]))
    d = int(rng.choice([3, 16, 32, 33, 64, 100, 128, 256, 512]))
    desc = f"case {case} {kind} d={d}"
    try:
        if kind == "shards":
            metric = int(rng.integers(0, 2)); storage = "bf16" if (metric == 0 and rng.random() < 0.3) else "f32"
            n = int(rng.choice([7, 100, 1000, 5000, 30000])); nq = int(rng.choice([1, 16, 17, 40])); k = int(rng.choice([1, 5, 10, 20, 33]))
            g = int(rng.integers(1, 6))
            xb = rng.random((n, d), dtype=np.float32) - np.float32(0.5 if metric == 0 else 0)
            if n > 50:
                xb[n - 3] = xb[2]
            xq = np.ascontiguousarray(xb[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.02))
            desc += f" metric={metric} storage={storage} n={n} nq={nq} k={k} shards={g}"
            rnd = (lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()) if storage == "bf16" else (lambda a: a)
            keys = []
            for r in range(g):
                lo, hi = n * r // g, n * (r + 1) // g
                idx = faiss.IndexFlat(d, metric, storage=storage)
                if hi > lo:
                    idx.add(xb[lo:hi])
                    keys.append(idx.search_keys_torch(torch.from_numpy(xq).cuda(), k, lo))
                else:   # an empty shard contributes a list of pads
                    keys.append(torch.full((nq, k), -1, dtype=torch.int64, device="cuda"))
            D, I = faiss.merge_keys_torch(torch.stack(keys), metric)
            Dr, Ir = ko.knn_exact(rnd(xb), rnd(xq), k, metric)
            assert_knn_matches(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, rnd(xb), rnd(xq), metric)
        elif kind == "assign":
            K = int(rng.choice([2, 16, 200, 256, 1000, 4096])); n = int(rng.choice([2048, 2049, 5000, 40000]))
            spherical = rng.random() < 0.5
            desc += f" K={K} n={n} spherical={spherical}"
            c = rng.standard_normal((K, d)).astype(np.float32)
            x = rng.integers(0, 256, (n, d)).astype(np.float32) if rng.random() < 0.5 else rng.standard_normal((n, d)).astype(np.float32)
            metric = 0 if spherical else 1
            if spherical:
                c = ko.normalize_rows(c)
            idx = faiss.IndexFlat(d, metric); idx.add(c)
            assert idx._assign_applies(n, 1)
            D, I = idx.assign_torch(torch.from_numpy(x).cuda())
            Dr, Ir = ko.knn_exact(c, x, 1, metric)
            scale = float(np.linalg.norm(x, axis=1).max() * np.linalg.norm(c, axis=1).max())
            assert_knn_matches(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, c, x, metric, rtol=max(1e-4, 4e-7 * scale))
        elif kind == "normalize":
            n = int(rng.choice([1, 5, 1000, 20000]))
            x = (rng.standard_normal((n, d)) * rng.choice([1e-3, 1.0, 1e4])).astype(np.float32)
            x[rng.integers(0, n)] = 0
            want = ko.normalize_rows(x.copy())
            a = x.copy(); faiss.normalize_L2(a)
            t = torch.from_numpy(x.copy()).cuda(); faiss.normalize_L2(t)
            for got in (a, t.cpu().numpy()):
                assert np.allclose(got, want, rtol=2e-6, atol=1e-30), np.abs(got - want).max()
                assert (got[np.all(x == 0, axis=1)] == 0).all()
        else:
            metric = int(rng.integers(0, 2)); n = int(rng.choice([0, 1, 17, 1000, 20000]))
            desc += f" metric={metric} n={n}"
            xb = rng.standard_normal((n, d)).astype(np.float32)
            idx = faiss.IndexFlat(d, metric)
            if n:
                idx.add(xb)
            path = os.path.join(tmp, f"i{case}.faiss")
            faiss.write_index(idx, path)
            back = faiss.read_index(path)
            os.unlink(path)
            assert back.ntotal == n and back.d == d and back.metric_type == metric
            if n:
                assert np.array_equal(back.reconstruct_n(0, n), xb)
                i0 = int(rng.integers(0, n)); m = int(rng.integers(0, n - i0 + 1))
                assert np.array_equal(idx.reconstruct_n(i0, m), xb[i0:i0 + m])
                xq = xb[:min(n, 3)].copy()
                D1, I1 = idx.search(xq, 4); D2, I2 = back.search(xq, 4)
                assert np.array_equal(I1, I2) and np.array_equal(D1, D2)
                idx.reset(); assert idx.ntotal == 0
                idx.add(xb[::-1].copy())
                D3, I3 = idx.search(xq, 1)
                Dr, Ir = ko.knn_exact(xb[::-1].copy(), xq, 1, metric)
                scale = float(np.linalg.norm(xq, axis=1).max() * np.linalg.norm(xb, axis=1).max())
                assert_knn_matches(D3, I3, Dr, Ir, xb[::-1].copy(), xq, metric, rtol=max(1e-4, 4e-7 * scale))
        runs[kind] += 1
    except Exception as e:
        fails += 1
        print("FAIL", desc, "->", repr(e)[:300], flush=True)
        if fails <= 3:
            traceback.print_exc()
print(f"fuzz_aux: passed {runs}, failed {fails}, seed {seed0}")
sys.exit(1 if fails else 0)
