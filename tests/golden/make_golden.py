"""Generate the golden kNN fixtures under tests/golden/ (SURVEY.md 8c).

The reference cannot run here (backend/config.py:46 is a SyntaxError, faiss is
absent) and ships no vectors of its own, so these fixtures come from the exact
float64 oracle (oracle/knn_oracle.py), cross-checked against torch-CPU
cdist/topk and sklearn brute-force NearestNeighbors.  PARITY UNPINNED with
respect to a real Faiss build; see oracle/knn_oracle.py.

Each .npz stores inputs (xb, xq), k, metric, expected I (int64), D (float32
rounded from float64) and the smallest float64 gap between consecutive ranks
1..k+1 per query, so a test can tell a real mismatch from a float32 near-tie.

Run:  python tests/golden/make_golden.py            (re)write the fixtures
      python tests/golden/make_golden.py --check    regenerate every fixture into a scratch directory and
                                                    compare it, array by array and bit for bit, with the
                                                    committed file (also a CPU test: tests/test_oracle.py)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import knn_oracle as ko  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE  # where emit() writes; --check points it at a scratch directory
L2, IP = ko.METRIC_L2, ko.METRIC_INNER_PRODUCT


def crosscheck(xb, xq, k, metric, I, D):
    import torch
    from sklearn.neighbors import NearestNeighbors

    if xb.shape[0] < k:
        return
    tb, tq = torch.from_numpy(xb).double(), torch.from_numpy(xq).double()
    if metric == L2:
        S = torch.cdist(tq, tb, compute_mode="donot_use_mm_for_euclid_dist") ** 2
        v, i = torch.topk(S, k, dim=1, largest=False)
        nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric="sqeuclidean").fit(xb.astype(np.float64))
        _, si = nn.kneighbors(xq.astype(np.float64))
        gap = ko.kth_gap(xb, xq, k, metric)
        ok = gap > 1e-9  # exact ties may be ordered differently by the cross-checkers
        assert (si[ok] == I[ok]).all(), "sklearn disagrees with the oracle"
    else:
        S = tq @ tb.T
        v, i = torch.topk(S, k, dim=1, largest=True)
        gap = ko.kth_gap(xb, xq, k, metric)
        ok = gap > 1e-9
    assert (i.numpy()[ok] == I[ok]).all(), "torch disagrees with the oracle"
    assert np.allclose(v.numpy()[ok], D[ok], rtol=1e-6, atol=1e-6)


def emit(name, xb, xq, k, metric, extra=None):
    D, I = ko.knn_exact(xb, xq, k, metric)
    gap = ko.kth_gap(xb, xq, k, metric) if xb.shape[0] > 1 else np.full(xq.shape[0], np.inf)
    crosscheck(xb, xq, k, metric, I, D)
    d = dict(xb=xb, xq=xq, k=np.int64(k), metric=np.int64(metric), I=I, D=D, gap=gap)
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"{name:28s} N={xb.shape[0]:5d} d={xb.shape[1]:4d} nq={xq.shape[0]:3d} k={k:3d} "
          f"min gap {gap.min():.3e}")


def main():
    rng = np.random.default_rng(0)
    # (1) small L2
    emit("l2_n1000_d128_k5", rng.random((1000, 128), dtype=np.float32), rng.random((8, 128), dtype=np.float32), 5, L2)
    # (2) d=512, L2 and IP-on-normalised (kept small: the .npz must stay a few 100 KB)
    xb = rng.random((256, 512), dtype=np.float32)
    xq = rng.random((4, 512), dtype=np.float32)
    emit("l2_n256_d512_k10", xb, xq, 10, L2)
    xbn, xqn = ko.normalize_rows(xb), ko.normalize_rows(xq)
    emit("ip_norm_n256_d512_k10", xbn, xqn, 10, IP)
    # (3) d not a multiple of 16/64: tails
    for d in (32, 100, 200):
        emit(f"l2_n500_d{d}_k7", rng.random((500, d), dtype=np.float32), rng.random((5, d), dtype=np.float32), 7, L2)
    # (4) k > N padding
    emit("l2_kgtN_n3_d16_k5", rng.random((3, 16), dtype=np.float32), rng.random((2, 16), dtype=np.float32), 5, L2)
    emit("ip_kgtN_n3_d16_k5", rng.random((3, 16), dtype=np.float32), rng.random((2, 16), dtype=np.float32), 5, IP)
    # (5) duplicate rows -> ties ordered by lowest id first
    base = rng.random((40, 64), dtype=np.float32)
    xb = np.concatenate([base, base[:20], base[:10]])
    emit("l2_dups_n70_d64_k6", xb, base[:6].copy(), 6, L2)
    emit("ip_dups_n70_d64_k6", xb, base[:6].copy(), 6, IP)
    # (7) nq = 1 and nq = 33 (straddles Faiss's 20-query switch and our 16-query tile)
    xb = rng.random((800, 64), dtype=np.float32)
    emit("l2_nq1_n800_d64_k10", xb, rng.random((1, 64), dtype=np.float32), 10, L2)
    emit("l2_nq33_n800_d64_k10", xb, rng.random((33, 64), dtype=np.float32), 10, L2)
    # k = 20 (NUM_IMAGES_TO_RETURN, backend/config.py:39) and k = 9 IP (siamese/config.py:98)
    emit("l2_n600_d128_k20", rng.random((600, 128), dtype=np.float32), rng.random((3, 128), dtype=np.float32), 20, L2)
    xb = ko.normalize_rows(rng.standard_normal((600, 128)).astype(np.float32))
    xq = ko.normalize_rows(rng.standard_normal((3, 128)).astype(np.float32))
    emit("ip_n600_d128_k9", xb, xq, 9, IP)
    # (8) k = 1 assignment: unit-norm centroids, IP-argmax == L2-argmin
    cent = ko.normalize_rows(rng.standard_normal((256, 128)).astype(np.float32))
    X = rng.integers(0, 256, (512, 128)).astype(np.float32)
    emit("assign_ip_n512_c256_d128", cent, X, 1, IP)
    emit("assign_l2_n512_c256_d128", cent, X, 1, L2)
    # (6) normalize_L2 incl. a zero row
    x = rng.standard_normal((9, 100)).astype(np.float32)
    x[3] = 0
    np.savez_compressed(os.path.join(OUT, "normalize_n9_d100.npz"), x=x, y=ko.normalize_rows(x))
    # (10) 8-way row shard + merge equals unsharded
    xb = rng.random((1000, 32), dtype=np.float32)
    xq = rng.random((6, 32), dtype=np.float32)
    parts_D, parts_I = [], []
    for r in range(8):
        lo, hi = 1000 * r // 8, 1000 * (r + 1) // 8
        Dp, Ip = ko.knn_exact(xb[lo:hi], xq, 10, L2, id_offset=lo)
        parts_D.append(Dp)
        parts_I.append(Ip)
    Dm, Im = ko.merge_shards(parts_D, parts_I, 10, L2)
    D, I = ko.knn_exact(xb, xq, 10, L2)
    assert (Im == I).all() and np.array_equal(Dm, D)
    emit("l2_shard8_n1000_d32_k10", xb, xq, 10, L2)


def emit_seeded(name, kind, seed, n, d, nq, k, metric):
    """SURVEY.md 8c's full-size fixtures, stored as seed + expected outputs (tests/knn_checks.py
    regenerates the inputs): the arrays themselves would be megabytes."""
    from tests.knn_checks import seeded_inputs

    xb, xq = seeded_inputs(kind, seed, n, d, nq)
    D, I = ko.knn_exact(xb, xq, k, metric)
    gap = ko.kth_gap(xb, xq, k, metric)
    crosscheck(xb, xq, k, metric, I, D)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), kind=kind, seed=np.int64(seed), n=np.int64(n), d=np.int64(d),
                        nq=np.int64(nq), k=np.int64(k), metric=np.int64(metric), I=I, D=D, gap=gap)
    print(f"{name:34s} N={n:5d} d={d:4d} nq={nq:4d} k={k:3d} min gap {gap.min():.3e}")


def main_seeded():
    # (2) at SURVEY.md 8c's size: N = 4096, d = 512, nq = 4, k = 10 -- L2 and IP on normalised rows
    emit_seeded("seeded_l2_n4096_d512_k10", "uniform", 20, 4096, 512, 4, 10, L2)
    emit_seeded("seeded_ip_norm_n4096_d512_k10", "uniform_unit", 21, 4096, 512, 4, 10, IP)
    # (8) at its size: 2048 x 128 descriptors against 256 unit-norm centroids, IP-argmax == L2-argmin
    emit_seeded("seeded_assign_ip_n2048_c256_d128", "assign", 22, 256, 128, 2048, 1, IP)
    emit_seeded("seeded_assign_l2_n2048_c256_d128", "assign", 22, 256, 128, 2048, 1, L2)


def check(scratch=None, verbose=True):
    """Regenerate everything into ``scratch`` and compare with the committed fixtures.  Returns the list of
    (file, problem) differences (empty = every committed .npz is reproduced bit for bit and none is missing
    or left over)."""
    import glob
    import tempfile

    global OUT
    saved = OUT
    problems = []
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        OUT = tmp
        try:
            if verbose:
                main()
                main_seeded()
            else:
                import contextlib
                import io

                with contextlib.redirect_stdout(io.StringIO()):
                    main()
                    main_seeded()
        finally:
            OUT = saved
        new = {os.path.basename(f) for f in glob.glob(os.path.join(tmp, "*.npz"))}
        old = {os.path.basename(f) for f in glob.glob(os.path.join(HERE, "*.npz"))}
        for name in sorted(old - new):
            problems.append((name, "committed but not generated"))
        for name in sorted(new - old):
            problems.append((name, "generated but not committed"))
        for name in sorted(new & old):
            a, b = np.load(os.path.join(tmp, name)), np.load(os.path.join(HERE, name))
            if sorted(a.files) != sorted(b.files):
                problems.append((name, f"keys {sorted(a.files)} != {sorted(b.files)}"))
                continue
            for key in a.files:
                x, y = a[key], b[key]
                if x.dtype != y.dtype or x.shape != y.shape or x.tobytes() != y.tobytes():
                    problems.append((name, f"array '{key}' differs"))
    return problems


if __name__ == "__main__":
    if "--check" in sys.argv:
        bad = check()
        for name, what in bad:
            print(f"MISMATCH {name}: {what}")
        print("golden fixtures reproduced bit for bit" if not bad else f"{len(bad)} differences")
        raise SystemExit(1 if bad else 0)
    if "--seeded" in sys.argv:   # only the seeded fixtures (the stored ones above stay byte-identical)
        main_seeded()
    else:
        main()
        main_seeded()
