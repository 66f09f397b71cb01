"""CPU tests: the oracle against the committed golden vectors and against its own
independent restatements (float64 exact, float32 C small-batch, float32 BLAS-form)."""
import glob
import os

import numpy as np
import pytest

from oracle import flat_oracle as fo
from oracle import knn_oracle as ko
from tests import keycodec as kc
from tests.knn_checks import assert_knn_matches, load_fixture

L2, IP = ko.METRIC_L2, ko.METRIC_INNER_PRODUCT
GOLDEN = sorted(g for g in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(g).startswith("normalize"))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(g)[:-4] for g in GOLDEN])
def test_numpy_oracle_reproduces_golden(path):
    z = load_fixture(path)
    D, I = ko.knn_exact(z["xb"], z["xq"], int(z["k"]), int(z["metric"]))
    assert np.array_equal(I, z["I"]) and np.array_equal(D, z["D"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(g)[:-4] for g in GOLDEN])
def test_c_restatement_matches_golden(path):
    z = load_fixture(path)
    xb, xq, k, metric = z["xb"], z["xq"], int(z["k"]), int(z["metric"])
    for nthreads in (0, 3):  # Faiss's own per-query scheme, and the slab-split one
        D, I, _ = fo.knn_flat(xb, xq, k, metric, nthreads)
        assert_knn_matches(D, I, z["D"], z["I"], xb, xq, metric, gap=z["gap"])


def test_blas_form_matches_exact():
    rng = np.random.default_rng(1)
    xb = rng.random((5000, 64), dtype=np.float32)
    xq = rng.random((33, 64), dtype=np.float32)
    for metric in (L2, IP):
        D, I = ko.knn_blas_f32(xb, xq, 10, metric)
        Dr, Ir = ko.knn_exact(xb, xq, 10, metric)
        assert_knn_matches(D, I, Dr, Ir, xb, xq, metric, gap=ko.kth_gap(xb, xq, 10, metric))


def test_dict_branch_semantics():
    """backend/siamese/test_index.py:58-69: normalise, per-row Euclidean distance,
    ascending argsort.  Same ranking as squared L2; values are its square root."""
    rng = np.random.default_rng(2)
    index = rng.standard_normal((300, 128))
    index /= np.linalg.norm(index, axis=1, keepdims=True)  # create_index.py:62-85 (float64 unit rows)
    emb = rng.standard_normal((1, 128))
    emb = emb / np.linalg.norm(emb)
    dist = np.array([np.linalg.norm(index[i, :] - emb) for i in range(len(index))])
    order = dist.argsort()[:9]
    D, I = ko.knn_exact(index.astype(np.float32), emb.astype(np.float32), 9, L2)
    assert np.array_equal(I[0], order)
    np.testing.assert_allclose(np.sqrt(D[0]), dist[order], rtol=1e-5)


def test_strict_gate_and_padding():
    xb = np.array([[0.0], [np.nan], [np.inf], [3e38]], np.float32)
    xq = np.array([[0.0]], np.float32)
    D, I = ko.knn_exact(xb, xq, 4, L2)  # (3e38)^2 = inf in f32 -> >= FLT_MAX, never enters
    assert I.tolist() == [[0, -1, -1, -1]] and D[0, 1] == ko.FLT_MAX
    Dc, Ic, _ = fo.knn_flat(xb, xq, 4, L2)
    assert np.array_equal(Ic, I) and np.array_equal(Dc, D)
    D, I = ko.knn_exact(np.zeros((0, 4), np.float32), np.zeros((2, 4), np.float32), 3, IP)
    assert (I == -1).all() and (D == -ko.FLT_MAX).all()


def test_normalize_oracles_agree(golden_dir):
    z = np.load(os.path.join(golden_dir, "normalize_n9_d100.npz"))
    assert np.array_equal(ko.normalize_rows(z["x"]), z["y"])
    y = z["x"].copy()
    fo.renorm_L2(y)
    np.testing.assert_allclose(y, z["y"], rtol=2e-6, atol=1e-7)
    assert np.array_equal(y[3], np.zeros(100, np.float32))
    np.testing.assert_allclose(np.linalg.norm(np.delete(y, 3, 0), axis=1), 1.0, rtol=1e-6)


def test_assignment_ip_equals_l2_on_unit_centroids(golden_dir):
    a = np.load(os.path.join(golden_dir, "assign_ip_n512_c256_d128.npz"))
    b = np.load(os.path.join(golden_dir, "assign_l2_n512_c256_d128.npz"))
    assert np.array_equal(a["I"], b["I"])  # quirk 5.9-7: arg-max IP == arg-min L2 for unit centroids
    assert np.array_equal(ko.assign_nearest(a["xb"], a["xb"], IP).ravel(), np.arange(256))


def test_shard_merge_equals_unsharded():
    rng = np.random.default_rng(3)
    xb = rng.random((999, 24), dtype=np.float32)
    xb[500] = xb[10]  # cross-shard duplicate: the tie must resolve to the lower global id
    xq = np.concatenate([rng.random((4, 24), dtype=np.float32), xb[10:11]])
    for metric in (L2, IP):
        Dp, Ip = [], []
        for r in range(5):
            lo, hi = 999 * r // 5, 999 * (r + 1) // 5
            d_, i_ = ko.knn_exact(xb[lo:hi], xq, 7, metric, id_offset=lo)
            Dp.append(d_)
            Ip.append(i_)
        D, I = ko.merge_shards(Dp, Ip, 7, metric)
        Dr, Ir = ko.knn_exact(xb, xq, 7, metric)
        assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
        # the packed-key image of the same merge (what travels in the all-gather)
        keys = np.stack([kc.encode(d_, i_, metric) for d_, i_ in zip(Dp, Ip)])  # (G, nq, k)
        merged = np.sort(keys.transpose(1, 0, 2).reshape(len(xq), -1), axis=1)[:, :7]
        Dk, Ik = kc.decode(merged, metric)
        assert np.array_equal(Ik, Ir) and np.array_equal(Dk, Dr)


def test_key_codec_is_order_preserving():
    v = np.array([-np.inf, -3e38, -1.5, -0.0, 0.0, 1e-30, 2.0, 3e38, np.inf], np.float32)
    o = kc.ord_f32(v).astype(np.int64)
    assert (np.diff(o) >= 0).all() and (np.diff(o)[[0, 1, 2, 4, 5, 6, 7]] > 0).all()
    assert np.array_equal(kc.unord_f32(kc.ord_f32(v)).view(np.uint32), v.view(np.uint32))


def test_lloyd_reference_against_sklearn():
    """The oracle's k-means iteration (what the GPU trainer is compared with) against an independent
    implementation: sklearn's Lloyd from the same initial centroids, the same number of iterations."""
    pytest.importorskip("sklearn")
    from sklearn.cluster import KMeans

    rng = np.random.default_rng(3)
    K, d, per = 9, 12, 120
    centres = rng.standard_normal((K, d)) * 6.0
    x = np.concatenate([centres[i] + rng.standard_normal((per, d)) for i in range(K)]).astype(np.float32)
    c0 = (centres + rng.standard_normal((K, d)) * 1.5).astype(np.float32)
    for niter in (1, 4):
        c, obj, lab = ko.lloyd_reference(x, c0, niter)
        sk = KMeans(n_clusters=K, init=c0.astype(np.float64), n_init=1, max_iter=niter, tol=0.0, algorithm="lloyd")
        sk.fit(x.astype(np.float64))
        np.testing.assert_allclose(c, sk.cluster_centers_, rtol=1e-9, atol=1e-9)
        assert len(obj) == niter and all(a >= b - 1e-9 for a, b in zip(obj, obj[1:]))   # never increases
    # spherical: unit centroids, objective = summed inner products, never decreases
    cs, objs, _ = ko.lloyd_reference(x, c0, 5, spherical=True)
    np.testing.assert_allclose(np.linalg.norm(cs, axis=1), 1.0, rtol=1e-12)
    assert all(b >= a - 1e-9 for a, b in zip(objs, objs[1:]))
    with pytest.raises(ValueError, match="empty cluster"):
        ko.lloyd_reference(x, np.concatenate([c0[:-1], c0[-1:] * 0 + 1e6]), 2)


# ---- standing independence checks.  The reference holds no fixture and Faiss is not
# importable here, so the oracle stays "parity unpinned"; what CAN be had in this container is agreement
# with two implementations that share no code with oracle/: sklearn's brute-force NearestNeighbors
# (sqeuclidean) and torch's float64 cdist / matmul + topk.  They run on every golden and on a fresh random
# draw per test run (the seed is printed on failure), against both oracles.
def _independent_topk(xb, xq, k, metric):
    """(ids by torch float64, ids by sklearn float64 or None, float64 scores of the torch ids)"""
    import torch

    tb, tq = torch.from_numpy(xb).double(), torch.from_numpy(xq).double()
    if metric == L2:
        from sklearn.neighbors import NearestNeighbors

        S = torch.cdist(tq, tb, compute_mode="donot_use_mm_for_euclid_dist") ** 2
        v, i = torch.topk(S, k, dim=1, largest=False)
        nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric="sqeuclidean").fit(xb.astype(np.float64))
        _, si = nn.kneighbors(xq.astype(np.float64))
        return i.numpy(), si, v.numpy()
    S = tq @ tb.T
    v, i = torch.topk(S, k, dim=1, largest=True)
    return i.numpy(), None, v.numpy()


def _assert_oracles_agree_with_independent(xb, xq, k, metric, tag):
    if xb.shape[0] < k:  # padding cases: the cross-checkers have no notion of -1 / FLT_MAX padding
        return
    gap = ko.kth_gap(xb, xq, k, metric)
    clear = gap > 1e-9  # exact float64 ties (duplicate rows) may be ordered differently by the cross-checkers
    ti, si, tv = _independent_topk(xb, xq, k, metric)
    D, I = ko.knn_exact(xb, xq, k, metric)
    assert np.array_equal(ti[clear], I[clear]), f"{tag}: torch float64 disagrees with knn_exact"
    if si is not None:
        assert np.array_equal(si[clear], I[clear]), f"{tag}: sklearn brute force disagrees with knn_exact"
    np.testing.assert_allclose(tv[clear], D[clear], rtol=1e-6, atol=1e-6, err_msg=tag)
    # on tied queries the SET of distances must still agree
    np.testing.assert_allclose(np.sort(tv, axis=1), np.sort(D.astype(np.float64), axis=1), rtol=1e-6, atol=1e-6,
                               err_msg=tag)
    # the float32 C restatement: ids equal where float32 resolves the ranks, near-ties otherwise
    for nthreads in (0, 3):
        Dc, Ic, _ = fo.knn_flat(xb, xq, k, metric, nthreads)
        assert_knn_matches(Dc, Ic, tv.astype(np.float32), ti.astype(np.int64), xb, xq, metric, gap=gap)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(g)[:-4] for g in GOLDEN])
def test_oracles_against_sklearn_and_torch_on_goldens(path):
    pytest.importorskip("sklearn")
    z = load_fixture(path)
    _assert_oracles_agree_with_independent(z["xb"], z["xq"], int(z["k"]), int(z["metric"]), os.path.basename(path))


def test_oracles_against_sklearn_and_torch_on_a_fresh_draw():
    pytest.importorskip("sklearn")
    seed = int.from_bytes(os.urandom(4), "little")
    rng = np.random.default_rng(seed)
    for case in range(6):
        n, d = int(rng.integers(50, 3000)), int(rng.integers(1, 300))
        nq, k = int(rng.integers(1, 40)), int(rng.integers(1, min(n, 40) + 1))
        kind = case % 3
        if kind == 0:
            xb, xq = rng.random((n, d), dtype=np.float32), rng.random((nq, d), dtype=np.float32)
        elif kind == 1:  # offset + spread: the CNN-embedding shape of the data (large common component)
            xb = (50.0 + rng.standard_normal((n, d))).astype(np.float32)
            xq = (50.0 + rng.standard_normal((nq, d))).astype(np.float32)
        else:            # unit rows (the "cosine" index of backend/utils.py:300-303)
            xb = ko.normalize_rows(rng.standard_normal((n, d)).astype(np.float32))
            xq = ko.normalize_rows(rng.standard_normal((nq, d)).astype(np.float32))
        for metric in (L2, IP):
            _assert_oracles_agree_with_independent(xb, xq, k, metric, f"seed {seed} case {case} metric {metric} "
                                                   f"n={n} d={d} nq={nq} k={k}")


def test_make_golden_check_reproduces_every_committed_fixture(golden_dir):
    """tests/golden/make_golden.py --check: every committed .npz comes out of the generator bit for bit
    (array by array), none missing, none left over -- the fixtures are what the script says they are."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    assert mg.check(verbose=False) == []


def test_c_oracle_under_address_and_ub_sanitizers(tmp_path):
    """The C restatement (oracle/flat_oracle.c) is the checker of every GPU parity test: a driver runs its entry
    points over ragged shapes (d not a multiple of 8, k larger than the index, an empty index, one row, ties,
    several OpenMP threads) in a build with AddressSanitizer and UndefinedBehaviorSanitizer; any report fails."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no host C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = tmp_path / "driver.c"
    drv.write_text(r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <float.h>
int oracle_knn_flat(const float*, int64_t, int, const float*, int64_t, int, int, float*, int64_t*, int);
void oracle_renorm_L2(size_t, size_t, float*);
float oracle_fvec_L2sqr(const float*, const float*, size_t);
float oracle_fvec_inner_product(const float*, const float*, size_t);
static unsigned s = 12345u;
static float rnd(void) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; }
int main(void) {
    const int shapes[][4] = {{0, 5, 3, 4}, {1, 1, 1, 1}, {1, 7, 2, 5}, {17, 3, 4, 20}, {100, 13, 5, 10},
                             {257, 64, 16, 7}, {1000, 37, 3, 1}, {64, 8, 64, 64}, {33, 100, 1, 40}};
    for (unsigned c = 0; c < sizeof(shapes) / sizeof(shapes[0]); c++) {
        const int n = shapes[c][0], d = shapes[c][1], nq = shapes[c][2], k = shapes[c][3];
        float* xb = malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * d);
        float* xq = malloc(sizeof(float) * (size_t)nq * d);
        float* D = malloc(sizeof(float) * (size_t)nq * k);
        int64_t* I = malloc(sizeof(int64_t) * (size_t)nq * k);
        for (int i = 0; i < n * d; i++) xb[i] = rnd();
        for (int i = 0; i < nq * d; i++) xq[i] = rnd();
        if (n > 3) for (int j = 0; j < d; j++) xb[(n - 1) * d + j] = xb[j];  /* a duplicate row: a tie */
        for (int metric = 0; metric <= 1; metric++)
            for (int nt = 1; nt <= 3; nt += 2) {
                oracle_knn_flat(xb, n, d, xq, nq, k, metric, D, I, nt);
                for (int q = 0; q < nq; q++)
                    for (int r = 0; r < k; r++) {
                        const int64_t id = I[q * k + r];
                        if (r < n ? (id < 0 || id >= n) : id != -1) { printf("bad id %d %d\n", c, r); return 1; }
                    }
            }
        if (n > 0) {
            oracle_renorm_L2((size_t)d, (size_t)n, xb);
            (void)oracle_fvec_L2sqr(xb, xq, (size_t)d);
            (void)oracle_fvec_inner_product(xb, xq, (size_t)d);
        }
        free(xb); free(xq); free(D); free(I);
    }
    puts("driver ok");
    return 0;
}
''')
    exe = tmp_path / "oracle_san"
    r = subprocess.run([gcc, "-g", "-O1", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", os.path.join(root, "oracle", "flat_oracle.c"), str(drv), "-lm", "-o", str(exe)],
                       capture_output=True, text=True)
    if r.returncode != 0 and ("lasan" in r.stderr or "lubsan" in r.stderr or "libasan" in r.stderr):
        pytest.skip("the sanitizer runtimes are not installed")
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", OMP_NUM_THREADS="3")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "driver ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, \
        (r.stdout[-500:], r.stderr[-2000:])
