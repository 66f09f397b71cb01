"""GPU parity tests: HIP kNN (through the C ABI) vs the CPU oracle.
Run on the MI355X box with:  python -m pytest tests -m gpu
Parity is UNPINNED with respect to a real Faiss build (see oracle/knn_oracle.py):
the expected values come from the exact float64 oracle and the committed
golden fixtures it generated."""
import glob
import os
import threading

import numpy as np
import pytest

from oracle import knn_oracle as ko
from tests.knn_checks import assert_knn_matches, load_fixture

pytestmark = pytest.mark.gpu

L2, IP = ko.METRIC_L2, ko.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def faiss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import image_search_engine_amd.faiss_compat as fc

    return fc


def make_index(faiss, metric, d):
    return faiss.IndexFlatL2(d) if metric == L2 else faiss.IndexFlatIP(d)


GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GOLDEN = [g for g in GOLDEN if not os.path.basename(g).startswith("normalize")]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(g)[:-4] for g in GOLDEN])
def test_golden_fixture(faiss, path):
    z = load_fixture(path)
    xb, xq, k, metric = z["xb"], z["xq"], int(z["k"]), int(z["metric"])
    index = make_index(faiss, metric, xb.shape[1])
    index.add(xb)
    assert index.ntotal == xb.shape[0]
    D, I = index.search(xq, k)
    n_mism = assert_knn_matches(D, I, z["D"], z["I"], xb, xq, metric, gap=z["gap"])
    if (z["gap"] == 0).all() or (z["gap"] > 1e-4).all():
        assert n_mism == 0


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("n,d,nq,k", [
    (10_000, 128, 16, 5),     # BASELINE config 1 shape
    (5_000, 512, 7, 10),
    (20_000, 64, 40, 20),     # 3 query tiles, k = NUM_IMAGES_TO_RETURN
    (3_000, 2048, 3, 10),     # ResNet-50 flatten width (backend/descriptors.py:166-168)
    (1_001, 48, 1, 32),       # ragged row count, one query, k at the single-pass limit
    (4_097, 96, 17, 1),       # k = 1
])
def test_random_vs_oracle(faiss, metric, n, d, nq, k):
    rng = np.random.default_rng(n + d + nq + k + metric)
    xb = rng.random((n, d), dtype=np.float32)
    xq = rng.random((nq, d), dtype=np.float32)
    index = make_index(faiss, metric, d)
    index.add(xb)
    D, I = index.search(xq, k)
    D_ref, I_ref = ko.knn_exact(xb, xq, k, metric)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, metric, gap=ko.kth_gap(xb, xq, k, metric))


@pytest.mark.parametrize("k", [29, 33, 64, 100, 257, 2048])
def test_large_k_multipass(faiss, k):
    rng = np.random.default_rng(k)
    xb = rng.random((3000, 40), dtype=np.float32)
    xq = rng.random((5, 40), dtype=np.float32)
    index = faiss.IndexFlatL2(40)
    index.add(xb)
    D, I = index.search(xq, k)
    D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2)


def test_against_c_restatement(faiss):
    """The float32 C restatement of Faiss's small-batch algorithm agrees too."""
    from oracle import flat_oracle as fo

    rng = np.random.default_rng(5)
    xb = rng.random((8000, 128), dtype=np.float32)
    xq = rng.random((9, 128), dtype=np.float32)
    for metric in (L2, IP):
        index = make_index(faiss, metric, 128)
        index.add(xb)
        D, I = index.search(xq, 10)
        Dc, Ic, _ = fo.knn_flat(xb, xq, 10, metric)
        assert_knn_matches(D, I, Dc, Ic, xb, xq, metric)


def test_incremental_add_and_reset(faiss):
    rng = np.random.default_rng(11)
    xb = rng.random((1500, 100), dtype=np.float32)
    xq = rng.random((4, 100), dtype=np.float32)
    index = faiss.IndexFlatL2(100)
    for lo, hi in ((0, 1), (1, 700), (700, 1500)):
        index.add(xb[lo:hi])
    assert index.ntotal == 1500
    D, I = index.search(xq, 8)
    D_ref, I_ref = ko.knn_exact(xb, xq, 8, L2)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2)
    assert np.array_equal(index.reconstruct_n(0, 1500), xb)
    index.reset()
    assert index.ntotal == 0
    D, I = index.search(xq, 3)
    assert (I == -1).all() and (D == np.finfo(np.float32).max).all()


def test_add_copies_and_matrix_input(faiss):
    rng = np.random.default_rng(3)
    xb = rng.random((300, 20), dtype=np.float32)
    keep = xb.copy()
    index = faiss.IndexFlatL2(20)
    index.add(np.matrix(xb))  # np.matrix accepted (backend/engine.py:96 passes .todense())
    xb[:] = 0  # the index owns its copy
    xq = np.matrix(rng.random((2, 20), dtype=np.float32))
    D, I = index.search(xq, 4)
    D_ref, I_ref = ko.knn_exact(keep, np.asarray(xq), 4, L2)
    assert_knn_matches(D, I, D_ref, I_ref, keep, np.asarray(xq, dtype=np.float32), L2)


def test_shape_errors(faiss):
    index = faiss.IndexFlatL2(16)
    with pytest.raises(AssertionError):
        index.add(np.zeros((3, 15), np.float32))
    index.add(np.zeros((3, 16), np.float32))
    with pytest.raises(AssertionError):
        index.search(np.zeros((1, 17), np.float32), 2)
    with pytest.raises(AssertionError):
        index.search(np.zeros((1, 16), np.float32), 0)
    with pytest.raises(RuntimeError):
        index.search(np.zeros((1, 16), np.float32), 5000)


def test_nan_and_inf_rows_never_returned(faiss):
    rng = np.random.default_rng(8)
    xb = rng.random((50, 16), dtype=np.float32)
    xb[7, 3] = np.nan
    xb[9, 0] = np.inf
    xq = rng.random((2, 16), dtype=np.float32)
    for metric in (L2, IP):
        index = make_index(faiss, metric, 16)
        index.add(xb)
        D, I = index.search(xq, 50)
        D_ref, I_ref = ko.knn_exact(xb, xq, 50, metric)
        # NaN scores never enter; an L2 distance of inf never enters; an inner product
        # of +inf does (it is strictly better than -FLT_MAX), as in Faiss's heap
        assert 7 not in I and (I[:, -1] == -1).all()
        assert np.array_equal(I < 0, I_ref < 0)
        for q in range(2):
            assert sorted(I[q].tolist()) == sorted(I_ref[q].tolist())


def test_normalize_L2(faiss, golden_dir):
    z = np.load(os.path.join(golden_dir, "normalize_n9_d100.npz"))
    x = z["x"].copy()
    assert faiss.normalize_L2(x) is None
    assert np.array_equal(x[3], np.zeros(100, np.float32))  # zero row untouched
    np.testing.assert_allclose(x, z["y"], rtol=2e-6, atol=1e-7)
    from oracle import flat_oracle as fo

    y = z["x"].copy()
    fo.renorm_L2(y)
    np.testing.assert_allclose(x, y, rtol=2e-6, atol=1e-7)
    with pytest.raises(AssertionError):
        faiss.normalize_L2(z["x"].astype(np.float64))


def test_cosine_is_ip_on_normalised(faiss):
    """create_search_index('cosine') semantics (backend/utils.py:300-303)."""
    rng = np.random.default_rng(21)
    xb = rng.standard_normal((2000, 128)).astype(np.float32)
    xq = rng.standard_normal((5, 128)).astype(np.float32)
    faiss.normalize_L2(xb)
    faiss.normalize_L2(xq)
    index = faiss.IndexFlatIP(128)
    index.add(xb)
    D, I = index.search(xq, 9)
    D_ref, I_ref = ko.knn_exact(xb, xq, 9, IP)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, IP)
    assert (D <= 1.0 + 1e-5).all()


def test_write_read_roundtrip(faiss, tmp_path):
    rng = np.random.default_rng(2)
    xb = rng.random((777, 33), dtype=np.float32)
    xq = rng.random((3, 33), dtype=np.float32)
    for metric in (L2, IP):
        index = make_index(faiss, metric, 33)
        index.add(xb)
        p = tmp_path / f"idx{metric}.faiss"
        faiss.write_index(index, str(p))
        back = faiss.read_index(str(p))
        assert (back.d, back.ntotal, back.metric_type) == (33, 777, metric)
        D0, I0 = index.search(xq, 6)
        D1, I1 = back.search(xq, 6)
        assert np.array_equal(I0, I1) and np.array_equal(D0, D1)


def test_concurrent_search_threads(faiss):
    """Flask request threads may enter index.search concurrently (backend/engine.py:137)."""
    rng = np.random.default_rng(4)
    xb = rng.random((20000, 64), dtype=np.float32)
    index = faiss.IndexFlatL2(64)
    index.add(xb)
    qs = [rng.random((3, 64), dtype=np.float32) for _ in range(8)]
    refs = [ko.knn_exact(xb, q, 5, L2) for q in qs]
    out = [None] * 8

    def work(i):
        for _ in range(5):
            out[i] = index.search(qs[i], 5)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(8):
        assert_knn_matches(out[i][0], out[i][1], refs[i][0], refs[i][1], xb, qs[i], L2)


def test_torch_device_api_and_determinism(faiss):
    import torch

    rng = np.random.default_rng(6)
    xb = rng.random((30000, 256), dtype=np.float32)
    xq = rng.random((16, 256), dtype=np.float32)
    index = faiss.IndexFlatL2(256)
    index.add_torch(torch.from_numpy(xb).cuda())
    tq = torch.from_numpy(xq).cuda()
    D1, I1 = index.search_torch(tq, 10)
    D2, I2 = index.search_torch(tq, 10)
    torch.cuda.synchronize()
    assert torch.equal(I1, I2) and torch.equal(D1, D2)  # bitwise reproducible
    D_ref, I_ref = ko.knn_exact(xb, xq, 10, L2)
    assert_knn_matches(D1.cpu().numpy(), I1.cpu().numpy(), D_ref, I_ref, xb, xq, L2)
    Dh, Ih = index.search(xq, 10)
    assert np.array_equal(Ih, I1.cpu().numpy()) and np.array_equal(Dh, D1.cpu().numpy())


def test_shard_keys_merge_equals_unsharded(faiss):
    """Row shards + packed-key merge == one index, single process."""
    import torch

    rng = np.random.default_rng(9)
    xb = rng.random((12345, 96), dtype=np.float32)
    xq = rng.random((20, 96), dtype=np.float32)
    tq = torch.from_numpy(xq).cuda()
    for metric in (L2, IP):
        whole = make_index(faiss, metric, 96)
        whole.add(xb)
        D0, I0 = whole.search(xq, 10)
        keys = []
        G = 4
        for r in range(G):
            lo, hi = 12345 * r // G, 12345 * (r + 1) // G
            sh = make_index(faiss, metric, 96)
            sh.add(xb[lo:hi])
            keys.append(sh.search_keys_torch(tq, 10, id_base=lo))
        D1, I1 = faiss.merge_keys_torch(torch.stack(keys), metric)
        assert np.array_equal(I0, I1.cpu().numpy())
        assert np.array_equal(D0, D1.cpu().numpy())


def test_full_size_properties(faiss):
    """BASELINE size (1M x 512, k = 10): size-independent properties instead of a
    full oracle pass: planted duplicates of the queries come back first at distance
    ~0, results are sorted, ids unique, and a 2-shard merge reproduces them."""
    import torch

    n, d, nq, k = 1_000_000, 512, 16, 10
    g = torch.Generator(device="cuda").manual_seed(1234)
    xb = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float32)
    xq = torch.rand((nq, d), generator=g, device="cuda", dtype=torch.float32)
    plant = torch.arange(nq, device="cuda") * 60_001 + 17
    xb[plant] = xq
    index = faiss.IndexFlatL2(d)
    index.add_torch(xb)
    D, I = index.search_torch(xq, k)
    torch.cuda.synchronize()
    Dn, In = D.cpu().numpy(), I.cpu().numpy()
    assert np.array_equal(In[:, 0], plant.cpu().numpy())
    assert (Dn[:, 0] <= 1e-3).all()
    assert (np.diff(Dn, axis=1) >= 0).all()
    assert all(len(set(r)) == k for r in In.tolist())
    # exact check of the reported distances against float64 on the returned rows
    rows = xb[I.reshape(-1)].double().reshape(nq, k, d)
    ref = ((rows - xq.double()[:, None, :]) ** 2).sum(-1).cpu().numpy()
    assert (np.abs(Dn - ref) <= 1e-4 * np.maximum(1, ref)).all()
    # every other row is no closer than the k-th (checked on a strided sample in float64)
    samp = xb[::97].double()
    ds = torch.cdist(xq.double(), samp).pow(2).cpu().numpy()
    kth = ref[:, -1:] * (1 + 1e-6)
    ids = np.arange(0, n, 97)
    for q in range(nq):
        closer = ids[ds[q] < ref[q, -1] * (1 - 1e-6)]
        assert set(closer.tolist()) <= set(In[q].tolist())
    del samp, ds, kth
    # two shards + merge
    half = n // 2
    a, b = faiss.IndexFlatL2(d), faiss.IndexFlatL2(d)
    a.add_torch(xb[:half])
    b.add_torch(xb[half:])
    keys = torch.stack([a.search_keys_torch(xq, k, 0), b.search_keys_torch(xq, k, half)])
    D2, I2 = faiss.merge_keys_torch(keys, L2)
    assert torch.equal(I2, I) and torch.equal(D2, D)


def _bf16_round(x):
    import torch

    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("metric", [IP, L2])
@pytest.mark.parametrize("n,d,nq,k", [(20_000, 512, 16, 10), (5_000, 100, 40, 5), (3_000, 2048, 3, 20)])
def test_bf16_storage_matches_oracle_on_rounded_data(faiss, metric, n, d, nq, k):
    """BASELINE config 5 path: rows and queries rounded to bf16, fp32 accumulation.  Against the
    exact oracle evaluated on the SAME rounded values only the summation order differs."""
    rng = np.random.default_rng(n + d + k + metric)
    xb = rng.standard_normal((n, d)).astype(np.float32)
    xq = rng.standard_normal((nq, d)).astype(np.float32)
    if metric == IP:  # cosine: normalised rows (backend/utils.py:300-303)
        xb /= np.linalg.norm(xb, axis=1, keepdims=True)
        xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    index = faiss.IndexFlat(d, metric, storage="bf16")
    index.add(xb)
    D, I = index.search(xq, k)
    xb_r, xq_r = _bf16_round(xb), _bf16_round(xq)
    D_ref, I_ref = ko.knn_exact(xb_r, xq_r, k, metric)
    assert_knn_matches(D, I, D_ref, I_ref, xb_r, xq_r, metric, gap=ko.kth_gap(xb_r, xq_r, k, metric))
    assert np.array_equal(index.reconstruct_n(0, 50), xb_r[:50])  # the index holds the rounded rows
    # recall against the unrounded exact answer is what config 5 reports
    _, I_exact = ko.knn_exact(xb, xq, k, metric)
    recall = np.mean([len(set(I[q]) & set(I_exact[q])) / k for q in range(nq)])
    assert recall >= 0.7, recall


@pytest.mark.parametrize("metric", [IP, L2])
@pytest.mark.parametrize("n,K,d", [(5000, 256, 128), (4099, 1000, 32), (3001, 200, 64), (2500, 77, 100), (2100, 4096, 128),
                                   (2048, 130, 512)])
def test_assignment_kernel_matches_oracle(faiss, metric, n, K, d):
    """k = 1 nearest-centroid assignment (backend/kmeans_faiss.py:46-50, BASELINE config 4 shape):
    SIFT-valued rows against unit-norm centroids through the dedicated MFMA kernel."""
    import torch

    rng = np.random.default_rng(n + K + d + metric)
    cent = ko.normalize_rows(rng.standard_normal((K, d)).astype(np.float32))
    X = rng.integers(0, 256, (n, d)).astype(np.float32)
    index = make_index(faiss, metric, d)
    index.add(cent)
    assert index._assign_applies(n, 1)
    D, I = index.search(X, 1)  # routed to the assignment kernel (n >= ASSIGN_MIN_NQ)
    D_ref, I_ref = ko.knn_exact(cent, X, 1, metric)
    assert_knn_matches(D, I, D_ref, I_ref, cent, X, metric, gap=ko.kth_gap(cent, X, 1, metric), rtol=2e-4)
    # the general scan path agrees (it is what smaller batches use)
    Ds, Is = index.search(X[:64], 1)
    assert_knn_matches(Ds, Is, D_ref[:64], I_ref[:64], cent, X[:64], metric, rtol=2e-4)
    # duplicate centroids: the lower id wins
    dup = make_index(faiss, metric, d)
    dup.add(np.concatenate([cent[:5], cent[:5]]))
    Dd, Id = dup.assign_torch(torch.from_numpy(X[:300]).cuda())
    assert (Id.cpu().numpy() < 5).all()


def test_l2_on_offset_data_like_cnn_embeddings(faiss):
    """Rows with a large common component and a small spread (post-ReLU CNN features):
    the expanded L2 form loses the neighbours to cancellation unless distances are taken
    around a shift vector.  Faiss's small-batch path (direct differences) is exact here."""
    rng = np.random.default_rng(77)
    n, d = 20000, 512
    xb = (40.0 + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
    xq = (xb[rng.integers(0, n, 16)] + 0.01 * rng.standard_normal((16, d))).astype(np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    assert np.abs(index.get_shift() - 40.0).max() < 0.01
    D, I = index.search(xq, 10)
    D_ref, I_ref = ko.knn_exact(xb, xq, 10, L2)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2, gap=ko.kth_gap(xb, xq, 10, L2))
    assert np.array_equal(index.reconstruct_n(0, 100), xb[:100])  # rows are stored unshifted
    # k = 1 assignment kernel on the same kind of data
    X = (40.0 + 0.05 * rng.standard_normal((4096, d))).astype(np.float32)
    small = faiss.IndexFlatL2(d)
    small.add(xb[:300])
    Da, Ia = small.search(X, 1)
    Dr, Ir = ko.knn_exact(xb[:300], X, 1, L2)
    assert_knn_matches(Da, Ia, Dr, Ir, xb[:300], X, L2, gap=ko.kth_gap(xb[:300], X, 1, L2))


def test_shape_sweep_against_oracle(faiss):
    """Many small (n, d, nq, k) combinations: odd d (scalar query staging), every query-tile
    count (T = 1, 2, 3 and ragged last tiles), k across the single-pass limit, both metrics."""
    rng = np.random.default_rng(2024)
    cases = []
    for d in (1, 3, 17, 33, 63, 65, 130, 257, 600, 1024):
        for nq in (1, 15, 17, 31, 33, 47, 49, 100):
            cases.append((int(rng.integers(20, 3000)), d, nq, int(rng.integers(1, 40))))
    rng.shuffle(cases)
    for n, d, nq, k in cases[:48]:
        metric = L2 if (n + d + nq) % 2 else IP
        xb = rng.random((n, d), dtype=np.float32)
        xq = rng.random((nq, d), dtype=np.float32)
        index = make_index(faiss, metric, d)
        index.add(xb)
        D, I = index.search(xq, k)
        D_ref, I_ref = ko.knn_exact(xb, xq, k, metric)
        try:
            assert_knn_matches(D, I, D_ref, I_ref, xb, xq, metric, gap=ko.kth_gap(xb, xq, k, metric))
        except AssertionError as e:
            raise AssertionError(f"n={n} d={d} nq={nq} k={k} metric={metric}: {e}")


@pytest.mark.parametrize("nq,k", [(40, 100), (20, 33), (50, 70)])
def test_large_k_with_several_query_tiles(faiss, nq, k):
    """k > 32 (floor-keyed passes) combined with 2 and 3 query tiles per pass."""
    rng = np.random.default_rng(nq * k)
    xb = rng.random((6000, 72), dtype=np.float32)
    xq = rng.random((nq, 72), dtype=np.float32)
    for metric in (L2, IP):
        index = make_index(faiss, metric, 72)
        index.add(xb)
        D, I = index.search(xq, k)
        D_ref, I_ref = ko.knn_exact(xb, xq, k, metric)
        assert_knn_matches(D, I, D_ref, I_ref, xb, xq, metric, gap=ko.kth_gap(xb, xq, k, metric))


def test_streams_and_threads_stress(faiss):
    """Several host threads, each on its own HIP stream, hammer one index through the device API:
    the event-ordered workspace slots must keep every batch's result intact."""
    import torch

    rng = np.random.default_rng(31)
    xb = rng.random((50_000, 128), dtype=np.float32)
    index = faiss.IndexFlatL2(128)
    index.add(xb)
    qs = [rng.random((nq, 128), dtype=np.float32) for nq in (1, 16, 17, 33, 48, 64)]
    refs = [ko.knn_exact(xb, q, 7, L2) for q in qs]
    errors = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            tq = torch.from_numpy(qs[i]).cuda()
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                for _ in range(30):
                    D, I = index.search_torch(tq, 7)
            st.synchronize()
            assert_knn_matches(D.cpu().numpy(), I.cpu().numpy(), refs[i][0], refs[i][1], xb, qs[i], L2)
        except Exception as e:  # surfaced in the main thread
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(qs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors


def test_threshold_exchange_on_off_and_tag_wrap(faiss, monkeypatch):
    """nq = 48 runs the three-tile kernel with the one-shot threshold exchange between blocks.
    Its sequence tags start just below the wrap here (test knob), so the searches below cross the
    point where the tags are wiped and restart; every result must stay identical to the oracle and
    to itself."""
    from oracle import flat_oracle as fo

    monkeypatch.setenv("ISE_XCHG_SEQ_START", str(0xFFFFFFF0 - 5))
    rng = np.random.default_rng(99)
    n, d, nq, k = 300_000, 96, 48, 10  # enough row tiles per wave for the exchange to run
    xb = rng.random((n, d), dtype=np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    Dr = Ir = None
    first = None
    for it in range(12):
        xq = rng.random((nq, d), dtype=np.float32) if it % 3 == 0 else xq  # noqa: F821
        D, I = index.search(xq, k)
        if it % 3 == 0:
            Dr, Ir, _ = fo.knn_flat(xb, xq, k, 1, 8)
            first = (D.copy(), I.copy())
        assert_knn_matches(D, I, Dr, Ir, xb, xq, 1)
        assert np.array_equal(I, first[1]) and np.array_equal(D, first[0])  # bitwise repeatable
    # large k (floor-keyed passes) through the same kernels
    D, I = index.search(xq, 70)
    Dr, Ir, _ = fo.knn_flat(xb, xq, 70, 1, 8)
    assert_knn_matches(D, I, Dr, Ir, xb, xq, 1)


def test_shards_with_three_query_tiles_equal_unsharded(faiss):
    """48 queries per pass (the kernel with the threshold exchange), shards large enough for the
    exchange to run, global ids through id_base: merged shards == one index, bit for bit."""
    import torch

    rng = np.random.default_rng(21)
    n, d, nq, k = 400_000, 64, 100, 10
    xb = rng.random((n, d), dtype=np.float32)
    xb[n - 7] = xb[11]                       # tie across shards: lower global id first
    xq = rng.random((nq, d), dtype=np.float32)
    xq[0] = xb[11]
    tq = torch.from_numpy(xq).cuda()
    for metric in (L2, IP):
        whole = make_index(faiss, metric, d)
        whole.add(xb)
        D0, I0 = whole.search(xq, k)
        if metric == L2:
            assert I0[0, 0] == 11 and I0[0, 1] == n - 7
        keys = []
        G = 2
        for r in range(G):
            lo, hi = n * r // G, n * (r + 1) // G
            sh = make_index(faiss, metric, d)
            sh.add(xb[lo:hi])
            keys.append(sh.search_keys_torch(tq, k, id_base=lo))
        D1, I1 = faiss.merge_keys_torch(torch.stack(keys), metric)
        assert np.array_equal(I0, I1.cpu().numpy())
        assert np.array_equal(D0, D1.cpu().numpy())


def test_benchmark_data_1000_query_sample(faiss):
    """SURVEY.md 8d parity gate: the benchmark's own index (1M x 512 uniform[0,1), default_rng 1234)
    against a 1000-query sample, ids identical to the CPU oracle (tie-aware), distances within
    1e-4 * max(1, |D|).  The oracle pass is the C restatement on the host cores (~10-20 s)."""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import host_cores, make_inputs
    from oracle import flat_oracle as fo

    n, d, nq, k = 1_000_000, 512, 1000, 10
    xb, xq = make_inputs(n, d, nq, 0, n)
    index = faiss.IndexFlatL2(d)
    index.add_torch(torch.from_numpy(xb).cuda())
    D, I = index.search(xq, k)                       # host API, batches of 48 queries per pass inside
    Dr, Ir, _ = fo.knn_flat(xb, xq, k, 1, host_cores())
    n_mism = assert_knn_matches(D, I, Dr, Ir, xb, xq, 1)
    assert n_mism <= 2, n_mism                      # float32 near-ties are rare on this data
    # the device API and a second call return the same bits
    D2, I2 = index.search_torch(torch.from_numpy(xq).cuda(), k)
    assert np.array_equal(I, I2.cpu().numpy()) and np.array_equal(D, D2.cpu().numpy())


def test_streams_and_threads_stress_with_exchange(faiss):
    """The stress test on an index long enough for the threshold exchange to run: host threads on
    their own streams, batch sizes that use the two- and three-tile kernels (and several passes),
    each stream's slot carrying its own exchange buffer and sequence tags."""
    import torch
    from oracle import flat_oracle as fo

    rng = np.random.default_rng(77)
    n, d, k = 400_000, 64, 10
    xb = rng.random((n, d), dtype=np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    qs = [rng.random((nq, d), dtype=np.float32) for nq in (32, 48, 100, 33, 150)]
    refs = [fo.knn_flat(xb, q, k, 1, 8)[:2] for q in qs]
    errors = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            tq = torch.from_numpy(qs[i]).cuda()
            torch.cuda.synchronize()
            outs = []
            with torch.cuda.stream(st):
                for _ in range(15):
                    outs.append(index.search_torch(tq, k))
            st.synchronize()
            for D, I in outs[::7]:
                assert_knn_matches(D.cpu().numpy(), I.cpu().numpy(), refs[i][0], refs[i][1], xb, qs[i], L2)
            assert all(torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][0], o[0]) for o in outs)
        except Exception as e:  # surfaced in the main thread
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(qs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors


@pytest.mark.parametrize("storage,metric", [("f32", 1), ("bf16", 0)])
def test_ten_million_rows_on_one_gpu(faiss, storage, metric):
    """BASELINE config 5's row count on ONE card (20.5 GB float32 / 10.2 GB bf16): every offset in the
    library is 64-bit clean.  The CPU oracle does not finish this size in seconds, so the checker here is a
    float64 brute force in torch on the device, chunk by chunk (rows are regenerated from per-chunk seeds);
    float32 L2 ids must be identical, bf16 ids identical on the rounded values the index holds."""
    import torch

    dev = torch.device("cuda")
    N, d, nq, k, CH = 10_000_000, 512, 16, 10, 1_000_000
    unit = storage == "bf16"

    def chunk(i):
        g = torch.Generator(device=dev).manual_seed(1000 + i)
        x = torch.rand((CH, d), generator=g, device=dev)
        if unit:
            x = x - 0.5
            x = x / x.norm(dim=1, keepdim=True)
        return x

    index = faiss.IndexFlat(d, metric, storage=storage)
    for i in range(N // CH):
        index.add_torch(chunk(i))
    assert index.ntotal == N
    g = torch.Generator(device=dev).manual_seed(7)
    xq = torch.rand((nq, d), generator=g, device=dev)
    if unit:
        xq = xq - 0.5
        xq = xq / xq.norm(dim=1, keepdim=True)
    xq[3] = chunk(9)[CH - 1]                       # the very last row is its own nearest neighbour
    D, I = index.search_torch(xq, k)
    best_d = torch.full((nq, k), float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    q64 = xq.to(torch.bfloat16).double() if unit else xq.double()
    for i in range(N // CH):
        x = chunk(i)
        x64 = x.to(torch.bfloat16).double() if unit else x.double()
        if metric == 1:
            s = (q64 * q64).sum(1, keepdim=True) + (x64 * x64).sum(1)[None, :] - 2.0 * q64 @ x64.T
        else:
            s = -(q64 @ x64.T)
        ids = torch.arange(i * CH, (i + 1) * CH, device=dev)[None, :].expand(nq, -1)
        cd, ci = torch.cat([best_d, s], 1), torch.cat([best_i, ids], 1)
        o = torch.argsort(cd, dim=1, stable=True)[:, :k]
        best_d, best_i = torch.gather(cd, 1, o), torch.gather(ci, 1, o)
        del x, x64, s
    assert I[3, 0].item() == N - 1
    assert torch.equal(I, best_i), (I != best_i).nonzero()[:5]
    got = D.double() if metric == 1 else -D.double()
    assert float((got - best_d).abs().max()) <= 1e-4
    if unit:
        # what BASELINE config 5 reports: recall@10 of the bf16 index against the float32-exact answer, on a
        # 1000-query sample (float64 brute force on the UNROUNDED rows; the large-batch bf16 path answers)
        nq2 = 1000
        xq2 = torch.rand((nq2, d), generator=g, device=dev) - 0.5
        xq2 = xq2 / xq2.norm(dim=1, keepdim=True)
        _, I2 = index.search_torch(xq2, k)
        bd = torch.full((nq2, k), float("inf"), dtype=torch.float64, device=dev)
        bi = torch.full((nq2, k), -1, dtype=torch.int64, device=dev)
        q64 = xq2.double()
        sub = 250_000
        for i in range(N // CH):
            x = chunk(i)
            for j in range(0, CH, sub):
                s_ = -(q64 @ x[j:j + sub].double().T)
                top = torch.topk(s_, k, dim=1, largest=False)
                cd = torch.cat([bd, top.values], 1)
                ci = torch.cat([bi, top.indices + (i * CH + j)], 1)
                o = torch.argsort(cd, dim=1, stable=True)[:, :k]
                bd, bi = torch.gather(cd, 1, o), torch.gather(ci, 1, o)
                del s_
            del x
        recall = float((I2[:, :, None] == bi[:, None, :]).any(2).float().mean())
        assert recall >= 0.97, recall
    del index
    torch.cuda.empty_cache()


@pytest.mark.parametrize("metric", [IP, L2])
def test_config4_full_size_assignment(faiss, metric):
    """BASELINE config 4 at its full size on one card: 50k images x 1k keypoints = 50 M SIFT-valued rows x 128
    against 4096 unit centroids, k = 1 (backend/kmeans_faiss.py:46-50 behind
    backend/bag_of_visual_words.py:98-106), generated on the device in chunks of 5 M rows as SURVEY.md 8d
    prescribes.  Every label is in range, and a 1500-row sample of every chunk is checked against the float64
    oracle (ids identical away from float32 near-ties)."""
    import torch

    K, d, chunk, nchunks = 4096, 128, 5_000_000, 10
    rng = np.random.default_rng(42)
    cent = ko.normalize_rows(rng.standard_normal((K, d)).astype(np.float32))
    index = make_index(faiss, metric, d)
    index.add(cent)
    assert index._assign_applies(chunk, 1)
    g = torch.Generator(device="cuda").manual_seed(4)
    total = 0
    for i in range(nchunks):
        X = torch.randint(0, 256, (chunk, d), generator=g, device="cuda", dtype=torch.int32).to(torch.float32)
        D, I = index.assign_torch(X)
        assert I.shape == (chunk, 1) and int(I.min()) >= 0 and int(I.max()) < K
        total += chunk
        pick = torch.randint(0, chunk, (1500,), generator=g, device="cuda")
        xs = X[pick].cpu().numpy()
        D_ref, I_ref = ko.knn_exact(cent, xs, 1, metric)
        assert_knn_matches(D[pick].cpu().numpy(), I[pick].cpu().numpy(), D_ref, I_ref, cent, xs, metric,
                           gap=ko.kth_gap(cent, xs, 1, metric), rtol=2e-4)
        del X, D, I
    assert total == 50_000_000
