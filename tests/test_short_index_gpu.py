"""GPU parity tests of the scan kernel of SHORT indexes (csrc/ise_short_scan.hpp): the reference's own
regime -- about 1 k images, one query per request (backend/utils.py:309-310, backend/engine.py:50-55) --
BASELINE config 2 (100k x 512) and the 125k-row shard of the 8-GPU run.

The kernel must leave, bit for bit, the per-block lists the streaming kernel leaves (so that the search
returns the same bits either way), and the results must match the CPU oracle.  Parity is UNPINNED with respect to a real Faiss build (see
oracle/knn_oracle.py)."""
import threading

import numpy as np
import pytest

from oracle import knn_oracle as ko
from tests.knn_checks import ATOL_UNIFORM, assert_knn_matches
from tests.test_exact_l2_gpu import env_knob, forced_exact, no_short

pytestmark = pytest.mark.gpu

L2, IP = ko.METRIC_L2, ko.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def faiss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import image_search_engine_amd.faiss_compat as fc

    return fc


def _bf16_round(x):
    import torch

    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


def _data(rng, n, d, nq, unit):
    xb = rng.random((n, d), dtype=np.float32)
    xq = rng.random((nq, d), dtype=np.float32)
    if unit:
        xb /= np.maximum(np.linalg.norm(xb, axis=1, keepdims=True), 1e-6)
        xq /= np.maximum(np.linalg.norm(xq, axis=1, keepdims=True), 1e-6)
    return xb, xq


SHAPES = [  # n, d: block counts from 1 to the full grid, tails in rows and in columns, rows up to the 2 KB limit
    (1, 8), (15, 3), (16, 32), (17, 100), (129, 128), (1000, 512), (4097, 512), (30_000, 200),
    (100_000, 128), (200_000, 64), (20_000, 500),
]


@pytest.mark.parametrize("metric,storage", [(L2, "f32"), (IP, "f32"), (L2, "bf16"), (IP, "bf16")])
@pytest.mark.parametrize("n,d", SHAPES)
def test_short_kernel_equals_the_streaming_path_and_the_oracle(faiss, metric, storage, n, d):
    """Every instantiation (float32 L2 in front of the exact re-rank, float32 inner product, bf16 rows), batches
    of 1 to 64 queries (one and two query tiles per pass), k = 1, 10 and the largest one pass takes, through the host entry point and the packed
    keys of the shard entry point (global ids): short-index kernel == streaming kernel, bit for bit, and both
    match the oracle (bf16: on the rounded values the index holds)."""
    import torch

    rng = np.random.default_rng(n * 31 + d + metric)
    nq_max = 64
    xb, xq_all = _data(rng, n, d, nq_max, unit=(metric == IP))
    index = faiss.IndexFlat(d, metric, storage=storage)
    index.add(xb)
    xb_r, xq_r_all = (xb, xq_all) if storage == "f32" else (_bf16_round(xb), _bf16_round(xq_all))
    kmax = 28 if (metric == L2 and storage == "f32") else 32   # float32 L2 keeps 4 spare candidates per query
    launched = index.short_stats()["short_batches"]
    # 17 .. 64 queries: two query tiles per pass (33 .. 64: two passes side by side in one launch)
    for nq, k in ((1, 10), (5, 1), (16, 10), (16, kmax), (7, 3), (17, 10), (32, kmax), (33, 5), (64, 10)):
        xq, xq_r = xq_all[:nq], xq_r_all[:nq]
        # one float32 L2 query against an index of up to 128 row tiles is answered by the direct scan (one launch)
        short = not (nq == 1 and metric == L2 and storage == "f32" and (n + 15) // 16 <= 128)
        D, I = index.search(xq, k)
        launched += 1 if short else 0
        assert index.short_stats() == {"short_batches": launched}, (nq, k)
        with no_short():
            Ds, Is = index.search(xq, k)
        assert index.short_stats()["short_batches"] == launched
        assert np.array_equal(I, Is) and np.array_equal(D, Ds), (nq, k)
        D_ref, I_ref = ko.knn_exact(xb_r, xq_r, k, metric)
        assert_knn_matches(D, I, D_ref, I_ref, xb_r, xq_r, metric, gap=ko.kth_gap(xb_r, xq_r, k, metric) if n > 1 else None)
        keys = index.search_keys_torch(torch.from_numpy(xq).cuda(), k, 12345)
        launched += 1 if short else 0
        Dm, Im = faiss.merge_keys_torch(keys[None], metric)
        assert np.array_equal(Im.cpu().numpy(), np.where(I >= 0, I + 12345, -1)) and np.array_equal(Dm.cpu().numpy(), D)


def test_what_does_not_take_the_short_kernel(faiss):
    """65 queries (more than two passes of two query tiles), k whose candidates need two passes, an index with more
    than 32 row tiles per block and rows of more than 2 KB keep the streaming kernels; the answers still agree
    with the oracle."""
    rng = np.random.default_rng(8)
    xb, xq = _data(rng, 50_000, 64, 65, unit=False)
    index = faiss.IndexFlatL2(64)
    index.add(xb)
    for nq, k in ((65, 10), (4, 33), (16, 100)):
        D, I = index.search(xq[:nq], k)
        D_ref, I_ref = ko.knn_exact(xb, xq[:nq], k, L2)
        assert_knn_matches(D, I, D_ref, I_ref, xb, xq[:nq], L2, gap=ko.kth_gap(xb, xq[:nq], k, L2), atol=ATOL_UNIFORM)
    assert index.short_stats()["short_batches"] == 0
    D, I = index.search(xq[:16], 32)                       # k = 32 + 4 spare candidates: still one pass
    assert index.short_stats()["short_batches"] == 1
    big = faiss.IndexFlatL2(16)
    big.add(rng.random((300_000, 16), dtype=np.float32))   # 18750 row tiles: 37 per block even with 512 blocks
    big.search(xq[:4, :16].copy(), 5)
    assert big.short_stats()["short_batches"] == 0
    wide = faiss.IndexFlatL2(2048)                          # the reference's own descriptor size: 8 KB rows
    xw = rng.random((3000, 2048), dtype=np.float32)
    wide.add(xw)
    Dw, Iw = wide.search(xw[5:9] + np.float32(0.001), 10)
    assert wide.short_stats()["short_batches"] == 0 and Iw[:, 0].tolist() == [5, 6, 7, 8]
    wide16 = faiss.IndexFlat(1024, L2, storage="bf16")      # 2 KB rows in bf16: still the short kernel
    wide16.add(xw[:, :1024].copy())
    wide16.search(xw[5:9, :1024].copy(), 10)
    assert wide16.short_stats()["short_batches"] == 1
    with env_knob("ISE_SHORT_TPB_MAX", 4):                  # the knob lowers the limit (A/B runs)
        index.search(xq[:16], 10)                           # 3125 row tiles: 8 per block at best
    assert index.short_stats()["short_batches"] == 1


def test_short_kernel_with_failed_certificates_and_forced_exact(faiss):
    """Data the filter cannot certify (two clusters 1e3 apart) and the test knob that fails every certificate:
    the re-rank lists the queries, the gated exact scan answers them; same bits as the streaming path."""
    rng = np.random.default_rng(9)
    n, d, nq, k = 24_000, 128, 16, 10
    off = np.zeros(d, np.float32)
    off[0] = 1000.0
    xb = 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    xb[n // 2:] += off
    xq = (xb[rng.integers(0, n, nq)] + 0.03 * rng.standard_normal((nq, d))).astype(np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    before = index.exact_stats()
    D, I = index.search(xq, k)
    after = index.exact_stats()
    assert after["exact_scan"] - before["exact_scan"] > 0 and after["reranked"] - before["reranked"] == nq
    D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2, gap=ko.kth_gap(xb, xq, k, L2))
    with no_short():
        Ds, Is = index.search(xq, k)
    assert np.array_equal(I, Is) and np.array_equal(D, Ds)
    uni = faiss.IndexFlatL2(d)
    xu = rng.random((n, d), dtype=np.float32)
    uni.add(xu)
    xqu = rng.random((nq, d), dtype=np.float32)
    D, I = uni.search(xqu, k)
    assert uni.exact_stats()["exact_scan"] == 0
    with forced_exact():
        Df, If = uni.search(xqu, k)
    assert uni.exact_stats()["exact_scan"] == nq and uni.short_stats() == {"short_batches": 2}
    assert np.array_equal(If, I) and np.array_equal(Df, D)


@pytest.mark.parametrize("metric", [L2, IP])
def test_many_batches_in_flight_on_many_streams(faiss, metric):
    """16 streams x 60 batches of different queries in flight over the six workspace slots, plus host threads
    searching at the same time.  Every result equals the one the batch gets alone."""
    import torch

    rng = np.random.default_rng(10 + metric)
    n, d, k, nb = 100_000, 128, 10, 8
    xb, xq_all = _data(rng, n, d, 16 * nb, unit=(metric == IP))
    index = faiss.IndexFlat(d, metric)
    index.add(xb)
    batches = [torch.from_numpy(xq_all[16 * b:16 * (b + 1)]).cuda() for b in range(nb)]
    want = [index.search_torch(q, k) for q in batches]
    for b in (0, nb - 1):
        D_ref, I_ref = ko.knn_exact(xb, xq_all[16 * b:16 * (b + 1)], k, metric)
        assert_knn_matches(want[b][0].cpu().numpy(), want[b][1].cpu().numpy(), D_ref, I_ref, xb,
                           xq_all[16 * b:16 * (b + 1)], metric, gap=ko.kth_gap(xb, xq_all[16 * b:16 * (b + 1)], k, metric))
    streams = [torch.cuda.Stream() for _ in range(16)]
    outs = [[(torch.empty((16, k), dtype=torch.float32, device="cuda"), torch.empty((16, k), dtype=torch.int64, device="cuda"))
             for _ in range(60)] for _ in streams]
    torch.cuda.synchronize()
    errors = []

    def host_caller(i):
        try:
            for r in range(20):
                b = (i + r) % nb
                D, I = index.search(xq_all[16 * b:16 * b + 1 + (i % 3)], k)
                m = 1 + (i % 3)
                if not (np.array_equal(I, want[b][1][:m].cpu().numpy()) and np.array_equal(D, want[b][0][:m].cpu().numpy())):
                    errors.append((i, r))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=host_caller, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    for r in range(60):
        for s, st in enumerate(streams):
            index.search_into(batches[(r + s) % nb], k, outs[s][r][0], outs[s][r][1], st.cuda_stream)
    [t.join() for t in th]
    torch.cuda.synchronize()
    assert not errors, errors[:3]
    for r in range(60):
        for s in range(16):
            b = (r + s) % nb
            assert torch.equal(outs[s][r][1], want[b][1]) and torch.equal(outs[s][r][0], want[b][0]), (r, s)


def test_config2_and_shard_sizes_at_full_size(faiss):
    """BASELINE config 2's index (100k x 512) and the 8-GPU run's per-rank shard (125k x 512), the bench's data:
    nq = 16 and nq = 1 through the short-index kernel, ids identical to the C oracle, distances within the
    absolute 1e-4; no certificate fails on this data."""
    from oracle import flat_oracle as fo

    fo.build()
    d, k = 512, 10
    xb = np.random.default_rng(1234).random((125_000, d), dtype=np.float32)
    xq = np.random.default_rng(4321).random((16, d), dtype=np.float32)
    for n in (100_000, 125_000):
        index = faiss.IndexFlatL2(d)
        index.add(xb[:n])
        Dc, Ic, _ = fo.knn_flat(xb[:n], xq, k, 1, 16)
        for nq in (16, 1):
            D, I = index.search(xq[:nq], k)
            assert_knn_matches(D, I, Dc[:nq], Ic[:nq], xb[:n], xq[:nq], 1, atol=ATOL_UNIFORM)
        assert index.short_stats() == {"short_batches": 2}
        assert index.exact_stats()["exact_scan"] == 0


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("n,d", [(1000, 2048), (4000, 2048), (1000, 1024), (999, 2047), (1000, 512), (5000, 500), (37, 2048)])
def test_small_indexes_spread_over_the_cus(faiss, metric, n, d):
    """The reference's own index size (about 1000 descriptors of 2048 floats, backend/utils.py:309-310,
    backend/descriptors.py:120) and its neighbours: the plans spread so small an index over the CUs -- single
    row tiles per block for rows of more than 2 KB, ~64 KB of rows per block otherwise (make_plan, the short-index
    plan, the one-query direct scan) -- and the results are the oracle's for batches of 1, 16, 17, 33 and 64
    queries (one and several query tiles per pass, one and several passes), through the host entry point and,
    packed as keys with global ids, through the shard entry point."""
    import torch

    rng = np.random.default_rng(n * 7 + d + metric)
    xb, xq_all = _data(rng, n, d, 64, unit=(metric == IP))
    index = faiss.IndexFlat(d, metric)
    index.add(xb)
    for nq, k in ((1, 20), (16, 10), (17, 10), (33, 5), (64, 20), (1, 1)):
        xq = xq_all[:nq]
        D, I = index.search(xq, k)
        D_ref, I_ref = ko.knn_exact(xb, xq, k, metric)
        assert_knn_matches(D, I, D_ref, I_ref, xb, xq, metric, gap=ko.kth_gap(xb, xq, k, metric) if n > k else None)
        keys = index.search_keys_torch(torch.from_numpy(xq).cuda(), k, 777)
        Dm, Im = faiss.merge_keys_torch(keys[None], metric)
        assert np.array_equal(Im.cpu().numpy(), np.where(I >= 0, I + 777, -1)) and np.array_equal(Dm.cpu().numpy(), D)
    # a row queried against itself comes back first (L2: at distance exactly 0)
    D0, I0 = index.search(xb[n // 2:n // 2 + 1], 1)
    assert I0[0, 0] == n // 2 and (metric == IP or D0[0, 0] == 0.0)
