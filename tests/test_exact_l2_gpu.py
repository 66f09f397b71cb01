"""GPU parity tests of the EXACT float32 L2 search (csrc/ise_exact.hpp): the streaming scan is a
filter keyed by a rigorous lower bound, candidates are re-ranked by the direct difference
sum (x - y)^2 -- what Faiss's IndexFlatL2 computes for the reference's one-query searches
(backend/engine.py:50-55; numpy restatement backend/siamese/test_index.py:58-69) -- and queries the
filter cannot certify go to an exact direct-difference scan.  The data here is built to break an
expanded-form kernel: means that drift, far-apart clusters, outliers, massive duplicates.
Expected values: the float64 oracle (parity unpinned w.r.t. a real Faiss build, see oracle/)."""
import os
import zlib

import numpy as np
import pytest

from oracle import knn_oracle as ko
from tests.knn_checks import ATOL_UNIFORM, assert_knn_matches

pytestmark = pytest.mark.gpu
L2 = ko.METRIC_L2


@pytest.fixture(scope="module")
def faiss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import image_search_engine_amd.faiss_compat as fc

    return fc


class env_knob:
    """A test knob of the library, set for the block: the library reads its knobs from the environment
    when ise_refresh_env_knobs() is called (include/ise_knn.h), never inside a search."""

    def __init__(self, name, value="1"):
        self.name, self.value = name, str(value)

    def __enter__(self):
        from image_search_engine_amd import _native as n

        os.environ[self.name] = self.value
        n.lib.ise_refresh_env_knobs()

    def __exit__(self, *a):
        from image_search_engine_amd import _native as n

        os.environ.pop(self.name, None)
        n.lib.ise_refresh_env_knobs()


def no_direct():
    """One-query float32 L2 batches against LONG indexes skip the filter and run the direct-difference scan
    alone; with this they take the filtered path like every other batch."""
    return env_knob("ISE_NO_DIRECT")


def forced_exact():
    """Every certificate fails inside the block: the exact fallback scan produces the results."""
    return env_knob("ISE_FORCE_EXACT")


def force_direct():
    """One-query float32 L2 batches run the direct-difference scan whatever the index length (by default indexes
    between ~2k and 262k rows take the filtered path behind the short-index scan kernel instead)."""
    return env_knob("ISE_DIRECT_SHORT_MAX_TILES", 1 << 30)


def no_short():
    """Short indexes are scanned by the streaming kernel (boot, thresholds) instead of the short-index kernel."""
    return env_knob("ISE_NO_SHORT")


def _adversarial(kind, rng, n, d):
    if kind == "cluster_sorted":      # rows sorted by class: the mean drifts all the way through the index
        c = np.sort(rng.integers(0, 8, n))
        centres = (rng.standard_normal((8, d)) * 20.0).astype(np.float32)
        xb = centres[c] + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    elif kind == "two_far_clusters":  # 1e3 apart, spread 1e-1: |y - mu|^2 ~ 2.5e5 against neighbour gaps ~ 1e-2
        off = np.zeros(d, np.float32)
        off[0] = 1000.0
        xb = 0.1 * rng.standard_normal((n, d)).astype(np.float32)
        xb[n // 2:] += off
    elif kind == "outlier_first":     # the first row is nothing like the rest
        xb = (5.0 + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
        xb[0] = 1e3
    elif kind == "huge_norm_rows":    # a few rows with 1e4 times the norm of the rest
        xb = rng.random((n, d), dtype=np.float32)
        xb[rng.integers(0, n, 5)] *= 1e4
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(xb, dtype=np.float32)


@pytest.mark.parametrize("adds", ["one", "several"])
@pytest.mark.parametrize("nq", [1, 16])
@pytest.mark.parametrize("kind", ["cluster_sorted", "two_far_clusters", "outlier_first", "huge_norm_rows"])
def test_l2_exact_on_adversarial_data(faiss, kind, nq, adds):
    rng = np.random.default_rng(zlib.crc32(f"{kind}{nq}".encode()))
    n, d, k = 24_000, 128, 10
    xb = _adversarial(kind, rng, n, d)
    xq = (xb[rng.integers(0, n, nq)] + 0.03 * rng.standard_normal((nq, d))).astype(np.float32)
    index = faiss.IndexFlatL2(d)
    if adds == "one":
        index.add(xb)
    else:  # a one-row first add, then uneven pieces with a search in between (the shift is refreshed lazily)
        index.add(xb[:1])
        index.add(xb[1:5000])
        index.search(xq, k)
        index.add(xb[5000:5003])
        index.add(xb[5003:])
    D, I = index.search(xq, k)
    D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
    n_mism = assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2, gap=ko.kth_gap(xb, xq, k, L2))
    assert n_mism == 0 or kind == "two_far_clusters"  # only float32 near-ties may differ, and only there
    # (the search between the adds met 5000 rows: above the direct scan's 128 row tiles, so the short kernel as well)
    assert index.short_stats() == {"short_batches": 2 if adds == "several" else 1}
    if nq == 1:  # a one-query batch through the direct scan alone and through the streaming kernels: the same bits
        with force_direct():
            d0 = index.host_stats()["direct_queries"]
            Df, If = index.search(xq, k)
            assert index.host_stats()["direct_queries"] == d0 + 1
        assert np.array_equal(If, I) and np.array_equal(Df, D)
    with no_short():
        Df, If = index.search(xq, k)
    assert np.array_equal(If, I) and np.array_equal(Df, D)
    st = index.exact_stats()
    assert st["reranked"] >= nq
    if kind == "two_far_clusters":
        assert st["exact_scan"] > 0, "the filter cannot certify this data: the exact scan must have run"


def test_certificate_holds_on_benign_data_and_forced_exact_agrees(faiss):
    """uniform[0,1) (the benchmark distribution): no query needs the exact scan; forcing it returns
    the same bits (both paths evaluate the same d(x, y)); distances within the ABSOLUTE 1e-4."""
    rng = np.random.default_rng(5)
    n, d = 50_000, 512
    xb = rng.random((n, d), dtype=np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    for nq, k in ((1, 10), (16, 10), (40, 20), (5, 1), (7, 32), (3, 33), (4, 100)):
        xq = rng.random((nq, d), dtype=np.float32)
        before = index.exact_stats()
        with no_direct():  # the filtered path is the subject here, for the one-query batch as well
            D, I = index.search(xq, k)
        after = index.exact_stats()
        assert after["reranked"] - before["reranked"] == nq
        assert after["exact_scan"] == before["exact_scan"], "certificate failed on uniform data"
        D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
        assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2, gap=ko.kth_gap(xb, xq, k, L2), atol=ATOL_UNIFORM)
        with forced_exact():
            Df, If = index.search(xq, k)
        assert index.exact_stats()["exact_scan"] - after["exact_scan"] == nq
        assert np.array_equal(If, I) and np.array_equal(Df, D)


@pytest.mark.parametrize("n,d", [(1, 8), (63, 20), (64, 128), (5000, 100), (70_000, 512), (20_000, 2100)])
def test_one_query_batches_run_the_direct_scan(faiss, n, d):
    """A float32 L2 batch of ONE query with k <= 32 -- the reference's request pattern, backend/engine.py:50-55 --
    is answered by the direct-difference scan alone (Faiss's nq < 20 algorithm as it stands: no filter, no
    certificate): same bits as the filtered path, ids identical to the float64 oracle, through the host, the
    device and the packed-key (shard) entry points, k from 1 to 32 and beyond the index size, odd d, duplicate
    rows (lowest id first), rows far from the origin; larger batches and k > 32 keep the filtered path."""
    import torch

    rng = np.random.default_rng(n + d)
    xb = (rng.random((n, d), dtype=np.float32) + np.float32(25.0 if n == 5000 else 0.0))
    if n >= 64:
        xb[n - 2] = xb[5]                      # duplicate rows: the lower id first
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    direct0 = index.host_stats()["direct_queries"]
    asked = 0
    with force_direct():  # (short indexes take the filtered path otherwise: compared below)
        _one_query_direct_cases(faiss, index, xb, n, d, direct0, asked)


def _one_query_direct_cases(faiss, index, xb, n, d, direct0, asked):
    import torch

    for k in (1, 10, 32):
        xq = (xb[min(5, n - 1):min(5, n - 1) + 1] + (np.float32(0.0) if k == 10 else np.float32(0.01))).astype(np.float32)
        D, I = index.search(xq, k)
        asked += 1
        D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
        assert assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2, gap=ko.kth_gap(xb, xq, k, L2)) == 0
        if k == 10 and n >= 64:
            assert I[0, 0] == 5 and I[0, 1] == n - 2 and D[0, 0] == 0.0 and D[0, 1] == 0.0
        with no_direct():
            Df, If = index.search(xq, k)
        assert np.array_equal(If, I) and np.array_equal(Df, D), (n, d, k)
        with no_direct(), no_short():
            Df, If = index.search(xq, k)
        assert np.array_equal(If, I) and np.array_equal(Df, D), (n, d, k)
        Dt, It = index.search_torch(torch.from_numpy(xq).cuda(), k)
        asked += 1
        assert np.array_equal(It.cpu().numpy(), I) and np.array_equal(Dt.cpu().numpy(), D)
        keys = index.search_keys_torch(torch.from_numpy(xq).cuda(), k, 1000)   # shard form: ids + 1000
        asked += 1
        Dm, Im = faiss.merge_keys_torch(keys[None], L2)
        assert np.array_equal(Im.cpu().numpy(), np.where(I >= 0, I + 1000, -1)) and np.array_equal(Dm.cpu().numpy(), D)
    assert index.host_stats()["direct_queries"] - direct0 == asked
    # not direct: two queries, k > 32 -- and the answers still agree with the one-query calls
    two = np.concatenate([xq, xq])
    D2, I2 = index.search(two, 32)
    D33, I33 = index.search(xq, 33)
    assert index.host_stats()["direct_queries"] - direct0 == asked
    assert np.array_equal(I2[0], I[0]) and np.array_equal(I2[1], I[0]) and np.array_equal(D2[1], D[0])
    assert np.array_equal(I33[0, :32], I[0]) and np.array_equal(D33[0, :32], D[0])


def test_massive_duplicates_go_through_the_exact_scan(faiss):
    """More copies of the nearest row than candidate slots: the filter cannot separate them from the
    rest, the exact scan returns the lowest ids (Faiss tie order)."""
    rng = np.random.default_rng(8)
    n, d, k = 6000, 64, 10
    xb = rng.random((n, d), dtype=np.float32)
    dup = np.sort(rng.choice(n, 200, replace=False))
    xb[dup] = xb[dup[0]]
    xq = np.stack([xb[dup[0]], xb[dup[0]] + np.float32(0.001)]).astype(np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    D, I = index.search(xq, k)
    assert np.array_equal(I[0], dup[:k]) and (D[0] == 0).all()
    assert np.array_equal(I[1], dup[:k])
    assert index.exact_stats()["exact_scan"] == 2
    D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2)


def test_exact_keys_shards_merge_to_the_unsharded_result(faiss):
    """Shards need not share anything: each re-ranks exactly on its own, the packed keys carry the
    direct-difference distance, and the merge equals one index -- also through the exact scan."""
    import torch

    rng = np.random.default_rng(12)
    n, d, k = 30_000, 96, 10
    xb = _adversarial("cluster_sorted", rng, n, d)
    xq = (xb[rng.integers(0, n, 20)] + 0.02 * rng.standard_normal((20, d))).astype(np.float32)
    tq = torch.from_numpy(xq).cuda()
    whole = faiss.IndexFlatL2(d)
    whole.add(xb)
    D0, I0 = whole.search(xq, k)
    for force in (False, True):
        keys = []
        for r in range(3):
            lo, hi = n * r // 3, n * (r + 1) // 3
            sh = faiss.IndexFlatL2(d)
            sh.add(xb[lo:hi])
            if force:
                with forced_exact():
                    keys.append(sh.search_keys_torch(tq, k, id_base=lo))
                    torch.cuda.synchronize()
            else:
                keys.append(sh.search_keys_torch(tq, k, id_base=lo))
        D1, I1 = faiss.merge_keys_torch(torch.stack(keys), L2)
        assert np.array_equal(I0, I1.cpu().numpy()) and np.array_equal(D0, D1.cpu().numpy())


def test_pinned_shift_changes_nothing_but_the_path(faiss):
    """Results do not depend on the shift vector: a deliberately bad one only sends queries to the
    exact scan."""
    rng = np.random.default_rng(3)
    n, d, k = 20_000, 128, 10
    xb = (3.0 + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
    xq = (3.0 + 0.2 * rng.standard_normal((8, d))).astype(np.float32)
    good = faiss.IndexFlatL2(d)
    good.add(xb)
    D0, I0 = good.search(xq, k)
    bad = faiss.IndexFlatL2(d)
    bad.set_shift(np.full(d, -3000.0, np.float32))
    bad.add(xb)
    D1, I1 = bad.search(xq, k)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    assert np.allclose(bad.get_shift(), -3000.0)
    assert good.exact_stats()["exact_scan"] == 0 and bad.exact_stats()["exact_scan"] > 0
    assert abs(float(good.get_shift().mean()) - 3.0) < 0.01


def test_shift_is_refreshed_as_the_index_grows(faiss):
    rng = np.random.default_rng(4)
    d = 64
    index = faiss.IndexFlatL2(d)
    index.add(np.full((1, d), 100.0, np.float32))      # an outlier first add
    xq = rng.random((2, d), dtype=np.float32)
    index.search(xq, 1)
    assert index.exact_stats()["shift_updates"] == 1 and np.allclose(index.get_shift(), 100.0)
    rows = rng.random((4000, d), dtype=np.float32)
    index.add(rows)
    D, I = index.search(xq, 5)
    assert index.exact_stats()["shift_updates"] == 2
    assert abs(float(index.get_shift().mean()) - (0.5 * 4000 + 100.0) / 4001) < 0.01
    index.add(rows[:100])                               # < a quarter more rows: the shift stays
    index.search(xq, 5)
    assert index.exact_stats()["shift_updates"] == 2
    xb = np.concatenate([np.full((1, d), 100.0, np.float32), rows, rows[:100]])
    D, I = index.search(xq, 5)
    D_ref, I_ref = ko.knn_exact(xb, xq, 5, L2)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2)


def test_reserve_workspaces_then_search_allocates_nothing_new(faiss):
    """ise_index_reserve_workspaces: sizes every slot ahead of the first batch (bench.py calls it
    before its warm-up); searches on many streams afterwards agree with the oracle."""
    import torch

    rng = np.random.default_rng(6)
    n, d, k, nq = 40_000, 128, 10, 16
    xb = rng.random((n, d), dtype=np.float32)
    xq = rng.random((nq, d), dtype=np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    index.reserve(nq, k)
    free0 = torch.cuda.mem_get_info()[0]
    tq = torch.from_numpy(xq).cuda()
    streams = [torch.cuda.Stream() for _ in range(16)]
    outs = [(torch.empty((nq, k), dtype=torch.float32, device="cuda"),
             torch.empty((nq, k), dtype=torch.int64, device="cuda")) for _ in streams]
    for s, (D, I) in zip(streams, outs):  # first use of a stream creates its hardware queue (runtime memory)
        with torch.cuda.stream(s):
            D.zero_()
    index.search_into(tq, k, outs[0][0], outs[0][1], streams[0].cuda_stream)  # loads the code objects
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    for _ in range(3):  # 16 streams over the library's six slots: every slot gets used
        for s, (D, I) in zip(streams, outs):
            index.search_into(tq, k, D, I, s.cuda_stream)
    torch.cuda.synchronize()
    # one slot's candidate lists alone are 1 MiB here: a lazily sized slot would show
    assert torch.cuda.mem_get_info()[0] >= free1 - (1 << 19), "a search allocated device memory after reserve()"
    del free0
    D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
    for D, I in outs:
        assert_knn_matches(D.cpu().numpy(), I.cpu().numpy(), D_ref, I_ref, xb, xq, L2, atol=ATOL_UNIFORM)


# ---------------------------------------------------------------- kernel instantiations round 1 never ran
@pytest.mark.parametrize("metric", [ko.METRIC_INNER_PRODUCT, L2])
@pytest.mark.parametrize("nq", [49, 64, 100])
def test_bf16_four_query_tiles(faiss, metric, nq):
    """scan_kernel<*, 8, 4, bf16>: chosen for nq >= 49 on bf16 rows (BASELINE config 5's large batches)."""
    import torch

    rng = np.random.default_rng(nq + metric)
    n, d, k = 30_000, 512, 10
    xb = rng.standard_normal((n, d)).astype(np.float32)
    xq = rng.standard_normal((nq, d)).astype(np.float32)
    rb = torch.from_numpy(xb).to(torch.bfloat16).to(torch.float32).numpy()
    rq = torch.from_numpy(xq).to(torch.bfloat16).to(torch.float32).numpy()
    index = faiss.IndexFlat(d, metric, storage="bf16")
    index.add(xb)
    D, I = index.search(xq, k)
    D_ref, I_ref = ko.knn_exact(rb, rq, k, metric)
    assert_knn_matches(D, I, D_ref, I_ref, rb, rq, metric, gap=ko.kth_gap(rb, rq, k, metric))


@pytest.mark.parametrize("storage,d", [("f32", 2100), ("f32", 2176), ("f32", 2240),
                                       ("bf16", 2304), ("bf16", 4400), ("bf16", 4480)])
def test_long_rows_four_wave_blocks(faiss, storage, d):
    """Rows of more than 8448 bytes (float32: 2112 < d <= 2240, bf16: 4224 < d <= 4480): 8 waves'
    candidate lists no longer fit beside a 16-query tile in the 160 KiB LDS, the host picks 4-wave
    blocks (scan_kernel<*, 4, 1, ...>); the shorter rows here are the longest the 8-wave kernel takes."""
    import torch

    rng = np.random.default_rng(d)
    n, nq, k = 2000, 5, 10
    xb = rng.random((n, d), dtype=np.float32)
    xq = rng.random((nq, d), dtype=np.float32)
    for metric in (L2, ko.METRIC_INNER_PRODUCT):
        index = faiss.IndexFlat(d, metric, storage=storage)
        index.add(xb)
        D, I = index.search(xq, k)
        if storage == "bf16":
            rb = torch.from_numpy(xb).to(torch.bfloat16).to(torch.float32).numpy()
            rq = torch.from_numpy(xq).to(torch.bfloat16).to(torch.float32).numpy()
        else:
            rb, rq = xb, xq
        D_ref, I_ref = ko.knn_exact(rb, rq, k, metric)
        assert_knn_matches(D, I, D_ref, I_ref, rb, rq, metric, gap=ko.kth_gap(rb, rq, k, metric))


def test_d_too_large_is_an_error_not_a_wrong_answer(faiss):
    from image_search_engine_amd._native import IseError

    for d, storage in ((2241, "f32"), (2304, "f32"), (4481, "bf16")):
        index = faiss.IndexFlat(d, L2, storage=storage)
        index.add(np.zeros((32, d), np.float32))
        with pytest.raises(IseError, match="d too large"):
            index.search(np.zeros((1, d), np.float32), 3)


def test_concurrent_host_searches_overlap_and_agree(faiss):
    """Flask request threads (backend/engine.py:137): the host API serves concurrent callers from a
    pool of contexts; every caller gets its own correct answer."""
    import threading

    rng = np.random.default_rng(10)
    n, d, k = 60_000, 128, 20
    xb = rng.random((n, d), dtype=np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    qs = [rng.random((1 + i % 3, d), dtype=np.float32) for i in range(12)]
    refs = [ko.knn_exact(xb, q, k, L2) for q in qs]
    errors = []

    def work(i):
        try:
            for _ in range(10):
                D, I = index.search(qs[i], k)
                assert_knn_matches(D, I, refs[i][0], refs[i][1], xb, qs[i], L2, atol=ATOL_UNIFORM)
        except Exception as e:
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(qs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors


@pytest.mark.parametrize("metric", [L2, 0])
def test_concurrent_small_searches_share_a_pass(faiss, metric):
    """Concurrent small host searches are combined into shared batches (include/ise_knn.h,
    ise_index_search_host): every caller gets, bit for bit, what it gets when it is alone -- one- to
    three-query calls, two different k in the mix (only equal k share a batch), a call too large to be
    combined running beside them -- and the counters show that calls did share batches."""
    import threading

    rng = np.random.default_rng(12 + metric)
    n, d = 200_000, 64
    xb = rng.random((n, d), dtype=np.float32) - np.float32(0.5 if metric == 0 else 0.0)
    index = faiss.IndexFlat(d, metric)
    index.add(xb)
    nthreads, per = 24, 12
    qs = [[rng.random((1 + (i + j) % 3, d), dtype=np.float32) - np.float32(0.5 if metric == 0 else 0.0)
           for j in range(per)] for i in range(nthreads)]
    ks = [10 if i % 4 else 20 for i in range(nthreads)]
    big = rng.random((40, d), dtype=np.float32)
    solo = [[index.search(q, ks[i]) for q in qs[i]] for i in range(nthreads)]   # one caller at a time
    solo_big = index.search(big, 10)
    s0 = index.host_stats()
    assert s0["combined_calls"] == nthreads * per and s0["combined_batches"] == nthreads * per  # alone: batches of one
    for i in (0, 5):  # and the solo answers are right
        Dr, Ir = ko.knn_exact(xb, qs[i][0], ks[i], metric)
        assert_knn_matches(solo[i][0][0], solo[i][0][1], Dr, Ir, xb, qs[i][0], metric, atol=ATOL_UNIFORM)
    errors = []
    start = threading.Barrier(nthreads + 1)

    def work(i):
        try:
            start.wait()
            for j, q in enumerate(qs[i]):
                D, I = index.search(q, ks[i])
                assert np.array_equal(I, solo[i][j][1]) and np.array_equal(D, solo[i][j][0]), (i, j)
        except Exception as e:
            errors.append((i, repr(e)))

    def work_big():
        try:
            start.wait()
            for _ in range(4):
                D, I = index.search(big, 10)
                assert np.array_equal(I, solo_big[1]) and np.array_equal(D, solo_big[0])
        except Exception as e:
            errors.append(("big", repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)] + [threading.Thread(target=work_big)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    s1 = index.host_stats()
    calls, batches = s1["combined_calls"] - s0["combined_calls"], s1["combined_batches"] - s0["combined_batches"]
    assert calls == nthreads * per            # the 40-query calls went their own way
    assert batches < calls, (batches, calls)  # 24 threads hammering one index: some calls shared a pass
    with pytest.raises(AssertionError):       # Faiss's own argument check (k > 0), before the library is entered
        index.search(qs[0][0], 0)
    D, I = index.search(qs[0][0], ks[0])
    assert np.array_equal(I, solo[0][0][1])


# ---------------------------------------------------------------- large batches: the GEMM-shaped path
@pytest.mark.parametrize("d,nq,k", [(512, 256, 10), (512, 300, 20), (128, 1024, 10), (256, 333, 1), (384, 260, 28),
                                    (512, 64, 10), (512, 100, 10)])
def test_large_batch_gemm_path_matches_oracle(faiss, d, nq, k):
    """nq >= 256 against a float32 L2 index of >= 128k rows: strided-sample thresholds, one GEMM-shaped
    pass, candidate select, exact re-rank (csrc/ise_gemm_scan.hpp).  Same answers as the oracle and as
    the streaming path (bit for bit: both end in the same direct-difference re-rank)."""
    from oracle import flat_oracle as fo

    rng = np.random.default_rng(d + nq + k)
    n = 140_000
    xb = rng.random((n, d), dtype=np.float32)
    xq = rng.random((nq, d), dtype=np.float32)
    xq[3] = xb[77]                      # an exact hit
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    before = index.exact_stats()
    D, I = index.search(xq, k)
    after = index.exact_stats()
    # from 256 queries on: the GEMM-shaped pass; below: two- and three-tile streaming passes
    assert after["gemm_chunks"] == before["gemm_chunks"] + (1 if nq >= 256 else 0), "unexpected path"
    assert after["exact_scan"] == before["exact_scan"], "certificate failed on uniform data"
    Dr, Ir, _ = fo.knn_flat(xb, xq, k, 1, 16)
    assert_knn_matches(D, I, Dr, Ir, xb, xq, L2, atol=ATOL_UNIFORM)
    assert I[3, 0] == 77 and D[3, 0] == 0.0
    # the streaming path on the same queries (batches below 256 queries): identical bits
    Ds = np.concatenate([index.search(xq[i:i + 48], k)[0] for i in range(0, nq, 48)])
    Is = np.concatenate([index.search(xq[i:i + 48], k)[1] for i in range(0, nq, 48)])
    assert np.array_equal(Is, I) and np.array_equal(Ds, D)
    with forced_exact():
        Df, If = index.search(xq, k)
    assert np.array_equal(If, I) and np.array_equal(Df, D)


def test_large_batch_gemm_path_on_cluster_sorted_rows(faiss):
    """Rows sorted by cluster, queries near every cluster: the strided sample still sees every cluster,
    thresholds stay tight, nothing overflows; and far-apart clusters go through the exact scan."""
    rng = np.random.default_rng(31)
    n, d, nq, k = 160_000, 128, 288, 10
    xb = _adversarial("cluster_sorted", rng, n, d)
    xq = (xb[rng.integers(0, n, nq)] + 0.03 * rng.standard_normal((nq, d))).astype(np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    D, I = index.search(xq, k)
    assert index.exact_stats()["gemm_chunks"] == 1
    D_ref, I_ref = ko.knn_exact(xb, xq, k, L2)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, L2, gap=ko.kth_gap(xb, xq, k, L2))
    xb2 = _adversarial("two_far_clusters", rng, n, d)
    xq2 = (xb2[rng.integers(0, n, 256)] + 0.03 * rng.standard_normal((256, d))).astype(np.float32)
    index2 = faiss.IndexFlatL2(d)
    index2.add(xb2)
    D2, I2 = index2.search(xq2, k)
    st = index2.exact_stats()
    assert st["gemm_chunks"] == 1 and st["exact_scan"] > 0
    D_ref, I_ref = ko.knn_exact(xb2, xq2, k, L2)
    assert_knn_matches(D2, I2, D_ref, I_ref, xb2, xq2, L2, gap=ko.kth_gap(xb2, xq2, k, L2))


def test_large_batch_path_on_concurrent_streams_and_threads(faiss):
    """The GEMM-shaped path keeps its buffers per workspace slot: batches of different sizes issued
    from several host threads on their own streams (and interleaved with small batches) stay correct
    and bitwise reproducible."""
    import threading

    import torch
    from oracle import flat_oracle as fo

    rng = np.random.default_rng(41)
    n, d, k = 150_000, 128, 10
    xb = rng.random((n, d), dtype=np.float32)
    index = faiss.IndexFlatL2(d)
    index.add(xb)
    qs = [rng.random((nq, d), dtype=np.float32) for nq in (256, 16, 700, 1100, 300)]
    refs = [fo.knn_flat(xb, q, k, 1, 8)[:2] for q in qs]
    errors = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            tq = torch.from_numpy(qs[i]).cuda()
            torch.cuda.synchronize()
            outs = []
            with torch.cuda.stream(st):
                for _ in range(6):
                    outs.append(index.search_torch(tq, k))
            st.synchronize()
            assert_knn_matches(outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy(), refs[i][0], refs[i][1], xb, qs[i], L2,
                               atol=ATOL_UNIFORM)
            assert all(torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][0], o[0]) for o in outs)
        except Exception as e:  # surfaced in the main thread
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(qs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert index.exact_stats()["gemm_chunks"] >= 6 * 5  # 256, 700, 1100 (two chunks) and 300 queries, six times each


@pytest.mark.parametrize("metric", [ko.METRIC_INNER_PRODUCT, L2])
@pytest.mark.parametrize("d,nq,k", [(512, 256, 10), (512, 1024, 10), (128, 300, 32), (384, 257, 1), (256, 128, 5)])
def test_bf16_large_batch_gemm_path(faiss, metric, d, nq, k):
    """BASELINE config 5's shape: bf16 rows, batches of >= 256 queries -> the bf16 GEMM-shaped pass
    (csrc/ise_gemm_bf16.hpp).  Against the exact oracle on the rounded values, and bit for bit against the
    streaming passes for the inner product (the same two accumulator chains)."""
    import torch

    rng = np.random.default_rng(d + nq + k + metric)
    n = 150_000
    xb = rng.standard_normal((n, d)).astype(np.float32)
    xq = rng.standard_normal((nq, d)).astype(np.float32)
    if metric == ko.METRIC_INNER_PRODUCT:  # cosine: normalised rows, as create_search_index("cosine") builds them
        xb, xq = ko.normalize_rows(xb), ko.normalize_rows(xq)
    rb = torch.from_numpy(xb).to(torch.bfloat16).to(torch.float32).numpy()
    rq = torch.from_numpy(xq).to(torch.bfloat16).to(torch.float32).numpy()
    index = faiss.IndexFlat(d, metric, storage="bf16")
    index.add(xb)
    D, I = index.search(xq, k)
    assert index.exact_stats()["gemm_chunks"] == 1
    D_ref, I_ref = ko.knn_exact(rb, rq, k, metric)
    assert_knn_matches(D, I, D_ref, I_ref, rb, rq, metric, gap=ko.kth_gap(rb, rq, k, metric))
    if metric == ko.METRIC_INNER_PRODUCT:
        Ds = np.concatenate([index.search(xq[i:i + 64], k)[0] for i in range(0, nq, 64)])
        Is = np.concatenate([index.search(xq[i:i + 64], k)[1] for i in range(0, nq, 64)])
        assert np.array_equal(Is, I) and np.array_equal(Ds, D)


def test_bf16_large_batch_overflow_reruns_through_the_streaming_passes(faiss):
    """5000 copies of one row: every copy passes any query's admit threshold, the candidate buffers
    overflow, and the streaming passes queued behind the GEMM path (gated on the overflow) answer
    instead -- lowest ids first."""
    import torch

    rng = np.random.default_rng(17)
    n, d, nq, k = 140_000, 128, 256, 10
    xb = ko.normalize_rows(rng.standard_normal((n, d)).astype(np.float32))
    dup = np.sort(rng.choice(n, 5000, replace=False))
    xb[dup] = xb[dup[0]]
    xq = np.ascontiguousarray(np.repeat(xb[dup[0]][None, :], nq, axis=0) + 0.001 * rng.standard_normal((nq, d)).astype(np.float32))
    rb = torch.from_numpy(xb).to(torch.bfloat16).to(torch.float32).numpy()
    rq = torch.from_numpy(xq).to(torch.bfloat16).to(torch.float32).numpy()
    index = faiss.IndexFlatIP(d, storage="bf16")
    index.add(xb)
    D, I = index.search(xq, k)
    assert index.exact_stats()["gemm_chunks"] == 1
    assert (I == dup[:k][None, :]).all(), "ties must resolve to the lowest ids"
    D_ref, I_ref = ko.knn_exact(rb, rq, k, ko.METRIC_INNER_PRODUCT)
    assert_knn_matches(D, I, D_ref, I_ref, rb, rq, ko.METRIC_INNER_PRODUCT)


@pytest.mark.parametrize("d,nq,k", [(512, 256, 10), (512, 1024, 10), (128, 300, 32), (384, 257, 1), (256, 1100, 5)])
def test_f32_inner_product_large_batch_gemm_path(faiss, d, nq, k):
    """The reference's default index type is "cosine" -- IndexFlatIP over normalised rows
    (backend/utils.py:293,300-303; backend/siamese/test_index.py:52-56).  Batches of >= 256 queries against float32
    inner-product rows take the GEMM-shaped pass (csrc/ise_gemm_scan.hpp, IPM) instead of re-reading the index per 48
    queries: same answers as the oracle, and bit for bit those of the streaming passes (the kernel sums a dot product
    in scan_kernel's order), which batches of 128 still take."""
    rng = np.random.default_rng(d + nq + k)
    n = 150_000
    xb = ko.normalize_rows(rng.standard_normal((n, d)).astype(np.float32))
    xq = ko.normalize_rows(rng.standard_normal((nq, d)).astype(np.float32))
    IP = ko.METRIC_INNER_PRODUCT
    index = faiss.IndexFlatIP(d)
    index.add(xb)
    D, I = index.search(xq, k)
    assert index.exact_stats()["gemm_chunks"] == (nq + 1023) // 1024
    D_ref, I_ref = ko.knn_exact(xb, xq, k, IP)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, IP, gap=ko.kth_gap(xb, xq, k, IP))
    before = index.exact_stats()["gemm_chunks"]
    Ds = np.concatenate([index.search(xq[i:i + 128], k)[0] for i in range(0, nq, 128)])
    Is = np.concatenate([index.search(xq[i:i + 128], k)[1] for i in range(0, nq, 128)])
    assert index.exact_stats()["gemm_chunks"] == before, "batches of 128 must take the streaming passes"
    assert np.array_equal(Is, I) and np.array_equal(Ds, D)


def test_f32_inner_product_large_batch_overflow_reruns_through_the_streaming_passes(faiss):
    """5000 copies of one row: every copy passes any query's admit threshold, the candidate buffers overflow, and
    the streaming passes queued behind the GEMM-shaped pass (gated on the overflow) answer instead -- lowest ids first."""
    rng = np.random.default_rng(18)
    n, d, nq, k = 140_000, 128, 256, 10
    IP = ko.METRIC_INNER_PRODUCT
    xb = ko.normalize_rows(rng.standard_normal((n, d)).astype(np.float32))
    dup = np.sort(rng.choice(n, 5000, replace=False))
    xb[dup] = xb[dup[0]]
    xq = np.ascontiguousarray(np.repeat(xb[dup[0]][None, :], nq, axis=0) + 0.001 * rng.standard_normal((nq, d)).astype(np.float32))
    index = faiss.IndexFlatIP(d)
    index.add(xb)
    D, I = index.search(xq, k)
    assert index.exact_stats()["gemm_chunks"] == 1
    assert (I == dup[:k][None, :]).all(), "ties must resolve to the lowest ids"
    D_ref, I_ref = ko.knn_exact(xb, xq, k, IP)
    assert_knn_matches(D, I, D_ref, I_ref, xb, xq, IP)
