"""Shared comparison helpers for the kNN parity tests (tests only)."""
import numpy as np

from oracle import knn_oracle as ko

RTOL = 1e-4  # north_star: distances within 1e-4 (fp32); applied as 1e-4 * max(1, |D|)


def true_scores(xb, xq, ids, metric):
    """float64 score of (query q, row ids[q, r]); -1 ids give nan."""
    out = np.full(ids.shape, np.nan)
    b = xb.astype(np.float64)
    for q in range(ids.shape[0]):
        ok = ids[q] >= 0
        rows = b[ids[q][ok]]
        x = xq[q].astype(np.float64)
        out[q, ok] = ((rows - x) ** 2).sum(1) if metric == ko.METRIC_L2 else rows @ x
    return out


ATOL_UNIFORM = 1e-4  # north_star's absolute bound, asserted where it is attainable: uniform[0,1) data


def assert_knn_matches(D, I, D_ref, I_ref, xb, xq, metric, gap=None, rtol=RTOL, atol=None):
    """Bit-exact ids wherever the float64 ranking is unambiguous; where two
    consecutive oracle ranks are closer than float32 can resolve, the returned
    row must still be a true near-tie at that rank.  Distances always within
    rtol * max(1, |D_ref|); with ``atol`` additionally within that ABSOLUTE bound
    (north_star: 1e-4 -- used on uniform[0,1) data, where |D| ~ d/6 keeps it above
    float32 rounding)."""
    assert D.dtype == np.float32 and I.dtype == np.int64
    assert D.shape == D_ref.shape and I.shape == I_ref.shape
    pad = I_ref < 0
    assert np.array_equal(I < 0, pad), "padding (-1) positions differ"
    assert np.array_equal(D[pad], D_ref[pad]), "padding distances must be +-FLT_MAX"
    tol = rtol * np.maximum(1.0, np.abs(D_ref.astype(np.float64)))
    err = np.abs(D.astype(np.float64) - D_ref.astype(np.float64))
    assert (err[~pad] <= tol[~pad]).all(), f"distance error {err[~pad].max():.3e} above tolerance"
    if atol is not None and (~pad).any():
        assert err[~pad].max() <= atol, f"distance error {err[~pad].max():.3e} above the absolute bound {atol:g}"
    for q in range(I.shape[0]):
        row = I[q][I[q] >= 0]
        assert len(set(row.tolist())) == len(row), "duplicate ids in one result row"
    mism = (I != I_ref) & ~pad
    if mism.any():
        ts = true_scores(xb, xq, I, metric)
        ref = D_ref.astype(np.float64)
        near = np.abs(ts - ref) <= 2e-6 * np.maximum(1.0, np.abs(ref)) + 1e-6
        bad = mism & ~near
        assert not bad.any(), (
            f"{bad.sum()} id mismatches that are not float32 near-ties, e.g. q={np.argwhere(bad)[0]}")
        if gap is not None:
            qs = np.unique(np.argwhere(mism)[:, 0])
            # "well separated" is relative to what float32 resolves at the distances' magnitude
            # (SIFT-valued descriptors: |D| ~ 3e6, one float32 ulp = 0.25)
            lim = 1e-4 + 2e-6 * np.abs(D_ref.astype(np.float64)).max(axis=1)
            assert (gap[qs] < lim[qs]).all(), "id mismatch on a query whose ranks are well separated"
    return int(mism.sum())


def seeded_inputs(kind: str, seed: int, n: int, d: int, nq: int):
    """Inputs of a SEEDED golden fixture (tests/golden/seeded_*.npz): the fixture stores the seed and
    the expected (I, D, gap) only -- SURVEY.md 8c's full-size cases (4096 x 512, 2048 x 128 vs 256)
    would be megabytes as stored arrays -- and the inputs are regenerated here, bit for bit."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d), dtype=np.float32), rng.random((nq, d), dtype=np.float32)
    if kind == "uniform_unit":   # IP on L2-normalised rows (create_search_index("cosine"))
        xb, xq = rng.random((n, d), dtype=np.float32), rng.random((nq, d), dtype=np.float32)
        return ko.normalize_rows(xb), ko.normalize_rows(xq)
    if kind == "assign":         # k = 1 assignment: n unit-norm centroids, nq SIFT-valued descriptors
        cent = ko.normalize_rows(rng.standard_normal((n, d)).astype(np.float32))
        return cent, rng.integers(0, 256, (nq, d)).astype(np.float32)
    raise ValueError(kind)


def load_fixture(path):
    """-> dict with xb, xq, k, metric, D, I, gap for stored and for seeded fixtures alike."""
    z = np.load(path)
    out = {key: z[key] for key in z.files}
    if "seed" in out:
        out["xb"], out["xq"] = seeded_inputs(str(out["kind"]), int(out["seed"]), int(out["n"]), int(out["d"]),
                                             int(out["nq"]))
    return out
