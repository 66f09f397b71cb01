"""GPU tests of the batched CNN extractor and the indexer -> engine flow
(backend/indexer.py:51-59, backend/engine.py:46-65) on synthetic images."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cnn_gpu_matches_cpu_fp32_reference():
    """Numerics of the device path against plain PyTorch fp32 on the CPU (same weights)."""
    from image_search_engine_amd.descriptors import CNNDescriptor

    gpu = CNNDescriptor(device="cuda")
    cpu = CNNDescriptor(device="cpu")
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(6)]
    imgs.append(rng.integers(0, 256, (333, 500, 3), dtype=np.uint8))
    fg = gpu.describe_batch(imgs).numpy()
    fc = cpu.describe_batch(imgs).numpy()
    assert fg.shape == (7, 2048)
    # fp32 convolutions on MIOpen vs oneDNN: relative error a few 1e-5 after 50 layers
    scale = np.abs(fc).max()
    assert np.abs(fg - fc).max() <= 2e-3 * scale
    cos = (fg * fc).sum(1) / (np.linalg.norm(fg, axis=1) * np.linalg.norm(fc, axis=1))
    assert (cos > 0.99999).all()
    one = gpu.describe(imgs[0])
    assert one.shape == (2048,) and not one.is_cuda
    np.testing.assert_allclose(one.numpy(), fg[0], rtol=1e-3, atol=1e-3 * scale)
    dev = gpu.extract_features_batch(imgs[:2])
    assert dev.is_cuda and dev.shape == (2, 2048)


def test_batched_preprocess_equals_the_one_image_path():
    """The staged upload (one pinned copy per batch, images of one size resized together) gives, row for
    row and in input order, exactly what preprocessing each image on its own gives -- mixed sizes,
    neighbours and non-neighbours of the same size, an image already 224 x 224, several batches in a row
    (the two staging buffers are reused)."""
    from image_search_engine_amd.descriptors import CNNDescriptor

    gpu = CNNDescriptor(device="cuda")
    rng = np.random.default_rng(5)
    shapes = [(375, 500), (375, 500), (224, 224), (500, 375), (375, 500), (64, 48), (224, 224), (375, 500)]
    for rnd in range(3):
        imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes[rnd:] + shapes[:rnd]]
        xb = gpu.preprocessor(imgs)
        assert xb.shape == (len(imgs), 3, 224, 224) and xb.is_contiguous(memory_format=torch.channels_last)
        for i, im in enumerate(imgs):
            assert torch.equal(xb[i], gpu.preprocessor([im])[0]), (rnd, i)
    cpu = CNNDescriptor(device="cpu")
    xc = cpu.preprocessor(imgs)
    assert torch.allclose(xb.cpu(), xc, atol=2e-5)   # bilinear weights on two devices: float rounding only
    assert gpu.preprocessor([]).shape == (0, 3, 224, 224)
    with pytest.raises(ValueError):
        gpu.preprocessor([np.zeros((8, 8), np.uint8)])


def test_small_batches_replay_a_graph_of_the_network(monkeypatch):
    """Batches of 1, 2, 4 ... 32 images go through a captured HIP graph (descriptors.CNNDescriptor._forward):
    same features as the eager network, for float32 and for bf16 autocast, replay after replay with
    different inputs, from several threads; other batch sizes and config.CNN_GRAPHS = False run eagerly."""
    import threading

    from image_search_engine_amd.config import Config
    from image_search_engine_amd.descriptors import CNNDescriptor

    rng = np.random.default_rng(6)
    for dtype, tol in ((torch.bfloat16, 2e-2), (torch.float32, 1e-4)):   # the float32 one serves the thread part below
        gpu = CNNDescriptor(device="cuda", dtype=dtype)
        for b in (1, 2, 3, 8, 32):
            imgs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(b)]
            with torch.no_grad():
                x = gpu.preprocessor(imgs)
                want = gpu._forward_eager(x)
                got = [gpu._forward(x).clone() for _ in range(2)]
                again = gpu._forward(gpu.preprocessor(imgs[::-1]))   # the same graph, other pixels
            scale = float(want.abs().max())
            for g in got:
                assert torch.allclose(g, want, rtol=tol, atol=tol * scale), (dtype, b)
            assert torch.allclose(again.flip(0), want, rtol=tol, atol=tol * scale)
            assert (b in gpu._graphs and gpu._graphs[b] is not None) == (b in gpu.GRAPH_BATCHES), (b, list(gpu._graphs))
    # threads replaying the batch-1 graph at once each get their own image's features
    yy, xx = np.mgrid[0:224, 0:224]
    imgs = [np.stack([(xx * (i + 1)) % 256, (yy * (8 - i)) % 256, np.full_like(xx, 30 * i)], -1).astype(np.uint8)
            for i in range(8)]                                  # structured and pairwise different
    monkeypatch.setattr(Config, "DESCRIBE_COMBINE_MAX", 0)   # every describe() is its own forward pass here
    want = [gpu.describe(im) for im in imgs]
    scale = max(float(w.abs().max()) for w in want)
    close = lambda a, b: torch.allclose(a, b, rtol=1e-3, atol=1e-3 * scale)  # noqa: E731 -- replays of one graph
    again = [gpu.describe(im) for im in imgs]
    assert all(close(a, w) for a, w in zip(again, want)), [float((a - w).abs().max()) / scale for a, w in zip(again, want)]
    assert not any(close(want[i], want[j]) for i in range(8) for j in range(i))   # a mix-up would show
    out = {}

    def work(i):
        out[i] = [gpu.describe(imgs[i]) for _ in range(5)]

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    bad = [(i, float((f - want[i]).abs().max())) for i in range(8) for f in out[i] if not close(f, want[i])]
    assert not bad, (bad, scale)
    monkeypatch.setattr(Config, "CNN_GRAPHS", False)
    eager = CNNDescriptor(device="cuda", dtype=torch.bfloat16)
    assert torch.allclose(eager.describe(imgs[0]), want[0], rtol=2e-2, atol=2e-2 * float(want[0].abs().max()))
    assert eager._graphs == {}


def test_indexer_then_engine_roundtrip(tmp_path, monkeypatch):
    from PIL import Image

    from image_search_engine_amd import engine, indexer
    from image_search_engine_amd.config import Config
    from image_search_engine_amd.descriptors import CNNDescriptor

    data = tmp_path / "data"
    data.mkdir()
    rng = np.random.default_rng(5)
    for i in range(24):
        arr = rng.integers(0, 256, (64 + 8 * (i % 3), 96, 3), dtype=np.uint8)
        Image.fromarray(arr).save(data / f"img_{i:02d}.png")
    monkeypatch.setattr(Config, "DATA_FOLDER_PATH", data)
    monkeypatch.setattr(Config, "MODELS_BASE_PATH", tmp_path / "models")
    monkeypatch.setattr(Config, "DNN_INDEX_PATH", tmp_path / "models" / "resnet50_dnn_index.faiss")
    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", tmp_path / "models" / "none.joblib")
    monkeypatch.setattr(Config, "DNN_BATCH_SIZE", 8)
    index = indexer.main()
    assert index.ntotal == 24 and index.d == 2048
    assert (tmp_path / "models" / "resnet50_dnn_index.faiss").exists()

    desc = CNNDescriptor(model=Config.DNN_MODEL)
    engine.load(desc=desc)
    assert engine.index.ntotal == 24 and len(engine.images_paths) == 24
    from image_search_engine_amd.descriptors import Describer

    target = engine.images_paths[7]
    feats = desc.describe(Describer({"x": desc}).read_image(target))
    preds = engine.run_image_query(feats, 5)
    assert len(preds) == 5
    dist, thumb, path = preds[0]
    assert path == str(target) and dist <= 1e-2 * max(1.0, preds[1][0]) and isinstance(thumb, str)
    assert all(preds[i][0] <= preds[i + 1][0] for i in range(4))
    # batched query form returns the same neighbours
    many = engine.run_image_queries(torch.stack([feats, feats]), 5)
    assert [p[2] for p in many[0]] == [p[2] for p in preds] == [p[2] for p in many[1]]
    # cosine index + normalised query (backend/utils.py:300-303, backend/engine.py:52-53)
    from image_search_engine_amd.utils import create_search_index

    xb = engine.index.reconstruct_n(0, 24)
    engine.index = create_search_index(xb, "cosine")
    np.testing.assert_allclose(np.linalg.norm(xb, axis=1), 1.0, rtol=1e-5)  # normalised in place (quirk 5.9-6)
    preds = engine.run_image_query(feats, 3, normalize=True)
    assert preds[0][2] == str(target) and abs(preds[0][0] - 1.0) < 1e-4


def test_faiss_kmeans_transform_is_nearest_centroid(golden_dir):
    """backend/kmeans_faiss.py:46-50 through the shim: spherical -> IP index over unit centroids."""
    import os

    from image_search_engine_amd.kmeans_faiss import FaissKMeans

    import image_search_engine_amd.faiss_compat as faiss

    z = np.load(os.path.join(golden_dir, "assign_ip_n512_c256_d128.npz"))
    # the reference reloads a saved codebook as FaissKMeans(n_clusters, index=read_index(...))
    # (backend/bag_of_visual_words.py:207-216); spherical k-means -> inner-product index
    index = faiss.IndexFlatIP(128)
    index.add(z["xb"])
    km = FaissKMeans(n_clusters=256, index=index)
    assert km.index.ntotal == 256
    I = km.transform(z["xq"])
    assert I.dtype == np.int64 and I.shape == (512, 1)
    assert np.array_equal(I, z["I"])
    hist, _ = np.histogram(I, bins=256)  # BoVW histogram as at backend/bag_of_visual_words.py:103
    assert hist.sum() == 512


def test_kmeans_train_lloyd_on_gpu():
    """spherical k-means as backend/kmeans_faiss.py:29-44 configures it.  Planted
    well-separated directions must be recovered; the objective must not decrease."""
    import image_search_engine_amd.faiss_compat as faiss
    from image_search_engine_amd.kmeans_faiss import FaissKMeans

    rng = np.random.default_rng(0)
    K, d, per = 12, 64, 300
    dirs = rng.standard_normal((K, d)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    X = np.concatenate([dirs[i] * rng.uniform(50, 100, (per, 1)) + 0.5 * rng.standard_normal((per, d))
                        for i in range(K)]).astype(np.float32)
    km = FaissKMeans(n_clusters=K, n_init=3, max_iter=25)
    km.fit(X)
    assert km.cluster_centers_.shape == (K, d)
    np.testing.assert_allclose(np.linalg.norm(km.cluster_centers_, axis=1), 1.0, rtol=1e-5)  # spherical
    labels = km.transform(X).ravel()
    truth = np.repeat(np.arange(K), per)
    # Lloyd from random rows may stop in a local optimum (as Faiss's does): ask for high purity,
    # and for exact recovery when started from the planted directions
    purity = sum(np.bincount(truth[labels == c]).max() for c in set(labels.tolist())) / len(labels)
    assert purity > 0.75, purity
    seeded = FaissKMeans(n_clusters=K, max_iter=5, init_centroids=dirs)
    seeded.fit(X)
    ls = seeded.transform(X).ravel()
    assert np.array_equal(ls, truth)
    obj = km.kmeans.obj
    assert (np.diff(obj) >= -1e-3 * np.abs(obj[:-1])).all()  # summed inner product never drops
    assert abs(km.inertia_ - obj[-1]) < 1e-6 * abs(obj[-1]) + 1e-6
    # plain (non-spherical) k-means through the same class
    plain = faiss.Kmeans(d=d, k=K, niter=10, nredo=1, seed=1, spherical=False)
    plain.train(X)
    _, I = plain.index.search(X, 1)
    assert len(set(I.ravel().tolist())) == K


@pytest.mark.parametrize("spherical", [False, True])
def test_kmeans_train_matches_the_oracle_iteration(spherical):
    """from given initial centroids (``init_centroids``, backend/kmeans_faiss.py:41) the trainer's
    iterations are the oracle's float64 Lloyd iterations (oracle/knn_oracle.py::lloyd_reference, itself checked
    against sklearn in tests/test_oracle.py): same centroids, same objective per iteration (``Kmeans.obj``),
    same final assignment -- also at the size where the MFMA assignment kernel takes over (n >= 2048).  What
    stays unpinned is Faiss's own seeding and its re-seeding of empty clusters (neither occurs here)."""
    import image_search_engine_amd.faiss_compat as faiss
    from oracle import knn_oracle as ko

    rng = np.random.default_rng(21)
    K, d, per = 24, 48, 150                     # n = 3600: the k = 1 assignment kernel runs the E-step
    centres = rng.standard_normal((K, d)) * 4.0
    x = np.concatenate([centres[i] + rng.standard_normal((per, d)) for i in range(K)]).astype(np.float32)
    c0 = (centres + rng.standard_normal((K, d)) * 0.8).astype(np.float32)
    niter = 6
    km = faiss.Kmeans(d=d, k=K, niter=niter, nredo=1, seed=5, spherical=spherical)
    last = km.train(x, init_centroids=c0)
    c_ref, obj_ref, lab_ref = ko.lloyd_reference(x, c0, niter, spherical=spherical)
    np.testing.assert_allclose(km.centroids, c_ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(km.obj, obj_ref, rtol=2e-5)
    assert km.obj.shape == (niter,) and abs(last - obj_ref[-1]) <= 2e-5 * abs(obj_ref[-1])
    # the index the reference then searches (kmeans_faiss.py:49) holds those centroids
    _, I = km.index.search(x, 1)
    c_fin = c_ref
    want = ko.assign_nearest(x, c_fin.astype(np.float32), ko.METRIC_INNER_PRODUCT if spherical else ko.METRIC_L2)
    assert (I.ravel() == np.asarray(want).ravel()).mean() > 0.999   # rows on a cell boundary may flip at 1e-5


def test_query_index_both_branches_match_reference_formulas():
    """backend/siamese/test_index.py:49-71: faiss branch = normalise + IP search; dict branch =
    normalise + per-row np.linalg.norm + argsort (non-squared L2, returns (indices, distances))."""
    import image_search_engine_amd.faiss_compat as faiss
    from image_search_engine_amd.query_index import query_index

    rng = np.random.default_rng(12)
    rows = rng.standard_normal((400, 128))
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)  # float64 unit rows (create_index.py:62-85)
    emb = rng.standard_normal((1, 128))
    # reference arithmetic, restated inline
    e = emb / np.linalg.norm(emb)
    dist = np.array([np.linalg.norm(rows[i, :] - e) for i in range(len(rows))])
    order = dist.argsort()[:9]
    ind, dis = query_index(emb.copy(), rows, "dict", 9)
    assert np.array_equal(np.asarray(ind), order)
    np.testing.assert_allclose(dis, dist[order], rtol=1e-5)
    # faiss branch over an inner-product index of the same rows: same neighbours, cos = 1 - d^2/2
    index = faiss.IndexFlatIP(128)
    index.add(rows.astype(np.float32))
    e32 = emb.astype(np.float32).copy()
    ind2, dis2 = query_index(e32, index, "faiss", 9)
    assert ind2 == order.tolist()
    np.testing.assert_allclose(np.linalg.norm(e32), 1.0, rtol=1e-6)  # normalised in place
    np.testing.assert_allclose(dis2, 1.0 - dist[order] ** 2 / 2.0, atol=1e-5)


def test_bench_contract_small():
    """bench.py keeps its JSON contract (one line, required keys, roofline + cpu_baseline)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--n", "60000", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["dtype"] == "f32" and r["config"]["workload"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(r["cpu_baseline"])
    assert r["ids_identical"] and r["recall_at_k"] == 1.0
    lat = r["batch_latency_us"]
    assert 0 < lat["p10"] <= lat["median"] <= lat["p90"]


# ---- visual-word histograms: one batched assignment + one histogram kernel
# against the reference's own loop -- clusterer.transform per image + np.histogram(bins=K)
def _reference_histograms(images_descriptions, labels_per_image, K):
    out = np.zeros((len(images_descriptions), K))
    for i, ids in enumerate(labels_per_image):
        out[i], _ = np.histogram(ids, bins=K)  # backend/bag_of_visual_words.py:103, no range=
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("K,d", [(200, 32), (4096, 128), (7, 16)])
def test_bovw_histograms_equal_numpy_loop(K, d):
    import image_search_engine_amd.faiss_compat as faiss
    from image_search_engine_amd import bag_of_visual_words as bovw
    from image_search_engine_amd.kmeans_faiss import FaissKMeans
    from oracle import knn_oracle as ko

    rng = np.random.default_rng(K)
    cent = rng.standard_normal((K, d)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    index = faiss.IndexFlatIP(d)
    index.add(cent)
    clusterer = FaissKMeans(K, index=index)
    # ragged images: empty, one keypoint, all keypoints in one cluster, few clusters, many keypoints
    lens = [0, 1, 5, 40, 300, 1000, 2500, 3, 0, 64]
    images = [rng.standard_normal((m, d)).astype(np.float32) for m in lens]
    images[2] = np.repeat(cent[K // 2][None] * 3.0, 5, axis=0)              # lo == hi
    images[3] = cent[rng.integers(K // 3, K // 3 + 3, 40)] * 2.0             # narrow label range
    images[7] = cent[[0, K - 1, K - 1]] * 1.5                                # hits both ends
    H = bovw.create_visual_word_histogram(images, clusterer, K)
    assert H.dtype == np.float64 and H.shape == (len(images), K)
    labels = [ko.assign_nearest(X, cent) if len(X) else np.zeros((0, 1), np.int64) for X in images]
    Href = _reference_histograms(images, labels, K)
    assert np.array_equal(H, Href)
    assert H.sum(axis=1).tolist() == [float(m) for m in lens]
    # per-image transform (the reference's call pattern) gives the same labels as the batch
    for X, lab in zip(images, labels):
        if len(X):
            assert np.array_equal(clusterer.transform(X), lab)


@pytest.mark.gpu
def test_bovw_histograms_batched_chunks_and_label_sweep(monkeypatch):
    """Many small images across several upload batches; every (lo, hi) label span of a small
    codebook, so every bin-edge rounding case of numpy's uniform-bin path is visited."""
    import torch

    import image_search_engine_amd.faiss_compat as faiss
    from image_search_engine_amd import _native as _n
    from image_search_engine_amd import bag_of_visual_words as bovw
    from image_search_engine_amd.kmeans_faiss import FaissKMeans

    K = 200
    # the kernel alone on synthetic labels: all spans [lo, hi] with every label in between
    spans = [(lo, hi) for lo in range(0, K, 7) for hi in range(lo, K, 5)]
    labs = [np.arange(lo, hi + 1, dtype=np.int64) for lo, hi in spans]
    offsets = np.zeros(len(labs) + 1, np.int64)
    np.cumsum([len(a) for a in labs], out=offsets[1:])
    dev = torch.device("cuda", 0)
    lab_d = torch.from_numpy(np.concatenate(labs)).to(dev)
    off_d = torch.from_numpy(offsets).to(dev)
    for bins in (K, 64, 4096, 3):
        out = torch.empty((len(labs), bins), dtype=torch.float64, device=dev)
        _n.check(_n.lib.ise_bovw_histogram_device(lab_d.data_ptr(), off_d.data_ptr(), len(labs), bins,
                                                  out.data_ptr(), 0, torch.cuda.current_stream(dev).cuda_stream))
        ref = np.stack([np.histogram(a, bins=bins)[0] for a in labs]).astype(np.float64)
        assert np.array_equal(out.cpu().numpy(), ref), bins

    # batching: force several uploads
    monkeypatch.setattr(bovw, "ROWS_PER_BATCH", 500)
    rng = np.random.default_rng(5)
    d = 24
    cent = rng.standard_normal((K, d)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    index = faiss.IndexFlatIP(d)
    index.add(cent)
    clusterer = FaissKMeans(K, index=index)
    images = [rng.standard_normal((int(m), d)).astype(np.float32) for m in rng.integers(0, 300, 40)]
    images[4] = rng.standard_normal((900, d)).astype(np.float32)  # one image larger than a batch
    H = bovw.create_visual_word_histogram(images, clusterer, K)
    ref = np.stack([np.histogram(clusterer.transform(X), bins=K)[0] if len(X) else np.zeros(K) for X in images])
    assert np.array_equal(H, ref.astype(np.float64))
    assert np.array_equal(bovw.create_visual_word_histogram([], clusterer, K), np.zeros((0, K)))


def test_config2_cnn_embeddings_to_l2_index_end_to_end():
    """BASELINE config 2 at a size the oracle covers: seeded random-init ResNet-50 (out_dim=512) on
    seeded synthetic uint8 images -> create_search_index("l2") -> search; neighbour ids identical
    to the exact CPU oracle on the very embeddings the index holds (backend/indexer.py:51-59,
    backend/engine.py:46-57).  Post-ReLU CNN features share a large common component (|y|^2 of the
    order 1e5 against neighbour distances below 1): the data that defeats an expanded-form kernel."""
    from image_search_engine_amd.descriptors import CNNDescriptor
    from image_search_engine_amd.utils import create_search_index
    from oracle import knn_oracle as ko
    from tests.knn_checks import assert_knn_matches

    n, nq, k, d = 3000, 16, 10, 512
    desc = CNNDescriptor(out_dim=d, seed=0)
    g = torch.Generator(device="cuda").manual_seed(1234)
    feats = []
    for i0 in range(0, n + nq, 250):
        imgs = torch.randint(0, 256, (min(250, n + nq - i0), 224, 224, 3), generator=g, device="cuda",
                             dtype=torch.uint8)
        feats.append(desc.extract_features_tensor(imgs).cpu())
    feats = torch.cat(feats).numpy()
    xb, xq = np.ascontiguousarray(feats[:n]), np.ascontiguousarray(feats[n:])
    assert xb.shape == (n, d) and np.isfinite(xb).all()
    index = create_search_index(xb.copy(), index_type="l2")
    D, I = index.search(xq, k)
    D_ref, I_ref = ko.knn_exact(xb, xq, k, ko.METRIC_L2)
    n_mism = assert_knn_matches(D, I, D_ref, I_ref, xb, xq, ko.METRIC_L2, gap=ko.kth_gap(xb, xq, k, ko.METRIC_L2))
    assert n_mism == 0
    # a row of the index queried against itself comes back first at distance 0
    D0, I0 = index.search(xb[17:18], 1)
    assert I0[0, 0] == 17 and D0[0, 0] == 0.0
    st = index.exact_stats()
    # both batches were scanned by the short-index kernel and re-ranked (3000 rows: above the one-query direct
    # scan's 128 row tiles)
    assert st["reranked"] == nq + 1 and index.short_stats() == {"short_batches": 2}
    assert index.host_stats()["direct_queries"] == 0


def test_config2_at_full_size():
    """BASELINE config 2 at its full size: 100 000 seeded synthetic images -> seeded random-init ResNet-50
    (out_dim = 512) -> L2 index -> k = 10 search, nq = 1, 16 and 1024, ids against the C oracle on the very
    embeddings the index holds (the float64 numpy oracle covers the 16-query batch as well)."""
    from image_search_engine_amd.descriptors import CNNDescriptor
    from image_search_engine_amd.utils import create_search_index
    from oracle import flat_oracle as fo
    from oracle import knn_oracle as ko
    from tests.knn_checks import assert_knn_matches

    n, d, k = 100_000, 512, 10
    desc = CNNDescriptor(out_dim=d, seed=0)
    g = torch.Generator(device="cuda").manual_seed(99)
    feats = []
    for i0 in range(0, n + 1024, 512):
        imgs = torch.randint(0, 256, (min(512, n + 1024 - i0), 224, 224, 3), generator=g, device="cuda",
                             dtype=torch.uint8)
        feats.append(desc.extract_features_tensor(imgs).cpu())
    feats = torch.cat(feats).numpy()
    xb, xq = np.ascontiguousarray(feats[:n]), np.ascontiguousarray(feats[n:])
    assert xb.shape == (n, d) and xq.shape == (1024, d) and np.isfinite(feats).all()
    index = create_search_index(xb.copy(), index_type="l2")
    assert index.ntotal == n
    for nq in (1, 16, 1024):
        D, I = index.search(xq[:nq], k)
        D_ref, I_ref, _ = fo.knn_flat(xb, xq[:nq], k, ko.METRIC_L2, 16)
        assert_knn_matches(D, I, D_ref, I_ref, xb, xq[:nq], ko.METRIC_L2)
    D64, I64 = ko.knn_exact(xb, xq[:16], k, ko.METRIC_L2)
    D, I = index.search(xq[:16], k)
    assert assert_knn_matches(D, I, D64, I64, xb, xq[:16], ko.METRIC_L2, gap=ko.kth_gap(xb, xq[:16], k, ko.METRIC_L2)) == 0
