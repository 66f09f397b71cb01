"""The serving wire format (backend/engine.py:68-107 as the front end reads it,
frontend/src/App.js:14-21,38-45) and the persisted row-id -> path list.

CPU tests drive the Flask route with test doubles standing in for the descriptor and the index (the
route itself is host logic); the GPU test runs it end to end on the MI355X: indexer.main ->
engine.load -> POST /similar_images."""
import base64
import io
import json

import numpy as np
import pytest

from image_search_engine_amd.config import Config
from oracle import knn_oracle as ko


def _png_bytes(arr_rgb):
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(arr_rgb).save(buf, format="PNG")
    return buf.getvalue()


def _write_images(folder, n, rng, size=48):
    from PIL import Image

    folder.mkdir(parents=True, exist_ok=True)
    paths = []
    for i in range(n):
        arr = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
        p = folder / f"img_{i:04d}.png"
        Image.fromarray(arr).save(p)
        paths.append(p)
    return paths


class _OracleIndex:
    """Test double for faiss.IndexFlatL2 (tests only): exact search by the CPU oracle."""

    def __init__(self, xb):
        self.xb, self.d, self.ntotal = xb, xb.shape[1], xb.shape[0]

    def search(self, x, k):
        return ko.knn_exact(self.xb, np.asarray(x, dtype=np.float32), k, ko.METRIC_L2)


class _MeanColourDescriptor:
    """Test double for CNNDescriptor: the mean BGR colour tiled to d values, as a flat tensor."""

    def __init__(self, d=12):
        self.d = d
        self.seen = []

    def describe(self, image):
        import torch

        self.seen.append(image)
        return torch.from_numpy(np.tile(image.reshape(-1, 3).mean(0), self.d // 3).astype(np.float32))


@pytest.fixture
def served(tmp_path):
    pytest.importorskip("flask")
    from image_search_engine_amd import engine

    rng = np.random.default_rng(0)
    paths = _write_images(tmp_path / "data", 30, rng)
    desc = _MeanColourDescriptor()
    from PIL import Image

    feats = np.stack([desc.describe(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]).numpy() for p in paths])
    saved = (engine.index, engine.images_paths, engine.descriptor)
    engine.index, engine.images_paths, engine.descriptor = _OracleIndex(feats), paths + [tmp_path / "gone.png"], desc
    yield engine, paths, feats
    engine.index, engine.images_paths, engine.descriptor = saved


def test_similar_images_wire_format(served):
    engine, paths, feats = served
    client = engine.create_app().test_client()
    from PIL import Image

    rgb = np.asarray(Image.open(paths[7]).convert("RGB"))
    resp = client.post("/similar_images", data={"image": (io.BytesIO(_png_bytes(rgb)), "query.png")},
                       content_type="multipart/form-data")
    assert resp.status_code == 200 and resp.mimetype == "application/json"
    assert resp.headers["Access-Control-Allow-Origin"] == "*"          # the reference enables CORS (engine.py:21)
    body = json.loads(resp.data)
    assert list(body.keys()) == ["prediction"]
    pred = body["prediction"]
    assert len(pred) == engine.config.NUM_IMAGES_TO_RETURN == 20      # backend/config.py:39
    # every entry is [distance, base64 thumbnail | null, path] -- what App.js destructures as [dist, im, path]
    for dist, im, path in pred:
        assert isinstance(dist, float) and isinstance(path, str) and (im is None or isinstance(im, str))
    assert pred[0][2] == str(paths[7]) and pred[0][0] == 0.0           # the uploaded image is its own best hit
    assert [p[0] for p in pred] == sorted(p[0] for p in pred)          # ascending squared L2
    thumb = Image.open(io.BytesIO(base64.decodebytes(pred[0][1].encode())))
    assert thumb.format == "JPEG" and max(thumb.size) <= 256           # backend/utils.py:44-62
    # the decoded upload reached the descriptor as BGR uint8 (cv2.imdecode order, backend/engine.py:42)
    assert np.array_equal(engine.descriptor.seen[-1], rgb[:, :, ::-1])
    D_ref, I_ref = ko.knn_exact(feats, feats[7:8], 20, ko.METRIC_L2)
    assert [p[2] for p in pred] == [str(paths[i]) for i in I_ref[0]]
    assert np.allclose([p[0] for p in pred], D_ref[0])


def test_similar_images_batch_endpoint(served):
    """/similar_images_batch: the field ``image`` repeated; every upload gets exactly the list /similar_images
    returns for it (one batched search behind it); 400 without files or with a non-image among them."""
    engine, paths, feats = served
    client = engine.create_app().test_client()
    from PIL import Image

    probes = [3, 11, 3]
    uploads = [(io.BytesIO(_png_bytes(np.asarray(Image.open(paths[i]).convert("RGB")))), f"q{j}.png")
               for j, i in enumerate(probes)]
    resp = client.post("/similar_images_batch", data={"image": uploads}, content_type="multipart/form-data")
    assert resp.status_code == 200 and resp.mimetype == "application/json"
    body = json.loads(resp.data)
    assert list(body.keys()) == ["predictions"] and len(body["predictions"]) == 3
    for i, pred in zip(probes, body["predictions"]):
        rgb = np.asarray(Image.open(paths[i]).convert("RGB"))
        one = client.post("/similar_images", data={"image": (io.BytesIO(_png_bytes(rgb)), "q.png")},
                          content_type="multipart/form-data")
        assert pred == json.loads(one.data)["prediction"]
        assert pred[0][2] == str(paths[i]) and pred[0][0] == 0.0
    assert client.post("/similar_images_batch").status_code == 400
    bad = client.post("/similar_images_batch", content_type="multipart/form-data",
                      data={"image": [(io.BytesIO(_png_bytes(rgb)), "ok.png"), (io.BytesIO(b"nope"), "bad.png")]})
    assert bad.status_code == 400 and b"bad.png" in bad.data


def test_similar_images_errors_and_missing_files(served):
    engine, paths, feats = served
    client = engine.create_app().test_client()
    assert client.post("/similar_images").status_code == 400                       # backend/engine.py:72-73
    assert client.post("/similar_images", data={"other": "x"}).status_code == 400
    bad = client.post("/similar_images", data={"image": (io.BytesIO(b"not an image"), "x.png")},
                      content_type="multipart/form-data")
    assert bad.status_code == 400
    assert client.get("/similar_images").status_code == 405
    # a hit whose file has vanished keeps its slot with a null thumbnail (backend/utils.py:51-54)
    paths[3].unlink()
    import torch

    pred = engine.run_image_query(torch.from_numpy(feats[3]), 2)   # a flat tensor, as descriptor.describe returns
    assert pred[0][1] is None and pred[0][2] == str(paths[3]) and pred[1][1] is not None
    batched = engine.run_image_queries(feats[2:5], 3)
    assert [len(b) for b in batched] == [3, 3, 3] and batched[1][0][2] == str(paths[3])


def test_describe_dataset_tracks_the_paths_it_described(tmp_path, monkeypatch):
    """A skipped image leaves no row: described_paths[i] is the path of row i (quirk 5.9-4)."""
    from image_search_engine_amd import descriptors as ds

    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", tmp_path / "absent.joblib")
    rng = np.random.default_rng(1)
    paths = _write_images(tmp_path / "data", 9, rng, size=16)
    paths[4].write_bytes(b"broken")                       # undecodable -> printed and skipped
    paths.insert(2, tmp_path / "data" / "missing.png")    # missing -> printed and skipped
    desc = _MeanColourDescriptor()
    describer = ds.Describer({"conv_features": desc}, batch_size=4)
    out = ds.describe_dataset(describer, np.array(paths).reshape(-1, 1))
    kept = [p for p in paths if p.name not in ("missing.png", paths[5].name)]
    assert len(out) == 8 and describer.described_paths == kept
    assert all(o.shape == (1, 12) for o in out)


def test_load_resnet50_weights_roundtrip(tmp_path):
    """CNNDescriptor(weights_path=...): a torchvision-style state_dict (with its fc head) is loaded
    before BatchNorm is folded and reproduces the source network's features."""
    import torch

    from image_search_engine_amd.descriptors import CNNDescriptor
    from image_search_engine_amd.resnet import load_resnet50_weights, resnet50_features

    src = resnet50_features(seed=123).eval()
    with torch.no_grad():  # non-trivial BatchNorm statistics, as a trained checkpoint has
        for m in src.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    state = dict(src.state_dict())
    state["fc.weight"] = torch.zeros(1000, 2048)
    state["fc.bias"] = torch.zeros(1000)
    path = tmp_path / "resnet50.pth"
    torch.save(state, path)
    x = torch.rand(2, 3, 64, 64)
    with torch.no_grad():
        want = src(x)
    desc = CNNDescriptor(seed=0, device="cpu", weights_path=path)   # folds BatchNorm after loading
    with torch.no_grad():
        got = desc.feature_extractor(x)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-3)          # values are O(100): folding changes rounding only
    assert not torch.allclose(CNNDescriptor(seed=0, device="cpu").feature_extractor(x), want, rtol=1e-2, atol=1.0)
    torch.save({"conv1.weight": torch.zeros(64, 3, 7, 7)}, tmp_path / "bad.pth")
    with pytest.raises(RuntimeError, match="not a ResNet-50"):
        load_resnet50_weights(resnet50_features(0), tmp_path / "bad.pth")


@pytest.mark.gpu
def test_indexer_engine_route_end_to_end_on_gpu(tmp_path, monkeypatch):
    """indexer.main (DNN branch, backend/indexer.py:51-59) -> paths file -> engine.load ->
    POST /similar_images, with a broken image in the data folder: ids still map to the right files."""
    pytest.importorskip("flask")
    import torch

    assert torch.cuda.is_available()
    from PIL import Image

    from image_search_engine_amd import descriptors as ds
    from image_search_engine_amd import engine, indexer, utils

    rng = np.random.default_rng(2)
    data = tmp_path / "data"
    paths = _write_images(data, 40, rng, size=64)
    (data / "img_0005.png").write_bytes(b"broken")       # skipped at index time
    models = tmp_path / "models"
    # the class, not the modules' instances: patching an instance would leave an attribute behind that
    # shadows the class for every later test
    monkeypatch.setattr(Config, "DATA_FOLDER_PATH", data)
    monkeypatch.setattr(Config, "DNN_INDEX_PATH", models / "resnet50_dnn_index.faiss")
    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", models / "absent.joblib")
    index = indexer.main()
    assert index.ntotal == 39
    rec = json.loads(engine.paths_file_for(models / "resnet50_dnn_index.faiss").read_text())
    listed = rec["paths"]
    assert rec["ntotal"] == 39 and rec["index_crc32"] == engine.file_crc32(models / "resnet50_dnn_index.faiss")
    assert len(listed) == 39 and str(data / "img_0005.png") not in listed
    saved = (engine.index, engine.images_paths, engine.descriptor)
    try:
        engine.load(desc=ds.CNNDescriptor())
        assert engine.index.ntotal == 39 and [str(p) for p in engine.images_paths] == listed
        client = engine.create_app().test_client()
        for probe in (3, 6, 39):                          # ids after the skipped image would be off by one with a fresh glob
            rgb = np.asarray(Image.open(paths[probe]).convert("RGB"))
            resp = client.post("/similar_images", data={"image": (io.BytesIO(_png_bytes(rgb)), "q.png")},
                               content_type="multipart/form-data")
            assert resp.status_code == 200
            pred = json.loads(resp.data)["prediction"]
            assert len(pred) == 20 and pred[0][2] == str(paths[probe]) and pred[0][0] <= 1e-3
            assert [p[0] for p in pred] == sorted(p[0] for p in pred)
    finally:
        engine.index, engine.images_paths, engine.descriptor = saved
    # a rebuild whose descriptions carry no paths (the cached-descriptions short circuit, quirk 5.9-5) must not
    # leave the old list beside the new index
    import joblib

    joblib.dump([np.full((1, 2048), i, np.float32) for i in range(3)], models / "cached.joblib")
    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", models / "cached.joblib")
    assert indexer.main().ntotal == 3
    assert not engine.paths_file_for(models / "resnet50_dnn_index.faiss").exists()


def test_process_pool_decoder_gives_the_same_descriptions(tmp_path, monkeypatch):
    """DECODE_PROCESSES > 0: spawned decode workers (no torch, no HIP in them) feed the same batches in the
    same order, skip the same broken files and leave the same described_paths as the thread pool -- with
    the pixels handed over in /dev/shm slots (the default), with slots too small for some of the images
    (those come back through the pipe) and without the ring; the ring file is gone after close()."""
    import os

    from image_search_engine_amd import descriptors as ds

    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", tmp_path / "absent.joblib")
    rng = np.random.default_rng(3)
    paths = _write_images(tmp_path / "data", 23, rng, size=24)
    paths += _write_images(tmp_path / "data2", 6, rng, size=40)      # 4800 B each: larger than the small slots below
    paths[7].write_bytes(b"broken")
    arr = np.array(paths).reshape(-1, 1)
    outs = {}
    for name, procs, slot_bytes in (("threads", 0, 3 << 20), ("ring", 3, 3 << 20), ("small slots", 3, 2048),
                                    ("no ring", 3, 0)):
        monkeypatch.setattr(Config, "DECODE_PROCESSES", procs)
        monkeypatch.setattr(Config, "DECODE_SLOT_BYTES", slot_bytes)
        describer = ds.Describer({"conv_features": _MeanColourDescriptor()}, batch_size=5)
        out = ds.describe_dataset(describer, arr)
        outs[name] = (np.concatenate([np.asarray(o) for o in out]), list(describer.described_paths))
        ring = getattr(describer, "_ring", None)
        assert (ring is not None) == (procs > 0 and slot_bytes > 0)
        ring_path = ring.path if ring is not None else None
        if ring is not None:
            assert os.path.exists(ring_path) and ring.slot_bytes == slot_bytes and ring.nslots >= 5 + 12
        describer.close()
        assert ring_path is None or not os.path.exists(ring_path)
    assert len(outs["threads"][1]) == 28
    for name in ("ring", "small slots", "no ring"):
        assert outs[name][1] == outs["threads"][1], name
        assert np.array_equal(outs[name][0], outs["threads"][0]), name


def test_paths_file_names_the_index_build_it_belongs_to(tmp_path):
    """a paths list left beside ANOTHER build of the index (same length, other rows) must not be
    used: the record carries the index file's row count and checksum, engine.read_paths_file checks both; the
    first format (a bare list) is still read, checked for its length only."""
    from image_search_engine_amd import engine

    idx = tmp_path / "x.faiss"
    idx.write_bytes(b"index build one")
    pf = engine.paths_file_for(idx)
    pf.write_text(json.dumps({"ntotal": 2, "index_crc32": engine.file_crc32(idx), "paths": ["a.png", "b.png"]}))
    assert [str(p) for p in engine.read_paths_file(idx, 2)] == ["a.png", "b.png"]
    assert engine.read_paths_file(idx, 3) is None                      # another row count
    idx.write_bytes(b"index build two")                                # same length list, another index file
    assert engine.read_paths_file(idx, 2) is None
    pf.write_text(json.dumps(["a.png", "b.png"]))                      # first format
    assert [str(p) for p in engine.read_paths_file(idx, 2)] == ["a.png", "b.png"]
    assert engine.read_paths_file(idx, 5) is None
    pf.unlink()
    assert engine.read_paths_file(idx, 2) is None


def test_runs_of_broken_files_never_hand_out_a_live_ring_slot(tmp_path, monkeypatch):
    """ring slots come from a free list and go back to it only when their image is skipped or its
    batch has been collected.  Runs of broken files (3 and 8 in a row) in front of a slow describe-only
    descriptor -- which reads the ring views one flush late -- used to let a decode worker overwrite pixels
    that were still pending (2 of 114 rows wrong)."""
    import time

    from image_search_engine_amd import descriptors as ds

    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", tmp_path / "absent.joblib")
    rng = np.random.default_rng(11)
    paths = _write_images(tmp_path / "data", 120, rng, size=20)
    for i in (9, 10, 11, 40, 41, 42, 43, 44, 45, 46, 47, 90):
        paths[i].write_bytes(b"broken")
    arr = np.array(paths).reshape(-1, 1)

    class _Slow(_MeanColourDescriptor):
        def describe(self, image):
            time.sleep(0.002)
            return super().describe(image)

    outs = {}
    for name, procs in (("threads", 0), ("ring", 3)):
        monkeypatch.setattr(Config, "DECODE_PROCESSES", procs)
        describer = ds.Describer({"conv_features": _Slow()}, batch_size=5)
        out = ds.describe_dataset(describer, arr)
        outs[name] = (np.concatenate([np.asarray(o) for o in out]), list(describer.described_paths))
        describer.close()
    assert len(outs["threads"][1]) == 108
    assert outs["ring"][1] == outs["threads"][1]
    assert np.array_equal(outs["ring"][0], outs["threads"][0])


@pytest.mark.parametrize("procs", [0, 2])
def test_describe_dataset_with_several_jobs_keeps_every_row_in_order(tmp_path, monkeypatch, procs):
    """describe_dataset runs describer.describe on ONE Describer from N_JOBS threads
    (backend/descriptors.py:125-129); the batch a call has launched but not collected is that call's own
    (it used to sit on the Describer, where the threads stole each other's: 190 of 200 rows came back)."""
    import time

    from image_search_engine_amd import descriptors as ds

    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", tmp_path / "absent.joblib")
    monkeypatch.setattr(Config, "N_JOBS", 4)
    monkeypatch.setattr(Config, "DECODE_PROCESSES", procs)
    rng = np.random.default_rng(12)
    paths = _write_images(tmp_path / "data", 200, rng, size=12)
    paths[17].write_bytes(b"broken")

    class _AsyncSlow(_MeanColourDescriptor):
        """describe_batch_async whose result is ready only a little later, like a device batch"""

        def describe_batch_async(self, images):
            import torch

            feats = torch.stack([_MeanColourDescriptor.describe(self, im) for im in images])

            class _H:
                def result(self_h):
                    time.sleep(0.002)
                    return feats

            return _H()

    describer = ds.Describer({"conv_features": _AsyncSlow()}, batch_size=7)
    out = ds.describe_dataset(describer, np.array(paths).reshape(-1, 1))
    describer.close()
    kept = [p for i, p in enumerate(paths) if i != 17]
    assert len(out) == 199 and describer.described_paths == kept
    from PIL import Image

    want = np.stack([np.tile(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1].reshape(-1, 3).mean(0), 4)
                     for p in kept]).astype(np.float32)
    assert np.array_equal(np.concatenate([np.asarray(o) for o in out]), want)


def test_describer_collects_async_batches_in_order(tmp_path, monkeypatch):
    """A descriptor with describe_batch_async has batch i + 1 launched before batch i is collected; the
    lists still come out in input order, a batch whose launch or result fails is redone image by image,
    and the last batch is collected at the end."""
    from image_search_engine_amd import descriptors as ds

    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", tmp_path / "absent.joblib")
    monkeypatch.setattr(Config, "DECODE_PROCESSES", 0)
    rng = np.random.default_rng(4)
    paths = _write_images(tmp_path / "data", 22, rng, size=16)
    events = []

    class _Async(_MeanColourDescriptor):
        def __init__(self):
            super().__init__()
            self.batches = 0

        def describe_batch_async(self, images):
            i = self.batches
            self.batches += 1
            events.append(("launch", i))
            if i == 1:
                raise RuntimeError("launch of batch 1 fails")
            feats = np.stack([_MeanColourDescriptor.describe(self, im).numpy() for im in images])

            class _H:
                def result(_self):
                    events.append(("result", i))
                    if i == 2:
                        raise RuntimeError("result of batch 2 fails")
                    return feats

            return _H()

    desc = _Async()
    describer = ds.Describer({"conv_features": desc}, batch_size=5)
    out = describer.describe(np.array(paths))
    got = np.concatenate([np.asarray(o) for o in out["conv_features"]])
    from PIL import Image

    want = np.stack([_MeanColourDescriptor().describe(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]).numpy()
                     for p in paths])
    assert np.array_equal(got, want) and out.paths["conv_features"] == paths
    # batch i is collected after batch i + 1 has been launched; the failed launch has no result() call
    assert events == [("launch", 0), ("launch", 1), ("result", 0), ("launch", 2), ("launch", 3), ("result", 2),
                      ("launch", 4), ("result", 3), ("result", 4)]


@pytest.mark.gpu
def test_process_pool_feed_on_the_gpu_equals_the_thread_pool(tmp_path, monkeypatch):
    """The whole feed on the MI355X -- decode processes, /dev/shm pixel slots, one pinned upload per batch,
    batch i + 1 launched before batch i is collected -- describes mixed-size files exactly as the plain
    thread-pool, collect-at-once configuration does: same rows, same order, same skipped files."""
    import torch

    from image_search_engine_amd import descriptors as ds

    assert torch.cuda.is_available()
    monkeypatch.setattr(Config, "BOVW_CORNER_DESCRIPTIONS_PATH", tmp_path / "absent.joblib")
    rng = np.random.default_rng(9)
    paths = []
    for j, size in enumerate((48, 64, 48, 80)):
        paths += _write_images(tmp_path / f"d{j}", 11, rng, size=size)
    paths[5].write_bytes(b"broken")
    arr = np.array(paths).reshape(-1, 1)
    desc = ds.CNNDescriptor()
    outs = {}
    for name, procs, asyn in (("threads, collect at once", 0, False), ("processes, async", 3, True),
                              ("threads, async", 0, True)):
        monkeypatch.setattr(Config, "DECODE_PROCESSES", procs)
        monkeypatch.setattr(Config, "DESCRIBE_ASYNC", asyn)
        describer = ds.Describer({"conv_features": desc}, batch_size=8)
        out = ds.describe_dataset(describer, arr)
        outs[name] = (np.concatenate([np.asarray(o) for o in out]), list(describer.described_paths))
        describer.close()
    ref = outs["threads, collect at once"]
    assert ref[0].shape == (43, 2048) and len(ref[1]) == 43
    for name in ("processes, async", "threads, async"):
        assert outs[name][1] == ref[1], name
        # same batches through the same network; the convolution library's kernels are not bitwise repeatable
        assert np.allclose(outs[name][0], ref[0], rtol=1e-4, atol=1e-4 * float(np.abs(ref[0]).max())), name
