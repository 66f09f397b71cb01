"""CPU tests of the host-side mirror of the reference surface (no GPU compute)."""
import struct

import numpy as np
import pytest

import image_search_engine_amd.faiss_compat as faiss
from image_search_engine_amd import utils
from image_search_engine_amd.config import Config, DnnModels, Method


def test_flat_file_layout_roundtrip():
    rng = np.random.default_rng(0)
    xb = rng.random((5, 3), dtype=np.float32)
    for metric, cc in ((faiss.METRIC_L2, b"IxF2"), (faiss.METRIC_INNER_PRODUCT, b"IxFI")):
        buf = faiss.serialize_flat(3, metric, xb)
        # [upstream-faiss] layout: fourcc, int32 d, int64 ntotal, 2 x int64 dummy, u8 trained,
        # int32 metric, u64 count, count float32
        assert buf[:4] == cc
        d, n, d1, d2, tr, mt = struct.unpack_from("<iqqqBi", buf, 4)
        assert (d, n, d1, d2, tr, mt) == (3, 5, 1 << 20, 1 << 20, 1, metric)
        (count,) = struct.unpack_from("<Q", buf, 4 + 4 + 8 * 3 + 1 + 4)
        assert count == 15 and len(buf) == 37 + 8 + 60
        d_, m_, x_ = faiss.parse_flat(buf)
        assert (d_, m_) == (3, metric) and np.array_equal(x_, xb)
    d_, m_, x_ = faiss.parse_flat(faiss.serialize_flat(7, faiss.METRIC_L2, np.zeros((0, 7), np.float32)))
    assert x_.shape == (0, 7)
    with pytest.raises(RuntimeError):
        faiss.parse_flat(b"IwPQ" + bytes(60))
    with pytest.raises(RuntimeError):
        faiss.parse_flat(faiss.serialize_flat(3, faiss.METRIC_L2, xb)[:-4])


def test_chunkIt_matches_reference_behaviour():
    # backend/utils.py:29-41 (float stepping; 'roughly equal parts')
    seq = list(range(10))
    assert utils.chunkIt(seq, 2) == [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]]
    assert utils.chunkIt(seq, 3) == [[0, 1, 2], [3, 4, 5], [6, 7, 8, 9]]
    assert sum(utils.chunkIt(list(range(7)), 4), []) == list(range(7))
    assert utils.chunkIt([], 2) == []
    arr = np.arange(6).reshape(-1, 1)  # describe_dataset passes an (N, 1) array
    parts = utils.chunkIt(arr, 2)
    assert [p.shape for p in parts] == [(3, 1), (3, 1)]


def test_config_attribute_names():
    c = Config()
    for name in ("RESIZE_SIZE", "EXTENSIONS", "NUM_IMAGES_TO_RETURN", "N_JOBS", "DATA_FOLDER_PATH",
                 "MODELS_BASE_PATH", "THUMBNAIL_SIZE", "DEVICE", "METHOD", "INDEX_TYPE", "DNN_MODEL",
                 "DNN_INDEX_PATH", "BOVW_CORNER_DESCRIPTIONS_PATH", "NUM_CLUSTERS", "LOGGING_FORMAT"):
        assert hasattr(c, name), name
    assert c.RESIZE_SIZE == 224 and c.NUM_IMAGES_TO_RETURN == 20 and c.INDEX_TYPE == "l2"
    assert c.DNN_MODEL == DnnModels.RESNET and c.METHOD == Method.DNN
    assert {m.name for m in Method} == {"BOVW", "DNN", "DHASH"}


def test_create_search_index_rejects_out_of_scope_types():
    with pytest.raises(NotImplementedError):
        utils.create_search_index(np.zeros((4, 16), np.float32), "cell-probe")
    with pytest.raises(ValueError):
        utils.create_search_index(np.zeros((4, 16), np.float32), "hamming")
    with pytest.raises(NotImplementedError):
        faiss.IndexIVFPQ(None, 16, 8, 16, 8)


def test_shard_bounds_cover_rows_exactly():
    from image_search_engine_amd.sharded import ShardedIndexFlat as S

    for n in (0, 1, 7, 1000, 1_000_000):
        for g in (1, 2, 3, 8):
            b = [S.shard_bounds(n, g, r) for r in range(g)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(g - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_bovw_module_surface_without_gpu():
    """The histogram half of BOVW mirrors the reference's names; with no index it refuses
    instead of falling back to a CPU path."""
    from image_search_engine_amd import bag_of_visual_words as bovw
    from image_search_engine_amd.kmeans_faiss import FaissKMeans

    assert callable(bovw.create_visual_word_histogram) and callable(bovw.run_clustering)
    b = bovw.BOVW(describer=None, n_clusters=8)
    assert b.n_clusters == 8 and b.clusterer is None and b.descriptions is None
    with pytest.raises(RuntimeError):
        bovw.create_visual_word_histogram([np.zeros((3, 4), np.float32)], FaissKMeans(8), 8)
