"""MI355X-native hot path of ManuelZ/image-search-engine.

Batched CNN feature extraction (PyTorch-ROCm) + exact brute-force L2 / inner
product kNN (hand-written HIP for gfx950 behind the C ABI of
``include/ise_knn.h``), behind the reference's own Python surface:

    reference module (backend/)   here
    ---------------------------   ---------------------------------------
    faiss (third party)           faiss_compat  IndexFlatL2/IP, normalize_L2,
                                                write_index/read_index, Kmeans
    utils.create_search_index     utils.create_search_index
    descriptors.CNNDescriptor     descriptors.CNNDescriptor (+ batched)
    descriptors.describe_dataset  descriptors.describe_dataset
    engine.run_image_query        engine.run_image_query
    indexer.main (DNN branch)     indexer.main
    kmeans_faiss.FaissKMeans      kmeans_faiss.FaissKMeans (transform)

The directory is named ``image-search-engine_amd``; import it as
``image_search_engine_amd`` (a thin alias package at the repo root).
"""
__version__ = "0.1.0"
