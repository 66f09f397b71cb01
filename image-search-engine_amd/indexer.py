"""Index build, DNN branch of the reference's ``indexer.main`` (backend/indexer.py:27-59):
paths -> describe_dataset -> np.concatenate -> create_search_index -> write_index.
The BOVW and DHASH branches are outside the scoped hot path."""
from __future__ import annotations

import json
import logging

import numpy as np

from . import faiss_compat as faiss
from .config import Config
from .descriptors import CNNDescriptor, Describer, describe_dataset
from .utils import create_search_index, get_images_paths

config = Config()


def main():
    print("Starting...")
    images_paths = get_images_paths()
    images_paths = np.array(images_paths).reshape(-1, 1)
    if config.METHOD != config.METHOD.DNN:
        raise NotImplementedError("only METHOD=DNN is on the scoped hot path (backend/indexer.py:51-59)")
    descriptor = CNNDescriptor(model=config.DNN_MODEL)
    describer = Describer({"conv_features": descriptor})
    descriptions = describe_dataset(describer, images_paths)
    descriptions = np.concatenate([np.asarray(d, dtype=np.float32) for d in descriptions])
    print("Creating index with features of size ", descriptions.shape)
    index = create_search_index(descriptions, index_type=config.INDEX_TYPE)
    config.DNN_INDEX_PATH.parent.mkdir(parents=True, exist_ok=True)
    faiss.write_index(index, str(config.DNN_INDEX_PATH))
    # row id -> image path of the rows actually indexed (skipped images leave no row): the engine
    # reads this instead of re-globbing the data folder (fixes SURVEY.md quirk 5.9-4)
    from .engine import file_crc32, paths_file_for

    described = [str(p) for p in np.asarray(describer.described_paths, dtype=object).ravel().tolist()]
    paths_file = paths_file_for(config.DNN_INDEX_PATH)
    if len(described) == index.ntotal:
        # the list names the index file it belongs to (row count + checksum): engine.load ignores a list
        # that was left beside another build of the index
        with open(paths_file, "w") as f:
            json.dump({"ntotal": int(index.ntotal), "index_crc32": file_crc32(config.DNN_INDEX_PATH),
                       "paths": described}, f)
    else:  # descriptions came from somewhere that does not track paths (the cached joblib file, quirk 5.9-5)
        print(f"WARNING: {len(described)} described paths for {index.ntotal} rows: no paths file written")
        if paths_file.exists():  # an older build's list must not sit beside the new index
            paths_file.unlink()
    return index


if __name__ == "__main__":
    logging.basicConfig(format=config.LOGGING_FORMAT, level=config.LOGGING_LEVEL)
    main()
