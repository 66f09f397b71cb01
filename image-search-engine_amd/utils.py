"""Index factory and small helpers with the reference's names
(backend/utils.py:29-41, 222-232, 293-330)."""
from __future__ import annotations

from . import faiss_compat as faiss
from .config import Config

config = Config()


def chunkIt(seq, num):
    """Divide a sequence into roughly equal parts (backend/utils.py:29-41)."""
    avg = len(seq) / float(num)
    out = []
    last = 0.0
    while last < len(seq):
        out.append(seq[int(last): int(last + avg)])
        last += avg
    return out


def get_images_paths():
    """rglob over EXTENSIONS under DATA_FOLDER_PATH (backend/utils.py:222-232)."""
    paths = []
    for ext in config.EXTENSIONS:
        paths.extend(config.DATA_FOLDER_PATH.rglob(ext))
    return paths


def get_image(image_path):
    """Thumbnail of an image as base64 JPEG (PNG if JPEG cannot encode it), None if the
    file is missing (backend/utils.py:44-62).  Display only."""
    import base64
    import io

    from PIL import Image

    size = config.THUMBNAIL_SIZE, config.THUMBNAIL_SIZE
    try:
        img = Image.open(image_path, mode="r")
    except FileNotFoundError:
        return None
    img.thumbnail(size, Image.LANCZOS)
    buf = io.BytesIO()
    try:
        img.save(buf, format="JPEG")
    except OSError:
        img.save(buf, format="PNG")
    return base64.encodebytes(buf.getvalue()).decode("ascii")


def create_search_index(data_array, index_type="cosine"):
    """backend/utils.py:293-330.  'cosine' -> IndexFlatIP over rows normalised IN
    PLACE in the caller's array (quirk 5.9-6); 'l2' -> IndexFlatL2; then add.
    'cell-probe' (IndexIVFPQ) is approximate and outside the scoped path."""
    num_features = data_array.shape[1]
    if index_type == "cosine":
        index = faiss.IndexFlatIP(num_features)
        faiss.normalize_L2(data_array)
    elif index_type == "l2":
        index = faiss.IndexFlatL2(num_features)
    elif index_type == "cell-probe":
        raise NotImplementedError("'cell-probe' (IndexIVFPQ) is outside the exact brute-force hot path")
    else:
        raise ValueError(f"unknown index_type {index_type!r}")
    index.add(data_array)
    print(f"There are {index.ntotal} images in the search index.")
    return index
