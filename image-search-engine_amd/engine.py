"""Query side with the reference's names (backend/engine.py:46-65): ``run_image_query``
over module-level ``index`` / ``images_paths`` / ``descriptor``, as the reference binds
them (backend/engine.py:28-35,110-135).  ``load()`` does what the reference's
``__main__`` block does for METHOD=DNN; the Flask route is a thin optional wrapper
(serving plumbing is a "next" row)."""
from __future__ import annotations

import json
import time

import numpy as np
import torch

from . import faiss_compat as faiss
from .config import Config, Method
from .utils import get_image, get_images_paths

config = Config()
index = None
images_paths = None
descriptor = None


def paths_file_for(index_path):
    """The row-id -> image-path list that ``indexer.main`` writes beside the index file."""
    from pathlib import Path

    p = Path(str(index_path))
    return p.with_name(p.name + ".paths.json")


def file_crc32(path) -> int:
    import zlib

    crc = 0
    with open(str(path), "rb") as f:
        while True:
            blk = f.read(1 << 22)
            if not blk:
                return crc
            crc = zlib.crc32(blk, crc)


def read_paths_file(index_path, ntotal):
    """The paths list written beside ``index_path``, or None when there is none or it belongs to another
    build of the index (row count or checksum differ; a bare list -- the first format -- is only checked
    for its length)."""
    from pathlib import Path

    pf = paths_file_for(index_path)
    if not pf.exists():
        return None
    with open(pf) as f:
        rec = json.load(f)
    if isinstance(rec, list):
        rec = {"paths": rec}
    paths = rec.get("paths", [])
    if len(paths) != ntotal or rec.get("ntotal", ntotal) != ntotal:
        print(f"WARNING: {pf} lists {len(paths)} paths for an index of {ntotal} rows: ignored")
        return None
    if "index_crc32" in rec and rec["index_crc32"] != file_crc32(index_path):
        print(f"WARNING: {pf} was written for another build of {index_path} (checksum differs): ignored")
        return None
    return [Path(p) for p in paths]


def load(index_path=None, paths=None, desc=None):
    """Bind the module globals (backend/engine.py:110-117 for METHOD=DNN).

    The reference maps a result id to ``images_paths[i]`` from a fresh ``rglob`` at start-up
    (backend/engine.py:61,112), so an image skipped at index time -- or a file added since --
    shifts every later id.  ``indexer.main`` here persists the paths of
    the rows it actually indexed beside the index; that list is used when present, the reference's
    glob otherwise."""
    global index, images_paths, descriptor
    index_path = index_path or config.DNN_INDEX_PATH
    index = faiss.read_index(str(index_path))
    listed = read_paths_file(index_path, index.ntotal) if paths is None else None
    if paths is not None:
        images_paths = paths
    elif listed is not None:
        images_paths = listed
    else:
        images_paths = get_images_paths()
    if len(images_paths) != index.ntotal:
        print(f"WARNING: {len(images_paths)} image paths for {index.ntotal} index rows: ids may not match paths")
    print(f"There are {index.ntotal} images in the index.")
    if desc is not None:
        descriptor = desc
        if hasattr(desc, "warm_up") and getattr(desc, "device", None) is not None and desc.device.type == "cuda":
            desc.warm_up()   # the batch shapes combined request threads will produce (descriptors.CNNDescriptor.describe)
    return index


def run_image_query(image_features, n_images, normalize=False):
    """backend/engine.py:46-65: tensor -> (1, d) float32, optional normalize_L2,
    index.search, ravel, id -> path -> thumbnail."""
    if isinstance(image_features, torch.Tensor):
        image_features = image_features.detach().cpu().numpy().reshape(1, -1)
    image_features = np.ascontiguousarray(image_features, dtype=np.float32)
    if normalize:
        faiss.normalize_L2(image_features)
    distances, indices = index.search(image_features, n_images)
    distances = distances.ravel().tolist()
    indices = indices.ravel().tolist()
    predictions = []
    for dist, i in zip(distances, indices):
        image_path = images_paths[i]
        image = get_image(image_path)
        predictions.append((dist, image, str(image_path)))
    return predictions


def run_image_queries(features, n_images, normalize=False):
    """Batched form (new capability): (nq, d) features -> list of per-query predictions
    with one index.search call."""
    if isinstance(features, torch.Tensor):
        features = features.detach().cpu().numpy()
    features = np.ascontiguousarray(features, dtype=np.float32).reshape(len(features), -1)
    if normalize:
        faiss.normalize_L2(features)
    D, I = index.search(features, n_images)
    return [[(float(d), get_image(images_paths[i]), str(images_paths[i])) for d, i in zip(dr, ir)]
            for dr, ir in zip(D.tolist(), I.tolist())]


def create_app():
    """POST /similar_images, multipart field ``image`` -> {"prediction": [[dist, b64, path], ...]}
    (backend/engine.py:68-107; frontend/src/App.js:14-21); POST /similar_images_batch for several files."""
    import io

    from flask import Flask, Response, request
    from PIL import Image

    app = Flask(__name__)

    @app.after_request
    def _cors(resp):  # flask_cors is not in this image; the reference enables CORS globally
        resp.headers["Access-Control-Allow-Origin"] = "*"
        return resp

    @app.route("/similar_images", methods=["POST"])
    def predict():
        if not request.files or "image" not in request.files:
            return Response("No file uploaded", status=400)
        try:
            rgb = np.asarray(Image.open(io.BytesIO(request.files["image"].read())).convert("RGB"))
        except Exception:
            return Response("Not an image", status=400)
        image = np.ascontiguousarray(rgb[:, :, ::-1])  # BGR like cv2.imdecode (backend/engine.py:42)
        start = time.time()
        if config.METHOD != Method.DNN:
            return Response("only METHOD=DNN is served", status=501)
        image_features = descriptor.describe(image)
        predictions = run_image_query(image_features, config.NUM_IMAGES_TO_RETURN)
        print(f"Took {time.time() - start:.2f} seconds.")
        return Response(response=json.dumps({"prediction": predictions}), status=200, mimetype="application/json")

    @app.route("/similar_images_batch", methods=["POST"])
    def predict_batch():
        """Several uploads in one request: the multipart field ``image`` repeated ->
        {"predictions": [prediction list of upload 0, of upload 1, ...]}, each list exactly what
        /similar_images returns for that file -- one batched forward pass, one index.search."""
        files = request.files.getlist("image") if request.files else []
        if not files:
            return Response("No file uploaded", status=400)
        if config.METHOD != Method.DNN:
            return Response("only METHOD=DNN is served", status=501)
        images = []
        for f in files:
            try:
                rgb = np.asarray(Image.open(io.BytesIO(f.read())).convert("RGB"))
            except Exception:
                return Response(f"Not an image: {f.filename}", status=400)
            images.append(np.ascontiguousarray(rgb[:, :, ::-1]))
        if hasattr(descriptor, "describe_batch"):
            feats = descriptor.describe_batch(images)
        else:
            feats = [descriptor.describe(im) for im in images]
        feats = np.stack([np.asarray(f, dtype=np.float32).reshape(-1) for f in feats])
        predictions = run_image_queries(feats, config.NUM_IMAGES_TO_RETURN)
        return Response(response=json.dumps({"predictions": predictions}), status=200, mimetype="application/json")

    return app
