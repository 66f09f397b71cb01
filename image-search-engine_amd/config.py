"""Configuration constants under the reference's attribute names (backend/config.py:8-109), so the
mirrored modules read ``config.X`` exactly as the reference's do.

Differences, all deliberate:
  * the reference's ``DATA_FOLDER_PATH = FILL THIS PATH`` (backend/config.py:46) is a SyntaxError;
    here the data and model folders come from ``$ISE_DATA_FOLDER`` / ``$ISE_MODELS_FOLDER``
    (defaults ``data`` / ``models``);
  * ``METHOD`` defaults to DNN, the scoped hot path (the reference defaults to BOVW,
    backend/config.py:61);
  * ``DNN_BATCH_SIZE``, ``DECODE_WORKERS`` and ``DECODE_PROCESSES`` are new knobs of the batched extractor;
  * knobs of out-of-scope subsystems (BoVW grid search, cluster scoring) are not carried over.
"""
import logging
import os
from dataclasses import dataclass
from enum import Enum
from pathlib import Path

# same member names and values as backend/config.py:8-16
Method = Enum("Method", ["BOVW", "DNN", "DHASH"])
DnnModels = Enum("DnnModels", ["RESNET", "BiT"])


def _folder(env: str, default: str) -> Path:
    return Path(os.environ.get(env, default))


_MODELS = _folder("ISE_MODELS_FOLDER", "models")


def _model_file(name: str) -> Path:
    return _MODELS / name


@dataclass
class Config:
    # Un-annotated class attributes are plain constants, not dataclass fields -- as in the reference.

    # ---- general
    LOGGING_LEVEL = logging.INFO
    LOGGING_FORMAT = "%(levelname)-5s: " + "@%(funcName)-25s | %(message)s"
    RESIZE_SIZE = 224                       # side of the square the CNN sees
    EXTENSIONS = tuple("*." + ext for ext in ("jpg", "jpeg", "png"))
    NUM_IMAGES_TO_RETURN = 20               # k of the served query
    N_JOBS = 1                              # joblib threads of describe_dataset
    THUMBNAIL_SIZE = 256
    DEVICE = "cuda"
    DATA_FOLDER_PATH = _folder("ISE_DATA_FOLDER", "data")
    MODELS_BASE_PATH = _MODELS

    # ---- which pipeline, which index
    METHOD = Method.DNN
    INDEX_TYPE = "l2"                       # "cosine" | "l2"  ("cell-probe" is out of scope)

    # ---- DNN path
    DNN_MODEL = DnnModels.RESNET
    DNN_INDEX_PATH = _model_file("resnet50_dnn_index.faiss")
    DNN_BATCH_SIZE = 64                     # images per forward pass (the reference runs batch 1)
    DECODE_WORKERS = 8                      # threads decoding images ahead of the GPU batches
    DECODE_PROCESSES = 0                    # > 0: decode in that many spawned processes instead (past ~2 k images/s)
    CNN_GRAPHS = True                       # batches of 1, 2, 4 ... 32 images replay a captured HIP graph of the network
    DESCRIBE_COMBINE_MAX = 32               # concurrent describe() calls share one forward pass of up to this many images (0: never)
    DESCRIBE_ASYNC = True                   # launch batch i + 1 on the device before collecting batch i
    DECODE_SLOT_BYTES = 3 << 20             # decode processes hand pixels over in /dev/shm slots of this size (0: pickle them)

    # ---- BoVW artefacts the hot path still consults
    BOVW_CORNER_DESCRIPTIONS_PATH = _model_file("bovw_corner_descriptions.joblib")  # quirk 5.9-5
    BOVW_KMEANS_INDEX_PATH = _model_file("bovw_kmeans_index.faiss")
    NUM_CLUSTERS = 200
