"""Configuration constants with the reference's attribute names
(backend/config.py:8-109).  The reference's ``DATA_FOLDER_PATH = FILL THIS PATH``
(backend/config.py:46) is a SyntaxError; here it defaults to ``./data`` and can
be overridden with the ISE_DATA_FOLDER environment variable."""
import logging
import os
from dataclasses import dataclass
from enum import Enum
from pathlib import Path


class Method(Enum):
    BOVW = 1
    DNN = 2
    DHASH = 3


class DnnModels(Enum):
    RESNET = 1
    BiT = 2


@dataclass
class Config:
    # un-annotated class attributes, i.e. plain constants, as in the reference
    LOGGING_LEVEL = logging.INFO
    LOGGING_FORMAT = "%(levelname)-5s: @%(funcName)-25s | %(message)s"
    RESIZE_SIZE = 224
    EXTENSIONS = ("*.jpg", "*.jpeg", "*.png")
    NUM_IMAGES_TO_RETURN = 20
    N_JOBS = 1
    DATA_FOLDER_PATH = Path(os.environ.get("ISE_DATA_FOLDER", "data"))
    MODELS_BASE_PATH = Path(os.environ.get("ISE_MODELS_FOLDER", "models"))
    THUMBNAIL_SIZE = 256
    DEVICE = "cuda"

    # the scoped hot path is the DNN method (the reference defaults to BOVW,
    # backend/config.py:61)
    METHOD = Method.DNN
    INDEX_TYPE = "l2"  # cosine, l2 ("cell-probe" is out of scope)

    DNN_MODEL = DnnModels.RESNET
    DNN_INDEX_PATH = MODELS_BASE_PATH / "resnet50_dnn_index.faiss"
    # images per forward pass of the batched extractor (new capability; the
    # reference runs batch 1, backend/descriptors.py:185-187)
    DNN_BATCH_SIZE = 64
    # threads decoding images ahead of the GPU batches (new; the reference decodes inline)
    DECODE_WORKERS = 8

    BOVW_CORNER_DESCRIPTIONS_PATH = MODELS_BASE_PATH / "bovw_corner_descriptions.joblib"
    BOVW_KMEANS_INDEX_PATH = MODELS_BASE_PATH / "bovw_kmeans_index.faiss"
    NUM_CLUSTERS = 200
