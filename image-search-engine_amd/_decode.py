"""Image decode worker for the process pool of ``Describer``.  Kept free of torch and
of the HIP library on purpose: spawned workers import only this module, PIL and numpy."""
from __future__ import annotations

import numpy as np


def read_image_bgr(path):
    """BGR uint8 HxWx3 like cv2.imread(IMREAD_COLOR) (backend/descriptors.py:64-68), decoded with PIL.
    Returns (path, array, None) or (path, None, error text): exceptions do not cross the process
    boundary as objects the parent would have to unpickle."""
    from PIL import Image

    try:
        with Image.open(str(path)) as im:
            rgb = np.asarray(im.convert("RGB"))
        return path, np.ascontiguousarray(rgb[:, :, ::-1]).astype(np.uint8), None
    except Exception as e:  # the caller prints and skips, as the reference does
        return path, None, f"Problem opening image ({e.__class__.__name__})"
