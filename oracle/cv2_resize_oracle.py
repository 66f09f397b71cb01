"""TEST INFRASTRUCTURE -- CPU restatement of the reference's image preprocessing, for bounding the product's
float preprocessing against it.  Only tests/ may import this module.

What it restates: backend/descriptors.py:153-159,185 -- ``A.Compose([A.Resize(224, 224, cv2.INTER_LINEAR),
A.Normalize(), ToTensorV2()])`` applied to the BGR uint8 image ``cv2.imread`` returned.  albumentations'
``Resize`` is ``cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)`` and ``Normalize()`` is
``(img - 255 * mean) / (255 * std)`` in float32 with the ImageNet mean / std [upstream-albumentations].

PARITY UNPINNED: neither cv2 nor albumentations is installed in this image (and must not be fetched), and the
reference holds no preprocessed fixture.  ``resize_linear_u8`` restates OpenCV's published uint8 bilinear
algorithm (modules/imgproc/src/resize.cpp [upstream-opencv, 4.x], recalled, not read here):
  * source position of destination pixel dx: fx = (dx + 0.5) * (src_w / dst_w) - 0.5 (double, kept as float),
    sx = floor(fx), fx -= sx; sx < 0 -> (0, fx = 0); sx >= src_w - 1 -> (src_w - 1, fx = 0): no anti-aliasing,
    edge pixels replicated;
  * coefficients in 11-bit fixed point: (round((1 - fx) * 2048), round(fx * 2048)) as int16 (cvRound: to even);
  * horizontal pass in int32: buf = S[sx] * a0 + S[sx + 1] * a1 (values scaled by 2048);
  * vertical pass: dst = (((b0 * (buf0 >> 4)) >> 16) + ((b1 * (buf1 >> 4)) >> 16) + 2) >> 2, as uint8;
  * exact 2x downscaling in both directions is taken by the INTER_AREA path instead (2 x 2 box average,
    rounded): ``cv::resize`` switches INTER_LINEAR to INTER_AREA there.
"""
from __future__ import annotations

import numpy as np

MEAN = np.array((0.485, 0.456, 0.406), np.float32)   # albumentations A.Normalize() defaults (ImageNet)
STD = np.array((0.229, 0.224, 0.225), np.float32)
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis_tables(src: int, dst: int):
    """(index of the left / upper source sample, int16 coefficient pair) per destination position"""
    scale = 1.0 / (float(dst) / float(src))            # double, as cv::resize derives it from the sizes
    ofs = np.empty(dst, np.int64)
    coef = np.empty((dst, 2), np.int64)
    for dx in range(dst):
        fx = np.float32((dx + 0.5) * scale - 0.5)
        sx = int(np.floor(fx))
        fx = np.float32(fx - np.float32(sx))
        if sx < 0:
            sx, fx = 0, np.float32(0.0)
        if sx >= src - 1:
            sx, fx = src - 1, np.float32(0.0)
        ofs[dx] = sx
        c0 = np.float32(np.float32(1.0) - fx) * np.float32(COEF_SCALE)
        c1 = np.float32(fx) * np.float32(COEF_SCALE)
        coef[dx, 0] = int(np.rint(c0))                 # saturate_cast<short>(float) = cvRound (ties to even)
        coef[dx, 1] = int(np.rint(c1))
    return ofs, coef


def resize_linear_u8(img: np.ndarray, size: tuple[int, int]) -> np.ndarray:
    """uint8 HxWxC -> uint8 size[1] x size[0] x C as ``cv2.resize(img, size, interpolation=cv2.INTER_LINEAR)``
    is understood to compute it (see the module docstring; ``size`` = (width, height) like cv2's dsize)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    dw, dh = int(size[0]), int(size[1])
    if (w, h) == (dw, dh):
        return img.copy()
    if w == 2 * dw and h == 2 * dh:                    # the INTER_AREA fast path of an exact 2x reduction
        s = img.astype(np.int64)
        box = s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2]
        return ((box + 2) >> 2).astype(np.uint8)
    xofs, alpha = _axis_tables(w, dw)
    yofs, beta = _axis_tables(h, dh)
    s = img.astype(np.int64)
    x1 = np.minimum(xofs + 1, w - 1)                   # coefficient 0 wherever the clamp bites
    buf = s[:, xofs, :] * alpha[None, :, 0, None] + s[:, x1, :] * alpha[None, :, 1, None]     # [h][dw][c], x 2048
    y1 = np.minimum(yofs + 1, h - 1)
    b0 = beta[:, 0][:, None, None]
    b1 = beta[:, 1][:, None, None]
    out = (((b0 * (buf[yofs] >> 4)) >> 16) + ((b1 * (buf[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess(img_bgr_u8: np.ndarray, size: int = 224) -> np.ndarray:
    """The reference's whole preprocessor: resize -> normalise -> CHW float32 (channel order untouched: the
    BGR image goes into the RGB statistics, quirk 5.9-1)."""
    r = resize_linear_u8(img_bgr_u8, (size, size)).astype(np.float32)
    x = (r - MEAN * np.float32(255.0)) * (np.float32(1.0) / (STD * np.float32(255.0)))
    return np.ascontiguousarray(x.transpose(2, 0, 1))
