"""CPU restatement of ``get_image_tensor`` (src/full_model/generate_reports_for_images.py:129-147).
TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference delegates the arithmetic to third-party code that is ABSENT from this image, so this part is
**parity unpinned** (hand KATs only, ``tests/test_oracle_kats.py``):

  * albumentations 1.x ``LongestMaxSize(512, cv2.INTER_AREA)``: ``scale = 512 / max(h, w)``, new size =
    ``py3round(dim * scale)`` per side, ``cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA)``;
  * OpenCV 4.x ``resize`` INTER_AREA for 8-bit single-channel DOWN-scaling (imgproc/src/resize.cpp):
      - both scales integer (``is_area_fast``): 2x2 -> ``(a + b + c + d + 2) >> 2``; otherwise the integer sum of the
        area times ``float(1 / area)``, ``saturate_cast<uchar>`` (round half to even);
      - otherwise ``computeResizeAreaTab`` (fractional coverage weights, double arithmetic, float weights) and
        ``ResizeArea_Invoker`` (float accumulation: x first in table order, then rows in order), ``saturate_cast``;
  * ``PadIfNeeded(512, 512, border_mode=cv2.BORDER_CONSTANT)``: centred, value 0, top/left = ``int((512 - n) / 2)``;
  * ``Normalize(mean=0.471, std=0.302)``, ``max_pixel_value=255``: float32 ``(x - mean*255) * (1 / (std*255))``;
  * ``ToTensorV2`` + ``unsqueeze(0)`` -> float32 [1, 1, 512, 512].
"""
from __future__ import annotations

import math

import numpy as np

IMAGE_INPUT_SIZE = 512
MEAN, STD = 0.471, 0.302


def py3round(x: float) -> int:
    return int(round(x))  # Python 3 round: half to even (albumentations.augmentations.geometric.functional.py3round)


def resize_area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: list of (dst index, src index, float32 weight) in OpenCV's order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA) for uint8 [h, w], down-scaling."""
    h, w = img.shape
    assert img.dtype == np.uint8 and new_h <= h and new_w <= w
    if (new_h, new_w) == (h, w):
        return img.copy()
    sx, sy = w / new_w, h / new_h
    isx, isy = int(round(sx)), int(round(sy))
    if abs(sx - isx) < np.finfo(np.float64).eps and abs(sy - isy) < np.finfo(np.float64).eps:
        blocks = img[: new_h * isy, : new_w * isx].reshape(new_h, isy, new_w, isx).astype(np.int64).sum(axis=(1, 3))
        if isx == 2 and isy == 2:
            return ((blocks + 2) >> 2).astype(np.uint8)
        v = blocks.astype(np.float32) * np.float32(1.0 / (isx * isy))
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)
    xtab, ytab = resize_area_tab(w, new_w, sx), resize_area_tab(h, new_h, sy)
    src = img.astype(np.float32)
    buf = np.zeros((h, new_w), dtype=np.float32)  # x pass: buf[dx] += S[sx] * alpha in table order
    for dx, s, a in xtab:
        buf[:, dx] += src[:, s] * a
    out = np.zeros((new_h, new_w), dtype=np.float32)
    first = np.ones((new_h,), dtype=bool)
    for dy, s, b in ytab:  # y pass: sum = beta * buf (first row of the group), then sum += beta * buf
        if first[dy]:
            out[dy] = b * buf[s]
            first[dy] = False
        else:
            out[dy] += b * buf[s]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def get_image_tensor_from_array(image: np.ndarray) -> np.ndarray:
    """uint8 [h, w] (what cv2.imread(..., IMREAD_UNCHANGED) returns for an 8-bit gray file) -> float32 [1,1,512,512]."""
    h, w = image.shape
    scale = IMAGE_INPUT_SIZE / float(max(h, w))
    if scale != 1.0:
        nh, nw = py3round(h * scale), py3round(w * scale)
        if scale > 1.0:
            raise NotImplementedError("INTER_AREA up-scaling (images smaller than 512) is not restated")
        image = resize_area_u8(image, nh, nw)
    h, w = image.shape
    top, left = int((IMAGE_INPUT_SIZE - h) / 2.0), int((IMAGE_INPUT_SIZE - w) / 2.0)
    padded = np.zeros((IMAGE_INPUT_SIZE, IMAGE_INPUT_SIZE), dtype=np.uint8)
    padded[top:top + h, left:left + w] = image
    mean = np.float32(MEAN) * np.float32(255.0)
    denom = np.float32(1.0) / (np.float32(STD) * np.float32(255.0))
    x = padded.astype(np.float32)
    x -= mean
    x *= denom
    return x[None, None]
