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
  * OpenCV 4.x ``resize`` INTER_AREA when EITHER axis is enlarged (scale = src / dst < 1 on an axis; images smaller
    than 512 px: ``LongestMaxSize`` up-scales them): "true area interpolation is only implemented for the case
    scale_x >= 1 && scale_y >= 1, in other cases it is emulated using some variant of bilinear" - the 8-bit
    fixed-point bilinear path (``HResizeLinear`` / ``VResizeLinear<uchar,int,short>``) with the AREA coordinate rule:
    ``s = floor(d * scale)``, ``f = (d + 1) - (s + 1) * inv_scale``, ``f = f <= 0 ? 0 : f - floor(f)`` (float32),
    weights ``saturate_cast<short>((1 - f, f) * 2048)``; columns with ``s + 1 >= width`` take ``S[width - 1] * 2048``;
    the second row index is clamped to ``height - 1``; vertical pass
    ``uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)``;
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


def _area_linear_coeffs(ssize: int, dsize: int):
    """Per destination index of the bilinear emulation: source index s, short weights (a0, a1) and the `edge` flag
    (s + 1 >= ssize: HResizeLinear's tail ``D[dx] = S[xofs[dx]] * ONE``), following resize.cpp's area_mode branch."""
    scale = 1.0 / (dsize / ssize)            # scale_x = 1. / inv_scale_x, double
    inv_scale = dsize / ssize
    d = np.arange(dsize, dtype=np.float64)
    s = np.floor(d * scale).astype(np.int64)
    f = ((d + 1.0) - (s + 1).astype(np.float64) * inv_scale).astype(np.float32)
    f = np.where(f <= 0, np.float32(0), f - np.floor(f)).astype(np.float32)
    edge = s + 1 >= ssize
    f = np.where(s >= ssize - 1, np.float32(0), f).astype(np.float32)
    s = np.minimum(s, ssize - 1)
    a0 = np.clip(np.rint((np.float32(1.0) - f) * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    a1 = np.clip(np.rint(f * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    return s, a0, a1, edge


def resize_area_as_linear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """cv2.resize(..., INTER_AREA) for uint8 [h, w] when an axis is enlarged: the fixed-point bilinear emulation."""
    h, w = img.shape
    src = img.astype(np.int64)
    sx, a0, a1, xedge = _area_linear_coeffs(w, new_w)
    sx1 = np.minimum(sx + 1, w - 1)
    hbuf = np.where(xedge[None, :], src[:, sx] * 2048, src[:, sx] * a0[None, :] + src[:, sx1] * a1[None, :])   # [h, new_w] int
    # the y loop of resize() applies no edge rule: weights from f as computed, second row clamped by the invoker
    scale_y, inv_y = 1.0 / (new_h / h), new_h / h
    dy = np.arange(new_h, dtype=np.float64)
    sy = np.floor(dy * scale_y).astype(np.int64)
    fy = ((dy + 1.0) - (sy + 1).astype(np.float64) * inv_y).astype(np.float32)
    fy = np.where(fy <= 0, np.float32(0), fy - np.floor(fy)).astype(np.float32)
    b0 = np.clip(np.rint((np.float32(1.0) - fy) * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    b1 = np.clip(np.rint(fy * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    r0 = np.clip(sy, 0, h - 1)
    r1 = np.clip(sy + 1, 0, h - 1)
    s0, s1 = hbuf[r0] >> 4, hbuf[r1] >> 4
    out = (((b0[:, None] * s0) >> 16) + ((b1[:, None] * s1) >> 16) + 2) >> 2
    return (out & 0xFF).astype(np.uint8)      # uchar(...) cast


def resize_area_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA) for uint8 [h, w]."""
    h, w = img.shape
    assert img.dtype == np.uint8
    if (new_h, new_w) == (h, w):
        return img.copy()
    if not (w / new_w >= 1 and h / new_h >= 1):
        return resize_area_as_linear_u8(img, new_h, new_w)
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
        nh, nw = max(py3round(h * scale), 1), max(py3round(w * scale), 1)
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
