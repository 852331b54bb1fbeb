"""GPU image preprocessing equal to ``get_image_tensor`` (src/full_model/generate_reports_for_images.py:129-147;
SURVEY.md 8(f) rank 4): LongestMaxSize(512, INTER_AREA; images smaller than 512 px are enlarged) -> centred zero PadIfNeeded(512, 512) ->
Normalize(0.471, 0.302) -> [1, 1, 512, 512] float32, on the HIP kernel ``rgrg_preprocess_u8_f32``.  File decoding
stays on the host (cv2 if installed, else PIL)."""
from __future__ import annotations

import torch

from . import _hip
from .constants import IMAGE_INPUT_SIZE

MEAN, STD = 0.471, 0.302  # generate_reports_for_images.py:29-30


def preprocess_image(image, device="cuda") -> torch.Tensor:
    """uint8 gray image [h, w] (numpy array or tensor, any device) -> float32 [1, 1, 512, 512] on ``device``."""
    img = torch.as_tensor(image)
    if img.dim() != 2 or img.dtype != torch.uint8:
        raise ValueError("preprocess_image expects an 8-bit single-channel image [h, w] (cv2.IMREAD_UNCHANGED of a gray file)")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _hip.RgrgHipError("rgrg_amd preprocesses images on the AMD GPU through librgrg_hip.so (there is no CPU fallback)")
    h, w = img.shape
    scale = IMAGE_INPUT_SIZE / float(max(h, w))
    nh, nw = (h, w) if scale == 1.0 else (int(round(h * scale)), int(round(w * scale)))  # py3round, albumentations
    nh, nw = max(nh, 1), max(nw, 1)
    src = img.to(dev).contiguous()
    out = torch.empty((1, 1, IMAGE_INPUT_SIZE, IMAGE_INPUT_SIZE), dtype=torch.float32, device=dev)
    lib = _hip.load()
    with torch.cuda.device(dev):
        _hip.check(lib.rgrg_preprocess_u8_f32(src.data_ptr(), h, w, w, nh, nw, MEAN, STD, out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "rgrg_preprocess_u8_f32")
    return out


def read_gray_image(image_path: str):
    """cv2.imread(path, cv2.IMREAD_UNCHANGED) for an 8-bit gray file; PIL when cv2 is not installed."""
    try:
        import cv2
        return cv2.imread(image_path, cv2.IMREAD_UNCHANGED)
    except ImportError:
        import numpy as np
        from PIL import Image
        with Image.open(image_path) as im:
            if im.mode != "L":
                raise ValueError(f"{image_path}: expected an 8-bit gray image, got PIL mode {im.mode}")
            return np.array(im, dtype=np.uint8)
