"""`save_img` with keras.preprocessing.image semantics (/root/reference/GAN_distributed.py:119-124):
the array is min-max rescaled to 0..255 (`scale=True` default) before JPEG/PNG encoding."""
from __future__ import annotations

from pathlib import Path

import numpy as np


def array_to_uint8(x: np.ndarray, scale: bool = True) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    if scale:
        x = x - x.min()
        m = x.max()
        if m != 0:
            x = x / m
        x = x * 255.0
    return np.clip(x, 0, 255).astype(np.uint8)


def save_img(path, x, scale: bool = True):
    """x: [H,W,1] or [H,W,3] or [H,W]."""
    from PIL import Image
    a = array_to_uint8(x, scale)
    if a.ndim == 3 and a.shape[-1] == 1:
        a = a[..., 0]
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray(a, mode="L" if a.ndim == 2 else "RGB").save(p)
    return p
