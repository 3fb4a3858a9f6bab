"""CPU restatement of the reference's image input transform (TEST INFRASTRUCTURE - never imported by the product).

Reference: ``ImageProcessor`` (data/preprocessing.py:42-48,73-78): ``transforms.Resize((S, S))`` -> ``ToTensor()`` ->
``Normalize(mean, std)`` on a PIL RGB image, with the ImageNet statistics of preprocessing.py:22-23.  The arithmetic lives in
third-party code that is not under /root/reference:

* torchvision (requirements.txt pins only ``>=0.15.0``; NOT installed here): ``Resize`` on a PIL image is
  ``img.resize((S, S), PIL.Image.BILINEAR)``; ``ToTensor`` is HWC uint8 -> CHW float32 ``.div(255)``; ``Normalize`` is
  ``tensor.sub_(mean).div_(std)`` with float32 mean / std.  Restated from its published definition.
* Pillow (installed: 12.2.0), ``src/libImaging/Resample.c``: two-pass separable convolution (horizontal, then vertical)
  on 8-bit channels with 22-bit fixed-point coefficients, rounding half up and clipping to [0, 255] after EACH pass.
  Restated below; pinned bit-exactly against ``PIL.Image.resize`` itself in tests/test_image_cpu.py.

Plain Python / numpy loops, written for small images.
"""
from __future__ import annotations

import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2     # Resample.c: #define PRECISION_BITS (32 - 8 - 2)


def bilinear_filter(x: float) -> float:
    """Resample.c bilinear_filter (support 1.0)."""
    if x < 0.0:
        x = -x
    if x < 1.0:
        return 1.0 - x
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box (in0 = 0, in1 = in_size).
    Returns (bounds int32 [out, 2] = (first tap, tap count), coefficients int32 [out, ksize], ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = bilinear_filter((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(v: int) -> int:
    v >>= PRECISION_BITS          # arithmetic shift, as the C code's lookup index
    return 0 if v < 0 else 255 if v > 255 else v


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``PIL.Image.fromarray(img, "RGB").resize((out_w, out_h), BILINEAR)`` for img uint8 [H, W, C]."""
    H, W, C = img.shape
    bx, kx, _ = precompute_coeffs(W, out_w)
    by, ky, _ = precompute_coeffs(H, out_h)
    tmp = np.zeros((H, out_w, C), np.uint8)
    src = img.astype(np.int64)
    for y in range(H):                      # ImagingResampleHorizontal_8bpc
        for xx in range(out_w):
            xmin, xmax = bx[xx]
            for c in range(C):
                s = 1 << (PRECISION_BITS - 1)
                for x in range(xmax):
                    s += int(src[y, x + xmin, c]) * int(kx[xx, x])
                tmp[y, xx, c] = _clip8(s)
    out = np.zeros((out_h, out_w, C), np.uint8)
    t64 = tmp.astype(np.int64)
    for yy in range(out_h):                 # ImagingResampleVertical_8bpc
        ymin, ymax = by[yy]
        for xx in range(out_w):
            for c in range(C):
                s = 1 << (PRECISION_BITS - 1)
                for y in range(ymax):
                    s += int(t64[y + ymin, xx, c]) * int(ky[yy, y])
                out[yy, xx, c] = _clip8(s)
    return out


def resize_bilinear_u8_fast(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Same arithmetic as ``resize_bilinear_u8`` with the pixel loops vectorised (int64 accumulators) - for the larger
    parity cases; tests hold it equal to the loop version and to PIL."""
    H, W, C = img.shape
    bx, kx, ksx = precompute_coeffs(W, out_w)
    by, ky, ksy = precompute_coeffs(H, out_h)
    half = 1 << (PRECISION_BITS - 1)
    src = img.astype(np.int64)
    acc = np.full((H, out_w, C), half, np.int64)
    for x in range(ksx):
        idx = np.minimum(bx[:, 0] + x, W - 1)
        on = (x < bx[:, 1]).astype(np.int64)
        acc += src[:, idx, :] * (kx[:, x].astype(np.int64) * on)[None, :, None]
    tmp = np.clip(acc >> PRECISION_BITS, 0, 255)
    acc = np.full((out_h, out_w, C), half, np.int64)
    for y in range(ksy):
        idx = np.minimum(by[:, 0] + y, H - 1)
        on = (y < by[:, 1]).astype(np.int64)
        acc += tmp[idx, :, :] * (ky[:, y].astype(np.int64) * on)[:, None, None]
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def to_tensor_normalize(img_u8: np.ndarray, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    """torchvision ``ToTensor`` + ``Normalize`` (reference preprocessing.py:46-47): uint8 [H, W, 3] -> float32 [3, H, W]."""
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    return t.sub_(m).div_(s)


def process_image(img_u8: np.ndarray, image_size: int = 224, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                  fast: bool = True) -> torch.Tensor:
    """The reference's ``val_transform`` (preprocessing.py:44-48,78) on a decoded RGB image."""
    f = resize_bilinear_u8_fast if fast else resize_bilinear_u8
    return to_tensor_normalize(f(img_u8, image_size, image_size), mean, std)
