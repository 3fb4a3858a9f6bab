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


# ------------------------------------------------------------------------------------------------------------------
# Training transform (reference data/preprocessing.py:52-70, ``augment=True``):
#   RandomResizedCrop(S, scale=(0.8, 1.0), ratio=(0.75, 1.33)) -> RandomHorizontalFlip(0.5) ->
#   ColorJitter(brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1) -> RandomRotation(5) -> ToTensor -> Normalize
# on a PIL image.  The random DRAWS are the loader's business (input.draw_train_params restates torchvision's
# ``get_params``); what is restated here is what each transform does to the pixels GIVEN its parameters:
#   * torchvision (absent; published ``transforms/_functional_pil.py``): resized_crop = ``img.crop(box).resize(size,
#     BILINEAR)``; hflip = ``transpose(FLIP_LEFT_RIGHT)``; adjust_brightness / contrast / saturation =
#     ``ImageEnhance.{Brightness, Contrast, Color}(img).enhance(f)``; adjust_hue = HSV split, uint8 wrap-around add of
#     ``uint8(hue * 255)`` to H, merge, back to RGB; rotate = ``img.rotate(angle, NEAREST, expand=False, fillcolor=0)``.
#   * Pillow 12.2.0 (installed): ``Blend.c`` (float32 ``in1 + alpha * (in2 - in1)``, truncated; clipped when alpha is
#     outside [0, 1]), ``Convert.c`` rgb2l / rgb2hsv / hsv2rgb, ``ImageEnhance.py`` (degenerate images), ``Image.rotate``
#     (matrix from rounded cos / sin) and ``Geometry.c`` affine_fixed (16.16 fixed point, nearest).
# ``train_transform_pil`` calls Pillow itself; ``train_transform`` is the numpy restatement.  tests/test_image_cpu.py holds
# them bit-identical (every op alone over exhaustive or random inputs, then whole chains).
def rgb2l(img: np.ndarray) -> np.ndarray:
    """Convert.c ``L24`` / rgb2l: uint8 [..., 3] -> uint8 [...]."""
    x = img.astype(np.int64)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(in1: np.ndarray, in2: np.ndarray, alpha: float) -> np.ndarray:
    """``Image.blend(in1, in2, alpha)`` (Blend.c): float32 multiply, float32 add, truncation; clipping only when
    extrapolating."""
    al = np.float32(alpha)
    d = (in2.astype(np.int32) - in1.astype(np.int32)).astype(np.float32)
    t = in1.astype(np.float32) + al * d
    if 0.0 <= float(al) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness(img: np.ndarray, f: float) -> np.ndarray:
    return blend_u8(np.zeros_like(img), img, f)


def contrast_mean(img: np.ndarray) -> int:
    """ImageEnhance.Contrast: ``int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5)``."""
    L = rgb2l(img)
    return int(float(int(L.astype(np.int64).sum())) / L.size + 0.5)


def adjust_contrast(img: np.ndarray, f: float) -> np.ndarray:
    return blend_u8(np.full_like(img, contrast_mean(img)), img, f)


def adjust_saturation(img: np.ndarray, f: float) -> np.ndarray:
    return blend_u8(np.repeat(rgb2l(img)[..., None], 3, -1), img, f)


def rgb2hsv(img: np.ndarray) -> np.ndarray:
    """Convert.c rgb2hsv (float32 ratios, double for the hue fold, truncation to uint8)."""
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    with np.errstate(all="ignore"):
        cr = (maxc - minc).astype(np.float32)
        s = cr / maxc.astype(np.float32)
        rc = (maxc - r).astype(np.float32) / cr
        gc = (maxc - g).astype(np.float32) / cr
        bc = (maxc - b).astype(np.float32) / cr
        h = np.where(r == maxc, (bc - gc).astype(np.float32),
                     np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc).astype(np.float32),
                              (4.0 + gc.astype(np.float64) - rc).astype(np.float32)))
        h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv2rgb(hsv: np.ndarray) -> np.ndarray:
    """Convert.c hsv2rgb (sector floor(h * 6 / 255), C ``round`` = half away from zero)."""
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    x = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(x)
    f = x - i
    fs = s / 255.0
    vf = v.astype(np.float64)
    p = np.clip(np.floor(vf * (1.0 - fs) + 0.5), 0, 255).astype(np.int32)
    q = np.clip(np.floor(vf * (1.0 - fs * f) + 0.5), 0, 255).astype(np.int32)
    t = np.clip(np.floor(vf * (1.0 - fs * (1.0 - f)) + 0.5), 0, 255).astype(np.int32)
    ii = i.astype(np.int32) % 6
    R = np.choose(ii, [v, q, p, p, t, v])
    G = np.choose(ii, [t, v, v, q, p, p])
    B = np.choose(ii, [p, p, t, v, v, q])
    g0 = s == 0
    return np.stack([np.where(g0, v, R), np.where(g0, v, G), np.where(g0, v, B)], -1).astype(np.uint8)


def hue_shift_u8(hue_factor: float) -> int:
    """torchvision adjust_hue: ``np.array(hue_factor * 255).astype(np.uint8)`` - truncation toward zero, modulo 256."""
    return int(hue_factor * 255) % 256


def adjust_hue(img: np.ndarray, hue_factor: float) -> np.ndarray:
    hsv = rgb2hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_u8(hue_factor)).astype(np.uint8)   # wraps, as np.uint8 +=
    return hsv2rgb(hsv)


def rotate_matrix(angle: float, w: int, h: int):
    """``Image.rotate(angle, expand=False, center=None)``: the affine matrix (output -> input), or None for a copy."""
    angle = angle % 360.0
    if angle == 0:
        return None
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2], m[5] = m[0] * -cx + m[1] * -cy + m[2], m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def _fix16(v: float) -> int:
    v = v * 65536.0 + 0.5                      # Geometry.c FIX = FLOOR(v * 65536.0 + 0.5)
    return int(math.floor(v)) if v < 0 else int(v)


def rotate_fixed_coeffs(angle: float, w: int, h: int):
    """Geometry.c affine_fixed's six 16.16 integers (a0, a1, a2, a3, a4, a5), or None when the rotation is a copy."""
    m = rotate_matrix(angle, w, h)
    if m is None:
        return None
    return (_fix16(m[0]), _fix16(m[1]), _fix16(m[2] + m[0] * 0.5 + m[1] * 0.5),
            _fix16(m[3]), _fix16(m[4]), _fix16(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_nearest(img: np.ndarray, angle: float) -> np.ndarray:
    """``img.rotate(angle, NEAREST, expand=False, fillcolor=0)`` (|angle| small: the affine_fixed path)."""
    h, w = img.shape[:2]
    c = rotate_fixed_coeffs(angle, w, h)
    if c is None:
        return img.copy()
    a0, a1, a2, a3, a4, a5 = c
    y, x = np.mgrid[0:h, 0:w]
    xin = (a2 + a1 * y + a0 * x) >> 16
    yin = (a5 + a4 * y + a3 * x) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


JITTER_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)   # torchvision ColorJitter fn_id 0..3


def train_augment_u8(img_u8: np.ndarray, p: dict, image_size: int = 224) -> np.ndarray:
    """The uint8 part of the training transform, numpy restatement.  ``p``: ``box`` (i, j, h, w), ``flip`` bool, ``order``
    (permutation of 0..3), ``brightness`` / ``contrast`` / ``saturation`` / ``hue`` factors, ``angle`` degrees."""
    i, j, h, w = p["box"]
    x = resize_bilinear_u8_fast(np.ascontiguousarray(img_u8[i:i + h, j:j + w]), image_size, image_size)
    if p["flip"]:
        x = np.ascontiguousarray(x[:, ::-1])
    factors = (p["brightness"], p["contrast"], p["saturation"], p["hue"])
    for fn in p["order"]:
        x = JITTER_OPS[fn](x, factors[fn])
    return rotate_nearest(x, p["angle"])


def train_augment_u8_pil(img_u8: np.ndarray, p: dict, image_size: int = 224) -> np.ndarray:
    """The same through Pillow itself, call for call as torchvision's PIL backend makes them."""
    from PIL import Image, ImageEnhance
    i, j, h, w = p["box"]
    im = Image.fromarray(np.ascontiguousarray(img_u8), "RGB").crop((j, i, j + w, i + h))
    im = im.resize((image_size, image_size), Image.BILINEAR)
    if p["flip"]:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    for fn in p["order"]:
        if fn == 0:
            im = ImageEnhance.Brightness(im).enhance(p["brightness"])
        elif fn == 1:
            im = ImageEnhance.Contrast(im).enhance(p["contrast"])
        elif fn == 2:
            im = ImageEnhance.Color(im).enhance(p["saturation"])
        else:
            hh, ss, vv = im.convert("HSV").split()
            np_h = np.array(hh, dtype=np.uint8)
            with np.errstate(over="ignore"):
                np_h += np.uint8(hue_shift_u8(p["hue"]))
            im = Image.merge("HSV", (Image.fromarray(np_h, "L"), ss, vv)).convert("RGB")
    im = im.rotate(p["angle"], Image.NEAREST, False, None, fillcolor=0)
    return np.asarray(im)


def process_image_train(img_u8: np.ndarray, p: dict, image_size: int = 224, mean=(0.485, 0.456, 0.406),
                        std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    """The reference's ``train_transform`` (preprocessing.py:52-70) on a decoded RGB image, given the random draws."""
    return to_tensor_normalize(train_augment_u8(img_u8, p, image_size), mean, std)
