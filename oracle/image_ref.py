"""CPU oracle of the input-pipeline arithmetic (TEST INFRASTRUCTURE ONLY -- never imported by the product).

The reference's train / eval transforms (/root/reference/datasets.py:121-144) are timm.create_transform / torchvision
transforms on PIL images; neither package is installed, so the arithmetic they delegate to is restated here in numpy
and PINNED AGAINST PILLOW ITSELF (present in the build container): tests/test_oracle_cpu.py compares every function
below with PIL.Image.resize / crop / transpose and PIL.ImageEnhance on random images, bit for bit.

  resize_u8        Pillow's two-pass separable resampling for 8-bit images (libImaging/Resample.c): per output sample a
                   window of `support * max(scale, 1)` input samples each side, filter weights normalised in double and
                   quantised to 22 fractional bits, an accumulator seeded with 1 << 21, result shifted and clamped to
                   uint8 -- after the HORIZONTAL pass and again after the VERTICAL pass.  bicubic (a = -0.5, support 2)
                   for the train transform (`interpolation='bicubic'`, datasets.py:131), bilinear (support 1) for eval
                   (torchvision Resize default, datasets.py:141).
  center_square    timm RandomResizedCropAndInterpolation with scale=(1,1), ratio=(1,1): the whole image when square,
                   else the centred min(W,H) square (its fallback; datasets.py:127-128).
  enhance_*        PIL.ImageEnhance Brightness / Contrast / Color as torchvision ColorJitter applies them to PIL images:
                   Image.blend(degenerate, image, factor) with truncation toward zero inside [0,1] and clamping outside;
                   L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16.
  to_tensor_normalize   ToTensor + Normalize(mean, std) -> float32 CHW.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def resample_coeffs(in_size, out_size, filt):
    """(bounds [out][2] = (first input sample, count), integer weights [out][ksize]) of one axis, as Pillow computes them."""
    f, support = FILTERS[filt]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)          # Pillow accumulates left to right in double: python floats do the same
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One resampling pass along `axis` (0 = rows/vertical, 1 = columns/horizontal) of an HWC uint8 image."""
    src = img.astype(np.int64)
    n_out = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n_out
    out = np.zeros(shape, dtype=np.uint8)
    for o in range(n_out):
        lo, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(np.delete(img.shape, axis), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(cnt):
            acc = acc + np.take(src, lo + t, axis=axis) * int(kk[o, t])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 0:
            out[o] = v
        else:
            out[:, o] = v
    return out


def resize_u8(img, out_h, out_w, filt="bicubic"):
    """HWC uint8 -> [out_h][out_w][C] uint8, Pillow's Image.resize((out_w, out_h), filt) (horizontal pass, then vertical;
    a pass whose size does not change is skipped, as Pillow does)."""
    h, w = img.shape[:2]
    out = img
    if w != out_w:
        b, k = resample_coeffs(w, out_w, filt)
        out = _pass(out, b, k, 1)
    if h != out_h:
        b, k = resample_coeffs(h, out_h, filt)
        out = _pass(out, b, k, 0)
    return out


def center_square(img):
    h, w = img.shape[:2]
    if h == w:
        return img
    side = min(h, w)
    top, left = (h - side) // 2, (w - side) // 2
    return img[top:top + side, left:left + side]


def _luma(img):
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def _blend(deg, img, factor):
    """PIL Image.blend(deg, img, factor) for uint8 (libImaging/Blend.c): interpolation truncates, extrapolation clamps."""
    a = np.float32(factor)
    d, i = deg.astype(np.float32), img.astype(np.float32)
    if 0.0 <= factor <= 1.0:
        # (UINT8)((int)in1 + alpha * ((int)in2 - (int)in1)): float arithmetic, truncation toward zero
        return (d + a * (i - d)).astype(np.int32).astype(np.uint8)
    t = d + a * (i - d)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def enhance_brightness(img, factor):
    return _blend(np.zeros_like(img), img, factor)


def enhance_contrast(img, factor):
    mean = int(_luma(img).astype(np.float64).mean() + 0.5)     # ImageStat.Stat(image.convert("L")).mean[0] + 0.5
    return _blend(np.full_like(img, mean), img, factor)


def enhance_color(img, factor):
    gray = _luma(img)
    return _blend(np.repeat(gray[..., None], 3, axis=2), img, factor)


ENHANCERS = (enhance_brightness, enhance_contrast, enhance_color)   # torchvision ColorJitter fn ids 0, 1, 2


def color_jitter(img, order, factors):
    """Apply brightness / contrast / saturation in `order` (a permutation of (0,1,2)) with `factors[op]`."""
    for op in order:
        img = ENHANCERS[op](img, float(factors[op]))
    return img


def to_tensor_normalize(img, mean, std):
    a = img.astype(np.float32) / np.float32(255.0)
    a = (a - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(a.transpose(2, 0, 1))
