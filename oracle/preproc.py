"""CPU restatement of the image pre-processing row H1 — TEST INFRASTRUCTURE ONLY (imported by
tests/ as the checker of videoglamm_amd/preproc.py + csrc/vg_preproc.hip; the product never imports it).

The reference resizes with torchvision / PIL (SAM: `resize(to_pil_image(x), target_size)`, R/utils/sam_transforms.py:44-49),
the CLIP processor (PIL bicubic) and cv2 (InternVideo2, R/model/videogpt_plus/model/internvideo/utils.py:105-143; cv2 is
absent here and videoglamm_amd/host.py stands in PIL's bilinear for it).  The resampling algorithm therefore lives in a
third-party dependency, Pillow (12.2.0 in this image; unpinned in R/requirements.txt): its published algorithm
(src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc) is restated
below with plain loops and PINNED against Pillow itself, which is importable here (tests/test_preproc.py runs both on
seeded images, up- and down-scaling, both filters, odd sizes: bit-exact).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_filter(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def bicubic_filter(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (bilinear_filter, 1.0), "bicubic": (bicubic_filter, 2.0)}


def precompute_coeffs(in_size, out_size, filt):
    """Resample.c:precompute_coeffs + normalize_coeffs_8bpc for the full box (0, in_size) -> (bounds [out,2], kk [out,ksize] int)."""
    f, fsupport = FILTERS[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation towards zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img, out_size, filt, axis):
    """one 8-bit pass along `axis` (0 = vertical, 1 = horizontal) of an [H,W,C] uint8 image."""
    bounds, kk = precompute_coeffs(img.shape[axis], out_size, filt)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[xmin + x] * kk[xx, x]
        out[xx] = clip8(acc)
    return np.moveaxis(out, 0, axis)


def pil_resize(img, hw, filt):
    """Image.resize((w, h), filter) of an RGB uint8 image: horizontal pass, then vertical, uint8 in between
    (Resample.c:ImagingResampleInner; a pass whose size does not change is skipped)."""
    h, w = hw
    if img.shape[1] != w:
        img = resample_axis(img, w, filt, 1)
    if img.shape[0] != h:
        img = resample_axis(img, h, filt, 0)
    return img
