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


# ----------------------------------------------------------------------------------------------------------------------
# OpenCV's cv2.resize(img, (w, h)) for uint8 images with the default INTER_LINEAR — what the reference's InternVideo2
# processor calls (R/model/videogpt_plus/model/internvideo/utils.py:124: `cv2.resize(x, target_size)`).
# SOURCE ABSENT: opencv-python is not installed in this image and is unpinned in R/requirements.txt (`opencv-python`), so this
# restates the PUBLISHED algorithm of OpenCV 4.x (modules/imgproc/src/resize.cpp) and is "parity unpinned" against cv2 itself;
# it is anchored on properties cv2 is known to have (identity, exact 2x2 box mean at 2x down-scaling, constants preserved)
# and on hand-computed vectors (tests/test_preproc.py).  The algorithm:
#   * cv::resize: dsize == ssize -> copy; INTER_LINEAR with an exact 2x down-scale on both axes is re-routed to the fast
#     INTER_AREA path ("INTER_AREA (fast) also is equal to INTER_LINEAR"): dst = (a + b + c + d + 2) >> 2;
#   * otherwise resizeGeneric_ with fixed-point taps (INTER_RESIZE_COEF_BITS = 11): for every output x,
#     fx = (float)((x + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx; sx < 0 -> (0, fx = 0); sx >= w - 1 -> (w - 1, fx = 0);
#     taps = saturate_cast<short>({1 - fx, fx} * 2048) (round half to even); rows likewise but WITHOUT zeroing fy: the two
#     source rows are clamped to [0, h - 1] instead;
#   * HResizeLinear: row buffer S[x] = src[sx] * a0 + src[sx + 1] * a1 (int32);
#   * VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>: dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
CV_COEF_BITS = 11


def cv_linear_taps(in_size, out_size, clamp_frac):
    """-> (first source index [out], second source index [out], taps int [out,2]).  clamp_frac: the horizontal rule (fraction
    zeroed at the borders); the vertical pass keeps the fraction and clamps the row indices."""
    scale = 1.0 / (float(out_size) / float(in_size))
    i0 = np.zeros(out_size, np.int64)
    i1 = np.zeros(out_size, np.int64)
    taps = np.zeros((out_size, 2), np.int64)
    for d in range(out_size):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        if clamp_frac:
            if s < 0:
                s, f = 0, np.float32(0.0)
            if s >= in_size - 1:
                s, f = in_size - 1, np.float32(0.0)
        c0, c1 = np.float32(1.0) - f, f
        taps[d, 0] = int(np.rint(np.float32(c0 * np.float32(1 << CV_COEF_BITS))))     # cvRound: round half to even
        taps[d, 1] = int(np.rint(np.float32(c1 * np.float32(1 << CV_COEF_BITS))))
        i0[d] = min(max(s, 0), in_size - 1)
        i1[d] = min(max(s + 1, 0), in_size - 1)
    return i0, i1, taps


def cv2_resize_linear_u8(img, hw):
    """cv2.resize(img, (w, h)) (INTER_LINEAR) of an [H,W,C] uint8 image, plain loops."""
    h, w = hw
    H, W, C = img.shape
    if (H, W) == (h, w):
        return img.copy()
    src = img.astype(np.int64)
    out = np.empty((h, w, C), np.uint8)
    if H == 2 * h and W == 2 * w:               # the INTER_AREA fast path the call is re-routed to
        for y in range(h):
            for x in range(w):
                out[y, x] = (src[2 * y, 2 * x] + src[2 * y, 2 * x + 1] + src[2 * y + 1, 2 * x] + src[2 * y + 1, 2 * x + 1] + 2) >> 2
        return out
    x0, x1, xa = cv_linear_taps(W, w, True)
    y0, y1, yb = cv_linear_taps(H, h, False)
    rows = np.empty((H, w, C), np.int64)        # HResizeLinear of every source row
    for x in range(w):
        rows[:, x] = src[:, x0[x]] * xa[x, 0] + src[:, x1[x]] * xa[x, 1]
    for y in range(h):
        s0, s1 = rows[y0[y]], rows[y1[y]]
        out[y] = (((yb[y, 0] * (s0 >> 4)) >> 16) + ((yb[y, 1] * (s1 >> 4)) >> 16) + 2) >> 2
    return out
