"""Row H1 of SURVEY.md §8a on the device (§8f row 1): frame sub-sampling stays on the host (index arithmetic), the
decoded uint8 frames are uploaded ONCE and the three model inputs are produced in HBM by csrc/vg_preproc.hip:

    images          [Te,3,224,224]   cv2.resize (INTER_LINEAR), (x/255 - mean)/std R/.../internvideo/utils.py:105-143
    context_images  [Te,3,336,336]   CLIP processor: bicubic short side 336, centre crop, /255, mean/std
                                                                                 R/utils/enc_preprocessors.py:120-166
    images_for_sam  [T,3,1024,1024]  resize longest side, (x - mean)/std, bilinear stretch to 1024^2
                                                                                 R/utils/sam_transforms.py:26-65

Same results as videoglamm_amd/host.py (bit for bit: the SAM / CLIP resampler is Pillow's integer arithmetic, whose coefficient
tables are computed here in float64 exactly like Pillow's precompute_coeffs / normalize_coeffs_8bpc; the InternVideo2 stream is
OpenCV's 11-bit fixed-point INTER_LINEAR with the tap tables of host.cv_linear_taps), at 0.75 MB of
PCIe traffic per 512^2 frame instead of 12.6 MB per SAM frame plus the encoder tensors.
"""
import functools

import numpy as np
import torch

from . import host, ops

PRECISION_BITS = 32 - 8 - 2
DEVICE = "cuda"


def _filter(name, x):
    x = np.abs(x)
    if name == "bilinear":
        return np.where(x < 1.0, 1.0 - x, 0.0)
    a = -0.5        # Pillow's bicubic
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


@functools.lru_cache(maxsize=64)
def resample_coeffs(in_size, out_size, filt):
    """Pillow's per-output-pixel taps for the full box: (bounds int32 [out,2] = (first, count), coeffs int32 [out,ksize]).
    Vectorised over the outputs; the taps are accumulated left to right like the C loop, so every rounding matches."""
    support0 = {"bilinear": 1.0, "bicubic": 2.0}[filt]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.int64)[None, :]
    w = _filter(filt, ((taps + xmin[:, None]) - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(taps < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, np.float64)
    for t in range(ksize):
        ww = ww + w[:, t]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.trunc(np.where(k < 0, -0.5 + k * (1 << PRECISION_BITS), 0.5 + k * (1 << PRECISION_BITS)))
    kk = np.where(taps < xmax[:, None], kk, 0).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), kk


@functools.lru_cache(maxsize=64)
def _coeffs_on(device, in_size, out_size, filt):
    b, k = resample_coeffs(in_size, out_size, filt)
    return torch.from_numpy(b).to(device), torch.from_numpy(k).to(device)


def resize_u8(x, hw, filt="bilinear"):
    """PIL.Image.resize((w, h), filt) of every image of x [N,H,W,C] uint8 (device): horizontal pass, then vertical."""
    h, w = hw
    if x.shape[2] != w:
        x = ops.resample_u8(x, w, 1, *_coeffs_on(x.device, x.shape[2], w, filt))
    if x.shape[1] != h:
        x = ops.resample_u8(x, h, 0, *_coeffs_on(x.device, x.shape[1], h, filt))
    return x


def _frames(frames):
    """list of [H,W,3] uint8 arrays (one clip: equal sizes) or an [N,H,W,3] array / tensor -> device uint8 tensor."""
    if not torch.is_tensor(frames):
        frames = torch.from_numpy(np.ascontiguousarray(np.stack(list(frames))))
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3, "frames are [N,H,W,3] uint8"
    return frames if frames.is_cuda else frames.to(DEVICE)


def sam_preprocess(frames, img_size=1024):
    """-> (images_for_sam [T,3,S,S] fp32 on device, resize_shape (th, tw))."""
    x = _frames(frames)
    th, tw = host.get_preprocess_shape(x.shape[1], x.shape[2], img_size)
    x = resize_u8(x, (th, tw), "bilinear")
    x = ops.normalize_u8(x, host.SAM_MEAN.flatten().tolist(), host.SAM_STD.flatten().tolist(), 0)
    if (th, tw) != (img_size, img_size):
        x = ops.bilinear(x.reshape(-1, th, tw), img_size, img_size).reshape(x.shape[0], 3, img_size, img_size)
    return x, (th, tw)


@functools.lru_cache(maxsize=64)
def _cv_taps_on(device, in_size, out_size, clamp_frac):
    i0, i1, taps = host.cv_linear_taps(in_size, out_size, clamp_frac)
    return torch.from_numpy(np.stack([i0, i1], 1).astype(np.int32)).to(device), torch.from_numpy(np.ascontiguousarray(taps)).to(device)


def resize_cv_u8(x, hw):
    """cv2.resize(img, (w, h)) (INTER_LINEAR) of every image of x [N,H,W,C] uint8 (device): one launch, both passes."""
    h, w = hw
    if tuple(x.shape[1:3]) == (h, w):
        return x
    if x.shape[1] == 2 * h and x.shape[2] == 2 * w:
        return ops.resize_cv_linear_u8(x, h, w)
    xi, xa = _cv_taps_on(x.device, x.shape[2], w, True)
    yi, yb = _cv_taps_on(x.device, x.shape[1], h, False)
    return ops.resize_cv_linear_u8(x, h, w, xi, xa, yi, yb)


def iv2_preprocess(frames, size=224):
    x = resize_cv_u8(_frames(frames), (size, size))
    return ops.normalize_u8(x, host.IV2_MEAN, host.IV2_STD, 1)


def clip_preprocess(frames, size=336):
    x = _frames(frames)
    h, w = x.shape[1:3]
    s = size / min(h, w)
    nh, nw = max(size, int(round(h * s))), max(size, int(round(w * s)))
    x = resize_u8(x, (nh, nw), "bicubic")
    return ops.normalize_u8(x, host.CLIP_MEAN, host.CLIP_STD, 1, crop=((nh - size) // 2, (nw - size) // 2, size, size))


def preprocess_vision(np_images, type="video", enc_preprocessor=None, sam_preprocessor=None, conv_generator=None, precision="fp16"):
    """preprocess_vision — R/chat.py:402-489 (same parameters, same five values in the same order) with the pixel work on the device: the clip
    is uploaded once as uint8.  np_images: B x T x (H x W x C), batch of one; a clip is a list of equal-sized frames or one [T,H,W,3] uint8
    array / device tensor.  The preprocessor objects only carry NUM_FRAMES here (the arithmetic is host.py's, bit for bit); anything but
    host.py's own classes is refused — use host.preprocess_vision with custom preprocessors.
    -> (enc_image, enc_context_image, image_sam, original_size_list, resize_list)."""
    assert len(np_images) == 1, "Batch size must be 1"
    if not (enc_preprocessor is None or isinstance(enc_preprocessor, host.EncPreprocessor_VideoGPTPlus)) or \
            not (sam_preprocessor is None or isinstance(sam_preprocessor, host.SAM_v2_Preprocess)):
        raise TypeError("the device pre-processing path implements host.EncPreprocessor_VideoGPTPlus / host.SAM_v2_Preprocess only")
    dt = host.precision_dtype(precision)
    cast = (lambda x: x) if dt == torch.float32 else (lambda x: x.to(dt))
    frames = _frames(np_images[0])
    T = frames.shape[0]
    original_size_list = [tuple(frames.shape[1:3])]
    sam, shape = sam_preprocess(frames)
    if type == "image":
        assert T == 1, "Time dimension must be 1"
        return [cast(clip_preprocess(frames))], None, [cast(sam)], original_size_list, [shape]
    if type != "video":
        raise ValueError(f"type must be 'video' or 'image', got {type!r}")
    num_frames = getattr(conv_generator, "NUM_FRAMES", None) or getattr(enc_preprocessor, "num_frames", None) or host.NUM_FRAMES
    n_enc = getattr(enc_preprocessor, "num_frames", None) or num_frames
    idx = host.pad_or_truncate(host.subsample_frames(list(range(T)), num_frames), n_enc)
    enc = frames[torch.tensor(idx, device=frames.device)]
    return [cast(iv2_preprocess(enc))], [cast(clip_preprocess(enc))], [cast(sam)], original_size_list, [shape]
