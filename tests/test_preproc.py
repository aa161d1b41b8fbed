"""Image pre-processing on the device: Pillow's 8-bit resampler restated (oracle/preproc.py) and
pinned against Pillow itself; the product's coefficient tables; the device pipeline (videoglamm_amd/preproc.py) against
the host pipeline (videoglamm_amd/host.py, whose SAM branch is pinned to the reference by tests/test_host_rows.py) —
bit-exact for everything that is integer or a single IEEE operation, 1e-5 for the fp32 bilinear stretch of non-square
SAM inputs."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import preproc as OP

PIL_FILTER = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC}
SIZES = [((37, 53), (64, 80)), ((64, 80), (37, 53)), ((48, 64), (96, 128)), ((50, 70), (17, 23)), ((33, 33), (33, 50)),
         ((120, 90), (224, 224)), ((7, 5), (3, 9)), ((1, 9), (4, 4))]


def image(hw, seed, smooth=False):
    rng = np.random.RandomState(seed)
    x = rng.randint(0, 256, hw + (3,)).astype(np.uint8)
    if smooth:      # a natural-looking picture: low-pass filtered noise (PIL's own upscale of a tiny random image)
        small = rng.randint(0, 256, (max(2, hw[0] // 16), max(2, hw[1] // 16), 3)).astype(np.uint8)
        x = np.array(Image.fromarray(small).resize((hw[1], hw[0]), Image.BICUBIC))
    return x


def clip(T, hw, seed):
    return [image(hw, seed + t, smooth=True) for t in range(T)]


# ------------------------------------------------------------------------------------------------ oracle / tables (CPU)
@pytest.mark.parametrize("filt", ["bilinear", "bicubic"])
def test_oracle_resampler_is_pillow(filt):
    for i, (src, dst) in enumerate(SIZES):
        img = image(src, i)
        ref = np.array(Image.fromarray(img).resize((dst[1], dst[0]), PIL_FILTER[filt]))
        np.testing.assert_array_equal(OP.pil_resize(img, dst, filt), ref)


def test_product_coefficient_tables():
    from videoglamm_amd import preproc as PP
    for i, o in [(37, 64), (64, 37), (50, 17), (120, 224), (512, 1024), (854, 1024), (480, 576), (1080, 224), (3, 7), (5, 1)]:
        for filt in ("bilinear", "bicubic"):
            b, k = OP.precompute_coeffs(i, o, filt)
            b2, k2 = PP.resample_coeffs(i, o, filt)
            np.testing.assert_array_equal(b, b2)
            np.testing.assert_array_equal(k, k2)


# ------------------------------------------------------------------------------------------------ OpenCV INTER_LINEAR (InternVideo2 stream)
CV_SIZES = [((37, 53), (224, 224)), ((480, 640), (224, 224)), ((448, 448), (224, 224)), ((224, 224), (224, 224)), ((300, 225), (224, 224)),
            ((225, 223), (224, 224)), ((5, 7), (3, 2)), ((64, 80), (37, 53)), ((1, 9), (4, 4)), ((720, 1280), (224, 224))]


def test_cv2_linear_properties_and_hand_vectors():
    """OpenCV is absent (source absent: parity unpinned against cv2 itself): the restatement is anchored on what cv2.resize is known to
    do — identity at equal size, the exact 2x2 box mean at 2x down-scaling (the INTER_AREA re-route), constants preserved, 2-tap support
    without antialiasing — and on vectors computed by hand from the published fixed-point formulas."""
    from videoglamm_amd import host
    for fn in (OP.cv2_resize_linear_u8, host.cv2_resize_linear_u8):
        img = image((40, 56), 3)
        np.testing.assert_array_equal(fn(img, (40, 56)), img)
        big = image((48, 64), 4).astype(np.int64)
        box = ((big[0::2, 0::2] + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        np.testing.assert_array_equal(fn(big.astype(np.uint8), (24, 32)), box)
        for v in (0, 1, 77, 254, 255):
            assert np.unique(fn(np.full((10, 12, 3), v, np.uint8), (23, 31))).tolist() == [v]
        # one row, 4 -> 3 pixels: fx = (d + 0.5) * 4/3 - 0.5 = 1/6, 3/2, 17/6 -> (s, f) = (0, 1/6), (1, 1/2), (2, 5/6);
        # taps round(2048 * {1 - f, f}) = (1707, 341), (1024, 1024), (341, 1707); rows: same row twice with (b0, b1) from
        # fy = (0 + 0.5) * 1 - 0.5 = 0 -> (2048, 0) [same height would be a copy, so use 2 -> 2 rows via a 2 x 4 -> 2 x 3 call]
        row = np.array([[[10], [200], [30], [255]]], np.uint8).repeat(2, axis=0)          # [2,4,1], two equal rows
        s = [10 * 1707 + 200 * 341, 200 * 1024 + 30 * 1024, 30 * 341 + 255 * 1707]
        want = [(((2048 * (v >> 4)) >> 16) + 0 + 2) >> 2 for v in s]
        got = fn(row, (2, 3))
        assert got[:, :, 0].tolist() == [want, want], (got[:, :, 0].tolist(), want)
        # no antialiasing: at 8x down-scaling an output pixel sees 2 source pixels per axis, not the 8x8 area
        spike = np.zeros((64, 64, 1), np.uint8)
        spike[5, 5] = 255                                           # between the taps of every output pixel (taps at 3,4 and 11,12)
        assert fn(spike, (8, 8)).max() == 0
        # up-scaling 2x: interior taps are (1536, 512) / (512, 1536) on both axes
        up = fn(np.array([[[0], [100]], [[0], [100]]], np.uint8), (2, 4))      # rows equal -> pure horizontal
        assert up[0, :, 0].tolist() == [0, 25, 75, 100]


def test_cv2_linear_host_equals_oracle():
    from videoglamm_amd import host
    for i, (src, dst) in enumerate(CV_SIZES):
        for smooth in (False, True):
            img = image(src, 50 + i, smooth=smooth)
            np.testing.assert_array_equal(host.cv2_resize_linear_u8(img, dst), OP.cv2_resize_linear_u8(img, dst))
    # upscaling by a 2-tap filter: Pillow's bilinear has the same support there, the two differ by rounding only
    img = image((37, 53), 9)
    pil = np.array(Image.fromarray(img).resize((224, 224), Image.BILINEAR)).astype(int)
    assert np.abs(pil - host.cv2_resize_linear_u8(img, (224, 224)).astype(int)).max() <= 1
    # down-scaling: Pillow antialiases (support grows with the scale), cv2 does not — the streams must NOT be the same any more
    img = image((480, 640), 10)
    pil = np.array(Image.fromarray(img).resize((224, 224), Image.BILINEAR)).astype(int)
    assert np.abs(pil - host.cv2_resize_linear_u8(img, (224, 224)).astype(int)).max() > 30


@pytest.mark.gpu
def test_resize_cv_hip(cuda):
    from videoglamm_amd import preproc as PP
    for i, (src, dst) in enumerate(CV_SIZES):
        imgs = np.stack([image(src, 70 + 10 * i + n, smooth=n == 1) for n in range(3)])
        got = PP.resize_cv_u8(torch.from_numpy(imgs).to(cuda), dst).cpu().numpy()
        for n in range(3):
            np.testing.assert_array_equal(got[n], OP.cv2_resize_linear_u8(imgs[n], dst))


# ------------------------------------------------------------------------------------------------ pipeline vs host.py
def check_pipeline(device, hw, T, num_frames):
    from videoglamm_amd import host, preproc as PP
    frames = clip(T, hw, 40)
    cg = host.ConvGenerator_VideoGPTPlus(num_frames=num_frames)
    ref = host.preprocess_vision([frames], conv_generator=cg, precision="fp32")
    got = PP.preprocess_vision([torch.from_numpy(np.stack(frames)).to(device)], conv_generator=cg, precision="fp32")
    assert got[3] == ref[3] == [tuple(hw)] and got[4] == ref[4]                   # original_size_list, resize_list (R/chat.py:489's order)
    for name, g, r in zip(("images", "context_images"), got[:2], ref[:2]):
        assert g[0].shape == r[0].shape and g[0].dtype == torch.float32
        assert torch.equal(g[0].cpu(), r[0].cpu()), (name, (g[0].cpu() - r[0].cpu()).abs().max())
    g, r = got[2][0].cpu(), ref[2][0].cpu()
    assert g.shape == r.shape
    # type="image": the CLIP tensor of the one frame at position 0, no context stream
    gi = PP.preprocess_vision([[frames[0]]], type="image", precision="fp32")
    ri = host.preprocess_vision([[frames[0]]], type="image", precision="fp32")
    assert gi[1] is None and ri[1] is None and gi[3] == ri[3] and gi[4] == ri[4]
    assert torch.equal(gi[0][0].cpu(), ri[0][0].cpu()) and gi[2][0].shape == ri[2][0].shape == (1, 3, 1024, 1024)
    assert PP.preprocess_vision([frames], conv_generator=cg, precision="bf16")[2][0].dtype == torch.bfloat16
    if ref[4][0] == (1024, 1024):
        assert torch.equal(g, r)                                                   # no stretch: bit-exact
    else:
        torch.testing.assert_close(g, r, rtol=1e-5, atol=1e-5)                     # fp32 bilinear stretch


def test_pipeline_cpu(cpu_ops, monkeypatch):
    from videoglamm_amd import preproc as PP
    monkeypatch.setattr(PP, "DEVICE", "cpu")
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))     # CPU tensors stand in for device tensors
    check_pipeline(torch.device("cpu"), (64, 64), 3, 4)                            # square: SAM 1024^2 bit-exact
    check_pipeline(torch.device("cpu"), (45, 80), 2, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("hw,T,nf", [((64, 64), 3, 4), ((45, 80), 5, 4), ((90, 60), 2, 8), ((512, 512), 8, 8), ((480, 854), 4, 4)])
def test_pipeline_hip(cuda, hw, T, nf):
    check_pipeline(cuda, hw, T, nf)


@pytest.mark.gpu
@pytest.mark.parametrize("filt", ["bilinear", "bicubic"])
def test_resize_u8_hip_is_pillow(cuda, filt):
    from videoglamm_amd import preproc as PP
    for i, (src, dst) in enumerate(SIZES + [((300, 200), (336, 504)), ((256, 256), (1024, 1024))]):
        imgs = np.stack([image(src, 10 * i + n, smooth=n == 1) for n in range(2)])
        got = PP.resize_u8(torch.from_numpy(imgs).to(cuda), dst, filt).cpu().numpy()
        for n in range(2):
            ref = np.array(Image.fromarray(imgs[n]).resize((dst[1], dst[0]), PIL_FILTER[filt]))
            np.testing.assert_array_equal(got[n], ref)
    # extremes: saturated images stay saturated through the negative bicubic lobes (clip8)
    sat = np.zeros((1, 40, 40, 3), np.uint8)
    sat[:, ::2] = 255
    got = PP.resize_u8(torch.from_numpy(sat).to(cuda), (23, 61), filt).cpu().numpy()[0]
    np.testing.assert_array_equal(got, np.array(Image.fromarray(sat[0]).resize((61, 23), PIL_FILTER[filt])))


@pytest.mark.gpu
def test_normalize_u8_hip(cuda):
    import _cpu_ops
    from videoglamm_amd import host, ops
    x = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (2, 19, 23, 3)).astype(np.uint8))
    for mode, mean, std in ((0, host.SAM_MEAN.flatten().tolist(), host.SAM_STD.flatten().tolist()), (1, host.CLIP_MEAN, host.CLIP_STD)):
        for crop in (None, (3, 5, 11, 13)):
            got = ops.normalize_u8(x.to(cuda), mean, std, mode, crop=crop)
            assert torch.equal(got.cpu(), _cpu_ops.normalize_u8(x, mean, std, mode, crop=crop))
    bf = ops.normalize_u8(x.to(cuda), host.CLIP_MEAN, host.CLIP_STD, 1, out_dtype=torch.bfloat16)
    assert torch.equal(bf.cpu(), _cpu_ops.normalize_u8(x, host.CLIP_MEAN, host.CLIP_STD, 1).to(torch.bfloat16))
    from videoglamm_amd import _lib
    with pytest.raises(_lib.VGKernelError):
        ops.normalize_u8(x.to(cuda), host.CLIP_MEAN, host.CLIP_STD, 1, crop=(10, 10, 11, 13))
