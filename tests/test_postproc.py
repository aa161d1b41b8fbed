"""Mask post-processing and evaluation metrics: oracle vs the reference's own outputs
(tests/golden/postproc.npz), host mirror (videoglamm_amd/postproc.py) on the CPU twins, and — on the MI355X — the HIP
kernels (vg_postproc.hip) against the oracle, bit-exact (integer / byte work)."""
import numpy as np
import pytest
import torch

import _golden as G
from oracle import postproc as OP


def gold():
    z = G.fixture("postproc.npz")
    return {k: v.numpy() for k, v in z.items()}


def brute_components(mask, conn):
    """flood fill in plain Python: the independent check of the scipy-based restatement (small inputs only)."""
    H, W = mask.shape
    lab = np.zeros((H, W), np.int32)
    cnt = np.zeros((H, W), np.int32)
    nb = [(-1, 0), (1, 0), (0, -1), (0, 1)] + ([(-1, -1), (-1, 1), (1, -1), (1, 1)] if conn == 8 else [])
    for y in range(H):
        for x in range(W):
            if mask[y, x] and lab[y, x] == 0:
                stack, comp = [(y, x)], []
                lab[y, x] = y * W + x + 1
                while stack:
                    cy, cx = stack.pop()
                    comp.append((cy, cx))
                    for dy, dx in nb:
                        ny, nx = cy + dy, cx + dx
                        if 0 <= ny < H and 0 <= nx < W and mask[ny, nx] and lab[ny, nx] == 0:
                            lab[ny, nx] = y * W + x + 1
                            stack.append((ny, nx))
                for cy, cx in comp:
                    cnt[cy, cx] = len(comp)
    return lab, cnt


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_oracle_matches_reference_outputs():
    g = gold()
    pred, gt = g["pred"].astype(bool), g["gt"].astype(bool)
    iou = np.array([[OP.compute_iou(p, q) for q in gt] for p in pred])
    np.testing.assert_array_equal(iou, g["iou"])
    assert OP.compute_miou(list(pred), list(gt)) == g["miou"]
    for i in range(pred.shape[0]):
        for j in range(gt.shape[0]):
            np.testing.assert_array_equal(np.asarray(OP.db_eval_iou(gt[j], pred[i]), np.float64), g["jaccard"][i, j])
        for t in range(pred.shape[1]):
            np.testing.assert_array_equal(OP.seg2bmap(pred[i, t]), g["bmap"][i, t].astype(bool))


@pytest.mark.parametrize("conn", [4, 8])
def test_oracle_components_vs_flood_fill(conn):
    m = OP.blobs((2, 23, 37), 7, density=0.45, smooth=1)
    lab, cnt = OP.connected_components(m, conn)
    for n in range(2):
        bl, bc = brute_components(m[n], conn)
        np.testing.assert_array_equal(lab[n], bl)
        np.testing.assert_array_equal(cnt[n], bc)


def test_oracle_blobs_holes_dilate():
    from scipy import ndimage as ndi
    m = OP.blobs((31, 45), 3, density=0.4, smooth=1)
    out = OP.remove_small_blobs(m, 6)
    _, cnt = brute_components(m, 4)
    np.testing.assert_array_equal(out, m & (cnt >= 6))
    assert OP.remove_small_blobs(m, 0) is not None and (OP.remove_small_blobs(m, 0) == m).all()
    s = np.where(m, 1.5, -2.0).astype(np.float32)
    filled = OP.fill_holes_in_mask_scores(s[None, None], 5)[0, 0]
    _, hc = brute_components(~m, 8)
    np.testing.assert_array_equal(filled, np.where(~m & (hc <= 5), np.float32(0.1), s))
    b = OP.seg2bmap(m)
    for r in (1, 3, 6):     # the shift-OR restatement of cv2.dilate vs scipy's binary dilation
        np.testing.assert_array_equal(OP.dilate(b, OP.disk(r)), ndi.binary_dilation(b, structure=OP.disk(r)))
    assert OP.f_measure(m, m) == 1 and OP.f_measure(m, np.zeros_like(m)) == 0
    assert OP.bound_pix((480, 854)) == 8


# ------------------------------------------------------------------------------------------------ host mirror
def check_host(device):
    from videoglamm_amd import postproc as PP
    g = gold()
    pred, gt = g["pred"].astype(bool), g["gt"].astype(bool)
    dp, dg = torch.from_numpy(pred).to(device), torch.from_numpy(gt).to(device)
    np.testing.assert_array_equal(PP.iou_matrix(dp, dg), g["iou"])                    # pinned to the reference
    assert PP.compute_miou(list(dp), list(dg)) == g["miou"]
    assert PP.compute_miou([], list(dg)) == 0.0
    assert PP.compute_iou(pred[0], gt[1]) == g["iou"][0, 1]                           # numpy in, uploaded
    for i in range(pred.shape[0]):
        for j in range(gt.shape[0]):
            np.testing.assert_array_equal(PP.db_eval_iou(dg[j], dp[i]), g["jaccard"][i, j])
            f = PP.db_eval_boundary(dg[j], dp[i])
            np.testing.assert_array_equal(f, OP.db_eval_boundary(gt[j], pred[i]))
    assert PP.db_eval_iou(dg[0, 0], dp[0, 0]) == g["jaccard"][0, 0, 0]
    assert PP.f_measure(dp[0, 0], dg[0, 0], bound_th=2) == OP.f_measure(pred[0, 0], gt[0, 0], 2)
    with pytest.raises(NotImplementedError):
        PP.db_eval_iou(dg[0], dp[0], void_pixels=dg[0])
    out = PP.remove_small_blobs(dp, 20)
    assert out.dtype == torch.bool and out.shape == dp.shape
    ref = np.stack([OP.remove_small_blobs(x, 20) for x in pred.reshape(-1, *pred.shape[-2:])]).reshape(pred.shape)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    assert PP.remove_small_blobs(dp, 0) is dp
    s = torch.from_numpy(np.where(pred[:, :, None], 2.0, -1.0).astype(np.float32)).to(device).flatten(0, 1)   # [N,1,H,W]
    np.testing.assert_array_equal(PP.fill_holes_in_mask_scores(s, 8).cpu().numpy(), OP.fill_holes_in_mask_scores(s.cpu().numpy(), 8))
    lab, cnt = PP.get_connected_components(s > 0)
    rl, rc = OP.connected_components(s.cpu().numpy()[:, 0] > 0, 8)
    np.testing.assert_array_equal(lab.cpu().numpy()[:, 0], rl)
    np.testing.assert_array_equal(cnt.cpu().numpy()[:, 0], rc)


def test_host_mirror_cpu(cpu_ops, monkeypatch):
    from videoglamm_amd import postproc as PP
    monkeypatch.setattr(PP, "DEVICE", "cpu")
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))    # CPU tensors stand in for device tensors
    check_host(torch.device("cpu"))


@pytest.mark.gpu
def test_host_mirror_hip(cuda):
    check_host(cuda)


# ------------------------------------------------------------------------------------------------ HIP kernels
SHAPES = [(1, 1, 1), (2, 7, 5), (3, 46, 83), (2, 64, 64), (2, 130, 257), (1, 480, 854)]


@pytest.mark.gpu
@pytest.mark.parametrize("conn", [4, 8])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("density", [0.1, 0.5, 0.9])
def test_connected_components_hip(cuda, shape, conn, density):
    from videoglamm_amd import ops
    m = OP.blobs(shape, 11 + shape[1], density=density, smooth=2 if shape[1] > 8 else 0)
    lab, cnt = ops.connected_components(torch.from_numpy(m).to(cuda), conn)
    rl, rc = OP.connected_components(m, conn)
    np.testing.assert_array_equal(lab.cpu().numpy(), rl)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)


@pytest.mark.gpu
def test_connected_components_hard_cases_hip(cuda):
    """long snakes (deep union-find chains), full and empty images, checkerboards (every pixel its own 4-component,
    one 8-component), a spiral."""
    from videoglamm_amd import ops
    H, W = 96, 200
    snake = np.zeros((H, W), bool)
    snake[::2] = True
    snake[1::4, -1] = True
    snake[3::4, 0] = True
    yy, xx = np.mgrid[:H, :W]
    checker = (yy + xx) % 2 == 0
    spiral = np.zeros((H, W), bool)
    y0, y1, x0, x1 = 0, H - 1, 0, W - 1
    while y0 <= y1 and x0 <= x1:
        spiral[y0, x0:x1 + 1] = True
        spiral[y0:y1 + 1, x1] = True
        spiral[y1, x0 + 2:x1 + 1] = True
        spiral[y0 + 2:y1 + 1, x0 + 2] = True
        y0, y1, x0, x1 = y0 + 2, y1 - 2, x0 + 2, x1 - 2
    m = np.stack([snake, checker, spiral, np.ones((H, W), bool), np.zeros((H, W), bool)])
    for conn in (4, 8):
        lab, cnt = ops.connected_components(torch.from_numpy(m).to(cuda), conn)
        rl, rc = OP.connected_components(m, conn)
        np.testing.assert_array_equal(lab.cpu().numpy(), rl)
        np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    # BASELINE-size property: 8 x 1024^2 frames; areas over distinct labels add up to the foreground size
    big = torch.from_numpy(OP.blobs((8, 1024, 1024), 5, density=0.5, smooth=4)).to(cuda)
    lab, cnt = ops.connected_components(big, 8)
    roots = lab == (torch.arange(1024 * 1024, device=cuda, dtype=torch.int32).view(1, 1024, 1024) + 1)
    assert int(cnt[roots].sum()) == int(big.sum())
    assert bool(((lab > 0) == big).all()) and bool((cnt[big] > 0).all())
    small = ops.remove_small_blobs(big, 20)
    cnt4 = ops.connected_components(big, 4)[1]                              # blobs are 4-connected components
    assert bool((small.bool() == (big & (cnt4 >= 20))).all())
    assert bool((ops.remove_small_blobs(small, 20) == small).all())          # idempotent


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_blobs_holes_counts_hip(cuda, shape):
    from videoglamm_amd import ops
    m = OP.blobs(shape, 21, density=0.45, smooth=1 if shape[1] > 8 else 0)
    g = OP.blobs(shape, 22, density=0.5, smooth=1 if shape[1] > 8 else 0)
    dm, dg = torch.from_numpy(m).to(cuda), torch.from_numpy(g).to(cuda)
    out = ops.remove_small_blobs(dm, 6).cpu().numpy().astype(bool)
    np.testing.assert_array_equal(out, np.stack([OP.remove_small_blobs(x, 6) for x in m]))
    s = np.where(m, 0.7, -0.3).astype(np.float32)
    s.flat[::17] = 0.0                                                      # exact zeros are background (score <= 0)
    np.testing.assert_array_equal(ops.fill_holes(torch.from_numpy(s).to(cuda), 4).cpu().numpy(), OP.fill_holes_in_mask_scores(s, 4))
    inter, uni = ops.mask_pair_counts(dm, dg)
    a, b = m.reshape(shape[0], -1), g.reshape(shape[0], -1)
    np.testing.assert_array_equal(inter.cpu().numpy(), (a[:, None] & b[None]).sum(-1))
    np.testing.assert_array_equal(uni.cpu().numpy(), (a[:, None] | b[None]).sum(-1))
    di, du = ops.mask_pair_counts(dm, dg, diagonal=True)
    np.testing.assert_array_equal(di.cpu().numpy(), (a & b).sum(-1))
    np.testing.assert_array_equal(du.cpu().numpy(), (a | b).sum(-1))
    u8 = torch.from_numpy((m * 200).astype(np.uint8)).to(cuda)              # any nonzero byte counts as set
    np.testing.assert_array_equal(ops.mask_pair_counts(u8, dg)[0].cpu().numpy(), inter.cpu().numpy())
    for r in (0, 1, OP.bound_pix(shape[1:]), 8):
        c = ops.boundary_counts(dm, dg, r).cpu().numpy()
        np.testing.assert_array_equal(c, np.array([OP.boundary_counts(x, y, r) for x, y in zip(m, g)]))


@pytest.mark.gpu
def test_postproc_argument_errors(cuda):
    from videoglamm_amd import _lib, ops
    m = torch.zeros(1, 8, 8, dtype=torch.bool, device=cuda)
    with pytest.raises(_lib.VGKernelError):
        ops.connected_components(m, 6)
    with pytest.raises(_lib.VGKernelError):
        ops.boundary_counts(m, m, 40)
    with pytest.raises(_lib.VGKernelError):
        ops.fill_holes(torch.zeros(1, 8, 8, device=cuda), 0)
    with pytest.raises(_lib.VGKernelError):
        ops.mask_pair_counts(m, torch.zeros(2, 8, 8, dtype=torch.bool, device=cuda), diagonal=True)
