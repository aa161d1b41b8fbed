"""Mask post-processing and evaluation metrics on the device.

Same names, argument meaning and return values as the reference's helpers, but the masks stay in HBM: the pixel work
(connected components, boundary maps, dilation, intersections) runs in videoglamm_amd/csrc/vg_postproc.hip and only
a handful of integer counts per mask pair cross PCIe; the scalar arithmetic on those counts (divisions, the greedy
pairing of compute_miou, the F formula) is the reference's float64 host arithmetic.

  remove_small_blobs           R/eval_gcg_infer.py:20-29
  get_connected_components     R/model/segment_anything_2/sam2/utils/misc.py:47-63
  fill_holes_in_mask_scores    R/model/segment_anything_2/sam2/utils/misc.py:216-227
  compute_iou / compute_miou   R/eval_gcg_metrics.py:26-60
  db_eval_iou                  R/eval_referdavis_metrics.py:147-176   (void_pixels is not supported)
  db_eval_boundary / f_measure R/eval_referdavis_metrics.py:178-259
"""
import numpy as np
import torch

from . import ops


DEVICE = "cuda"     # where host inputs (numpy arrays / CPU tensors) are uploaded to


def _dev(x):
    """numpy / torch, host / device -> device tensor (bool stays bool)."""
    if not torch.is_tensor(x):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x if x.is_cuda else x.to(DEVICE)


def _mask(x):
    x = _dev(x)
    return x if x.dtype in (torch.bool, torch.uint8) else x != 0


def remove_small_blobs(binary_mask, min_size=0):
    """clear 4-connected blobs of fewer than min_size pixels; [..., H, W] (every leading index is its own image).
    Returns a device tensor of the input's dtype (uint8 for numpy bool input)."""
    m = _mask(binary_mask)
    if min_size <= 0:
        return m
    out = ops.remove_small_blobs(m, min_size)
    return out.view(torch.bool) if m.dtype == torch.bool else out


def get_connected_components(mask):
    """mask [N,1,H,W] (1 = foreground) -> (labels, counts) int32 [N,1,H,W], 8-connectivity."""
    return ops.connected_components(_mask(mask), 8)


def fill_holes_in_mask_scores(mask, max_area):
    assert max_area > 0, "max_area must be positive"
    return ops.fill_holes(_dev(mask).float(), max_area)


def iou_matrix(pred_masks, gt_masks):
    """[P, ...] x [G, ...] -> float64 [P, G] of |p & g| / |p | g| (nan where both are empty, like compute_iou)."""
    inter, uni = ops.mask_pair_counts(_mask(pred_masks), _mask(gt_masks))
    inter, uni = inter.cpu().numpy(), uni.cpu().numpy()
    with np.errstate(invalid="ignore", divide="ignore"):
        return inter / uni


def compute_iou(mask1, mask2):
    return iou_matrix(_mask(mask1)[None], _mask(mask2)[None])[0, 0]


def compute_miou(pred_masks, gt_masks):
    """greedy one-to-one pairing by descending IoU, mean of the paired IoUs (0.0 when nothing pairs)."""
    pred_masks, gt_masks = list(pred_masks), list(gt_masks)
    if not pred_masks or not gt_masks:
        return 0.0
    iou = iou_matrix(torch.stack([_mask(p) for p in pred_masks]), torch.stack([_mask(g) for g in gt_masks]))
    paired = []
    while iou.size > 0 and np.max(iou) > 0:
        i, j = np.unravel_index(np.argmax(iou, axis=None), iou.shape)
        paired.append(iou[i, j])
        iou = np.delete(iou, i, axis=0)
        iou = np.delete(iou, j, axis=1)
    return np.mean(paired) if paired else 0.0


def db_eval_iou(annotation, segmentation, void_pixels=None):
    """per-frame Jaccard of [H,W] or [T,H,W] masks; 1 where the union is empty."""
    if void_pixels is not None:
        raise NotImplementedError("void_pixels are not used on this path (R/eval_referdavis_metrics.py passes None)")
    a, s = _mask(annotation), _mask(segmentation)
    assert a.shape == s.shape, f"Annotation({tuple(a.shape)}) and segmentation:{tuple(s.shape)} dimensions do not match."
    inter, uni = ops.mask_pair_counts(a.reshape(-1, *a.shape[-2:]), s.reshape(-1, *s.shape[-2:]), diagonal=True)   # all frames, one launch
    inter, uni = inter.cpu().numpy(), uni.cpu().numpy()
    with np.errstate(invalid="ignore", divide="ignore"):
        j = inter / uni
    j[np.isclose(uni, 0)] = 1
    return j[0] if a.dim() == 2 else j


def _f_from_counts(n_fg, n_gt, fg_match, gt_match):
    if n_fg == 0 and n_gt > 0:
        precision, recall = 1, 0
    elif n_fg > 0 and n_gt == 0:
        precision, recall = 0, 1
    elif n_fg == 0 and n_gt == 0:
        precision, recall = 1, 1
    else:
        precision = fg_match / float(n_fg)
        recall = gt_match / float(n_gt)
    if precision + recall == 0:
        return 0
    return 2 * precision * recall / (precision + recall)


def db_eval_boundary(annotation, segmentation, void_pixels=None, bound_th=0.008):
    """boundary F-measure per frame of [H,W] or [T,H,W] masks."""
    if void_pixels is not None:
        raise NotImplementedError("void_pixels are not used on this path (R/eval_referdavis_metrics.py passes None)")
    a, s = _mask(annotation), _mask(segmentation)
    assert a.shape == s.shape
    if a.dim() not in (2, 3):
        raise ValueError(f"db_eval_boundary does not support tensors with {a.dim()} dimensions")
    H, W = a.shape[-2:]
    radius = int(bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm((H, W))))
    counts = ops.boundary_counts(s, a, radius).cpu().numpy()
    f = np.array([_f_from_counts(*(int(v) for v in row)) for row in counts], dtype=np.float64)
    return f[0] if a.dim() == 2 else f


def f_measure(foreground_mask, gt_mask, void_pixels=None, bound_th=0.008):
    return db_eval_boundary(gt_mask, foreground_mask, void_pixels, bound_th)
