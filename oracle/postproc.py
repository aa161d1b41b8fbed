"""CPU restatement of the reference's mask post-processing and evaluation metrics.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker
of the HIP kernels in videoglamm_amd/csrc/vg_postproc.hip; the product path never imports it.

numpy only (scipy.ndimage.label for the labelling itself).  Pinning status:
  * compute_iou / compute_miou (R/eval_gcg_metrics.py:26-60), db_eval_iou and _seg2bmap
    (R/eval_referdavis_metrics.py:147-176, 262-305) are pinned: tests/golden/postproc.npz holds the outputs of the
    reference's own functions on seeded masks (tests/golden/make_golden.py:gen_postproc).
  * remove_small_blobs (R/eval_gcg_infer.py:20-29) and f_measure (R/eval_referdavis_metrics.py:194-259) call
    scikit-image (`remove_small_objects`, `disk`) and OpenCV (`cv2.dilate`), both absent from this image and unpinned
    in R/requirements.txt:8,36: their published algorithms are restated below — PARITY UNPINNED for those two steps.
  * get_connected_components (R/model/segment_anything_2/sam2/utils/misc.py:47-63) is a CUDA extension
    (sam2/csrc/connected_components.cu) that cannot be built here (needs nvcc): the partition and the areas it
    documents (8-connectivity) are restated with scipy — PARITY UNPINNED for the label values, which nothing consumes.
"""
import numpy as np
from scipy import ndimage as ndi


def connected_components(mask, connectivity=8):
    """[N,H,W] bool -> (labels, counts) int32 [N,H,W].  labels: 0 on background, 1 + smallest linear index of the
    component on foreground (the canonical form the HIP kernel produces); counts: component area per pixel.
    misc.py:47-63 documents the outputs; connected_components.cu:213-282 is 8-connectivity."""
    mask = np.asarray(mask).astype(bool)
    structure = ndi.generate_binary_structure(2, 1 if connectivity == 4 else 2)
    labels = np.zeros(mask.shape, np.int32)
    counts = np.zeros(mask.shape, np.int32)
    H, W = mask.shape[-2:]
    idx = np.arange(H * W, dtype=np.int64).reshape(H, W)
    for n in range(mask.shape[0]):
        lab, k = ndi.label(mask[n], structure)
        if k == 0:
            continue
        first = np.asarray(ndi.minimum(idx, lab, index=np.arange(1, k + 1))).astype(np.int64)
        size = np.bincount(lab.ravel(), minlength=k + 1)
        fg = lab > 0
        labels[n][fg] = first[lab[fg] - 1] + 1
        counts[n][fg] = size[lab[fg]]
    return labels, counts


def remove_small_blobs(binary_mask, min_size=0):
    """R/eval_gcg_infer.py:20-29.  skimage.morphology.remove_small_objects(bool image, min_size) with its default
    connectivity=1: label with the 4-neighbourhood, bincount, clear components with size < min_size."""
    binary_mask = np.asarray(binary_mask)
    if min_size > 0:
        dtype = binary_mask.dtype
        m = binary_mask.astype(bool)
        lab, _ = ndi.label(m, ndi.generate_binary_structure(m.ndim, 1))
        sizes = np.bincount(lab.ravel())
        too_small = sizes < min_size
        too_small[0] = False
        out = m.copy()
        out[too_small[lab]] = False
        binary_mask = out.astype(dtype)
    return binary_mask


def fill_holes_in_mask_scores(mask, max_area):
    """misc.py:216-227 on [N,1,H,W] or [N,H,W] float scores."""
    assert max_area > 0, "max_area must be positive"
    mask = np.asarray(mask, np.float32)
    flat = mask.reshape(-1, *mask.shape[-2:])
    labels, areas = connected_components(flat <= 0, 8)
    is_hole = (labels > 0) & (areas <= max_area)
    return np.where(is_hole, np.float32(0.1), flat).reshape(mask.shape)


def compute_iou(mask1, mask2):
    """R/eval_gcg_metrics.py:26-37 (0/0 -> nan like the reference)."""
    intersection = np.logical_and(mask1, mask2)
    union = np.logical_or(mask1, mask2)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.sum(intersection) / np.sum(union)


def miou_from_matrix(iou_matrix):
    """greedy one-to-one pairing of R/eval_gcg_metrics.py:50-58."""
    iou_matrix = np.array(iou_matrix, dtype=np.float64)
    paired = []
    while iou_matrix.size > 0 and np.max(iou_matrix) > 0:
        i, j = np.unravel_index(np.argmax(iou_matrix, axis=None), iou_matrix.shape)
        paired.append(iou_matrix[i, j])
        iou_matrix = np.delete(iou_matrix, i, axis=0)
        iou_matrix = np.delete(iou_matrix, j, axis=1)
    return np.mean(paired) if paired else 0.0


def compute_miou(pred_masks, gt_masks):
    """R/eval_gcg_metrics.py:40-60."""
    pred_masks, gt_masks = list(pred_masks), list(gt_masks)
    m = np.zeros((len(pred_masks), len(gt_masks)))
    for i, p in enumerate(pred_masks):
        for j, g in enumerate(gt_masks):
            m[i, j] = compute_iou(p, g)
    return miou_from_matrix(m)


def db_eval_iou(annotation, segmentation):
    """R/eval_referdavis_metrics.py:147-176 without void pixels: per-frame Jaccard, 1 where the union is empty."""
    annotation = np.asarray(annotation).astype(bool)
    segmentation = np.asarray(segmentation).astype(bool)
    inters = np.sum(segmentation & annotation, axis=(-2, -1))
    union = np.sum(segmentation | annotation, axis=(-2, -1))
    with np.errstate(invalid="ignore", divide="ignore"):
        j = inters / union
    if np.ndim(j) == 0:
        return 1 if np.isclose(union, 0) else j
    j[np.isclose(union, 0)] = 1
    return j


def seg2bmap(seg):
    """R/eval_referdavis_metrics.py:262-305, same-size case: 1-pixel boundary towards the origin."""
    seg = np.asarray(seg).astype(bool)
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    s[:-1, :] = seg[1:, :]
    se[:-1, :-1] = seg[1:, 1:]
    b = (seg ^ e) | (seg ^ s) | (seg ^ se)
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = 0
    return b


def disk(radius):
    """skimage.morphology.disk: (2r+1)^2 footprint, x^2 + y^2 <= r^2."""
    r = int(radius)
    L = np.arange(-r, r + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2) <= r ** 2


def dilate(img, footprint):
    """cv2.dilate(img, kernel) on a 0/1 image: max over the (centre-anchored) footprint, pixels outside the image do
    not contribute (OpenCV's default morphology border)."""
    img = np.asarray(img).astype(bool)
    H, W = img.shape
    r = footprint.shape[0] // 2
    out = np.zeros_like(img)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            if not footprint[dy + r, dx + r]:
                continue
            ys, ye = max(0, -dy), min(H, H - dy)
            xs, xe = max(0, -dx), min(W, W - dx)
            if ys < ye and xs < xe:
                out[ys:ye, xs:xe] |= img[ys + dy:ye + dy, xs + dx:xe + dx]
    return out


def bound_pix(shape, bound_th=0.008):
    """R/eval_referdavis_metrics.py:215-216."""
    return int(bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(shape)))


def boundary_counts(foreground_mask, gt_mask, radius):
    """(n_fg, n_gt, fg_match, gt_match) of f_measure, R/eval_referdavis_metrics.py:218-236."""
    fg_boundary = seg2bmap(foreground_mask)
    gt_boundary = seg2bmap(gt_mask)
    fp = disk(radius)
    fg_dil = dilate(fg_boundary, fp)
    gt_dil = dilate(gt_boundary, fp)
    return (int(fg_boundary.sum()), int(gt_boundary.sum()), int((fg_boundary & gt_dil).sum()), int((gt_boundary & fg_dil).sum()))


def f_from_counts(n_fg, n_gt, fg_match, gt_match):
    """R/eval_referdavis_metrics.py:238-259."""
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


def f_measure(foreground_mask, gt_mask, bound_th=0.008):
    """R/eval_referdavis_metrics.py:194-259 without void pixels."""
    r = bound_pix(np.asarray(foreground_mask).shape, bound_th)
    return f_from_counts(*boundary_counts(foreground_mask, gt_mask, r))


def db_eval_boundary(annotation, segmentation, bound_th=0.008):
    """R/eval_referdavis_metrics.py:178-191."""
    annotation, segmentation = np.asarray(annotation), np.asarray(segmentation)
    if annotation.ndim == 3:
        return np.array([f_measure(segmentation[t], annotation[t], bound_th) for t in range(annotation.shape[0])])
    return f_measure(segmentation, annotation, bound_th)


def blobs(shape, seed, density=0.5, smooth=3):
    """seeded blobby test masks [..., H, W] (box-filtered noise above a quantile) — shared by tests, smoke and bench."""
    rng = np.random.RandomState(seed)
    x = rng.rand(*shape).astype(np.float32)
    for ax in (-2, -1):
        for _ in range(smooth):
            x = (np.roll(x, 1, ax) + x + np.roll(x, -1, ax)) / 3
    return x > np.quantile(x, 1 - density)
