"""Post-processing kernels (vg_postproc.hip) on the MI355X: time per call and bytes/s against the HBM roof.
usage: python tools/bench_postproc.py [reps]   — algorithmic bytes: the caller-visible inputs + outputs once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


class OP:       # seeded blobby masks (box-filtered noise above a quantile); no reference logic involved
    @staticmethod
    def blobs(shape, seed, density=0.5, smooth=3):
        rng = np.random.RandomState(seed)
        x = rng.rand(*shape).astype(np.float32)
        for ax in (-2, -1):
            for _ in range(smooth):
                x = (np.roll(x, 1, ax) + x + np.roll(x, -1, ax)) / 3
        return x > np.quantile(x, 1 - density)
from videoglamm_amd import ops      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


for name, shape, dens in [("8 x 1024^2 (C1 SAM frames)", (8, 1024, 1024), 0.5), ("32 x 480x854 (DAVIS clip)", (32, 480, 854), 0.3),
                          ("8 x 512^2 (C1 output masks)", (8, 512, 512), 0.5), ("64 x 256^2 (low-res scores)", (64, 256, 256), 0.5)]:
    m = torch.from_numpy(OP.blobs(shape, 1, density=dens, smooth=4)).to(dev)
    g = torch.from_numpy(OP.blobs(shape, 2, density=dens, smooth=4)).to(dev)
    s = torch.where(m, 1.0, -1.0)
    px = m.numel()
    rows = [("connected_components(8)", lambda: ops.connected_components(m, 8), 9 * px),
            ("remove_small_blobs(20)", lambda: ops.remove_small_blobs(m, 20), 2 * px),
            ("fill_holes(8)", lambda: ops.fill_holes(s, 8), 8 * px),
            ("mask_pair_counts diag", lambda: ops.mask_pair_counts(m, g, diagonal=True), 2 * px),
            ("boundary_counts r=8", lambda: ops.boundary_counts(m, g, 8), 2 * px)]
    print(name)
    for label, fn, nbytes in rows:
        us = timed(fn)
        print(f"  {label:26s} {us:9.1f} us   {nbytes / us / 1e3:8.1f} GB/s algorithmic  ({px / us / 1e3:.2f} Gpixel/s)")
