"""Per-mask IoU of the bf16 HIP path against the fp32 reference fixture (micro SAM2), framewise branch — shows whether a
mIoU change is spread over all masks (an accuracy change) or is one multimask/stability selection flipping (a discrete
decision that bf16 rounding can move either way)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _golden as G  # noqa: E402
from oracle import seeded  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402
from videoglamm_amd.sam2 import SAM2  # noqa: E402

cuda = torch.device("cuda:0")
fx = G.fixture("sam2_micro.npz")
T, N, H, W = [int(v) for v in fx["meta"]]
sd = G.weights("sam2_micro_manifest.json", 1, seeded.sam2_overrides())
for dt in (torch.float32, torch.bfloat16):
    m = SAM2(Params(sd, cuda, dt), "", G.sam2_cfg())
    images = G.rnd((T, 3, m.S, m.S), 11).to(cuda)
    text = G.rnd((N, 256), 12, 0.5).to(cuda)
    fw, _ = m.framewise_branch(images, text, (H, W))
    a = (fw.cpu() > 0).numpy()
    b = (fx["framewise_logits"] > 0).numpy()
    ious = [[float((a[t, n] & b[t, n]).sum() / max((a[t, n] | b[t, n]).sum(), 1)) for n in range(N)] for t in range(T)]
    print(dt, "per-mask IoU [t][n]:", np.round(np.array(ious), 4).tolist())
