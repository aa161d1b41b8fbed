"""Pre-processing row H1 on the MI355X (videoglamm_amd/preproc.py) vs the host pipeline (videoglamm_amd/host.py + H2D).
usage: python tools/bench_preproc.py [reps]   — C1 clip: 8 frames of 512^2 (and a 480x854 clip), Te = 8."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoglamm_amd import host, ops, preproc  # noqa: E402

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
    return e0.elapsed_time(e1) / reps * 1e3


for hw in ((512, 512), (480, 854)):
    rng = np.random.RandomState(0)
    frames = [rng.randint(0, 256, hw + (3,)).astype(np.uint8) for _ in range(8)]
    stack = np.stack(frames)
    t0 = time.time()
    cg = host.ConvGenerator_VideoGPTPlus(num_frames=8)
    ref = host.preprocess_vision([frames], conv_generator=cg, precision="fp32")
    t_host = time.time() - t0
    t0 = time.time()
    on_dev = [t[0].to(dev) for t in ref[:3]]
    torch.cuda.synchronize()
    t_h2d = time.time() - t0
    fp32_bytes = sum(t[0].numel() * 4 for t in ref[:3])
    pinned = torch.from_numpy(stack).pin_memory()
    us_up = timed(lambda: pinned.to(dev, non_blocking=True))
    x = torch.from_numpy(stack).to(dev)
    us_all = timed(lambda: preproc.preprocess_vision([x], conv_generator=cg, precision="fp32"))
    us_sam = timed(lambda: preproc.sam_preprocess(x))
    us_iv2 = timed(lambda: preproc.iv2_preprocess(x))
    us_clip = timed(lambda: preproc.clip_preprocess(x))
    out_bytes = fp32_bytes + stack.nbytes
    print(f"{hw[0]}x{hw[1]} x 8 frames: host pipeline {t_host * 1e3:.0f} ms + H2D of {fp32_bytes / 1e6:.0f} MB fp32 {t_h2d * 1e3:.1f} ms"
          f"  |  device: upload {stack.nbytes / 1e6:.1f} MB uint8 {us_up:.0f} us + kernels {us_all:.0f} us"
          f" (sam {us_sam:.0f}, iv2 {us_iv2:.0f}, clip {us_clip:.0f}) = {out_bytes / us_all / 1e3:.0f} GB/s of input+output bytes")
