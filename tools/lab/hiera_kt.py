"""Hiera-L + FPN alone (32 frames in two 16-frame chunks, as the C2 clip runs it), for rocprofv3 --kernel-trace --stats: which kernels hold the 101 ms"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from videoglamm_amd import sam2 as S2, synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = synth.SAM2_L
sd = synth.device_state_dict(synth.sam2_manifest(cfg), dev, torch.bfloat16)
m = S2.SAM2(Params(sd, dev, torch.bfloat16), "", cfg)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
img = torch.randn(T, 3, 1024, 1024, device=dev, dtype=torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(reps):
    if i == 1:
        e0.record()
    f = m.hiera_frames(img, None)
    del f
e1.record()
torch.cuda.synchronize()
print(f"hiera+fpn {T} frames: {e0.elapsed_time(e1) / (reps - 1):.2f} ms per clip")
