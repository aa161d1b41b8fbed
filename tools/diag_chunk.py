"""Diagnostic (GPU box): frame-batched vs frame-serial SAM2 (Hiera + mask decoder) in both dtypes — per-item arithmetic must not depend on the batch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videoglamm_amd import ops, synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402
from videoglamm_amd.sam2 import SAM2  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = synth.SAM2_L
sd16 = synth.device_state_dict(synth.sam2_manifest(cfg), dev, torch.bfloat16)
g = torch.Generator().manual_seed(7)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img = torch.randn(T, 3, 1024, 1024, generator=g).to(dev)
text = (torch.randn(1, 256, generator=g) * 0.5).to(dev)
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    sd = sd16 if dt == torch.bfloat16 else {k: v.float() for k, v in sd16.items()}
    m = SAM2(Params(sd, dev, dt), "", cfg)
    m.frame_chunk = T
    fb = m.hiera_frames(img)
    lb, _ = m.framewise_branch(img, text, (256, 256), frame_feats=fb)
    m.frame_chunk = 1
    fs = m.hiera_frames(img)
    ls, _ = m.framewise_branch(img, text, (256, 256), frame_feats=fs)
    lx, _ = m.framewise_branch(img, text, (256, 256), frame_feats=fb)      # batched Hiera features, serial decoder
    for t in range(T):
        d = [float((fb[t][lv].float() - fs[t][lv].float()).abs().max()) for lv in range(3)]
        print(f"{name} frame {t}: fpn max|batched - serial| {d}  logits max diff (all batched vs all serial) {float((lb[t] - ls[t]).abs().max()):.3e}"
              f"  (decoder only: {float((lb[t] - lx[t]).abs().max()):.3e})  |logit| max {float(ls[t].abs().max()):.2f}")
    del m
