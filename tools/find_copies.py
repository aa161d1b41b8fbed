"""Debug aid: list the call sites inside videoglamm_amd that make torch materialise a copy (.contiguous() on a strided
view, torch.cat, .to(dtype)) during one Hiera forward — every such copy is an extra HBM round trip outside the kernels."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoglamm_amd import synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402
from videoglamm_amd.sam2 import SAM2  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.SAM2_L
sd = synth.device_state_dict(synth.sam2_manifest(cfg, "model.visual_model."), dev, torch.bfloat16)
sam = SAM2(Params(sd, dev, torch.bfloat16), "model.visual_model.", cfg)
img = torch.randn(2, 3, 1024, 1024, device=dev)
sam.forward_image(img)
sites = collections.Counter()
orig = torch.Tensor.contiguous


def spy(self, *a, **k):
    if not self.is_contiguous():
        fr = [f for f in traceback.extract_stack() if "videoglamm_amd" in f.filename]
        sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1]) + f"  {tuple(self.shape)}"] += 1
    return orig(self, *a, **k)


torch.Tensor.contiguous = spy
sam.forward_image(img)
torch.Tensor.contiguous = orig
for k, v in sites.most_common(30):
    print(v, k)
