"""the mask decoder at C4's clip size (512 instances) alone in a process, for rocprofv3 --kernel-trace --stats: which of its kernels holds the time"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from videoglamm_amd import sam2 as S2, synth  # noqa: E402
from videoglamm_amd.params import Params  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = synth.SAM2_L
sd = synth.device_state_dict(synth.sam2_manifest(cfg), dev, torch.bfloat16)


class M:
    sam2 = S2.SAM2(Params(sd, dev, torch.bfloat16), "", cfg)


print(bench.mask_decoder_record(M, dev, reps=5))
