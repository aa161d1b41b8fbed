"""Helpers shared by the oracle / parity tests: fixtures + the seeded weights they were made with."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, GOLD)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import seeded  # noqa: E402
import configs  # noqa: E402,F401


def fixture(name):
    z = np.load(os.path.join(GOLD, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def manifest(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def weights(manifest_name, seed, overrides=None):
    return seeded.seeded_state_dict(manifest(manifest_name), seed, overrides)


def rnd(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def sam2_cfg():
    c = configs.SAM2_MICRO
    return dict(image_size=c["image_size"], trunk=dict(c["trunk"]))
