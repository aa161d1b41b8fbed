"""ORACLE support (test infrastructure): deterministic synthetic weights keyed by parameter NAME.

Both tests/golden/make_golden.py (which loads them into the reference's nn.Modules) and the tests /
bench (which feed the same dict to the oracle and to the HIP path) regenerate identical tensors from
(name, shape, seed) with torch's CPU generator, so no weight file is ever committed.
"""
import math
import zlib

import torch


def seeded_tensor(name, shape, seed=0):
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
    shape = tuple(shape)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))
    if len(shape) == 1 and name.endswith(".weight"):
        return 1.0 + 0.05 * torch.randn(shape, generator=g)  # norm gains / per-channel scales
    return 0.1 * torch.randn(shape, generator=g)


def seeded_state_dict(manifest, seed=0, overrides=None):
    """manifest: {name: shape}.  overrides: {name: tensor | callable(tensor)->tensor}."""
    sd = {n: seeded_tensor(n, s, seed) for n, s in manifest.items()}
    for n, v in (overrides or {}).items():
        sd[n] = v(sd[n]) if callable(v) else v
    return sd


def sam2_overrides(prefix=""):
    """Random-init SAM2 predicts 'no object' (object_score_logits < 0 -> all masks = -1024):
    bias the object-score head positive so the mask path is exercised."""
    return {prefix + "sam_mask_decoder.pred_obj_score_head.layers.2.bias": lambda t: t * 0 + 4.0}


def sam2_noobj_overrides(c, k, prefix=""):
    """score head rescaled so that objects disappear and reappear over a clip: the random-init head answers ~ -0.5 with a spread
    of 0.01 between frames / objects; score' = k * (score + c) puts the values on both sides of 0 with a usable margin
    (tests/golden/make_golden.py:gen_sam2_noobj picks c on the reference and stores it in the fixture)."""
    h = prefix + "sam_mask_decoder.pred_obj_score_head.layers.2."
    return {h + "weight": lambda t: t * k, h + "bias": lambda t: t * 0 + c * k}
