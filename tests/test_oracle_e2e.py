"""End-to-end oracle vs the reference's own VideoGLaMM inference() (both SAM2 branches) on the tiny
Llama composition — tests/golden/e2e_tiny.npz (token ids must be identical, masks identical)."""
import numpy as np
import torch

import _golden as G
from oracle import pipeline, seeded

torch.set_grad_enabled(False)


def e2e_setup():
    fx = G.fixture("e2e_tiny.npz")
    E = G.configs.E2E
    sd = G.weights("e2e_manifest.json", 5, seeded.sam2_overrides("model.visual_model."))
    cfg = dict(seg_token_idx=int(fx["seg_token_idx"]),
               iv2=dict(depth=E["iv2"]["depth"], num_heads=E["iv2"]["num_heads"], patch_size=E["iv2"]["patch_size"]),
               clip=dict(num_heads=E["clip"]["num_heads"], num_layers=E["clip"]["num_layers"], patch_size=E["clip"]["patch_size"]),
               llm=dict(E["llm"]),
               sam2=dict(image_size=G.configs.SAM2_E2E["image_size"], trunk=dict(G.configs.SAM2_E2E["trunk"])))
    te, S, T = E["te"], G.configs.SAM2_E2E["image_size"], E["t_sam"]
    inputs = dict(images=G.rnd((te, 3, 224, 224), 41), context_images=G.rnd((te, 3, 336, 336), 42),
                  images_for_sam=G.rnd((T, 3, S, S), 43), input_ids=fx["input_ids"].long(), original_size=(40, 56),
                  max_new_tokens=E["max_new_tokens"])
    return fx, sd, cfg, inputs


def _check(branch):
    fx, sd, cfg, inputs = e2e_setup()
    key = "video" if branch else "framewise"
    ids, seg = pipeline.inference(sd, cfg, use_sam2_video_branch=branch, **inputs)
    assert ids.tolist() == fx[f"{key}_output_ids"].long().tolist()
    ref = fx[f"{key}_masks"].numpy() > 0.5
    got = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
    assert got.shape == ref.shape
    inter, union = (got & ref).sum(), (got | ref).sum()
    assert inter / union > 0.9995, inter / union  # IoU as R/eval_gcg_metrics.py:26-35


def test_e2e_framewise():
    _check(False)


def test_e2e_video_branch():
    _check(True)


def image_setup():
    fx = G.fixture("e2e_image.npz")
    _, sd, cfg, _ = e2e_setup()
    cfg = dict(cfg, seg_token_idx=int(fx["seg_token_idx"]))
    S = G.configs.SAM2_E2E["image_size"]
    inputs = dict(images=G.rnd((1, 3, 336, 336), 51), context_images=None, images_for_sam=G.rnd((1, 3, S, S), 52),
                  input_ids=fx["input_ids"].long(), original_size=(40, 56), max_new_tokens=G.configs.E2E["max_new_tokens"])
    return fx, sd, cfg, inputs


def test_e2e_image_prompt():
    """single-image prompt (context_images=None): CLIP -> image_mm_projector without pooling, one SAM frame"""
    fx, sd, cfg, inputs = image_setup()
    ids, seg = pipeline.inference(sd, cfg, use_sam2_video_branch=False, **inputs)
    assert ids.tolist() == fx["output_ids"].long().tolist()
    ref = fx["masks"].numpy() > 0.5
    got = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
    assert got.shape == ref.shape and (got & ref).sum() / (got | ref).sum() > 0.9995
