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


SEAM_TOL = dict(rtol=1e-3, atol=1e-3)     # mask logits within 1e-3 (fp32), through the LLM -> text_hidden_fcs -> SAM2 seam


def check_seam(fx, key, emb, logits):
    """[SEG] embeddings [N,256] and the mask logits BEFORE the `> 0` vs what the reference's own inference() handed across the seam
    (make_golden.py:_capture_seam)."""
    torch.testing.assert_close(emb.float().cpu(), fx[f"{key}_seg_emb"], **SEAM_TOL)
    torch.testing.assert_close(logits.float().cpu(), fx[f"{key}_logits"], **SEAM_TOL)


def _check(branch, eight=False):
    fx, sd, cfg, inputs = e2e_setup()
    key = ("video" if branch else "framewise") + ("8" if eight else "")
    if eight:       # C4's shape: eight [SEG] ids in the prompt + the emitted ones = 14 objects
        inputs = dict(inputs, input_ids=fx["input_ids8"].long())
    cap = {}
    ids, seg = pipeline.inference(sd, cfg, use_sam2_video_branch=branch, capture=cap, **inputs)
    assert ids.tolist() == fx[f"{key}_output_ids"].long().tolist()
    ref = fx[f"{key}_masks"].numpy() > 0.5
    got = np.stack([np.stack([seg[t][k] for k in sorted(seg[t])]) for t in sorted(seg)])
    assert got.shape == ref.shape and (not eight or ref.shape[1] >= 8)
    inter, union = (got & ref).sum(), (got | ref).sum()
    assert inter / union > 0.9995, inter / union  # IoU as R/eval_gcg_metrics.py:26-35
    check_seam(fx, key, cap["emb"], cap["logits"])


def test_e2e_framewise():
    _check(False)


def test_e2e_video_branch():
    _check(True)


def test_e2e_framewise_8_objects():
    _check(False, True)


def test_e2e_video_branch_8_objects():
    _check(True, True)


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
