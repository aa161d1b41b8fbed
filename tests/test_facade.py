"""The drop-in surface of row H3 beyond inference(): model_forward(inference=True) / forward(**kwargs)
(R/model/VideoGLaMM.py:325-508, 897-900), the two "no [SEG] was emitted" behaviours (:732 raises, :840-842 returns an
empty dict), EOS stopping, and the refusals of what is out of scope — on the CPU twins and on the HIP kernels."""
import numpy as np
import pytest
import torch

torch.set_grad_enabled(False)


def build(device, **over):
    from test_oracle_e2e import e2e_setup
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    fx, sd, cfg, inp = e2e_setup()
    cfg = dict(cfg, **over)
    return fx, inp, VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=device)


def check_model_forward(device):
    """teacher-forcing the ids the model emitted reproduces the hidden states, so model_forward(inference=True) must give
    the framewise logits whose sign is the reference's masks (tests/golden/e2e_tiny.npz)."""
    fx, inp, m = build(device)
    out_ids = fx["framewise_output_ids"].long()[None]
    hw = inp["original_size"]
    kw = dict(images_for_sam=[inp["images_for_sam"]], images=[inp["images"]], context_images=[inp["context_images"]], input_ids=out_ids,
              label_list=[torch.zeros(hw)], masks_list=["gt"], inference=True)
    out = m.forward(**kw)                                         # forward(**kwargs) -> model_forward
    assert out["gt_masks"] == ["gt"] and len(out["pred_masks"]) == 1
    logits = torch.stack([x.float().cpu() for x in out["pred_masks"][0]])          # [T, N, H, W]
    ref = fx["framewise_masks"].numpy() > 0.5
    got = logits.numpy() > 0
    assert got.shape == ref.shape and (got & ref).sum() / (got | ref).sum() > 0.999
    # a prompt without any [SEG]: one empty [0,H,W] tensor per frame (VideoGLaMM.py:436-446 pads nothing at batch 1)
    none = m.model_forward(**dict(kw, input_ids=inp["input_ids"][None]))
    assert len(none["pred_masks"][0]) == inp["images_for_sam"].shape[0] and all(tuple(x.shape) == (0,) + tuple(hw) for x in none["pred_masks"][0])
    with pytest.raises(NotImplementedError):
        m.model_forward(**dict(kw, inference=False))            # training losses are out of scope
    with pytest.raises(NotImplementedError):
        m.forward(past_key_values=None, input_ids=out_ids)        # the bare LM forward is not part of this path


def check_no_seg_and_eos(device):
    fx, inp, m = build(device, seg_token_idx=10 ** 6)             # an id the LLM can never emit
    args = ([inp["images"]], [inp["context_images"]], [inp["images_for_sam"]], inp["input_ids"][None], [(1024, 1024)], [inp["original_size"]])
    with pytest.raises(AttributeError):                           # the reference dereferences a tuple here (VideoGLaMM.py:732)
        m.inference(*args, max_new_tokens=3)
    ids, segs = m.inference(*args, max_new_tokens=3, use_sam2_video_branch=True)
    assert segs == [{}] and ids.shape[1] == inp["input_ids"].numel() + 3             # VideoGLaMM.py:840-842
    first = int(ids[0, inp["input_ids"].numel()])
    fx, inp, m = build(device, seg_token_idx=10 ** 6, eos_token_id=first)
    ids, segs = m.inference(*args, max_new_tokens=8, use_sam2_video_branch=True)
    assert ids.shape[1] == inp["input_ids"].numel() + 1 and int(ids[0, -1]) == first   # stops on EOS, EOS included
    with pytest.raises(AssertionError):                           # batch size is 1 (VideoGLaMM.py:252-253)
        m.inference([inp["images"]] * 2, [inp["context_images"]] * 2, [inp["images_for_sam"]] * 2, inp["input_ids"][None].repeat(2, 1),
                    [(1024, 1024)] * 2, [inp["original_size"]] * 2)


def test_model_forward_cpu(cpu_ops, monkeypatch):
    from videoglamm_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    check_model_forward(torch.device("cpu"))


def test_no_seg_and_eos_cpu(cpu_ops, monkeypatch):
    from videoglamm_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    check_no_seg_and_eos(torch.device("cpu"))


@pytest.mark.gpu
def test_model_forward_hip_fp32(cuda):
    check_model_forward(cuda)


@pytest.mark.gpu
def test_no_seg_and_eos_hip_fp32(cuda):
    check_no_seg_and_eos(cuda)
