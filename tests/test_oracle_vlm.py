"""The LLM-side oracle (oracle/vlm.py) vs outputs of the reference's InternVideo2 and of HF
CLIPVisionModel / LlamaModel (the third-party arithmetic the reference calls), tests/golden/vlm_tiny.npz."""
import torch

import _golden as G
from oracle import vlm as O

torch.set_grad_enabled(False)
TOL = dict(rtol=1e-4, atol=1e-4)


def setup_module(m):
    m.fx = G.fixture("vlm_tiny.npz")


def test_internvideo2():
    c = G.configs.IV2_TINY
    sd = G.weights("iv2_tiny_manifest.json", 2)
    cfg = dict(depth=c["depth"], num_heads=c["num_heads"], patch_size=c["patch_size"])
    out = O.iv2_forward(sd, "", cfg, G.rnd((2, 4, 3, c["img_size"], c["img_size"]), 31))
    torch.testing.assert_close(out, fx["iv2_out"], **TOL)


def test_clip():
    c = G.configs.CLIP_TINY
    sd = G.weights("clip_tiny_manifest.json", 3)
    cfg = dict(num_heads=c["num_heads"], num_layers=c["num_layers"], patch_size=c["patch_size"])
    out = O.clip_forward(sd, "", cfg, G.rnd((3, 3, c["img_size"], c["img_size"]), 32))
    torch.testing.assert_close(out, fx["clip_out"], **TOL)


def test_llama():
    c = G.configs.LLAMA_TINY
    sd = G.weights("llama_tiny_manifest.json", 4)
    out = O.llama_forward(sd, "", c | dict(num_layers=c["num_layers"]), G.rnd((1, 45, c["hidden"]), 33)[0])
    torch.testing.assert_close(out, fx["llama_out"], **TOL)


def test_phi3():
    """fused qkv_proj / gate_up_proj decoder (the released checkpoint's LLM) vs HF Phi3Model, tests/golden/phi3_tiny.npz"""
    c = G.configs.PHI3_TINY
    sd = G.weights("phi3_tiny_manifest.json", 5)
    out = O.llama_forward(sd, "", c, G.rnd((1, 45, c["hidden"]), 34)[0])
    torch.testing.assert_close(out, G.fixture("phi3_tiny.npz")["phi3_out"], **TOL)


def test_phi3_sliding_window():
    """the window mask (position i sees [i - w, i]) on a sequence that crosses it, vs HF Phi3Model with the same visible keys"""
    c = G.configs.PHI3_TINY_WIN
    sd = G.weights("phi3_tiny_manifest.json", 5)
    out = O.llama_forward(sd, "", c, G.rnd((1, 45, c["hidden"]), 34)[0])
    torch.testing.assert_close(out, G.fixture("phi3_tiny.npz")["phi3_win_out"], **TOL)
    full = O.llama_forward(sd, "", G.configs.PHI3_TINY, G.rnd((1, 45, c["hidden"]), 34)[0])
    assert (out[:12] - full[:12]).abs().max() < 1e-5 and (out[12:] - full[12:]).abs().max() > 1e-4     # rows inside the window are untouched
