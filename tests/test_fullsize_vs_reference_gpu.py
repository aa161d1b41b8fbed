"""The HIP path (fp32 parity mode) against the REFERENCE at full architecture size (-m gpu): tests/golden/fullsize_*.npz hold outputs of the
reference's own modules (sam2_hiera_l.yaml unmodified, InternVideo2-1B, HF CLIP-L/336, HF LlamaModel at Llama-3-8B width) on the name-seeded
weights and seeded inputs regenerated here (tests/golden/make_golden_fullsize.py).  The north star's bar: mask logits within
1e-3 * max(1, |logit|) in fp32 — on the framewise branch and, FREE-RUNNING, on the video branch (both sides round their own memories to
bf16 and attend to them); towers / LLM rows at the same bar."""
import pytest
import torch

import _golden as G
from fullsize_keys import sub
from oracle import seeded
from videoglamm_amd import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def bf16w(sd):
    return {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in sd.items()}


def close(got, want, tol, what):
    got = got.float().cpu()
    err = ((got - want).abs() / want.abs().clamp_min(1.0)).max()
    print(f"  {what}: max err / max(1,|ref|) {float(err):.2e} (|ref| max {float(want.abs().max()):.2f})")
    assert got.shape == want.shape and torch.isfinite(got).all() and float(err) <= tol, (what, float(err))


@pytest.fixture(scope="module")
def sam2(cuda):
    from videoglamm_amd.params import Params
    from videoglamm_amd.sam2 import SAM2
    sd = bf16w(seeded.seeded_state_dict(synth.sam2_manifest(synth.SAM2_L), 2, seeded.sam2_overrides()))
    return SAM2(Params(sd, cuda, torch.float32), "", synth.SAM2_L)


def test_sam2_large_framewise_frame_fp32_vs_reference(cuda, sam2):
    fx = G.fixture("fullsize_sam2.npz")
    g = torch.Generator().manual_seed(9)
    img = torch.randn(1, 3, 1024, 1024, generator=g)
    text = torch.randn(2, 256, generator=g) * 0.5
    logits, low = sam2.framewise_branch(img.to(cuda), text.to(cuda), (480, 640))
    close(sub("low", low.float().cpu().reshape(2, 1, 256, 256)), fx["fw_low"], 1e-3, "SAM2-L framewise low-res logits vs the reference")
    close(sub("logits", logits[0]), fx["fw_logits"], 1e-3, "SAM2-L framewise logits at 480x640 vs the reference")
    frac = (logits[0].float().cpu() > 0).float().mean(dim=(1, 2))
    torch.testing.assert_close(frac, fx["fw_mask_frac"], rtol=0, atol=2e-4)


def test_sam2_large_video_clip_fp32_free_running_vs_reference(cuda, sam2):
    """T = 3, N = 2, free-running (own bf16-rounded memories) against SAM2VideoPredictor's outputs on the same weights and frames"""
    fx = G.fixture("fullsize_sam2.npz")
    g = torch.Generator().manual_seed(19)
    images = torch.randn(9, 3, 1024, 1024, generator=g)[:3].contiguous()
    text = torch.randn(2, 256, generator=g) * 0.5
    trace = {}
    vid = sam2.video_branch(images.to(cuda), text.to(cuda), (480, 640), trace)
    low = trace["low_res"].float().cpu()
    for t in range(3):
        close(sub("low", low[t]), fx["vid_low"][t], 1e-3, f"low-res logits frame {t} vs the reference (free-running)")
    close(trace["obj_ptr"], fx["vid_obj_ptr"], 1e-3, "object pointers")
    sc = torch.stack([trace["frame0_obj_logits"].view(-1)] + [trace[f"obj_logits_{t}"].view(-1) for t in (1, 2)])
    close(sc, fx["vid_obj_scores"], 1e-3, "object scores")
    close(sub("logits", vid), fx["vid_logits"], 1e-3, "mask logits at 480x640")
    for t in (0, 1):
        got = trace["maskmem"][t].float().cpu()                      # [N, 4096, 64] token-major
        got = sub("maskmem", got.permute(0, 2, 1).reshape(2, 64, 64, 64))
        want = fx[f"vid_maskmem{t}"]
        far = float((~torch.isclose(got, want, rtol=1e-2, atol=2e-3)).float().mean())
        print(f"  memory of frame {t}: fraction more than one bf16 step from the reference's {far:.1e}")
        assert far < (5e-3 if t == 0 else 2e-4), far


def test_towers_and_llm_fp32_vs_reference(cuda):
    from videoglamm_amd.params import Params
    from videoglamm_amd.vlm import LlamaDecoder, VisionTowers
    fx = G.fixture("fullsize_vlm.npz")
    full = synth.vlm_manifest(synth.videoglamm_llama3_8b())
    # InternVideo2-1B, one chunk
    c = synth.IV2_1B
    p = "model.vision_tower.vision_encoder."
    sd = bf16w(seeded.seeded_state_dict({k: v for k, v in full.items() if k.startswith(p)}, 7))
    t = VisionTowers(Params(sd, cuda, torch.float32), dict(iv2=dict(depth=c["depth"], num_heads=c["num_heads"], patch_size=c["patch_size"])))
    vid = torch.randn(1, 4, 3, 224, 224, generator=torch.Generator().manual_seed(51))
    out = t.iv2(vid.to(cuda)).float().cpu()                         # [1, 1024, 1408]: the CLS row is dropped
    want = fx["iv2_out"]                                            # every 16th token INCLUDING the CLS row 0: rows 16, 32, ... = out rows 15, 31, ...
    close(out[:, 15::16], want[:, 1:], 1e-3, "InternVideo2-1B tokens vs the reference")
    del t
    # CLIP-L/336, two frames
    c = synth.CLIP_L_336
    p = "model.image_vision_tower.vision_tower.vision_model."
    sd = bf16w(seeded.seeded_state_dict({k: v for k, v in full.items() if k.startswith(p)}, 8))
    t = VisionTowers(Params(sd, cuda, torch.float32), dict(clip=dict(num_layers=c["num_layers"], num_heads=c["num_heads"], patch_size=c["patch_size"])))
    img = torch.randn(2, 3, 336, 336, generator=torch.Generator().manual_seed(52))
    close(sub("tokens16", t.clip(img.to(cuda))), fx["clip_out"], 1e-3, "CLIP-L/336 hidden_states[-2] vs the reference")
    del t
    # Llama-3-8B width, 2 layers: prefill of 248 rows + 8 rows through the decode kernels
    c = dict(synth.LLAMA3_8B, num_layers=2, vocab=8192)
    man = {k: v for k, v in synth.vlm_manifest(dict(synth.videoglamm_llama3_8b(), llm=c)).items() if k.startswith(("model.layers.", "model.norm"))}
    sd = bf16w(seeded.seeded_state_dict(man, 5))
    x = (torch.randn(256, c["hidden"], generator=torch.Generator().manual_seed(53)) * 0.5).to(torch.bfloat16).float().to(cuda)
    dec = LlamaDecoder(Params(sd, cuda, torch.float32), c, 1024, use_graph=False)
    close(dec.forward(x)[::8], fx["llama_out"], 1e-3, "Llama-3-8B-width prefill vs HF LlamaModel")
    dec = LlamaDecoder(Params(sd, cuda, torch.float32), c, 1024, use_graph=False)
    rows = [dec.forward(x[:248])] + [dec.forward(x[i:i + 1]) for i in range(248, 256)]
    close(torch.cat(rows)[::8], fx["llama_out"], 1e-3, "prefill + 8 cached rows vs HF LlamaModel")
