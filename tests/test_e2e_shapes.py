"""End-to-end against the ORACLE on ragged shapes the golden fixtures do not cover: a one-frame clip, odd output sizes,
one / three [SEG] objects, both SAM2 branches, Te = 4 and 8.  Random weights never emit [SEG], so the product forces it at
given decode steps (after the full lm_head + argmax) and the oracle — which restates the reference and has no such knob —
replays the emitted ids by teacher forcing (smoke_check.py:_oracle, the same scheme as __graft_entry__.smoke())."""
import numpy as np
import pytest
import torch

torch.set_grad_enabled(False)

CFG = dict(seg_token_idx=77, projector_depth=2,
           iv2=dict(img_size=224, patch_size=14, embed_dim=64, depth=2, num_heads=4, mlp_hidden=128),
           clip=dict(img_size=336, patch_size=14, hidden=64, mlp=128, num_layers=3, num_heads=4),
           llm=dict(vocab=96, hidden=64, ffn=176, num_layers=2, num_heads=4, num_kv_heads=2, rms_eps=1e-5, rope_theta=10000.0),
           sam2=dict(image_size=256, trunk=dict(embed_dim=16, num_heads=1, stages=[1, 2, 3, 1], global_att_blocks=[4, 5],
                                                window_spec=[8, 4, 8, 4], window_pos_embed_bkg_spatial_size=[7, 7])))
CASES = [  # T frames, Te encoder frames, output (H, W), forced [SEG] steps, video branch
    (1, 4, (37, 53), {0: 77}, False),
    (1, 4, (37, 53), {2: 77}, True),
    (2, 8, (64, 48), {1: 77, 2: 77, 4: 77}, False),
    (3, 4, (33, 61), {0: 77, 3: 77}, True),
]


def check(device, case):
    from oracle import pipeline, seeded
    from videoglamm_amd import synth
    from videoglamm_amd.model import VideoGLaMMForCausalLM
    from smoke_check import _oracle

    T, te, hw, forced, branch = case
    cfg = dict(CFG, forced_tokens=forced)
    sd = seeded.seeded_state_dict(synth.manifest(cfg), 3, seeded.sam2_overrides("model.visual_model."))
    g = torch.Generator().manual_seed(7 + T + te)
    S = cfg["sam2"]["image_size"]
    images, context = torch.randn(te, 3, 224, 224, generator=g), torch.randn(te, 3, 336, 336, generator=g)
    sam = torch.randn(T, 3, S, S, generator=g)
    ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((te,), -200), torch.randint(3, 76, (5,), generator=g)])
    m = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device=device))
    out_ids, segs = m.inference([images.to(device)], [context.to(device)], [sam.to(device)], ids[None], [(S, S)], [hw], max_new_tokens=6,
                                use_sam2_video_branch=branch)
    ref_ids, ref_logits = _oracle(pipeline, sd, cfg, images, context, sam, out_ids[0], ids.numel(), hw, branch)
    assert out_ids[0].tolist() == ref_ids.tolist()
    got = np.stack([np.stack([segs[0][t][k] for k in sorted(segs[0][t])]) for t in sorted(segs[0])])
    ref = (ref_logits > 0).numpy()
    assert got.shape == ref.shape == (T, len(forced)) + hw
    agree = (got == ref).mean()
    assert agree > 0.999, agree


@pytest.mark.parametrize("case", CASES)
def test_shapes_cpu(cpu_ops, monkeypatch, case):
    from videoglamm_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    check(torch.device("cpu"), case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_shapes_hip_fp32(cuda, case):
    check(cuda, case)
