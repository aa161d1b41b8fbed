"""TEST INFRASTRUCTURE (repo root, outside the product package: it imports oracle/).  __graft_entry__.smoke(): one small invocation of the hot path on cuda:0 (HIP kernels, fp32 parity mode)
checked against the oracle (CPU restatement of the reference) on the same seeded weights and inputs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))


def run():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import pipeline, seeded
    from videoglamm_amd import _lib, synth
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    lib = _lib.load()
    assert lib.vg_init(0) > 0, lib.vg_last_error()
    torch.set_grad_enabled(False)
    cfg = dict(seg_token_idx=77, projector_depth=2,
               iv2=dict(img_size=224, patch_size=14, embed_dim=64, depth=3, num_heads=4, mlp_hidden=128),
               clip=dict(img_size=336, patch_size=14, hidden=64, mlp=128, num_layers=3, num_heads=4),
               llm=dict(vocab=96, hidden=64, ffn=176, num_layers=2, num_heads=4, num_kv_heads=2, rms_eps=1e-5, rope_theta=10000.0),
               sam2=dict(image_size=256, trunk=dict(embed_dim=16, num_heads=1, stages=[1, 2, 3, 1], global_att_blocks=[4, 5],
                                                    window_spec=[8, 4, 8, 4], window_pos_embed_bkg_spatial_size=[7, 7])),
               forced_tokens={1: 77, 3: 77})
    sd = seeded.seeded_state_dict(synth.manifest(cfg), 3, seeded.sam2_overrides("model.visual_model."))
    g = torch.Generator().manual_seed(5)
    te, T, S, hw = 4, 3, 256, (40, 56)
    images, context = torch.randn(te, 3, 224, 224, generator=g), torch.randn(te, 3, 336, 336, generator=g)
    sam = torch.randn(T, 3, S, S, generator=g)
    ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((te,), -200), torch.randint(3, 76, (6,), generator=g)])
    m = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, cfg, torch_dtype=torch.float32, device="cuda:0"))
    for branch in (False, True):
        out_ids, segs = m.inference([images.cuda()], [context.cuda()], [sam.cuda()], ids[None], [(S, S)], [hw], max_new_tokens=5,
                                    use_sam2_video_branch=branch)
        # oracle with the same forced tokens: feed the emitted ids back as a teacher-forced prompt
        ocfg = dict(cfg)
        ref_ids, ref_logits = _oracle(pipeline, sd, ocfg, images, context, sam, out_ids[0], ids.numel(), hw, branch)
        assert out_ids[0].tolist() == ref_ids.tolist(), (out_ids[0].tolist(), ref_ids.tolist())
        got = torch.stack([torch.stack([torch.from_numpy(segs[0][t][k]) for k in sorted(segs[0][t])]) for t in sorted(segs[0])])
        ref = ref_logits > 0
        agree = (got == ref).float().mean().item()
        assert agree > 0.999, f"mask agreement {agree}"
        print(f"smoke[{'video' if branch else 'framewise'}]: ids {out_ids[0].tolist()[-5:]} masks {tuple(got.shape)} agree={agree:.5f}")
    print("smoke ok")


def _oracle(pipeline, sd, cfg, images, context, sam, emitted_ids, n_prompt, hw, branch):
    """Oracle run: greedy ids are checked where they were free, and the forced steps are replayed by teacher
    forcing (the oracle has no forced_tokens knob — it restates the reference)."""
    from oracle import sam2 as osam, vlm as ovlm

    visual = ovlm.encode_visual(sd, cfg, images, context)
    ids = emitted_ids[:n_prompt].clone()
    forced = cfg.get("forced_tokens") or {}
    hidden = None
    for step in range(emitted_ids.numel() - n_prompt):
        x, _ = ovlm.splice(sd, "model.", ids, visual, cfg["seg_token_idx"])
        hidden = ovlm.llama_forward(sd, "model.", cfg["llm"], x)
        nxt = int(torch.argmax(torch.nn.functional.linear(hidden[-1], sd["lm_head.weight"])))
        nxt = forced.get(step, nxt)
        ids = torch.cat([ids, torch.tensor([nxt])])
    added = hidden.shape[0] - (ids.shape[0] - 1)
    seg_mask = torch.cat([torch.zeros(added, dtype=torch.bool), ids[1:] == cfg["seg_token_idx"]])
    fc = "model.text_hidden_fcs.0."
    emb = ovlm.lin(sd, fc + "2", torch.relu(ovlm.lin(sd, fc + "0", hidden)))[seg_mask]
    p = "model.visual_model."
    if branch:
        logits, _ = osam.video_branch(sd, p, cfg["sam2"], sam, emb, hw)
        logits = torch.stack(logits)[:, :, 0]
    else:
        logits, _ = osam.framewise_branch(sd, p, cfg["sam2"], sam, emb, hw)
        logits = torch.stack(logits)
    return ids, logits


if __name__ == "__main__":
    run()
