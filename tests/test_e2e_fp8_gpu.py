"""BASELINE config C4's LLM path end to end on the HIP kernels: the façade with the fp8 prefill GEMMs (e4m3 MFMA, per-token /
per-channel scales) and fp8 decode weights vs the SAME model on the bf16 path — a 2-layer decoder of Llama-3-8B width, micro
towers / SAM2, 8 forced [SEG] objects, both SAM2 branches.  The fp8 run is teacher-forced to the bf16 run's ids (both see the
same sequence); compared: the lm_head logits of every step (relative error, argmax on clear margins), the [SEG] embeddings that
cross the LLM -> SAM2 seam, and the masks (IoU as R/eval_gcg_metrics.py:26-35)."""
import pytest
import torch

torch.set_grad_enabled(False)


def _cfg():
    from videoglamm_amd import synth
    return dict(seg_token_idx=4095, projector_depth=2,
                iv2=dict(img_size=224, patch_size=14, embed_dim=128, depth=3, num_heads=4, mlp_hidden=256),
                clip=dict(img_size=336, patch_size=14, hidden=128, mlp=256, num_layers=3, num_heads=4),
                llm=dict(synth.LLAMA3_8B, num_layers=2, vocab=4096),
                sam2=dict(image_size=256, trunk=dict(embed_dim=16, num_heads=1, stages=[1, 2, 3, 1], global_att_blocks=[4, 5],
                                                     window_spec=[8, 4, 8, 4], window_pos_embed_bkg_spatial_size=[7, 7])))


@pytest.mark.gpu
@pytest.mark.parametrize("branch", [False, True])
def test_e2e_fp8_llm_path_vs_bf16(cuda, branch):
    from videoglamm_amd import synth
    from videoglamm_amd.model import VideoGLaMMForCausalLM

    cfg = _cfg()
    G, te, T, hw = 28, 4, 3, (48, 64)
    forced = {2 + 3 * i: cfg["seg_token_idx"] for i in range(8)}          # eight [SEG] objects (C4)
    sd = synth.device_state_dict(synth.manifest(cfg), cuda, torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    images, context = torch.randn(te, 3, 224, 224, generator=g).to(cuda), torch.randn(te, 3, 336, 336, generator=g).to(cuda)
    sam = torch.randn(T, 3, 256, 256, generator=g).to(cuda)
    ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((te,), -200), torch.randint(3, 4000, (20,), generator=g)])[None]
    S = 208 * te + ids.shape[1] - te

    def run(c):
        m = synth.install_forced_tokens(VideoGLaMMForCausalLM(sd, c, torch_dtype=torch.bfloat16, device=cuda))
        m.capture = {}
        out_ids, segs = m.inference([images], [context], [sam], ids, [(256, 256)], [hw], max_new_tokens=G, use_sam2_video_branch=branch)
        dec = m.P._decoder
        n = out_ids.shape[1] - ids.shape[1]
        rows = dec.hid_all[S - 1:S - 1 + n].float()                        # the final-norm states that emitted the n tokens
        logits = rows @ m.P.t("lm_head.weight").float().t()                # (checker arithmetic in torch, outside the product path)
        return out_ids[0].tolist(), m.capture, logits

    ids16, cap16, lg16 = run(dict(cfg, forced_tokens=forced))
    gen = ids16[ids.shape[1]:]
    assert sum(t == cfg["seg_token_idx"] for t in gen) == 8 and cap16["emb"].shape == (8, 256)
    c8 = dict(cfg, forced_tokens={i: t for i, t in enumerate(gen)})
    c8["llm"] = dict(cfg["llm"], prefill_gemm="fp8", decode_weights="fp8")
    ids8, cap8, lg8 = run(c8)
    assert ids8 == ids16
    rel = ((lg16 - lg8).norm(dim=1) / lg16.norm(dim=1)).max().item()
    cos = torch.nn.functional.cosine_similarity(lg16, lg8).min().item()
    top2 = lg16.topk(2, dim=1).values
    margin, noise = top2[:, 0] - top2[:, 1], (lg16 - lg8).pow(2).mean(dim=1).sqrt()
    clear = margin > 4 * noise
    agree = (lg16.argmax(1) == lg8.argmax(1))
    free = [i for i in range(len(gen)) if i not in forced]
    model_agree = sum(cap16["argmax"][i] == cap8["argmax"][i] for i in free) / len(free)
    ecos = torch.nn.functional.cosine_similarity(cap16["emb"].float(), cap8["emb"].float()).min().item()
    a, b = cap16["logits"] > 0, cap8["logits"] > 0
    iou = ((a & b).sum(dim=(0, 2, 3)).double() / (a | b).sum(dim=(0, 2, 3)).double().clamp_min(1))
    print(f"fp8 LLM path vs bf16 ({'video' if branch else 'framewise'} branch): logits rel err {rel:.3f}, cosine {cos:.4f}, clear-margin steps "
          f"{int(clear.sum())}/{len(gen)}, argmax agreement {model_agree:.2f} of the free steps, [SEG] embedding cosine {ecos:.4f}, "
          f"mask IoU mean {iou.mean():.4f} min {iou.min():.4f}")
    assert rel < 0.15 and cos > 0.99, (rel, cos)                 # e4m3: ~3.6 % rms per element, it does not average out of a dot product
    assert bool(agree[clear].all()), "an argmax with a clear bf16 margin flipped on the fp8 path"
    assert ecos > 0.99 and iou.mean() > 0.97 and iou.min() > 0.93, (ecos, iou.tolist())
