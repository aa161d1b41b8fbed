"""Golden vectors at FULL ARCHITECTURE SIZE.  RUN ONLY IN THE BUILD CONTAINER (needs /root/reference):

    python tests/golden/make_golden_fullsize.py [sam2|iv2|clip|llama|vlm|all]

The micro fixtures (make_golden.py) pin the arithmetic of every stage to the reference at embed_dim 16 / one head; the shape-dependent
code — head_dim 72 / 88, the 16-token window of Hiera stage 3, the 14x14 -> 7x7 background position embedding interpolated to 256x256
tokens, q-pooling at real strides, the 64x64-token memory attention with 4096-row memories — only runs at the real dimensions.  Here the
REFERENCE'S OWN modules are instantiated from its unmodified configuration (R/model/segment_anything_2/sam2_configs/sam2_hiera_l.yaml;
InternVideo2-1B as R/model/videogpt_plus/model/internvideo/internvideo2_stage2_config_vision.py:14-80 configures it; HF CLIP-L/336 and a
2-layer HF LlamaModel at Llama-3-8B width), loaded with the name-seeded weights the full-size GPU tests already use, and run on CPU in fp32:

* fullsize_sam2.npz — the inputs of tests/test_fullsize_gpu.py:test_sam2_large_frame_bf16_vs_oracle (one framewise frame, N = 2) and the
  first T = 3 frames of tests/test_video_fullsize_gpu.py's clip (N = 2) through SAM2VideoPredictor: FPN features, low-res logits, masks,
  object pointers, object scores, frame-0 / frame-1 memories.  Large tensors are stored SUBSAMPLED on fixed strides (the strides are in the
  file); a wrong window, head split or interpolation moves every element, so a strided sample pins them as well as the whole tensor.
* fullsize_vlm.npz — one InternVideo2-1B chunk (4 frames), two CLIP-L/336 frames (every 16th token), a 2-layer Llama-3-8B-width prefill
  of 256 rows (every 8th row).

Only data is written; no reference source is copied.  Weights are regenerated from (name, shape, seed) by oracle/seeded.py.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import as ri  # noqa: E402
from oracle import seeded  # noqa: E402
from videoglamm_amd import synth  # noqa: E402  (manifests + architecture presets only: no device code runs here)
from fullsize_keys import SUB, sub  # noqa: E402

torch.set_grad_enabled(False)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().float().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote", path, f"{os.path.getsize(path) / 1e3:.0f} kB", {k: tuple(np.asarray(v).shape) for k, v in arrays.items()})


def bf16_weights(sd):
    """the full-size tests keep every matrix bf16-representable so that both modes of the product read the same values"""
    return {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in sd.items()}


def load_named(module, prefix, seed, overrides=None):
    """seeded weights keyed by the CHECKPOINT name (prefix + module key): the tests regenerate them from synth's manifests"""
    sd0 = module.state_dict()
    man = {prefix + k: list(v.shape) for k, v in sd0.items()}
    sd = bf16_weights(seeded.seeded_state_dict(man, seed, overrides))
    module.load_state_dict({k: sd[prefix + k].to(v.dtype) for k, v in sd0.items()})
    return man


def gen_sam2():
    import make_golden as mg
    ri.install()
    cfg = ri.sam2_model_cfg("sam2_hiera_l.yaml", True)                      # UNMODIFIED: dims 144..1152, heads 2/4/8/16, windows 8/4/16/8, global 23/33/43
    m = ri.build_sam2(cfg)
    mg.patch_predictor_device()
    man = load_named(m, "", 2, seeded.sam2_overrides())
    want = synth.sam2_manifest(synth.SAM2_L)
    missing = {k for k in want if k not in man or list(man[k]) != list(want[k])}
    assert not missing, sorted(missing)[:5]                                 # the product's manifest names exist, with these shapes, in the reference's module
    out = {}
    # ---- framewise frame (inputs of test_sam2_large_frame_bf16_vs_oracle)
    g = torch.Generator().manual_seed(9)
    img = torch.randn(1, 3, 1024, 1024, generator=g)
    text = torch.randn(2, 256, generator=g) * 0.5
    H, W = 480, 640
    t0 = time.time()
    bo = m.forward_image(img)
    for i in range(3):
        out[f"fw_fpn{i}"] = sub(f"fpn{i}", bo["backbone_fpn"][i])
    out["fw_pos2"] = sub("fpn2", bo["vision_pos_enc"][2])
    sparse, dense = m.sam_prompt_encoder(points=None, boxes=None, masks=None, text_embeds=text.unsqueeze(1))
    _, emb, _, _ = m._prepare_backbone_features(bo)
    emb[-1] = emb[-1] + m.no_mem_embed
    sizes = [(256, 256), (128, 128), (64, 64)]
    feats = [f.permute(1, 2, 0).view(1, -1, *s) for f, s in zip(emb[::-1], sizes[::-1])][::-1]
    low, iou, _, _ = m.sam_mask_decoder(image_embeddings=feats[-1], image_pe=m.sam_prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sparse,
                                        dense_prompt_embeddings=dense, multimask_output=False, repeat_image=True, high_res_features=feats[:-1])
    masks4, iou4, tok4, obj = m.sam_mask_decoder.predict_masks(image_embeddings=feats[-1], image_pe=m.sam_prompt_encoder.get_dense_pe(),
                                                               sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense, repeat_image=True,
                                                               high_res_features=feats[:-1])
    out["fw_low"] = sub("low", low)                                                      # [N,1,256,256] -> strided
    out["fw_iou"] = iou
    out["fw_iou4"], out["fw_tokens4"], out["fw_obj"] = iou4, tok4, obj
    out["fw_masks4"] = sub("low", masks4)
    logits = torch.nn.functional.interpolate(low.float(), (H, W), mode="bilinear", align_corners=False)[:, 0]
    out["fw_logits"] = sub("logits", logits)
    out["fw_mask_frac"] = (logits > 0).float().mean(dim=(1, 2))
    print(f"framewise frame: {time.time() - t0:.0f} s; |low| max {float(low.abs().max()):.2f}; mask fraction {out['fw_mask_frac'].tolist()}")
    # ---- video clip: the first 3 frames of test_sam2_large_video_branch_fp32_vs_oracle's 9 (a shorter clip is a prefix of a longer one:
    #      make_golden.gen_sam2_long checks that on the reference)
    g = torch.Generator().manual_seed(19)
    images = torch.randn(9, 3, 1024, 1024, generator=g)[:3].contiguous()
    text = torch.randn(2, 256, generator=g) * 0.5
    T, N = 3, 2
    rec = []
    hook = m.sam_mask_decoder.pred_obj_score_head.register_forward_hook(lambda mod, i, o: rec.append(o.flatten().clone()))
    t0 = time.time()
    vid, frames = mg._run_ref_video(m, images, text, H, W)
    hook.remove()
    print(f"video clip T={T} N={N}: {time.time() - t0:.0f} s")
    sc = torch.stack([torch.cat(rec[:N])] + rec[N:N + T - 1])
    assert sc.shape == (T, N) and (sc > 0).all(), sc
    lowv = torch.stack([f["pred_masks"] for f in frames])                                 # [T,N,1,256,256]
    out["vid_low"] = sub("low", lowv)
    out["vid_low_absmax"] = lowv.abs().flatten(1).max(dim=1).values
    out["vid_obj_ptr"] = torch.stack([f["obj_ptr"] for f in frames])
    out["vid_obj_scores"] = sc
    out["vid_logits"] = sub("logits", vid)                                                # [T,N,480,640] -> strided
    out["vid_mask_frac"] = (vid > 0).float().mean(dim=(2, 3))
    for t in (0, 1):
        out[f"vid_maskmem{t}"] = sub("maskmem", frames[t]["maskmem_features"].float())     # [N,64,64,64], bf16-valued
    out["vid_maskmem_pos"] = sub("maskmem", frames[1]["maskmem_pos_enc"][-1].float())
    print("video: |low| max per frame", out["vid_low_absmax"].tolist(), "mask fraction", out["vid_mask_frac"].tolist())
    save("fullsize_sam2.npz", **out)


def _merge_save(name, new):
    """parts are generated one at a time (InternVideo2-1B alone takes minutes): keep what the file already holds"""
    path = os.path.join(HERE, name)
    old = dict(np.load(path)) if os.path.exists(path) else {}
    old.update({k: (v.detach().float().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in new.items()})
    save(name, **old)


def gen_iv2():
    ri.install()
    from model.videogpt_plus.model.internvideo.internvideo2 import PretrainInternVideo2
    out = {}
    # ---- InternVideo2-1B: the arguments R/.../internvideo2_stage2_config_vision.py:14-80 gives build_vit (depth 40, 1408, 16 heads of 88,
    #      mlp_ratio 48/11, qk-norm, sep pos embed, clip_return_layer 6 / student interval 1; the fused / flash paths off: CPU)
    c = synth.IV2_1B
    t0 = time.time()
    iv2 = PretrainInternVideo2(in_chans=3, img_size=c["img_size"], patch_size=c["patch_size"], embed_dim=c["embed_dim"], depth=c["depth"],
                               num_heads=c["num_heads"], mlp_ratio=48 / 11, clip_embed_dim=768, attn_pool_num_heads=16, qkv_bias=False,
                               drop_path_rate=0.0, init_values=1e-5, qk_normalization=True, use_flash_attn=False, use_fused_rmsnorm=False,
                               use_fused_mlp=False, num_frames=4, tubelet_size=1, sep_image_video_pos_embed=True, clip_teacher_embed_dim=3200,
                               clip_teacher_final_dim=768, clip_return_layer=6, clip_student_return_interval=1).eval()
    p = "model.vision_tower.vision_encoder."
    man = load_named(iv2, p, 7)
    want = {k: v for k, v in synth.vlm_manifest(synth.videoglamm_llama3_8b()).items() if k.startswith(p)}
    bad = {k for k in want if k not in man or list(man[k]) != list(want[k])}
    assert not bad, sorted(bad)[:5]
    vid = torch.randn(1, 4, 3, 224, 224, generator=torch.Generator().manual_seed(51))
    o = iv2(vid.permute(0, 2, 1, 3, 4), None, False, x_vis_return_idx=-2, x_vis_only=True)     # InternVideo2_Stage2V.forward (internvideo/utils.py:229-238)
    print(f"InternVideo2-1B chunk: {time.time() - t0:.0f} s, out {tuple(o.shape)}, |x| max {float(o.abs().max()):.2f}")
    out["iv2_shape"] = np.array(o.shape)
    out["iv2_out"] = sub("tokens16", o)
    _merge_save("fullsize_vlm.npz", out)


def gen_clip():
    ri.install()
    from transformers import CLIPVisionConfig, CLIPVisionModel
    out = {}
    # ---- CLIP-L/336 (HF CLIPVisionModel, as R/.../multimodal_encoder/clip_encoder.py:34-72 selects hidden_states[-2][:, 1:])
    c = synth.CLIP_L_336
    t0 = time.time()
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=c["hidden"], intermediate_size=c["mlp"], num_hidden_layers=c["num_layers"],
                                            num_attention_heads=c["num_heads"], image_size=c["img_size"], patch_size=c["patch_size"],
                                            hidden_act="quick_gelu", attn_implementation="eager")).eval()
    # transformers 5.x names the tower's parameters without the "vision_model." level the 4.41 checkpoints carry: seed by the CHECKPOINT name
    p = "model.image_vision_tower.vision_tower." + ("" if next(iter(clip.state_dict())).startswith("vision_model.") else "vision_model.")
    man = load_named(clip, p, 8)
    want = {k: v for k, v in synth.vlm_manifest(synth.videoglamm_llama3_8b()).items() if k.startswith(p)}
    bad = {k for k in want if k not in man or list(man[k]) != list(want[k])}
    assert not bad, sorted(bad)[:5]
    img = torch.randn(2, 3, 336, 336, generator=torch.Generator().manual_seed(52))
    o = clip(img, output_hidden_states=True).hidden_states[-2][:, 1:]
    print(f"CLIP-L/336 x2: {time.time() - t0:.0f} s, out {tuple(o.shape)}, |x| max {float(o.abs().max()):.2f}")
    out["clip_shape"] = np.array(o.shape)
    out["clip_out"] = sub("tokens16", o)
    _merge_save("fullsize_vlm.npz", out)


def gen_llama():
    ri.install()
    from transformers import LlamaConfig, LlamaModel
    out = {}
    # ---- 2 decoder layers at Llama-3-8B width: the weights of tests/test_fullsize_gpu.py:_llama2 (seed 5), 256 bf16-valued rows (seed 53)
    c = dict(synth.LLAMA3_8B, num_layers=2, vocab=8192)
    t0 = time.time()
    llm = LlamaModel(LlamaConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["ffn"], num_hidden_layers=2,
                                 num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"], rms_norm_eps=c["rms_eps"],
                                 rope_theta=c["rope_theta"], max_position_embeddings=8192, attn_implementation="eager")).eval()
    man = load_named(llm, "model.", 5)
    x = (torch.randn(256, c["hidden"], generator=torch.Generator().manual_seed(53)) * 0.5).to(torch.bfloat16).float()
    o = llm(inputs_embeds=x[None]).last_hidden_state[0]
    print(f"Llama-3-8B-width x2 layers, 256 rows: {time.time() - t0:.0f} s, |h| max {float(o.abs().max()):.2f}")
    out["llama_out"] = o[::8].contiguous()
    _merge_save("fullsize_vlm.npz", out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    for part, fn in (("iv2", gen_iv2), ("clip", gen_clip), ("llama", gen_llama)):
        if what in (part, "vlm", "all"):
            fn()
    if what in ("sam2", "all"):
        gen_sam2()
