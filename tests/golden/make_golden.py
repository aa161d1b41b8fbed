"""Golden-vector generator.  RUN ONLY IN THE BUILD CONTAINER (needs /root/reference):

    python tests/golden/make_golden.py [sam2|vlm|e2e|all]

Imports the reference (recipe: _ref_import.py), loads the name-seeded synthetic weights of
oracle/seeded.py into the reference's own nn.Modules, runs them on seeded inputs on CPU/fp32 and
stores inputs' seeds + outputs as small .npz fixtures, plus the {name: shape} manifests the tests need
to regenerate the same weights.  Only data is written; no reference source is copied.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import as ri  # noqa: E402
from oracle import seeded  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)

# ---- the micro configurations shared with the tests (tests/golden/configs.py) --------------------
from configs import CLIP_TINY, IV2_TINY, LLAMA_TINY, SAM2_E2E, SAM2_MICRO  # noqa: E402


def rnd(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().float().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote", path, {k: tuple(np.asarray(v).shape) for k, v in arrays.items()}, f"{os.path.getsize(path) / 1e3:.0f} kB")


def load_seeded(module, seed, overrides=None, manifest_name=None):
    sd0 = module.state_dict()
    manifest = {k: list(v.shape) for k, v in sd0.items()}
    sd = seeded.seeded_state_dict(manifest, seed, overrides)
    module.load_state_dict({k: v.to(sd0[k].dtype) for k, v in sd.items()})
    if manifest_name:
        with open(os.path.join(HERE, manifest_name), "w") as f:
            json.dump(manifest, f)
        print("wrote", manifest_name, len(manifest), "tensors", sum(int(np.prod(s)) for s in manifest.values()), "params")
    return sd


def patch_predictor_device():
    from model.segment_anything_2.sam2 import sam2_video_predictor as svp

    orig = svp.SAM2VideoPredictor.init_state_from_tensor

    def patched(self, *a, **k):  # hard-coded torch.device("cuda") at sam2_video_predictor.py:143-147
        real = torch.device
        try:
            torch.device = lambda *aa, **kk: real("cpu")
            return orig(self, *a, **k)
        finally:
            torch.device = real

    if not getattr(svp.SAM2VideoPredictor, "_vg_patched", False):
        svp.SAM2VideoPredictor.init_state_from_tensor = torch.inference_mode()(patched)
        svp.SAM2VideoPredictor._vg_patched = True


def build_ref_sam2(seed=1):
    ri.install()
    cfg = ri.sam2_model_cfg("sam2_hiera_l.yaml", True, trunk_override=SAM2_MICRO["trunk"],
                            neck_channels=SAM2_MICRO["neck_channels"], image_size=SAM2_MICRO["image_size"])
    m = ri.build_sam2(cfg)
    patch_predictor_device()
    load_seeded(m, seed, seeded.sam2_overrides(), "sam2_micro_manifest.json")
    return m


def gen_sam2():
    m = build_ref_sam2()
    S = SAM2_MICRO["image_size"]
    T, N, H, W = 5, 2, 40, 56
    images = rnd((T, 3, S, S), 11)
    text = rnd((N, 256), 12, 0.5)
    out = {}
    # --- S1: forward_image (Hiera + FPN + conv_s0/s1)
    bo = m.forward_image(images[0:1])
    out["fpn0"], out["fpn1"], out["fpn2"] = bo["backbone_fpn"]
    out["pos2"] = bo["vision_pos_enc"][2]
    # --- S7/S8 framewise decode exactly as VideoGLaMM.inference_framewise does (R/model/VideoGLaMM.py:689-749)
    sparse, dense = m.sam_prompt_encoder(points=None, boxes=None, masks=None, text_embeds=text.unsqueeze(1))
    out["dense_pe"] = m.sam_prompt_encoder.get_dense_pe()
    lows, fw = [], []
    for t in range(T):
        bo = m.forward_image(images[t:t + 1])
        _, emb, _, _ = m._prepare_backbone_features(bo)
        emb[-1] = emb[-1] + m.no_mem_embed
        sizes = [(S // 4, S // 4), (S // 8, S // 8), (S // 16, S // 16)]
        feats = [f.permute(1, 2, 0).view(1, -1, *s) for f, s in zip(emb[::-1], sizes[::-1])][::-1]
        low, iou, _, _ = m.sam_mask_decoder(image_embeddings=feats[-1], image_pe=m.sam_prompt_encoder.get_dense_pe(),
                                            sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                                            multimask_output=False, repeat_image=True, high_res_features=feats[:-1])
        lows.append(low)
        fw.append(torch.nn.functional.interpolate(low.float(), (H, W), mode="bilinear", align_corners=False)[:, 0])
    out["framewise_low"] = torch.stack(lows)
    out["framewise_logits"] = torch.stack(fw)
    # all-4-token decoder output on frame 0 (pre multimask selection)
    masks4, iou4, tok4, obj = m.sam_mask_decoder.predict_masks(
        image_embeddings=feats[-1], image_pe=m.sam_prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sparse,
        dense_prompt_embeddings=dense, repeat_image=True, high_res_features=feats[:-1])
    out["dec_masks4"], out["dec_iou4"], out["dec_tokens4"], out["dec_obj"] = masks4, iou4, tok4, obj
    # --- video branch exactly as VideoGLaMM.inference_video_branch does (R/model/VideoGLaMM.py:844-877)
    state = m.init_state_from_tensor(images, H, W)
    m.reset_state(state)
    feat = text.unsqueeze(1)
    for k in range(N):
        m.add_new_text(inference_state=state, frame_idx=0, obj_id=k, text=feat[k].unsqueeze(0))
    vid = []
    for fidx, obj_ids, logits in m.propagate_in_video(state):
        vid.append(logits.clone())
    out["video_logits"] = torch.stack(vid)
    od = state["output_dict"]
    out["video_low_res"] = torch.stack([od["cond_frame_outputs"][0]["pred_masks"]] + [od["non_cond_frame_outputs"][t]["pred_masks"] for t in range(1, T)])
    out["video_obj_ptr"] = torch.stack([od["cond_frame_outputs"][0]["obj_ptr"]] + [od["non_cond_frame_outputs"][t]["obj_ptr"] for t in range(1, T)])
    out["video_maskmem0"] = od["cond_frame_outputs"][0]["maskmem_features"].float()
    out["video_maskmem1"] = od["non_cond_frame_outputs"][1]["maskmem_features"].float()
    # --- S5 / S9 modules in isolation
    hw = (S // 16) ** 2
    curr, cpos = rnd((hw, N, 256), 21), rnd((hw, N, 256), 22)
    mem, mpos = rnd((2 * hw + 8, N, 64), 23), rnd((2 * hw + 8, N, 64), 24)
    out["memattn_out"] = m.memory_attention(curr=[curr], curr_pos=[cpos], memory=mem, memory_pos=mpos, num_obj_ptr_tokens=8)
    pix, msk = rnd((N, 256, S // 16, S // 16), 25), rnd((N, 1, S, S), 26, 3.0)
    mo = m.memory_encoder(pix, torch.sigmoid(msk) * 20 - 10, skip_mask_sigmoid=True)
    out["memenc_feat"], out["memenc_pos"] = mo["vision_features"], mo["vision_pos_enc"][0]
    save("sam2_micro.npz", meta=np.array([T, N, H, W]), **out)


from make_golden_keys import LONG_MEM_FRAMES  # noqa: E402


def gen_sam2_long():
    """video branch over T = 18 frames, N = 2 objects: past the 7-slot memory bank (t >= 8: the oldest memory drops out, the tpos slots
    roll) and past the 16-pointer cap (t >= 17: the oldest non-conditioning pointer drops out) — R/.../sam2_base.py:536-633,
    sam2_video_predictor.py:744-827.  The first 9 frames of this run ARE the T = 9 run of the same clip up to the pointer window
    (min(T, 16) pointers: no difference before frame 16), so one fixture pins both lengths."""
    m = build_ref_sam2()
    S = SAM2_MICRO["image_size"]
    T, N, H, W = 18, 2, 40, 56
    images = rnd((T, 3, S, S), 41)
    text = rnd((N, 256), 42, 0.5)
    state = m.init_state_from_tensor(images, H, W)
    m.reset_state(state)
    for k in range(N):
        m.add_new_text(inference_state=state, frame_idx=0, obj_id=k, text=text.unsqueeze(1)[k].unsqueeze(0))
    vid = [logits.clone() for _, _, logits in m.propagate_in_video(state)]
    od = state["output_dict"]
    frames = [od["cond_frame_outputs"][0]] + [od["non_cond_frame_outputs"][t] for t in range(1, T)]
    out = dict(meta=np.array([T, N, H, W]), video_logits=torch.stack(vid)[:, :, 0], low_res=torch.stack([f["pred_masks"] for f in frames]),
               obj_ptr=torch.stack([f["obj_ptr"] for f in frames]))
    for t in LONG_MEM_FRAMES:
        out[f"maskmem_{t}"] = frames[t]["maskmem_features"].float()
    # the same clip cut to 9 frames: must equal the first 9 frames of the long run (checked here, against the reference itself)
    state9 = m.init_state_from_tensor(images[:9], H, W)
    m.reset_state(state9)
    for k in range(N):
        m.add_new_text(inference_state=state9, frame_idx=0, obj_id=k, text=text.unsqueeze(1)[k].unsqueeze(0))
    vid9 = torch.stack([logits.clone() for _, _, logits in m.propagate_in_video(state9)])[:, :, 0]
    assert torch.equal(vid9, out["video_logits"][:9]), "T = 9 is not a prefix of T = 18 in the reference"
    frac = (out["video_logits"] > 0).float().mean(dim=(1, 2, 3))
    print("mask fraction per frame:", [round(float(f), 3) for f in frac])
    save("sam2_video_long.npz", **out)


def _run_ref_video(m, images, text, H, W):
    state = m.init_state_from_tensor(images, H, W)
    m.reset_state(state)
    for k in range(text.shape[0]):
        m.add_new_text(inference_state=state, frame_idx=0, obj_id=k, text=text.unsqueeze(1)[k].unsqueeze(0))
    vid = [logits.clone() for _, _, logits in m.propagate_in_video(state)]
    od = state["output_dict"]
    T = images.shape[0]
    frames = [od["cond_frame_outputs"][0]] + [od["non_cond_frame_outputs"][t] for t in range(1, T)]
    return torch.stack(vid)[:, :, 0], frames


def gen_sam2_noobj():
    """video branch with objects that DISAPPEAR (object_score_logits <= 0): NO_OBJ_SCORE fill of the masks, the no_obj_ptr mix
    of the pointers, the -1024 mask through the bilinear upsample + memory encoder — R/.../sam2_base.py:355-364,390-401,
    sam2_video_predictor.py:571-612.  The random-init score head barely tells objects / frames apart (spread 0.01 around -0.5),
    so its last layer is rescaled: score' = K (score + c), seeded.sam2_noobj_overrides(c, K).  c is searched on the reference
    itself among values that split the two objects on frame 0, for the largest margin |score'| over all (frame, object) pairs
    (a margin far above fp32 summation noise keeps every present / absent decision implementation-independent); c, K and the
    score table are stored in the fixture."""
    m = build_ref_sam2()
    S = SAM2_MICRO["image_size"]
    T, N, H, W = 9, 2, 40, 56
    images = rnd((T, 3, S, S), 71)
    text = rnd((N, 256), 72, 0.5)
    rec = []
    hook = m.sam_mask_decoder.pred_obj_score_head.register_forward_hook(lambda mod, i, o: rec.append(o.flatten().clone()))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    K, best = 30.0, None
    for c in [0.556 + 0.0005 * i for i in range(23)]:
        sd = dict(sd0)
        for name, fn in seeded.sam2_noobj_overrides(c, K).items():
            sd[name] = fn(sd0[name])
        m.load_state_dict(sd)
        rec.clear()
        vid, frames = _run_ref_video(m, images, text, H, W)
        sc = torch.stack([torch.cat(rec[:N])] + rec[N:])                           # [T,N]: frame 0 is one call per object
        pres = sc > 0
        ok = bool(pres.any(0).all() and (~pres).any(0).all() and (pres[:, 0] != pres[:, 1]).sum() >= 2 and pres[0, 0] != pres[0, 1])
        margin = float(sc.abs().min())
        print(f"c {c:.4f} margin {margin:.4f} ok {ok} pattern {pres.int().tolist()}")
        if ok and (best is None or margin > best[0]):
            best = (margin, c, vid, frames, sc)
    hook.remove()
    margin, c, vid, frames, sc = best
    assert margin > 0.03, margin
    print("chosen c", c, "margin", margin, "scores", sc.tolist())
    out = dict(meta=np.array([T, N, H, W]), score_c=np.float64(c), score_k=np.float64(K), obj_scores=sc, video_logits=vid,
               low_res=torch.stack([f["pred_masks"] for f in frames]), obj_ptr=torch.stack([f["obj_ptr"] for f in frames]))
    for t in range(T):
        out[f"maskmem_{t}"] = frames[t]["maskmem_features"].float()
    save("sam2_noobj.npz", **out)


def gen_vlm():
    ri.install()
    from model.videogpt_plus.model.internvideo.internvideo2 import PretrainInternVideo2
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaModel

    out = {}
    c = IV2_TINY
    iv2 = PretrainInternVideo2(in_chans=3, img_size=c["img_size"], patch_size=c["patch_size"], embed_dim=c["embed_dim"], depth=c["depth"],
                               num_heads=c["num_heads"], mlp_ratio=c["mlp_ratio"], clip_embed_dim=32, attn_pool_num_heads=4, qkv_bias=False,
                               drop_path_rate=0.0, init_values=1e-5, qk_normalization=True, use_flash_attn=False, use_fused_rmsnorm=False,
                               use_fused_mlp=False, num_frames=4, tubelet_size=1, sep_image_video_pos_embed=True, clip_teacher_embed_dim=32,
                               clip_teacher_final_dim=16, clip_return_layer=1, clip_student_return_interval=1).eval()
    load_seeded(iv2, 2, None, "iv2_tiny_manifest.json")
    vid = rnd((2, 4, 3, c["img_size"], c["img_size"]), 31)
    # InternVideo2_Stage2V.forward (internvideo/utils.py:229-238)
    out["iv2_out"] = iv2(vid.permute(0, 2, 1, 3, 4), None, False, x_vis_return_idx=-2, x_vis_only=True)

    c = CLIP_TINY
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=c["hidden"], intermediate_size=c["mlp"], num_hidden_layers=c["num_layers"],
                                            num_attention_heads=c["num_heads"], image_size=c["img_size"], patch_size=c["patch_size"],
                                            hidden_act="quick_gelu", attn_implementation="eager")).eval()
    load_seeded(clip, 3, None, "clip_tiny_manifest.json")
    img = rnd((3, 3, c["img_size"], c["img_size"]), 32)
    # CLIPVisionTower.forward/feature_select (clip_encoder.py:34-72)
    out["clip_out"] = clip(img, output_hidden_states=True).hidden_states[-2][:, 1:]

    c = LLAMA_TINY
    llm = LlamaModel(LlamaConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["ffn"], num_hidden_layers=c["num_layers"],
                                 num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"], rms_norm_eps=c["rms_eps"],
                                 rope_theta=c["rope_theta"], max_position_embeddings=4096, attn_implementation="eager")).eval()
    load_seeded(llm, 4, None, "llama_tiny_manifest.json")
    emb = rnd((1, 45, c["hidden"]), 33)
    out["llama_out"] = llm(inputs_embeds=emb).last_hidden_state[0]
    save("vlm_tiny.npz", **out)


def gen_phi3():
    """HF Phi3Model (the decoder the reference's VideoGPTPlusPhi3ForCausalLM wraps, language_model/phi3.py:29-40) on a tiny
    config with the released checkpoint's structure: fused qkv_proj / gate_up_proj, MHA, sliding_window 2047."""
    ri.install()
    from transformers import Phi3Config, Phi3Model
    from configs import PHI3_TINY

    c = PHI3_TINY
    llm = Phi3Model(Phi3Config(vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["ffn"], num_hidden_layers=c["num_layers"],
                               num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"], rms_norm_eps=c["rms_eps"],
                               rope_theta=c["rope_theta"], max_position_embeddings=4096, original_max_position_embeddings=4096,
                               sliding_window=c["sliding_window"], pad_token_id=0, bos_token_id=1, eos_token_id=2,
                               attn_implementation="eager")).eval()
    load_seeded(llm, 5, None, "phi3_tiny_manifest.json")
    emb = rnd((1, 45, c["hidden"]), 34)
    out = dict(phi3_out=llm(inputs_embeds=emb).last_hidden_state[0])
    # the same weights with a window the 45-token sequence crosses.  transformers 5.x masks kv <= q - sliding_window (sliding_window keys
    # visible, own position included); the reference's pin 4.41.0 shows sliding_window + 1 keys — so HF-5 sliding_window = 12 is the mask
    # of a 4.41 model with sliding_window = 11 (PHI3_TINY_WIN), which is what the oracle / product are given
    from configs import PHI3_TINY_WIN
    w5 = PHI3_TINY_WIN["sliding_window"] + 1
    win = Phi3Model(Phi3Config(vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["ffn"], num_hidden_layers=c["num_layers"],
                               num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"], rms_norm_eps=c["rms_eps"],
                               rope_theta=c["rope_theta"], max_position_embeddings=4096, original_max_position_embeddings=4096,
                               sliding_window=w5, pad_token_id=0, bos_token_id=1, eos_token_id=2, attn_implementation="eager")).eval()
    win.load_state_dict(llm.state_dict())
    out["phi3_win_out"] = win(inputs_embeds=emb).last_hidden_state[0]
    assert (out["phi3_win_out"] - out["phi3_out"]).abs().max() > 1e-3, "the window must change the result"
    save("phi3_tiny.npz", **out)


def build_ref_e2e(use_video_branch):
    """Compose VideoGLaMM_SAM2 with the Llama wrapper exactly the way R/model/VideoGLaMM.py:155-173,882-903
    composes it with Phi-3 (the reference ships no Llama composition — SURVEY headline 3)."""
    ri.install()
    os.chdir(ri.R)
    import model.VideoGLaMM as vg
    from model.videogpt_plus.model.language_model.llama3_1 import (VideoGPTPlusLlamaConfig, VideoGPTPlusLlamaForCausalLM,
                                                                    VideoGPTPlusLlamaModel)
    from model.videogpt_plus.model.internvideo.internvideo2 import PretrainInternVideo2
    from model.videogpt_plus.model.internvideo.utils import InternVideo2_Stage2V
    from model.videogpt_plus.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from model.videogpt_plus.model.multimodal_projector.builder import build_vision_projector
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from configs import E2E

    def micro_sam(*a, **k):
        cfg = ri.sam2_model_cfg("sam2_hiera_l.yaml", use_video_branch, trunk_override=SAM2_E2E["trunk"],
                                neck_channels=SAM2_E2E["neck_channels"], image_size=SAM2_E2E["image_size"])
        if not use_video_branch:
            cfg["_target_"] = "model.segment_anything_2.sam2.modeling.sam2_base.SAM2Base"
        return ri.build_sam2(cfg)

    patch_predictor_device()
    vg.build_sam2 = micro_sam
    vg.build_sam2_video_predictor = micro_sam

    class VideoGLaMMLlamaModel(vg.VideoGLaMMMetaModel, VideoGPTPlusLlamaModel):
        def __init__(self, config, **kwargs):
            super().__init__(config, **kwargs)
            self.config.use_cache = False
            self.config.mm_vision_select_feature = "patch"

    class VideoGLaMMLlamaForCausalLM(VideoGPTPlusLlamaForCausalLM, vg.VideoGLaMM_SAM2):
        def __init__(self, config, **kwargs):
            super(VideoGPTPlusLlamaForCausalLM, self).__init__(config)
            self.model = VideoGLaMMLlamaModel(config, **kwargs)
            self.lm_head = torch.nn.Linear(config.hidden_size, config.vocab_size, bias=False)
            self.post_init()

        def forward(self, **kwargs):
            if "past_key_values" in kwargs:
                return VideoGPTPlusLlamaForCausalLM.forward(self, **kwargs)
            return vg.VideoGLaMM_SAM2.model_forward(self, **kwargs)

        def super_forward(self, **kwargs):
            return VideoGPTPlusLlamaForCausalLM.forward(self, **kwargs)

    c = E2E["llm"]
    config = VideoGPTPlusLlamaConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["ffn"], num_hidden_layers=c["num_layers"],
                                     num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"], rms_norm_eps=c["rms_eps"],
                                     rope_theta=c["rope_theta"], max_position_embeddings=8192, attn_implementation="eager",
                                     bos_token_id=1, eos_token_id=2, pad_token_id=0)
    config.mm_projector_type = "mlp2x_gelu"
    config.image_mm_projector_type = "mlp2x_gelu"
    m = VideoGLaMMLlamaForCausalLM(config, train_mask_decoder=False, out_dim=256, mask_decoder_itm=False, use_sam2=True,
                                   use_sam2_video_branch=use_video_branch)
    m.config.seg_token_idx = E2E["seg_token_idx"]
    iv = E2E["iv2"]

    class _IV2(torch.nn.Module):
        forward = InternVideo2_Stage2V.forward
        dtype = property(lambda self: self.vision_encoder.patch_embed.proj.weight.dtype)

    tower = _IV2()
    tower.vision_encoder = PretrainInternVideo2(in_chans=3, img_size=iv["img_size"], patch_size=iv["patch_size"], embed_dim=iv["embed_dim"],
                                                depth=iv["depth"], num_heads=iv["num_heads"], mlp_ratio=iv["mlp_ratio"], clip_embed_dim=32,
                                                attn_pool_num_heads=4, qkv_bias=False, drop_path_rate=0.0, init_values=1e-5, qk_normalization=True,
                                                use_flash_attn=False, use_fused_rmsnorm=False, use_fused_mlp=False, num_frames=4, tubelet_size=1,
                                                sep_image_video_pos_embed=True, clip_teacher_embed_dim=32, clip_teacher_final_dim=16,
                                                clip_return_layer=1, clip_student_return_interval=1)
    cl = E2E["clip"]
    ctower = CLIPVisionTower.__new__(CLIPVisionTower)
    torch.nn.Module.__init__(ctower)
    ctower.is_loaded, ctower.select_layer, ctower.select_feature = True, -2, "patch"
    ctower.vision_tower = CLIPVisionModel(CLIPVisionConfig(hidden_size=cl["hidden"], intermediate_size=cl["mlp"], num_hidden_layers=cl["num_layers"],
                                                           num_attention_heads=cl["num_heads"], image_size=cl["img_size"], patch_size=cl["patch_size"],
                                                           hidden_act="quick_gelu", attn_implementation="eager"))
    m.model.vision_tower = tower
    m.model.image_vision_tower = ctower
    m.model.mm_projector = build_vision_projector(config, image_mm_projector=False)
    m.model.image_mm_projector = build_vision_projector(config, image_mm_projector=True)
    m.eval()
    return m


def _capture_seam(m):
    """record what crosses the LLM -> SAM2 seam inside the reference's inference(): the [SEG] embeddings handed to the prompt
    encoder (text_embeds, R/model/VideoGLaMM.py:692-695 / sam2_video_predictor.py:415-495) and the mask logits at output
    resolution BEFORE the `> 0` (postprocess_masks, VideoGLaMM.py:746-749; propagate_in_video, :869-875)."""
    cap = dict(emb=[], logits=[])
    vm = m.model.visual_model
    vm.sam_prompt_encoder.register_forward_pre_hook(
        lambda mod, args, kw: cap["emb"].append(kw["text_embeds"].detach().float().reshape(-1, 256).clone()) if kw.get("text_embeds") is not None else None,
        with_kwargs=True)
    post = m.model.postprocess_masks

    def post_rec(*a, **k):
        out = post(*a, **k)
        cap["logits"].append(out[:, 0].detach().float().clone())
        return out

    m.model.postprocess_masks = post_rec
    if hasattr(vm, "propagate_in_video"):
        prop = vm.propagate_in_video

        def prop_rec(*a, **k):
            for fidx, oids, logits in prop(*a, **k):
                cap["logits"].append(logits[:, 0].detach().float().clone())
                yield fidx, oids, logits

        vm.propagate_in_video = prop_rec
    return cap


def _run_e2e(m, branch, images, context, sam, ids, H, W, S, max_new_tokens, fixtures, key):
    cap = _capture_seam(m)
    out_ids, segs = m.inference(images=[images], context_images=[context], images_for_sam=[sam], input_ids=ids,
                                resize_list=[(S, S)], original_size_list=[(H, W)], max_new_tokens=max_new_tokens,
                                use_sam2_video_branch=branch)
    fixtures[f"{key}_output_ids"] = out_ids[0].numpy()
    seg = segs[0]
    frames = sorted(seg.keys())
    objs = sorted(seg[frames[0]].keys()) if frames else []
    fixtures[f"{key}_masks"] = np.stack([np.stack([seg[t][k] for k in objs]) for t in frames]) if frames else np.zeros((0,))
    fixtures[f"{key}_seg_emb"] = torch.cat(cap["emb"]).numpy()                       # [N,256]
    fixtures[f"{key}_logits"] = torch.stack(cap["logits"]).numpy()                  # [T,N,H,W] before the threshold
    assert fixtures[f"{key}_seg_emb"].shape == (len(objs), 256) and fixtures[f"{key}_logits"].shape[:2] == (len(frames), len(objs))
    assert np.array_equal(fixtures[f"{key}_logits"] > 0, fixtures[f"{key}_masks"])
    print(key, "output_ids", out_ids[0].tolist(), "n_seg", len(objs), "mask px", [int(seg[t][k].sum()) for t in frames for k in objs][:8])


def gen_e2e():
    from configs import E2E

    fixtures = {}
    manifest_done = False
    for branch in (False, True):
        m = build_ref_e2e(branch)
        sd0 = m.state_dict()
        # drop the towers' unused heads from the manifest? keep everything: names are the checkpoint contract
        ov = seeded.sam2_overrides("model.visual_model.")
        load_seeded(m, 5, ov, None if manifest_done else "e2e_manifest.json")
        manifest_done = True
        # make the random LM emit [SEG] deterministically: large lm_head row norm for the seg token
        te, S = E2E["te"], SAM2_E2E["image_size"]
        T, H, W = E2E["t_sam"], 40, 56
        images = rnd((te, 3, 224, 224), 41)
        context = rnd((te, 3, 336, 336), 42)
        sam = rnd((T, 3, S, S), 43)
        g = torch.Generator().manual_seed(44)
        ids = torch.cat([torch.tensor([1, 5, 6]), torch.full((te,), -200), torch.randint(3, E2E["llm"]["vocab"], (12,), generator=g)])[None]
        # Version-skew guard (transformers 5.x here vs the reference's pinned 4.41): after the first outer
        # forward with output_hidden_states=True, 5.x records every CLIP layer output twice, so
        # hidden_states[-2] silently becomes the LAST layer.  Evaluate the reference's CLIPVisionTower once
        # while the 4.41 semantics still hold (3 entries for 2 layers) and pin its forward to that result.
        ctower = m.get_model().get_image_vision_tower()
        assert len(ctower.vision_tower(context[:1], output_hidden_states=True).hidden_states) == E2E["clip"]["num_layers"] + 1
        clip_feats = ctower(context, select_feature="patch")
        ctower.forward = lambda imgs, select_feature="patch", batch_size=128: clip_feats if imgs.shape == context.shape else (_ for _ in ()).throw(RuntimeError("unexpected CLIP input"))
        key = "video" if branch else "framewise"
        with torch.no_grad():
            # a random LM never emits token 300: pick as [SEG] a token it DOES emit (SURVEY §8c gotcha);
            # the choice is recorded in the fixture so the tests use the same index
            probe = m.generate(images=[images], context_images=[context], input_ids=ids, max_new_tokens=E2E["max_new_tokens"],
                               num_beams=1, use_cache=False)
            gen = probe[0, ids.shape[1]:].tolist()
            seg_idx = gen[1]
            m.config.seg_token_idx = seg_idx
            fixtures["seg_token_idx"] = np.array(seg_idx)
            print("generated", gen, "-> seg_token_idx", seg_idx)
            _run_e2e(m, branch, images, context, sam, ids, H, W, S, E2E["max_new_tokens"], fixtures, key)
            fixtures["input_ids"] = ids[0].numpy()
            # C4's shape: EIGHT [SEG] objects.  seg_token_mask covers the prompt as well (output_ids[:, 1:] == seg_token_idx,
            # R/model/VideoGLaMM.py:630-633,803-806), so eight [SEG] ids in the prompt's text give eight objects whose embeddings come
            # from prompt rows (the multi-turn case), plus whatever the model emits on top
            ids8 = torch.cat([ids[0, :3 + te + 4], torch.full((8,), seg_idx), ids[0, 3 + te + 4:]])[None]
            _run_e2e(m, branch, images, context, sam, ids8, H, W, S, E2E["max_new_tokens"], fixtures, key + "8")
            fixtures["input_ids8"] = ids8[0].numpy()
    save("e2e_tiny.npz", **fixtures)


def gen_e2e_image():
    """single-image prompt: inference(images=[img], context_images=None, ...) — the reference then takes the image path
    of prepare_inputs_labels_for_multimodal (arch.py:243-245,393-397: encode_images + project(input_type="image"): CLIP
    patch features -> image_mm_projector, no pooling, one <image> placeholder) and one SAM frame, framewise branch."""
    from configs import E2E

    m = build_ref_e2e(False)
    load_seeded(m, 5, seeded.sam2_overrides("model.visual_model."), None)
    S, H, W = SAM2_E2E["image_size"], 40, 56
    img = rnd((1, 3, 336, 336), 51)
    sam = rnd((1, 3, S, S), 52)
    g = torch.Generator().manual_seed(53)
    ids = torch.cat([torch.tensor([1, 5, 6, -200]), torch.randint(3, E2E["llm"]["vocab"], (12,), generator=g)])[None]
    ctower = m.get_model().get_image_vision_tower()
    assert len(ctower.vision_tower(img, output_hidden_states=True).hidden_states) == E2E["clip"]["num_layers"] + 1
    feats = ctower(img, select_feature="patch")           # 4.41 semantics pinned (see gen_e2e)
    ctower.forward = lambda imgs, select_feature="patch", batch_size=128: feats if imgs.shape == img.shape else (_ for _ in ()).throw(RuntimeError("unexpected CLIP input"))
    fixtures = {}
    with torch.no_grad():
        probe = m.generate(images=[img], context_images=None, input_ids=ids, max_new_tokens=E2E["max_new_tokens"], num_beams=1, use_cache=False)
        gen = probe[0, ids.shape[1]:].tolist()
        seg_idx = gen[1]
        m.config.seg_token_idx = seg_idx
        print("generated", gen, "-> seg_token_idx", seg_idx)
        out_ids, segs = m.inference(images=[img], context_images=None, images_for_sam=[sam], input_ids=ids, resize_list=[(S, S)],
                                    original_size_list=[(H, W)], max_new_tokens=E2E["max_new_tokens"], use_sam2_video_branch=False)
    seg = segs[0]
    frames = sorted(seg.keys())
    objs = sorted(seg[frames[0]].keys())
    fixtures["seg_token_idx"] = np.array(seg_idx)
    fixtures["input_ids"] = ids[0].numpy()
    fixtures["output_ids"] = out_ids[0].numpy()
    fixtures["masks"] = np.stack([np.stack([seg[t][k] for k in objs]) for t in frames])
    print("image mode: output_ids", out_ids[0].tolist(), "n_seg", len(objs), "mask px", [int(seg[t][k].sum()) for t in frames for k in objs])
    save("e2e_image.npz", **fixtures)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("e2e_image", "all"):
        gen_e2e_image()
    if what in ("sam2", "all"):
        gen_sam2()
    if what in ("sam2_long", "all"):
        gen_sam2_long()
    if what in ("sam2_noobj", "all"):
        gen_sam2_noobj()
    if what in ("vlm", "all"):
        gen_vlm()
    if what in ("phi3", "all"):
        gen_phi3()
    if what in ("e2e", "all"):
        gen_e2e()


# ---- host rows (H1/H2): run the reference's own helper functions on toy inputs --------------------------
class ToyTokenizer:
    """whitespace tokenizer with a BOS id, enough for tokenizer_image_token / templates."""
    bos_token_id = 1

    def __init__(self):
        self.vocab = {}

    def __call__(self, text):
        ids = [self.bos_token_id]
        for w in text.replace("\n", " \n ").split(" "):
            if w == "":
                continue
            ids.append(self.vocab.setdefault(w, 10 + len(self.vocab)))
        return type("Enc", (), {"input_ids": ids})()


def gen_host():
    ri.install()
    from PIL import Image
    import torchvision.transforms.functional as tvf

    tvf.to_pil_image = Image.fromarray
    tvf.resize = lambda img, size: img.resize((size[1], size[0]), Image.BILINEAR)  # torchvision default for PIL inputs
    import model.segment_anything.utils.transforms as tr
    tr.resize, tr.to_pil_image = tvf.resize, tvf.to_pil_image
    from utils.sam_transforms import sam_preprocess
    from model.videogpt_plus.mm_utils import tokenizer_image_token
    from model.videogpt_plus import conversation as conv

    out = {}
    g = np.random.RandomState(7)
    frame = g.randint(0, 256, size=(60, 80, 3)).astype(np.uint8)
    x, shape = sam_preprocess(frame, model_type="sam2")
    out["sam_pre_sub"], out["sam_pre_shape"] = x[:, ::16, ::16], np.array(shape)
    out["sam_pre_mean"] = x.mean(dim=(1, 2))
    tok = ToyTokenizer()
    for name, key in (("phi3_instruct", "phi3"), ("llama3_1", "llama3_1")):
        c = conv.conv_templates[name].copy()
        c.messages = []
        c.append_message(c.roles[0], "<image>" * 4 + "\n" + "Please segment the red car .")
        c.append_message(c.roles[1], "")
        out[f"ids_{key}"] = tokenizer_image_token(c.get_prompt(), tok, return_tensors="pt").numpy()
    # CLIP stream of H1: the reference's own EncPreprocessor_VideoGPTPlus.preprocess (R/utils/enc_preprocessors.py:120-166) with the
    # processor it would fetch from the hub ("openai/clip-vit-large-patch14-336": no network here) constructed offline from that
    # checkpoint's published preprocessor_config.json values (size 336 shortest edge, bicubic = resample 3, centre crop 336, CLIP mean /
    # std), and the cv2-based InternVideo2 processor (cv2 absent) replaced by an empty stand-in: only "context_images" is taken
    from transformers import CLIPImageProcessor
    from utils.enc_preprocessors import EncPreprocessor_VideoGPTPlus
    enc = EncPreprocessor_VideoGPTPlus.__new__(EncPreprocessor_VideoGPTPlus)
    enc.num_frames, enc.frame_resolution_iv, enc.frame_resolution_clip = 4, 224, 336
    enc.image_processor = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, do_resize=True, do_center_crop=True,
                                             do_normalize=True, do_rescale=True, do_convert_rgb=True, resample=3,
                                             image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711])
    enc.video_processor = type("NoIV2", (), {"preprocess": staticmethod(lambda frames: {"pixel_values": []})})()
    clip_frames = [g.randint(0, 256, size=s).astype(np.uint8) for s in ((60, 80, 3), (500, 400, 3), (336, 336, 3))]   # up-, down-scaling, identity
    ctx = enc.preprocess(list(clip_frames))["context_images"]
    assert len(ctx) == 4 and torch.equal(ctx[2], ctx[3])                           # three frames padded to num_frames by repeating the last
    out["clip_pre_sub"] = torch.stack(ctx)[:, :, ::7, ::7]
    out["clip_pre_mean"] = torch.stack(ctx).mean(dim=(2, 3))
    out["clip_pre_patch"] = torch.stack(ctx)[:, :, 100:132, 200:232]
    save("host_rows.npz", **out)


def gen_host_pv():
    """The reference's own chat.preprocess_vision (R/chat.py:402-489), type="video" and type="image", called the way R/chat.py:540-553 calls it:
    what sits at each of the five return positions.  The CLIP processor is built offline as in gen_host; the cv2-based InternVideo2 processor
    (cv2 absent) is a stand-in that returns one zero tensor per frame, so position 0 of the video case pins shape and position only."""
    ri.install()
    ri._mod("decord", VideoReader=None, cpu=None)                    # chat.py imports it at module level; load_video is not called
    from PIL import Image
    import torchvision.transforms.functional as tvf

    tvf.to_pil_image = Image.fromarray
    tvf.resize = lambda img, size: img.resize((size[1], size[0]), Image.BILINEAR)
    import model.segment_anything.utils.transforms as tr
    tr.resize, tr.to_pil_image = tvf.resize, tvf.to_pil_image
    import importlib
    chat = importlib.import_module("chat")
    from transformers import CLIPImageProcessor
    from utils.enc_preprocessors import EncPreprocessor_VideoGPTPlus
    from utils.sam_transforms import SAM_v2_Preprocess
    enc = EncPreprocessor_VideoGPTPlus.__new__(EncPreprocessor_VideoGPTPlus)
    enc.num_frames, enc.frame_resolution_iv, enc.frame_resolution_clip = 4, 224, 336
    enc.image_processor = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, do_resize=True, do_center_crop=True,
                                             do_normalize=True, do_rescale=True, do_convert_rgb=True, resample=3,
                                             image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711])
    enc.video_processor = type("NoIV2", (), {"preprocess": staticmethod(lambda frames: {"pixel_values": [torch.zeros(3, 224, 224) for _ in frames]})})()
    conv_generator = type("Conv", (), {"NUM_FRAMES": 4})()
    out = {}
    frames = [np.random.RandomState(20 + i).randint(0, 256, size=(48, 64, 3)).astype(np.uint8) for i in range(6)]     # 6 frames > NUM_FRAMES: sub-sampled
    ret = chat.preprocess_vision([list(frames)], type="video", enc_preprocessor=enc, sam_preprocessor=SAM_v2_Preprocess(),
                                 conv_generator=conv_generator, precision="fp32")
    assert len(ret) == 5
    out["video_pos0_shape"] = np.array(ret[0][0].shape)
    out["video_pos1_sub"], out["video_pos1_mean"] = ret[1][0][:, :, ::7, ::7], ret[1][0].mean(dim=(2, 3))
    out["video_pos2_sub"], out["video_pos2_mean"] = ret[2][0][:, :, ::16, ::16], ret[2][0].mean(dim=(2, 3))
    out["video_pos3"], out["video_pos4"] = np.array(ret[3]), np.array(ret[4])
    image = np.random.RandomState(31).randint(0, 256, size=(60, 80, 3)).astype(np.uint8)
    ret = chat.preprocess_vision([[image]], type="image", enc_preprocessor=enc, sam_preprocessor=SAM_v2_Preprocess(),
                                 conv_generator=conv_generator, precision="fp32")
    assert len(ret) == 5 and ret[1] is None
    out["image_pos0_sub"], out["image_pos0_mean"] = ret[0][0][:, :, ::7, ::7], ret[0][0].mean(dim=(2, 3))
    out["image_pos2_sub"], out["image_pos2_mean"] = ret[2][0][:, :, ::16, ::16], ret[2][0].mean(dim=(2, 3))
    out["image_pos3"], out["image_pos4"] = np.array(ret[3]), np.array(ret[4])
    # ConvGenerator_VideoGPTPlus.apply_for_chat(type='image' / 'video', use_mm_start_end False / True) on the toy tokenizer
    from utils.conv_generator import ConvGenerator_VideoGPTPlus
    for base in ("phi3", "llama3_1"):
        for mm in (False, True):
            cg = ConvGenerator_VideoGPTPlus(use_mm_start_end=mm, base_type=base)
            cg.NUM_FRAMES = 4
            for kind in ("video", "image"):
                out[f"chat_ids_{base}_{int(mm)}_{kind}"] = cg.apply_for_chat("Please segment the red car .", type=kind, tokenizer=ToyTokenizer()).numpy()
    save("host_pv.npz", **out)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("host", "all")):
    gen_host()

if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("host_pv", "all")):
    gen_host_pv()


def gen_postproc():
    """mask post-processing / evaluation metrics: outputs of the reference's OWN functions on seeded masks.
    compute_iou / compute_miou (eval_gcg_metrics.py), db_eval_iou / _seg2bmap (eval_referdavis_metrics.py) are plain numpy
    and run here; remove_small_blobs / f_measure need scikit-image + OpenCV (absent) and stay unpinned (oracle/postproc.py)."""
    import math

    from oracle import postproc as op

    ns = {"np": np, "math": math}
    compute_iou, compute_miou = ri.functions("eval_gcg_metrics.py", ["compute_iou", "compute_miou"], ns)
    db_eval_iou, seg2bmap = ri.functions("eval_referdavis_metrics.py", ["db_eval_iou", "_seg2bmap"], ns)
    T, H, W = 3, 46, 83                       # odd width, not a multiple of the 64-pixel wave segment
    pred = np.stack([op.blobs((T, H, W), 100 + i, density=0.35 + 0.1 * i) for i in range(3)])     # [P,T,H,W]
    gt = np.stack([op.blobs((T, H, W), 200 + i, density=0.4) for i in range(2)])                   # [G,T,H,W]
    gt[1] = pred[0] ^ op.blobs((T, H, W), 300, density=0.05, smooth=1)                              # one good match
    pred[2, 1] = False                                                                             # an empty frame
    gt[0, 1] = False
    iou = np.array([[compute_iou(p, g) for g in gt] for p in pred])
    miou = compute_miou(list(pred), list(gt))
    jac = np.stack([np.asarray(db_eval_iou(gt[j], pred[i]), np.float64) for i in range(3) for j in range(2)]).reshape(3, 2, T)
    bmap = np.stack([np.stack([seg2bmap(pred[i, t].copy()) for t in range(T)]) for i in range(3)])
    save("postproc.npz", pred=pred.astype(np.uint8), gt=gt.astype(np.uint8), iou=iou, miou=np.float64(miou), jaccard=jac,
         bmap=bmap.astype(np.uint8))


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("postproc", "all")):
    gen_postproc()


def gen_ingest():
    """checkpoint-ingest helpers: the reference's OWN interpolate_pos_embed_internvideo2_new
    (R/model/videogpt_plus/model/internvideo/pos_embed.py:247-307) on a seeded positional embedding: 8 frames of a 4x4 grid ->
    4 frames of a 6x6 grid (temporal linear + spatial bicubic), and the temporal-only case 8 -> 4."""
    import logging
    import types
    ns = {"torch": torch, "logger": logging.getLogger("golden")}
    (interp,) = ri.functions("model/videogpt_plus/model/internvideo/pos_embed.py", ["interpolate_pos_embed_internvideo2_new"], ns)
    C = 8
    pos = rnd((1, 1 + 8 * 16, C), 61)
    out = {"pos_in": pos}
    for tag, side in (("t_and_s", 6), ("t_only", 4)):
        model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=4 * side * side), pos_embed=torch.zeros(1, 1 + 4 * side * side, C),
                                      num_frames=4, tubelet_size=1)
        ck = {"pos_embed": pos.clone()}
        interp(ck, model, orig_t_size=8)
        out["pos_" + tag] = ck["pos_embed"]
    save("ingest.npz", **out)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("ingest", "all")):
    gen_ingest()
