"""Checkpoint ingest of the RELEASED artefact layout: what the reference assembles at start-up from
four places is read here into one {reference state-dict name: tensor} dict + the architecture config `model.py` takes.

  1. the HF model directory (`--llava_version_or_path`): LLM + `mm_projector` / `image_mm_projector` + `text_hidden_fcs`
     + SAM2 under `model.visual_model.*`, as *.safetensors or pytorch_model*.bin shards, with the HF `config.json`
     — `VideoGLaMMForCausalLM.from_pretrained`, R/chat.py:277-284;
  2. the InternVideo2 stage-2 checkpoint (`mm_vision_tower`, a torch .pt holding {"model" | "module": state_dict} with
     the text tower and heads beside `vision_encoder.*`) — setup_internvideo2V, R/model/videogpt_plus/model/internvideo/
     utils.py:60-91 (load_state_dict(strict=False): only `vision_encoder.*` lands; num_frames == origin_num_frames == 4,
     so its positional-embedding interpolation is the identity);
  3. the CLIP directory (`image_mm_vision_tower`, openai/clip-vit-large-patch14-336: `vision_model.*` beside the text
     tower) — CLIPVisionTower.load_model, R/model/videogpt_plus/model/multimodal_encoder/clip_encoder.py;
  4. optionally a stand-alone SAM2 checkpoint ({"model": state_dict}) — _load_checkpoint,
     R/model/segment_anything_2/sam2/build_sam.py:92-112.

Tensors stay on the host here; `params.Params` packs them for the kernels (the `.gamma -> .weight` rename of the memory
fuser's LayerScale is applied there).  One-time load work, outside the hot path.
"""
import glob
import json
import os

import torch

from . import synth

IV2_PREFIX = "model.vision_tower.vision_encoder."
CLIP_PREFIX = "model.image_vision_tower.vision_tower."
SAM2_PREFIX = "model.visual_model."


def _read(path):
    """one checkpoint file -> flat {name: tensor}."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("model", "module", "state_dict"):       # internvideo/utils.py:75-81, build_sam.py:96
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            return obj[key]
    return obj


def _read_dir(path):
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
    sd = {}
    for f in files:
        sd.update(_read(f))
    return sd


def _read_any(path):
    return _read_dir(path) if os.path.isdir(path) else _read(path)


def load_state_dict(model_dir, vision_tower=None, image_vision_tower=None, sam2_checkpoint=None, lora_dir=None, iv2_origin_num_frames=None):
    """-> ({reference name: tensor}, hf_config dict or None).  vision_tower / image_vision_tower default to the paths the
    HF config names (`mm_vision_tower`, `image_mm_vision_tower`) when the directory does not already carry the towers.
    lora_dir: a LoRA training output (adapter + non_lora_trainables.bin) merged over the base like
    load_videogptplus_model_from_pretrained does; iv2_origin_num_frames: frame count the InternVideo2 checkpoint was trained with
    (its pos_embed is interpolated to the model's 4 frames when it differs — the released config has 4 == 4)."""
    sd = _read_dir(model_dir)
    hf = None
    cfg_path = os.path.join(model_dir, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path) as fh:
            hf = json.load(fh)
        gen_path = os.path.join(model_dir, "generation_config.json")     # HF generate() stops on generation_config's eos ids
        if os.path.exists(gen_path):
            with open(gen_path) as fh:
                hf["_generation_eos_token_id"] = json.load(fh).get("eos_token_id")
    if not any(k.startswith(IV2_PREFIX) for k in sd):
        vision_tower = vision_tower or (hf or {}).get("mm_vision_tower")
        if not vision_tower:
            raise FileNotFoundError("the model directory has no InternVideo2 weights and no vision_tower path was given")
        iv2 = {k: v for k, v in _read_any(vision_tower).items() if k.startswith("vision_encoder.")}
        if not iv2:
            raise KeyError(f"{vision_tower}: no vision_encoder.* tensors")
        origin = iv2_origin_num_frames if iv2_origin_num_frames is not None else (hf or {}).get("iv2_origin_num_frames")
        if origin and int(origin) != 4:
            # every *pos_embed / clip_pos_embed tensor but img_pos_embed, like interpolate_pos_embed_internvideo2_new's key loop
            # (pos_embed.py:248-252); time only: the patch grid of the checkpoint is the model's (224 / 14 = 16 per side)
            for k in [k for k in iv2 if "pos_embed" in k and "img_pos_embed" not in k]:
                pe = iv2[k]
                grid = int(round(((pe.shape[1] - 1) // int(origin)) ** 0.5))
                iv2[k] = interpolate_iv2_pos_embed(pe, int(origin), 4, grid)
        sd.update({"model.vision_tower." + k: v for k, v in iv2.items()})
    if not any(k.startswith(CLIP_PREFIX) for k in sd):
        image_vision_tower = image_vision_tower or (hf or {}).get("image_mm_vision_tower")
        if not image_vision_tower:
            raise FileNotFoundError("the model directory has no CLIP weights and no image_vision_tower path was given")
        clip = {k: v for k, v in _read_any(image_vision_tower).items() if k.startswith("vision_model.")}
        if not clip:
            raise KeyError(f"{image_vision_tower}: no vision_model.* tensors")
        sd.update({CLIP_PREFIX + k: v for k, v in clip.items()})
    if sam2_checkpoint:
        sd.update({SAM2_PREFIX + k: v for k, v in _read_any(sam2_checkpoint).items()})
    if not any(k.startswith(SAM2_PREFIX) for k in sd):
        raise FileNotFoundError("the model directory has no SAM2 weights (model.visual_model.*) and no sam2_checkpoint was given")
    if lora_dir:
        nl = os.path.join(lora_dir, "non_lora_trainables.bin")
        if not os.path.exists(nl):     # the reference's own error (train_ds_with_videogptplus.py:166)
            raise FileNotFoundError("Lora is specified in the model path, however could not find non-LORA trainables weights.")
        merge_non_lora_trainables(sd, nl)
        merge_lora(sd, lora_dir)
    return sd, hf


def interpolate_iv2_pos_embed(pos, orig_t, new_t, new_side, num_extra=1):
    """interpolate_pos_embed_internvideo2_new for one tensor — R/model/videogpt_plus/model/internvideo/pos_embed.py:247-307, called
    from setup_internvideo2V (internvideo/utils.py:83-88) when the checkpoint's frame count / patch grid differs from the model's:
    pos [1, extra + orig_t * s * s, C] -> [1, extra + new_t * new_side^2, C]; linear along time first, then bicubic
    (align_corners=False) over the patch grid; the extra (cls) tokens are kept."""
    C = pos.shape[-1]
    extra, tok = pos[:, :num_extra], pos[:, num_extra:]
    side = int(round((tok.shape[1] // orig_t) ** 0.5))
    assert orig_t * side * side == tok.shape[1], f"pos_embed of {tok.shape[1]} tokens is not {orig_t} frames of a square grid"
    dt = pos.dtype
    tok = tok.float()
    if orig_t != new_t:
        t = tok.view(1, orig_t, -1, C).permute(0, 2, 3, 1).reshape(-1, C, orig_t)
        t = torch.nn.functional.interpolate(t, size=new_t, mode="linear")
        tok = t.view(1, -1, C, new_t).permute(0, 3, 1, 2).reshape(1, -1, C)
    if side != new_side:
        t = tok.reshape(-1, side, side, C).permute(0, 3, 1, 2)
        t = torch.nn.functional.interpolate(t, size=(new_side, new_side), mode="bicubic", align_corners=False)
        tok = t.permute(0, 2, 3, 1).reshape(-1, new_t, new_side, new_side, C).flatten(1, 3)
    return torch.cat((extra.float(), tok), dim=1).to(dt)


def merge_non_lora_trainables(sd, path):
    """non_lora_trainables.bin of a LoRA training run laid over the base state dict — load_videogptplus_model_from_pretrained,
    R/train_ds_with_videogptplus.py:163-171: keys lose a leading "base_model." and, when any key starts with "model.model.", a
    leading "model."; load_state_dict(strict=False) = known names are overwritten, unknown ones ignored.  -> number of tensors taken."""
    extra = _read(path)
    extra = {(k[11:] if k.startswith("base_model.") else k): v for k, v in extra.items()}
    if any(k.startswith("model.model.") for k in extra):
        extra = {(k[6:] if k.startswith("model.") else k): v for k, v in extra.items()}
    n = 0
    for k, v in extra.items():
        if k in sd:
            assert tuple(sd[k].shape) == tuple(v.shape), f"{k}: {tuple(v.shape)} does not fit {tuple(sd[k].shape)}"
            sd[k] = v
            n += 1
    return n


def merge_lora(sd, lora_dir):
    """PeftModel.from_pretrained(model, dir).merge_and_unload() on a state dict (R/train_ds_with_videogptplus.py:173-178; peft==0.12.0,
    R/requirements.txt:33 — source absent here: the published LoRA merge W += (lora_alpha / r) * B @ A, "parity unpinned").
    Reads adapter_config.json (r, lora_alpha; use_rslora -> alpha / sqrt(r)) and adapter_model.{safetensors,bin} whose keys are
    base_model.model.<module>.lora_A.weight / lora_B.weight.  -> number of weights merged."""
    with open(os.path.join(lora_dir, "adapter_config.json")) as fh:
        ac = json.load(fh)
    r, alpha = int(ac["r"]), float(ac["lora_alpha"])
    scaling = alpha / (r ** 0.5) if ac.get("use_rslora") else alpha / r
    f = next((os.path.join(lora_dir, n) for n in ("adapter_model.safetensors", "adapter_model.bin") if os.path.exists(os.path.join(lora_dir, n))), None)
    if f is None:
        raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin under {lora_dir}")
    ad = _read(f)
    n = 0
    for k, a in ad.items():
        if not k.endswith("lora_A.weight") and ".lora_A." not in k:
            continue
        kb = k.replace("lora_A", "lora_B")
        mod = k.split(".lora_A")[0]
        mod = mod[len("base_model.model."):] if mod.startswith("base_model.model.") else mod
        name = mod + ".weight"
        if name not in sd:
            raise KeyError(f"LoRA adapter targets {name}, which the base checkpoint does not have")
        w = sd[name]
        delta = (ad[kb].float() @ a.float()) * scaling
        assert tuple(delta.shape) == tuple(w.shape), f"{name}: LoRA delta {tuple(delta.shape)} vs weight {tuple(w.shape)}"
        sd[name] = (w.float() + delta).to(w.dtype)
        n += 1
    return n


def _count(sd, prefix, pattern):
    """number of distinct indices i of keys prefix + pattern.format(i) + ..."""
    idx = set()
    head = prefix + pattern
    for k in sd:
        if k.startswith(head):
            idx.add(int(k[len(head):].split(".")[0]))
    return len(idx)


def derive_config(sd, hf=None, seg_token_idx=None):
    """architecture config for model.VideoGLaMMForCausalLM from the HF config.json (LLM hyper-parameters) and the tensor
    SHAPES (depths, widths).  Head counts and the SAM2 trunk layout are not recoverable from shapes: they come from the
    presets of the released composition (InternVideo2-1B, CLIP-L/336, SAM2 hiera-L / hiera-T) and are checked against the
    widths; anything that does not fit raises instead of guessing."""
    hf = hf or {}
    emb = sd["model.embed_tokens.weight"]
    D = emb.shape[1]
    fused = "model.layers.0.self_attn.qkv_proj.weight" in sd
    n_layers = _count(sd, "model.", "layers.")
    heads = hf.get("num_attention_heads")
    if heads is None:
        raise KeyError("config.json lacks num_attention_heads: the head count cannot be read off the tensor shapes")
    kv = hf.get("num_key_value_heads", heads)
    hd = D // heads
    ffn = sd["model.layers.0.mlp.down_proj.weight"].shape[1]
    if fused:
        assert sd["model.layers.0.self_attn.qkv_proj.weight"].shape[0] == (heads + 2 * kv) * hd, "qkv_proj rows != (H + 2 Hkv) * head_dim"
    else:
        assert sd["model.layers.0.self_attn.k_proj.weight"].shape[0] == kv * hd, "k_proj rows != Hkv * head_dim"
    llm = dict(vocab=emb.shape[0], hidden=D, ffn=ffn, num_layers=n_layers, num_heads=heads, num_kv_heads=kv,
               rms_eps=float(hf.get("rms_norm_eps", 1e-5)), rope_theta=float(hf.get("rope_theta", 10000.0)))
    if fused:
        llm["fused_proj"] = True
    if hf.get("sliding_window"):
        llm["sliding_window"] = int(hf["sliding_window"])
    assert hf.get("num_hidden_layers", n_layers) == n_layers, "config.json num_hidden_layers != layers in the checkpoint"

    iv2_w = sd[IV2_PREFIX + "pos_embed"].shape[-1]
    iv2 = dict(synth.IV2_1B, embed_dim=iv2_w, depth=_count(sd, IV2_PREFIX, "blocks."),
               patch_size=sd[IV2_PREFIX + "patch_embed.proj.weight"].shape[-1], mlp_hidden=sd[IV2_PREFIX + "blocks.0.mlp.fc1.weight"].shape[0])
    iv2["num_heads"] = int(hf.get("iv2_num_heads", synth.IV2_1B["num_heads"]))
    assert iv2_w % iv2["num_heads"] == 0, f"InternVideo2 width {iv2_w} is not divisible by {iv2['num_heads']} heads"

    v = CLIP_PREFIX + ("vision_model." if CLIP_PREFIX + "vision_model.embeddings.class_embedding" in sd else "")
    clip_w = sd[v + "embeddings.class_embedding"].shape[0]
    patch = sd[v + "embeddings.patch_embedding.weight"].shape[-1]
    n_pos = sd[v + "embeddings.position_embedding.weight"].shape[0]
    clip = dict(hidden=clip_w, patch_size=patch, img_size=int(round((n_pos - 1) ** 0.5)) * patch, mlp=sd[v + "encoder.layers.0.mlp.fc1.weight"].shape[0],
                num_layers=int(hf.get("clip_num_layers", _count(sd, v, "encoder.layers."))), num_heads=int(hf.get("clip_num_heads", synth.CLIP_L_336["num_heads"])))
    assert clip_w % clip["num_heads"] == 0

    t = SAM2_PREFIX + "image_encoder.trunk."
    trunk_w = sd[t + "patch_embed.proj.weight"].shape[0]
    n_blocks = _count(sd, t, "blocks.")
    preset = next((p for p in (synth.SAM2_L, synth.SAM2_T) if p["trunk"]["embed_dim"] == trunk_w and sum(p["trunk"]["stages"]) == n_blocks), None)
    if "sam2" in hf:
        sam2 = hf["sam2"]
    elif preset is not None:
        sam2 = dict(image_size=preset["image_size"], trunk=dict(preset["trunk"]))
    else:
        raise KeyError(f"SAM2 trunk (width {trunk_w}, {n_blocks} blocks) matches no preset: pass the trunk layout under config.json['sam2']")

    depth = 1 if "model.mm_projector.weight" in sd else len({k.split(".")[2] for k in sd if k.startswith("model.mm_projector.")})
    if seg_token_idx is None:
        seg_token_idx = hf.get("seg_token_idx", emb.shape[0] - 1)     # "[SEG]" is the last added token (R/chat.py:297-300)
    cfg = dict(seg_token_idx=int(seg_token_idx), iv2=iv2, clip=clip, llm=llm, sam2=sam2, projector_depth=depth)
    # HF generate() stops on generation_config.json's eos ids when that file has them, on config.json's otherwise
    gen_eos = eos_ids(hf.get("_generation_eos_token_id"))
    eos = gen_eos or eos_ids(hf.get("eos_token_id"))
    if eos:
        cfg["eos_token_id"] = eos
        cfg["eos_from_generation_config"] = bool(gen_eos)
    for k in ("bos_token_id", "pad_token_id"):
        if hf.get(k) is not None:
            cfg[k] = hf[k]
    return cfg


def eos_ids(*sources):
    """int / list / None values -> sorted list of distinct EOS ids (stock Phi-3-mini: generation_config eos = [32000, 32001, 32007])."""
    out = set()
    for s in sources:
        if s is None:
            continue
        out.update([int(s)] if isinstance(s, int) else [int(x) for x in s])
    return sorted(out)
