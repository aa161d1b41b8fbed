"""Synthetic checkpoints: the {name: shape} manifest of a VideoGLaMM checkpoint for a given architecture
config (reference state-dict naming) and random-init weights generated directly on the device.

There is no network for real checkpoints (and no Llama-based VideoGLaMM checkpoint exists publicly —
R/model/VideoGLaMM.py:882, R/chat.py:279-285), so benchmarks and smoke tests run on these.  The manifest is
checked against manifests dumped from the reference's own nn.Modules (tests/test_synth.py).
"""
import math
import zlib

import torch

from .sam2 import hiera_layout

# ---- architecture presets -------------------------------------------------------------------------------
SAM2_L = dict(image_size=1024, trunk=dict(embed_dim=144, num_heads=2, stages=[2, 6, 36, 4], global_att_blocks=[23, 33, 43],
                                          window_pos_embed_bkg_spatial_size=[7, 7], window_spec=[8, 4, 16, 8]))
SAM2_T = dict(image_size=1024, trunk=dict(embed_dim=96, num_heads=1, stages=[1, 2, 7, 2], global_att_blocks=[5, 7, 9],
                                          window_pos_embed_bkg_spatial_size=[7, 7], window_spec=[8, 4, 14, 7]))
IV2_1B = dict(img_size=224, patch_size=14, embed_dim=1408, depth=40, num_heads=16, mlp_hidden=6144)
CLIP_L_336 = dict(img_size=336, patch_size=14, hidden=1024, mlp=4096, num_layers=24, num_heads=16)
LLAMA3_8B = dict(vocab=128257, hidden=4096, ffn=14336, num_layers=32, num_heads=32, num_kv_heads=8, rms_eps=1e-5, rope_theta=500000.0)


# the LLM of the released VideoGLaMM checkpoint (microsoft/Phi-3-mini-4k-instruct + [SEG], R/chat.py:31): fused
# qkv_proj / gate_up_proj tensors, MHA with head_dim 96, sliding window 2047
PHI3_MINI = dict(vocab=32065, hidden=3072, ffn=8192, num_layers=32, num_heads=32, num_kv_heads=32, rms_eps=1e-5, rope_theta=10000.0,
                 sliding_window=2047, fused_proj=True)


def videoglamm_phi3_mini():
    """the released composition: Phi-3-mini (+[SEG]) + InternVideo2-1B + CLIP-L/336 + SAM2-L."""
    return dict(seg_token_idx=32064, iv2=IV2_1B, clip=CLIP_L_336, llm=PHI3_MINI, sam2=SAM2_L, projector_depth=2)


def videoglamm_llama3_8b():
    """BASELINE configs C1-C3: Llama-3-8B (+[SEG]) + InternVideo2-1B + CLIP-L/336 + SAM2-L, mlp2x_gelu adapters."""
    return dict(seg_token_idx=128256, iv2=IV2_1B, clip=CLIP_L_336, llm=LLAMA3_8B, sam2=SAM2_L, projector_depth=2)


# ---- manifests ------------------------------------------------------------------------------------------
def _lin(m, name, n, k, bias=True):
    m[name + ".weight"] = [n, k]
    if bias:
        m[name + ".bias"] = [n]


def _ln(m, name, c):
    m[name + ".weight"] = [c]
    m[name + ".bias"] = [c]


def sam2_manifest(cfg, p=""):
    m = {}
    tr = cfg["trunk"]
    blocks, stage_ends = hiera_layout(tr)
    t = p + "image_encoder.trunk."
    m[t + "pos_embed"] = [1, tr["embed_dim"], *tr["window_pos_embed_bkg_spatial_size"]]
    m[t + "pos_embed_window"] = [1, tr["embed_dim"], tr["window_spec"][0], tr["window_spec"][0]]
    m[t + "patch_embed.proj.weight"] = [tr["embed_dim"], 3, 7, 7]
    m[t + "patch_embed.proj.bias"] = [tr["embed_dim"]]
    for i, b in enumerate(blocks):
        q = f"{t}blocks.{i}."
        _ln(m, q + "norm1", b["dim"])
        _lin(m, q + "attn.qkv", 3 * b["dim_out"], b["dim"])
        _lin(m, q + "attn.proj", b["dim_out"], b["dim_out"])
        _ln(m, q + "norm2", b["dim_out"])
        _lin(m, q + "mlp.layers.0", 4 * b["dim_out"], b["dim_out"])
        _lin(m, q + "mlp.layers.1", b["dim_out"], 4 * b["dim_out"])
        if b["dim"] != b["dim_out"]:
            _lin(m, q + "proj", b["dim_out"], b["dim"])
    chans = [blocks[i]["dim_out"] for i in stage_ends[::-1]]
    for j, c in enumerate(chans):
        m[f"{p}image_encoder.neck.convs.{j}.conv.weight"] = [256, c, 1, 1]
        m[f"{p}image_encoder.neck.convs.{j}.conv.bias"] = [256]
    ma = p + "memory_attention."
    for i in range(4):
        l = f"{ma}layers.{i}."
        for a, kin in (("self_attn", 256), ("cross_attn_image", 64)):
            _lin(m, l + a + ".q_proj", 256, 256)
            _lin(m, l + a + ".k_proj", 256, kin)
            _lin(m, l + a + ".v_proj", 256, kin)
            _lin(m, l + a + ".out_proj", 256, 256)
        _lin(m, l + "linear1", 2048, 256)
        _lin(m, l + "linear2", 256, 2048)
        for n in ("norm1", "norm2", "norm3"):
            _ln(m, l + n, 256)
    _ln(m, ma + "norm", 256)
    me = p + "memory_encoder."
    cin = 1
    for i in range(4):
        cout = cin * 4
        m[f"{me}mask_downsampler.encoder.{3 * i}.weight"] = [cout, cin, 3, 3]
        m[f"{me}mask_downsampler.encoder.{3 * i}.bias"] = [cout]
        _ln(m, f"{me}mask_downsampler.encoder.{3 * i + 1}", cout)
        cin = cout
    m[me + "mask_downsampler.encoder.12.weight"] = [256, 256, 1, 1]
    m[me + "mask_downsampler.encoder.12.bias"] = [256]
    m[me + "pix_feat_proj.weight"] = [256, 256, 1, 1]
    m[me + "pix_feat_proj.bias"] = [256]
    for i in range(2):
        l = f"{me}fuser.layers.{i}."
        m[l + "weight"] = [256]
        m[l + "dwconv.weight"] = [256, 1, 7, 7]
        m[l + "dwconv.bias"] = [256]
        _ln(m, l + "norm", 256)
        _lin(m, l + "pwconv1", 1024, 256)
        _lin(m, l + "pwconv2", 256, 1024)
    m[me + "out_proj.weight"] = [64, 256, 1, 1]
    m[me + "out_proj.bias"] = [64]
    m[p + "maskmem_tpos_enc"] = [7, 1, 1, 64]
    m[p + "no_mem_embed"] = [1, 1, 256]
    m[p + "no_mem_pos_enc"] = [1, 1, 256]
    m[p + "no_obj_ptr"] = [1, 256]
    m[p + "mask_downsample.weight"] = [1, 1, 4, 4]
    m[p + "mask_downsample.bias"] = [1]
    pe = p + "sam_prompt_encoder."
    m[pe + "pe_layer.positional_encoding_gaussian_matrix"] = [2, 128]
    for i in range(4):
        m[f"{pe}point_embeddings.{i}.weight"] = [1, 256]
    m[pe + "not_a_point_embed.weight"] = [1, 256]
    m[pe + "mask_downscaling.0.weight"] = [4, 1, 2, 2]
    m[pe + "mask_downscaling.0.bias"] = [4]
    _ln(m, pe + "mask_downscaling.1", 4)
    m[pe + "mask_downscaling.3.weight"] = [16, 4, 2, 2]
    m[pe + "mask_downscaling.3.bias"] = [16]
    _ln(m, pe + "mask_downscaling.4", 16)
    m[pe + "mask_downscaling.6.weight"] = [256, 16, 1, 1]
    m[pe + "mask_downscaling.6.bias"] = [256]
    m[pe + "no_mask_embed.weight"] = [1, 256]
    d = p + "sam_mask_decoder."
    tt = d + "transformer."

    def att(name, internal):
        _lin(m, name + ".q_proj", internal, 256)
        _lin(m, name + ".k_proj", internal, 256)
        _lin(m, name + ".v_proj", internal, 256)
        _lin(m, name + ".out_proj", 256, internal)

    for i in range(2):
        l = f"{tt}layers.{i}."
        att(l + "self_attn", 256)
        att(l + "cross_attn_token_to_image", 128)
        att(l + "cross_attn_image_to_token", 128)
        _lin(m, l + "mlp.layers.0", 2048, 256)
        _lin(m, l + "mlp.layers.1", 256, 2048)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            _ln(m, l + n, 256)
    att(tt + "final_attn_token_to_image", 128)
    _ln(m, tt + "norm_final_attn", 256)
    m[d + "iou_token.weight"] = [1, 256]
    m[d + "mask_tokens.weight"] = [4, 256]
    m[d + "obj_score_token.weight"] = [1, 256]
    m[d + "output_upscaling.0.weight"] = [256, 64, 2, 2]
    m[d + "output_upscaling.0.bias"] = [64]
    _ln(m, d + "output_upscaling.1", 64)
    m[d + "output_upscaling.3.weight"] = [64, 32, 2, 2]
    m[d + "output_upscaling.3.bias"] = [32]
    m[d + "conv_s0.weight"] = [32, 256, 1, 1]
    m[d + "conv_s0.bias"] = [32]
    m[d + "conv_s1.weight"] = [64, 256, 1, 1]
    m[d + "conv_s1.bias"] = [64]
    for i in range(4):
        for j, (n, k) in enumerate(((256, 256), (256, 256), (32, 256))):
            _lin(m, f"{d}output_hypernetworks_mlps.{i}.layers.{j}", n, k)
    for j, (n, k) in enumerate(((256, 256), (256, 256), (4, 256))):
        _lin(m, f"{d}iou_prediction_head.layers.{j}", n, k)
    for j, (n, k) in enumerate(((256, 256), (256, 256), (1, 256))):
        _lin(m, f"{d}pred_obj_score_head.layers.{j}", n, k)
    for j in range(3):
        _lin(m, f"{p}obj_ptr_proj.layers.{j}", 256, 256)
    return m


def vlm_manifest(cfg):
    m = {}
    c = cfg["iv2"]
    p = "model.vision_tower.vision_encoder."
    C, L = c["embed_dim"], (c["img_size"] // c["patch_size"]) ** 2
    hid = c.get("mlp_hidden", int(C * c.get("mlp_ratio", 4)))
    m[p + "patch_embed.proj.weight"] = [C, 3, 1, c["patch_size"], c["patch_size"]]
    m[p + "patch_embed.proj.bias"] = [C]
    m[p + "cls_token"] = [1, 1, C]
    m[p + "pos_embed"] = [1, 1 + 4 * L, C]
    for i in range(c["depth"] - 1):  # the last block never runs on this path; it is not materialised
        b = f"{p}blocks.{i}."
        m[b + "norm1.weight"] = [C]
        m[b + "attn.qkv.weight"] = [3 * C, C]
        m[b + "attn.q_norm.weight"] = [C]
        m[b + "attn.k_norm.weight"] = [C]
        _lin(m, b + "attn.proj", C, C)
        m[b + "ls1.gamma"] = [C]
        m[b + "norm2.weight"] = [C]
        _lin(m, b + "mlp.fc1", hid, C)
        _lin(m, b + "mlp.fc2", C, hid)
        m[b + "ls2.gamma"] = [C]
    c = cfg["clip"]
    v = "model.image_vision_tower.vision_tower.vision_model."
    C, L = c["hidden"], (c["img_size"] // c["patch_size"]) ** 2
    m[v + "embeddings.class_embedding"] = [C]
    m[v + "embeddings.patch_embedding.weight"] = [C, 3, c["patch_size"], c["patch_size"]]
    m[v + "embeddings.position_embedding.weight"] = [L + 1, C]
    _ln(m, v + "pre_layrnorm", C)
    for i in range(c["num_layers"] - 1):  # hidden_states[-2]: the last layer is never read
        b = f"{v}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(m, b + "self_attn." + n, C, C)
        _ln(m, b + "layer_norm1", C)
        _ln(m, b + "layer_norm2", C)
        _lin(m, b + "mlp.fc1", c["mlp"], C)
        _lin(m, b + "mlp.fc2", C, c["mlp"])
    c = cfg["llm"]
    D, hd = c["hidden"], c["hidden"] // c["num_heads"]
    for name, kin in (("model.mm_projector", cfg["iv2"]["embed_dim"]), ("model.image_mm_projector", cfg["clip"]["hidden"])):
        depth = cfg.get("projector_depth", 2)
        if depth == 1:
            _lin(m, name, D, kin)
        else:
            _lin(m, name + ".0", D, kin)
            for j in range(1, depth):
                _lin(m, f"{name}.{2 * j}", D, D)
    m["model.embed_tokens.weight"] = [c["vocab"], D]
    for i in range(c["num_layers"]):
        l = f"model.layers.{i}."
        if c.get("fused_proj"):      # Phi-3 checkpoint layout
            m[l + "self_attn.qkv_proj.weight"] = [(c["num_heads"] + 2 * c["num_kv_heads"]) * hd, D]
            m[l + "mlp.gate_up_proj.weight"] = [2 * c["ffn"], D]
        else:
            m[l + "self_attn.q_proj.weight"] = [c["num_heads"] * hd, D]
            m[l + "self_attn.k_proj.weight"] = [c["num_kv_heads"] * hd, D]
            m[l + "self_attn.v_proj.weight"] = [c["num_kv_heads"] * hd, D]
            m[l + "mlp.gate_proj.weight"] = [c["ffn"], D]
            m[l + "mlp.up_proj.weight"] = [c["ffn"], D]
        m[l + "self_attn.o_proj.weight"] = [D, c["num_heads"] * hd]
        m[l + "mlp.down_proj.weight"] = [D, c["ffn"]]
        m[l + "input_layernorm.weight"] = [D]
        m[l + "post_attention_layernorm.weight"] = [D]
    m["model.norm.weight"] = [D]
    m["lm_head.weight"] = [c["vocab"], D]
    _lin(m, "model.text_hidden_fcs.0.0", D, D)
    _lin(m, "model.text_hidden_fcs.0.2", 256, D)
    return m


def manifest(cfg):
    m = vlm_manifest(cfg)
    m.update(sam2_manifest(cfg["sam2"], "model.visual_model."))
    return m


# ---- random-init weights on the device ------------------------------------------------------------------------
def device_state_dict(man, device, dtype, seed=0):
    """Same distribution family as oracle/seeded.py (fan-in scaled normals), generated on the device."""
    sd = {}
    for name, shape in man.items():
        g = torch.Generator(device=device).manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32).mul_(1.0 / math.sqrt(max(fan_in, 1)))
        elif name.endswith(".weight"):
            t = 1.0 + 0.05 * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        else:
            t = 0.1 * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        keep32 = len(shape) < 2 or name.endswith("positional_encoding_gaussian_matrix")
        sd[name] = t if keep32 else t.to(dtype)
    k = "model.visual_model.sam_mask_decoder.pred_obj_score_head.layers.2.bias"
    if k in sd:
        sd[k] = torch.full_like(sd[k], 4.0)  # random-init SAM2 otherwise predicts "no object" everywhere
    return sd


# ---- decode-time token hook for synthetic weights ------------------------------------------------------
def forced_tokens_hook(mapping):
    """{decode step: token id} -> a VideoGLaMMForCausalLM.token_hook: random-init weights never emit [SEG], so benchmarks and whole-workload
    tests replace the emitted token at fixed steps (after the full lm_head + argmax: no work is skipped) — and parity tests teacher-force a
    second model to the first one's ids the same way.  None / {} -> None (plain greedy decoding)."""
    if not mapping:
        return None
    table = {int(k): int(v) for k, v in mapping.items()}

    def hook(step, tok):
        return table.get(step)
    hook.forced_table = table      # vlm.generate applies a table-carrying hook on the device (LlamaDecoder.set_forced): no host round trip per token
    return hook


def install_forced_tokens(model, mapping=None):
    """model.token_hook = forced_tokens_hook(mapping); mapping defaults to the harness's cfg["forced_tokens"] entry.  Returns the model."""
    model.token_hook = forced_tokens_hook(model.cfg.get("forced_tokens") if mapping is None else mapping)
    return model
