"""ORACLE (test infrastructure, never shipped in the product path): CPU fp32 restatement of the
LLM side of the VideoGLaMM hot path — InternVideo2 video tower, CLIP ViT image tower, V-L adapters with
pooling, embedding splice, Llama decoder, greedy decode, [SEG] extraction and the L-V adapter.
R/ = /root/reference/VideoGLaMM/.  Third-party arithmetic (HF transformers==4.41.0: LlamaModel,
CLIPVisionModel, GenerationMixin.generate greedy search) is restated from its published algorithm and
pinned by golden vectors produced by running the reference + HF in the build container
(tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200  # R/model/videogpt_plus/constants.py


def lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


# ----------------------------------------------------------------------------- L1 InternVideo2
def rms_norm(x, w, eps):
    """R/model/videogpt_plus/model/internvideo/internvideo2.py:134-145 (== HF LlamaRMSNorm)."""
    dt = x.dtype
    x = x.to(torch.float32)
    x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return w * x.to(dt)


def iv2_forward(sd, p, cfg, video):
    """InternVideo2_Stage2V.forward -> PretrainInternVideo2.forward(x_vis_return_idx=-2, x_vis_only=True)
    R/.../internvideo/utils.py:229-238 ; internvideo2.py:585-651,190-209,265-316.
    video: [nc, T=4, 3, H, W] -> [nc, 1 + T*L, C] (pre-final-norm tokens of block depth-2)."""
    depth, heads, ps = cfg["depth"], cfg["num_heads"], cfg["patch_size"]
    x = video.permute(0, 2, 1, 3, 4)  # [B,C,T,H,W]
    x = F.conv3d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=(1, ps, ps))
    x = x.flatten(3).permute(0, 2, 3, 1)  # B,T,L,C
    B, T, L, C = x.shape
    x = x.reshape(B, T * L, C)
    x = torch.cat((sd[p + "cls_token"].expand(B, -1, -1), x), dim=1) + sd[p + "pos_embed"]
    hd = C // heads
    for i in range(depth):
        b = f"{p}blocks.{i}."
        h = rms_norm(x, sd[b + "norm1.weight"], 1e-6)
        qkv = F.linear(h, sd[b + "attn.qkv.weight"]).reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        N = q.shape[2]
        q = rms_norm(q.transpose(1, 2).flatten(-2, -1), sd[b + "attn.q_norm.weight"], 1e-6).view(B, N, heads, hd).transpose(1, 2)
        k = rms_norm(k.transpose(1, 2).flatten(-2, -1), sd[b + "attn.k_norm.weight"], 1e-6).view(B, N, heads, hd).transpose(1, 2)
        a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        h = lin(sd, b + "attn.proj", (a @ v).transpose(1, 2).reshape(B, N, C))
        x = x + h.float() * sd[b + "ls1.gamma"].float()
        h = rms_norm(x, sd[b + "norm2.weight"], 1e-6)
        h = lin(sd, b + "mlp.fc2", F.gelu(lin(sd, b + "mlp.fc1", h)))
        x = x + h.float() * sd[b + "ls2.gamma"].float()
        if i == depth - 2:
            break
    return x


# ----------------------------------------------------------------------------- L2 CLIP ViT
def clip_forward(sd, p, cfg, images):
    """CLIPVisionTower.forward(select_feature='patch') over HF CLIPVisionModel, hidden_states[-2], CLS dropped
    R/.../multimodal_encoder/clip_encoder.py:34-72 ; HF modeling_clip.py (CLIPVisionEmbeddings, CLIPEncoderLayer).
    images: [T,3,H,W] -> [T, L, C]."""
    heads, layers, ps = cfg["num_heads"], cfg["num_layers"], cfg["patch_size"]
    # transformers==4.41 (reference pin) names the tensors "<tower>.vision_model.*"; 5.x dropped the level
    v = p + "vision_model." if (p + "vision_model.embeddings.class_embedding") in sd else p
    x = F.conv2d(images, sd[v + "embeddings.patch_embedding.weight"], stride=ps).flatten(2).transpose(1, 2)
    B, L, C = x.shape
    x = torch.cat([sd[v + "embeddings.class_embedding"].expand(B, 1, -1), x], dim=1) + sd[v + "embeddings.position_embedding.weight"]
    x = F.layer_norm(x, (C,), sd[v + "pre_layrnorm.weight"], sd[v + "pre_layrnorm.bias"], 1e-5)
    hd = C // heads
    for i in range(layers - 1):  # select_layer = -2: the last encoder layer's output is never used
        b = f"{v}encoder.layers.{i}."
        h = F.layer_norm(x, (C,), sd[b + "layer_norm1.weight"], sd[b + "layer_norm1.bias"], 1e-5)
        q = lin(sd, b + "self_attn.q_proj", h).view(B, -1, heads, hd).transpose(1, 2)
        k = lin(sd, b + "self_attn.k_proj", h).view(B, -1, heads, hd).transpose(1, 2)
        vv = lin(sd, b + "self_attn.v_proj", h).view(B, -1, heads, hd).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
        h = lin(sd, b + "self_attn.out_proj", (a @ vv).transpose(1, 2).reshape(B, -1, C))
        x = x + h
        h = F.layer_norm(x, (C,), sd[b + "layer_norm2.weight"], sd[b + "layer_norm2.bias"], 1e-5)
        h = lin(sd, b + "mlp.fc1", h)
        h = lin(sd, b + "mlp.fc2", h * torch.sigmoid(1.702 * h))
        x = x + h
    return x[:, 1:]


# ----------------------------------------------------------------------------- L3 adapters + pooling
def projector(sd, name, x):
    """build_vision_projector: 'linear' or 'mlpNx_gelu' — R/.../multimodal_projector/builder.py:17-54."""
    if name + ".weight" in sd:
        return lin(sd, name, x)
    i = 0
    x = lin(sd, f"{name}.{i}", x)
    while f"{name}.{i + 2}.weight" in sd:
        i += 2
        x = lin(sd, f"{name}.{i}", F.gelu(x))
    return x


def adaptive_pool(x, shape):
    """apply_adaptive_avg_pooling — R/model/videogpt_plus/model/arch.py:88-96."""
    b, n, c = x.shape
    h = int(n ** 0.5)
    x = F.adaptive_avg_pool2d(x.permute(0, 2, 1).reshape(b, -1, h, h), shape)
    return x.flatten(2).transpose(1, 2)


def project_video(sd, p, video_features, context_features):
    """project(input_type='video') — arch.py:164-191. video_features [nc, 4*L, Dv] (CLS dropped),
    context_features [Te, Lc, Dc] -> [1, Te*144 + nc*4*64, D] (context first, then video)."""
    vf = projector(sd, p + "mm_projector", video_features)
    nc, tl, d = vf.shape
    vf = adaptive_pool(vf.reshape(nc * 4, tl // 4, d), (8, 8)).reshape(nc, -1, d)
    cf = adaptive_pool(projector(sd, p + "image_mm_projector", context_features), (12, 12))
    cf = cf.reshape(1, -1, d)
    return torch.cat([cf[0]] + [vf[i] for i in range(nc)], dim=0).unsqueeze(0)


# ----------------------------------------------------------------------------- L4 splice
def splice(sd, p, input_ids, visual, seg_token_idx):
    """prepare_inputs_labels_for_multimodal for ONE sample with one run of <image> placeholders
    (arch.py:205-216,271-371,453-475,538-552). input_ids [L] (with -200), visual [Nv, D]
    -> inputs_embeds [S, D], seg_token_mask [S] bool."""
    ids = input_ids
    seg = torch.cat([ids[1:] == seg_token_idx, torch.zeros(1, dtype=torch.bool)])
    pos = torch.where(ids == IMAGE_TOKEN_INDEX)[0]
    emb = sd[p + "embed_tokens.weight"]
    if pos.numel() == 0:
        return emb[ids], seg
    s, e = int(pos[0]), int(pos[-1])
    x = torch.cat([emb[ids[:s]], visual, emb[ids[e + 1:]]], dim=0)
    m = torch.cat([seg[:s], torch.zeros(visual.shape[0], dtype=torch.bool), seg[e + 1:]])
    return x, m


# ----------------------------------------------------------------------------- L5 Llama decoder
def rope_tables(hd, n, theta):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = torch.arange(n).float()[:, None] * inv[None]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    return torch.cat((-x[..., x.shape[-1] // 2:], x[..., : x.shape[-1] // 2]), dim=-1)


def llama_forward(sd, p, cfg, x):
    """HF LlamaModel / Phi3Model over inputs_embeds (causal, no cache) + final norm. x: [S,D] -> [S,D] normed hidden.
    HF modeling_llama.py: LlamaDecoderLayer / LlamaAttention (repeat_kv GQA) / LlamaMLP / LlamaRMSNorm; modeling_phi3.py is
    the same arithmetic with fused qkv_proj / gate_up_proj weights (the LLM of the released checkpoint,
    R/model/videogpt_plus/model/language_model/phi3.py:29-40) and a sliding window w = cfg["sliding_window"] (2047 for
    Phi-3-mini-4k): position i attends to [i - w, i], i.e. w + 1 keys.  That is the mask of the reference's pinned
    transformers==4.41.0 (modeling_attn_mask_utils.AttentionMaskConverter._make_causal_mask: context_mask =
    tril(ones, diagonal = -w - 1) filled with min; its flash-attention path passes window_size = (w, w): the same keys).
    SOURCE ABSENT (4.41.0 is not installed here; restated from its published code): the window EDGE is "parity unpinned";
    the arithmetic is pinned by HF 5.15's Phi3Model run with sliding_window = w + 1, whose mask (kv > q - sliding_window)
    shows exactly these keys (tests/golden/phi3_tiny.npz: phi3_win_out)."""
    S, D = x.shape
    H, Hkv, eps = cfg["num_heads"], cfg["num_kv_heads"], cfg["rms_eps"]
    hd = D // H
    cos, sin = rope_tables(hd, S, cfg["rope_theta"])
    mask = torch.full((S, S), float("-inf")).triu(1)
    if cfg.get("sliding_window"):
        mask = mask.masked_fill(torch.ones(S, S, dtype=torch.bool).tril(-int(cfg["sliding_window"]) - 1), float("-inf"))
    for i in range(cfg["num_layers"]):
        l = f"{p}layers.{i}."
        h = rms_norm(x, sd[l + "input_layernorm.weight"], eps)
        if l + "self_attn.qkv_proj.weight" in sd:   # HF Phi3Attention: one fused projection, rows q | k | v
            qkv = lin(sd, l + "self_attn.qkv_proj", h)
            q, k, v = qkv[:, : H * hd], qkv[:, H * hd:(H + Hkv) * hd], qkv[:, (H + Hkv) * hd:]
        else:
            q, k, v = (lin(sd, l + f"self_attn.{n}_proj", h) for n in "qkv")
        q = q.reshape(S, H, hd).transpose(0, 1)
        k = k.reshape(S, Hkv, hd).transpose(0, 1)
        v = v.reshape(S, Hkv, hd).transpose(0, 1)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        k = k.repeat_interleave(H // Hkv, dim=0)
        v = v.repeat_interleave(H // Hkv, dim=0)
        a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5 + mask, dim=-1)
        x = x + lin(sd, l + "self_attn.o_proj", (a @ v).transpose(0, 1).reshape(S, D))
        h = rms_norm(x, sd[l + "post_attention_layernorm.weight"], eps)
        if l + "mlp.gate_up_proj.weight" in sd:     # HF Phi3MLP: gate, up = gate_up_proj(h).chunk(2, -1); down(up * silu(gate))
            gate, up = lin(sd, l + "mlp.gate_up_proj", h).chunk(2, dim=-1)
        else:
            gate, up = lin(sd, l + "mlp.gate_proj", h), lin(sd, l + "mlp.up_proj", h)
        x = x + lin(sd, l + "mlp.down_proj", F.silu(gate) * up)
    return rms_norm(x, sd[p + "norm.weight"], eps)


# ----------------------------------------------------------------------------- generate + L6 [SEG]
def encode_visual(sd, cfg, images, context_images):
    """encode_videos + project — arch.py:121-151,164-191.  images [Te,3,224,224], context [Te,3,336,336]."""
    if context_images is None:
        # image prompt (arch.py:110-119,243-245,393-397): CLIP patch features of the image(s) -> image_mm_projector, NO
        # pooling, tokens of all images concatenated ([t, 576, D] -> [t*576, D])
        cf = clip_forward(sd, "model.image_vision_tower.vision_tower.", cfg["clip"], images)
        pf = projector(sd, "model.image_mm_projector", cf)
        return pf.reshape(-1, pf.shape[-1])
    te = images.shape[0]
    chunks = images.reshape(te // 4, 4, *images.shape[1:])
    vf = iv2_forward(sd, "model.vision_tower.vision_encoder.", cfg["iv2"], chunks)[:, 1:]
    cf = clip_forward(sd, "model.image_vision_tower.vision_tower.", cfg["clip"], context_images)
    return project_video(sd, "model.", vf, cf)[0]


def generate(sd, cfg, images, context_images, input_ids, max_new_tokens, eos_token_id=None, trace=None):
    """VideoGLaMM_SAM2.inference_* steps A–D (R/model/VideoGLaMM.py:609-655 / 781-831): greedy
    generate(use_cache=False) restated as full re-forward of the LM per token (vision encoded once — the
    towers are deterministic, so re-encoding per token as the reference does changes nothing), then
    pred_embeddings = text_hidden_fcs(hidden of the last step)[seg_token_mask].
    input_ids: [L] int64 with -200 placeholders -> (output_ids [L+G], pred_embeddings [N,256])."""
    seg_idx = cfg["seg_token_idx"]
    visual = encode_visual(sd, cfg, images, context_images)
    ids = input_ids.clone()
    hidden = None
    for _ in range(max_new_tokens):
        x, _ = splice(sd, "model.", ids, visual, seg_idx)
        hidden = llama_forward(sd, "model.", cfg["llm"], x)
        nxt = int(torch.argmax(F.linear(hidden[-1], sd["lm_head.weight"])))
        ids = torch.cat([ids, torch.tensor([nxt])])
        if eos_token_id is not None and nxt == eos_token_id:
            break
    added = hidden.shape[0] - (ids.shape[0] - 1)
    seg_mask = torch.cat([torch.zeros(added, dtype=torch.bool), ids[1:] == seg_idx])
    fc = "model.text_hidden_fcs.0."
    emb = lin(sd, fc + "2", F.relu(lin(sd, fc + "0", hidden)))
    if trace is not None:
        trace["visual"] = visual
        trace["hidden"] = hidden
    return ids, emb[seg_mask]
