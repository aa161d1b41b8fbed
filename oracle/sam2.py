"""ORACLE (test infrastructure, never shipped in the product path): CPU fp32 restatement of the
reference's SAM2 promptable pixel decoder, written as plain functions over the reference's own
state-dict names.  R/ = /root/reference/VideoGLaMM/model/segment_anything_2/sam2/.

Pinned against the reference itself: tests/golden/make_golden.py imports the reference in the build
container, runs it on seeded inputs/weights and stores the outputs that tests/test_oracle_*.py
compare these functions with (SURVEY.md §8c — the reference ships no tests/golden vectors).
"""
import math

import torch
import torch.nn.functional as F

NO_OBJ_SCORE = -1024.0  # R/modeling/sam2_base.py:17


def sdpa(q, k, v):
    """F.scaled_dot_product_attention in explicit math form; q,k,v: [B,H,N,D]."""
    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    return torch.softmax(s, dim=-1) @ v


def lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def layer_norm(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def layer_norm2d(sd, name, x, eps=1e-6):
    """R/modeling/sam2_utils.py:137-149 (NCHW, per-pixel over channels)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[name + ".weight"][:, None, None] * x + sd[name + ".bias"][:, None, None]


def mlp(sd, name, x, num_layers, act=F.relu, sigmoid_output=False):
    """R/modeling/sam2_utils.py:108-132."""
    for i in range(num_layers):
        x = lin(sd, f"{name}.layers.{i}", x)
        if i < num_layers - 1:
            x = act(x)
    return torch.sigmoid(x) if sigmoid_output else x


# ----------------------------------------------------------------------------- Hiera + FPN (S1)
def hiera_layout(cfg):
    """Per-block (dim, dim_out, heads, window, q_stride) — R/modeling/backbones/hieradet.py:196-259."""
    stages, window_spec = cfg["stages"], cfg["window_spec"]
    depth = sum(stages)
    stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][: cfg.get("q_pool", 3)]
    embed_dim, num_heads, cur_stage = cfg["embed_dim"], cfg["num_heads"], 1
    blocks = []
    for i in range(depth):
        dim_out = embed_dim
        window = window_spec[cur_stage - 1]
        if i in cfg["global_att_blocks"]:
            window = 0
        if i - 1 in stage_ends:
            dim_out = int(embed_dim * 2.0)
            num_heads = int(num_heads * 2.0)
            cur_stage += 1
        blocks.append(dict(dim=embed_dim, dim_out=dim_out, heads=num_heads, window=window, q_stride=2 if i in q_pool_blocks else 0))
        embed_dim = dim_out
    return blocks, stage_ends


def window_partition(x, ws):
    """R/modeling/backbones/utils.py:16-38."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    """R/modeling/backbones/utils.py:41-62."""
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def _maxpool_nhwc(x):
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def hiera_block(sd, p, blk, x):
    """MultiScaleBlock + MultiScaleAttention — R/modeling/backbones/hieradet.py:37-168."""
    shortcut = x
    x = layer_norm(sd, p + "norm1", x, 1e-6)
    if blk["dim"] != blk["dim_out"]:
        shortcut = lin(sd, p + "proj", x)
        if blk["q_stride"]:
            shortcut = _maxpool_nhwc(shortcut)
    ws = blk["window"]
    H, W = x.shape[1], x.shape[2]
    pad_hw = None
    if ws > 0:
        x, pad_hw = window_partition(x, ws)
    B, h, w, _ = x.shape
    nh = blk["heads"]
    qkv = lin(sd, p + "attn.qkv", x).reshape(B, h * w, 3, nh, -1)
    q, k, v = torch.unbind(qkv, 2)
    if blk["q_stride"]:
        q = _maxpool_nhwc(q.reshape(B, h, w, -1))
        h, w = q.shape[1:3]
        q = q.reshape(B, h * w, nh, -1)
    o = sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(B, h, w, -1)
    x = lin(sd, p + "attn.proj", o)
    if blk["q_stride"]:
        ws = blk["window"] // 2
        H, W = shortcut.shape[1:3]
        pad_hw = (H + (ws - H % ws) % ws, W + (ws - W % ws) % ws) if ws > 0 else None
    if blk["window"] > 0:
        x = window_unpartition(x, ws, pad_hw, (H, W))
    x = shortcut + x
    return x + mlp(sd, p + "mlp", layer_norm(sd, p + "norm2", x, 1e-6), 2, act=F.gelu)


def hiera_pos_embed(sd, p, hw):
    """R/modeling/backbones/hieradet.py:269-277 (bicubic background + tiled window embedding)."""
    h, w = hw
    window_embed = sd[p + "pos_embed_window"]
    pos = F.interpolate(sd[p + "pos_embed"], size=(h, w), mode="bicubic")
    pos = pos + window_embed.tile([x // y for x, y in zip(pos.shape, window_embed.shape)])
    return pos.permute(0, 2, 3, 1)


def hiera_forward(sd, p, cfg, img):
    """R/modeling/backbones/hieradet.py:279-295 ; img [B,3,H,W] -> stage outputs NCHW (high->low res)."""
    blocks, stage_ends = hiera_layout(cfg)
    x = F.conv2d(img, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(sd, p, x.shape[1:3])
    outs = []
    for i, blk in enumerate(blocks):
        x = hiera_block(sd, f"{p}blocks.{i}.", blk, x)
        if i in stage_ends:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


def pos_embed_sine(num_pos_feats, h, w, temperature=10000.0):
    """PositionEmbeddingSine.forward (normalize=True) — R/modeling/position_encoding.py:78-111 -> [C,h,w]."""
    npf = num_pos_feats // 2
    scale = 2 * math.pi
    y = torch.arange(1, h + 1, dtype=torch.float32).view(-1, 1).repeat(1, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, -1).repeat(h, 1)
    eps = 1e-6
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)


def forward_image(sd, p, cfg, img):
    """SAM2Base.forward_image + ImageEncoder + FpnNeck (scalp=1, top-down levels [2,3], nearest) —
    R/modeling/sam2_base.py:465-477 ; backbones/image_encoder.py:29-42,101-133.
    Returns (backbone_fpn [3 levels, conv_s0/s1 applied], vision_pos_enc [3 levels])."""
    xs = hiera_forward(sd, p + "image_encoder.trunk.", cfg["trunk"], img)
    n = len(xs) - 1
    out, pos = [None] * len(xs), [None] * len(xs)
    prev = None
    for i in range(n, -1, -1):
        lat = F.conv2d(xs[i], sd[f"{p}image_encoder.neck.convs.{n - i}.conv.weight"], sd[f"{p}image_encoder.neck.convs.{n - i}.conv.bias"])
        if i in (2, 3) and prev is not None:
            prev = lat + F.interpolate(prev.float(), scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
        pos[i] = pos_embed_sine(256, prev.shape[-2], prev.shape[-1])[None].repeat(prev.shape[0], 1, 1, 1)
    out, pos = out[:-1], pos[:-1]  # scalp = 1
    out[0] = F.conv2d(out[0], sd[p + "sam_mask_decoder.conv_s0.weight"], sd[p + "sam_mask_decoder.conv_s0.bias"])
    out[1] = F.conv2d(out[1], sd[p + "sam_mask_decoder.conv_s1.weight"], sd[p + "sam_mask_decoder.conv_s1.bias"])
    return out, pos


# ----------------------------------------------------------------------------- prompt encoder (S3)
def dense_pe(sd, p, size):
    """PromptEncoder.get_dense_pe — R/modeling/sam/prompt_encoder.py:68-77,216-228 -> [1,C,h,w]."""
    h, w = size
    g = sd[p + "sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    grid = torch.ones((h, w), dtype=g.dtype)
    y = (grid.cumsum(dim=0) - 0.5) / h
    x = (grid.cumsum(dim=1) - 0.5) / w
    c = 2 * torch.stack([x, y], dim=-1) - 1
    c = 2 * math.pi * (c @ g)
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).permute(2, 0, 1)[None]


def prompt_encoder(sd, p, cfg, n, text_embeds, with_empty_point):
    """PromptEncoder.forward(points?, None, None, text_embeds) — R/modeling/sam/prompt_encoder.py:143-189.
    with_empty_point=True reproduces _forward_sam_heads' padding point (label -1) + the pad=True extra
    point of _embed_points (sam2_base.py:310-313; prompt_encoder.py:79-101) -> 2 not-a-point tokens."""
    es = cfg["image_size"] // 16
    sparse = torch.empty((n, 0, 256))
    if with_empty_point:
        nap = sd[p + "sam_prompt_encoder.not_a_point_embed.weight"]  # [1,256]
        sparse = torch.cat([sparse, nap[None].expand(n, 2, 256)], dim=1)  # PE zeroed for label -1
    if text_embeds is not None:
        sparse = torch.cat([sparse, text_embeds], dim=1)
    dense = sd[p + "sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(n, -1, es, es)
    return sparse, dense


# ----------------------------------------------------------------------------- mask decoder (S7, S8)
def attn(sd, p, q, k, v, heads):
    """sam/transformer.py Attention.forward :236-260."""
    q, k, v = lin(sd, p + "q_proj", q), lin(sd, p + "k_proj", k), lin(sd, p + "v_proj", v)

    def sep(x):
        b, n, c = x.shape
        return x.reshape(b, n, heads, c // heads).transpose(1, 2)

    o = sdpa(sep(q), sep(k), sep(v)).transpose(1, 2)
    return lin(sd, p + "out_proj", o.reshape(o.shape[0], o.shape[1], -1))


def two_way_transformer(sd, p, src, pos_src, tokens):
    """TwoWayTransformer / TwoWayAttentionBlock — R/modeling/sam/transformer.py:69-115,160-193."""
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos_src.flatten(2).permute(0, 2, 1)
    queries, query_pe = tokens, tokens
    for i in range(2):
        lp = f"{p}layers.{i}."
        if i == 0:
            queries = attn(sd, lp + "self_attn.", queries, queries, queries, 8)
        else:
            q = queries + query_pe
            queries = queries + attn(sd, lp + "self_attn.", q, q, queries, 8)
        queries = layer_norm(sd, lp + "norm1", queries)
        q, k = queries + query_pe, keys + key_pe
        queries = layer_norm(sd, lp + "norm2", queries + attn(sd, lp + "cross_attn_token_to_image.", q, k, keys, 8))
        queries = layer_norm(sd, lp + "norm3", queries + mlp(sd, lp + "mlp", queries, 2))
        q, k = queries + query_pe, keys + key_pe
        keys = layer_norm(sd, lp + "norm4", keys + attn(sd, lp + "cross_attn_image_to_token.", k, q, queries, 8))
    q, k = queries + query_pe, keys + key_pe
    queries = layer_norm(sd, p + "norm_final_attn", queries + attn(sd, p + "final_attn_token_to_image.", q, k, keys, 8))
    return queries, keys


def mask_decoder_predict(sd, p, image_embeddings, image_pe, sparse, dense, repeat_image, high_res):
    """MaskDecoder.predict_masks — R/modeling/sam/mask_decoder.py:168-245.
    -> masks [N,4,4h,4w], iou_pred [N,4], mask_tokens_out [N,4,256], object_score_logits [N,1]."""
    d = p + "sam_mask_decoder."
    out_tokens = torch.cat([sd[d + "obj_score_token.weight"], sd[d + "iou_token.weight"], sd[d + "mask_tokens.weight"]], dim=0)
    tokens = torch.cat((out_tokens[None].expand(sparse.size(0), -1, -1), sparse), dim=1)
    src = torch.repeat_interleave(image_embeddings, tokens.shape[0], dim=0) if repeat_image else image_embeddings
    src = src + dense
    pos_src = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(sd, d + "transformer.", src, pos_src, tokens)
    iou_token_out = hs[:, 1, :]
    mask_tokens_out = hs[:, 2:6, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    feat_s0, feat_s1 = high_res
    up = F.conv_transpose2d(src, sd[d + "output_upscaling.0.weight"], sd[d + "output_upscaling.0.bias"], stride=2)
    up = F.gelu(layer_norm2d(sd, d + "output_upscaling.1", up + feat_s1))
    up = F.gelu(F.conv_transpose2d(up, sd[d + "output_upscaling.3.weight"], sd[d + "output_upscaling.3.bias"], stride=2) + feat_s0)
    hyper_in = torch.stack([mlp(sd, f"{d}output_hypernetworks_mlps.{i}", mask_tokens_out[:, i, :], 3) for i in range(4)], dim=1)
    b, c, h, w = up.shape
    masks = (hyper_in @ up.view(b, c, h * w)).view(b, -1, h, w)
    iou_pred = mlp(sd, d + "iou_prediction_head", iou_token_out, 3, sigmoid_output=True)
    obj = mlp(sd, d + "pred_obj_score_head", hs[:, 0, :], 3)
    return masks, iou_pred, mask_tokens_out, obj


def dynamic_multimask_via_stability(masks, iou, delta=0.05, thresh=0.98):
    """R/modeling/sam/mask_decoder.py:247-295."""
    mm, mi = masks[:, 1:], iou[:, 1:]
    best = torch.argmax(mi, dim=-1)
    bi = torch.arange(mi.size(0))
    best_m, best_i = mm[bi, best].unsqueeze(1), mi[bi, best].unsqueeze(1)
    sm, si = masks[:, 0:1], iou[:, 0:1]
    fl = sm.flatten(-2)
    area_i = torch.sum(fl > delta, dim=-1).float()
    area_u = torch.sum(fl > -delta, dim=-1).float()
    stab = torch.where(area_u > 0, area_i / area_u, 1.0)
    ok = stab >= thresh
    return torch.where(ok[..., None, None].expand_as(sm), sm, best_m), torch.where(ok.expand_as(si), si, best_i)


def mask_decoder(sd, p, image_embeddings, image_pe, sparse, dense, multimask_output, repeat_image, high_res):
    """MaskDecoder.forward — R/modeling/sam/mask_decoder.py:110-166."""
    masks, iou, tok, obj = mask_decoder_predict(sd, p, image_embeddings, image_pe, sparse, dense, repeat_image, high_res)
    if multimask_output:
        return masks[:, 1:], iou[:, 1:], tok[:, 1:], obj  # use_multimask_token_for_obj_ptr: true
    m, i = dynamic_multimask_via_stability(masks, iou)
    return m, i, tok[:, 0:1], obj


# ----------------------------------------------------------------------------- memory attention (S5)
def axial_cis(dim, end_x, end_y, theta=10000.0):
    """compute_axial_cis — R/modeling/position_encoding.py:174-191 (complex64 [end_x*end_y, dim/2])."""
    fx = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    t = torch.arange(end_x * end_y, dtype=torch.float32)
    tx, ty = (t % end_x).float(), torch.div(t, end_x, rounding_mode="floor").float()
    fxo, fyo = torch.outer(tx, fx), torch.outer(ty, fx)
    return torch.cat([torch.polar(torch.ones_like(fxo), fxo), torch.polar(torch.ones_like(fyo), fyo)], dim=-1)


def apply_rotary(xq, xk, cis, repeat_k):
    """apply_rotary_enc — R/modeling/position_encoding.py:194-216; x: [B,H,N,D]."""
    xq_ = torch.view_as_complex(xq.float().reshape(*xq.shape[:-1], -1, 2))
    c = cis.view(1, 1, *cis.shape)
    q_out = torch.view_as_real(xq_ * c).flatten(3)
    if xk.shape[-2] == 0:
        return q_out, xk
    xk_ = torch.view_as_complex(xk.float().reshape(*xk.shape[:-1], -1, 2))
    if repeat_k:
        c = c.repeat(1, 1, xk_.shape[-2] // xq_.shape[-2], 1)
    return q_out, torch.view_as_real(xk_ * c).flatten(3)


def rope_attn(sd, p, q, k, v, num_k_exclude_rope, repeat_k):
    """RoPEAttention.forward (1 head, d=256) — R/modeling/sam/transformer.py:289-327."""
    q, k, v = lin(sd, p + "q_proj", q)[:, None], lin(sd, p + "k_proj", k)[:, None], lin(sd, p + "v_proj", v)[:, None]
    side = int(math.sqrt(q.shape[-2]))
    cis = axial_cis(q.shape[-1], side, side)
    nk = k.size(-2) - num_k_exclude_rope
    q, k_rot = apply_rotary(q, k[:, :, :nk], cis, repeat_k)
    k = torch.cat([k_rot, k[:, :, nk:]], dim=2)
    return lin(sd, p + "out_proj", sdpa(q, k, v)[:, 0])


def memory_attention(sd, p, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens):
    """MemoryAttention.forward, 4 layers, batch-first inside — R/modeling/memory_attention.py:119-169,60-99.
    curr/curr_pos: [HW,B,256]; memory/memory_pos: [M,B,64] -> [HW,B,256]."""
    m = p + "memory_attention."
    out = (curr + 0.1 * curr_pos).transpose(0, 1)
    memory, memory_pos = memory.transpose(0, 1), memory_pos.transpose(0, 1)
    for i in range(4):
        lp = f"{m}layers.{i}."
        t2 = layer_norm(sd, lp + "norm1", out)
        out = out + rope_attn(sd, lp + "self_attn.", t2, t2, t2, 0, False)
        t2 = layer_norm(sd, lp + "norm2", out)
        out = out + rope_attn(sd, lp + "cross_attn_image.", t2, memory + memory_pos, memory, num_obj_ptr_tokens, True)
        t2 = layer_norm(sd, lp + "norm3", out)
        out = out + lin(sd, lp + "linear2", F.relu(lin(sd, lp + "linear1", t2)))
    return layer_norm(sd, m + "norm", out).transpose(0, 1)


# ----------------------------------------------------------------------------- memory encoder (S9)
def memory_encoder(sd, p, pix_feat, masks):
    """MemoryEncoder.forward(skip_mask_sigmoid=True) — R/modeling/memory_encoder.py:159-182,17-118.
    pix_feat [B,256,h,w], masks [B,1,16h,16w] -> (features [B,64,h,w], pos [B,64,h,w])."""
    e = p + "memory_encoder."
    x = masks
    for i in range(4):
        x = F.conv2d(x, sd[f"{e}mask_downsampler.encoder.{3 * i}.weight"], sd[f"{e}mask_downsampler.encoder.{3 * i}.bias"], stride=2, padding=1)
        x = F.gelu(layer_norm2d(sd, f"{e}mask_downsampler.encoder.{3 * i + 1}", x))
    x = F.conv2d(x, sd[e + "mask_downsampler.encoder.12.weight"], sd[e + "mask_downsampler.encoder.12.bias"])
    x = F.conv2d(pix_feat, sd[e + "pix_feat_proj.weight"], sd[e + "pix_feat_proj.bias"]) + x
    for i in range(2):
        l = f"{e}fuser.layers.{i}."
        inp = x
        x = F.conv2d(x, sd[l + "dwconv.weight"], sd[l + "dwconv.bias"], padding=3, groups=x.shape[1])
        x = layer_norm2d(sd, l + "norm", x).permute(0, 2, 3, 1)
        x = lin(sd, l + "pwconv2", F.gelu(lin(sd, l + "pwconv1", x)))
        x = (sd[l + "weight"] * x).permute(0, 3, 1, 2)
        x = inp + x
    x = F.conv2d(x, sd[e + "out_proj.weight"], sd[e + "out_proj.bias"])
    pos = pos_embed_sine(64, x.shape[-2], x.shape[-1])[None].repeat(x.shape[0], 1, 1, 1)
    return x, pos


# ----------------------------------------------------------------------------- SAM heads + tracking (S4, S6, S10)
def forward_sam_heads(sd, p, cfg, backbone_features, high_res, text_inputs, multimask_output):
    """SAM2Base._forward_sam_heads with points=None, mask=None — R/modeling/sam2_base.py:251-411."""
    B = backbone_features.size(0)
    sparse, dense = prompt_encoder(sd, p, cfg, B, text_inputs, with_empty_point=True)
    es = cfg["image_size"] // 16
    low, ious, tokens, obj = mask_decoder(sd, p, backbone_features, dense_pe(sd, p, (es, es)), sparse, dense,
                                          multimask_output, False, high_res)
    is_obj = obj > 0
    low_pre_where = low
    low = torch.where(is_obj[:, None, None], low, torch.tensor(NO_OBJ_SCORE)).float()
    high = F.interpolate(low, size=(cfg["image_size"],) * 2, mode="bilinear", align_corners=False)
    tok = tokens[:, 0]
    if multimask_output:
        best = torch.argmax(ious, dim=-1)
        bi = torch.arange(B)
        low_best, high_best = low[bi, best].unsqueeze(1), high[bi, best].unsqueeze(1)
        tok = tokens[bi, best]
    else:
        low_best, high_best = low, high
    obj_ptr = mlp(sd, p + "obj_ptr_proj", tok, 3)
    lam = is_obj.float()
    obj_ptr = lam * obj_ptr + (1 - lam) * sd[p + "no_obj_ptr"]
    return dict(low_multi=low, low_multi_pre_where=low_pre_where, ious=ious, low=low_best, high=high_best,
                obj_ptr=obj_ptr, obj_logits=obj)


def encode_new_memory(sd, p, cfg, feat_top, high_res_masks, is_mask_from_pts):
    """SAM2Base._encode_new_memory — R/modeling/sam2_base.py:666-704 ; feat_top: [HW,B,256]."""
    B = feat_top.size(1)
    es = cfg["image_size"] // 16
    pix = feat_top.permute(1, 2, 0).view(B, 256, es, es)
    m = (high_res_masks > 0).float() if is_mask_from_pts else torch.sigmoid(high_res_masks)  # binarize_mask_from_pts_for_mem_enc
    m = m * 20.0 - 10.0
    return memory_encoder(sd, p, pix, m)


def video_branch(sd, p, cfg, images, text_embeds, video_hw):
    """VideoGLaMM video branch over one clip: init_state_from_tensor -> add_new_text per object ->
    propagate_in_video — R/model/VideoGLaMM.py:834-877 ; R/sam2_video_predictor.py:108-180,415-495,
    520-636,674-827,921-1017 ; R/modeling/sam2_base.py:495-664,706-803.
    images [T,3,S,S]; text_embeds [N,256] -> list over frames of logits [N,1,H,W] (fp32) + trace dict."""
    T, N = images.shape[0], text_embeds.shape[0]
    S = cfg["image_size"]
    es = S // 16
    feat_cache = {}

    def feats(t, bs):
        if t not in feat_cache:
            feat_cache.clear()
            feat_cache[t] = forward_image(sd, p, cfg, images[t:t + 1].float())
        fpn, pos = feat_cache[t]
        vf = [x.expand(bs, -1, -1, -1).flatten(2).permute(2, 0, 1) for x in fpn]
        vp = [x.expand(bs, -1, -1, -1).flatten(2).permute(2, 0, 1) for x in pos]
        sizes = [(x.shape[-2], x.shape[-1]) for x in pos]
        return vf, vp, sizes

    def high_res(vf, sizes):
        return [x.permute(1, 2, 0).view(x.size(1), x.size(2), *s) for x, s in zip(vf[:-1], sizes[:-1])]

    trace = {}
    # --- frame 0: one object at a time, batch 1, no memory encoder (add_new_text)
    per_obj = []
    for k in range(N):
        vf, vp, sizes = feats(0, 1)
        pix = (vf[-1] + sd[p + "no_mem_embed"]).permute(1, 2, 0).view(1, 256, es, es)  # directly_add_no_mem_embed
        per_obj.append(forward_sam_heads(sd, p, cfg, pix, high_res(vf, sizes), text_embeds[k:k + 1].unsqueeze(1), True))
    trace["frame0_low_multi_pre_where"] = torch.cat([o["low_multi_pre_where"] for o in per_obj])
    trace["frame0_ious"] = torch.cat([o["ious"] for o in per_obj])
    trace["frame0_obj_logits"] = torch.cat([o["obj_logits"] for o in per_obj])
    # --- preflight consolidation (low-res 256-grid masks, obj ptrs) + memory encoder on frame 0
    low0 = torch.cat([o["low"] for o in per_obj])            # [N,1,S/4,S/4]
    ptr0 = torch.cat([o["obj_ptr"] for o in per_obj])        # [N,256]
    vf, vp, sizes = feats(0, N)
    high0 = F.interpolate(low0, size=(S, S), mode="bilinear", align_corners=False)
    mem0, mempos = encode_new_memory(sd, p, cfg, vf[-1], high0, True)
    mem0 = mem0.to(torch.bfloat16)                           # sam2_video_predictor.py:1011
    maskmem_pos = mempos[0:1]                                # cached constant (:1019-1042)
    cond = dict(maskmem_features=mem0, obj_ptr=ptr0)
    non_cond = {}
    outs = [low0]
    trace["obj_ptr"], trace["maskmem"] = [ptr0], [mem0.float()]
    tpos = sd[p + "maskmem_tpos_enc"]
    for t in range(1, T):
        vf, vp, sizes = feats(t, N)
        # _prepare_memory_conditioned_features (num_maskmem 7, stride 1, cond frame 0 only)
        prevs = [(0, cond)]
        for t_pos in range(1, 7):
            t_rel = 7 - t_pos
            prevs.append((t_pos, non_cond.get(t - t_rel)))   # both branches reduce to frame_idx - t_rel at r=1
        mems, mposs = [], []
        for t_pos, prev in prevs:
            if prev is None:
                continue
            mems.append(prev["maskmem_features"].flatten(2).permute(2, 0, 1))
            enc = maskmem_pos.expand(N, -1, -1, -1).flatten(2).permute(2, 0, 1)
            mposs.append(enc + tpos[7 - t_pos - 1])
        ptrs = [cond["obj_ptr"]]
        for t_diff in range(1, min(T, 16)):
            tt = t - t_diff
            if tt < 0:
                break
            if tt in non_cond:
                ptrs.append(non_cond[tt]["obj_ptr"])
        obj_ptrs = torch.stack(ptrs, dim=0)                   # [P,N,256]
        obj_pos = obj_ptrs.new_zeros(len(ptrs), N, 64)        # add_tpos_enc_to_obj_ptrs: false
        obj_ptrs = obj_ptrs.reshape(-1, N, 4, 64).permute(0, 2, 1, 3).flatten(0, 1)
        obj_pos = obj_pos.repeat_interleave(4, dim=0)
        # NB: torch.cat promotes the bf16 memories to fp32 (values stay bf16-rounded)
        memory = torch.cat(mems + [obj_ptrs], dim=0)
        memory_pos = torch.cat(mposs + [obj_pos], dim=0)
        pix = memory_attention(sd, p, vf[-1], vp[-1], memory, memory_pos, obj_ptrs.shape[0])
        pix = pix.permute(1, 2, 0).view(N, 256, es, es)
        o = forward_sam_heads(sd, p, cfg, pix, high_res(vf, sizes), None, True)
        mem, _ = encode_new_memory(sd, p, cfg, vf[-1], o["high"], False)
        non_cond[t] = dict(maskmem_features=mem.to(torch.bfloat16), obj_ptr=o["obj_ptr"])
        outs.append(o["low"])
        trace["obj_ptr"].append(o["obj_ptr"])
        trace["maskmem"].append(non_cond[t]["maskmem_features"].float())
        if t == 1:
            trace["frame1_pix_feat_with_mem"] = pix
            trace["frame1_low_multi_pre_where"] = o["low_multi_pre_where"]
            trace["frame1_maskmem_features"] = mem
        trace[f"obj_logits_{t}"] = o["obj_logits"]
    H, W = video_hw
    video_res = [F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False) for x in outs]
    trace["low_res"] = torch.stack(outs)
    trace["obj_ptr"] = torch.stack(trace["obj_ptr"])          # [T,N,256]
    return video_res, trace


def framewise_branch(sd, p, cfg, images, text_embeds, video_hw):
    """VideoGLaMM framewise decode — R/model/VideoGLaMM.py:205-241,676-766: per frame, Hiera + no_mem_embed,
    1 sparse token (text only), multimask_output=False (dynamic stability fallback), repeat_image=True."""
    T, N = images.shape[0], text_embeds.shape[0]
    es = cfg["image_size"] // 16
    sparse, dense = prompt_encoder(sd, p, cfg, N, text_embeds.unsqueeze(1), with_empty_point=False)
    pe = dense_pe(sd, p, (es, es))
    outs, lows = [], []
    for t in range(T):
        fpn, _ = forward_image(sd, p, cfg, images[t:t + 1])
        emb = fpn[-1] + sd[p + "no_mem_embed"].view(1, 256, 1, 1)
        low, _, _, _ = mask_decoder(sd, p, emb, pe, sparse, dense, False, True, fpn[:-1])
        lows.append(low)
        outs.append(F.interpolate(low.float(), video_hw, mode="bilinear", align_corners=False)[:, 0])
    return outs, torch.stack(lows)
