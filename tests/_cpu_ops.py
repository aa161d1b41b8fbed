"""Plain-PyTorch statement of every operator in videoglamm_amd/ops.py (test infrastructure only).

Used (a) on the GPU box as the fp32 reference each HIP kernel is compared with, and (b) on CPU to
exercise the host-side graph code (videoglamm_amd/*.py) without a GPU by monkeypatching
``videoglamm_amd.ops``.  Never imported by the product.
"""
import torch
import torch.nn.functional as F

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3, 4, 5


def _act(x, act):
    if act == ACT_GELU:
        return F.gelu(x)
    if act == ACT_QUICK_GELU:
        return x * torch.sigmoid(1.702 * x)
    if act == ACT_RELU:
        return F.relu(x)
    if act == ACT_SILU:
        return F.silu(x)
    if act == ACT_SIGMOID:
        return torch.sigmoid(x)
    return x


def linear(x, w, bias=None, act=ACT_NONE, gamma=None, residual=None, out_dtype=None, out=None, glu=False):
    if glu:
        return swiglu(linear(x, w, bias))
    odt = out_dtype if out_dtype is not None else x.dtype
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    y = _act(y, act)
    if gamma is not None:
        y = y * gamma
    if residual is not None:
        y = y + residual.float()
    y = y.to(odt)
    if out is not None:
        out.copy_(y)
        return out
    return y


def linear_rows(x, w, bias=None, act=ACT_NONE, residual=None, out=None, ln=None, add=None, rope=None, force_fused=False):
    """vg_gemm_rows as the separate statements it fuses (intermediates rounded to the storage dtype like the separate launches)."""
    K, N = x.shape[-1], w.shape[0]
    M = x.numel() // K
    h = x
    if ln is not None:
        h = layernorm(x, ln[0], ln[1], ln[2])
    elif add is not None:
        h = (x.reshape(-1, add.shape[0], K).float() + add.float()).to(x.dtype).reshape(x.shape)
    y = linear(h, w, bias, act, None, residual)
    if rope is not None:
        cos, sin, cols, ch, rpb, r0, r1, grid = rope
        heads, B = cols // ch, M // rpb
        yv = y.reshape(B, rpb, N).clone()
        part = yv[:, r0:r1, :cols].reshape(B, r1 - r0, heads, ch).permute(0, 2, 1, 3).reshape(B * heads, r1 - r0, ch).clone(memory_format=torch.contiguous_format)
        rope_axial_(part, cos, sin, r1 - r0, grid)
        yv[:, r0:r1, :cols] = part.view(B, heads, r1 - r0, ch).permute(0, 2, 1, 3).reshape(B, r1 - r0, cols)
        y = yv.reshape(*x.shape[:-1], N)
    if out is not None:
        out.copy_(y.reshape(out.shape))
        return out
    return y


def heads_blockdiag(x, TP):
    N, nt = x.shape[0], x.shape[1]
    bd = torch.zeros(N, 8, TP, 8, 16, dtype=x.dtype, device=x.device)
    torch.diagonal(bd, dim1=1, dim2=3)[:, :nt].copy_(x.view(N, nt, 8, 16).permute(0, 1, 3, 2))
    return bd.view(N * 8 * TP, 128)


def heads_blockdiag_gather(full, N, nt, TP):
    o_full = full.view(N, 8, TP, 8, 16)
    return torch.diagonal(o_full, dim1=1, dim2=3)[:, :nt].permute(0, 1, 3, 2).reshape(N, nt, 128)


def linear_ln(x, ln, w, bias=None, act=ACT_NONE, window=None):
    xn = layernorm(x, ln[0], ln[1], ln[2])
    return linear_window(xn, w, bias, *window, scatter=False, act=act) if window is not None else linear(xn, w, bias, act=act)


def mlp_rows(x, ln, w1, b1, w2, b2, force=False):
    h = linear(layernorm(x, ln[0], ln[1], ln[2]), w1, b1, ACT_GELU)
    return linear(h, w2, b2, ACT_NONE, None, x)


def linear_window(x, w, bias, B, H, W, ws, scatter, act=ACT_NONE, gamma=None, residual=None):
    if scatter:
        y = linear(x, w, bias, act, gamma)
        y = window_unpartition(y, ws, B, H, W)
        return y if residual is None else (y.float() + residual.float()).to(y.dtype)
    return linear(window_partition(x.view(B, H, W, -1), ws), w, bias, act, gamma)


def bmm_nt(a, w, out_dtype=None, shared_a=False):
    odt = out_dtype if out_dtype is not None else a.dtype
    if shared_a:
        a = a[:, :w.shape[-1]]
    return (a.float() @ w.float().transpose(-1, -2)).to(odt)


def attention(q, k, v, scale, causal=False, window=0):
    B, Sq, Hq, D = q.shape
    Skv, Hkv = k.shape[1], k.shape[2]
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(Hq // Hkv, dim=1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(Hq // Hkv, dim=1)
    s = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        i = torch.arange(Sq, device=q.device)[:, None]
        j = torch.arange(Skv, device=q.device)[None, :]
        s = s.masked_fill(j > i + (Skv - Sq), float("-inf"))
        if window:      # the query's own position and the window - 1 before it
            s = s.masked_fill(j <= i + (Skv - Sq) - window, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ vf).permute(0, 2, 1, 3).contiguous().to(q.dtype)


def attention_dv(q, k, v, scale):
    return attention(q, k, v, scale)


def window_attention(q, k, v, scale):
    return None          # (the dedicated window kernels are a device-only route)


def attention_windows(q, k, v, scale):
    return attention(q, k, v, scale)


def mask_upscale(x, w0, b0, s1, ln_w, ln_b, eps, w1, b1, s0, hyper, es):
    """mask_decoder.py:225-245 on channels-last tensors (sam2.py: mask_decoder's unfused chain, in fp32): ConvT as GEMM + pixel shuffle, + s1,
    LayerNorm2d, GELU, ConvT, + s0, GELU, hyper . upscaled."""
    N, Bi = x.shape[0], s1.shape[0]
    rep = lambda t: t.float().repeat(N // Bi, 1, 1)      # noqa: E731   instance n -> image n % Bi
    g = x.float() @ w0.float().t()
    up = pixel_shuffle2(g, b0, N, es, es, 64).view(N, 4 * es * es, 64) + rep(s1)
    up = F.gelu(F.layer_norm(up, (64,), ln_w, ln_b, eps))
    g = up @ w1.float().t()
    up = F.gelu(pixel_shuffle2(g, b1, N, 2 * es, 2 * es, 32).view(N, 16 * es * es, 32) + rep(s0))
    return (hyper.float() @ up.transpose(1, 2)).view(N, 4, 4 * es, 4 * es)


def twoway_image_update(xpe, x, u2, c2, w2t, bo, ln_w, ln_b, eps, pe, nt, TP):
    """image -> token cross-attention of a two-way block on the image rows, in the fused form (sam2.py: _i2t_fused):
    scores = xpe . u2^T + c2 over columns (head h, token t) = h * TP + t; softmax over t < nt inside each head; y = a . w2t^T + bo;
    x' = LayerNorm(x + y); returns (x', x' + pe)."""
    N, P = u2.shape[0], x.shape[1]
    xpe, x = xpe.repeat(N // x.shape[0], 1, 1), x.repeat(N // x.shape[0], 1, 1)      # instance n reads input slot n % Nx
    s = (xpe.float() @ u2.float().transpose(1, 2) + c2.float()[:, None, :]).view(N, P, 8, TP)
    s[..., nt:] = float("-inf")
    a = torch.softmax(s, dim=-1).view(N, P, 8 * TP)
    y = (a @ w2t.float().transpose(1, 2)).to(x.dtype).float() + bo
    xn = F.layer_norm(x.float() + y, (x.shape[-1],), ln_w, ln_b, eps).to(x.dtype)
    return xn, (xn.float() + pe.float()).to(x.dtype)


def layernorm(x, w, b, eps, out_dtype=None):
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps).to(out_dtype or x.dtype)


def rmsnorm(x, w, eps, out_dtype=None):
    xf = x.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype).float()
    if w is not None:
        y = y * w
    return y.to(out_dtype or x.dtype)


def axpby(a, b, alpha=1.0, beta=1.0, out_dtype=None):
    odt = out_dtype or a.dtype
    if b is None:
        return (alpha * a.float() + beta).to(odt)
    reps = a.numel() // b.numel()
    bb = b.float().reshape(-1).repeat(reps).reshape(a.shape)
    return (alpha * a.float() + beta * bb).to(odt)


def add(a, b):
    return axpby(a, b, 1.0, 1.0)


def activation(x, act, out_dtype=None):
    return _act(x.float(), act).to(out_dtype or x.dtype)


def swiglu(gu):
    Fh = gu.shape[-1] // 2
    g = F.silu(gu[..., :Fh].float()).to(gu.dtype).float()
    return (g * gu[..., Fh:].float()).to(gu.dtype)


def cast(x, dtype):
    return x.to(dtype)


def where_rows(cond, a, b=None, fill=0.0, out=None):
    rows = cond.numel()
    a2 = a.reshape(rows, -1)
    if b is not None:
        other = b.reshape(-1).repeat(a2.shape[1] // b.numel()).to(a.dtype)[None, :].expand_as(a2)
    else:
        other = torch.full_like(a2, fill)
    y = torch.where(cond.reshape(rows, 1) > 0, a2, other).reshape(a.shape)
    if out is not None:
        out.copy_(y.reshape(out.shape))
        return out
    return y


def mask_for_mem(x, binarize, scale, bias, out_dtype):
    m = (x > 0).float() if binarize else torch.sigmoid(x)
    return (m * scale + bias).to(out_dtype)


def bilinear_mask(x, Ho, Wo):
    return threshold(bilinear(x, Ho, Wo))


def threshold(x):
    return (x > 0).to(torch.uint8)


def rope_half_(x, cos, sin, pos0):
    S, H, D = x.shape
    c = cos[pos0:pos0 + S].to(x.dtype)[:, None, :]
    s = sin[pos0:pos0 + S].to(x.dtype)[:, None, :]
    x1, x2 = x[..., : D // 2].clone(), x[..., D // 2:].clone()
    x[..., : D // 2] = x1 * c + (-x2) * s
    x[..., D // 2:] = x2 * c + x1 * s
    return x


def rope_axial_(x, cos, sin, n_rope, n_grid):
    B, N, C = x.shape
    if n_rope == 0:
        return x
    xr = x[:, :n_rope].float().reshape(B, n_rope, C // 2, 2)
    reps = n_rope // n_grid
    c = cos.repeat(reps, 1)[None]
    s = sin.repeat(reps, 1)[None]
    a, b = xr[..., 0], xr[..., 1]
    out = torch.stack([a * c - b * s, a * s + b * c], dim=-1).reshape(B, n_rope, C)
    x[:, :n_rope] = out.to(x.dtype)
    return x


def rope_axial_heads_(x, heads, cos, sin, n_rope, n_grid):
    Ch = 2 * cos.shape[1]
    for h in range(heads):
        part = x[..., h * Ch:(h + 1) * Ch].contiguous()
        x[..., h * Ch:(h + 1) * Ch] = rope_axial_(part, cos, sin, n_rope, n_grid)
    return x


def embed(ids, table):
    return table[ids.reshape(-1)]


def argmax(x, out=None):
    r = torch.argmax(x.float(), dim=-1)
    if out is not None:
        out.copy_(r.reshape(out.shape))
        return out
    return r


def attention_decode(q, k_cache, v_cache, pos_dev, scale, window=0):
    n = int(pos_dev[0]) + q.shape[1]
    return attention(q, k_cache[:n].unsqueeze(0), v_cache[:n].unsqueeze(0), scale, causal=True, window=window)


def rope_kv_append_(qkv, k_cache, v_cache, cos, sin, H, Hkv, D, pos0=0, pos_dev=None):
    pos = int(pos_dev[0]) if pos_dev is not None else pos0
    S = qkv.shape[0]
    q = qkv[:, : H * D].view(S, H, D)
    k = qkv[:, H * D:(H + Hkv) * D].reshape(S, Hkv, D).clone()
    rope_half_(q, cos, sin, pos)
    rope_half_(k, cos, sin, pos)
    k_cache[pos:pos + S] = k
    v_cache[pos:pos + S] = qkv[:, (H + Hkv) * D:].reshape(S, Hkv, D)
    return qkv


def decode_gemv(x, w, norm_w=None, eps=0.0, residual=None, glu=False, out_dtype=None, out=None):
    h = rmsnorm(x, norm_w, eps) if norm_w is not None else x
    return linear(h.view(1, -1), w, residual=residual, out_dtype=out_dtype, out=out, glu=glu)


def decode_attention_workspace(H, Hkv, D, max_len, device):
    return torch.zeros(1)


def decode_attention(qkv, k_cache, v_cache, cos, sin, H, Hkv, D, pos_dev, scale, ws, window=0, keys_per_wg=0):   # (keys_per_wg: a work split of the HIP kernel, no arithmetic meaning)
    qkv = qkv.clone()
    rope_kv_append_(qkv, k_cache, v_cache, cos, sin, H, Hkv, D, 0, pos_dev)
    q = qkv[:, : H * D].view(1, 1, H, D)
    return attention_decode(q, k_cache, v_cache, pos_dev, scale, window).view(1, H * D)


def store_row_(src, dst, idx_dev, idx_off=0):
    dst.view(-1, src.numel())[int(idx_dev[0]) + idx_off] = src.reshape(-1)
    return dst


def add_int_(p, v):
    p += v
    return p


def decode_advance_(pos_dev, inc, tok_dev=None, step_dev=None, forced=None, hist=None, raw=None, rope=None):
    """vg_decode_advance as stated in include/vg_kernels.h"""
    if tok_dev is not None:
        k = int(step_dev[0])
        t0 = int(tok_dev[0])
        t = t0
        if forced is not None and k < forced.numel() and int(forced[k]) >= 0:
            t = int(forced[k])
        if raw is not None and k < raw.numel():
            raw[k] = t0
        if hist is not None and k < hist.numel():
            hist[k] = t
        tok_dev[0] = t
        step_dev[0] = k + 1
    pos_dev += inc
    if rope is not None:
        cos, sin, rope_cs = rope
        hd = cos.shape[1]
        rope_cs[:hd] = cos[int(pos_dev[0])]
        rope_cs[hd:] = sin[int(pos_dev[0])]


def multimask_select(masks, ious, tokens, mode, delta=0.05, thresh=0.98):
    N = masks.shape[0]
    bi = torch.arange(N)
    best = torch.argmax(ious[:, 1:], dim=-1) + 1
    if mode == 0:
        fl = masks[:, 0].flatten(1)
        ai, au = (fl > delta).sum(-1).float(), (fl > -delta).sum(-1).float()
        stab = torch.where(au > 0, ai / au, torch.ones_like(au))
        sel = torch.where(stab >= thresh, torch.zeros_like(best), best)
        tok = tokens[:, 0]
    else:
        sel = best
        tok = tokens[bi, sel]
    return masks[bi, sel].unsqueeze(1).contiguous(), ious[bi, sel], tok.contiguous(), sel.to(torch.int32)


def permute5(x, dims, strides):
    return torch.as_strided(x, dims, strides).contiguous()


def im2col(x, kh, kw, stride, pad, kpad):
    B, H, W, C = x.shape
    cols = F.unfold(x.float().permute(0, 3, 1, 2), (kh, kw), padding=pad, stride=stride)  # [B, C*kh*kw, L]
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    cols = cols.reshape(B, C, kh * kw, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, kh * kw * C)
    out = torch.zeros(B * Ho * Wo, kpad, dtype=x.dtype, device=x.device)
    out[:, : kh * kw * C] = cols.to(x.dtype)
    return out, Ho, Wo


def dwconv(x, w, bias, k):
    B, H, W, C = x.shape
    wt = w.reshape(k, k, C).permute(2, 0, 1)[:, None]  # [C,1,k,k]
    y = F.conv2d(x.float().permute(0, 3, 1, 2), wt, bias, padding=k // 2, groups=C)
    return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)


def conv3s2_ln_gelu(x, w, bias, ln_w, ln_b, eps):
    """memory_encoder.py:17-63 on channels-last tensors: Conv2d(3, stride 2, pad 1) -> LayerNorm2d -> GELU."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    if (Cin, Cout) not in ((1, 4), (4, 16)):
        return None
    wt = w.float()[:, : 9 * Cin].reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), wt, bias, stride=2, padding=1).permute(0, 2, 3, 1)
    y = F.gelu(F.layer_norm(y, (Cout,), ln_w, ln_b, eps))
    return y.contiguous().to(x.dtype)


def pixel_shuffle2(g, bias, B, H, W, C):
    y = g.float().reshape(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C)
    if bias is not None:
        y = y + bias
    return y.to(g.dtype)


def pool2(x, is_max):
    xf = x.float().permute(0, 3, 1, 2)
    y = F.max_pool2d(xf, 2, 2) if is_max else F.avg_pool2d(xf, 2, 2)
    return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)


def window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def window_unpartition(win, ws, B, H, W):
    C = win.shape[-1]
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    x = win.reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    return x[:, :H, :W].contiguous()


def bilinear(x, Ho, Wo):
    return F.interpolate(x[:, None].float(), size=(Ho, Wo), mode="bilinear", align_corners=False)[:, 0]


def upsample2_add(lateral, top, out=None):
    up = top.float().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    y = (lateral.float() + up).to(lateral.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


# ---- mask post-processing / evaluation counts: the numpy restatement (oracle/postproc.py) behind the same signatures
def _np_mask(m):
    return m.detach().cpu().numpy().astype(bool)


def connected_components(mask, connectivity=8):
    from oracle import postproc as _op
    m = _np_mask(mask)
    lab, cnt = _op.connected_components(m.reshape(-1, *m.shape[-2:]), connectivity)
    return torch.from_numpy(lab.reshape(m.shape)), torch.from_numpy(cnt.reshape(m.shape))


def remove_small_blobs(mask, min_size):
    from oracle import postproc as _op
    import numpy as _n
    m = _np_mask(mask)
    out = _n.stack([_op.remove_small_blobs(x, min_size) for x in m.reshape(-1, *m.shape[-2:])]).reshape(m.shape)
    return torch.from_numpy(out.astype(_n.uint8))


def fill_holes(scores, max_area):
    from oracle import postproc as _op
    return torch.from_numpy(_op.fill_holes_in_mask_scores(scores.detach().cpu().numpy(), max_area))


def mask_pair_counts(a, b, diagonal=False):
    a, b = _np_mask(a), _np_mask(b)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    if diagonal:
        return torch.from_numpy((a & b).sum(1)), torch.from_numpy((a | b).sum(1))
    return torch.from_numpy((a[:, None] & b[None]).sum(-1)), torch.from_numpy((a[:, None] | b[None]).sum(-1))


def boundary_counts(fg, gt, radius):
    from oracle import postproc as _op
    import numpy as _n
    f, g = _np_mask(fg), _np_mask(gt)
    f, g = f.reshape(-1, *f.shape[-2:]), g.reshape(-1, *g.shape[-2:])
    return torch.from_numpy(_n.array([_op.boundary_counts(x, y, radius) for x, y in zip(f, g)], dtype=_n.int64))


# ---- image pre-processing: numpy statements of the two kernels (integer taps / IEEE normalisation)
def resample_u8(x, out_size, axis, bounds, coeffs):
    import numpy as _n
    a = _n.moveaxis(x.numpy().astype(_n.int64), 2 if axis == 1 else 1, 0)          # resized axis first
    b, k = bounds.numpy(), coeffs.numpy().astype(_n.int64)
    out = _n.empty((out_size,) + a.shape[1:], _n.uint8)
    for o in range(out_size):
        acc = _n.full(a.shape[1:], 1 << 21, _n.int64)
        for t in range(b[o, 1]):
            acc += a[b[o, 0] + t] * k[o, t]
        out[o] = _n.clip(acc >> 22, 0, 255)
    return torch.from_numpy(_n.ascontiguousarray(_n.moveaxis(out, 0, 2 if axis == 1 else 1)))


def resize_cv_linear_u8(x, Ho, Wo, xi=None, xa=None, yi=None, yb=None):
    """numpy statement of vg_resize_cv_linear_u8: the two fixed-point passes from the given index / tap tables."""
    import numpy as _n
    a = x.numpy().astype(_n.int64)
    if xi is None:
        return torch.from_numpy(((a[:, 0::2, 0::2] + a[:, 0::2, 1::2] + a[:, 1::2, 0::2] + a[:, 1::2, 1::2] + 2) >> 2).astype(_n.uint8))
    xi, xa, yi, yb = (t.numpy().astype(_n.int64) for t in (xi, xa, yi, yb))
    rows = a[:, :, xi[:, 0]] * xa[None, None, :, 0, None] + a[:, :, xi[:, 1]] * xa[None, None, :, 1, None]
    out = (((yb[None, :, 0, None, None] * (rows[:, yi[:, 0]] >> 4)) >> 16) + ((yb[None, :, 1, None, None] * (rows[:, yi[:, 1]] >> 4)) >> 16) + 2) >> 2
    return torch.from_numpy(out.astype(_n.uint8))


def normalize_u8(x, mean, std, mode, crop=None, out_dtype=torch.float32):
    import numpy as _n
    top, left, h, w = crop if crop is not None else (0, 0, x.shape[1], x.shape[2])
    a = x.numpy()[:, top:top + h, left:left + w]
    if mode == 0:
        y = (a.astype(_n.float32) - _n.asarray(mean, _n.float32)) / _n.asarray(std, _n.float32)
    else:
        y = ((a.astype(_n.float64) / 255.0 - _n.asarray(mean, _n.float64)) / _n.asarray(std, _n.float64)).astype(_n.float32)
    return torch.from_numpy(_n.ascontiguousarray(_n.transpose(y, (0, 3, 1, 2)))).to(out_dtype)


def mlp3_pack(w):
    G, n_out, n_in = w.shape
    T = -(-n_out // 32)
    wp = torch.zeros(G, T * 32, n_in, dtype=w.dtype, device=w.device)
    wp[:, :n_out] = w
    return wp.view(G, T, 32, n_in // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().view(G, T, n_in // 16, 64, 8)


def _mlp3_unpack(wp, n_out):
    G, T, nk = wp.shape[:3]
    return wp.view(G, T, nk, 2, 32, 8).permute(0, 1, 4, 2, 3, 5).reshape(G, T * 32, nk * 16)[:, :n_out]


def mlp3_grouped(x, G, w0, b0, w1, b1, w2, b2, out, sigmoid_mask=0):
    Hd, No = b0.shape[1], b2.shape[1]
    w0, w1, w2 = _mlp3_unpack(w0, Hd), _mlp3_unpack(w1, Hd), _mlp3_unpack(w2, No)
    for g in range(G):
        h = F.relu(x[:, g, :].float() @ w0[g].float().t() + b0[g]).to(x.dtype)
        h = F.relu(h.float() @ w1[g].float().t() + b1[g]).to(x.dtype)
        y = h.float() @ w2[g].float().t() + b2[g]
        if (sigmoid_mask >> g) & 1:
            y = torch.sigmoid(y)
        out[:, g, :No] = y.to(out.dtype)
    return out


ALL = [n for n, f in list(globals().items()) if callable(f) and not n.startswith("_") and n not in ("torch", "F")]
