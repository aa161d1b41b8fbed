"""SAM2 promptable pixel decoder on the MI355X kernel library (rows S1–S11 of SURVEY.md §8a).

Host-side graph only: every arithmetic step is a C-ABI kernel call through ``ops``; torch is used for
device allocation, views, concatenation and indexing.  Layout is channels-last everywhere
([B,H,W,C] == [B, tokens, C]), so 1x1 convs / LayerNorm2d / FPN laterals are plain row GEMMs / row norms
and the reference's NCHW<->(HW)NC permutes (R/.../sam2/modeling/sam2_base.py:479-493) disappear.
R/ = /root/reference/VideoGLaMM/model/segment_anything_2/sam2/.
"""
import math
import os

import torch

from . import ops

NO_OBJ_SCORE = -1024.0  # R/modeling/sam2_base.py:17
_SELFATTN_FUSED = True     # memory self-attention: fused q|k|v projection + one RoPE launch
_MEMENC_FUSED = True       # memory encoder: fused conv + LayerNorm2d + GELU stages
_MEMATTN_LOWRANK = True    # memory cross-attention: v-projection behind the attention
# (module flags, not environment knobs: tests/test_host_sam2.py switches them off together to run the reference's own order of operations)
_MEMBANK = True      # r05: the propagation's memories and object pointers live in ONE preallocated bank per clip (no per-frame cat / copies) and the
#                      memory attention runs on the fused short-row GEMMs (vg_gemm_rows); False = r04's per-frame assembly (tests switch it off together
#                      with the flags above to run the reference's own order of operations)
PTR_ROWS = 64        # 16 object pointers x 4 tokens of 64 channels (max_obj_ptrs_in_encoder, sam2_base.py:590-626)


def _stack_views(ts):
    """cat(ts, 0) without the copy when ts are consecutive [1,...] slices of one batched tensor (what hiera_frames
    hands out): 67 MB of FPN features per 8-frame chunk would otherwise take another trip through HBM."""
    t0 = ts[0]
    step = t0.stride(0) * t0.element_size()
    if all(t.shape == t0.shape and t.stride() == t0.stride() and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr()
           and t.data_ptr() == t0.data_ptr() + j * step for j, t in enumerate(ts)) and t0.shape[0] == 1:
        return t0.as_strided((len(ts),) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())
    return torch.cat(ts, dim=0)


def hiera_layout(cfg):
    """(dim, dim_out, heads, window, q_stride) per block — R/modeling/backbones/hieradet.py:196-259."""
    stages, window_spec = cfg["stages"], cfg["window_spec"]
    stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][: cfg.get("q_pool", 3)]
    dim, heads, cur = cfg["embed_dim"], cfg["num_heads"], 1
    blocks = []
    for i in range(sum(stages)):
        dim_out, window = dim, window_spec[cur - 1]
        if i in cfg["global_att_blocks"]:
            window = 0
        if i - 1 in stage_ends:
            dim_out, heads, cur = dim * 2, heads * 2, cur + 1
        blocks.append(dict(dim=dim, dim_out=dim_out, heads=heads, window=window, q_stride=2 if i in q_pool_blocks else 0))
        dim = dim_out
    return blocks, stage_ends


def _sine_pos(num_pos_feats, h, w, temperature=10000.0):
    """PositionEmbeddingSine (normalize=True) as an [h*w, C] table — R/modeling/position_encoding.py:78-111."""
    npf = num_pos_feats // 2
    y = torch.arange(1, h + 1, dtype=torch.float32).view(-1, 1).repeat(1, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, -1).repeat(h, 1)
    y = y / (y[-1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, num_pos_feats)


def _axial_cos_sin(dim, side, theta=10000.0):
    """compute_axial_cis as cos/sin tables [side*side, dim/2] — R/modeling/position_encoding.py:174-191."""
    fx = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    t = torch.arange(side * side, dtype=torch.float32)
    tx, ty = (t % side).float(), torch.div(t, side, rounding_mode="floor").float()
    ang = torch.cat([torch.outer(tx, fx), torch.outer(ty, fx)], dim=-1)
    return ang.cos(), ang.sin()


class SAM2:
    def __init__(self, params, prefix, cfg):
        """params: Params over the checkpoint; prefix: e.g. "model.visual_model."; cfg: {image_size, trunk{...}}."""
        self.P, self.p, self.cfg = params, prefix, cfg
        self.dtype, self.device = params.dtype, params.device
        self.S = cfg["image_size"]
        self.es = self.S // 16
        self.blocks, self.stage_ends = hiera_layout(cfg["trunk"])
        # frames batched per Hiera launch group (r02, 32-frame clip: 16 frames 145.8 ms, 8 frames 156.2, 4 frames 171.8: bigger
        # batches fill the tails of the 128-tile grids) and per framewise mask-decoder launch group (its GEMMs are M = 4096 rows
        # per (frame, object): the more pairs per launch the better)
        self.frame_chunk, self.decode_chunk = 16, 128

    def video_static_feats(self, n):
        """persistent per-level buffers [n, h, w, c] the graph-replayed propagation reads (video_branch_graphed): handed to hiera_frames as `bufs`,
        Hiera's last kernels write the clip's features straight into them — the replay then has nothing to copy (r05: 96 staging copies = 268 MB
        per 32-frame clip before).  The features hiera_frames returns for these buffers are views of them: valid until the next clip of this length
        is encoded.  Buffers of lengths without a cached graph are dropped when the graph cache evicts (video_branch_graphed)."""
        st = self.__dict__.setdefault("_video_static", {})
        if n not in st:
            st[n] = [torch.empty((n,) + shp, dtype=self.dtype, device=self.device) for shp in self.level_shapes()]
        return st[n]

    def hiera_frames(self, images, frames=None, bufs=None):
        """forward_image over many frames in chunks -> list (per frame) of [1,h,w,c] level views.  bufs: optional per-level [len(frames), h, w, c]
        destinations (video_static_feats)."""
        frames = list(range(images.shape[0])) if frames is None else frames
        out = {}
        # ONE buffer per level for all requested frames: the chunks' last kernels write straight into their rows, so consecutive frames stay
        # consecutive views across chunk borders and the consumers' batches (_stack_views) never copy (r03: a 32-pair mask-decoder batch over two
        # 16-frame chunks cat 268 MB of FPN levels per C2 clip)
        n = len(frames)
        if bufs is None:
            bufs = [torch.empty((n,) + shp, dtype=self.dtype, device=self.device) for shp in self.level_shapes()]
        for c0 in range(0, n, self.frame_chunk):
            fr = frames[c0:c0 + self.frame_chunk]
            fpn = self.forward_image(images[fr[0]:fr[-1] + 1] if fr == list(range(fr[0], fr[-1] + 1)) else images[fr],
                                     out=[b[c0:c0 + len(fr)] for b in bufs])
            for j, t in enumerate(fr):
                out[t] = [f[j:j + 1] for f in fpn]
        return out

    # ------------------------------------------------------------------ small helpers
    def level_shapes(self):
        """(h, w, c) of the three FPN levels forward_image returns, channel widths read off the weights that produce them (conv_s0, conv_s1, the
        neck's top lateral): hiera_frames' clip-wide buffers and dist.py's exchanges are sized from here, not from SAM2-L's 32 / 64 / 256."""
        S = self.S
        c0 = self.P.w(self.p + "sam_mask_decoder.conv_s0").shape[0]
        c1 = self.P.w(self.p + "sam_mask_decoder.conv_s1").shape[0]
        c2 = self.P.w(self.p + "image_encoder.neck.convs.0.conv").shape[0]
        return [(S // 4, S // 4, c0), (S // 8, S // 8, c1), (S // 16, S // 16, c2)]

    def lin(self, name, x, **kw):
        return ops.linear(x, self.P.w(self.p + name), self.P.b(self.p + name), **kw)

    def ln(self, name, x, eps=1e-5, **kw):
        return ops.layernorm(x, self.P.f32(self.p + name + ".weight"), self.P.f32(self.p + name + ".bias"), eps, **kw)

    def mlp(self, name, x, n, act=ops.ACT_RELU, sigmoid_output=False, out_dtype=None, out=None):
        """R/modeling/sam2_utils.py:108-132.  out: optional destination of the last layer (a row-strided 2-D view)."""
        for i in range(n):
            last = i == n - 1
            x = self.lin(f"{name}.layers.{i}", x, act=(ops.ACT_SIGMOID if (last and sigmoid_output) else (ops.ACT_NONE if last else act)),
                         out_dtype=out_dtype if last else None, out=out if last else None)
        return x

    def _heads_grouped(self):
        """the small three-layer heads ride vg_mlp3_grouped in the bf16 mode (fp32 parity mode keeps one vg_gemm per layer)"""
        return self.dtype == torch.bfloat16

    def _mlp3_stack(self, names):
        """the parameters of several equally shaped three-layer MLPs stacked for vg_mlp3_grouped: (w0 [G,Hd,K], b0, w1, b1, w2 [G,No,Hd], b2), computed once"""
        def make(i, bias):
            stack = lambda: torch.stack([self.P.sd[f"{self.p}{n}.layers.{i}.{'bias' if bias else 'weight'}"].float() for n in names])      # noqa: E731
            return stack if bias else (lambda: ops.mlp3_pack(stack()))      # (weights in the kernel's fragment order)
        key = tuple(names)
        return tuple(self.P.const(("mlp3", key, i, bias), make(i, bias), dtype=torch.float32 if bias else None) for i in range(3) for bias in (False, True))

    # ------------------------------------------------------------------ S1 Hiera + FPN
    def _hiera_pos(self, h, w):
        def make():
            sd, p = self.P.sd, self.p + "image_encoder.trunk."
            pe = torch.nn.functional.interpolate(sd[p + "pos_embed"].float().cpu(), size=(h, w), mode="bicubic")
            we = sd[p + "pos_embed_window"].float().cpu()
            pe = pe + we.tile([x // y for x, y in zip(pe.shape, we.shape)])
            return pe.permute(0, 2, 3, 1).reshape(h * w, -1)
        return self.P.const(("hiera_pos", h, w), make)

    def _hiera_block(self, i, blk, x):
        """MultiScaleBlock — R/modeling/backbones/hieradet.py:37-168.  x: [B,H,W,C]."""
        p = f"image_encoder.trunk.blocks.{i}."
        B, H, W, _ = x.shape
        do, nh = blk["dim_out"], blk["heads"]
        hd = do // nh
        shortcut = x
        ws = blk["window"]
        wq, bq = self.P.w(self.p + p + "attn.qkv"), self.P.b(self.p + p + "attn.qkv")
        if blk["dim"] != do:
            xn = self.ln(p + "norm1", x, 1e-6)           # two consumers (the shortcut's projection and q|k|v): the norm is its own launch
            shortcut = self.lin(p + "proj", xn)
            if blk["q_stride"]:
                shortcut = ops.pool2(shortcut, True)
            # window_partition rides in the qkv GEMM's A-row gather (padding rows read zeros, like the reference's F.pad)
            qkv = ops.linear_window(xn, wq, bq, B, H, W, ws, scatter=False) if ws > 0 else self.lin(p + "attn.qkv", xn.view(B, H * W, -1))
        else:
            # norm1 -> q|k|v (+ the window gather) as ONE launch at the widths of stages 1 / 2 (ops.linear_ln: vg_gemm_ln), two launches elsewhere
            ln1 = (self.P.f32(self.p + p + "norm1.weight"), self.P.f32(self.p + p + "norm1.bias"), 1e-6)
            qkv = ops.linear_ln(x, ln1, wq, bq, window=(B, H, W, ws)) if ws > 0 else ops.linear_ln(x.view(B, H * W, -1), ln1, wq, bq)
        if ws > 0:
            h = w = ws
            Bw = qkv.shape[0]
        else:
            h, w, Bw = H, W, B
        qkv = qkv.view(Bw, h * w, 3, nh, hd)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        if blk["q_stride"]:
            q = ops.pool2(q.reshape(Bw, h, w, do) if q.is_contiguous() else qkv.view(Bw, h, w, 3 * do)[..., :do], True)
            h, w = h // 2, w // 2
            q = q.view(Bw, h * w, nh, hd)
        if ws > 0 and not blk["q_stride"]:
            o = ops.attention_windows(q, k, v, hd ** -0.5).view(Bw, h * w, do)    # per-window kernels / windows packed into query tiles
        else:
            o = ops.window_attention(q, k, v, hd ** -0.5) if ws > 0 else None    # q-pooled windows (4 x 16, 16 x 64) on their own kernel
            o = (o if o is not None else ops.attention(q, k, v, hd ** -0.5)).view(Bw, h * w, do)
        Hs, Ws = shortcut.shape[1:3]
        if ws > 0:
            # proj + window_unpartition + residual add in one pass (rows scattered to their pixels by the epilogue)
            wse = ws // 2 if blk["q_stride"] else ws
            x = ops.linear_window(o, self.P.w(self.p + p + "attn.proj"), self.P.b(self.p + p + "attn.proj"), B, Hs, Ws, wse,
                                  scatter=True, residual=shortcut)
        else:
            x = self.lin(p + "attn.proj", o, residual=shortcut.view(B, Hs * Ws, do)).view(B, Hs, Ws, do)
        # x + mlp(norm2(x)): one launch at the widths of stages 1 and 2 (vg_mlp_rows; bf16), three launches otherwise (ops.mlp_rows falls back)
        return ops.mlp_rows(x, (self.P.f32(self.p + p + "norm2.weight"), self.P.f32(self.p + p + "norm2.bias"), 1e-6),
                            self.P.w(self.p + p + "mlp.layers.0"), self.P.b(self.p + p + "mlp.layers.0"),
                            self.P.w(self.p + p + "mlp.layers.1"), self.P.b(self.p + p + "mlp.layers.1"))

    def forward_image(self, img, out=None):
        """SAM2Base.forward_image (Hiera + FpnNeck scalp=1 + conv_s0/s1) — R/modeling/sam2_base.py:465-477,
        backbones/image_encoder.py:29-42,101-133.  img: [B,3,S,S] (any float dtype, NCHW like the reference's
        preprocessing emits) -> fpn = [[B,S/4,S/4,32], [B,S/8,S/8,64], [B,S/16,S/16,256]] channels-last."""
        B, _, H, W = img.shape
        x = ops.permute5(img.contiguous(), (B, H, W, 3, 1), (3 * H * W, W, 1, H * W, 0)).view(B, H, W, 3)
        x = ops.cast(x, self.dtype)
        t = self.p + "image_encoder.trunk."
        C0 = self.cfg["trunk"]["embed_dim"]
        wpe = self.P.conv_w(t + "patch_embed.proj")
        cols, Ho, Wo = ops.im2col(x, 7, 7, 4, 3, wpe.shape[1])
        x = ops.linear(cols, wpe, self.P.b(t + "patch_embed.proj")).view(B, Ho, Wo, C0)
        x = ops.add(x, self._hiera_pos(Ho, Wo))
        feats = []
        for i, blk in enumerate(self.blocks):
            x = self._hiera_block(i, blk, x)
            if i in self.stage_ends:
                feats.append(x)
        n = len(feats) - 1
        dst = out                                   # optional [B, ...] destination per returned level (hiera_frames' clip-wide buffers)
        out, prev = [None] * len(feats), None
        for i in range(n, -1, -1):
            top_dst = dst[2] if (dst is not None and i == 2) else None      # level 2 is returned as the neck makes it
            if i in (2, 3) and prev is not None:
                prev = ops.upsample2_add(self.lin(f"image_encoder.neck.convs.{n - i}.conv", feats[i]), prev, out=top_dst)
            else:
                prev = self.lin(f"image_encoder.neck.convs.{n - i}.conv", feats[i], out=top_dst)
            out[i] = prev
        out = out[:-1]
        out[0] = self.lin("sam_mask_decoder.conv_s0", out[0], out=None if dst is None else dst[0])
        out[1] = self.lin("sam_mask_decoder.conv_s1", out[1], out=None if dst is None else dst[1])
        return out

    def vision_pos(self):
        """vision_pos_enc of the top level as [es*es, 256] (constant)."""
        return self.P.const(("vision_pos", self.es), lambda: _sine_pos(256, self.es, self.es))

    # ------------------------------------------------------------------ S3 prompt encoder
    def dense_pe(self):
        """PromptEncoder.get_dense_pe as [es*es, 256] — R/modeling/sam/prompt_encoder.py:68-77,216-228 (constant)."""
        def make():
            g = self.P.sd[self.p + "sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].float().cpu()
            grid = torch.ones((self.es, self.es))
            y, x = (grid.cumsum(0) - 0.5) / self.es, (grid.cumsum(1) - 0.5) / self.es
            c = 2 * math.pi * ((2 * torch.stack([x, y], dim=-1) - 1) @ g)
            return torch.cat([c.sin(), c.cos()], dim=-1).reshape(self.es * self.es, 256)
        return self.P.const("dense_pe", make)

    def sparse_prompt(self, n, text_embeds, with_empty_point):
        """PromptEncoder.forward sparse part — prompt_encoder.py:143-189 (+ sam2_base.py:310-313 padding points)."""
        if with_empty_point and text_embeds is None:
            # the tracked frames' prompt (two padding points, no text) is a constant of the model: made once per object count, and mask_decoder
            # recognises it (its token tensor is then a constant too — no per-frame cat / copy)
            def make():
                nap = self.P.t(self.p + "sam_prompt_encoder.not_a_point_embed.weight")
                return nap.view(1, 1, 256).expand(n, 2, 256).contiguous()
            t = self.P.const(("sparse_empty", n), make)
            t._vg_const_prompt = True      # an explicit mark on THIS tensor object (a view or a slice of it does not carry it); read-only by contract
            return t
        parts = []
        if with_empty_point:
            nap = self.P.t(self.p + "sam_prompt_encoder.not_a_point_embed.weight")
            parts.append(nap.view(1, 1, 256).expand(n, 2, 256))
        if text_embeds is not None:
            parts.append(ops.cast(text_embeds, self.dtype))
        return torch.cat(parts, dim=1).contiguous()

    # ------------------------------------------------------------------ S7/S8 mask decoder
    def _attn(self, name, q, k, v, heads, residual=None):
        """sam/transformer.py Attention.forward :236-260 (+ the caller's residual add fused into out_proj)."""
        q, k, v = self.lin(name + ".q_proj", q), self.lin(name + ".k_proj", k), self.lin(name + ".v_proj", v)
        B, nq, c = q.shape
        hd = c // heads
        o = ops.attention(q.view(B, nq, heads, hd), k.view(B, -1, heads, hd), v.view(B, -1, heads, hd), hd ** -0.5)
        return self.lin(name + ".out_proj", o.view(B, nq, c), residual=residual)

    def _two_way(self, src, tokens):
        """TwoWayTransformer — R/modeling/sam/transformer.py:69-115,160-193.  src [N,HW,256], tokens [N,nt,256]."""
        t = "sam_mask_decoder.transformer."
        key_pe = self.dense_pe()
        queries, keys, query_pe = tokens, src, tokens
        for i in range(2):
            l = f"{t}layers.{i}."
            if i == 0:
                queries = self._attn(l + "self_attn", queries, queries, queries, 8)
            else:
                q = ops.add(queries, query_pe)
                queries = self._attn(l + "self_attn", q, q, queries, 8, residual=queries)
            queries = self.ln(l + "norm1", queries)
            q, k = ops.add(queries, query_pe), ops.add(keys, key_pe)
            queries = self.ln(l + "norm2", self._attn(l + "cross_attn_token_to_image", q, k, keys, 8, residual=queries))
            h = self.lin(l + "mlp.layers.0", queries, act=ops.ACT_RELU)
            queries = self.ln(l + "norm3", self.lin(l + "mlp.layers.1", h, residual=queries))
            q = ops.add(queries, query_pe)
            keys = self.ln(l + "norm4", self._attn(l + "cross_attn_image_to_token", k, q, queries, 8, residual=keys))
        q, k = ops.add(queries, query_pe), ops.add(keys, key_pe)
        queries = self.ln(t + "norm_final_attn", self._attn(t + "final_attn_token_to_image", q, k, keys, 8, residual=queries))
        return queries, keys

    # ---- fused two-way transformer (r04): the image side of a block in TWO passes over the 4096 x 256 keys instead of a dozen.
    # Algebra (exact in real arithmetic; sam/transformer.py:236-260 Attention with 8 heads of 16 channels, internal dim 128):
    #   token -> image:  scores[h,t,r] = scale * q_h[t] . ((x_r + pe_r) Wk_h^T + bk_h)  =  (x_r + pe_r) . u[h,t]  + const(h,t)   with
    #                    u[h,t] = scale * Wk_h^T q_h[t]  in R^256 — the constant cancels in the softmax over r — and
    #                    sum_r a[h,t,r] (x_r Wv_h^T + bv_h) = (sum_r a[h,t,r] x_r) Wv_h^T + bv_h:  ONE single-head attention of head dim 256 with the
    #                    8 nt vectors u as queries, x + pe as keys and x as values (vg_attention's head-dim-256 kernel), then 16 channels per (h, t).
    #   image -> token:  scores[r,h,t] = (x_r + pe_r) . u2[h,t] + c2[h,t]  (u2 = scale Wq_h^T k_h[t], c2 = scale bq_h . k_h[t]), softmax over t per head,
    #                    out_proj(sum_t a v_h[t]) = a[r,:] . W2 with W2[(h,t)] = Wo[:, head h] v_h[t] in R^256: vg_twoway_image_update does the two small
    #                    GEMMs, the per-head softmax, the residual, LayerNorm and the + pe of the next pass in one kernel.
    # The k / v / q projections of the 4096 image rows (5 x 4096 x 256 x 128 MACs per block) are gone; the token side keeps the library's small GEMMs.
    def _heads_bd(self, x, TP):
        """[N, nt, 128] -> block-diagonal [N * 8 * TP, 128]: row (h, t) holds head h's 16 channels of token t, zeros elsewhere (one launch)."""
        return ops.heads_blockdiag(x, TP)

    def _tw_const(self, name, kind):
        w = lambda n: self.P.sd[self.p + name + n + ".weight"].float()      # noqa: E731
        b = lambda n: self.P.sd[self.p + name + n + ".bias"].float()        # noqa: E731
        scale = 16 ** -0.5
        if kind == "ukT":        # [256, 128]: U = Qbd . (scale Wk)  ->  linear weight[d][k] = scale Wk[k][d]
            return self.P.const((name, kind), lambda: scale * w(".k_proj").t())
        if kind == "uqT":        # image -> token: scale Wq^T
            return self.P.const((name, kind), lambda: scale * w(".q_proj").t())
        if kind == "cq":         # [1, 128]: scale bq, the single input row of c2 = (scale bq) . Kbd^T
            return self.P.const((name, kind), lambda: scale * b(".q_proj")[None])
        raise KeyError(kind)

    def _t2i_fused(self, name, queries, query_pe, xpe, x, TP):
        N, nt = queries.shape[0], queries.shape[1]
        NC, P, Bi = 8 * TP, x.shape[1], x.shape[0]
        q_tok = self.lin(name + ".q_proj", ops.add(queries, query_pe))
        U = ops.linear(self._heads_bd(q_tok, TP), self._tw_const(name, "ukT")).view(N // Bi, Bi, NC, 1, 256)
        # instance i attends image i % Bi: one attention call per group of Bi instances (Bi = N after the first block: a single call)
        zn = [ops.attention(U[g], xpe.view(Bi, P, 1, 256), x.view(Bi, P, 1, 256), 1.0) for g in range(N // Bi)]     # softmax-weighted means of the image rows
        zn = zn[0] if len(zn) == 1 else torch.cat(zn, dim=0)
        o_full = self.lin(name + ".v_proj", zn.view(N * NC, 256))                                   # (+ bv: the weights of a (h, t) sum to one)
        o = ops.heads_blockdiag_gather(o_full, N, nt, TP)                                           # head h's channels of row (h, t)
        return self.lin(name + ".out_proj", o, residual=queries)

    def _i2t_fused(self, name, norm, queries, query_pe, xpe, x, pe, TP):
        N, nt = queries.shape[0], queries.shape[1]
        NC = 8 * TP
        k_bd = self._heads_bd(self.lin(name + ".k_proj", ops.add(queries, query_pe)), TP)
        v_bd = self._heads_bd(self.lin(name + ".v_proj", queries), TP)
        u2 = ops.linear(k_bd, self._tw_const(name, "uqT")).view(N, NC, 256)
        # r06: both come out of their GEMM in the layout the kernel reads — c2 as the ONE output row of (scale bq) . Kbd^T (the roles of rows and weights
        # swapped: a contiguous fp32 [N NC]), W2^T [N, 256, NC] as a batched Wo . Vbd[n]^T with Wo shared over the batch (stride 0) — instead of a strided
        # column copy and a transpose copy (two ATen launches per block, four per tracked frame)
        c2 = ops.linear(self._tw_const(name, "cq"), k_bd, out_dtype=torch.float32).view(N, NC)
        w2t = ops.bmm_nt(self.P.w(self.p + name + ".out_proj"), v_bd.view(N, NC, -1), shared_a=True)                    # [N, 256, NC]
        return ops.twoway_image_update(xpe, x, u2, c2, w2t, self.P.b(self.p + name + ".out_proj"), self.P.f32(self.p + norm + ".weight"),
                                       self.P.f32(self.p + norm + ".bias"), 1e-5, pe, nt, TP)

    def _two_way_fused(self, src, tokens):
        """TwoWayTransformer (R/modeling/sam/transformer.py:69-115,160-193) with the image side fused — see the comment block above.
        src [Bi, HW, 256] with Bi | N: instance i works on image i % Bi; the first block reads the shared embeddings, its image -> token pass
        writes one updated copy per instance."""
        t = "sam_mask_decoder.transformer."
        key_pe = self.dense_pe()
        nt = tokens.shape[1]
        TP = 8 if nt <= 8 else 16
        queries, keys, query_pe = tokens, src, tokens
        kpe = ops.add(keys, key_pe)
        for i in range(2):
            l = f"{t}layers.{i}."
            if i == 0:
                queries = self._attn(l + "self_attn", queries, queries, queries, 8)
            else:
                q = ops.add(queries, query_pe)
                queries = self._attn(l + "self_attn", q, q, queries, 8, residual=queries)
            queries = self.ln(l + "norm1", queries)
            queries = self.ln(l + "norm2", self._t2i_fused(l + "cross_attn_token_to_image", queries, query_pe, kpe, keys, TP))
            h = self.lin(l + "mlp.layers.0", queries, act=ops.ACT_RELU)
            queries = self.ln(l + "norm3", self.lin(l + "mlp.layers.1", h, residual=queries))
            keys, kpe = self._i2t_fused(l + "cross_attn_image_to_token", l + "norm4", queries, query_pe, kpe, keys, key_pe, TP)
        queries = self.ln(t + "norm_final_attn", self._t2i_fused(t + "final_attn_token_to_image", queries, query_pe, kpe, keys, TP))
        return queries, keys

    def mask_decoder(self, image_embed, sparse, high_res, repeat_image):
        """MaskDecoder.predict_masks — R/modeling/sam/mask_decoder.py:168-245.
        image_embed [Bi,HW,256], sparse [N,ns,256], high_res = (s0 [Bi,(4es)^2,32], s1 [Bi,(2es)^2,64]) with Bi | N: instance i decodes image
        i % Bi (Bi = N: one image per instance; Bi = 1 = the reference's repeat_image; 1 < Bi < N: the framewise branch's (object, frame) pairs,
        object-major — r04: a frame's objects share its embedding and high-resolution features instead of reading per-object copies).
        -> masks fp32 [N,4,4es,4es], iou fp32 [N,4], mask tokens [N,4,256], object score logits fp32 [N,1]."""
        d = "sam_mask_decoder."
        N, es, Bi = sparse.shape[0], self.es, image_embed.shape[0]
        assert N % Bi == 0 and (repeat_image or Bi > 1 or N == 1)
        out_tokens = self.P.const("mask_decoder_out_tokens", lambda: torch.cat([self.P.t(self.p + d + "obj_score_token.weight"), self.P.t(self.p + d + "iou_token.weight"),
                                                                                 self.P.t(self.p + d + "mask_tokens.weight")], dim=0))
        mk_tokens = lambda: torch.cat([out_tokens.unsqueeze(0).expand(N, -1, -1), sparse], dim=1).contiguous()      # noqa: E731
        # the constant empty prompt (sparse_prompt marks it) has a constant token tensor: cached per object count, never written in place (queries / query_pe)
        tokens = self.P.const(("mask_decoder_tokens_empty", N), mk_tokens) if getattr(sparse, "_vg_const_prompt", False) else mk_tokens()
        no_mask = self.P.t(self.p + "sam_prompt_encoder.no_mask_embed.weight").view(-1)
        src = ops.add(image_embed, no_mask).view(Bi, es * es, 256)  # dense prompt = no_mask_embed broadcast (prompt_encoder.py:183-187)
        # the fused image side needs the bf16 kernels (vg_twoway_image_update, head-dim-256 attention); fp32 parity mode keeps the reference's order
        fused = self.dtype == torch.bfloat16 and os.environ.get("VG_TWOWAY_FUSED", "1") != "0" or os.environ.get("VG_TWOWAY_FUSED") == "2"
        if fused:
            hs, src = self._two_way_fused(src, tokens)
        else:
            if Bi < N:      # the unfused chain works on one image copy per instance
                src = src.unsqueeze(0).expand(N // Bi, Bi, es * es, 256).reshape(N, es * es, 256)
            hs, src = self._two_way(src, tokens)
        iou_tok, mask_toks = hs[:, 1, :], hs[:, 2:6, :]
        s0, s1 = high_res            # [Bi, ...]: ops.add broadcasts them over the instance groups (instance i -> image i % Bi)
        nhy = self.P.w(self.p + f"{d}output_hypernetworks_mlps.0.layers.2").shape[0]
        hyper = torch.empty(N, 4, nhy, dtype=hs.dtype, device=hs.device)
        if self._heads_grouped():      # r06: the four hypernetwork MLPs in one launch (vg_mlp3_grouped) instead of twelve
            ops.mlp3_grouped(mask_toks, 4, *self._mlp3_stack([f"{d}output_hypernetworks_mlps.{i}" for i in range(4)]), out=hyper)
        else:
            for i in range(4):       # (row-strided views in and out: no per-token copies, no stack)
                self.mlp(f"{d}output_hypernetworks_mlps.{i}", mask_toks[:, i, :], 3, out=hyper[:, i, :])
        if fused:
            # r04: upscaling + hypernetwork product in one kernel (the two [N, 16384, 64] / [N, 65536, 32] intermediates never exist)
            masks = ops.mask_upscale(src, self.P.convT_w(self.p + d + "output_upscaling.0"), self.P.b(self.p + d + "output_upscaling.0"), s1.view(Bi, 4 * es * es, 64),
                                     self.P.f32(self.p + d + "output_upscaling.1.weight"), self.P.f32(self.p + d + "output_upscaling.1.bias"), 1e-6,
                                     self.P.convT_w(self.p + d + "output_upscaling.3"), self.P.b(self.p + d + "output_upscaling.3"), s0.view(Bi, 16 * es * es, 32),
                                     hyper, es)
        else:
            g = ops.linear(src, self.P.convT_w(self.p + d + "output_upscaling.0"))
            up = ops.pixel_shuffle2(g, self.P.b(self.p + d + "output_upscaling.0"), N, es, es, 64)
            up = ops.add(up, s1)
            up = ops.activation(self.ln(d + "output_upscaling.1", up, 1e-6), ops.ACT_GELU)
            g = ops.linear(up, self.P.convT_w(self.p + d + "output_upscaling.3"))
            up = ops.pixel_shuffle2(g, self.P.b(self.p + d + "output_upscaling.3"), N, 2 * es, 2 * es, 32)
            up = ops.activation(ops.add(up, s0), ops.ACT_GELU)                       # [N,4es,4es,32]
            masks = ops.bmm_nt(hyper, up.view(N, 16 * es * es, 32), out_dtype=torch.float32).view(N, 4, 4 * es, 4 * es)
        if self._heads_grouped():      # one launch per head instead of three (their outputs stay contiguous [N, 4] / [N, 1] for the selection kernels)
            iou = torch.empty(N, 1, 4, dtype=torch.float32, device=hs.device)
            obj = torch.empty(N, 1, 1, dtype=torch.float32, device=hs.device)
            ops.mlp3_grouped(hs[:, 1:2, :], 1, *self._mlp3_stack([d + "iou_prediction_head"]), out=iou, sigmoid_mask=1)
            ops.mlp3_grouped(hs[:, 0:1, :], 1, *self._mlp3_stack([d + "pred_obj_score_head"]), out=obj)
            iou, obj = iou.view(N, 4), obj.view(N, 1)
        else:
            iou = self.mlp(d + "iou_prediction_head", iou_tok, 3, sigmoid_output=True, out_dtype=torch.float32)
            obj = self.mlp(d + "pred_obj_score_head", hs[:, 0, :], 3, out_dtype=torch.float32)
        return masks, iou, mask_toks.contiguous(), obj

    # ------------------------------------------------------------------ S5 memory attention
    def _rope_attn(self, name, q_in, k_in, v_in, n_exclude, residual):
        """RoPEAttention.forward (1 head) — R/modeling/sam/transformer.py:289-327."""
        if _SELFATTN_FUSED and q_in is k_in and k_in is v_in and n_exclude == 0 and q_in.dtype == torch.bfloat16:
            # self-attention (r04): ONE q|k|v projection (the rows are read once), ONE RoPE launch over the q | k columns of it, attention on views
            w, b = self.P.fused([self.p + name + ".q_proj", self.p + name + ".k_proj", self.p + name + ".v_proj"])
            qkv = ops.linear(q_in, w, b)
            B, nq, c3 = qkv.shape
            c = c3 // 3
            cos, sin = self.P.const(("axial_cos", c, nq), lambda: _axial_cos_sin(c, int(math.sqrt(nq)))[0], torch.float32), \
                self.P.const(("axial_sin", c, nq), lambda: _axial_cos_sin(c, int(math.sqrt(nq)))[1], torch.float32)
            ops.rope_axial_heads_(qkv, 2, cos, sin, nq, nq)
            o = ops.attention(qkv[..., :c].unsqueeze(2), qkv[..., c:2 * c].unsqueeze(2), qkv[..., 2 * c:].unsqueeze(2), c ** -0.5)
            return self.lin(name + ".out_proj", o.view(B, nq, c), residual=residual)
        q, k, v = self.lin(name + ".q_proj", q_in), self.lin(name + ".k_proj", k_in), self.lin(name + ".v_proj", v_in)
        B, nq, c = q.shape
        cos, sin = self.P.const(("axial_cos", c, nq), lambda: _axial_cos_sin(c, int(math.sqrt(nq)))[0], torch.float32), \
            self.P.const(("axial_sin", c, nq), lambda: _axial_cos_sin(c, int(math.sqrt(nq)))[1], torch.float32)
        ops.rope_axial_(q, cos, sin, nq, nq)
        ops.rope_axial_(k, cos, sin, k.shape[1] - n_exclude, nq)
        o = ops.attention(q.view(B, nq, 1, c), k.view(B, -1, 1, c), v.view(B, -1, 1, c), c ** -0.5)
        return self.lin(name + ".out_proj", o.view(B, nq, c), residual=residual)

    def _rope_attn_lowrank(self, name, q_in, k_in, mem, n_exclude, residual):
        """The memory cross-attention with the v-projection BEHIND the attention (r04): the reference computes softmax(q k^T) (M Wv^T + b_v)
        (R/modeling/sam/transformer.py:289-327 with kv_in_dim = 64, memory_attention.py:60-99); the softmax rows sum to one, so that is
        (softmax(q k^T) M) Wv^T + b_v — the attention's PV half then works on the memory's own 64 dims (vg_attention_dv: a quarter of the PV
        MFMAs and V bytes) and the projection runs on the nq query rows instead of the ~7 x nq memory rows."""
        q, k = self.lin(name + ".q_proj", q_in), self.lin(name + ".k_proj", k_in)
        B, nq, c = q.shape
        cos, sin = self.P.const(("axial_cos", c, nq), lambda: _axial_cos_sin(c, int(math.sqrt(nq)))[0], torch.float32), \
            self.P.const(("axial_sin", c, nq), lambda: _axial_cos_sin(c, int(math.sqrt(nq)))[1], torch.float32)
        ops.rope_axial_(q, cos, sin, nq, nq)
        ops.rope_axial_(k, cos, sin, k.shape[1] - n_exclude, nq)
        pm = ops.attention_dv(q.view(B, nq, 1, c), k.view(B, -1, 1, c), mem.view(B, -1, 1, mem.shape[-1]), c ** -0.5)
        o = self.lin(name + ".v_proj", pm.view(B, nq, mem.shape[-1]))
        return self.lin(name + ".out_proj", o, residual=residual)

    def memory_attention(self, curr, memory, memory_pos, num_obj_ptr_tokens):
        """MemoryAttention.forward — R/modeling/memory_attention.py:119-169,60-99 (batch-first throughout).
        curr [N,HW,256]; memory, memory_pos [N,M,64] -> [N,HW,256]."""
        m = "memory_attention."
        out = ops.axpby(curr, self.vision_pos(), 1.0, 0.1)
        mem_k = ops.add(memory, memory_pos)
        for i in range(4):
            l = f"{m}layers.{i}."
            t2 = self.ln(l + "norm1", out)
            out = self._rope_attn(l + "self_attn", t2, t2, t2, 0, out)
            t2 = self.ln(l + "norm2", out)
            cross = self._rope_attn_lowrank if _MEMATTN_LOWRANK and memory.shape[-1] == 64 else self._rope_attn
            out = cross(l + "cross_attn_image", t2, mem_k, memory, num_obj_ptr_tokens, out)
            t2 = self.ln(l + "norm3", out)
            out = self.lin(l + "linear2", self.lin(l + "linear1", t2, act=ops.ACT_RELU), residual=out)
        return self.ln(m + "norm", out)

    def _merged_vo(self, name):
        """(W_out W_v [256, 64], W_out b_v + b_out): the value and output projections of the low-rank memory cross-attention as ONE matrix —
        ((P M) Wv^T + bv) Wo^T + bo = (P M) (Wo Wv)^T + (Wo bv + bo), exact in real arithmetic (product formed in fp32, rounded once)."""
        def make():
            wv, wo = self.P.sd[self.p + name + ".v_proj.weight"].float(), self.P.sd[self.p + name + ".out_proj.weight"].float()
            return wo @ wv
        def make_b():
            wo = self.P.sd[self.p + name + ".out_proj.weight"].float()
            return wo @ self.P.sd[self.p + name + ".v_proj.bias"].float() + self.P.sd[self.p + name + ".out_proj.bias"].float()
        return self.P.const(("merged_vo", name), make), self.P.const(("merged_vo_b", name), make_b, torch.float32)

    def memory_attention_bank(self, curr, bank, lo, hi, pos, nptr):
        """MemoryAttention.forward (R/modeling/memory_attention.py:119-169,60-99) on the clip's memory bank (r05).
        curr [N,HW,256]; bank [N,R,64]: rows [lo, hi) are this frame's memory — first nptr object-pointer tokens (no RoPE, no positional term:
        num_k_exclude_rope), then whole 4096-token memories; pos [R,64]: the positional table aligned with the bank's rows (maskmem_pos +
        maskmem_tpos_enc by age, zeros on the pointer rows).  Per layer: norm -> q|k|v -> RoPE as one launch, attention, out_proj + residual,
        norm -> q -> RoPE as one launch, cross-attention on the un-projected memory (vg_attention_dv), W_out W_v + residual as one GEMM,
        norm -> linear1 -> ReLU as one launch, linear2 + residual; the four layers' key projections of (memory + pos) + their RoPE are ONE
        launch per frame.  bf16: vg_gemm_rows; the fp32 parity mode runs the same statements on the separate kernels (ops.linear_rows)."""
        m = "memory_attention."
        N, nq, C = curr.shape
        Mv = hi - lo
        side = int(math.sqrt(nq))
        cos = self.P.const(("axial_cos", C, nq), lambda: _axial_cos_sin(C, side)[0], torch.float32)
        sin = self.P.const(("axial_sin", C, nq), lambda: _axial_cos_sin(C, side)[1], torch.float32)
        lnp = lambda n: (self.P.f32(self.p + n + ".weight"), self.P.f32(self.p + n + ".bias"), 1e-5)      # noqa: E731
        out = ops.axpby(curr, self.vision_pos(), 1.0, 0.1)
        mem = bank[:, lo:hi]
        wk4, bk4 = self.P.fused([f"{self.p}{m}layers.{i}.cross_attn_image.k_proj" for i in range(4)])
        k4 = ops.linear_rows(mem, wk4, bk4, add=pos[lo:hi], rope=(cos, sin, 4 * C, C, Mv, nptr, Mv, nq))          # [N, Mv, 4 C]
        for i in range(4):
            l = f"{m}layers.{i}."
            sa = self.p + l + "self_attn"
            wqkv, bqkv = self.P.fused([sa + ".q_proj", sa + ".k_proj", sa + ".v_proj"])
            qkv = ops.linear_rows(out, wqkv, bqkv, ln=lnp(l + "norm1"), rope=(cos, sin, 2 * C, C, nq, 0, nq, nq))
            o = ops.attention(qkv[..., :C].unsqueeze(2), qkv[..., C:2 * C].unsqueeze(2), qkv[..., 2 * C:].unsqueeze(2), C ** -0.5)
            out = self.lin(l + "self_attn.out_proj", o.view(N, nq, C), residual=out)
            ca = l + "cross_attn_image"
            q = ops.linear_rows(out, self.P.w(self.p + ca + ".q_proj"), self.P.b(self.p + ca + ".q_proj"), ln=lnp(l + "norm2"),
                                rope=(cos, sin, C, C, nq, 0, nq, nq))
            pm = ops.attention_dv(q.view(N, nq, 1, C), k4[..., i * C:(i + 1) * C].unsqueeze(2), mem.unsqueeze(2), C ** -0.5)
            wvo, bvo = self._merged_vo(ca)
            out = ops.linear(pm.view(N, nq, mem.shape[-1]), wvo, bvo, residual=out)
            h = ops.linear_rows(out, self.P.w(self.p + l + "linear1"), self.P.b(self.p + l + "linear1"), act=ops.ACT_RELU, ln=lnp(l + "norm3"))
            out = self.lin(l + "linear2", h, residual=out)
        return self.ln(m + "norm", out)

    # ------------------------------------------------------------------ S9 memory encoder
    def memory_encoder(self, pix_feat, mask, out=None):
        """MemoryEncoder.forward(skip_mask_sigmoid=True) — R/modeling/memory_encoder.py:159-182,17-118.
        pix_feat [N,es,es,256]; mask [N,S,S,1] (already scaled) -> features [N,es*es,64] (written into `out`, a [N,es*es,64] view whose
        object stride is free — the memory bank's slot — when given)."""
        e = "memory_encoder."
        N = pix_feat.shape[0]
        x, H = mask, self.S
        for i in range(4):
            w = self.P.conv_w(f"{self.p}{e}mask_downsampler.encoder.{3 * i}")
            if _MEMENC_FUSED:      # the few-channel stages (1 -> 4 -> 16): conv + LayerNorm2d + GELU in one pass (r04)
                ln = f"{self.p}{e}mask_downsampler.encoder.{3 * i + 1}"
                y = ops.conv3s2_ln_gelu(x, w, self.P.b(f"{self.p}{e}mask_downsampler.encoder.{3 * i}"), self.P.f32(ln + ".weight"), self.P.f32(ln + ".bias"), 1e-6)
                if y is not None:
                    x = y
                    continue
            cols, Ho, Wo = ops.im2col(x, 3, 3, 2, 1, w.shape[1])
            x = ops.linear(cols, w, self.P.b(f"{self.p}{e}mask_downsampler.encoder.{3 * i}"))
            x = ops.activation(self.ln(f"{e}mask_downsampler.encoder.{3 * i + 1}", x, 1e-6), ops.ACT_GELU).view(N, Ho, Wo, -1)
        x = self.lin(e + "mask_downsampler.encoder.12", x)
        x = self.lin(e + "pix_feat_proj", pix_feat, residual=x)
        for i in range(2):
            l = f"{e}fuser.layers.{i}."
            h = ops.dwconv(x, self.P.dw_w(self.p + l + "dwconv"), self.P.b(self.p + l + "dwconv"), 7)
            if _MEMBANK:       # norm -> pwconv1 -> GELU in one launch (bf16; the parity mode runs the separate kernels inside linear_rows)
                h = ops.linear_rows(h, self.P.w(self.p + l + "pwconv1"), self.P.b(self.p + l + "pwconv1"), act=ops.ACT_GELU,
                                    ln=(self.P.f32(self.p + l + "norm.weight"), self.P.f32(self.p + l + "norm.bias"), 1e-6))
            else:
                h = self.lin(l + "pwconv1", self.ln(l + "norm", h, 1e-6), act=ops.ACT_GELU)
            x = ops.linear(h, self.P.w(self.p + l + "pwconv2"), self.P.b(self.p + l + "pwconv2"), gamma=self.P.f32(self.p + l + "weight"), residual=x)
        return self.lin(e + "out_proj", x.view(N, self.es * self.es, -1), out=out).view(N, self.es * self.es, 64)

    def maskmem_pos(self):
        return self.P.const(("maskmem_pos", self.es), lambda: _sine_pos(64, self.es, self.es))

    # ------------------------------------------------------------------ S6 SAM heads
    def forward_sam_heads(self, pix_feat, high_res, text_inputs, multimask_output=True, ptr_out=None):
        """SAM2Base._forward_sam_heads (points=None, masks=None) — R/modeling/sam2_base.py:251-411.
        pix_feat [N,HW,256]; returns low-res best mask fp32 [N,1,4es,4es] (NO_OBJ-filled), high-res [N,1,S,S],
        obj_ptr [N,256], object score logits [N,1].  ptr_out: optional [N, 4, 64] destination of the object pointers (their rows of the memory bank)."""
        N = pix_feat.shape[0]
        sparse = self.sparse_prompt(N, text_inputs, with_empty_point=True)
        masks, iou, toks, obj = self.mask_decoder(pix_feat, sparse, high_res, repeat_image=False)
        low, _, tok, _ = ops.multimask_select(masks, iou, toks, 1 if multimask_output else 0)
        low = ops.where_rows(obj, low, None, NO_OBJ_SCORE)
        high = ops.bilinear(low.view(N, 4 * self.es, 4 * self.es), self.S, self.S).view(N, 1, self.S, self.S)
        if self._heads_grouped():
            ptr = ops.mlp3_grouped(tok.view(N, 1, -1), 1, *self._mlp3_stack(["obj_ptr_proj"]), out=torch.empty(N, 1, tok.shape[-1], dtype=tok.dtype, device=tok.device)).view(N, -1)
        else:
            ptr = self.mlp("obj_ptr_proj", tok, 3)
        ptr = ops.where_rows(obj, ptr, self.P.t(self.p + "no_obj_ptr").view(-1), out=ptr_out)
        return dict(low=low, high=high, obj_ptr=ptr, obj_logits=obj, low_multi_pre_where=masks[:, 1:], ious=iou[:, 1:])

    def encode_new_memory(self, feat_top, high_res_masks, is_mask_from_pts, out=None):
        """SAM2Base._encode_new_memory — R/modeling/sam2_base.py:666-704; features come back bf16-rounded the way
        the predictor stores them (sam2_video_predictor.py:967,1011).  out: optional [N, es*es, 64] destination (a slot of the memory bank)."""
        N = high_res_masks.shape[0]
        m = ops.mask_for_mem(high_res_masks.view(N, self.S, self.S, 1), is_mask_from_pts, 20.0, -10.0, self.dtype)
        if self.dtype == torch.bfloat16:        # the model's activations ARE the predictor's storage format: the last GEMM writes the slot
            return self.memory_encoder(feat_top.view(N, self.es, self.es, 256), m, out=out)
        mem = ops.cast(ops.cast(self.memory_encoder(feat_top.view(N, self.es, self.es, 256), m), torch.bfloat16), self.dtype)
        if out is not None:
            out.copy_(mem)
            return out
        return mem

    # ------------------------------------------------------------------ S10 video branch / S11 framewise
    def video_branch(self, images, text_embeds, video_hw, trace=None, frame_feats=None, as_masks=False, mem_override=None):
        """init_state_from_tensor -> add_new_text per object -> propagate_in_video for one clip
        (R/model/VideoGLaMM.py:834-877; R/sam2_video_predictor.py:108-180,415-495,520-636,674-827,921-1017;
        R/modeling/sam2_base.py:495-664,706-803).  images [T,3,S,S]; text_embeds [N,256].
        frame_feats: optional precomputed forward_image outputs per frame (frame-sharded Hiera).
        mem_override (parity tests): {frame: [N, es*es, 64] memory} stored in the bank INSTEAD of the frame's own encoded memory (which is still
        computed and traced) — the bank is bf16-rounded like the reference's, so two fp32 implementations that differ in summation order store
        memories one rounding step apart in a few elements; teacher-forcing the bank separates that from an arithmetic difference.
        -> video-res logits fp32 [T,N,H,W]."""
        T, N = images.shape[0], text_embeds.shape[0]
        es, hw = self.es, self.es * self.es
        H, W = video_hw

        if frame_feats is None:  # Hiera is frame-independent: batch it up front (the recurrence below only reads it)
            frame_feats = self.hiera_frames(images)
        if _MEMBANK:
            return self._video_branch_bank(T, text_embeds, video_hw, trace, frame_feats, as_masks, mem_override)

        def feats(t, bs):
            fpn = frame_feats[t]
            if bs > 1:
                fpn = [f.expand(bs, -1, -1, -1).contiguous() for f in fpn]
            return fpn

        no_mem = self.P.t(self.p + "no_mem_embed").view(-1)
        # frame 0: objects added one at a time (batch 1 each), no memory encoder yet
        fpn0 = feats(0, 1)
        pix0 = ops.add(fpn0[2].view(1, hw, 256), no_mem)          # directly_add_no_mem_embed
        outs0 = [self.forward_sam_heads(pix0, (fpn0[0], fpn0[1]), text_embeds[k:k + 1].unsqueeze(1)) for k in range(N)]
        low0 = torch.cat([o["low"] for o in outs0])
        ptr0 = torch.cat([o["obj_ptr"] for o in outs0])
        if trace is not None:
            trace["frame0_low_multi_pre_where"] = torch.cat([o["low_multi_pre_where"] for o in outs0])
            trace["frame0_obj_logits"] = torch.cat([o["obj_logits"] for o in outs0])
            trace["obj_ptr"], trace["maskmem"] = [ptr0], []
        # preflight: consolidate + memory-encode frame 0 (binarised mask, is_mask_from_pts=True)
        high0 = ops.bilinear(low0.view(N, 4 * es, 4 * es), self.S, self.S).view(N, 1, self.S, self.S)
        top0 = fpn0[2].view(1, hw, 256).expand(N, -1, -1).contiguous()
        cond = dict(mem=self.encode_new_memory(top0, high0, True), ptr=ptr0)
        if trace is not None:
            trace["maskmem"].append(cond["mem"])
        if mem_override is not None and 0 in mem_override:
            cond["mem"] = mem_override[0].to(cond["mem"])
        non_cond = {}
        lows = [low0]
        tpos = self.P.t(self.p + "maskmem_tpos_enc").view(7, 64)
        mpos = self.maskmem_pos()
        for t in range(1, T):
            fpn = feats(t, N)
            mems, mposs = [], []
            for t_pos in range(0, 7):
                prev = cond if t_pos == 0 else non_cond.get(t - (7 - t_pos))
                if prev is None:
                    continue
                mems.append(prev["mem"])
                mposs.append(ops.add(mpos, tpos[7 - t_pos - 1]).unsqueeze(0).expand(N, -1, -1))
            ptrs = [cond["ptr"]]
            for t_diff in range(1, min(T, 16)):
                if t - t_diff < 0:
                    break
                if (t - t_diff) in non_cond:
                    ptrs.append(non_cond[t - t_diff]["ptr"])
            obj_ptrs = torch.stack(ptrs, dim=1).reshape(N, len(ptrs) * 4, 64)   # each 256-d pointer = 4 tokens of 64
            memory = torch.cat(mems + [obj_ptrs], dim=1).contiguous()
            memory_pos = torch.cat(mposs + [torch.zeros_like(obj_ptrs)], dim=1).contiguous()
            top = fpn[2].view(N, hw, 256)
            pix = self.memory_attention(top, memory, memory_pos, obj_ptrs.shape[1])
            o = self.forward_sam_heads(pix, (fpn[0], fpn[1]), None)
            non_cond[t] = dict(mem=self.encode_new_memory(top, o["high"], False), ptr=o["obj_ptr"])
            non_cond.pop(t - 16, None)
            lows.append(o["low"])
            if trace is not None:
                trace["obj_ptr"].append(o["obj_ptr"])
                trace["maskmem"].append(non_cond[t]["mem"])
                trace[f"obj_logits_{t}"] = o["obj_logits"]
            if mem_override is not None and t in mem_override:
                non_cond[t]["mem"] = mem_override[t].to(non_cond[t]["mem"])
            if trace is not None and t == 1:
                trace["frame1_pix_feat_with_mem"] = pix
                trace["frame1_low_multi_pre_where"] = o["low_multi_pre_where"]
        low = torch.stack(lows)                                                  # [T,N,1,4es,4es]
        if trace is not None:
            trace["low_res"] = low
            trace["obj_ptr"] = torch.stack(trace["obj_ptr"])       # [T,N,256]; trace["maskmem"][t]: [N, es*es, 64] (bf16-rounded)
        up = ops.bilinear_mask if as_masks else ops.bilinear       # as_masks=True: uint8 (logit > 0) in one pass
        return up(low.view(T * N, 4 * es, 4 * es), H, W).view(T, N, H, W)

    def _bank_pos(self, ages):
        """[PTR_ROWS + 7 hw, 64] positional table aligned with the memory bank's rows for one age pattern: slot s holds maskmem_pos +
        maskmem_tpos_enc[ages[s]] (sam2_base.py:566-580: t_pos = 0 for the conditioning frame -> entry 6, a frame t_rel behind -> entry
        t_rel - 1), zeros on the pointer rows and on empty slots.  At most 12 patterns per clip length; built on first use (never inside a capture:
        the eager pass in front of it fills the cache)."""
        hw = self.es * self.es

        def make():
            tpos = self.P.t(self.p + "maskmem_tpos_enc").view(7, 64)
            mpos = self.maskmem_pos()
            tab = torch.zeros(PTR_ROWS + 7 * hw, 64, dtype=self.dtype, device=self.device)
            for sl, a in enumerate(ages):
                if a is not None:
                    tab[PTR_ROWS + sl * hw:PTR_ROWS + (sl + 1) * hw] = ops.add(mpos, tpos[a])
            return tab
        return self.P.const(("bank_pos", self.es) + tuple(-1 if a is None else a for a in ages), make)

    def _video_branch_bank(self, T, text_embeds, video_hw, trace, frame_feats, as_masks, mem_override):
        """video_branch() on a preallocated memory bank (r05).  bank [N, PTR_ROWS + 7 hw, 64]: rows [PTR_ROWS - 4 (j + 1), PTR_ROWS - 4 j) hold
        object pointer j (j = 0: the conditioning frame's, j = 1 + (t - 1) % 15: frame t's — a ring over the 15 most recent frames), rows
        [PTR_ROWS + s hw, PTR_ROWS + (s + 1) hw) memory slot s (s = 0: the conditioning frame, s = 1 + (t - 1) % 6: frame t — a ring over the 6
        most recent).  The valid rows of a frame are one contiguous range [lo, hi): the memory encoder's last GEMM and the pointer selection
        write their rows in place, the attention kernels read the range through strides — no concatenation, no copies.  The keys' order inside
        the softmax differs from the reference's (memories oldest-first, then pointers newest-first), which is a summation order, not a value."""
        N = text_embeds.shape[0]
        es, hw = self.es, self.es * self.es
        H, W = video_hw
        nptr_max = min(T, 16)                              # max_obj_ptrs_in_encoder = min(num_frames, 16)
        bank = torch.empty(N, PTR_ROWS + 7 * hw, 64, dtype=self.dtype, device=self.device)
        ptr_rows = lambda j: slice(PTR_ROWS - 4 * (j + 1), PTR_ROWS - 4 * j)        # noqa: E731
        mem_rows = lambda sl: slice(PTR_ROWS + sl * hw, PTR_ROWS + (sl + 1) * hw)   # noqa: E731

        def feats(t, bs):
            fpn = frame_feats[t]
            if bs > 1:
                fpn = [f.expand(bs, -1, -1, -1).contiguous() for f in fpn]
            return fpn

        def store_memory(t, sl, top, high, from_pts):
            mem = self.encode_new_memory(top, high, from_pts, out=bank[:, mem_rows(sl)])
            if trace is not None:
                trace["maskmem"].append(mem.clone())
            if mem_override is not None and t in mem_override:
                bank[:, mem_rows(sl)].copy_(mem_override[t])

        no_mem = self.P.t(self.p + "no_mem_embed").view(-1)
        # frame 0: objects added one at a time (batch 1 each), no memory encoder yet
        fpn0 = feats(0, 1)
        pix0 = ops.add(fpn0[2].view(1, hw, 256), no_mem)          # directly_add_no_mem_embed
        outs0 = [self.forward_sam_heads(pix0, (fpn0[0], fpn0[1]), text_embeds[k:k + 1].unsqueeze(1), ptr_out=bank[k:k + 1, ptr_rows(0)])
                 for k in range(N)]
        low0 = torch.cat([o["low"] for o in outs0])
        if trace is not None:
            trace["frame0_low_multi_pre_where"] = torch.cat([o["low_multi_pre_where"] for o in outs0])
            trace["frame0_obj_logits"] = torch.cat([o["obj_logits"] for o in outs0])
            trace["obj_ptr"], trace["maskmem"] = [bank[:, ptr_rows(0)].reshape(N, 256).clone()], []
        # preflight: consolidate + memory-encode frame 0 (binarised mask, is_mask_from_pts=True)
        high0 = ops.bilinear(low0.view(N, 4 * es, 4 * es), self.S, self.S).view(N, 1, self.S, self.S)
        top0 = fpn0[2].view(1, hw, 256).expand(N, -1, -1).contiguous() if N > 1 else fpn0[2].view(1, hw, 256)
        store_memory(0, 0, top0, high0, True)
        lows = [low0]
        for t in range(1, T):
            fpn = feats(t, N)
            ages = [6] + [None] * 6                                   # slot -> entry of maskmem_tpos_enc
            for tp in range(max(1, t - 6), t):
                ages[1 + (tp - 1) % 6] = t - tp - 1
            nmem = 1 + min(t - 1, 6)
            nptr = 4 * (1 + min(t - 1, nptr_max - 1))
            lo, hi = PTR_ROWS - nptr, PTR_ROWS + nmem * hw
            top = fpn[2].view(N, hw, 256)
            pix = self.memory_attention_bank(top, bank, lo, hi, self._bank_pos(ages), nptr)
            # this frame's pointer replaces the oldest of the ring AFTER the attention has read it (stream order)
            o = self.forward_sam_heads(pix, (fpn[0], fpn[1]), None, ptr_out=bank[:, ptr_rows(1 + (t - 1) % 15)])
            lows.append(o["low"])
            if trace is not None:
                trace["obj_ptr"].append(bank[:, ptr_rows(1 + (t - 1) % 15)].reshape(N, 256).clone())
                trace[f"obj_logits_{t}"] = o["obj_logits"]
                if t == 1:
                    trace["frame1_pix_feat_with_mem"] = pix
                    trace["frame1_low_multi_pre_where"] = o["low_multi_pre_where"]
            store_memory(t, 1 + (t - 1) % 6, top, o["high"], False)
        low = torch.stack(lows)                                                  # [T,N,1,4es,4es]
        if trace is not None:
            trace["low_res"] = low
            trace["obj_ptr"] = torch.stack(trace["obj_ptr"])       # [T,N,256]; trace["maskmem"][t]: [N, es*es, 64] (bf16-rounded)
        up = ops.bilinear_mask if as_masks else ops.bilinear       # as_masks=True: uint8 (logit > 0) in one pass
        return up(low.view(T * N, 4 * es, 4 * es), H, W).view(T, N, H, W)

    def video_branch_graphed(self, images, text_embeds, video_hw, frame_feats, as_masks=False):
        """video_branch() replayed from a HIP graph.  The propagation is ~150 small launches per frame with no host decision
        in between (mask selection, object scores and memory selection all stay on the device), so the launch sequence of a
        (T, N, output size) configuration is captured once — after one eager pass that packs the weights — and replayed for
        every later clip: the GPU no longer waits for Python between launches (r02, C2 clip: 16 % of the step idle in the eager
        loop; 378 -> 361 ms).  Inputs are copied into the graph's static buffers; the result is a copy of the graph's output
        buffer.  At most four (T, N, size) configurations are kept (least recently used goes first)."""
        T, N = images.shape[0], text_embeds.shape[0]
        key = (T, N, tuple(video_hw), text_embeds.dtype, bool(as_masks))
        graphs = self.__dict__.setdefault("_video_graphs", {})
        ent = graphs.pop(key, None)
        if ent is None:
            while len(graphs) >= 4:
                graphs.pop(next(iter(graphs)))
            # the persistent feature buffers of clip lengths no cached graph reads any more go with them (8.4 MB per frame at SAM2-L: a long-lived
            # process fed varied clip lengths would otherwise grow without bound)
            st_all = self.__dict__.get("_video_static", {})
            for n in [n for n in st_all if n != T and all(k[0] != n for k in graphs)]:
                del st_all[n]
            self.video_branch(images, text_embeds, video_hw, frame_feats=frame_feats, as_masks=as_masks)      # eager once: lazy weight packing, kernel attributes
            # static inputs of the graph: the persistent feature buffers themselves when the caller's features already live there
            # (model.inference_video_branch hands them to Hiera), private copies otherwise
            static = self.__dict__.get("_video_static", {}).get(T)
            in_place = static is not None and all(frame_feats[t][lv].data_ptr() == static[lv][t:t + 1].data_ptr() for t in range(T) for lv in range(3))
            st_feats = {t: ([static[lv][t:t + 1] for lv in range(3)] if in_place else [torch.empty_like(f) for f in frame_feats[t]]) for t in range(T)}
            st_emb = torch.empty_like(text_embeds)
            for t in range(T):
                for d, f in zip(st_feats[t], frame_feats[t]):
                    if d.data_ptr() != f.data_ptr():
                        d.copy_(f)
            st_emb.copy_(text_embeds)
            torch.cuda.synchronize()
            try:
                g, kept = torch.cuda.CUDAGraph(keep_graph=True), True      # (the hipGraph_t stays readable: video_graph_nodes() below)
            except TypeError:                                  # torch < 2.8: no keep_graph — the node census is then unavailable, the replay is the same
                g, kept = torch.cuda.CUDAGraph(), False
            with ops.graph_capture(g):      # (thread-local capture mode, cyclic GC held off: ops.graph_capture)
                out = self.video_branch(images, st_emb, video_hw, frame_feats=st_feats, as_masks=as_masks)
            if kept:
                g.instantiate()
            ent = (g, st_feats, st_emb, out)
        graphs[key] = ent                       # (re-inserted last: most recently used)
        g, st_feats, st_emb, out = ent
        for t in range(T):
            for d, f in zip(st_feats[t], frame_feats[t]):
                if d.data_ptr() != f.data_ptr():       # (features Hiera wrote into the graph's own buffers need no staging)
                    d.copy_(f)
        st_emb.copy_(text_embeds)
        g.replay()
        return out.clone()

    def video_graph_nodes(self):
        """{(T, N, (H, W), as_masks): (kernel, memcpy, other) node counts} of the cached propagation graphs: launches per replayed clip."""
        return {(k[0], k[1], k[2], k[4]): ops.graph_node_counts(ent[0]) for k, ent in self.__dict__.get("_video_graphs", {}).items()
                if hasattr(ent[0], "raw_cuda_graph")}

    def framewise_branch(self, images, text_embeds, video_hw, frame_feats=None, frames=None, as_masks=False):
        """VideoGLaMM framewise decode — R/model/VideoGLaMM.py:205-241,676-766.  One mask-decoder batch per frame
        (N objects, repeat_image), multimask_output=False with the stability fallback.  -> (logits fp32 [T,N,H,W], low-res logits);
        as_masks=True: the first value is the thresholded uint8 masks (logit > 0) instead, made in one pass from the low-res
        logits (vg_bilinear_mask) — the fp32 logits at output resolution (134 MB for 32 x 1024^2) never exist."""
        N = text_embeds.shape[0]
        es, hw = self.es, self.es * self.es
        H, W = video_hw
        frames = list(range(images.shape[0])) if frames is None else frames
        sparse = self.sparse_prompt(N, text_embeds.unsqueeze(1), with_empty_point=False)
        ns = sparse.shape[1]
        no_mem = self.P.t(self.p + "no_mem_embed").view(-1)
        lows = []
        # frames are independent here: a chunk of frames x all objects goes through Hiera and the mask decoder as
        # ONE batch (identical per-item arithmetic to the reference's frame-serial loop, far fewer/larger launches)
        step = max(1, self.decode_chunk // max(N, 1)) if frame_feats is not None else self.frame_chunk
        for c0 in range(0, len(frames), step):
            fr = frames[c0:c0 + step]
            Tc = len(fr)
            if frame_feats is not None:
                fpn = [_stack_views([frame_feats[t][lv] for t in fr]) for lv in range(3)]
            else:
                fpn = self.forward_image(images[fr[0]:fr[-1] + 1] if fr == list(range(fr[0], fr[-1] + 1)) else images[fr])
            emb = ops.add(fpn[2].view(Tc, hw, 256), no_mem)
            # (object, frame) pairs, object-major: instance o * Tc + t decodes frame t with object o's prompt.  r04: the frame's embedding and
            # high-resolution features are SHARED by its objects (mask_decoder: instance i -> image i % Tc) — the per-object copies
            # (3.1 GB per 64 x 8 clip, written and read back) are gone.
            sp = sparse if Tc == 1 else sparse.unsqueeze(1).expand(N, Tc, ns, 256).reshape(N * Tc, ns, 256)
            masks, iou, toks, _ = self.mask_decoder(emb, sp, (fpn[0].view(Tc, 16 * hw, 32), fpn[1].view(Tc, 4 * hw, 64)), repeat_image=Tc == 1)
            low, _, _, _ = ops.multimask_select(masks, iou, toks, 0)
            low = low.view(N, Tc, 1, 4 * es, 4 * es)
            lows.append(low.transpose(0, 1) if N > 1 and Tc > 1 else low.view(Tc, N, 1, 4 * es, 4 * es))
        low = torch.cat(lows, dim=0)                                             # [T,N,1,4es,4es]
        up = ops.bilinear_mask if as_masks else ops.bilinear
        return up(low.view(len(frames) * N, 4 * es, 4 * es), H, W).view(len(frames), N, H, W), low
