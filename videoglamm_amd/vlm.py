"""LLM side of the VideoGLaMM hot path on the MI355X kernel library (rows L1–L6 of SURVEY.md §8a):
InternVideo2 video tower, CLIP ViT image tower, V-L adapters + 2x2 pooling, embedding splice, Llama
decoder with a KV cache, greedy decode with [SEG] hidden capture, L-V adapter.

Scheduling differs from the reference on purpose (results are identical): the towers run ONCE per clip and
the LLM keeps a KV cache, where the reference re-encodes the clip and re-runs the full sequence for every
generated token (generate(use_cache=False), R/model/VideoGLaMM.py:617-626,790-799;
R/model/videogpt_plus/model/language_model/llama3_1.py:110-134).
"""
import os

import torch

from . import ops

IMAGE_TOKEN_INDEX = -200  # R/model/videogpt_plus/constants.py


class VisionTowers:
    def __init__(self, params, cfg):
        self.P, self.cfg = params, cfg
        self.dtype = params.dtype

    # ------------------------------------------------------------------ shared
    def _patches(self, imgs, ps, wname, bias):
        """non-overlapping patch embedding as im2col + GEMM. imgs [B,3,H,W] NCHW -> [B, L, C]."""
        B, _, H, W = imgs.shape
        x = ops.permute5(imgs.contiguous(), (B, H, W, 3, 1), (3 * H * W, W, 1, H * W, 0)).view(B, H, W, 3)
        x = ops.cast(x, self.dtype)
        w = self.P.conv_w(wname)
        cols, Ho, Wo = ops.im2col(x, ps, ps, ps, 0, w.shape[1])
        return ops.linear(cols, w, bias).view(B, Ho * Wo, -1)

    # ------------------------------------------------------------------ L1 InternVideo2
    def iv2(self, video):
        """InternVideo2_Stage2V.forward / PretrainInternVideo2.forward(x_vis_return_idx=-2, x_vis_only=True)
        R/.../internvideo/utils.py:229-238; internvideo2.py:585-651,190-209,265-316.
        video [nc,4,3,H,W] -> [nc, 4*L, C] (CLS already dropped, arch.py:145)."""
        p, c = "model.vision_tower.vision_encoder.", self.cfg["iv2"]
        nc, T = video.shape[:2]
        heads = c["num_heads"]
        x = self._patches(video.reshape(nc * T, *video.shape[2:]), c["patch_size"], p + "patch_embed.proj", self.P.b(p + "patch_embed.proj"))
        C = x.shape[-1]
        x = x.view(nc, -1, C)
        cls = self.P.t(p + "cls_token").view(1, 1, C).expand(nc, 1, C)
        x = ops.add(torch.cat([cls, x], dim=1).contiguous(), self.P.t(p + "pos_embed").view(-1))
        n, hd = x.shape[1], C // heads
        for i in range(c["depth"] - 1):  # the loop breaks after block depth-2 (internvideo2.py:640-642)
            b = f"{p}blocks.{i}."
            h = ops.rmsnorm(x, self.P.f32(b + "norm1.weight"), 1e-6)
            qkv = ops.linear(h, self.P.w(b + "attn.qkv")).view(nc * n, 3 * C)
            q = ops.rmsnorm(qkv[:, :C], self.P.f32(b + "attn.q_norm.weight"), 1e-6).view(nc, n, heads, hd)
            k = ops.rmsnorm(qkv[:, C:2 * C], self.P.f32(b + "attn.k_norm.weight"), 1e-6).view(nc, n, heads, hd)
            v = qkv.view(nc, n, 3, heads, hd)[:, :, 2]
            o = ops.attention(q, k, v, hd ** -0.5).view(nc, n, C)
            x = ops.linear(o, self.P.w(b + "attn.proj"), self.P.b(b + "attn.proj"), gamma=self.P.f32(b + "ls1.gamma"), residual=x)
            h = ops.rmsnorm(x, self.P.f32(b + "norm2.weight"), 1e-6)
            h = ops.linear(h, self.P.w(b + "mlp.fc1"), self.P.b(b + "mlp.fc1"), act=ops.ACT_GELU)
            x = ops.linear(h, self.P.w(b + "mlp.fc2"), self.P.b(b + "mlp.fc2"), gamma=self.P.f32(b + "ls2.gamma"), residual=x)
        return x[:, 1:]

    # ------------------------------------------------------------------ L2 CLIP
    def clip(self, images):
        """CLIPVisionTower.forward('patch'): HF CLIPVisionModel hidden_states[-2] minus CLS —
        R/.../multimodal_encoder/clip_encoder.py:34-72.  images [T,3,H,W] -> [T, L, C]."""
        c = self.cfg["clip"]
        p = "model.image_vision_tower.vision_tower."
        v = p + "vision_model." if self.P.has(p + "vision_model.embeddings.class_embedding") else p
        heads = c["num_heads"]
        x = self._patches(images, c["patch_size"], v + "embeddings.patch_embedding", None)
        T, L, C = x.shape
        cls = self.P.t(v + "embeddings.class_embedding").view(1, 1, C).expand(T, 1, C)
        x = ops.add(torch.cat([cls, x], dim=1).contiguous(), self.P.t(v + "embeddings.position_embedding.weight").view(-1))
        x = ops.layernorm(x, self.P.f32(v + "pre_layrnorm.weight"), self.P.f32(v + "pre_layrnorm.bias"), 1e-5)
        n, hd = L + 1, C // heads
        for i in range(c["num_layers"] - 1):  # select_layer=-2: the last layer's output is never read
            b = f"{v}encoder.layers.{i}."
            h = ops.layernorm(x, self.P.f32(b + "layer_norm1.weight"), self.P.f32(b + "layer_norm1.bias"), 1e-5)
            wqkv, bqkv = self.P.fused([b + "self_attn.q_proj", b + "self_attn.k_proj", b + "self_attn.v_proj"])
            qkv = ops.linear(h, wqkv, bqkv).view(T, n, 3, heads, hd)
            o = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], hd ** -0.5).view(T, n, C)
            x = ops.linear(o, self.P.w(b + "self_attn.out_proj"), self.P.b(b + "self_attn.out_proj"), residual=x)
            h = ops.layernorm(x, self.P.f32(b + "layer_norm2.weight"), self.P.f32(b + "layer_norm2.bias"), 1e-5)
            h = ops.linear(h, self.P.w(b + "mlp.fc1"), self.P.b(b + "mlp.fc1"), act=ops.ACT_QUICK_GELU)
            x = ops.linear(h, self.P.w(b + "mlp.fc2"), self.P.b(b + "mlp.fc2"), residual=x)
        return x[:, 1:]

    # ------------------------------------------------------------------ L3 adapters + pooling
    def _projector(self, name, x):
        """'linear' or 'mlpNx_gelu' — R/.../multimodal_projector/builder.py:17-54."""
        if self.P.has(name + ".weight"):
            return ops.linear(x, self.P.w(name), self.P.b(name))
        i = 0
        while self.P.has(f"{name}.{i + 2}.weight"):
            x = ops.linear(x, self.P.w(f"{name}.{i}"), self.P.b(f"{name}.{i}"), act=ops.ACT_GELU)
            i += 2
        return ops.linear(x, self.P.w(f"{name}.{i}"), self.P.b(f"{name}.{i}"))

    def encode(self, images, context_images, comm=None):
        """encode_videos + project(input_type='video') — R/model/videogpt_plus/model/arch.py:121-151,164-191.
        images [Te,3,224,224], context [Te,3,336,336] -> visual tokens [Te*144 + Te*64, D] (context first);
        context_images None: images [t,3,336,336] -> [t*576, D].
        comm (dist.FrameSharder, world > 1): this rank encodes only its block of CLIP frames and of 4-frame InternVideo2
        chunks; the projected, pooled tokens are all-gathered (every op up to there is per frame / per chunk)."""
        if context_images is None:
            # image prompt (encode_images + project(input_type="image"), arch.py:110-119,393-397): CLIP patch features of
            # the image(s) -> image_mm_projector, no pooling, all tokens concatenated
            cf = self._projector("model.image_mm_projector", self.clip(images).contiguous())
            return cf.view(-1, cf.shape[-1])
        te = images.shape[0]
        assert te % 4 == 0, "the video encoder consumes 4-frame chunks (arch.py:133)"
        video = images.view(te // 4, 4, *images.shape[1:])
        # opt-in (VG_TOWERS_SHARDED=1): a rank that encodes fewer frames runs GEMMs of another M, which can take another tile route
        # (another fp32 summation order): in bf16 the visual tokens — and with them greedy ids — may then differ from the
        # single-GPU run.  The default keeps the whole LLM side replicated: N-GPU ids == 1-GPU ids by construction.
        sharded = comm is not None and comm.world > 1 and os.environ.get("VG_TOWERS_SHARDED", "0") == "1"
        if sharded:
            c0, cn = comm.block(te)
            v0, vn = comm.block(te // 4)
            context_images, video = context_images[c0:c0 + cn], video[v0:v0 + vn]
        else:
            cn, vn = te, te // 4

        def video_tokens():
            if vn == 0:
                return None
            vf = self._projector("model.mm_projector", self.iv2(video).contiguous())      # [nc, 4*L, D]
            D = vf.shape[-1]
            g = int(round((vf.shape[1] // 4) ** 0.5))
            # arch.py:173,177-178 hard-code 8x8 / 12x12 outputs; with the 16x16 / 24x24 token grids of the shipped
            # towers adaptive_avg_pool2d is exactly a 2x2 mean
            assert g == 16, "the video token grid must be 16x16"
            return ops.pool2(vf.view(vn * 4, g, g, D), False).view(-1, D)                 # 16x16 -> 8x8 per frame

        def context_tokens():
            if cn == 0:
                return None
            cf = self._projector("model.image_mm_projector", self.clip(context_images).contiguous())   # [Te, Lc, D]
            D = cf.shape[-1]
            g = int(round(cf.shape[1] ** 0.5))
            assert g == 24, "the context token grid must be 24x24"
            return ops.pool2(cf.view(cn, g, g, D), False).view(-1, D)                      # 24x24 -> 12x12 per frame

        if images.is_cuda and os.environ.get("VG_TOWERS_OVERLAP", "1") != "0" and vn and cn:
            # the two towers are independent and their GEMMs are small (M = 2050 / 4616 rows: ~1 round of tiles with a
            # long tail each): InternVideo2 goes to a second stream so that the tails of one fill under the other
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=images.device)
            main = torch.cuda.current_stream(images.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                vf = video_tokens()
            cf = context_tokens()
            main.wait_stream(self._side)
            vf.record_stream(main)
        else:
            vf = video_tokens()
            cf = context_tokens()
        if sharded:
            D = self.P.sd["model.embed_tokens.weight"].shape[1]          # the projectors end in the LLM width
            cf = comm.gather_rows(cf, te, 144, (D,), self.P.dtype, images.device)
            vf = comm.gather_rows(vf, te // 4, 4 * 64, (D,), self.P.dtype, images.device)
        return torch.cat([cf, vf], dim=0)


class LlamaDecoder:
    """HF LlamaModel / Phi3Model arithmetic (RMSNorm, rotate-half RoPE, GQA attention, SwiGLU) with a KV cache.  Phi-3 (the
    released checkpoint's LLM, R/model/videogpt_plus/model/language_model/phi3.py:29-40) is the same graph with the
    q|k|v and gate|up projections already fused in the checkpoint, and a sliding window: with cfg["sliding_window"] = w (2047 for
    Phi-3-mini-4k) position i attends to positions [i - w, i] — w + 1 keys, the mask transformers==4.41.0 (the reference's pin) builds
    in modeling_attn_mask_utils._make_causal_mask (diagonal = -w - 1) and its flash-attention path (window_size = (w, w)) — applied
    inside the prefill and decode attention kernels.  The released model at the reference's default NUM_FRAMES = 16 (S ~ 3370) needs it.

    Prefill runs eagerly (sequence length varies per clip).  A decode step is fully static — the token id, the
    cache position and the KV length all live in device memory (tok_dev / pos_dev) — so it is captured ONCE into
    a HIP graph (torch.cuda.CUDAGraph around the same C-ABI launches) and replayed per generated token: ~330
    kernel launches per token stop costing host time.  One instance is kept per model and max_len bucket."""
    HIST = 4096      # generated tokens per clip the device-side history holds (the reference's chat caps max_new_tokens at 512)

    def __init__(self, params, cfg, max_len, use_graph=None):
        self.P, self.c = params, cfg
        c = cfg
        self.D, self.H, self.Hkv = c["hidden"], c["num_heads"], c["num_kv_heads"]
        self.hd = self.D // self.H
        self.max_len = max_len
        self.window = int(c["sliding_window"]) + 1 if c.get("sliding_window") else 0     # visible keys, own position included
        dev, dt = params.device, params.dtype
        # zero-filled: vg_decode_attention2 masks rows past the position instead of skipping them (every row must hold finite values)
        self.kc = [torch.zeros(max_len, self.Hkv, self.hd, dtype=dt, device=dev) for _ in range(c["num_layers"])]
        self.vc = [torch.zeros(max_len, self.Hkv, self.hd, dtype=dt, device=dev) for _ in range(c["num_layers"])]
        inv = 1.0 / (c["rope_theta"] ** (torch.arange(0, self.hd, 2, dtype=torch.int64).float() / self.hd))
        fr = torch.arange(max_len).float()[:, None] * inv[None]
        self.cos = fr.cos().to(dev).contiguous()
        self.sin = fr.sin().to(dev).contiguous()
        self.pos = 0
        self.pos_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tok_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        # device-side loop state (vg_decode_advance): step counter, emitted / raw (pre-forcing) token history, forced-token table
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.hist = torch.zeros(self.HIST, dtype=torch.int64, device=dev)
        self.raw = torch.zeros(self.HIST, dtype=torch.int64, device=dev)
        self.forced = torch.full((self.HIST,), -1, dtype=torch.int64, device=dev)
        self.rope_cs = torch.zeros(self.hd, dtype=torch.float32, device=dev)     # [cos row | sin row] of *pos_dev for vg_decode_qkv_rope
        self.amax_acc = torch.zeros(1, dtype=torch.int64, device=dev)            # vg_argmax_partial's accumulator (vg_decode_step_end leaves it zero)
        self.hid_all = torch.empty(max_len, self.D, dtype=dt, device=dev)   # final-norm state of every position
        self.use_graph = (dev.type == "cuda") if use_graph is None else use_graph
        self.graph, self.graphs = None, {}
        self.kpw, self.kpw_min = 0, int(os.environ.get("VG_DEC_KPW_MIN", "2048"))
        self.attn_ws = None
        es = 2 if dt == torch.bfloat16 else 4
        ffn = c.get("ffn") or params.t("model.layers.0.mlp.down_proj.weight").shape[1]
        # the fused decode kernels' shape limits (vg_decode_gemv / vg_decode_attention); VG_DECODE_FUSED=0 = A/B knob
        # fp8 (e4m3) weights with per-row scales in the decode step's MLP GEMVs and the lm_head (cfg["decode_weights"] == "fp8": the decode
        # side of BASELINE config C4's "fp8 LLM path"; bf16 activations, fp32 accumulation; the prefill GEMMs stay bf16)
        self.w8 = c.get("decode_weights") == "fp8"
        # fp8 MFMA prefill GEMMs (cfg["prefill_gemm"] == "fp8"): vg_quantize_fp8_rows + vg_gemm_f8; the attention itself stays bf16
        self.f8_prefill = c.get("prefill_gemm") == "fp8"
        assert not self.f8_prefill or dt == torch.bfloat16, "the fp8 prefill pairs with a bf16 model"
        if self.w8:
            assert dt == torch.bfloat16 and self.D in (3072, 4096, 8192) and ffn in (8192, 14336), "fp8 decode weights: bf16 model, supported row lengths"
        self.fused_decode = (os.environ.get("VG_DECODE_FUSED", "1") != "0" and (self.H // self.Hkv) in (1, 2, 4, 8)
                             and self.hd % (16 // es) == 0 and self.hd * es <= 512 and max_len <= 8192
                             and max(self.D, ffn) * es <= 65536 and self.D % (16 // es) == 0 and ffn % (16 // es) == 0)
        assert self.fused_decode or not self.w8, "fp8 decode weights need the fused decode kernels"
        # vg_decode_layer: which roles of a layer run as one chained launch.  VG_DECODE_CHAIN=1 (attention + o_proj) / 3 (+ the MLP); default 0 =
        # separate launches: measured on C2 the chained launch only ties (27.4 us vs 20.1 + 8.0; 85.4 vs 84.7 for the whole layer — every
        # device-side hand-off is a fabric round trip, as the launch boundary it replaces is)
        self.chain_roles, self.chain_flags, self.chain_err = 0, None, None
        if self.fused_decode and dev.type == "cuda":
            self.chain_roles = min(ops.decode_layer_roles(self.H, self.Hkv, self.hd, self.D, ffn, dt), int(os.environ.get("VG_DECODE_CHAIN", "0")))
            if self.chain_roles == 2:
                self.chain_roles = 1
        # r06: RoPE + KV append in the q|k|v GEMV's epilogue, attention as a wave-private flash pass (bf16, head_dim 128); VG_DECODE_ROPE=0 = A/B knob
        self.rope_path = (self.fused_decode and dev.type == "cuda" and not self.chain_roles and os.environ.get("VG_DECODE_ROPE", "1") != "0"
                          and ops.decode_rope_path(self.H, self.Hkv, self.hd, self.D, dt))
        self.kpw2 = int(os.environ.get("VG_DEC2_KPW", "256"))

    def reset(self):
        self.pos = 0
        self.pos_dev.zero_()
        self.step_dev.zero_()

    def _layers(self, x, pos0, pos_dev, kv_hook=None):
        """decoder stack on x [S,D]; KV appended at pos (host value pos0, or *pos_dev when given).
        kv_hook(i): called after layer i wrote its K/V rows and before its attention (the sequence-parallel prefill
        all-gathers the other ranks' rows there)."""
        P, c = self.P, self.c
        S = x.shape[0]
        f8 = self.f8_prefill and S > 16 and pos_dev is None
        for i in range(c["num_layers"]):
            l = f"model.layers.{i}."
            qkv_names = [l + "self_attn.q_proj", l + "self_attn.k_proj", l + "self_attn.v_proj"]
            gu_names = [l + "mlp.gate_proj", l + "mlp.up_proj"]
            h = ops.rmsnorm(x, P.f32(l + "input_layernorm.weight"), c["rms_eps"])
            if f8:      # fp8 MFMA prefill (config C4): activations quantised per token, weights per output channel, fp32 accumulation
                qkv = ops.linear_f8(*ops.quantize_fp8(h), *P.fp8(qkv_names, stored=l + "self_attn.qkv_proj"))
            else:
                wqkv, _ = P.fused(qkv_names, stored=l + "self_attn.qkv_proj")
                qkv = ops.linear(h, wqkv)
            ops.rope_kv_append_(qkv, self.kc[i], self.vc[i], self.cos, self.sin, self.H, self.Hkv, self.hd, pos0, pos_dev)
            if kv_hook is not None:
                kv_hook(i)
            q = qkv[:, : self.H * self.hd].view(1, S, self.H, self.hd)
            if pos_dev is None:
                n = pos0 + S
                o = ops.attention(q, self.kc[i][:n].unsqueeze(0), self.vc[i][:n].unsqueeze(0), self.hd ** -0.5, causal=True,
                                  window=self.window)
            else:
                o = ops.attention_decode(q, self.kc[i], self.vc[i], pos_dev, self.hd ** -0.5, window=self.window)
            if f8:
                x = ops.linear_f8(*ops.quantize_fp8(o.view(S, self.D)), *P.fp8(l + "self_attn.o_proj"), residual=x)
                h = ops.rmsnorm(x, P.f32(l + "post_attention_layernorm.weight"), c["rms_eps"])
                a = ops.linear_f8(*ops.quantize_fp8(h), *P.fp8(gu_names, stored=l + "mlp.gate_up_proj"), glu=True)
                x = ops.linear_f8(*ops.quantize_fp8(a), *P.fp8(l + "mlp.down_proj"), residual=x)
                continue
            x = ops.linear(o.view(S, self.D), P.w(l + "self_attn.o_proj"), residual=x)
            h = ops.rmsnorm(x, P.f32(l + "post_attention_layernorm.weight"), c["rms_eps"])
            wgu, _ = P.fused(gu_names, stored=l + "mlp.gate_up_proj")
            # gate|up in one GEMM; for the decode step the SwiGLU runs in that GEMV's epilogue (ops.linear(glu=True))
            x = ops.linear(ops.linear(h, wgu, glu=True), P.w(l + "mlp.down_proj"), residual=x)
        return ops.rmsnorm(x, P.f32("model.norm.weight"), c["rms_eps"])

    def _layers_decode(self, x, rope_row=True):
        """the same stack for ONE new row at position *pos_dev, on the fused decode kernels: 5 launches per layer
        (norm+qkv[+rope+append], [rope+append+]attention+merge, o+residual, norm+gate|up+SwiGLU, down+residual) instead of 9."""
        P, c = self.P, self.c
        if self.attn_ws is None:
            self.attn_ws = ops.decode_attention_workspace(self.H, self.Hkv, self.hd, self.max_len, x.device)
        if self.chain_roles:
            if self.chain_flags is None:
                self.chain_flags = ops.decode_layer_flags(c["num_layers"], x.device)
                self.chain_err = torch.zeros((), dtype=torch.int32, device=x.device)
            self.chain_err.add_(self.chain_flags[:, 1].sum())      # the previous token's gave-up words survive the memset below (generate() checks)
            self.chain_flags.zero_()          # one memset per token: every layer's arrival stripes and go flags
        if self.rope_path and rope_row:
            ops.decode_advance_(self.pos_dev, 0, rope=(self.cos, self.sin, self.rope_cs))      # the cos / sin row of *pos_dev, once per token (decode_step_begin does it inside a captured step)
        for i in range(c["num_layers"]):
            l = f"model.layers.{i}."
            qkv_names = [l + "self_attn.q_proj", l + "self_attn.k_proj", l + "self_attn.v_proj"]
            gu_names = [l + "mlp.gate_proj", l + "mlp.up_proj"]
            wqkv, _ = P.fused(qkv_names, stored=l + "self_attn.qkv_proj")
            if self.rope_path:
                # norm -> q|k|v -> RoPE -> append in one launch; the attention starts on its K / V loads (4 launches per layer + the MLP's 2)
                q = ops.decode_qkv_rope(x, wqkv, P.f32(l + "input_layernorm.weight"), c["rms_eps"], self.kc[i], self.vc[i], self.rope_cs,
                                        self.pos_dev, self.H, self.Hkv, self.hd)
                o = ops.decode_attention2(q, self.kc[i], self.vc[i], self.H, self.Hkv, self.hd, self.pos_dev, self.hd ** -0.5, self.attn_ws,
                                          window=self.window, keys_per_wg=self.kpw2)
                x = ops.decode_gemv(o, P.w(l + "self_attn.o_proj"), residual=x)
                qkv = None
            else:
                qkv = ops.decode_gemv(x, wqkv, norm_w=P.f32(l + "input_layernorm.weight"), eps=c["rms_eps"])
            if qkv is None:
                pass
            elif self.chain_roles:
                # r03: attention, o_proj and (bf16 Llama widths) the MLP as roles of ONE launch whose GEMV workgroups fetch their weight
                # rows while the producer role still runs (vg_decode_layer) — bit-identical to the separate launches below
                w_o = P.w(l + "self_attn.o_proj")
                if self.chain_roles == 3 and not self.w8:
                    wgu, _ = P.fused(gu_names, stored=l + "mlp.gate_up_proj")
                    x = ops.decode_layer(qkv, self.kc[i], self.vc[i], self.cos, self.sin, self.H, self.Hkv, self.hd, self.pos_dev, self.hd ** -0.5,
                                         self.attn_ws, self.chain_flags[i], w_o, x, window=self.window,
                                         mlp=(P.f32(l + "post_attention_layernorm.weight"), c["rms_eps"], wgu, P.w(l + "mlp.down_proj")))
                    continue
                x = ops.decode_layer(qkv, self.kc[i], self.vc[i], self.cos, self.sin, self.H, self.Hkv, self.hd, self.pos_dev, self.hd ** -0.5,
                                     self.attn_ws, self.chain_flags[i], w_o, x, window=self.window)
            else:
                o = ops.decode_attention(qkv, self.kc[i], self.vc[i], self.cos, self.sin, self.H, self.Hkv, self.hd,
                                         self.pos_dev, self.hd ** -0.5, self.attn_ws, window=self.window, keys_per_wg=self.kpw)
                x = ops.decode_gemv(o, P.w(l + "self_attn.o_proj"), residual=x)
            if self.w8:
                # fp8 weights + row scales for the MLP (81 % of a layer's bytes) — the attention projections stay bf16: at K = 4096
                # an fp8 row is a single batch of loads per lane and the per-row reduction eats the gain (10.4 vs 9.1 us measured)
                a = ops.decode_gemv_w8(x, *P.fp8(gu_names, stored=l + "mlp.gate_up_proj"), norm_w=P.f32(l + "post_attention_layernorm.weight"),
                                       eps=c["rms_eps"], glu=True)
                x = ops.decode_gemv_w8(a, *P.fp8(l + "mlp.down_proj"), residual=x)
            else:
                wgu, _ = P.fused(gu_names, stored=l + "mlp.gate_up_proj")
                a = ops.decode_gemv(x, wgu, norm_w=P.f32(l + "post_attention_layernorm.weight"), eps=c["rms_eps"], glu=True)
                x = ops.decode_gemv(a, P.w(l + "mlp.down_proj"), residual=x)
        return ops.rmsnorm(x, P.f32("model.norm.weight"), c["rms_eps"])

    def forward(self, x):
        """eager: x [S,D] new tokens appended at self.pos -> final-normed hidden [S,D] (also kept in hid_all)."""
        S, pos = x.shape[0], self.pos
        assert pos + S <= self.max_len
        h = self._layers(x, pos, None)
        self.hid_all[pos:pos + S].copy_(h)
        self.pos = pos + S
        self.pos_dev.fill_(self.pos)
        return h

    def forward_sharded(self, x, comm):
        """sequence-parallel prefill over comm.world ranks (every rank holds the whole prompt x [S,D] and the whole model):
        rank r runs rows [r*m, (r+1)*m) (m = ceil(S/world)) through every layer; per layer the new K/V rows are all-gathered
        straight into every rank's KV cache (S/world x 2 x Hkv x hd elements per rank: ~0.9 MB at S = 1697, world = 8) and a
        row attends causally to the keys [0, its position].  The projections of a row do not depend on the other rows, so the
        prefill costs 1/world per rank (plus one weight pass) instead of a whole one on every rank; the final-norm rows are
        all-gathered at the end (the replicated decode continues from the last one)."""
        c = self.c
        S, W, r = x.shape[0], comm.world, comm.rank
        m = -(-S // W)
        a = min(r * m, S)
        n = max(min(m, S - a), 0)                       # this rank's rows [a, a+n)
        assert self.pos == 0 and W * m <= self.max_len
        send = torch.zeros(m, 2, self.Hkv, self.hd, dtype=x.dtype, device=x.device)
        recv = torch.empty(W * m, 2, self.Hkv, self.hd, dtype=x.dtype, device=x.device)

        def gather(i):
            if n:
                send[:n, 0].copy_(self.kc[i][a:a + n])
                send[:n, 1].copy_(self.vc[i][a:a + n])
            comm.all_gather_into(recv, send)
            self.kc[i][:W * m].copy_(recv[:, 0])        # rows >= S are padding: never read, overwritten by the decode appends
            self.vc[i][:W * m].copy_(recv[:, 1])

        hsend = torch.zeros(m, self.D, dtype=x.dtype, device=x.device)
        if n:
            hsend[:n].copy_(self._layers(x[a:a + n].contiguous(), a, None, kv_hook=gather))   # the same stack as forward(), bf16 or fp8 GEMMs
        else:
            for i in range(c["num_layers"]):      # a rank without rows still takes part in every gather
                gather(i)
        # every rank gets every final-norm row: the decode continues from the last one, and [SEG] tokens that sit INSIDE the prompt
        # (multi-turn) take their hidden state from prompt rows (S x D elements once per clip: 14 MB at S = 1697)
        hrecv = torch.empty(W * m, self.D, dtype=x.dtype, device=x.device)
        comm.all_gather_into(hrecv, hsend)
        self.hid_all[:W * m].copy_(hrecv)
        last = self.hid_all[S - 1:S].clone()
        self.pos = S
        self.pos_dev.fill_(S)
        return last

    def next_token(self, hidden_row):
        """lm_head + argmax of one final-norm row -> tok_dev (device int64[1])."""
        if self.w8 and hidden_row.shape[0] == 1:
            logits = ops.decode_gemv_w8(hidden_row.contiguous(), *self.P.fp8("lm_head"), out_dtype=torch.float32)
        else:
            logits = ops.linear(hidden_row, self.P.w("lm_head"), out_dtype=torch.float32)
        ops.argmax(logits.view(1, -1), out=self.tok_dev)

    def _decode_step(self):
        """static step: consume tok_dev at position *pos_dev, emit the next token into tok_dev, advance pos_dev."""
        if not (self.fused_decode and self.P.device.type == "cuda"):
            x = ops.embed(self.tok_dev, self.P.t("model.embed_tokens.weight"))
            h = self._layers_decode(x) if self.fused_decode else self._layers(x, 0, self.pos_dev)
            ops.store_row_(h, self.hid_all, self.pos_dev)
            self.next_token(h)
            self.advance(1)
            return
        # head and tail of the step as one launch each (embed + cos / sin row; token decode + row store + bookkeeping): 3 launches where there were 8
        x = ops.decode_step_begin(self.tok_dev, self.P.t("model.embed_tokens.weight"), self.pos_dev,
                                  rope=(self.cos, self.sin, self.rope_cs) if self.rope_path else None)
        h = self._layers_decode(x, rope_row=False)
        if self.w8:
            logits = ops.decode_gemv_w8(h.contiguous(), *self.P.fp8("lm_head"), out_dtype=torch.float32)
        else:
            logits = ops.linear(h, self.P.w("lm_head"), out_dtype=torch.float32)
        ops.argmax_partial(logits.view(-1), self.amax_acc)
        ops.decode_step_end(self.amax_acc, self.tok_dev, self.pos_dev, self.step_dev, h, self.hid_all, self.forced, self.hist, self.raw)

    def advance(self, inc):
        """the host's part of HF generate()'s loop, on the device: force / record the token in tok_dev, count the step, move the position"""
        ops.decode_advance_(self.pos_dev, inc, self.tok_dev, self.step_dev, self.forced, self.hist, self.raw)

    def set_forced(self, table):
        """{step: token id} applied to the emitted tokens on the device (step 0 = the token the prefill emits); None / {} = no forcing"""
        f = torch.full((self.HIST,), -1, dtype=torch.int64)
        for k, v in (table or {}).items():
            if 0 <= int(k) < self.HIST:
                f[int(k)] = int(v)
        self.forced.copy_(f.to(self.forced.device))

    def decode_step(self):
        assert self.pos + 1 <= self.max_len
        # long caches: two 64-key blocks per decode-attention workgroup (half the partials to publish, arrive and merge: the merge of a C2 prompt's
        # 54 splits took two passes); a launch parameter, so each setting has its own captured graph.  VG_DEC_KPW_MIN = first position that uses it
        self.kpw = 128 if (self.pos >= self.kpw_min and not self.rope_path) else 0
        self.graph = self.graphs.get(self.kpw)
        if not self.use_graph:
            self._decode_step()
        else:
            if self.graph is None:
                # one eager step first (lazy weight packing, kernel attribute setup), then rewind and capture
                snap_tok, snap_pos, snap_step = self.tok_dev.clone(), self.pos_dev.clone(), self.step_dev.clone()
                self._decode_step()
                torch.cuda.synchronize()
                self.tok_dev.copy_(snap_tok)
                self.pos_dev.copy_(snap_pos)
                self.step_dev.copy_(snap_step)
                g = torch.cuda.CUDAGraph()
                # thread-local capture mode: in multi-GPU runs the RCCL watchdog thread of torch.distributed polls events of
                # finished collectives; under the default (global) mode such a call from another thread can invalidate the capture
                with ops.graph_capture(g):      # (thread-local capture mode, cyclic GC held off: ops.graph_capture)
                    self._decode_step()
                self.graph = self.graphs[self.kpw] = g
                self.tok_dev.copy_(snap_tok)
                self.pos_dev.copy_(snap_pos)
                self.step_dev.copy_(snap_step)
            self.graph.replay()
        self.pos += 1


def splice(params, input_ids, visual):
    """prepare_inputs_labels_for_multimodal for one sample with one run of <image> placeholders —
    R/model/videogpt_plus/model/arch.py:271-371,453-467.  input_ids [L] (host, with -200) -> embeds [S,D]."""
    ids = input_ids
    pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten()
    table = params.t("model.embed_tokens.weight")
    dev = params.device
    if pos.numel() == 0:
        return ops.embed(ids.to(dev), table)
    s, e = int(pos[0]), int(pos[-1])
    parts = [ops.embed(ids[:s].to(dev), table), visual]
    if e + 1 < ids.numel():
        parts.append(ops.embed(ids[e + 1:].to(dev), table))
    return torch.cat(parts, dim=0).contiguous()


def stage_mark(stages, name):
    """diagnostics (bench.py's per-stage split): when `stages` is a list, wait for the device and append (name, host time)."""
    if stages is not None:
        import time
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        stages.append((name, time.perf_counter()))


def generate(params, cfg, towers, images, context_images, input_ids, max_new_tokens, eos_token_id=None, visual=None,
             token_hook=None, after_prefill=None, comm=None, trace=None, stages=None):
    """Steps A–D of VideoGLaMM_SAM2.inference_* (R/model/VideoGLaMM.py:609-655 / 781-831) with encode-once +
    KV-cache scheduling.  The hidden state the reference gathers for a [SEG] at output position p is the
    final-norm state of position p-1 = the row that produced the token, captured here as it is
    emitted.  token_hook(step, token_id) -> token_id | None: called with every emitted token AFTER the full lm_head + argmax (the place of a
    logits processor in HF generate()); a returned id replaces the token (teacher forcing in tests; synth.forced_tokens_hook for the synthetic-weight
    benchmark, whose random weights never emit [SEG] — no work is skipped).
    trace: optional dict; trace["argmax"] receives the model's own argmax of every step (before any forcing).
    eos_token_id: one id or several (HF generate() stops on any id of generation_config.eos_token_id).
    input_ids: host int64 [L] -> (output_ids host int64 [L+G], pred_embeddings device [N,256])."""
    seg_idx = cfg["seg_token_idx"]
    eos = set() if eos_token_id is None else ({int(eos_token_id)} if isinstance(eos_token_id, int) else {int(e) for e in eos_token_id})
    stage_mark(stages, "start")
    if visual is None:
        visual = towers.encode(images, context_images, comm)
    stage_mark(stages, "towers")
    x = splice(params, input_ids, visual)
    need = x.shape[0] + max_new_tokens + 1
    dec = getattr(params, "_decoder", None)
    if dec is None or dec.max_len < need:
        dec = LlamaDecoder(params, cfg["llm"], -(-need // 1024) * 1024)
        params._decoder = dec          # KV cache + captured decode graph are reused across clips
    dec.reset()
    if comm is not None and comm.world > 1 and x.shape[0] >= 64 * comm.world and os.environ.get("VG_PREFILL_SHARDED", "0") == "1":
        hidden = dec.forward_sharded(x, comm)   # opt-in sequence-parallel prefill: 1/world of the rows per rank (row chunks of another
        #                                         M take other GEMM routes: bf16 ids may differ from the single-GPU run; default = replicated)
    else:
        hidden = dec.forward(x)[-1:]            # hid_all rows 0..S-1: final-norm states of the spliced prompt
    added = x.shape[0] - input_ids.numel()      # "num_newly_added_tokens" (VideoGLaMM.py:613,786)
    ids = input_ids.tolist()
    if max_new_tokens > 0:
        dec.next_token(hidden)
    stage_mark(stages, "prefill")
    if after_prefill is not None:
        after_prefill()   # e.g. enqueue the (LLM-independent, MFMA-bound) Hiera pass on a side stream so that it
        #                   overlaps the HBM-bound decode loop below
    # The loop.  Forcing / recording / the position bump can happen on the device (LlamaDecoder.advance), so with no Python hook in the way step k + 1
    # can be enqueued BEFORE token k is read back (VG_DECODE_AHEAD=1: tokens return on a side stream behind per-step events; the step launched while
    # an EOS token was in flight is wasted — it writes cache rows past the end that nobody reads).  Built for VERDICT r05 item 1(c) and measured r06 on C2
    # (tools/lab/r06_c2_ab.sh, same box): the synchronous hand-over costs ~16 us per token in the serial decode stage (2.98 ms wall vs 2.966 ms of graph
    # time), and the run-ahead loop's events + side-stream copies make every replayed step ~0.17 ms LONGER (3.13 vs 2.97 ms per token; clip 252.6 vs
    # 248.2 ms) — so the default is the synchronous loop; the opt-in stays for runtimes where the hand-over is the larger term.
    # token_hook: a dict-carrying hook (synth.forced_tokens_hook: hook.forced_table = {step: id}) can be applied on the device; any other callable
    # is opaque Python and always keeps the synchronous hand-over.
    table = getattr(token_hook, "forced_table", None) if token_hook is not None else {}
    ahead = (table is not None and os.environ.get("VG_DECODE_AHEAD", "0") == "1" and max_new_tokens <= dec.HIST and params.device.type == "cuda"
             and dec.use_graph)
    if max_new_tokens > 0:
        dec.set_forced(table if ahead else None)
        dec.step_dev.zero_()
        dec.advance(0)                       # token 0 (emitted by the prefill): forced / recorded; the position stays
    if ahead:
        # tokens come back on a side stream that waits only for the step that emitted them (an event per step), never for the step enqueued after it
        main = torch.cuda.current_stream()
        rb = dec.__dict__.setdefault("_rb_stream", torch.cuda.Stream())
        host = dec.__dict__.setdefault("_rb_host", torch.zeros(2, dec.HIST, dtype=torch.int64).pin_memory())
        evs = [torch.cuda.Event()]           # evs[j]: token j is in hist[j] / raw[j]
        evs[0].record(main)
        known = 0
        while max_new_tokens > 0:
            if len(evs) < max_new_tokens:    # run ahead by one: step len(evs) - 1 consumes token len(evs) - 1 and emits token len(evs)
                dec.decode_step()
                evs.append(torch.cuda.Event())
                evs[-1].record(main)
                upto = len(evs) - 1          # read what the steps BEFORE this one emitted
            else:
                upto = len(evs)              # nothing left to launch: read the last token too
            with torch.cuda.stream(rb):
                rb.wait_event(evs[upto - 1])
                host[0, known:upto].copy_(dec.hist[known:upto], non_blocking=True)
                if trace is not None:
                    host[1, known:upto].copy_(dec.raw[known:upto], non_blocking=True)
            rb.synchronize()
            stop = False
            for j in range(known, upto):
                if trace is not None:
                    trace.setdefault("argmax", []).append(int(host[1, j]))
                ids.append(int(host[0, j]))
                known += 1
                if ids[-1] in eos or known == max_new_tokens:
                    stop = True
                    break
            if stop:
                break
    else:
        for step in range(max_new_tokens):
            nxt = int(dec.tok_dev[0])
            if trace is not None:
                trace.setdefault("argmax", []).append(nxt)
            if token_hook is not None:
                repl = token_hook(step, nxt)
                if repl is not None and int(repl) != nxt:
                    nxt = int(repl)
                    dec.tok_dev.fill_(nxt)
            ids.append(nxt)
            if nxt in eos or step == max_new_tokens - 1:
                break
            dec.decode_step()
    if dec.chain_roles and dec.chain_flags is not None and int(dec.chain_err) + int(dec.chain_flags[:, 1].sum()):
        # VG_DECODE_CHAIN: a workgroup of a chained layer launch stopped waiting for its producer role (bounded wait) — the ids above are not to be trusted
        raise ops._lib.VGKernelError("vg_decode_layer: a device-side wait gave up (flags[1] set); rerun with VG_DECODE_CHAIN=0")
    out_ids = torch.tensor(ids, dtype=torch.int64)
    # seg_token_mask = (output_ids[:,1:] == seg) left-padded by `added` (VideoGLaMM.py:630-633,803-806):
    # the row picked for a [SEG] at output position j is j-1+added, i.e. the state that emitted it
    rows = [j - 1 + added for j in range(1, len(ids)) if ids[j] == seg_idx]
    if not rows:
        stage_mark(stages, "decode")
        return out_ids, torch.empty(0, 256, dtype=params.dtype, device=params.device)
    h = dec.hid_all[torch.tensor(rows, device=params.device)]
    fc = "model.text_hidden_fcs.0."
    h = ops.linear(h, params.w(fc + "0"), params.b(fc + "0"), act=ops.ACT_RELU)
    emb = ops.linear(h, params.w(fc + "2"), params.b(fc + "2"))
    stage_mark(stages, "decode")
    return out_ids, emb
