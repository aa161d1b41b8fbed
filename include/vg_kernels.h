/*
 * vg_kernels.h — C ABI of libvgkernels.so, the hand-written gfx950 (MI355X / CDNA4) kernel
 * library behind the VideoGLaMM inference hot path.
 *
 * Conventions:
 *   - plain pointers and sizes only; every buffer (inputs, outputs, workspace) is owned by the
 *     caller and lives in device memory; no hidden allocation, no implicit synchronisation;
 *   - every entry point is stateless, re-entrant and stream-ordered on the hipStream_t passed last;
 *   - return value 0 = VG_OK, negative = error code; vg_last_error() gives a thread-local message;
 *   - activations / weights are VG_F32 or VG_BF16 (raw uint16 storage), accumulation is always fp32;
 *     bias / norm-weight / LayerScale vectors are always fp32;
 *   - tensors are token-major ("channels last"): [tokens, channels] with channels contiguous.
 *
 * R/ = the reference tree /root/reference/VideoGLaMM/. The reference has no native boundary on this
 * path (its only native file, sam2/csrc/connected_components.cu, is dead code on it —
 * R/model/segment_anything_2/sam2/sam2_video_predictor.py:971-975); each entry point below therefore
 * cites the PyTorch call site(s) whose arithmetic it replaces.
 */
#ifndef VG_KERNELS_H
#define VG_KERNELS_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vg_stream_t; /* hipStream_t */

#define VG_F32 0
#define VG_BF16 1

#define VG_OK 0
#define VG_ERR_ARG (-1)
#define VG_ERR_LAUNCH (-2)
#define VG_ERR_UNSUPPORTED (-3)

#define VG_ACT_NONE 0
#define VG_ACT_GELU 1       /* exact erf GELU: nn.GELU() */
#define VG_ACT_QUICK_GELU 2 /* x*sigmoid(1.702x): HF CLIP "quick_gelu" */
#define VG_ACT_RELU 3
#define VG_ACT_SILU 4
#define VG_ACT_SIGMOID 5

int vg_version(void);
const char* vg_last_error(void);
/* Binds nothing; checks that `device` is a gfx950 part. Returns CU count (>0) or a negative code. */
int vg_init(int device);
/* Node census of a captured HIP graph (hipGraph_t): counts[0] = kernel nodes, counts[1] = memcpy nodes, counts[2] = every other node.
 * Measurement aid for the graph-replayed loops (the SAM2 propagation the reference runs as Python calls per frame,
 * R/model/segment_anything_2/sam2/sam2_video_predictor.py:744-827; the decode step): launches per replay, read off the graph itself. */
int vg_graph_node_counts(void* graph, int64_t* counts);

/* ---- dense contraction (MFMA) ------------------------------------------------------------------
 * C[b] = ((act(A[b] @ W[b]^T + bias)) * gamma) + R[b]      A:[M,K] lda, W:[N,K] ldw, C:[M,N] ldc
 * Replaces every nn.Linear / 1x1 conv / im2col'd conv on the path, e.g.
 *   R/model/videogpt_plus/model/internvideo/internvideo2.py:193,207,257-261 (qkv/proj/fc1/fc2 + LayerScale :283-284),
 *   R/model/segment_anything_2/sam2/modeling/backbones/hieradet.py:56,82 ; sam2_utils.py:126-131 (MLP),
 *   R/model/segment_anything_2/sam2/modeling/sam/transformer.py:238-258 (q/k/v/out proj),
 *   R/model/segment_anything_2/sam2/modeling/sam/mask_decoder.py:227-234 (hypernetwork product, batched),
 *   HF LlamaAttention/LlamaMLP projections and lm_head (R/model/videogpt_plus/model/language_model/llama3_1.py:63-75).
 * in_dtype applies to A and W; out_dtype to C and R. K must be a multiple of 8 (bf16) / 4 (f32) and
 * lda/ldw multiples of the same; M, N arbitrary. batch>1 uses element strides sA/sW/sC/sR (sW may be 0).
 */
int vg_gemm(const void* A, int64_t lda, int64_t sA, const void* W, int64_t ldw, int64_t sW,
            void* C, int64_t ldc, int64_t sC, const float* bias, const float* gamma,
            const void* R, int64_t ldr, int64_t sR, int M, int N, int K, int batch,
            int in_dtype, int out_dtype, int act, int a_op, vg_stream_t stream);
/* a_op = 1: W holds 2N rows, gate rows then up rows; C[:, n] = silu(A·W[n] + b[n]) * (A·W[N+n] + b[N+n]) —
 * HF LlamaMLP act(gate_proj(x)) * up_proj(x) with the SwiGLU done in the epilogue (gate / up / act rounded to the
 * output dtype like the separate modules do).  M > 16 additionally needs 16-byte aligned C rows, no R / gamma / act. */

/* Short-row contraction with a row-wise producer and a column-pair consumer in the same launch (bf16, K in {64, 128, 192, 256}, N % 64 == 0):
 *   C = act(pro(A) @ W^T + bias) [axial RoPE] [+ R]
 *   pro: LayerNorm over the K columns (ln_w / ln_b / ln_eps; statistics in fp32, result rounded to bf16 — vg_layernorm followed by vg_gemm), or
 *        A + A2 with A2 [a2_rows, K] repeated over blocks of a2_rows rows (rounded to bf16 — vg_axpby followed by vg_gemm), or nothing (NULLs);
 *   RoPE (rope_cos != NULL): vg_rope_axial_heads on the bf16-rounded product's first rope_cols columns (heads of rope_ch channels), rows
 *        [rope_r0, rope_r1) of every block of rows_per_block rows, token = (row - rope_r0) % rope_grid; no activation / residual with it.
 *   a_block_stride != 0: A is a strided [B, rows_per_block, K] view — row m at (m / rows_per_block) * a_block_stride + (m % rows_per_block) * lda.
 * Replaces, per launch, the module sequences norm1 -> self_attn.{q,k,v}_proj -> apply_rotary_enc, norm2 -> cross_attn_image.q_proj -> rotary,
 * (memory + memory_pos) -> cross_attn_image.k_proj -> rotary with num_k_exclude_rope, norm3 -> linear1 -> ReLU of
 * R/model/segment_anything_2/sam2/modeling/memory_attention.py:60-99 + sam/transformer.py:289-327, and norm -> pwconv1 -> GELU of
 * memory_encoder.py:96-118.  The fp32 parity mode runs the separate entry points. */
int vg_gemm_rows(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, const void* R, int64_t ldr,
                 int64_t M, int N, int K, int act, const float* ln_w, const float* ln_b, float ln_eps, const void* A2, int64_t lda2,
                 int a2_rows, const float* rope_cos, const float* rope_sin, int rope_cols, int rope_ch, int rows_per_block, int64_t a_block_stride,
                 int rope_r0, int rope_r1, int rope_grid, int dtype, vg_stream_t stream);

/* A whole pre-norm MLP block on narrow rows in one launch (bf16; C in {144, 288}: Hiera stages 1 and 2; H % 32 == 0):
 *   y = x + W2 @ gelu(W1 @ LayerNorm(x) + b1) + b2        W1: [H, C] contiguous rows, W2: [C, H] contiguous rows, exact-erf GELU
 * — x = x + self.mlp(self.norm2(x)) of R/model/segment_anything_2/sam2/modeling/backbones/hieradet.py:160-168 (MLP: sam2_utils.py:108-132).
 * The LayerNorm output and the hidden activation are rounded to bf16 like the separate launches' tensors; they never leave the CU.
 * vg_mlp_rows_supported(C, H) != 0: this (C, H) has an instantiation; the fp32 parity mode runs vg_layernorm + vg_gemm x 2. */
int vg_mlp_rows_supported(int C, int H);
int vg_mlp_rows(const void* x, int64_t ldx, void* y, int64_t ldy, const float* ln_w, const float* ln_b, float eps, const void* w1,
                const float* b1, const void* w2, const float* b2, int64_t M, int C, int H, int dtype, vg_stream_t stream);

/* The same contraction with Hiera's window_partition / window_unpartition (backbones/utils.py:16-38,41-60, called from
 * hieradet.py:128-136,147-148) folded into it.  GEMM row m is the WINDOW-order row index (window (b,wy,wx), token (r,c)),
 * M = B*ceil(H/ws)*ceil(W/ws)*ws*ws including the zero padding rows the reference pads with.
 *   mode 1: A is the IMAGE-order tensor [B,H,W,K] (row stride lda); row m is gathered from it (padding rows read
 *           zero_row, K zeros) and C is written in window order — window_partition + qkv projection in one pass.
 *   mode 2: A is window-order; C (row stride ldc) and R (ldr) are IMAGE-order [B,H,W,N]: row m is scattered to its
 *           pixel, padding rows are dropped — proj + window_unpartition + residual add in one pass. */
int vg_gemm_window(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                   const float* gamma, const void* R, int64_t ldr, int N, int K, int in_dtype, int out_dtype, int act,
                   int mode, int B, int H, int Wd, int ws, const void* zero_row, vg_stream_t stream);

/* y = act(LayerNorm_K(A rows) @ W^T + bias) — the norm1 -> q|k|v and norm2 -> fc1 pairs of Hiera's MultiScaleBlock at the widths of stages 1 and 2
 * (R/model/segment_anything_2/sam2/modeling/backbones/hieradet.py:117-123, 160-166), the LayerNorm applied to the rows in registers.  Built for the
 * row-register route only: bf16, K in {144, 288}, N a multiple of 16, >= 65536 rows, act none | GELU; VG_ERR_UNSUPPORTED otherwise (the caller then runs
 * vg_layernorm + vg_gemm / vg_gemm_window).  window = 1: rows gathered from an image-order [B,H,W,K] tensor as vg_gemm_window's mode 1 (M is ignored;
 * padding rows stay zero BEHIND the norm, as F.pad of the normalised tensor does); zero_row as in vg_gemm_window. */
int vg_gemm_ln(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
               const float* ln_w, const float* ln_b, float ln_eps, int64_t M, int N, int K, int act,
               int window, int B, int H, int Wd, int ws, const void* zero_row, int dtype, vg_stream_t stream);


/* ---- attention (flash-style, LDS-staged QK tiles, in-register online softmax) -------------------
 * O[b,i,h,:] = softmax_j(scale * Q[b,i,h,:]·K[b,j,g,:] (+causal mask)) @ V[b,j,g,:],  g = h / (Hq/Hkv)
 * causal = 1: key j visible to query i iff j <= i + (Skv - Sq).  causal = c >= 2: the same with a sliding window of c
 * visible keys, the query's own position included: i + (Skv - Sq) - c < j <= i + (Skv - Sq) (HF Phi-3's sliding_window mask,
 * transformers==4.41.0 modeling_attn_mask_utils._make_causal_mask: c = sliding_window + 1).  causal = -w (w > 0): block-diagonal mask — the
 * sequence is a pack of independent w-token windows (key j visible to query i iff j/w == i/w; Sq == Skv), which lets
 * Hiera's 16- / 64-token windows share 128-query tiles.  D % 8 == 0, D <= 256.
 * Strides are in elements: *_sb batch, *_ss token, *_sh head; the head dim is contiguous.
 * Replaces F.scaled_dot_product_attention / naive softmax(QK^T)V at
 *   R/model/segment_anything_2/sam2/modeling/backbones/hieradet.py:72-76,
 *   R/model/segment_anything_2/sam2/modeling/sam/transformer.py:249-255,316-322,
 *   R/model/videogpt_plus/model/internvideo/internvideo2.py:200-206,
 *   HF CLIPAttention / LlamaAttention (third-party, transformers==4.41.0).
 */
int vg_attention(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv,
                 int Sq, int Skv, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb,
                 int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb,
                 int64_t o_ss, int64_t o_sh, float scale, int causal, int dtype, vg_stream_t stream);

/* Attention inside independent windows: O[w,i,h,:] = softmax_j(scale * Q[w,i,h,:].K[w,j,h,:]) @ V[w,j,h,:], i in [0, wq), j in
 * [0, wtok) — Hiera's MultiScaleAttention on window_partition'ed tokens (R/model/segment_anything_2/sam2/modeling/backbones/
 * hieradet.py:37-83, backbones/utils.py:16-38; wq < wtok: the q-pooled first block of a stage, hieradet.py:64-68).  bf16 only.
 *   wq == wtok == 256, D in {64, 72, 80}: one workgroup per (window, head), the window's K and V staged once (stage 3);
 *   D == 72 and (wq, wtok) in {(16,16), (64,64), (4,16), (16,64)}: one wave per (window, head) on 16x16x32 MFMA tiles.
 * *_sb = window stride, *_ss token stride, *_sh head stride (elements); O may be a strided view as well. */
int vg_window_attention(const void* Q, const void* K, const void* V, void* O, int Bw, int H, int wq, int wtok, int D,
                        int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                        int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                        float scale, int dtype, vg_stream_t stream);

/* Same contraction with the KV range split over `nsplit` workgroups per (query tile, head) and a merge pass —
 * the decode-step shape (Sq = 1, Skv = thousands: one workgroup per head would walk the whole KV cache
 * serially).  workspace: fp32, >= B*Hq*nsplit*Sq*(D+2) floats, caller-owned.  nsplit = 1 == vg_attention. */
int vg_attention_splitkv(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv,
                         int Sq, int Skv, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb,
                         int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb,
                         int64_t o_ss, int64_t o_sh, float scale, int causal, int dtype, float* workspace,
                         int64_t ws_floats, int nsplit, const int* skv_dev, vg_stream_t stream);
/* skv_dev (may be NULL): when given, the kernel uses Skv = *skv_dev + Sq read from device memory instead of the
 * host value, so one captured decode step can be replayed from a HIP graph as the KV cache grows. */

/* Attention whose VALUES (and output) have DV dims while queries / keys have D (one head group, no mask): SAM2's memory cross-attention
 * with the v-projection moved behind the attention — the reference computes softmax(q k^T) (M Wv^T + b)
 * (R/model/segment_anything_2/sam2/modeling/sam/transformer.py:289-327, memory_attention.py:60-99); softmax rows sum to one, so that equals
 * (softmax(q k^T) M) Wv^T + b: the projection then runs on the 4096 query rows instead of the ~28 000 memory rows and the PV half of the
 * attention works on the memory's own 64 dims.  Built for bf16, D in (128, 256], DV = 64 (VG_ERR_UNSUPPORTED otherwise).  V rows [.., DV], O rows [.., DV];
 * workspace (nsplit > 1): fp32, >= B*H*nsplit*Sq*(DV+2) floats. */
int vg_attention_dv(const void* Q, const void* K, const void* V, void* O, int B, int H, int Sq, int Skv, int D, int DV,
                    int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb,
                    int64_t v_ss, int64_t v_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float scale, int dtype,
                    float* workspace, int64_t ws_floats, int nsplit, vg_stream_t stream);

/* Fused HF rotate-half RoPE + KV-cache append for one decoder layer (HF LlamaAttention.forward: apply_rotary_pos_emb
 * then cache update).  qkv: fused projection output [S, (H+2*Hkv)*D], row stride ld; q rotated in place, rotated k
 * and plain v written to k_cache/v_cache[pos+s] ([max_len,Hkv,D]).  pos = pos_dev ? *pos_dev : pos0. */
int vg_rope_kv_append(void* qkv, int64_t ld, void* k_cache, void* v_cache, const float* cos, const float* sin,
                      int S, int H, int Hkv, int D, int pos0, const int* pos_dev, int dtype, vg_stream_t stream);
/* dst[(*idx_dev + idx_off)*n + i] = src[i]; *p += v — device-indexed bookkeeping for graph-replayed decode. */
int vg_store_row(const void* src, void* dst, int64_t n, const int* idx_dev, int idx_off, int dtype, vg_stream_t stream);
int vg_add_int(int* p, int v, vg_stream_t stream);

/* ---- fused single-token decode step (HF LlamaDecoderLayer.forward at q_len = 1, transformers==4.41.0; called per
 * generated token from R/model/VideoGLaMM.py:616-628 / 789-801 via self.generate) ---------------------------------
 * vg_decode_gemv: y[N] = f(x)[K] . W[N,K]^T (+ R[N]).  norm_w != NULL: f = LlamaRMSNorm(eps) with weight norm_w (fp32
 *   copy of the norm weight) — input_layernorm / post_attention_layernorm fused into the q|k|v and gate|up projections.
 *   glu != 0: W is [2N,K] = gate rows | up rows and y[n] = silu(x.gate_n) * (x.up_n) (LlamaMLP).  Row stride of W = ldw.
 * vg_decode_attention: qkv = fused projection row [(H+2*Hkv)*D] of the new token at position p = *pos_dev.  Applies
 *   rotate-half RoPE (cos/sin tables [max_len, D/2]) to q and k, appends k/v to the caches ([max_len,Hkv,D]) and writes
 *   softmax(q.K[lo..p]^T * scale).V[lo..p] to out [H*D], lo = window > 0 ? max(0, p + 1 - window) : 0 (window = number of
 *   visible positions, the new one included; 0 = the whole cache).  workspace: fp32, vg_decode_attention_ws_floats() floats, must be
 *   zero-filled ONCE by the caller before the first launch (it ends with self-resetting per-head counters).  keys_per_wg: 0 / 64 =
 *   one workgroup per 64 cache positions and KV head; 128 = a hint for long caches (>= ~2048 positions): two 64-key blocks per
 *   workgroup, so half as many partial results are published and merged (same result up to fp32 summation order; honoured for the
 *   bf16 Llama-3 / Phi-3 head shapes, ignored elsewhere). */
int vg_decode_gemv(const void* x, const void* W, int64_t ldw, void* y, const float* norm_w, float eps,
                   const void* R, int N, int K, int glu, int in_dtype, int out_dtype, vg_stream_t stream);
int64_t vg_decode_attention_ws_floats(int H, int Hkv, int D, int max_len);
int vg_decode_attention(const void* qkv, void* k_cache, void* v_cache, const float* cos, const float* sin,
                        void* out, int H, int Hkv, int D, int max_len, int window, float scale, const int* pos_dev,
                        float* workspace, int64_t ws_floats, int keys_per_wg, int dtype, vg_stream_t stream);

/* r06 — the decode step with the RoPE in the projection and the attention as a wave-private flash pass (bf16, head_dim 128):
 * vg_decode_qkv_rope: q|k|v projection of the new token (RMSNorm fused as in vg_decode_gemv) with rotate-half RoPE of the q and k heads and the
 *   KV-cache append in the GEMV's epilogue: q_out [H*D] receives the rotated query heads, k_cache / v_cache ([max_len,Hkv,D]) row *pos_dev the
 *   rotated key / the value row.  rope_cs: fp32 [2][D/2] = cos row | sin row of position *pos_dev (kept current by vg_decode_advance).
 *   Replaces HF LlamaAttention.forward's q_proj / k_proj / v_proj + apply_rotary_pos_emb + cache update at q_len = 1
 *   (R/model/videogpt_plus/model/language_model/llama3_1.py:40-108 via generate).
 * vg_decode_attention2: softmax(q.K[lo..p]^T * scale).V[lo..p] for the pre-rotated q against caches that already hold row p; keys_per_wg 128 / 256.
 *   The caches must hold FINITE values in every row (allocate them zero-filled): rows past p are masked, not skipped.  workspace: the
 *   vg_decode_attention one (zero-filled once; the last Hkv words are self-resetting counters).
 * vg_decode_advance: the decode loop's bookkeeping on the device.  tok / step given (both or neither) — k = *step; raw[k] = *tok; *tok = forced[k] >= 0 ?
 *   forced[k] : *tok; hist[k] = *tok; *step = k + 1 — then *pos += inc (0 / 1); rope_cs given — rope_cs = rows *pos of cos / sin ([max_len, half_dim],
 *   after the increment).  forced / hist / raw may be NULL.  What HF generate()'s loop does on the host between two forward calls (logits
 *   processor, input_ids append, cache_position + 1). */
int vg_decode_qkv_rope_supported(int H, int Hkv, int D, int K, int dtype);
int vg_decode_qkv_rope(const void* x, const void* Wqkv, int64_t ldw, const float* norm_w, float eps, void* q_out, void* k_cache,
                       void* v_cache, const float* rope_cs, const int* pos_dev, int H, int Hkv, int D, int K, int dtype,
                       vg_stream_t stream);
int vg_decode_attention2_supported(int H, int Hkv, int D, int dtype);
int vg_decode_attention2(const void* q, const void* k_cache, const void* v_cache, void* out, int H, int Hkv, int D, int max_len,
                         int window, float scale, const int* pos_dev, float* workspace, int64_t ws_floats, int keys_per_wg, int dtype,
                         vg_stream_t stream);
/* The head and tail of a captured decode step in one launch each: vg_decode_step_begin — x = table[*tok] (vg_embed of one row) and rope_cs = rows *pos of cos /
 *   sin (rope_cs may be NULL); vg_argmax_partial — vg_argmax's first stage: packed (value, ~index) keys atomicMax-ed into acc[row], acc zero on entry;
 *   vg_decode_step_end — *tok = index decoded from acc[0] (acc[0] left zero), hid_all[*pos] = row (vg_store_row), then vg_decode_advance's bookkeeping with
 *   inc = 1.  Same results as vg_embed, vg_decode_advance, vg_argmax, vg_store_row, vg_decode_advance in that order. */
int vg_decode_step_begin(const int64_t* tok, const void* table, void* x, int D, int dtype, const int* pos, const float* cos, const float* sin,
                         float* rope_cs, int half_dim, vg_stream_t stream);
int vg_argmax_partial(const void* x, int64_t rows, int n, uint64_t* acc, int dtype, vg_stream_t stream);
int vg_decode_step_end(uint64_t* acc, int64_t* tok, int* pos, int* step, const int64_t* forced, int n_forced, int64_t* hist, int64_t* raw, int cap,
                       const void* row, void* hid_all, int D, int dtype, vg_stream_t stream);
int vg_decode_advance(int64_t* tok, int* pos, int* step, const int64_t* forced, int n_forced, int64_t* hist, int64_t* raw, int cap,
                      const float* cos, const float* sin, float* rope_cs, int half_dim, int inc, vg_stream_t stream);

/* vg_decode_layer (r03): everything of a decoder layer behind the q|k|v projection as ONE launch — vg_decode_attention, then
 *   y_o = attn_out . Wo^T + resid (vg_decode_gemv), and when Wgu != NULL also act = SwiGLU(RMSNorm(y_o; norm_w, eps) . Wgu^T) and
 *   y = act . Wdown^T + y_o.  The GEMV workgroups request their weight rows before the row they multiply exists and wait for it on
 *   device-side arrival counters (csrc/vg_decode.hip: decode_layer_kernel), so the weight stream no longer idles behind the
 *   latency-bound attention and between the GEMVs.  Results are bit-identical to the separate calls.  The same zero-filled
 *   workspace as vg_decode_attention; flags: vg_decode_layer_flag_ints() ints, 128-byte aligned, zero-filled by the caller before
 *   EVERY launch (one memset per token over the regions of all layers); word [1] != 0 afterwards = a device-side wait gave up.
 *   vg_decode_layer_roles: 0 = shape not covered (use the separate calls), 1 = attention + o_proj
 *   (pass Wgu = NULL), 3 = the MLP as well.  HF LlamaDecoderLayer.forward at q_len = 1 (R/model/VideoGLaMM.py:616-628 via generate). */
int vg_decode_layer_roles(int H, int Hkv, int D, int hidden, int inter, int dtype);
int64_t vg_decode_layer_flag_ints(void);
int vg_decode_layer(const void* qkv, void* k_cache, void* v_cache, const float* cos, const float* sin, void* attn_out,
                    int H, int Hkv, int D, int max_len, int window, float scale, const int* pos_dev,
                    float* workspace, int64_t ws_floats, int32_t* flags,
                    const void* Wo, int64_t ldo, const void* resid, void* y_o,
                    const float* norm_w, float eps, const void* Wgu, int64_t ldgu, void* act,
                    const void* Wdown, int64_t lddown, void* y, int hidden, int inter, int dtype, vg_stream_t stream);

/* ---- row normalisation -------------------------------------------------------------------------
 * LayerNorm over the last dim (biased variance, two-pass fp32): nn.LayerNorm and LayerNorm2d
 * (R/model/segment_anything_2/sam2/modeling/sam2_utils.py:137-149 — channels-last makes them identical).
 * RMSNorm: R/model/videogpt_plus/model/internvideo/internvideo2.py:134-145 and HF LlamaRMSNorm.
 * x rows have stride ldx elements, y rows ldy.
 */
int vg_layernorm(const void* x, int64_t ldx, const float* w, const float* b, void* y, int64_t ldy,
                 int64_t rows, int C, float eps, int in_dtype, int out_dtype, vg_stream_t stream);
int vg_rmsnorm(const void* x, int64_t ldx, const float* w, void* y, int64_t ldy, int64_t rows, int C,
               float eps, int in_dtype, int out_dtype, vg_stream_t stream);

/* ---- SAM2 two-way transformer, image side of the image -> token cross-attention, fused (r04) --------
 * Replaces, for the 4096 image rows of every (frame, object) instance, the chain q_proj -> Attention (8 heads x 16, keys = the nt prompt/output
 * tokens) -> out_proj -> + residual -> LayerNorm -> + dense PE of TwoWayAttentionBlock.forward
 * (R/model/segment_anything_2/sam2/modeling/sam/transformer.py:185-193 with Attention.forward :236-260) by one pass over the rows:
 *   s = xpe . u2^T + c2  (columns (head h, token t) = h * TP + t),  a = softmax over t < nt inside each head,  y = a . w2t^T + bo,
 *   x_out = LayerNorm(x + y),  xpe_out = x_out + pe.
 * u2 [N, 8 TP, 256], c2 [N, 8 TP] (fp32), w2t [N, 256, 8 TP] carry the token side (k / v projections folded into the q / out projection weights:
 * videoglamm_amd/sam2.py:_i2t_fused); x_out, xpe_out [N, P, 256]; xpe, x [x_instances, P, 256] — instance n reads slot n % x_instances (the first
 * block of a frame's objects shares the frame's embedding); pe [P, 256]; bf16 only; TP = 8 or 16; nt <= TP.
 */
int vg_twoway_image_update(const void* xpe, const void* x, const void* u2, const float* c2, const void* w2t, const float* bo,
                           const float* ln_w, const float* ln_b, float eps, const void* pe, void* x_out, void* xpe_out,
                           int N, int x_instances, int P, int nt, int TP, int dtype, vg_stream_t stream);
/* Token-side layout of the fused two-way path: gather = 0: x [N, nt, 128] -> out [N, 8, TP, 128], row (h, t) = head h's 16 channels of token t,
 * zeros elsewhere (the per-head products of Attention.forward, sam/transformer.py:236-260, as ONE small GEMM on block-diagonal rows);
 * gather = 1: the inverse read — x [N, 8, TP, 128] -> out [N, nt, 128], channel c of token t from row (c / 16, t).  bf16 / fp32. */
int vg_heads_blockdiag(const void* x, void* out, int N, int nt, int TP, int gather, int dtype, vg_stream_t stream);

/* Output upscaling + hypernetwork product of the mask decoder, fused (r04): MaskDecoder.predict_masks
 * (R/model/segment_anything_2/sam2/modeling/sam/mask_decoder.py:225-245): ConvTranspose2d(256 -> 64, k2 s2) + feat_s1 -> LayerNorm2d -> GELU ->
 * ConvTranspose2d(64 -> 32, k2 s2) + feat_s0 -> GELU -> masks[k] = hyper_in[k] . upscaled (k = 0..3).
 * x [N, es*es, 256] (the two-way transformer's image output); w0 [4*64, 256], w1 [4*32, 64]: the ConvT weights as GEMM weights, row (dy*2+dx)*Cout + co
 * (Params.convT_w); s1 [images, (2es)^2, 64], s0 [images, (4es)^2, 32] channels-last, instance n uses image n % images; hyper [N, 4, 32];
 * masks fp32 [N, 4, 4es, 4es].  bf16 only.
 */
int vg_mask_upscale(const void* x, const void* w0, const float* b0, const void* s1, const float* ln_w, const float* ln_b, float eps,
                    const void* w1, const float* b1, const void* s0, const void* hyper, float* masks, int N, int images, int es, int dtype,
                    vg_stream_t stream);

/* ---- pointwise --------------------------------------------------------------------------------- */
/* out[i] = alpha*a[i] + beta*b[i % b_period]  (b may be NULL -> alpha*a[i] + beta) */
int vg_axpby(const void* a, const void* b, void* out, int64_t n, float alpha, float beta,
             int64_t b_period, int a_dtype, int b_dtype, int out_dtype, vg_stream_t stream);
/* y = act(x) */
int vg_activation(const void* x, void* y, int64_t n, int act, int in_dtype, int out_dtype,
                  vg_stream_t stream);
/* y[m,f] = silu(gu[m,f]) * gu[m,F+f]  (HF LlamaMLP: down(act(gate(x))*up(x)); gate|up packed) */
int vg_swiglu(const void* gu, void* y, int64_t M, int F, int dtype, vg_stream_t stream);
/* out = in converted */
int vg_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, vg_stream_t stream);
/* out[n,i] = cond[n] > 0 ? a[n,i] : (b ? b[i % b_period] : fill)   (fp32 cond); ld_out: elements between consecutive rows of out (0 = inner:
 * contiguous) — the object pointers go straight into their rows of the memory bank.
 * R/model/segment_anything_2/sam2/modeling/sam2_base.py:355-364,390-401 */
int vg_where_rows(const float* cond, const void* a, const void* b, void* out, int64_t rows,
                  int64_t inner, int64_t b_period, float fill, int64_t ld_out, int dtype, vg_stream_t stream);
/* mask fed to the memory encoder: out = (binarize ? (x>0) : sigmoid(x)) * scale + bias
 * R/model/segment_anything_2/sam2/modeling/sam2_base.py:684-693 */
int vg_mask_for_mem(const float* x, void* out, int64_t n, int binarize, float scale, float bias,
                    int out_dtype, vg_stream_t stream);
/* out[i] = x[i] > 0  (uint8)  R/model/VideoGLaMM.py:763,873 */
int vg_threshold(const float* x, uint8_t* out, int64_t n, vg_stream_t stream);
/* HF rotate-half RoPE, in place on x:[S,H,D] (token stride x_ss, head stride x_sh); cos/sin:[*,D/2] f32,
 * row used for token s is pos0+s.  HF modeling_llama.apply_rotary_pos_emb. */
int vg_rope_half(void* x, int64_t x_ss, int64_t x_sh, const float* cos, const float* sin, int S, int H,
                 int D, int pos0, int dtype, vg_stream_t stream);
/* SAM2 axial (complex-pair) RoPE, in place on x:[B,N,C] contiguous, only tokens n < n_rope;
 * token n uses table row n % n_grid.  cos/sin:[n_grid,C/2] f32.
 * R/model/segment_anything_2/sam2/modeling/position_encoding.py:174-216, sam/transformer.py:306-312 */
int vg_rope_axial(void* x, const float* cos, const float* sin, int B, int N, int C, int n_rope,
                  int n_grid, int dtype, vg_stream_t stream);
/* The same rotation on a STRIDED view holding H heads of Ch channels per row (bf16): x[b, n, h*Ch + c] at x + b*sb + n*ld (elements), every head
 * with the one [n_grid, Ch/2] table — the q | k columns of a fused q|k|v projection in one in-place launch (SAM2 memory self-attention, r04). */
int vg_rope_axial_heads(void* x, int64_t ld, int64_t sb, const float* cos, const float* sin, int B, int H, int Ch, int n_rope,
                        int n_grid, int dtype, vg_stream_t stream);
/* out[i,:] = table[ids[i],:]   (nn.Embedding) */
int vg_embed(const int64_t* ids, const void* table, void* out, int64_t n, int D, int dtype,
             vg_stream_t stream);
/* out[r] = first index of the row maximum (torch.argmax)  x:[rows,n] */
int vg_argmax(const void* x, int64_t rows, int n, int64_t* out, int dtype, vg_stream_t stream);

/* SAM2 mask selection, one call per decoder batch: masks fp32 [N,4,HW], ious fp32 [N,4], tokens [N,4,C].
 * mode 0: dynamic multimask via stability (R/.../sam/mask_decoder.py:247-295), token 0 returned;
 * mode 1: best predicted IoU among tokens 1..3 (R/.../sam2_base.py:376-386), that token returned.
 * out_mask [N,HW] fp32, out_iou [N] fp32, out_token [N,C] (may be NULL), out_idx [N] int32 (may be NULL). */
int vg_multimask_select(const float* masks, const float* ious, const void* tokens, float* out_mask,
                        float* out_iou, void* out_token, int* out_idx, int N, int64_t HW, int C, float delta,
                        float thresh, int mode, int token_dtype, vg_stream_t stream);

/* ---- spatial / layout (all NHWC) ---------------------------------------------------------------- */
/* 5-D gather copy: out (contiguous, dims n0..n4) = in at element strides s0..s4 */
int vg_permute5(const void* in, void* out, const int64_t dims[5], const int64_t strides[5], int dtype,
                vg_stream_t stream);
/* x:[B,H,W,C] -> cols:[B*Ho*Wo, Kpad], column (ky*kw+kx)*C + c, zero padded to Kpad.
 * Feeds vg_gemm for nn.Conv2d: hieradet patch embed (R/.../backbones/utils.py:65-95), MaskDownSampler
 * (R/.../memory_encoder.py:17-58), IV2/CLIP patch embeds. */
int vg_im2col(const void* x, void* cols, int B, int H, int W, int C, int kh, int kw, int stride,
              int pad, int Kpad, int dtype, vg_stream_t stream);
/* depthwise kxk conv, stride 1, pad k/2, NHWC; w:[k*k, C] f32, bias:[C] f32  (CXBlock dwconv,
 * R/.../memory_encoder.py:81-87) */
int vg_dwconv(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C,
              int k, int dtype, vg_stream_t stream);
/* Conv2d(k = 3, stride 2, pad 1) + bias -> LayerNorm over the channels (eps) -> GELU(erf) in ONE pass, channels-last:
 * x:[B,H,W,Cin] -> y:[B,(H+1)/2,(W+1)/2,Cout]; w:[Cout, ldw] in the tensors' dtype with column (ky*3+kx)*Cin + c (vg_im2col's column order);
 * bias / ln_w / ln_b fp32.  The few-channel stages of SAM2's MaskDownSampler (R/.../memory_encoder.py:17-63: 1->4->16 channels at
 * 512^2 / 256^2 outputs, LayerNorm2d + GELU after every conv) — built for exactly the pairs 1->4 and 4->16, VG_ERR_UNSUPPORTED otherwise. */
int vg_conv3s2_ln_gelu(const void* x, const void* w, int64_t ldw, const float* bias, const float* ln_w, const float* ln_b, float eps,
                       void* y, int B, int H, int W, int Cin, int Cout, int dtype, vg_stream_t stream);
/* ConvTranspose2d k2 s2 tail: g:[B,H,W,4,C] (GEMM output, tap = dy*2+dx) -> y:[B,2H,2W,C] (+bias)
 * R/.../sam/mask_decoder.py:64-73 */
int vg_pixel_shuffle2(const void* g, const float* bias, void* y, int B, int H, int W, int C, int dtype,
                      vg_stream_t stream);
/* 2x2 stride-2 max / mean pooling over an NHWC grid (Hiera q-pool hieradet.py:20-34,64-66;
 * adaptive_avg_pool2d with integer ratio 2, R/model/videogpt_plus/model/arch.py:88-96) */
int vg_pool2(const void* x, void* y, int B, int H, int W, int C, int64_t x_pix_stride, int is_max,
             int dtype, vg_stream_t stream);
/* window partition with zero padding / reverse with crop (R/.../backbones/utils.py:16-62) */
int vg_window_partition(const void* x, void* win, int B, int H, int W, int C, int ws, int dtype,
                        vg_stream_t stream);
int vg_window_unpartition(const void* win, void* x, int B, int H, int W, int C, int ws, int dtype,
                          vg_stream_t stream);
/* bilinear resize, align_corners=False, no antialias, planar fp32 [N,Hi,Wi] -> [N,Ho,Wo]
 * (F.interpolate at sam2_base.py:368-374, sam2_video_predictor.py:509-516, VideoGLaMM.py:147-153) */
int vg_bilinear(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, vg_stream_t stream);
/* vg_bilinear + (logit > 0) in one pass: uint8 masks [N,Ho,Wo] (0 / 1) from low-resolution fp32 logits [N,Hi,Wi] — the final
 * `postprocess_masks(...) > 0` / `video_res_masks > 0.0` of R/model/VideoGLaMM.py:147-153,757-766,869-875 without materialising
 * the fp32 logits at output resolution.  Bit-identical to vg_threshold(vg_bilinear(x)). */
int vg_bilinear_mask(const float* in, uint8_t* out, int N, int Hi, int Wi, int Ho, int Wo, vg_stream_t stream);
/* y:[B,2H,2W,C] = lateral + nearest2x(top:[B,H,W,C])   (FPN top-down, image_encoder.py:113-127) */
int vg_upsample2_add(const void* lateral, const void* top, void* y, int B, int H, int W, int C,
                     int dtype, vg_stream_t stream);

/* vg_decode_gemv with fp8 weights (config C4's LLM path, decode side): W8:[N or 2N, K] bytes in OCP e4m3, one fp32
 * scale per weight row (w ~ scale[n] * fp8); x, the norm and the epilogue are unchanged (bf16 activations, fp32
 * accumulation): y[n] = scale[n] * sum_k fp8(W8[n,k]) * xn[k].  Halves the bytes a decode step streams.
 * K in {3072, 4096, 8192, 14336}. */
int vg_decode_gemv_w8(const void* x, const uint8_t* W8, int64_t ldw, const float* wscale, void* y, const float* norm_w,
                      float eps, const void* R, int N, int K, int glu, int out_dtype, vg_stream_t stream);
/* vg_gemm for grids that leave most of the chip idle (few 128x128 tiles, long K): K is cut into ksplit slices that run as
 * separate workgroups, the fp32 partial tiles go to `workspace` (>= ksplit*M*N floats) and one pass sums them and applies
 * bias / act / gamma / residual.  Same result up to fp32 summation order.  M > 16, N % 8 == 0, no batch / GLU / window. */
int vg_gemm_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                   const float* gamma, const void* R, int64_t ldr, int M, int N, int K, int in_dtype, int out_dtype, int act,
                   int ksplit, float* workspace, int64_t ws_floats, vg_stream_t stream);
/* fp8 (OCP e4m3) prefill path of BASELINE config C4.
 * vg_quantize_fp8_rows: q[m,:] = e4m3(x[m,:] / scale[m]) with scale[m] = absmax(x[m,:]) / 448 (round to nearest even, what
 * torch's .to(float8_e4m3fn) does); x bf16 / fp32 [M,K], K % 4 == 0.
 * vg_gemm_f8: C[m,n] = scale_a[m] * scale_w[n] * sum_k A8[m,k] * W8[n,k]  (+ bias[n]) (+ R[m,n]), fp32 accumulation on
 * v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (twice the bf16 MFMA rate at half the operand bytes);
 * a_op = 1: W8 = [gate; up] rows, C[m,n] = silu(gate)*up like vg_gemm.  M > 16 (prefill only), K % 16 == 0. */
int vg_quantize_fp8_rows(const void* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int64_t M, int K, int dtype,
                         vg_stream_t stream);
int vg_gemm_f8(const uint8_t* A8, int64_t lda, const float* a_scale, const uint8_t* W8, int64_t ldw, const float* w_scale,
               void* C, int64_t ldc, const float* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K,
               int out_dtype, int a_op, vg_stream_t stream);
/* Which kernel an (M, N, K) GEMM of vg_gemm / vg_gemm_window is routed to with the current knobs (measurement aid:
 * bench.py attributes per-launch times to kernels with it): 0 gemm_skinny_kernel (M <= 16), 1 gemm_tile_glds_kernel,
 * 2 gemm_tile_k64b_kernel, 3 gemm_tile_w128_kernel, 4 gemm_tile_s128_kernel.  N = output columns (F for a_op == 1).
 * Launches nothing. */
int vg_gemm_route(int64_t M, int64_t N, int64_t K, int in_dtype, int a_op, int windowed);

/* ---- mask post-processing and evaluation counts: integer / byte work ---- */
/* Connected components of N binary images [N,H,W] (uint8, foreground = nonzero), connectivity 4 or 8.
 * labels: 0 on background, 1 + the smallest linear pixel index (y*W + x) of the component on foreground;
 * counts: area of the pixel's component, 0 on background.  Replaces _C.get_connected_componnets
 * (R/model/segment_anything_2/sam2/csrc/connected_components.cu:213-282, 8-connectivity; called from
 * R/model/segment_anything_2/sam2/utils/misc.py:47-63): same partition and areas; the reference's label VALUES are an
 * artefact of its 2x2-block union-find, only labels > 0 and the areas are consumed (misc.py:223-224).
 * Odd H / W are accepted (the reference asserts even sizes, connected_components.cu:226-227). */
int vg_connected_components(const uint8_t* mask, int32_t* labels, int32_t* counts, int N, int H, int W,
                            int connectivity, vg_stream_t stream);
/* out = mask with every 4-connected component of fewer than min_size pixels cleared (uint8 0/1).
 * remove_small_blobs, R/eval_gcg_infer.py:20-29 (skimage.morphology.remove_small_objects on a bool image).
 * ws_labels / ws_counts: int32 [N,H,W] scratch. */
int vg_remove_small_blobs(const uint8_t* mask, uint8_t* out, int32_t* ws_labels, int32_t* ws_counts, int N, int H,
                          int W, int min_size, vg_stream_t stream);
/* out = scores with every 8-connected background (score <= 0) component of area <= max_area set to 0.1
 * fill_holes_in_mask_scores, R/model/segment_anything_2/sam2/utils/misc.py:216-227 (switched off in the reference's
 * predictor, sam2_video_predictor.py:971-975: a flag on this side too). */
int vg_fill_holes(const float* scores, float* out, int32_t* ws_labels, int32_t* ws_counts, int N, int H, int W,
                  int max_area, vg_stream_t stream);
/* inter[p,g] = |a_p & b_g|, uni[p,g] = |a_p | b_g| over L pixels; a:[P,L], b:[G,L] uint8 (nonzero = set).
 * diagonal != 0 (P == G): only the pairs (i, i), outputs [P] — the per-frame Jaccard of one object.
 * compute_iou R/eval_gcg_metrics.py:26-37; db_eval_iou R/eval_referdavis_metrics.py:147-176 (no void pixels). */
int vg_mask_pair_counts(const uint8_t* a, const uint8_t* b, int64_t* inter, int64_t* uni, int P, int G, int64_t L,
                        int diagonal, vg_stream_t stream);
/* out[n] = {n_fg, n_gt, fg_match, gt_match}: pixels of the two 1-pixel boundary maps (_seg2bmap,
 * R/eval_referdavis_metrics.py:262-305, same-size case) and how many of each lie within the disk of the given radius
 * (skimage disk: dx^2 + dy^2 <= r^2; cv2.dilate border = ignore) of the other one — the integer part of f_measure,
 * R/eval_referdavis_metrics.py:194-259.  fg, gt: uint8 [N,H,W]. */
int vg_boundary_counts(const uint8_t* fg, const uint8_t* gt, int64_t* out, int N, int H, int W, int radius,
                       vg_stream_t stream);

/* ---- small MLP heads (r06) ----
 * vg_mlp3_grouped: G independent three-layer MLPs (Linear, ReLU, Linear, ReLU, Linear; sigmoid on the outputs of head g when bit g of sig_mask is set)
 * in one launch: SAM2's output_hypernetworks_mlps (G = 4), iou_prediction_head + pred_obj_score_head (G = 2) and obj_ptr_proj (G = 1) —
 * R/modeling/sam/mask_decoder.py:232-245, R/modeling/sam2_base.py:425-431, MLP = R/modeling/sam2_utils.py:108-132.  bf16 rows and weights.
 *   x: head g reads rows x + g * x_gs + r * x_rs (r < R; strides in elements, multiples of 8), K inputs each.
 *   w0, w1, w2: the heads' nn.Linear weights ([Hd, K], [Hd, Hd], [No, Hd] per head) stacked over the heads and PACKED in MFMA fragment order, so that a wave's
 *   load is one contiguous KiB: [G][ceil(out / 32)][in / 16][64][8] with element (g, t, s, lane, e) = W_g[32 t + lane % 32][16 s + 8 (lane / 32) + e], rows past
 *   `out` zero (videoglamm_amd/ops.py:mlp3_pack is the reference packing).  b0 [G, Hd], b1 [G, Hd], b2 [G, No]: fp32, plain.
 *   out: element (r, g, c) at out + r * o_rs + g * o_gs + c (c < No), out_dtype bf16 or fp32.  K, Hd multiples of 16 in [16, 256], No in [1, 256].
 * The activations between the layers are rounded to bf16, as vg_gemm's bf16 outputs are. */
int vg_mlp3_grouped(const void* x, int64_t x_rs, int64_t x_gs, const void* w0, const float* b0, const void* w1, const float* b1, const void* w2,
                    const float* b2, void* out, int64_t o_rs, int64_t o_gs, int out_dtype, int G, int R, int K, int Hd, int No, unsigned sig_mask,
                    vg_stream_t stream);

/* ---- image pre-processing: uint8 frames in HBM -> the three model inputs ---- */
/* One 8-bit pass of Pillow's separable resampler over in:[N,H,W,C] uint8 (C <= 4): axis 1 resizes W -> out_size
 * (out:[N,H,out_size,C]), axis 0 resizes H -> out_size (out:[N,out_size,W,C]).  bounds:[out_size,2] = (first input
 * index, tap count), coeffs:[out_size,ksize] = 22-bit fixed-point weights (Pillow src/libImaging/Resample.c:
 * precompute_coeffs + normalize_coeffs_8bpc; computed on the host by videoglamm_amd/preproc.py).
 * out = clip8((2^21 + sum_k in[first + k] * coeff[k]) >> 22): bit-exact with PIL.Image.resize — what
 * torchvision's resize(to_pil_image(x), size) runs at R/utils/sam_transforms.py:44-49 and the CLIP processor's
 * bicubic resize at R/utils/enc_preprocessors.py:120-166. */
int vg_resample_u8(const uint8_t* in, uint8_t* out, int N, int H, int W, int C, int out_size, int axis,
                   const int32_t* bounds, const int32_t* coeffs, int ksize, vg_stream_t stream);
/* OpenCV's cv2.resize(img, (Wo, Ho)) with the default INTER_LINEAR on N uint8 images [N,H,W,C] -> [N,Ho,Wo,C]: the InternVideo2
 * stream of the reference's pre-processing (R/model/videogpt_plus/model/internvideo/utils.py:124).  2 taps per axis in 11-bit
 * fixed point (imgproc/src/resize.cpp: HResizeLinear + VResizeLinear with FixedPtCast<int, uchar, 22>), no antialiasing.
 * xi / yi: int32 [Wo,2] / [Ho,2] source indices (already clamped), xa / yb: int32 [Wo,2] / [Ho,2] taps (sum 2048), computed by
 * the host in OpenCV's float32 arithmetic (videoglamm_amd/host.py:cv_linear_taps).  An exact 2x down-scale on both axes takes
 * the INTER_AREA fast path cv::resize re-routes it to ((a+b+c+d+2)>>2; tables may be NULL).  H == Ho && W == Wo is refused. */
int vg_resize_cv_linear_u8(const uint8_t* in, uint8_t* out, int N, int H, int W, int C, int Ho, int Wo, const int32_t* xi,
                           const int32_t* xa, const int32_t* yi, const int32_t* yb, vg_stream_t stream);

/* in:[N,H,W,3] uint8, crop (top,left,h,w) -> out:[N,3,h,w] planar.  mean / std: three doubles each on the HOST.
 * mode 0: fp32 (x - mean) / std on 0..255 values (SAM: R/utils/sam_transforms.py:50-55);
 * mode 1: (x / 255 - mean) / std evaluated in fp64, rounded once (the encoder processors' numpy arithmetic,
 * R/model/videogpt_plus/model/internvideo/utils.py:105-143, R/utils/enc_preprocessors.py:120-166). */
int vg_normalize_u8(const uint8_t* in, void* out, int N, int H, int W, int top, int left, int h, int w,
                    const double* mean, const double* std, int mode, int out_dtype, vg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VG_KERNELS_H */
