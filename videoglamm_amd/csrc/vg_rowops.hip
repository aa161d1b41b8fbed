// Row-wise kernels: LayerNorm, RMSNorm, argmax.  One wave (64 lanes) per row, 4 rows per workgroup;
// fp32 statistics, two-pass variance (mean first) like torch's native_layer_norm.
#include "vg_common.h"
#include <math.h>

template <typename TI, typename TO, bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const TI* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                   const float* __restrict__ b, TO* __restrict__ y, int64_t ldy,
                                                   int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + row * ldx;
  TO* yr = y + row * ldy;
  float mean = 0.f;
  if (!RMS) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += vg_elt<TI>::ld(xr + c);
    mean = wave_sum(s) / (float)C;
  }
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = vg_elt<TI>::ld(xr + c) - mean;
    v += d * d;
  }
  v = wave_sum(v) / (float)C;
  const float rstd = rsqrtf(v + eps);
  for (int c = lane; c < C; c += 64) {
    float o = (vg_elt<TI>::ld(xr + c) - mean) * rstd;
    if (RMS) {
      // reference casts the normalised value back to the input dtype before the weight multiply
      // (internvideo2.py:140-145, HF LlamaRMSNorm)
      if (sizeof(TI) == 2) o = bf2f(f2bf(o));
      o = w ? o * w[c] : o;
    } else {
      o = o * (w ? w[c] : 1.f) + (b ? b[c] : 0.f);
    }
    vg_elt<TO>::st(yr + c, o);
  }
}

template <bool RMS>
static int launch_norm(const void* x, int64_t ldx, const float* w, const float* b, void* y, int64_t ldy,
                       int64_t rows, int C, float eps, int in_dtype, int out_dtype, hipStream_t st) {
  if (rows == 0) return VG_OK;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (in_dtype == VG_F32 && out_dtype == VG_F32)
    norm_kernel<float, float, RMS><<<grid, 256, 0, st>>>((const float*)x, ldx, w, b, (float*)y, ldy, rows, C, eps);
  else if (in_dtype == VG_BF16 && out_dtype == VG_BF16)
    norm_kernel<bf16_t, bf16_t, RMS><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, w, b, (bf16_t*)y, ldy, rows, C, eps);
  else if (in_dtype == VG_BF16 && out_dtype == VG_F32)
    norm_kernel<bf16_t, float, RMS><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, w, b, (float*)y, ldy, rows, C, eps);
  else if (in_dtype == VG_F32 && out_dtype == VG_BF16)
    norm_kernel<float, bf16_t, RMS><<<grid, 256, 0, st>>>((const float*)x, ldx, w, b, (bf16_t*)y, ldy, rows, C, eps);
  else {
    vg_set_error("norm: bad dtypes %d -> %d", in_dtype, out_dtype);
    return VG_ERR_ARG;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_layernorm(const void* x, int64_t ldx, const float* w, const float* b, void* y, int64_t ldy,
                            int64_t rows, int C, float eps, int in_dtype, int out_dtype, vg_stream_t stream) {
  VG_CHECK(x && y && rows >= 0 && C > 0, VG_ERR_ARG, "vg_layernorm: bad args");
  return launch_norm<false>(x, ldx, w, b, y, ldy, rows, C, eps, in_dtype, out_dtype, (hipStream_t)stream);
}
extern "C" int vg_rmsnorm(const void* x, int64_t ldx, const float* w, void* y, int64_t ldy, int64_t rows, int C,
                          float eps, int in_dtype, int out_dtype, vg_stream_t stream) {
  VG_CHECK(x && y && rows >= 0 && C > 0, VG_ERR_ARG, "vg_rmsnorm: bad args");
  return launch_norm<true>(x, ldx, w, nullptr, y, ldy, rows, C, eps, in_dtype, out_dtype, (hipStream_t)stream);
}

// argmax: one 256-thread workgroup per row; ties resolve to the lowest index (torch.argmax on CPU).
__global__ __launch_bounds__(256) void argmax_kernel(const void* __restrict__ x, int n, int64_t* __restrict__ out, int dt) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int64_t row = blockIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = ld_any(x, row * n + i, dt);
    if (v > best || (v == best && i < bi) || bi == 0x7fffffff) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    out[row] = bi;
  }
}

extern "C" int vg_argmax(const void* x, int64_t rows, int n, int64_t* out, int dtype, vg_stream_t stream) {
  VG_CHECK(x && out && rows >= 0 && n > 0, VG_ERR_ARG, "vg_argmax: bad args");
  if (rows == 0) return VG_OK;
  argmax_kernel<<<dim3((unsigned)rows), 256, 0, (hipStream_t)stream>>>(x, n, out, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// Mask selection after the SAM2 mask decoder (one 256-thread workgroup per object).
//   mode 0 (multimask_output=False): dynamic multimask via stability — keep mask 0 when its stability
//          score |{m>delta}| / |{m>-delta}| >= thresh, else the best-IoU mask of tokens 1..3
//          (sam/mask_decoder.py:247-295); token out = token 0.
//   mode 1 (multimask_output=True): best-IoU of tokens 1..3 (sam2_base.py:376-386); token out = that token.
__global__ __launch_bounds__(256) void multimask_select_kernel(const float* __restrict__ masks, const float* __restrict__ ious,
                                                               const void* __restrict__ tokens, float* __restrict__ out_mask,
                                                               float* __restrict__ out_iou, void* __restrict__ out_token,
                                                               int* __restrict__ out_idx, int64_t HW, int C, float delta,
                                                               float thresh, int mode, int tok_dt) {
  __shared__ float s_i[4], s_u[4];
  __shared__ int s_sel;
  const int n = blockIdx.x;
  const float* m = masks + (int64_t)n * 4 * HW;
  const float* io = ious + n * 4;
  int best = 1;
  for (int k = 2; k < 4; ++k) if (io[k] > io[best]) best = k;  // argmax, first max wins
  int sel = best;
  if (mode == 0) {
    float ai = 0.f, au = 0.f;
    for (int64_t i = threadIdx.x; i < HW; i += 256) {
      const float v = m[i];
      ai += v > delta ? 1.f : 0.f;
      au += v > -delta ? 1.f : 0.f;
    }
    ai = wave_sum(ai); au = wave_sum(au);
    if ((threadIdx.x & 63) == 0) { s_i[threadIdx.x >> 6] = ai; s_u[threadIdx.x >> 6] = au; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float ti = s_i[0] + s_i[1] + s_i[2] + s_i[3], tu = s_u[0] + s_u[1] + s_u[2] + s_u[3];
      const float stab = tu > 0.f ? ti / tu : 1.0f;
      s_sel = stab >= thresh ? 0 : best;
    }
    __syncthreads();
    sel = s_sel;
  }
  const float* src = m + (int64_t)sel * HW;
  for (int64_t i = threadIdx.x; i < HW; i += 256) out_mask[(int64_t)n * HW + i] = src[i];
  const int tsel = mode == 0 ? 0 : sel;
  if (tokens && out_token)
    for (int c = threadIdx.x; c < C; c += 256)
      st_any(out_token, (int64_t)n * C + c, tok_dt, ld_any(tokens, ((int64_t)n * 4 + tsel) * C + c, tok_dt));
  if (threadIdx.x == 0) {
    out_iou[n] = io[sel];
    if (out_idx) out_idx[n] = sel;
  }
}

extern "C" int vg_multimask_select(const float* masks, const float* ious, const void* tokens, float* out_mask,
                                   float* out_iou, void* out_token, int* out_idx, int N, int64_t HW, int C, float delta,
                                   float thresh, int mode, int token_dtype, vg_stream_t stream) {
  VG_CHECK(masks && ious && out_mask && out_iou && N >= 0 && HW > 0, VG_ERR_ARG, "vg_multimask_select: bad args");
  if (N == 0) return VG_OK;
  multimask_select_kernel<<<dim3(N), 256, 0, (hipStream_t)stream>>>(masks, ious, tokens, out_mask, out_iou, out_token, out_idx,
                                                                     HW, C, delta, thresh, mode, token_dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
