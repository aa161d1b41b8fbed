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
