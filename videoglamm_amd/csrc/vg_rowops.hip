// Row-wise kernels: LayerNorm, RMSNorm, argmax.  One wave (64 lanes) per row, 4 rows per workgroup;
// fp32 statistics, two-pass variance (mean first) like torch's native_layer_norm.
#include "vg_common.h"
#include <math.h>
#include <stdlib.h>

template <typename T> struct RowVec;
template <> struct RowVec<bf16_t> {
  static constexpr int N = 8;
  typedef u32x4_t Raw;
  static __device__ __forceinline__ Raw ldraw(const bf16_t* p) { return *(const u32x4_t*)p; }
  static __device__ __forceinline__ void unpack(const Raw& r, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(r[e] << 16); v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    const u32x4_t r = *(const u32x4_t*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(r[e] << 16); v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
  }
};
template <> struct RowVec<float> {
  static constexpr int N = 4;
  typedef f32x4_t Raw;
  static __device__ __forceinline__ Raw ldraw(const float* p) { return *(const f32x4_t*)p; }
  static __device__ __forceinline__ void unpack(const Raw& r, float (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r[e];
  }
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const f32x4_t r = *(const f32x4_t*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r[e];
  }
};
template <typename TO, int N> __device__ __forceinline__ void row_store(TO* p, const float (&v)[N]) {
  if constexpr (sizeof(TO) == 2 && N == 8) {
    u32x4_t r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = f2bf2(v[2 * e], v[2 * e + 1]);
    *(u32x4_t*)p = r;
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) vg_elt<TO>::st(p + e, v[e]);
  }
}

// TPR threads cooperate on one row: 64 (one wave per row, 4 rows per workgroup) for tall inputs, 256 (whole
// workgroup per row) when there are few rows (LLM decode: 1 x 4096).  16-byte loads when VEC, scalar otherwise.
template <typename TI, typename TO, bool RMS, int TPR, bool VEC>
__global__ __launch_bounds__(256) void norm_kernel(const TI* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                   const float* __restrict__ b, TO* __restrict__ y, int64_t ldy,
                                                   int64_t rows, int C, float eps) {
  __shared__ float red[4];
  constexpr int RPB = 256 / TPR;
  constexpr int NV = RowVec<TI>::N;
  const int t = threadIdx.x % TPR;
  auto reduce = [&](float v) {
    if constexpr (TPR == 16) return group_sum<16>(v);
    v = wave_sum(v);
    if constexpr (TPR == 256) {
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      v = red[0] + red[1] + red[2] + red[3];
      __syncthreads();
    }
    return v;
  };
  if constexpr (VEC) {
    if (C <= TPR * NV * 4) {
      // The whole row fits in registers (<= 4 x 16-byte chunks per lane): ONE global read, two in-register passes.
      // Persistent over row groups (grid-stride) with the NEXT group's row requested before the current one is
      // reduced: a one-row-per-wave launch is latency-bound (load -> two dependent reductions -> store) at ~1 TB/s.
      typedef typename RowVec<TI>::Raw Raw;
      const int64_t ngroups = (rows + RPB - 1) / RPB;
      const int sub = threadIdx.x / TPR;
      auto rowload = [&](int64_t grp, Raw (&r)[4]) {
        const int64_t rw = min(grp * RPB + sub, rows - 1);   // clamped: the load is unconditional
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (t + j * TPR) * NV;
          r[j] = RowVec<TI>::ldraw(x + rw * ldx + (c < C ? c : 0));
        }
      };
      // the lane's columns are the same for every row it visits: weight / bias live in registers across the row loop
      // (per-element scalar loads of w and b were 16 of the 17 load instructions per 16 bytes of x: ~1 TB/s)
      float wr[4][NV], br[4][NV];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (t + j * TPR) * NV;
#pragma unroll
        for (int e4 = 0; e4 < NV; e4 += 4) {
          const f32x4_t wv = (w && c < C) ? *(const f32x4_t*)(w + c + e4) : f32x4_t{1.f, 1.f, 1.f, 1.f};
          const f32x4_t bv = (b && c < C) ? *(const f32x4_t*)(b + c + e4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) { wr[j][e4 + e] = wv[e]; br[j][e4 + e] = bv[e]; }
        }
      }
      Raw cur[4], nxt[4];
      rowload(blockIdx.x, cur);
      for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        rowload(min(grp + (int64_t)gridDim.x, ngroups - 1), nxt);
        const int64_t row = grp * RPB + sub;
        const bool live = row < rows;          // TPR == 256: block-uniform; 64 / 16: uniform over the lanes that reduce together
        float v[4][NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (t + j * TPR) * NV;
          RowVec<TI>::unpack(cur[j], v[j]);
          if (c < C) {
#pragma unroll
            for (int e = 0; e < NV; ++e) s += v[j][e];
          }
        }
        const float mean_r = RMS ? 0.f : reduce(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (t + j * TPR) * NV;
          if (c < C) {
#pragma unroll
            for (int e = 0; e < NV; ++e) { const float d = v[j][e] - mean_r; q += d * d; }
          }
        }
        const float rstd_r = rsqrtf(reduce(q) / (float)C + eps);
        TO* yr = y + row * ldy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (t + j * TPR) * NV;
          if (live && c < C) {
            float o[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) {
              float n = (v[j][e] - mean_r) * rstd_r;
              if (RMS) {
                if (sizeof(TI) == 2) n = bf2f(f2bf(n));
                o[e] = w ? n * wr[j][e] : n;
              } else {
                o[e] = n * wr[j][e] + br[j][e];
              }
            }
            row_store<TO, NV>(yr + c, o);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
      }
      return;
    }
  }
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;
  if (row >= rows) return;  // TPR == 256: block-uniform; TPR == 64: wave-uniform and no block barrier is used
  const TI* xr = x + row * ldx;
  TO* yr = y + row * ldy;
  float mean = 0.f;
  if (!RMS) {
    float s = 0.f;
    if constexpr (VEC) {
      for (int c = t * NV; c < C; c += TPR * NV) { float v[NV]; RowVec<TI>::ld(xr + c, v);
#pragma unroll
        for (int e = 0; e < NV; ++e) s += v[e]; }
    } else {
      for (int c = t; c < C; c += TPR) s += vg_elt<TI>::ld(xr + c);
    }
    mean = reduce(s) / (float)C;
  }
  float q = 0.f;
  if constexpr (VEC) {
    for (int c = t * NV; c < C; c += TPR * NV) { float v[NV]; RowVec<TI>::ld(xr + c, v);
#pragma unroll
      for (int e = 0; e < NV; ++e) { const float d = v[e] - mean; q += d * d; } }
  } else {
    for (int c = t; c < C; c += TPR) { const float d = vg_elt<TI>::ld(xr + c) - mean; q += d * d; }
  }
  const float rstd = rsqrtf(reduce(q) / (float)C + eps);
  auto fin = [&](float xv, int c) {
    float o = (xv - mean) * rstd;
    if (RMS) {
      // reference casts the normalised value back to the input dtype before the weight multiply
      // (internvideo2.py:140-145, HF LlamaRMSNorm)
      if (sizeof(TI) == 2) o = bf2f(f2bf(o));
      return w ? o * w[c] : o;
    }
    return o * (w ? w[c] : 1.f) + (b ? b[c] : 0.f);
  };
  if constexpr (VEC) {
    for (int c = t * NV; c < C; c += TPR * NV) {
      float v[NV];
      RowVec<TI>::ld(xr + c, v);
#pragma unroll
      for (int e = 0; e < NV; ++e) v[e] = fin(v[e], c + e);
      row_store<TO, NV>(yr + c, v);
    }
  } else {
    for (int c = t; c < C; c += TPR) vg_elt<TO>::st(yr + c, fin(vg_elt<TI>::ld(xr + c), c));
  }
}

// Short bf16 rows (Hiera's LayerNorms: C = 144 / 288 / 576 / 1152, 65536 - 1M rows per launch).  TPR lanes own a row and EVERY lane
// carries NJ 16-byte chunks of it (576 = 8 lanes x 9 chunks: no masked lanes; the one-wave-per-row kernel above issues a second,
// 12 %-used load per row there and holds weight / bias in 64 registers: 1.8 TB/s on stage 3).  Weight and bias live in LDS as fp32,
// the raw row stays packed in registers (unpacked on the fly in each of the three passes), the next row group is requested before
// the current one is reduced.  Same two-pass fp32 statistics as norm_kernel.
template <bool RMS, int TPR, int NJ>
__global__ __launch_bounds__(256) void norm_short_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, bf16_t* __restrict__ y, int64_t ldy,
                                                         int64_t rows, int C, float eps) {
  extern __shared__ __attribute__((aligned(16))) float nwb[];      // gamma[C] | beta[C]
  constexpr int RPB = 256 / TPR;
  for (int i = threadIdx.x; i < C; i += 256) { nwb[i] = w ? w[i] : 1.f; nwb[C + i] = b ? b[i] : 0.f; }
  __syncthreads();
  const int t = threadIdx.x % TPR, sub = threadIdx.x / TPR;
  const int64_t ngroups = (rows + RPB - 1) / RPB;
  auto reduce = [&](float v) { return group_sum<TPR>(v); };
  auto rowload = [&](int64_t grp, u32x4_t (&r)[NJ]) {
    const int64_t rw = min(grp * RPB + sub, rows - 1);       // clamped: the load is unconditional
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = (t + j * TPR) * 8;
      r[j] = *(const u32x4_t*)(x + rw * ldx + (c < C ? c : 0));
    }
  };
  u32x4_t cur[NJ], nxt[NJ];
  rowload(blockIdx.x, cur);
  const float invC = 1.0f / (float)C;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    rowload(min(grp + (int64_t)gridDim.x, ngroups - 1), nxt);
    const int64_t row = grp * RPB + sub;
    float s = 0.f;
    if (!RMS) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if ((t + j * TPR) * 8 < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s += __uint_as_float(cur[j][e] << 16) + __uint_as_float(cur[j][e] & 0xffff0000u);
        }
    }
    const float mean = RMS ? 0.f : reduce(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if ((t + j * TPR) * 8 < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = __uint_as_float(cur[j][e] << 16) - mean, d1 = __uint_as_float(cur[j][e] & 0xffff0000u) - mean;
          q += d0 * d0 + d1 * d1;
        }
      }
    const float rstd = rsqrtf(reduce(q) * invC + eps);
    if (row < rows) {
      bf16_t* yr = y + row * ldy;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = (t + j * TPR) * 8;
        if (c < C) {
          const f32x4_t w0 = *(const f32x4_t*)(nwb + c), w1 = *(const f32x4_t*)(nwb + c + 4);
          const f32x4_t b0 = *(const f32x4_t*)(nwb + C + c), b1 = *(const f32x4_t*)(nwb + C + c + 4);
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float n0 = (__uint_as_float(cur[j][e] << 16) - mean) * rstd, n1 = (__uint_as_float(cur[j][e] & 0xffff0000u) - mean) * rstd;
            if (RMS) { n0 = bf2f(f2bf(n0)); n1 = bf2f(f2bf(n1)); }       // the reference rounds the normalised value before the weight
            const float g0 = e < 2 ? w0[2 * e] : w1[2 * e - 4], g1 = e < 2 ? w0[2 * e + 1] : w1[2 * e - 3];
            const float a0 = e < 2 ? b0[2 * e] : b1[2 * e - 4], a1 = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
            o[2 * e] = RMS ? n0 * g0 : n0 * g0 + a0;
            o[2 * e + 1] = RMS ? n1 * g1 : n1 * g1 + a1;
          }
          row_store<bf16_t, 8>(yr + c, o);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) cur[j] = nxt[j];
  }
}

template <bool RMS, int TPR, int NJ>
static void launch_norm_short(const bf16_t* x, int64_t ldx, const float* w, const float* b, bf16_t* y, int64_t ldy, int64_t rows, int C,
                              float eps, hipStream_t st) {
  const int64_t ngroups = (rows + 256 / TPR - 1) / (256 / TPR);
  const int64_t cap = 256 * (NJ >= 9 ? 4 : 8);          // resident workgroups per CU at the kernel's register footprint
  const unsigned grid = (unsigned)(ngroups < cap ? ngroups : cap);
  norm_short_kernel<RMS, TPR, NJ><<<grid, 256, 2 * C * sizeof(float), st>>>(x, ldx, w, b, y, ldy, rows, C, eps);
}

template <typename TI, typename TO, bool RMS>
static void launch_norm_t(const void* x, int64_t ldx, const float* w, const float* b, void* y, int64_t ldy, int64_t rows,
                          int C, float eps, hipStream_t st) {
  constexpr int NV = RowVec<TI>::N;
  const bool vec = (C % NV == 0) && (ldx % NV == 0) && (ldy % NV == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) % 16 == 0);
  const TI* xi = (const TI*)x;
  TO* yo = (TO*)y;
  const unsigned pgrid = 256 * 8;   // persistent cap of the register-resident path: 8 workgroups per CU
  if constexpr (sizeof(TI) == 2 && sizeof(TO) == 2) {
    if (vec && rows >= 1024 && (((uintptr_t)w | (uintptr_t)b) % 16 == 0)) {
      // tower / LLM-prefill widths with a few thousand rows: norm_kernel's per-workgroup prologue (64 weight / bias registers per lane, loaded
      // before the first row) outweighs the 4 rows a workgroup then normalises — 23 MB in 42 us at InternVideo2's 4100 x 1408 (r02 trace)
      const bf16_t* xs = (const bf16_t*)x;
      bf16_t* ys = (bf16_t*)y;
      if (C == 1408) return launch_norm_short<RMS, 64, 3>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
      if (C == 1024) return launch_norm_short<RMS, 64, 2>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
      if (C == 4096) return launch_norm_short<RMS, 64, 8>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
      if (C == 3072) return launch_norm_short<RMS, 64, 6>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
    }
    if (vec && rows >= 4096 && (((uintptr_t)w | (uintptr_t)b) % 16 == 0)) {      // Hiera's LayerNorm widths, every lane loaded
      const bf16_t* xs = (const bf16_t*)x;
      bf16_t* ys = (bf16_t*)y;
      if (C == 576) return launch_norm_short<RMS, 8, 9>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
      if (C == 1152) return launch_norm_short<RMS, 16, 9>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
      if (C == 288) return launch_norm_short<RMS, 8, 5>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
      if (C == 144) return launch_norm_short<RMS, 8, 3>(xs, ldx, w, b, ys, ldy, rows, C, eps, st);
    }
  }
  if (rows < 1024 || (vec && C > 64 * NV * 4)) {   // few rows, or rows too long for one wave's registers
    dim3 grid((unsigned)rows);
    if (vec && C <= 256 * NV * 4 && grid.x > pgrid) grid.x = pgrid;
    if (vec) norm_kernel<TI, TO, RMS, 256, true><<<grid, 256, 0, st>>>(xi, ldx, w, b, yo, ldy, rows, C, eps);
    else norm_kernel<TI, TO, RMS, 256, false><<<grid, 256, 0, st>>>(xi, ldx, w, b, yo, ldy, rows, C, eps);
  } else if (vec && C <= 16 * NV * 4) {   // short rows (Hiera stage 1-2: C = 144 / 288 in bf16): 16 lanes per row, 16 rows per workgroup.
    // Only rows that fit the 16-lane register-resident path come here: that path is persistent, the fall-back below it is one
    // row group per workgroup and must never see a capped grid (r02: fp32 rows of 256 < C <= 512 were capped at 32768 rows)
    dim3 grid((unsigned)((rows + 15) / 16 < (int64_t)pgrid ? (rows + 15) / 16 : (int64_t)pgrid));
    norm_kernel<TI, TO, RMS, 16, true><<<grid, 256, 0, st>>>(xi, ldx, w, b, yo, ldy, rows, C, eps);
  } else {
    dim3 grid((unsigned)((rows + 3) / 4));
    if (vec && grid.x > pgrid) grid.x = pgrid;
    if (vec) norm_kernel<TI, TO, RMS, 64, true><<<grid, 256, 0, st>>>(xi, ldx, w, b, yo, ldy, rows, C, eps);
    else norm_kernel<TI, TO, RMS, 64, false><<<grid, 256, 0, st>>>(xi, ldx, w, b, yo, ldy, rows, C, eps);
  }
}

template <bool RMS>
static int launch_norm(const void* x, int64_t ldx, const float* w, const float* b, void* y, int64_t ldy,
                       int64_t rows, int C, float eps, int in_dtype, int out_dtype, hipStream_t st) {
  if (rows == 0) return VG_OK;
  if (in_dtype == VG_F32 && out_dtype == VG_F32) launch_norm_t<float, float, RMS>(x, ldx, w, b, y, ldy, rows, C, eps, st);
  else if (in_dtype == VG_BF16 && out_dtype == VG_BF16) launch_norm_t<bf16_t, bf16_t, RMS>(x, ldx, w, b, y, ldy, rows, C, eps, st);
  else if (in_dtype == VG_BF16 && out_dtype == VG_F32) launch_norm_t<bf16_t, float, RMS>(x, ldx, w, b, y, ldy, rows, C, eps, st);
  else if (in_dtype == VG_F32 && out_dtype == VG_BF16) launch_norm_t<float, bf16_t, RMS>(x, ldx, w, b, y, ldy, rows, C, eps, st);
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

// argmax over long rows (lm_head logits, 128k entries): up to 64 workgroups per row atomicMax a packed
// (order-preserving value bits, ~index) key into out[row] (zeroed first), then the key is decoded in place.
// Ties -> lowest index, NaN-free inputs assumed (torch.argmax semantics on finite logits).
__device__ __forceinline__ uint64_t amax_key(float v, int i) {
  uint32_t u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone map of float order onto unsigned order
  return ((uint64_t)u << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
}
__global__ __launch_bounds__(256) void argmax_stage1(const void* __restrict__ x, int n, unsigned long long* __restrict__ acc, int nb, int dt) {
  __shared__ uint64_t sk[4];
  const int64_t row = blockIdx.y;
  uint64_t best = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nb * 256) {
    const uint64_t k = amax_key(ld_any(x, row * n + i, dt), i);
    best = k > best ? k : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t ok = __shfl_xor(best, o, 64);
    best = ok > best ? ok : best;
  }
  if ((threadIdx.x & 63) == 0) sk[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) best = sk[w] > best ? sk[w] : best;
    atomicMax(acc + row, (unsigned long long)best);   // out[] doubles as the per-row key accumulator
  }
}
__global__ __launch_bounds__(256) void argmax_stage2(int64_t* __restrict__ out, int64_t rows) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r < rows) out[r] = (int64_t)(0xffffffffu - (uint32_t)(((uint64_t)out[r]) & 0xffffffffu));
}

extern "C" int vg_argmax(const void* x, int64_t rows, int n, int64_t* out, int dtype, vg_stream_t stream) {
  VG_CHECK(x && out && rows >= 0 && n > 0, VG_ERR_ARG, "vg_argmax: bad args");
  if (rows == 0) return VG_OK;
  int nb = (n + 2047) / 2048;
  if (nb > 64) nb = 64;
  if (nb < 1) nb = 1;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, (size_t)rows * sizeof(int64_t), st) != hipSuccess) {
    vg_set_error("vg_argmax: memset failed");
    return VG_ERR_LAUNCH;
  }
  argmax_stage1<<<dim3(nb, (unsigned)rows), 256, 0, st>>>(x, n, (unsigned long long*)out, nb, dtype);
  argmax_stage2<<<dim3((unsigned)((rows + 255) / 256)), 256, 0, st>>>(out, rows);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// vg_argmax's first stage alone (r06, the captured decode step): acc[row] must be zero on entry (vg_decode_step_end decodes the key and leaves it zero)
extern "C" int vg_argmax_partial(const void* x, int64_t rows, int n, uint64_t* acc, int dtype, vg_stream_t stream) {
  VG_CHECK(x && acc && rows > 0 && n > 0, VG_ERR_ARG, "vg_argmax_partial: bad args");
  int nb = (n + 2047) / 2048;
  if (nb > 64) nb = 64;
  argmax_stage1<<<dim3(nb, (unsigned)rows), 256, 0, (hipStream_t)stream>>>(x, n, (unsigned long long*)acc, nb, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// Mask selection after the SAM2 mask decoder (one 256-thread workgroup per object).
//   mode 0 (multimask_output=False): dynamic multimask via stability — keep mask 0 when its stability
//          score |{m>delta}| / |{m>-delta}| >= thresh, else the best-IoU mask of tokens 1..3
//          (sam/mask_decoder.py:247-295); token out = token 0.
//   mode 1 (multimask_output=True): best-IoU of tokens 1..3 (sam2_base.py:376-386); token out = that token.
// (r06: 1024 threads and 16-byte accesses per workgroup — the 256-thread scalar version took 207 us per 128 instances at C4's clip size, 0.8 of the mask
// decoder's 11.6 ms; counts are exact in fp32 up to 2^24 pixels, so the order of the partial sums does not matter)
__global__ __launch_bounds__(1024) void multimask_select_kernel(const float* __restrict__ masks, const float* __restrict__ ious,
                                                                const void* __restrict__ tokens, float* __restrict__ out_mask,
                                                                float* __restrict__ out_iou, void* __restrict__ out_token,
                                                                int* __restrict__ out_idx, int64_t HW, int C, float delta,
                                                                float thresh, int mode, int tok_dt) {
  __shared__ float s_i[16], s_u[16];
  __shared__ int s_sel;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* m = masks + (int64_t)n * 4 * HW;
  const float* io = ious + n * 4;
  int best = 1;
  for (int k = 2; k < 4; ++k) if (io[k] > io[best]) best = k;  // argmax, first max wins
  int sel = best;
  const bool vec = (HW & 3) == 0 && ((((uintptr_t)masks) | ((uintptr_t)out_mask)) & 15) == 0;
  if (mode == 0) {
    float ai = 0.f, au = 0.f;
    if (vec) {
      const f32x4_t* m4 = (const f32x4_t*)m;
      for (int64_t i = tid; i < HW / 4; i += 1024) {
        const f32x4_t v = m4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { ai += v[e] > delta ? 1.f : 0.f; au += v[e] > -delta ? 1.f : 0.f; }
      }
    } else {
      for (int64_t i = tid; i < HW; i += 1024) {
        const float v = m[i];
        ai += v > delta ? 1.f : 0.f;
        au += v > -delta ? 1.f : 0.f;
      }
    }
    ai = wave_sum(ai); au = wave_sum(au);
    if ((tid & 63) == 0) { s_i[tid >> 6] = ai; s_u[tid >> 6] = au; }
    __syncthreads();
    if (tid == 0) {
      float ti = 0.f, tu = 0.f;
      for (int w = 0; w < 16; ++w) { ti += s_i[w]; tu += s_u[w]; }
      const float stab = tu > 0.f ? ti / tu : 1.0f;
      s_sel = stab >= thresh ? 0 : best;
    }
    __syncthreads();
    sel = s_sel;
  }
  const float* src = m + (int64_t)sel * HW;
  if (vec) {
    const f32x4_t* s4 = (const f32x4_t*)src;
    f32x4_t* d4 = (f32x4_t*)(out_mask + (int64_t)n * HW);
    for (int64_t i = tid; i < HW / 4; i += 1024) d4[i] = s4[i];
  } else {
    for (int64_t i = tid; i < HW; i += 1024) out_mask[(int64_t)n * HW + i] = src[i];
  }
  const int tsel = mode == 0 ? 0 : sel;
  if (tokens && out_token)
    for (int c = tid; c < C; c += 1024)
      st_any(out_token, (int64_t)n * C + c, tok_dt, ld_any(tokens, ((int64_t)n * 4 + tsel) * C + c, tok_dt));
  if (tid == 0) {
    out_iou[n] = io[sel];
    if (out_idx) out_idx[n] = sel;
  }
}

// mode 1 (multimask_output=True: the best-IoU token, no stability pass — the SAM heads of the video branch): nothing to reduce, so the selected
// mask is copied by many workgroups per object with 16-byte accesses (r05: one workgroup per object took 62 us for a 256 KB mask — a tracked
// frame's slowest small kernel)
__global__ __launch_bounds__(256) void multimask_best_kernel(const float* __restrict__ masks, const float* __restrict__ ious,
                                                             const void* __restrict__ tokens, float* __restrict__ out_mask,
                                                             float* __restrict__ out_iou, void* __restrict__ out_token,
                                                             int* __restrict__ out_idx, int64_t HW, int C, int tok_dt) {
  const int n = blockIdx.y;
  const float* io = ious + n * 4;
  int sel = 1;
  for (int k = 2; k < 4; ++k) if (io[k] > io[sel]) sel = k;  // argmax, first max wins
  const f32x4_t* src = (const f32x4_t*)(masks + ((int64_t)n * 4 + sel) * HW);
  f32x4_t* dst = (f32x4_t*)(out_mask + (int64_t)n * HW);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
  if (blockIdx.x == 0) {
    if (tokens && out_token)
      for (int c = threadIdx.x; c < C; c += 256)
        st_any(out_token, (int64_t)n * C + c, tok_dt, ld_any(tokens, ((int64_t)n * 4 + sel) * C + c, tok_dt));
    if (threadIdx.x == 0) {
      out_iou[n] = io[sel];
      if (out_idx) out_idx[n] = sel;
    }
  }
}

extern "C" int vg_multimask_select(const float* masks, const float* ious, const void* tokens, float* out_mask,
                                   float* out_iou, void* out_token, int* out_idx, int N, int64_t HW, int C, float delta,
                                   float thresh, int mode, int token_dtype, vg_stream_t stream) {
  VG_CHECK(masks && ious && out_mask && out_iou && N >= 0 && HW > 0, VG_ERR_ARG, "vg_multimask_select: bad args");
  if (N == 0) return VG_OK;
  if (mode == 1 && HW % 4 == 0 && ((uintptr_t)masks & 15) == 0 && ((uintptr_t)out_mask & 15) == 0) {
    const int nb = (int)((HW / 4 + 255) / 256);
    multimask_best_kernel<<<dim3(nb < 64 ? nb : 64, N), 256, 0, (hipStream_t)stream>>>(masks, ious, tokens, out_mask, out_iou, out_token, out_idx, HW, C, token_dtype);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  multimask_select_kernel<<<dim3(N), 1024, 0, (hipStream_t)stream>>>(masks, ious, tokens, out_mask, out_iou, out_token, out_idx,
                                                                     HW, C, delta, thresh, mode, token_dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
