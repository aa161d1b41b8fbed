// Single-token decode step of the LLM: one new row against the KV cache and against every
// weight matrix.  The step is HBM-bound (all weights stream once per token) and latency-bound (a chain of ~9 dependent
// launches per layer, each paying ~4-5 us of ramp on top of its bytes), so the kernels here exist to cut launches:
//   decode_gemv_kernel  : [RMSNorm ->] x.W^T [-> SwiGLU] [+ residual]   (norm and GLU fused into the GEMV)
//   decode_attn_kernel  : RoPE(q,k) + KV-cache append + split-KV attention + merge of the splits by the last
//                         workgroup to finish (no separate rope / combine launches)
// Arithmetic mirrors the stand-alone kernels (vg_rmsnorm, vg_gemm skinny path, vg_rope_kv_append) step for step.
#include "vg_common.h"
#include <type_traits>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ void dec_unpack(const u32x4_t& v, float* f);
template <> __device__ __forceinline__ void dec_unpack<float>(const u32x4_t& v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(v[e]);
}
template <> __device__ __forceinline__ void dec_unpack<bf16_t>(const u32x4_t& v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(v[e] << 16); f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ u32x4_t dec_pack(const float* f);
template <> __device__ __forceinline__ u32x4_t dec_pack<float>(const float* f) {
  u32x4_t v = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
  return v;
}
template <> __device__ __forceinline__ u32x4_t dec_pack<bf16_t>(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = f2bf2(f[2 * e], f[2 * e + 1]);
  return v;
}
template <typename T> __device__ __forceinline__ float dec_dot(const u32x4_t& a, const u32x4_t& b);
template <> __device__ __forceinline__ float dec_dot<float>(const u32x4_t& a, const u32x4_t& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s = fmaf(__uint_as_float(a[e]), __uint_as_float(b[e]), s);
  return s;
}
template <> __device__ __forceinline__ float dec_dot<bf16_t>(const u32x4_t& a, const u32x4_t& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16), s);
    s = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u), s);
  }
  return s;
}
// 16 fp8 (OCP e4m3) weights of one 16-byte chunk against the 16 bf16 activations of two x chunks
__device__ __forceinline__ float dec_dot_w8(const u32x4_t& w, const u32x4_t& x0, const u32x4_t& x1) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[e], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[e], true);
    const uint32_t xa = e < 2 ? x0[2 * e] : x1[2 * e - 4], xb = e < 2 ? x0[2 * e + 1] : x1[2 * e - 3];
    s = fmaf(lo[0], __uint_as_float(xa << 16), s);
    s = fmaf(lo[1], __uint_as_float(xa & 0xffff0000u), s);
    s = fmaf(hi[0], __uint_as_float(xb << 16), s);
    s = fmaf(hi[1], __uint_as_float(xb & 0xffff0000u), s);
  }
  return s;
}
template <typename T> __device__ __forceinline__ float dec_round(float v) { return v; }
template <> __device__ __forceinline__ float dec_round<bf16_t>(float v) { return bf2f(f2bf(v)); }

// ---------------------------------------------------------------------------------------------------------------
// GEMV.  x (one row, K elements) is staged once per workgroup into LDS — normalised on the way when norm_w is given
// (HF LlamaRMSNorm: fp32 mean of squares, x*rstd cast to the activation dtype, then * weight) — and every wave streams
// PAIRS of weight rows against it: rows (2i, 2i+1), or (i, N+i) = gate|up of output i when GLU.
struct DecGemvArgs {
  const void* x; const void* W; void* y; const float* nw; const void* R;
  int N, K; int64_t ldw; float eps; int ppw;
  const float* wscale;   // fp8 weights only: one fp32 scale per weight row
  // RoPE epilogue (vg_decode_qkv_rope): the rows are the fused q|k|v projection of the new token; pairs are (d, d + D/2) of one head, rotated in
  // registers with the CURRENT position's cos / sin row (rope_cs: [2][D/2] fp32 at a fixed address, refreshed by vg_decode_advance), q heads go to
  // y, the new key / value rows straight into the caches at *pos_dev
  const float* rope_cs; const int* pos_dev; void* kc; void* vc; int H, Hkv;
};

template <typename T, typename TO, bool GLU>
__device__ __forceinline__ void dec_gemv_store(const DecGemvArgs& p, int pi, float a0, float a1) {
  const int n0 = GLU ? pi : 2 * pi, n1 = GLU ? p.N + pi : 2 * pi + 1;
  if (p.wscale) {        // fp8 weights: the row scales leave the dot products
    a0 *= p.wscale[n0];
    a1 *= p.wscale[GLU ? n1 : min(n1, p.N - 1)];
  }
  TO* y = (TO*)p.y;
  const TO* R = (const TO*)p.R;
  if constexpr (GLU) {
    float g = a0, u = a1;
    if (sizeof(T) == 2) { g = bf2f(f2bf(g)); u = bf2f(f2bf(u)); }   // gate/up projections materialise in bf16 in HF
    g = vg_silu(g);
    if (sizeof(T) == 2) g = bf2f(f2bf(g));
    float v = g * u;
    if (R) v += vg_elt<TO>::ld(R + n0);
    vg_elt<TO>::st(y + n0, v);
  } else {
    float v = a0;
    if (R) v += vg_elt<TO>::ld(R + n0);
    vg_elt<TO>::st(y + n0, v);
    if (n1 < p.N) {
      v = a1;
      if (R) v += vg_elt<TO>::ld(R + n1);
      vg_elt<TO>::st(y + n1, v);
    }
  }
}

// Fast path, K = a whole number of batches of chunks.  A wave owns `ppw` consecutive pairs and walks them as one flat sequence of
// batches (2 rows x 4 chunks per lane), software-pipelined two batches deep = 16 x 16-byte loads in flight per lane.
// Everything the kernel must wait for is requested up front, oldest-needed first (vmcnt retires in order): x, the norm
// weights, then both pipeline stages of W — the norm runs underneath the HBM round trip.  The loop body is branch-free
// and issues no stores: results go to LDS and are written out (with the residual / SwiGLU epilogue) once at the end,
// because on gfx9 a pending store makes every later vmcnt wait a full drain.
constexpr int DEC_MAX_PPW = 64;

// NB batches per row pair, CPB 16-byte chunks per lane per row per batch: K = NB * CPB * 64 chunks (CPB = 4 unless K only
// divides by 128 chunks, e.g. Phi-3's hidden 3072 = 3 x 2 x 64 x 8).
// W8: the weights are fp8 (OCP e4m3, one byte each) with a per-row scale; a weight chunk then covers 16 K elements = TWO
// x chunks, and NB x CPB x 64 counts weight chunks (K = 16 x that).
// RHD = D/2 > 0: the RoPE form above (pairs (d, d + RHD) of a head instead of adjacent rows)
template <typename T, typename TO, bool GLU, int NB, int CPB = 4, bool W8 = false, int RHD = 0>
__global__ __launch_bounds__(256) void decode_gemv_fast_kernel(DecGemvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char dec_smem[];
  __shared__ float red[4];
  __shared__ float res[4][DEC_MAX_PPW][2];
  __shared__ float rope_lds[RHD > 0 ? 2 * RHD : 1];
  static_assert(RHD == 0 || (!GLU && !W8 && 2 * RHD <= 256), "RoPE form: plain bf16 / fp32 rows");
  constexpr int KPC = 16 / sizeof(T);
  constexpr int NWV = KPC / 4;     // float4 loads of norm weight per chunk
  static_assert(!W8 || sizeof(T) == 2, "fp8 weights pair with bf16 activations");
  constexpr int NCH = NB * CPB * 64 * (W8 ? 2 : 1);   // 16-byte chunks of x (with W8 a weight row has half as many)
  constexpr int XN = (NCH + 255) / 256;          // chunks of x per thread (the last one may be partial: NCH % 256 == 128)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t* xs = (u32x4_t*)dec_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npair = GLU ? p.N : (p.N + 1) / 2;
  const int gw = blockIdx.x * 4 + wave;
  typedef typename std::conditional<W8, uint8_t, T>::type WT;
  const WT* W = (const WT*)p.W;
  const int p0 = gw * p.ppw;
  const int np = max(min(p0 + p.ppw, npair) - p0, 0);     // pairs of this wave
  const int total = np * NB;

  // ---- 1. every load up front
  u32x4_t xr[XN];
#pragma unroll
  for (int i = 0; i < XN; ++i) xr[i] = ((const u32x4_t*)p.x)[min(tid + 256 * i, NCH - 1)];
  f32x4_t nwr[XN][NWV];
  if (p.nw) {
#pragma unroll
    for (int i = 0; i < XN; ++i)
#pragma unroll
      for (int j = 0; j < NWV; ++j) nwr[i][j] = ((const f32x4_t*)p.nw)[min(tid + 256 * i, NCH - 1) * NWV + j];
  }
  float ropev = 0.f;               // cos (tid < RHD) / sin (RHD <= tid < 2 RHD) of the current position: requested BEFORE the weights, so the wait below costs the stream nothing
  if constexpr (RHD > 0) ropev = p.rope_cs[min(tid, 2 * RHD - 1)];
  int ipi = p0, icb = 0;           // issue cursor (pair, batch within the pair)
  u32x4_t va0[CPB], va1[CPB], vb0[CPB], vb1[CPB];
  auto issue = [&](u32x4_t (&v0)[CPB], u32x4_t (&v1)[CPB]) {
    const int pc = min(ipi, npair - 1);                      // clamped: the two prologue issues are unconditional
    const int n0 = RHD > 0 ? (pc / max(RHD, 1)) * (2 * RHD) + pc % max(RHD, 1) : GLU ? pc : 2 * pc;
    const int n1 = RHD > 0 ? n0 + RHD : GLU ? p.N + pc : min(2 * pc + 1, p.N - 1);
    const u32x4_t* w0 = (const u32x4_t*)(W + (int64_t)n0 * p.ldw) + icb * (64 * CPB) + lane;
    const u32x4_t* w1 = (const u32x4_t*)(W + (int64_t)n1 * p.ldw) + icb * (64 * CPB) + lane;
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      v0[u] = __builtin_nontemporal_load(w0 + u * 64);
      v1[u] = __builtin_nontemporal_load(w1 + u * 64);
    }
    if (++icb == NB) { icb = 0; ++ipi; }
  };
  issue(va0, va1);
  issue(vb0, vb1);

  // ---- 2. stage x (normalised) into LDS
  {
    float rstd = 1.f;
    if (p.nw) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        float f[KPC];
        dec_unpack<T>(xr[i], f);
        if (tid + 256 * i < NCH) {
#pragma unroll
          for (int e = 0; e < KPC; ++e) ss += f[e] * f[e];
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) red[wave] = ss;
      __syncthreads();
      rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)p.K + p.eps);
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        float f[KPC];
        dec_unpack<T>(xr[i], f);
#pragma unroll
        for (int e = 0; e < KPC; ++e) f[e] = dec_round<T>(f[e] * rstd) * nwr[i][e / 4][e % 4];
        xr[i] = dec_pack<T>(f);
      }
    }
#pragma unroll
    for (int i = 0; i < XN; ++i)
      if (tid + 256 * i < NCH) xs[tid + 256 * i] = xr[i];
    if constexpr (RHD > 0) {
      if (tid < 2 * RHD) rope_lds[tid] = ropev;
    }
    __syncthreads();
  }

  // ---- 3. the stream
  float a0 = 0.f, a1 = 0.f;
  int cpl = 0, ccb = 0;            // consume cursor (local pair, batch within the pair)
  auto consume = [&](const u32x4_t (&v0)[CPB], const u32x4_t (&v1)[CPB]) {
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      const int wc = ccb * (64 * CPB) + u * 64 + lane;
      if constexpr (W8) {
        const u32x4_t xa = xs[2 * wc], xb = xs[2 * wc + 1];
        a0 += dec_dot_w8(v0[u], xa, xb);
        a1 += dec_dot_w8(v1[u], xa, xb);
      } else {
        const u32x4_t xv = xs[wc];
        a0 += dec_dot<T>(v0[u], xv);
        a1 += dec_dot<T>(v1[u], xv);
      }
    }
    if (++ccb == NB) {
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      if (lane == 0) { res[wave][cpl][0] = a0; res[wave][cpl][1] = a1; }
      a0 = 0.f;
      a1 = 0.f;
      ccb = 0;
      ++cpl;
    }
  };
  int b = 0;
  for (; b + 4 <= total; b += 2) {      // on entry: batch b in set a, batch b+1 in set b
    consume(va0, va1);
    issue(va0, va1);
    consume(vb0, vb1);
    issue(vb0, vb1);
  }
  const int rem = total - b;
  if (rem == 3) {
    consume(va0, va1);
    issue(va0, va1);
    consume(vb0, vb1);
    consume(va0, va1);
  } else if (rem == 2) {
    consume(va0, va1);
    consume(vb0, vb1);
  } else if (rem == 1) {
    consume(va0, va1);
  }
  // ---- 4. epilogue: one lane per pair
  if constexpr (RHD > 0) {
    const int pos = *p.pos_dev;
    for (int i = lane; i < np; i += 64) {
      const int pi = p0 + i, hh = pi / RHD, d = pi % RHD;
      const float x1 = dec_round<TO>(res[wave][i][0]), x2 = dec_round<TO>(res[wave][i][1]);      // the projection materialises in the activation dtype (HF)
      float o1 = x1, o2 = x2;
      TO* dst;
      if (hh < p.H + p.Hkv) {      // q and k heads: rotate-half RoPE, the arithmetic of vg_rope_kv_append / decode_attn_kernel
        const float c = rope_lds[d], sv = rope_lds[RHD + d];
        if (sizeof(TO) == 2) {
          const float cb = bf2f(f2bf(c)), sb = bf2f(f2bf(sv));
          o1 = bf2f(f2bf(x1 * cb)) + bf2f(f2bf(-x2 * sb));
          o2 = bf2f(f2bf(x2 * cb)) + bf2f(f2bf(x1 * sb));
        } else {
          o1 = x1 * c - x2 * sv;
          o2 = x2 * c + x1 * sv;
        }
        dst = hh < p.H ? (TO*)p.y + (int64_t)hh * (2 * RHD) : (TO*)p.kc + ((int64_t)pos * p.Hkv + (hh - p.H)) * (2 * RHD);
      } else {
        dst = (TO*)p.vc + ((int64_t)pos * p.Hkv + (hh - p.H - p.Hkv)) * (2 * RHD);
      }
      vg_elt<TO>::st(dst + d, o1);
      vg_elt<TO>::st(dst + d + RHD, o2);
    }
  } else {
    for (int i = lane; i < np; i += 64) dec_gemv_store<T, TO, GLU>(p, p0 + i, res[wave][i][0], res[wave][i][1]);
  }
}

template <typename T, typename TO, bool GLU>
__global__ __launch_bounds__(256) void decode_gemv_kernel(DecGemvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char dec_smem[];
  __shared__ float red[4];
  constexpr int KPC = 16 / sizeof(T);
  u32x4_t* xs = (u32x4_t*)dec_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = p.K / KPC;
  const int npair = GLU ? p.N : (p.N + 1) / 2;
  const int gw = blockIdx.x * 4 + wave;
  const T* W = (const T*)p.W;
  {
    const u32x4_t* xg = (const u32x4_t*)p.x;
    float ss = 0.f;
    for (int c = tid; c < nch; c += 256) {
      const u32x4_t v = xg[c];
      xs[c] = v;
      if (p.nw) {
        float f[KPC];
        dec_unpack<T>(v, f);
#pragma unroll
        for (int e = 0; e < KPC; ++e) ss += f[e] * f[e];
      }
    }
    if (p.nw) {
      ss = wave_sum(ss);
      if (lane == 0) red[wave] = ss;
      __syncthreads();
      const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)p.K + p.eps);
      for (int c = tid; c < nch; c += 256) {   // each thread re-reads only the chunks it wrote itself
        float f[KPC];
        dec_unpack<T>(xs[c], f);
#pragma unroll
        for (int e = 0; e < KPC; ++e) f[e] = dec_round<T>(f[e] * rstd) * p.nw[c * KPC + e];
        xs[c] = dec_pack<T>(f);
      }
    }
    __syncthreads();
  }
  const int TW = gridDim.x * 4;
  for (int i = gw; i < npair; i += TW) {
    const int n0 = GLU ? i : 2 * i;
    const int n1 = GLU ? p.N + i : min(2 * i + 1, p.N - 1);
    const u32x4_t* w0 = (const u32x4_t*)(W + (int64_t)n0 * p.ldw);
    const u32x4_t* w1 = (const u32x4_t*)(W + (int64_t)n1 * p.ldw);
    float a0 = 0.f, a1 = 0.f;
    int c = lane;
    for (; c + 3 * 64 < nch; c += 4 * 64) {
      u32x4_t v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { v0[u] = w0[c + u * 64]; v1[u] = w1[c + u * 64]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const u32x4_t xv = xs[c + u * 64];
        a0 += dec_dot<T>(v0[u], xv);
        a1 += dec_dot<T>(v1[u], xv);
      }
    }
    for (; c < nch; c += 64) {
      const u32x4_t xv = xs[c];
      a0 += dec_dot<T>(w0[c], xv);
      a1 += dec_dot<T>(w1[c], xv);
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) dec_gemv_store<T, TO, GLU>(p, i, a0, a1);
  }
}

template <typename T, typename TO, bool GLU>
static int launch_decode_gemv(DecGemvArgs p, hipStream_t st) {
  constexpr int KPC = 16 / sizeof(T);
  static int bpc = -1;
  if (bpc < 0) {
    // max workgroups per CU the row pairs are spread over (measured constant).  2 = what is resident (these kernels take 254 VGPRs): with 4 the q|k|v
    // projection ran 768 workgroups in one and a half rounds (r03: 3.23 -> 3.205 ms per token on C2)
    bpc = 2;
  }
  const int npair = GLU ? p.N : (p.N + 1) / 2;
  const int maxw = 256 * bpc * 4;
  int ppw = (npair + maxw - 1) / maxw;              // row pairs per wave
  // prefer a split that gives every CU the same number of workgroups (a multiple of 256 workgroups)
  for (int c = ppw; c <= 2 * ppw; ++c)
    if (((npair + 4 * c - 1) / (4 * c)) % 256 == 0 && npair % (4 * c) == 0) { ppw = c; break; }
  if (ppw > DEC_MAX_PPW) ppw = DEC_MAX_PPW;
  const int blocks = (npair + 4 * ppw - 1) / (4 * ppw);
  p.ppw = ppw;
  const size_t lds = (size_t)p.K * sizeof(T);
  const int nch = p.K / KPC;
  if (nch == 384) {        // Phi-3-mini hidden size: three batches of two chunks per lane
    decode_gemv_fast_kernel<T, TO, GLU, 3, 2><<<blocks, 256, lds, st>>>(p);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  const int nb = nch % 256 == 0 ? nch / 256 : 0;
  switch (nb) {
    case 1: decode_gemv_fast_kernel<T, TO, GLU, 1><<<blocks, 256, lds, st>>>(p); break;
    case 2: decode_gemv_fast_kernel<T, TO, GLU, 2><<<blocks, 256, lds, st>>>(p); break;
    case 4: decode_gemv_fast_kernel<T, TO, GLU, 4><<<blocks, 256, lds, st>>>(p); break;
    case 7: decode_gemv_fast_kernel<T, TO, GLU, 7><<<blocks, 256, lds, st>>>(p); break;
    case 8: decode_gemv_fast_kernel<T, TO, GLU, 8><<<blocks, 256, lds, st>>>(p); break;
    default: decode_gemv_kernel<T, TO, GLU><<<blocks, 256, lds, st>>>(p); break;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

template <typename TO, bool GLU>
static int launch_decode_gemv_w8(DecGemvArgs p, hipStream_t st) {
  static int bpc = -1;
  if (bpc < 0) {
    bpc = 4;
  }
  const int npair = GLU ? p.N : (p.N + 1) / 2;
  const int maxw = 256 * bpc * 4;
  int ppw = (npair + maxw - 1) / maxw;
  for (int c = ppw; c <= 2 * ppw; ++c)
    if (((npair + 4 * c - 1) / (4 * c)) % 256 == 0 && npair % (4 * c) == 0) { ppw = c; break; }
  if (ppw > DEC_MAX_PPW) ppw = DEC_MAX_PPW;
  const int blocks = (npair + 4 * ppw - 1) / (4 * ppw);
  p.ppw = ppw;
  const size_t lds = (size_t)p.K * 2;
  const int nchw = p.K / 16;               // weight chunks per row
  if (nchw % 256 == 0 && nchw / 256 == 1) decode_gemv_fast_kernel<bf16_t, TO, GLU, 1, 4, true><<<blocks, 256, lds, st>>>(p);
  else if (nchw % 256 == 0 && nchw / 256 == 2) decode_gemv_fast_kernel<bf16_t, TO, GLU, 2, 4, true><<<blocks, 256, lds, st>>>(p);
  else if (nchw == 7 * 128) decode_gemv_fast_kernel<bf16_t, TO, GLU, 7, 2, true><<<blocks, 256, lds, st>>>(p);
  else if (nchw == 3 * 64) decode_gemv_fast_kernel<bf16_t, TO, GLU, 3, 1, true><<<blocks, 256, lds, st>>>(p);
  else {
    vg_set_error("vg_decode_gemv_w8: K=%d is not one of the supported row lengths (3072, 4096, 8192, 14336)", p.K);
    return VG_ERR_UNSUPPORTED;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_decode_gemv_w8(const void* x, const uint8_t* W8, int64_t ldw, const float* wscale, void* y, const float* norm_w,
                                 float eps, const void* R, int N, int K, int glu, int out_dtype, vg_stream_t stream) {
  VG_CHECK(x && W8 && wscale && y && N > 0 && K > 0, VG_ERR_ARG, "vg_decode_gemv_w8: bad args N=%d K=%d", N, K);
  VG_CHECK(K % 16 == 0 && ldw % 16 == 0, VG_ERR_ARG, "vg_decode_gemv_w8: K/ldw must be multiples of 16 (K=%d ldw=%lld)", K, (long long)ldw);
  VG_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)W8 & 15) == 0 && ((uintptr_t)norm_w & 15) == 0, VG_ERR_ARG,
           "vg_decode_gemv_w8: x/W/norm_w must be 16-byte aligned");
  VG_CHECK((int64_t)K * 2 <= 64 * 1024, VG_ERR_UNSUPPORTED, "vg_decode_gemv_w8: K=%d does not fit the LDS staging", K);
  VG_CHECK(out_dtype == VG_BF16 || out_dtype == VG_F32, VG_ERR_ARG, "vg_decode_gemv_w8: bad out_dtype %d", out_dtype);
  DecGemvArgs p{x, W8, y, norm_w, R, N, K, ldw, eps, 1, wscale};
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == VG_BF16) return glu ? launch_decode_gemv_w8<bf16_t, true>(p, st) : launch_decode_gemv_w8<bf16_t, false>(p, st);
  return glu ? launch_decode_gemv_w8<float, true>(p, st) : launch_decode_gemv_w8<float, false>(p, st);
}

extern "C" int vg_decode_gemv(const void* x, const void* W, int64_t ldw, void* y, const float* norm_w, float eps,
                              const void* R, int N, int K, int glu, int in_dtype, int out_dtype, vg_stream_t stream) {
  VG_CHECK(x && W && y && N > 0 && K > 0, VG_ERR_ARG, "vg_decode_gemv: bad args N=%d K=%d", N, K);
  VG_CHECK(in_dtype == VG_BF16 || in_dtype == VG_F32, VG_ERR_ARG, "vg_decode_gemv: bad in_dtype %d", in_dtype);
  const int kpc = in_dtype == VG_BF16 ? 8 : 4, es = in_dtype == VG_BF16 ? 2 : 4;
  VG_CHECK(K % kpc == 0 && ldw % kpc == 0, VG_ERR_ARG, "vg_decode_gemv: K/ldw must be multiples of %d (K=%d ldw=%lld)", kpc, K, (long long)ldw);
  VG_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)norm_w & 15) == 0, VG_ERR_ARG,
           "vg_decode_gemv: x/W/norm_w must be 16-byte aligned");
  VG_CHECK((int64_t)K * es <= 64 * 1024, VG_ERR_UNSUPPORTED, "vg_decode_gemv: K=%d does not fit the LDS staging", K);
  DecGemvArgs p{x, W, y, norm_w, R, N, K, ldw, eps, 1, nullptr};
  hipStream_t st = (hipStream_t)stream;
#define VG_DEC_GEMV(TI, TOO) return glu ? launch_decode_gemv<TI, TOO, true>(p, st) : launch_decode_gemv<TI, TOO, false>(p, st)
  if (in_dtype == VG_BF16 && out_dtype == VG_BF16) VG_DEC_GEMV(bf16_t, bf16_t);
  if (in_dtype == VG_BF16 && out_dtype == VG_F32) VG_DEC_GEMV(bf16_t, float);
  if (in_dtype == VG_F32 && out_dtype == VG_F32) VG_DEC_GEMV(float, float);
#undef VG_DEC_GEMV
  vg_set_error("vg_decode_gemv: unsupported dtype combination %d -> %d", in_dtype, out_dtype);
  return VG_ERR_UNSUPPORTED;
}

// q|k|v projection of the new token with RMSNorm in front and RoPE + KV-cache append behind it (r06): the rows of a rotate-half pair
// (d, d + D/2) go to ONE lane of the GEMV's epilogue, so the rotation is two fmas on values that are in registers anyway, the new key / value
// rows are written where the attention kernel reads them, and that kernel (vg_decode_attention2) starts on its K / V loads instead of on a
// RoPE phase.  rope_cs = [cos row | sin row] of position *pos_dev (vg_decode_advance keeps it current): a fixed address, no dependent load.
extern "C" int vg_decode_qkv_rope_supported(int H, int Hkv, int D, int K, int dtype) {
  if (H <= 0 || Hkv <= 0 || D != 128) return 0;
  if (dtype == VG_BF16) return K == 4096 || K == 2048;
  if (dtype == VG_F32) return K == 4096;
  return 0;
}

template <typename T, int NB>
static int launch_decode_qkv_rope(DecGemvArgs p, hipStream_t st) {
  const int npair = p.N / 2;
  const int maxw = 256 * 2 * 4;
  int ppw = (npair + maxw - 1) / maxw;
  for (int c = ppw; c <= 2 * ppw; ++c)
    if (((npair + 4 * c - 1) / (4 * c)) % 256 == 0 && npair % (4 * c) == 0) { ppw = c; break; }
  if (ppw > DEC_MAX_PPW) ppw = DEC_MAX_PPW;
  p.ppw = ppw;
  const int blocks = (npair + 4 * ppw - 1) / (4 * ppw);
  decode_gemv_fast_kernel<T, T, false, NB, 4, false, 64><<<blocks, 256, (size_t)p.K * sizeof(T), st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_decode_qkv_rope(const void* x, const void* Wqkv, int64_t ldw, const float* norm_w, float eps, void* q_out, void* k_cache,
                                  void* v_cache, const float* rope_cs, const int* pos_dev, int H, int Hkv, int D, int K, int dtype,
                                  vg_stream_t stream) {
  VG_CHECK(x && Wqkv && q_out && k_cache && v_cache && rope_cs && pos_dev, VG_ERR_ARG, "vg_decode_qkv_rope: null pointer");
  VG_CHECK(vg_decode_qkv_rope_supported(H, Hkv, D, K, dtype), VG_ERR_UNSUPPORTED, "vg_decode_qkv_rope: H=%d Hkv=%d D=%d K=%d dtype=%d not covered", H, Hkv, D, K,
           dtype);
  const int kpc = dtype == VG_BF16 ? 8 : 4;
  VG_CHECK(ldw % kpc == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)Wqkv & 15) == 0 && ((uintptr_t)norm_w & 15) == 0, VG_ERR_ARG,
           "vg_decode_qkv_rope: alignment (16 bytes; ldw a multiple of %d)", kpc);
  DecGemvArgs p{x, Wqkv, q_out, norm_w, nullptr, (H + 2 * Hkv) * D, K, ldw, eps, 1, nullptr, rope_cs, pos_dev, k_cache, v_cache, H, Hkv};
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VG_BF16) return K == 4096 ? launch_decode_qkv_rope<bf16_t, 2>(p, st) : launch_decode_qkv_rope<bf16_t, 1>(p, st);
  return launch_decode_qkv_rope<float, 4>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention of the one new token.  grid = (ceil(max_len/64) splits, Hkv): a workgroup owns 64 keys of one KV head and
// the G = H/Hkv query heads that share it.  All K and V bytes of the split are requested up front (one HBM round
// trip), RoPE of q and of the new k runs while they are in flight, the workgroup that owns position `pos` appends the
// new k/v rows to the cache, and the last workgroup of a KV head to finish merges the per-split (max, sum, acc)
// partials — ordering by an agent-scope arrival counter that resets itself, so the launch is graph-replayable.
// vg_decode_layer's flag region (ints; 32-int = 128-byte lines so that no two hot words share one): line 0 = [0] KV heads merged, [1] a wait gave up;
// lines 4 + 16 r + s = arrival stripe s of GEMV role r; lines 36 + r = the stripes of role r completed; lines 40 + 64 b + i = go flag i of boundary b
constexpr int DEC_LINE = 32, DEC_STRIPES = 16, DEC_GO = 64;
constexpr int DEC_CHAIN_INTS = (40 + 3 * DEC_GO) * DEC_LINE;
struct DecAttnArgs {
  const void* qkv; void* kc; void* vc; const float* cs; const float* sn; void* o;
  float* ws; int* cnt; const int* pos_dev;
  int H, Hkv, D, nsplit; float scale;
  int window;   // > 0: only the last `window` positions (the new one included) are attended (sliding-window LLMs); 0: all
};

__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_load_dword sc1
}
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // global_store_dword sc1 (write-through)
}

template <typename TO> __device__ __forceinline__ float ld_agent_elt(const TO* p);
template <> __device__ __forceinline__ float ld_agent_elt<float>(const float* p) { return ld_agent(p); }
template <> __device__ __forceinline__ float ld_agent_elt<bf16_t>(const bf16_t* p) {
  return bf2f(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));       // global_load_ushort sc1
}
template <typename TO> __device__ __forceinline__ void st_agent_elt(TO* p, float v);
template <> __device__ __forceinline__ void st_agent_elt<float>(float* p, float v) { st_agent(p, v); }
template <> __device__ __forceinline__ void st_agent_elt<bf16_t>(bf16_t* p, float v) {
  __hip_atomic_store(p, f2bf(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_store_short sc1
}

// DT = compile-time head_dim (0 = run-time p.D): with it every load loop below has a constant trip count, so the
// kernel is straight-line up to the first wait and the compiler's vmcnt bookkeeping stays exact.
// AG (decode_layer_kernel): the merged output row leaves write-through (sc1) and *done is bumped once per KV head after it — the
// o_proj workgroups of the same launch wait on that counter and read the row with sc1 loads.
// LB = 64-key blocks per workgroup (2 at long caches: half as many partials to publish, arrive and merge — the merge then takes ONE pass of <= 32)
template <typename T, int G, int DT, bool AG, int LB = 1>
__device__ __forceinline__ void dec_attn_body(const DecAttnArgs& p, const int s, const int kvh, char* dec_smem, int& ticket, int* done) {
  constexpr int KPC = 16 / sizeof(T);
  constexpr int L = 64 * LB;
  const int pos = *p.pos_dev;
  const int active = pos / L + 1;
  const int lo = p.window > 0 ? max(0, pos + 1 - p.window) : 0;     // first visible position
  const int first = lo / L;                                         // first split with a visible key
  if (s >= active || s < first) return;
  const int nact = active - first;
  const int D = DT ? DT : p.D;
  const int CH = D / KPC, hd = D / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = s * L;
  const int nk = min(L, pos + 1 - j0);
  const int KP = 256 / CH, cidx = tid % CH, kslot = min(tid / CH, KP - 1);
  const bool vlive = tid / CH < KP;
  const int NKI = DT ? (DT / KPC + 3) / 4 : 8;                 // K chunks per lane (wave w takes chunks w, w+4, ...)
  const int NVI = DT ? (64 + 256 / (DT / KPC) - 1) / (256 / (DT / KPC)) : 8;   // V rows per thread
  const int NRI = DT ? ((G + 1) * (DT / 2) + 255) / 256 : ((G + 1) * hd + 255) / 256;   // rope pairs per thread
  float* qs = (float*)dec_smem;       // [G][D] roped q
  float* knew = qs + G * D;           // [D] roped new k
  float* vnew = knew + D;             // [D]
  float* sp = vnew + D;               // [4 waves][G][64] partial scores
  float* sc = sp + LB * 4 * G * 64;   // [G][LB][64] exp(score - max)      (sp: [LB][4 waves][G][64])
  float* ms = sc + LB * G * 64;       // [G] max | [G] sum
  float* po = ms + 2 * G + ((4 - ((2 * G) & 3)) & 3);   // [KP][G][D] partial outputs

  // ---- 1. every global load up front, the early-needed (L2-resident) ones first: vmcnt retires in order
  const T* qkv = (const T*)p.qkv;
  const float* cs = p.cs + (int64_t)pos * hd;
  const float* sn = p.sn + (int64_t)pos * hd;
  float rx1[5], rx2[5], rc[5], rs_[5];
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    if (it < NRI) {
      const int idx = min(tid + 256 * it, (G + 1) * hd - 1);
      const int gh = idx / hd, d = idx % hd;
      const T* src = gh < G ? qkv + (int64_t)(kvh * G + gh) * D : qkv + (int64_t)(p.H + kvh) * D;
      rx1[it] = vg_elt<T>::ld(src + d);
      rx2[it] = vg_elt<T>::ld(src + d + hd);
      rc[it] = cs[d];
      rs_[it] = sn[d];
    }
  }
  const float vn = vg_elt<T>::ld(qkv + (int64_t)(p.H + p.Hkv + kvh) * D + min(tid, D - 1));
  const int64_t rs = (int64_t)p.Hkv * D;
  const T* kb = (const T*)p.kc + ((int64_t)j0 * p.Hkv + kvh) * D;
  const T* vb = (const T*)p.vc + ((int64_t)j0 * p.Hkv + kvh) * D;
  u32x4_t kreg[LB][8], vreg[LB][8];
#pragma unroll
  for (int sub = 0; sub < LB; ++sub) {
    const int jl = min(sub * 64 + lane, nk - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < NKI) kreg[sub][i] = *(const u32x4_t*)(kb + jl * rs + min(wave + 4 * i, CH - 1) * KPC);
  }
#pragma unroll
  for (int sub = 0; sub < LB; ++sub) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < NVI) vreg[sub][i] = *(const u32x4_t*)(vb + min(sub * 64 + kslot + KP * i, nk - 1) * rs + cidx * KPC);
  }
  // ---- 2. RoPE of the G query heads and of the new key; the new value row (vg_rope_kv_append arithmetic)
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    if (it < NRI) {
      const int idx = tid + 256 * it;
      const int gh = idx / hd, d = idx % hd;
      const float x1 = rx1[it], x2 = rx2[it], c = rc[it], sv = rs_[it];
      float o1, o2;
      if (sizeof(T) == 2) {
        const float cb = bf2f(f2bf(c)), sb = bf2f(f2bf(sv));
        o1 = bf2f(f2bf(x1 * cb)) + bf2f(f2bf(-x2 * sb));
        o2 = bf2f(f2bf(x2 * cb)) + bf2f(f2bf(x1 * sb));
      } else {
        o1 = x1 * c - x2 * sv;
        o2 = x2 * c + x1 * sv;
      }
      o1 = dec_round<T>(o1);
      o2 = dec_round<T>(o2);
      if (idx < (G + 1) * hd) {
        float* dst = gh < G ? qs + gh * D : knew;
        dst[d] = o1;
        dst[d + hd] = o2;
      }
    }
  }
  if (tid < D) vnew[tid] = vn;
  __syncthreads();
  if (s == active - 1) {   // this split holds position `pos`: append the new rows to the cache
    T* kdst = (T*)p.kc + ((int64_t)pos * p.Hkv + kvh) * D;
    T* vdst = (T*)p.vc + ((int64_t)pos * p.Hkv + kvh) * D;
    if (tid < D) {
      vg_elt<T>::st(kdst + tid, knew[tid]);
      vg_elt<T>::st(vdst + tid, vnew[tid]);
    }
  }
  // ---- 3. q.k: lane = key, wave w covers 16-byte chunks w, w+4, ... of the head dimension
#pragma unroll
  for (int sub = 0; sub < LB; ++sub) {
    const bool isnew = (j0 + sub * 64 + lane == pos);
    float part[G];
#pragma unroll
    for (int g = 0; g < G; ++g) part[g] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = wave + 4 * i;
      if (i < NKI && c < CH) {
        float kf[KPC];
        dec_unpack<T>(kreg[sub][i], kf);
        if (isnew) {
#pragma unroll
          for (int e = 0; e < KPC; ++e) kf[e] = knew[c * KPC + e];
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int e = 0; e < KPC; ++e) part[g] = fmaf(kf[e], qs[g * D + c * KPC + e], part[g]);
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) sp[((sub * 4 + wave) * G + g) * 64 + lane] = part[g];
  }
  __syncthreads();
  for (int g = wave; g < G; g += 4) {
    float v[LB];
    bool seen[LB];
    float mx = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < LB; ++sub) {
      const float* q4 = sp + (sub * 4 * G + g) * 64 + lane;
      v[sub] = (q4[0] + q4[G * 64] + q4[2 * G * 64] + q4[3 * G * 64]) * p.scale;
      seen[sub] = sub * 64 + lane < nk && j0 + sub * 64 + lane >= lo;
      if (!seen[sub]) v[sub] = -INFINITY;
      mx = fmaxf(mx, v[sub]);
    }
    const float m = wave_max(mx);
    float es = 0.f;
#pragma unroll
    for (int sub = 0; sub < LB; ++sub) {
      const float e = seen[sub] ? __expf(v[sub] - m) : 0.f;
      sc[(g * LB + sub) * 64 + lane] = e;
      es += e;
    }
    const float l = wave_sum(es);
    if (lane == 0) { ms[g] = m; ms[G + g] = l; }
  }
  __syncthreads();
  // ---- 4. p.v: thread = (key slot, 16-byte chunk of the head dimension)
  {
    float acc[G][KPC];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < KPC; ++e) acc[g][e] = 0.f;
#pragma unroll
    for (int sub = 0; sub < LB; ++sub)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = sub * 64 + kslot + KP * i;
      if (i < NVI && vlive && kslot + KP * i < 64 && j < nk) {      // (KP does not divide 64 at head dim 96: the last row round overshoots the block)
        float vf[KPC];
        dec_unpack<T>(vreg[sub][i], vf);
        if (j0 + j == pos) {
#pragma unroll
          for (int e = 0; e < KPC; ++e) vf[e] = vnew[cidx * KPC + e];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float pj = sc[g * LB * 64 + j];
#pragma unroll
          for (int e = 0; e < KPC; ++e) acc[g][e] = fmaf(pj, vf[e], acc[g][e]);
        }
      }
    }
    if (vlive) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < KPC; ++e) po[(kslot * G + g) * D + cidx * KPC + e] = acc[g][e];
    }
  }
  __syncthreads();
  // ---- 5. publish the split's partials write-through (sc1): the merging workgroup reads them with sc1 loads, so the
  //         hand-off needs no cache write-back / invalidate fences (MI355X_MICROARCH.md, valid hand-off forms)
  float* wsp = p.ws + ((int64_t)kvh * p.nsplit + s) * G * (D + 2);
  for (int o = tid; o < G * D; o += 256) {
    const int g = o / D, d = o % D;
    float sum = 0.f;
    for (int k = 0; k < KP; ++k) sum += po[(k * G + g) * D + d];
    st_agent(wsp + g * (D + 2) + d, sum);
  }
  if (tid < G) { st_agent(wsp + tid * (D + 2) + D, ms[tid]); st_agent(wsp + tid * (D + 2) + D + 1, ms[G + tid]); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) ticket = __hip_atomic_fetch_add(&p.cnt[kvh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != nact - 1) return;
  // ---- 6. merge by the last workgroup of this KV head to arrive.  All loads of a pass (the (max, sum) of every
  //         split and two accumulator columns per thread) are requested together: one fabric round trip per pass.
  const float* wbase = p.ws + ((int64_t)kvh * p.nsplit + first) * G * (D + 2);   // the visible splits [first, active)
  float* cw = sp;                    // [nact][G] max -> weight      (sp is free again; nsplit*G*2 <= 4*G*64)
  float* cl = sp + p.nsplit * G;     // [nact][G] sum
  T* out = (T*)p.o;
  const int nml = nact * G;
  for (int o0 = 0; o0 < G * D; o0 += 512) {
    int oo[2], og[2], od[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      oo[k] = o0 + tid + 256 * k;
      const int oc = min(oo[k], G * D - 1);
      og[k] = oc / D;
      od[k] = oc % D;
    }
    float num[2] = {0.f, 0.f};
    for (int sb = 0; sb < nact; sb += 32) {
      float mv[2], lv[2];
      if (o0 == 0 && sb == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {       // nml <= 128 * 8: two slots per thread cover 512, the rest loops below
          const float* w = wbase + (int64_t)min(tid + 256 * k, nml - 1) * (D + 2);
          mv[k] = ld_agent(w + D);
          lv[k] = ld_agent(w + D + 1);
        }
      }
      float v[2][32];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int u = 0; u < 32; ++u)
          v[k][u] = ld_agent(wbase + ((int64_t)min(sb + u, nact - 1) * G + og[k]) * (D + 2) + od[k]);
      if (o0 == 0 && sb == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (tid + 256 * k < nml) { cw[tid + 256 * k] = mv[k]; cl[tid + 256 * k] = lv[k]; }
        for (int t = tid + 512; t < nml; t += 256) {
          const float* w = wbase + (int64_t)t * (D + 2);
          cw[t] = ld_agent(w + D);
          cl[t] = ld_agent(w + D + 1);
        }
        __syncthreads();
        for (int gg = wave; gg < G; gg += 4) {
          float M = -INFINITY;
          for (int s2 = lane; s2 < nact; s2 += 64) M = fmaxf(M, cw[s2 * G + gg]);
          M = wave_max(M);
          float den = 0.f;
          for (int s2 = lane; s2 < nact; s2 += 64) {
            const float wgt = __expf(cw[s2 * G + gg] - M);
            cw[s2 * G + gg] = wgt;
            den = fmaf(wgt, cl[s2 * G + gg], den);
          }
          den = wave_sum(den);
          if (lane == 0) ms[gg] = 1.0f / den;
        }
        __syncthreads();
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int u = 0; u < 32; ++u)
          if (sb + u < nact) num[k] = fmaf(cw[(sb + u) * G + og[k]], v[k][u], num[k]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (oo[k] < G * D) {
        if constexpr (AG) st_agent_elt<T>(out + (int64_t)(kvh * G + og[k]) * D + od[k], num[k] * ms[og[k]]);
        else vg_elt<T>::st(out + (int64_t)(kvh * G + og[k]) * D + od[k], num[k] * ms[og[k]]);
      }
  }
  if (tid == 0) __hip_atomic_store(&p.cnt[kvh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if constexpr (AG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) ticket = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket == p.Hkv - 1 && tid < DEC_GO)        // every head's row is in memory: release the o_proj workgroups (64 flags, 8 pollers each)
      __hip_atomic_store(done + (40 + tid) * DEC_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T, int G, int DT, int LB = 1>
__global__ __launch_bounds__(256) void decode_attn_kernel(DecAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char dec_smem[];
  __shared__ int ticket;
  dec_attn_body<T, G, DT, false, LB>(p, blockIdx.x, blockIdx.y, dec_smem, ticket, nullptr);
}

template <typename T, int G, int DT, int LB = 1>
static void launch_decode_attn_gd(const DecAttnArgs& p, dim3 grid, size_t lds, hipStream_t st) {
  static size_t lds_cap = 64 * 1024;   // raise the dynamic-LDS cap only when a shape needs it (never inside a replay)
  if (lds > lds_cap) {
    (void)hipFuncSetAttribute((const void*)decode_attn_kernel<T, G, DT, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_cap = lds;
  }
  decode_attn_kernel<T, G, DT, LB><<<grid, 256, lds, st>>>(p);
}

template <typename T, int G>
static void launch_decode_attn_g(const DecAttnArgs& p, dim3 grid, size_t lds, hipStream_t st) {
  if (p.D == 128) launch_decode_attn_gd<T, G, 128>(p, grid, lds, st);
  else if (p.D == 96) launch_decode_attn_gd<T, G, 96>(p, grid, lds, st);      // Phi-3-mini: the released checkpoint's LLM
  else launch_decode_attn_gd<T, G, 0>(p, grid, lds, st);
}

template <typename T>
static int launch_decode_attn(const DecAttnArgs& p, int G, int keys_per_wg, hipStream_t st) {
  constexpr int KPC = 16 / sizeof(T);
  const int KP = 256 / (p.D / KPC);
  size_t lds = sizeof(float) * ((size_t)G * p.D + 2 * p.D + 4 * G * 64 + G * 64 + 2 * G + 4 + (size_t)KP * G * p.D);
  dim3 grid(p.nsplit, p.Hkv);
  if constexpr (sizeof(T) == 2) {
    // 128 keys per workgroup (the caller's hint for long caches): the two shapes the pipeline decodes with — Llama-3 (G = 4, d = 128), Phi-3 (MHA, d = 96)
    if (keys_per_wg == 128 && ((G == 4 && p.D == 128) || (G == 1 && p.D == 96))) {
      lds += sizeof(float) * (4 * G * 64 + G * 64);
      if (G == 4) launch_decode_attn_gd<T, 4, 128, 2>(p, grid, lds, st);
      else launch_decode_attn_gd<T, 1, 96, 2>(p, grid, lds, st);
      VG_LAUNCH_CHECK();
      return VG_OK;
    }
  }
  switch (G) {
    case 1: launch_decode_attn_g<T, 1>(p, grid, lds, st); break;
    case 2: launch_decode_attn_g<T, 2>(p, grid, lds, st); break;
    case 4: launch_decode_attn_g<T, 4>(p, grid, lds, st); break;
    case 8: launch_decode_attn_g<T, 8>(p, grid, lds, st); break;
    default:
      vg_set_error("vg_decode_attention: H/Hkv = %d not in {1,2,4,8}", G);
      return VG_ERR_UNSUPPORTED;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int64_t vg_decode_attention_ws_floats(int H, int Hkv, int D, int max_len) {
  if (H <= 0 || Hkv <= 0 || D <= 0 || max_len <= 0 || H % Hkv) return -1;
  const int64_t nsplit = (max_len + 63) / 64;
  return (int64_t)Hkv * nsplit * (H / Hkv) * (D + 2) + Hkv;
}

extern "C" int vg_decode_attention(const void* qkv, void* k_cache, void* v_cache, const float* cos, const float* sin,
                                   void* out, int H, int Hkv, int D, int max_len, int window, float scale, const int* pos_dev,
                                   float* workspace, int64_t ws_floats, int keys_per_wg, int dtype, vg_stream_t stream) {
  VG_CHECK(qkv && k_cache && v_cache && cos && sin && out && pos_dev && workspace, VG_ERR_ARG, "vg_decode_attention: null pointer");
  VG_CHECK(keys_per_wg == 0 || keys_per_wg == 64 || keys_per_wg == 128, VG_ERR_ARG, "vg_decode_attention: keys_per_wg %d not in {0, 64, 128}", keys_per_wg);
  VG_CHECK(H > 0 && Hkv > 0 && H % Hkv == 0 && D > 0 && D % 2 == 0 && max_len > 0, VG_ERR_ARG,
           "vg_decode_attention: bad shape H=%d Hkv=%d D=%d max_len=%d", H, Hkv, D, max_len);
  VG_CHECK(dtype == VG_BF16 || dtype == VG_F32, VG_ERR_ARG, "vg_decode_attention: bad dtype %d", dtype);
  VG_CHECK(window >= 0, VG_ERR_ARG, "vg_decode_attention: window %d < 0", window);
  const int kpc = dtype == VG_BF16 ? 8 : 4, es = dtype == VG_BF16 ? 2 : 4;
  VG_CHECK(D % kpc == 0 && D * es <= 512, VG_ERR_UNSUPPORTED, "vg_decode_attention: head_dim %d unsupported (multiple of %d, <= %d)", D, kpc, 512 / es);
  VG_CHECK((((uintptr_t)k_cache | (uintptr_t)v_cache) & 15) == 0, VG_ERR_ARG, "vg_decode_attention: caches must be 16-byte aligned");
  const int64_t need = vg_decode_attention_ws_floats(H, Hkv, D, max_len);
  VG_CHECK(ws_floats >= need, VG_ERR_ARG, "vg_decode_attention: workspace %lld < %lld floats", (long long)ws_floats, (long long)need);
  const int nsplit = (max_len + 63) / 64;
  VG_CHECK(nsplit <= 128, VG_ERR_UNSUPPORTED, "vg_decode_attention: max_len %d > 8192", max_len);
  DecAttnArgs p{qkv, k_cache, v_cache, cos, sin, out, workspace, (int*)(workspace + (need - Hkv)), pos_dev, H, Hkv, D, nsplit, scale, window};
  if (dtype == VG_BF16) return launch_decode_attn<bf16_t>(p, H / Hkv, keys_per_wg, (hipStream_t)stream);
  return launch_decode_attn<float>(p, H / Hkv, keys_per_wg, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// One launch per decoder layer behind the q|k|v projection (r03): attention, o_proj (+ residual), and — bf16 Llama
// widths — RMSNorm + gate|up + SwiGLU and down_proj (+ residual) as ROLES of one grid.  A workgroup's role follows
// from its index, roles are laid out in dependency order, and the dispatcher hands out workgroups in index order, so
// every workgroup a waiting one depends on is already running: a GEMV workgroup requests its first two batches of
// weight rows (16 x 16 bytes per lane — all of its rows for o_proj / down_proj), THEN waits for the producer role,
// then reads the input row with sc1 loads.  What the separate launches serialised — the weight stream's ramp behind
// the latency-bound attention, and each GEMV's ramp behind the previous one's tail — now overlaps.  Arithmetic
// (order of every sum) is the separate kernels': results are bit-identical.
// Hand-off: a producer workgroup stores sc1 (write-through) -> s_waitcnt vmcnt(0) -> barrier -> relaxed agent-scope
// add on one of 16 arrival stripes (own 128-byte lines: 512 simultaneous arrivals on ONE word serialise at the
// memory side); who completes a stripe adds to the role's counter, who completes that raises 64 go flags (own lines);
// a consumer workgroup polls go flag (index % 64) with one lane, s_sleep between polls (first version: every waiting
// workgroup polled the word next to the attention tickets — the pollers starved the tickets, +17 us per layer), then
// barrier -> sc1 loads (MI355X_MICROARCH.md, valid hand-off forms: the one the split merge above uses).  The wait is
// bounded: after ~0.5 s a workgroup sets the gave-up word and carries on (wrong numbers the host reports, not a hung
// queue).  The caller zero-fills the flag region before every launch (one memset per token for all layers).
struct DecRoleArgs {
  DecGemvArgs g;
  int nblocks;
};
struct DecLayerArgs {
  DecAttnArgs a;
  DecRoleArgs r[3];     // o_proj, gate|up, down
  int nroles;           // 1: attention + o_proj; 3: + the MLP
  int* flags;
};

__device__ __forceinline__ void dec_chain_wait(int* flags, const int boundary, const int wg) {
  if (threadIdx.x == 0) {
    const int* go = flags + (40 + boundary * DEC_GO + (wg & (DEC_GO - 1))) * DEC_LINE;
    int spins = 0;
    while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(16);
      if (++spins > (1 << 20)) {
        __hip_atomic_store(flags + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  asm volatile("" ::: "memory");
}

// thread 0 of a finished workgroup of role r (n workgroups): true for the one arrival that completes the role
__device__ __forceinline__ bool dec_chain_arrive(int* flags, const int r, const int wg, const int n) {
  const int st = wg & (DEC_STRIPES - 1);
  const int in_stripe = (n - st + DEC_STRIPES - 1) / DEC_STRIPES;
  if (__hip_atomic_fetch_add(flags + (4 + DEC_STRIPES * r + st) * DEC_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != in_stripe - 1) return false;
  const int nst = min(n, DEC_STRIPES);
  return __hip_atomic_fetch_add(flags + (36 + r) * DEC_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nst - 1;
}

// dec_gemv_store for a role: the residual row was written by another role of this launch (sc1 loads: this XCD's L2 may hold the line from an earlier
// launch); AGST: somebody in this launch reads y (sc1 stores)
template <typename T, typename TO, bool GLU, bool AGST>
__device__ __forceinline__ void dec_gemv_store_ag(const DecGemvArgs& p, int pi, float a0, float a1) {
  const int n0 = GLU ? pi : 2 * pi, n1 = GLU ? p.N + pi : 2 * pi + 1;
  TO* y = (TO*)p.y;
  const TO* R = (const TO*)p.R;
  auto put = [&](int n, float v) {
    if (R) v += ld_agent_elt<TO>(R + n);
    if constexpr (AGST) st_agent_elt<TO>(y + n, v);
    else vg_elt<TO>::st(y + n, v);
  };
  if constexpr (GLU) {
    float g = a0, u = a1;
    if (sizeof(T) == 2) { g = bf2f(f2bf(g)); u = bf2f(f2bf(u)); }
    g = vg_silu(g);
    if (sizeof(T) == 2) g = bf2f(f2bf(g));
    put(n0, g * u);
  } else {
    put(n0, a0);
    if (n1 < p.N) put(n1, a1);
  }
}

// decode_gemv_fast_kernel's body as role `role` (0..2): weights first, wait on boundary `role`, x by sc1 loads; `release`: raise boundary role + 1
template <typename T, typename TO, bool GLU, bool NORM, int NB, int CPB>
__device__ __forceinline__ void dec_gemv_role(const DecGemvArgs& p, const int wg, const int nwg, char* dec_smem, float* red, float (*res)[DEC_MAX_PPW][2],
                                               int* flags, const int role, const bool release, int& bcast) {
  constexpr int KPC = 16 / sizeof(T);
  constexpr int NWV = KPC / 4;
  constexpr int NCH = NB * CPB * 64;
  constexpr int XN = (NCH + 255) / 256;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t* xs = (u32x4_t*)dec_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npair = GLU ? p.N : (p.N + 1) / 2;
  const int gw = wg * 4 + wave;
  const T* W = (const T*)p.W;
  const int p0 = gw * p.ppw;
  const int np = max(min(p0 + p.ppw, npair) - p0, 0);
  const int total = np * NB;

  // ---- 1. the weights (and the norm weights: constants of the layer) before the input row exists
  int ipi = p0, icb = 0;
  u32x4_t va0[CPB], va1[CPB], vb0[CPB], vb1[CPB];
  auto issue = [&](u32x4_t (&v0)[CPB], u32x4_t (&v1)[CPB]) {
    const int pc = min(ipi, npair - 1);
    const int n0 = GLU ? pc : 2 * pc;
    const int n1 = GLU ? p.N + pc : min(2 * pc + 1, p.N - 1);
    const u32x4_t* w0 = (const u32x4_t*)(W + (int64_t)n0 * p.ldw) + icb * (64 * CPB) + lane;
    const u32x4_t* w1 = (const u32x4_t*)(W + (int64_t)n1 * p.ldw) + icb * (64 * CPB) + lane;
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      v0[u] = __builtin_nontemporal_load(w0 + u * 64);
      v1[u] = __builtin_nontemporal_load(w1 + u * 64);
    }
    if (++icb == NB) { icb = 0; ++ipi; }
  };
  issue(va0, va1);
  issue(vb0, vb1);
  f32x4_t nwr[NORM ? XN : 1][NWV];
  if constexpr (NORM) {
#pragma unroll
    for (int i = 0; i < XN; ++i)
#pragma unroll
      for (int j = 0; j < NWV; ++j) nwr[i][j] = ((const f32x4_t*)p.nw)[min(tid + 256 * i, NCH - 1) * NWV + j];
  }

  // ---- 2. the producer role's row
  dec_chain_wait(flags, role, wg);
  u32x4_t xr[XN];
#pragma unroll
  for (int i = 0; i < XN; ++i) {
    const uint32_t* src = (const uint32_t*)p.x + (int64_t)min(tid + 256 * i, NCH - 1) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) xr[i][j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  {
    float rstd = 1.f;
    if constexpr (NORM) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        float f[KPC];
        dec_unpack<T>(xr[i], f);
        if (tid + 256 * i < NCH) {
#pragma unroll
          for (int e = 0; e < KPC; ++e) ss += f[e] * f[e];
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) red[wave] = ss;
      __syncthreads();
      rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)p.K + p.eps);
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        float f[KPC];
        dec_unpack<T>(xr[i], f);
#pragma unroll
        for (int e = 0; e < KPC; ++e) f[e] = dec_round<T>(f[e] * rstd) * nwr[i][e / 4][e % 4];
        xr[i] = dec_pack<T>(f);
      }
    }
#pragma unroll
    for (int i = 0; i < XN; ++i)
      if (tid + 256 * i < NCH) xs[tid + 256 * i] = xr[i];
    __syncthreads();
  }

  // ---- 3. the stream (decode_gemv_fast_kernel's)
  float a0 = 0.f, a1 = 0.f;
  int cpl = 0, ccb = 0;
  auto consume = [&](const u32x4_t (&v0)[CPB], const u32x4_t (&v1)[CPB]) {
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      const u32x4_t xv = xs[ccb * (64 * CPB) + u * 64 + lane];
      a0 += dec_dot<T>(v0[u], xv);
      a1 += dec_dot<T>(v1[u], xv);
    }
    if (++ccb == NB) {
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      if (lane == 0) { res[wave][cpl][0] = a0; res[wave][cpl][1] = a1; }
      a0 = 0.f;
      a1 = 0.f;
      ccb = 0;
      ++cpl;
    }
  };
  int b = 0;
  for (; b + 4 <= total; b += 2) {
    consume(va0, va1);
    issue(va0, va1);
    consume(vb0, vb1);
    issue(vb0, vb1);
  }
  const int rem = total - b;
  if (rem == 3) {
    consume(va0, va1);
    issue(va0, va1);
    consume(vb0, vb1);
    consume(va0, va1);
  } else if (rem == 2) {
    consume(va0, va1);
    consume(vb0, vb1);
  } else if (rem == 1) {
    consume(va0, va1);
  }
  // ---- 4. epilogue; a role somebody waits for leaves write-through and arrives
  if (!release) {
    for (int i = lane; i < np; i += 64) dec_gemv_store_ag<T, TO, GLU, false>(p, p0 + i, res[wave][i][0], res[wave][i][1]);
    return;
  }
  for (int i = lane; i < np; i += 64) dec_gemv_store_ag<T, TO, GLU, true>(p, p0 + i, res[wave][i][0], res[wave][i][1]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) bcast = dec_chain_arrive(flags, role, wg, nwg) ? 1 : 0;
  __syncthreads();
  if (bcast && tid < DEC_GO) __hip_atomic_store(flags + (40 + (role + 1) * DEC_GO + tid) * DEC_LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NBH / CPBH: batches x chunks of a hidden-size row (o_proj, gate|up); NBI: batches of an intermediate-size row (down_proj), 0 = no MLP roles
template <typename T, int G, int DT, int NBH, int CPBH, int NBI>
__global__ __launch_bounds__(256, 2) void decode_layer_kernel(DecLayerArgs q) {
  extern __shared__ __attribute__((aligned(16))) char dec_smem[];
  __shared__ int ticket;
  __shared__ float red[4];
  __shared__ float res[4][DEC_MAX_PPW][2];
  int w = blockIdx.x;
  const int nattn = q.a.nsplit * q.a.Hkv;
  if (w < nattn) {
    dec_attn_body<T, G, DT, true>(q.a, w % q.a.nsplit, w / q.a.nsplit, dec_smem, ticket, q.flags);
    return;
  }
  w -= nattn;
  if (w < q.r[0].nblocks) {
    dec_gemv_role<T, T, false, false, NBH, CPBH>(q.r[0].g, w, q.r[0].nblocks, dec_smem, red, res, q.flags, 0, q.nroles > 1, ticket);
    return;
  }
  if constexpr (NBI > 0) {
    w -= q.r[0].nblocks;
    if (w < q.r[1].nblocks) {
      dec_gemv_role<T, T, true, true, NBH, CPBH>(q.r[1].g, w, q.r[1].nblocks, dec_smem, red, res, q.flags, 1, true, ticket);
      return;
    }
    w -= q.r[1].nblocks;
    dec_gemv_role<T, T, false, false, NBI, 4>(q.r[2].g, w, q.r[2].nblocks, dec_smem, red, res, q.flags, 2, false, ticket);
  }
}

static int dec_role_blocks(int npair, int* ppw_out) {     // launch_decode_gemv's split
  constexpr int bpc = 2;
  const int maxw = 256 * bpc * 4;
  int ppw = (npair + maxw - 1) / maxw;
  for (int c = ppw; c <= 2 * ppw; ++c)
    if (((npair + 4 * c - 1) / (4 * c)) % 256 == 0 && npair % (4 * c) == 0) { ppw = c; break; }
  if (ppw > DEC_MAX_PPW) ppw = DEC_MAX_PPW;
  *ppw_out = ppw;
  return (npair + 4 * ppw - 1) / (4 * ppw);
}

// 0: not available for this shape; 1: attention + o_proj; 3: + the MLP roles
extern "C" int vg_decode_layer_roles(int H, int Hkv, int D, int hidden, int inter, int dtype) {
  if (H <= 0 || Hkv <= 0 || H % Hkv || (int64_t)H * D != hidden) return 0;
  if (dtype == VG_BF16 && H / Hkv == 4 && D == 128 && hidden == 4096) return inter == 14336 ? 3 : 1;   // Llama-3-8B widths
  if (dtype == VG_F32 && H / Hkv == 4 && D == 128 && hidden == 4096) return 1;                          // fp32 mode of the same
  if (dtype == VG_BF16 && H == Hkv && D == 96 && hidden == 3072) return 1;                              // Phi-3-mini
  return 0;
}
extern "C" int64_t vg_decode_layer_flag_ints(void) { return DEC_CHAIN_INTS; }

template <typename T, int G, int DT, int NBH, int CPBH, int NBI>
static int launch_decode_layer(const DecLayerArgs& q, int blocks, size_t lds, hipStream_t st) {
  static size_t lds_cap = 64 * 1024;
  if (lds > lds_cap) {
    (void)hipFuncSetAttribute((const void*)decode_layer_kernel<T, G, DT, NBH, CPBH, NBI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_cap = lds;
  }
  decode_layer_kernel<T, G, DT, NBH, CPBH, NBI><<<blocks, 256, lds, st>>>(q);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_decode_layer(const void* qkv, void* k_cache, void* v_cache, const float* cos, const float* sin, void* attn_out,
                               int H, int Hkv, int D, int max_len, int window, float scale, const int* pos_dev,
                               float* workspace, int64_t ws_floats, int32_t* flags,
                               const void* Wo, int64_t ldo, const void* resid, void* y_o,
                               const float* norm_w, float eps, const void* Wgu, int64_t ldgu, void* act,
                               const void* Wdown, int64_t lddown, void* y, int hidden, int inter, int dtype, vg_stream_t stream) {
  VG_CHECK(qkv && k_cache && v_cache && cos && sin && attn_out && pos_dev && workspace && flags && Wo && resid && y_o, VG_ERR_ARG, "vg_decode_layer: null pointer");
  const int roles = vg_decode_layer_roles(H, Hkv, D, hidden, inter, dtype);
  const int want = Wgu ? 3 : 1;
  VG_CHECK(roles >= want, VG_ERR_UNSUPPORTED, "vg_decode_layer: H=%d Hkv=%d D=%d hidden=%d inter=%d dtype=%d: %d role(s) available, %d asked for",
           H, Hkv, D, hidden, inter, dtype, roles, want);
  VG_CHECK(want == 1 || (norm_w && Wdown && act && y), VG_ERR_ARG, "vg_decode_layer: the MLP roles need norm_w, Wgu, Wdown, act, y");
  VG_CHECK(window >= 0 && max_len > 0, VG_ERR_ARG, "vg_decode_layer: bad window / max_len");
  const int es = dtype == VG_BF16 ? 2 : 4, kpc = 16 / es;
  VG_CHECK(ldo % kpc == 0 && (!Wgu || (ldgu % kpc == 0 && lddown % kpc == 0)), VG_ERR_ARG, "vg_decode_layer: weight row strides must be multiples of %d", kpc);
  VG_CHECK((((uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)Wo | (uintptr_t)Wgu | (uintptr_t)Wdown | (uintptr_t)norm_w | (uintptr_t)attn_out |
             (uintptr_t)y_o | (uintptr_t)act) & 15) == 0 && ((uintptr_t)flags & 127) == 0, VG_ERR_ARG, "vg_decode_layer: alignment (16 bytes; flags 128)");
  const int64_t need = vg_decode_attention_ws_floats(H, Hkv, D, max_len);
  VG_CHECK(ws_floats >= need, VG_ERR_ARG, "vg_decode_layer: workspace %lld < %lld floats", (long long)ws_floats, (long long)need);
  const int nsplit = (max_len + 63) / 64;
  VG_CHECK(nsplit <= 128, VG_ERR_UNSUPPORTED, "vg_decode_layer: max_len %d > 8192", max_len);
  DecLayerArgs q{};
  q.a = DecAttnArgs{qkv, k_cache, v_cache, cos, sin, attn_out, workspace, (int*)(workspace + (need - Hkv)), pos_dev, H, Hkv, D, nsplit, scale, window};
  q.flags = flags;
  q.nroles = want;
  int ppw = 1;
  q.r[0].nblocks = dec_role_blocks((hidden + 1) / 2, &ppw);
  q.r[0].g = DecGemvArgs{attn_out, Wo, y_o, nullptr, resid, hidden, hidden, ldo, 0.f, ppw, nullptr};
  int blocks = nsplit * Hkv + q.r[0].nblocks;
  if (want == 3) {
    q.r[1].nblocks = dec_role_blocks(inter, &ppw);
    q.r[1].g = DecGemvArgs{y_o, Wgu, act, norm_w, nullptr, inter, hidden, ldgu, eps, ppw, nullptr};
    q.r[2].nblocks = dec_role_blocks((hidden + 1) / 2, &ppw);
    q.r[2].g = DecGemvArgs{act, Wdown, y, nullptr, y_o, hidden, inter, lddown, 0.f, ppw, nullptr};
    blocks += q.r[1].nblocks + q.r[2].nblocks;
  }
  const int G = H / Hkv, KP = 256 / (D / kpc);
  size_t lds = sizeof(float) * ((size_t)G * D + 2 * D + 4 * G * 64 + G * 64 + 2 * G + 4 + (size_t)KP * G * D);
  const size_t xbytes = (size_t)(want == 3 ? inter : hidden) * es;
  if (xbytes > lds) lds = xbytes;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VG_BF16 && D == 128) {
    if (want == 3) return launch_decode_layer<bf16_t, 4, 128, 2, 4, 7>(q, blocks, lds, st);
    return launch_decode_layer<bf16_t, 4, 128, 2, 4, 0>(q, blocks, lds, st);
  }
  if (dtype == VG_BF16) return launch_decode_layer<bf16_t, 1, 96, 3, 2, 0>(q, blocks, lds, st);
  return launch_decode_layer<float, 4, 128, 4, 4, 0>(q, blocks, lds, st);
}
