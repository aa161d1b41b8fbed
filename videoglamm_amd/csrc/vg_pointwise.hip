// Pointwise / gather kernels (HBM-bound): grid-stride, coalesced, dtype-erased scalar accessors.
#include "vg_common.h"
#include <math.h>

static inline dim3 pw_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;  // 256 CUs x 16 resident workgroups, grid-stride the rest
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}
#define PW_LOOP(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

__global__ __launch_bounds__(256) void axpby_kernel(const void* a, const void* b, void* out, int64_t n, float alpha,
                                                    float beta, int64_t bp, int adt, int bdt, int odt) {
  PW_LOOP(i, n) {
    const float av = ld_any(a, i, adt);
    const float bv = b ? ld_any(b, i % bp, bdt) : 1.f;
    st_any(out, i, odt, alpha * av + beta * bv);
  }
}
// bf16 a / out, 8 elements (16 bytes) per thread; b bf16 or fp32 with a period that is a multiple of 8 (position embeddings, residuals)
template <bool BF32>
__global__ __launch_bounds__(256) void axpby_vec_kernel(const bf16_t* a, const void* b, bf16_t* out, int64_t n8, float alpha, float beta, int64_t bp8) {
  PW_LOOP(i, n8) {
    const u32x4_t av = *(const u32x4_t*)(a + i * 8);
    float bv[8];
    const int64_t j = i % bp8;
    if constexpr (BF32) {
      const f32x4_t b0 = *(const f32x4_t*)((const float*)b + j * 8), b1 = *(const f32x4_t*)((const float*)b + j * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
    } else {
      const u32x4_t bw = *(const u32x4_t*)((const bf16_t*)b + j * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bv[2 * e] = __uint_as_float(bw[e] << 16); bv[2 * e + 1] = __uint_as_float(bw[e] & 0xffff0000u); }
    }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a0 = __uint_as_float(av[e] << 16), a1 = __uint_as_float(av[e] & 0xffff0000u);
      o[e] = f2bf2(alpha * a0 + beta * bv[2 * e], alpha * a1 + beta * bv[2 * e + 1]);
    }
    *(u32x4_t*)(out + i * 8) = o;
  }
}
extern "C" int vg_axpby(const void* a, const void* b, void* out, int64_t n, float alpha, float beta,
                        int64_t b_period, int a_dtype, int b_dtype, int out_dtype, vg_stream_t stream) {
  VG_CHECK(a && out && n >= 0 && (!b || b_period > 0), VG_ERR_ARG, "vg_axpby: bad args");
  if (n == 0) return VG_OK;
  if (b && a_dtype == VG_BF16 && out_dtype == VG_BF16 && n % 8 == 0 && b_period % 8 == 0 &&
      (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0) {
    if (b_dtype == VG_F32) axpby_vec_kernel<true><<<pw_grid(n / 8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)a, b, (bf16_t*)out, n / 8, alpha, beta, b_period / 8);
    else axpby_vec_kernel<false><<<pw_grid(n / 8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)a, b, (bf16_t*)out, n / 8, alpha, beta, b_period / 8);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  axpby_kernel<<<pw_grid(n), 256, 0, (hipStream_t)stream>>>(a, b, out, n, alpha, beta, b ? b_period : 1, a_dtype, b_dtype, out_dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void act_kernel(const void* x, void* y, int64_t n, int act, int idt, int odt) {
  PW_LOOP(i, n) st_any(y, i, odt, vg_act(ld_any(x, i, idt), act));
}
extern "C" int vg_activation(const void* x, void* y, int64_t n, int act, int in_dtype, int out_dtype, vg_stream_t stream) {
  VG_CHECK(x && y && n >= 0, VG_ERR_ARG, "vg_activation: bad args");
  if (n == 0) return VG_OK;
  act_kernel<<<pw_grid(n), 256, 0, (hipStream_t)stream>>>(x, y, n, act, in_dtype, out_dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void swiglu_kernel(const void* gu, void* y, int64_t M, int F, int dt) {
  const int64_t n = M * F;
  PW_LOOP(i, n) {
    const int64_t m = i / F;
    const int f = (int)(i - m * F);
    float g = ld_any(gu, m * 2 * F + f, dt);
    const float u = ld_any(gu, m * 2 * F + F + f, dt);
    g = vg_silu(g);
    if (dt == VG_BF16) g = bf2f(f2bf(g));  // HF: act(gate) materialised in bf16 before the product
    st_any(y, i, dt, g * u);
  }
}
extern "C" int vg_swiglu(const void* gu, void* y, int64_t M, int F, int dtype, vg_stream_t stream) {
  VG_CHECK(gu && y && M >= 0 && F > 0, VG_ERR_ARG, "vg_swiglu: bad args");
  if (M == 0) return VG_OK;
  swiglu_kernel<<<pw_grid(M * F), 256, 0, (hipStream_t)stream>>>(gu, y, M, F, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void cast_kernel(const void* in, void* out, int64_t n, int idt, int odt) {
  PW_LOOP(i, n) st_any(out, i, odt, ld_any(in, i, idt));
}
extern "C" int vg_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, vg_stream_t stream) {
  VG_CHECK(in && out && n >= 0, VG_ERR_ARG, "vg_cast: bad args");
  if (n == 0) return VG_OK;
  cast_kernel<<<pw_grid(n), 256, 0, (hipStream_t)stream>>>(in, out, n, in_dtype, out_dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void where_rows_kernel(const float* cond, const void* a, const void* b, void* out,
                                                         int64_t rows, int64_t inner, int64_t bp, float fill, int dt, int64_t ld_out) {
  const int64_t n = rows * inner;
  PW_LOOP(i, n) {
    const int64_t r = i / inner, c = i - r * inner;
    float v;
    if (cond[r] > 0.f) v = ld_any(a, i, dt);
    else v = b ? ld_any(b, c % bp, dt) : fill;
    st_any(out, r * ld_out + c, dt, v);
  }
}
extern "C" int vg_where_rows(const float* cond, const void* a, const void* b, void* out, int64_t rows, int64_t inner,
                             int64_t b_period, float fill, int64_t ld_out, int dtype, vg_stream_t stream) {
  VG_CHECK(cond && a && out && rows >= 0 && inner > 0 && (ld_out == 0 || ld_out >= inner), VG_ERR_ARG, "vg_where_rows: bad args");
  if (rows == 0) return VG_OK;
  where_rows_kernel<<<pw_grid(rows * inner), 256, 0, (hipStream_t)stream>>>(cond, a, b, out, rows, inner, b ? b_period : 1, fill, dtype, ld_out ? ld_out : inner);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void mask_for_mem_kernel(const float* x, void* out, int64_t n, int binarize, float scale,
                                                           float bias, int odt) {
  PW_LOOP(i, n) {
    const float v = x[i];
    // torch.sigmoid in fp32 (sam2_base.py:689) — use the accurate expf here, the value feeds a recurrence
    const float m = binarize ? (v > 0.f ? 1.f : 0.f) : 1.0f / (1.0f + expf(-v));
    st_any(out, i, odt, m * scale + bias);
  }
}
extern "C" int vg_mask_for_mem(const float* x, void* out, int64_t n, int binarize, float scale, float bias,
                               int out_dtype, vg_stream_t stream) {
  VG_CHECK(x && out && n >= 0, VG_ERR_ARG, "vg_mask_for_mem: bad args");
  if (n == 0) return VG_OK;
  mask_for_mem_kernel<<<pw_grid(n), 256, 0, (hipStream_t)stream>>>(x, out, n, binarize, scale, bias, out_dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void threshold_kernel(const float* x, uint8_t* out, int64_t n) {
  PW_LOOP(i, n) out[i] = x[i] > 0.f ? 1 : 0;
}
extern "C" int vg_threshold(const float* x, uint8_t* out, int64_t n, vg_stream_t stream) {
  VG_CHECK(x && out && n >= 0, VG_ERR_ARG, "vg_threshold: bad args");
  if (n == 0) return VG_OK;
  threshold_kernel<<<pw_grid(n), 256, 0, (hipStream_t)stream>>>(x, out, n);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void rope_half_kernel(void* x, int64_t x_ss, int64_t x_sh, const float* cs, const float* sn,
                                                        int S, int H, int D, int pos0, int dt) {
  const int hd = D / 2;
  const int64_t n = (int64_t)S * H * hd;
  PW_LOOP(i, n) {
    const int d = (int)(i % hd);
    const int64_t t = i / hd;
    const int hh = (int)(t % H);
    const int s = (int)(t / H);
    const int64_t base = (int64_t)s * x_ss + (int64_t)hh * x_sh;
    const float c = cs[(int64_t)(pos0 + s) * hd + d], sv = sn[(int64_t)(pos0 + s) * hd + d];
    const float x1 = ld_any(x, base + d, dt), x2 = ld_any(x, base + d + hd, dt);
    float o1, o2;
    if (dt == VG_BF16) {
      // HF computes q*cos + rotate_half(q)*sin with every product rounded to the tensor dtype
      const float cb = bf2f(f2bf(c)), sb = bf2f(f2bf(sv));
      o1 = bf2f(f2bf(x1 * cb)) + bf2f(f2bf(-x2 * sb));
      o2 = bf2f(f2bf(x2 * cb)) + bf2f(f2bf(x1 * sb));
    } else {
      o1 = x1 * c - x2 * sv;
      o2 = x2 * c + x1 * sv;
    }
    st_any(x, base + d, dt, o1);
    st_any(x, base + d + hd, dt, o2);
  }
}
extern "C" int vg_rope_half(void* x, int64_t x_ss, int64_t x_sh, const float* cos, const float* sin, int S, int H, int D,
                            int pos0, int dtype, vg_stream_t stream) {
  VG_CHECK(x && cos && sin && S >= 0 && H > 0 && D > 0 && D % 2 == 0, VG_ERR_ARG, "vg_rope_half: bad args");
  if (S == 0) return VG_OK;
  rope_half_kernel<<<pw_grid((int64_t)S * H * D / 2), 256, 0, (hipStream_t)stream>>>(x, x_ss, x_sh, cos, sin, S, H, D, pos0, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void rope_axial_kernel(void* x, const float* cs, const float* sn, int B, int N, int C,
                                                         int n_rope, int n_grid, int dt) {
  const int hc = C / 2;
  const int64_t n = (int64_t)B * n_rope * hc;
  PW_LOOP(i, n) {
    const int pr = (int)(i % hc);
    const int64_t t = i / hc;
    const int tok = (int)(t % n_rope);
    const int b = (int)(t / n_rope);
    const int64_t base = ((int64_t)b * N + tok) * C + 2 * pr;
    const int g = tok % n_grid;
    const float c = cs[(int64_t)g * hc + pr], s = sn[(int64_t)g * hc + pr];
    const float a = ld_any(x, base, dt), bb = ld_any(x, base + 1, dt);
    st_any(x, base, dt, a * c - bb * s);
    st_any(x, base + 1, dt, a * s + bb * c);
  }
}
// bf16, four complex pairs (16 bytes) per thread, rows of ld elements holding H heads of Ch channels each (one table for every head): the q | k
// columns of a fused q|k|v projection are rotated in ONE launch, in place, as a strided view (r04).  Same arithmetic per pair as the scalar kernel
// (a c - b s, a s + b c in fp32, one rounding), so the two agree bit for bit.
__global__ __launch_bounds__(256) void rope_axial_vec_kernel(bf16_t* x, int64_t ld, int64_t sb, const float* __restrict__ cs, const float* __restrict__ sn,
                                                             int B, int H, int Ch, int n_rope, int n_grid) {
  const int q8 = Ch / 8, hc = Ch / 2;
  const int64_t n = (int64_t)B * n_rope * H * q8;
  PW_LOOP(i, n) {
    const int c8 = (int)(i % q8);
    int64_t t = i / q8;
    const int h = (int)(t % H); t /= H;
    const int tok = (int)(t % n_rope);
    const int b = (int)(t / n_rope);
    bf16_t* px = x + (int64_t)b * sb + (int64_t)tok * ld + h * Ch + c8 * 8;
    const int g = tok % n_grid;
    const f32x4_t c = *(const f32x4_t*)(cs + (int64_t)g * hc + c8 * 4), sv = *(const f32x4_t*)(sn + (int64_t)g * hc + c8 * 4);
    u32x4_t v = *(const u32x4_t*)px;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = __uint_as_float(v[e] << 16), bb = __uint_as_float(v[e] & 0xffff0000u);
      v[e] = (uint32_t)f2bf(a * c[e] - bb * sv[e]) | ((uint32_t)f2bf(a * sv[e] + bb * c[e]) << 16);
    }
    *(u32x4_t*)px = v;
  }
}
extern "C" int vg_rope_axial_heads(void* x, int64_t ld, int64_t sb, const float* cos, const float* sin, int B, int H, int Ch, int n_rope, int n_grid,
                                   int dtype, vg_stream_t stream) {
  VG_CHECK(x && cos && sin && B > 0 && H > 0 && Ch > 0 && Ch % 8 == 0 && n_rope >= 0 && n_grid > 0 && ld >= (int64_t)H * Ch && ld % 8 == 0 && sb % 8 == 0 &&
               ((uintptr_t)x & 15) == 0 && dtype == VG_BF16, VG_ERR_ARG, "vg_rope_axial_heads: bad args (bf16, 16-byte aligned rows, Ch a multiple of 8)");
  if (n_rope == 0) return VG_OK;
  rope_axial_vec_kernel<<<pw_grid((int64_t)B * n_rope * H * (Ch / 8)), 256, 0, (hipStream_t)stream>>>((bf16_t*)x, ld, sb, cos, sin, B, H, Ch, n_rope, n_grid);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
extern "C" int vg_rope_axial(void* x, const float* cos, const float* sin, int B, int N, int C, int n_rope, int n_grid,
                             int dtype, vg_stream_t stream) {
  VG_CHECK(x && cos && sin && B > 0 && N > 0 && C % 2 == 0 && n_rope >= 0 && n_rope <= N && n_grid > 0, VG_ERR_ARG,
           "vg_rope_axial: bad args");
  if (n_rope == 0) return VG_OK;
  if (dtype == VG_BF16 && C % 8 == 0 && ((uintptr_t)x & 15) == 0)
    return vg_rope_axial_heads(x, C, (int64_t)N * C, cos, sin, B, 1, C, n_rope, n_grid, dtype, stream);
  rope_axial_kernel<<<pw_grid((int64_t)B * n_rope * C / 2), 256, 0, (hipStream_t)stream>>>(x, cos, sin, B, N, C, n_rope, n_grid, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void embed_kernel(const int64_t* ids, const void* table, void* out, int64_t n, int D, int dt) {
  const int64_t tot = n * D;
  PW_LOOP(i, tot) {
    const int64_t r = i / D;
    const int c = (int)(i - r * D);
    st_any(out, i, dt, ld_any(table, ids[r] * D + c, dt));
  }
}
extern "C" int vg_embed(const int64_t* ids, const void* table, void* out, int64_t n, int D, int dtype, vg_stream_t stream) {
  VG_CHECK(ids && table && out && n >= 0 && D > 0, VG_ERR_ARG, "vg_embed: bad args");
  if (n == 0) return VG_OK;
  embed_kernel<<<pw_grid(n * D), 256, 0, (hipStream_t)stream>>>(ids, table, out, n, D, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// Fused RoPE + KV-cache append for one LLM layer.  qkv: [S, (H + 2*Hkv)*D] (fused projection output, row stride
// ld).  q is rotated in place; k is rotated and written to k_cache[pos + s]; v is copied to v_cache[pos + s].
// pos = pos_dev ? *pos_dev : pos0 — reading the position from device memory keeps a decode step replayable
// inside a HIP graph.
__global__ __launch_bounds__(256) void rope_kv_append_kernel(void* qkv, int64_t ld, void* kc, void* vc, const float* cs,
                                                             const float* sn, int S, int H, int Hkv, int D, int pos0,
                                                             const int* pos_dev, int dt) {
  const int pos = pos_dev ? *pos_dev : pos0;
  const int hd = D / 2;
  const int HT = H + 2 * Hkv;
  const int64_t n = (int64_t)S * HT * hd;
  PW_LOOP(i, n) {
    const int d = (int)(i % hd);
    const int64_t t = i / hd;
    const int hh = (int)(t % HT);
    const int s = (int)(t / HT);
    const int64_t src = (int64_t)s * ld + (int64_t)hh * D;
    const float x1 = ld_any(qkv, src + d, dt), x2 = ld_any(qkv, src + d + hd, dt);
    if (hh >= H + Hkv) {  // v head: plain copy into the cache
      const int64_t dst = ((int64_t)(pos + s) * Hkv + (hh - H - Hkv)) * D;
      st_any(vc, dst + d, dt, x1);
      st_any(vc, dst + d + hd, dt, x2);
      continue;
    }
    const float c = cs[(int64_t)(pos + s) * hd + d], sv = sn[(int64_t)(pos + s) * hd + d];
    float o1, o2;
    if (dt == VG_BF16) {
      const float cb = bf2f(f2bf(c)), sb = bf2f(f2bf(sv));
      o1 = bf2f(f2bf(x1 * cb)) + bf2f(f2bf(-x2 * sb));
      o2 = bf2f(f2bf(x2 * cb)) + bf2f(f2bf(x1 * sb));
    } else {
      o1 = x1 * c - x2 * sv;
      o2 = x2 * c + x1 * sv;
    }
    if (hh < H) {
      st_any(qkv, src + d, dt, o1);
      st_any(qkv, src + d + hd, dt, o2);
    } else {
      const int64_t dst = ((int64_t)(pos + s) * Hkv + (hh - H)) * D;
      st_any(kc, dst + d, dt, o1);
      st_any(kc, dst + d + hd, dt, o2);
    }
  }
}
// bf16, 16-byte path: a thread owns 8 consecutive dims of the first half of a head and the matching 8 of the second half (two
// 16-byte loads, two 16-byte stores; the scalar kernel above moves 2 bytes per access: 108 us per Llama layer of the C2 prefill)
__global__ __launch_bounds__(256) void rope_kv_append_vec_kernel(bf16_t* qkv, int64_t ld, bf16_t* kc, bf16_t* vc, const float* cs,
                                                                 const float* sn, int S, int H, int Hkv, int D, int pos0, const int* pos_dev) {
  const int pos = pos_dev ? *pos_dev : pos0;
  const int hd = D / 2, cpr = hd / 8;         // 16-byte chunks per half head
  const int HT = H + 2 * Hkv;
  const int64_t n = (int64_t)S * HT * cpr;
  PW_LOOP(i, n) {
    const int c = (int)(i % cpr);
    const int64_t t = i / cpr;
    const int hh = (int)(t % HT);
    const int s = (int)(t / HT);
    bf16_t* src = qkv + (int64_t)s * ld + (int64_t)hh * D + c * 8;
    const u32x4_t a = *(const u32x4_t*)src, b = *(const u32x4_t*)(src + hd);
    if (hh >= H + Hkv) {  // v head: plain copy into the cache
      bf16_t* dst = vc + ((int64_t)(pos + s) * Hkv + (hh - H - Hkv)) * D + c * 8;
      *(u32x4_t*)dst = a;
      *(u32x4_t*)(dst + hd) = b;
      continue;
    }
    const float* cp = cs + (int64_t)(pos + s) * hd + c * 8;
    const float* sp = sn + (int64_t)(pos + s) * hd + c * 8;
    const f32x4_t c0 = *(const f32x4_t*)cp, c1 = *(const f32x4_t*)(cp + 4), s0 = *(const f32x4_t*)sp, s1 = *(const f32x4_t*)(sp + 4);
    float o1[8], o2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x1 = __uint_as_float((e & 1) ? (a[e >> 1] & 0xffff0000u) : (a[e >> 1] << 16));
      const float x2 = __uint_as_float((e & 1) ? (b[e >> 1] & 0xffff0000u) : (b[e >> 1] << 16));
      const float cb = bf2f(f2bf(e < 4 ? c0[e] : c1[e - 4])), sb = bf2f(f2bf(e < 4 ? s0[e] : s1[e - 4]));
      o1[e] = bf2f(f2bf(x1 * cb)) + bf2f(f2bf(-x2 * sb));
      o2[e] = bf2f(f2bf(x2 * cb)) + bf2f(f2bf(x1 * sb));
    }
    u32x4_t r1, r2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r1[e] = f2bf2(o1[2 * e], o1[2 * e + 1]); r2[e] = f2bf2(o2[2 * e], o2[2 * e + 1]); }
    bf16_t* dst = hh < H ? src : kc + ((int64_t)(pos + s) * Hkv + (hh - H)) * D + c * 8;
    *(u32x4_t*)dst = r1;
    *(u32x4_t*)(dst + hd) = r2;
  }
}
extern "C" int vg_rope_kv_append(void* qkv, int64_t ld, void* k_cache, void* v_cache, const float* cos, const float* sin,
                                 int S, int H, int Hkv, int D, int pos0, const int* pos_dev, int dtype, vg_stream_t stream) {
  VG_CHECK(qkv && k_cache && v_cache && cos && sin && S >= 0 && H > 0 && Hkv > 0 && D > 0 && D % 2 == 0, VG_ERR_ARG,
           "vg_rope_kv_append: bad args");
  if (S == 0) return VG_OK;
  if (dtype == VG_BF16 && D % 16 == 0 && ld % 8 == 0 && S > 1 &&
      ((((uintptr_t)qkv | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)cos | (uintptr_t)sin) & 15) == 0)) {
    rope_kv_append_vec_kernel<<<pw_grid((int64_t)S * (H + 2 * Hkv) * D / 16), 256, 0, (hipStream_t)stream>>>(
        (bf16_t*)qkv, ld, (bf16_t*)k_cache, (bf16_t*)v_cache, cos, sin, S, H, Hkv, D, pos0, pos_dev);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  rope_kv_append_kernel<<<pw_grid((int64_t)S * (H + 2 * Hkv) * D / 2), 256, 0, (hipStream_t)stream>>>(
      qkv, ld, k_cache, v_cache, cos, sin, S, H, Hkv, D, pos0, pos_dev, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// dst[(*idx_dev + idx_off) * n + i] = src[i]  — row store at a device-resident index (graph-replayable)
__global__ __launch_bounds__(256) void store_row_kernel(const void* src, void* dst, int64_t n, const int* idx_dev, int idx_off, int dt) {
  const int64_t base = ((int64_t)(*idx_dev) + idx_off) * n;
  PW_LOOP(i, n) {
    if (dt == VG_BF16) ((bf16_t*)dst)[base + i] = ((const bf16_t*)src)[i];
    else ((float*)dst)[base + i] = ((const float*)src)[i];
  }
}
extern "C" int vg_store_row(const void* src, void* dst, int64_t n, const int* idx_dev, int idx_off, int dtype, vg_stream_t stream) {
  VG_CHECK(src && dst && idx_dev && n > 0, VG_ERR_ARG, "vg_store_row: bad args");
  store_row_kernel<<<pw_grid(n), 256, 0, (hipStream_t)stream>>>(src, dst, n, idx_dev, idx_off, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ void add_int_kernel(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += v; }
extern "C" int vg_add_int(int* p, int v, vg_stream_t stream) {
  VG_CHECK(p, VG_ERR_ARG, "vg_add_int: null");
  add_int_kernel<<<1, 64, 0, (hipStream_t)stream>>>(p, v);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
