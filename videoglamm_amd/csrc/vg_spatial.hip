// Spatial / layout kernels on NHWC tensors (HBM-bound; the channel dim is the coalesced one).
#include "vg_common.h"
#include <math.h>

static inline dim3 sp_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}
#define SP_LOOP(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

struct Perm5 { int64_t d[5]; int64_t s[5]; };
__global__ __launch_bounds__(256) void permute5_kernel(const void* in, void* out, Perm5 p, int dt) {
  const int64_t n = p.d[0] * p.d[1] * p.d[2] * p.d[3] * p.d[4];
  SP_LOOP(i, n) {
    int64_t r = i;
    const int64_t i4 = r % p.d[4]; r /= p.d[4];
    const int64_t i3 = r % p.d[3]; r /= p.d[3];
    const int64_t i2 = r % p.d[2]; r /= p.d[2];
    const int64_t i1 = r % p.d[1]; r /= p.d[1];
    const int64_t src = r * p.s[0] + i1 * p.s[1] + i2 * p.s[2] + i3 * p.s[3] + i4 * p.s[4];
    if (dt == VG_BF16) ((bf16_t*)out)[i] = ((const bf16_t*)in)[src];
    else ((float*)out)[i] = ((const float*)in)[src];
  }
}
extern "C" int vg_permute5(const void* in, void* out, const int64_t dims[5], const int64_t strides[5], int dtype,
                           vg_stream_t stream) {
  VG_CHECK(in && out && dims && strides, VG_ERR_ARG, "vg_permute5: null");
  Perm5 p;
  int64_t n = 1;
  for (int i = 0; i < 5; ++i) { p.d[i] = dims[i]; p.s[i] = strides[i]; n *= dims[i]; VG_CHECK(dims[i] >= 0, VG_ERR_ARG, "vg_permute5: negative dim"); }
  if (n == 0) return VG_OK;
  permute5_kernel<<<sp_grid(n), 256, 0, (hipStream_t)stream>>>(in, out, p, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void im2col_kernel(const void* x, void* cols, int B, int H, int W, int C, int kh, int kw,
                                                     int stride, int pad, int Ho, int Wo, int Kpad, int dt) {
  const int64_t n = (int64_t)B * Ho * Wo * Kpad;
  const int Kreal = kh * kw * C;
  SP_LOOP(i, n) {
    const int kk = (int)(i % Kpad);
    int64_t r = i / Kpad;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float v = 0.f;
    if (kk < Kreal) {
      const int c = kk % C;
      const int t = kk / C;
      const int kx = t % kw, ky = t / kw;
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld_any(x, (((int64_t)b * H + iy) * W + ix) * C + c, dt);
    }
    st_any(cols, i, dt, v);
  }
}
extern "C" int vg_im2col(const void* x, void* cols, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                         int Kpad, int dtype, vg_stream_t stream) {
  VG_CHECK(x && cols && B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && Kpad >= kh * kw * C,
           VG_ERR_ARG, "vg_im2col: bad args");
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  im2col_kernel<<<sp_grid((int64_t)B * Ho * Wo * Kpad), 256, 0, (hipStream_t)stream>>>(x, cols, B, H, W, C, kh, kw, stride, pad, Ho, Wo, Kpad, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void dwconv_kernel(const void* x, const float* w, const float* bias, void* y, int B, int H,
                                                     int W, int C, int k, int dt) {
  const int64_t n = (int64_t)B * H * W * C;
  const int pad = k / 2;
  SP_LOOP(i, n) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % W); r /= W;
    const int oy = (int)(r % H);
    const int b = (int)(r / H);
    float acc = bias ? bias[c] : 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy - pad + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox - pad + kx;
        if (ix < 0 || ix >= W) continue;
        acc = fmaf(ld_any(x, (((int64_t)b * H + iy) * W + ix) * C + c, dt), w[(ky * k + kx) * C + c], acc);
      }
    }
    st_any(y, i, dt, acc);
  }
}
// bf16, 8 channels x 4 consecutive output pixels per thread (r04): 16-byte loads, a loaded input pixel feeds up to four outputs and a row of taps is
// loaded once per strip — 42 load instructions per output instead of 147 two-byte ones.  The taps of an output are accumulated in the scalar
// kernel's order (ky, then kx, out-of-image taps skipped), so the two kernels agree bit for bit.  Measured on the memory encoder's fuser
// ([8, 64, 64, 256], k = 7; C4's clip on the video branch): 365 us per launch with the scalar kernel = 58 ms of a 907 ms clip.
template <int K>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           bf16_t* __restrict__ y, int B, int H, int W, int C) {
  constexpr int P = K / 2, SW = 4;
  const int CG = C / 8, WS = (W + SW - 1) / SW;
  const int64_t n = (int64_t)B * H * WS * CG;
  SP_LOOP(i, n) {
    const int cg = (int)(i % CG);
    int64_t r = i / CG;
    const int sx = (int)(r % WS); r /= WS;
    const int oy = (int)(r % H);
    const int b = (int)(r / H);
    const int c0 = cg * 8, ox0 = sx * SW;
    float acc[SW][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float bv = bias ? bias[c0 + e] : 0.f;
#pragma unroll
      for (int j = 0; j < SW; ++j) acc[j][e] = bv;
    }
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy - P + ky;
      if (iy < 0 || iy >= H) continue;
      float wk[K][8];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const f32x4_t w0 = *(const f32x4_t*)(w + (int64_t)(ky * K + kx) * C + c0), w1 = *(const f32x4_t*)(w + (int64_t)(ky * K + kx) * C + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wk[kx][e] = w0[e]; wk[kx][4 + e] = w1[e]; }
      }
      const bf16_t* row = x + (((int64_t)b * H + iy) * W) * C + c0;
#pragma unroll
      for (int t = 0; t < SW + K - 1; ++t) {
        const int ix = ox0 - P + t;
        if (ix < 0 || ix >= W) continue;
        const u32x4_t xv = *(const u32x4_t*)(row + (int64_t)ix * C);
        float xf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { xf[2 * e] = __uint_as_float(xv[e] << 16); xf[2 * e + 1] = __uint_as_float(xv[e] & 0xffff0000u); }
#pragma unroll
        for (int j = 0; j < SW; ++j) {
          const int kx = t - j;
          if (kx >= 0 && kx < K) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(xf[e], wk[kx][e], acc[j][e]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SW; ++j) {
      if (ox0 + j < W) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf2(acc[j][2 * e], acc[j][2 * e + 1]);
        *(u32x4_t*)(y + (((int64_t)b * H + oy) * W + ox0 + j) * C + c0) = o;
      }
    }
  }
}

extern "C" int vg_dwconv(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k,
                         int dtype, vg_stream_t stream) {
  VG_CHECK(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && (k & 1), VG_ERR_ARG, "vg_dwconv: bad args");
  if (dtype == VG_BF16 && k == 7 && C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15) == 0) {
    const int64_t n = (int64_t)B * H * ((W + 3) / 4) * (C / 8);
    dwconv_strip_kernel<7><<<sp_grid(n), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, w, bias, (bf16_t*)y, B, H, W, C);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  dwconv_kernel<<<sp_grid((int64_t)B * H * W * C), 256, 0, (hipStream_t)stream>>>(x, w, bias, y, B, H, W, C, k, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// SAM2 memory encoder, mask downsampler stages with few channels (R/modeling/memory_encoder.py:17-63: Conv2d(k = 3, stride 2, pad 1) -> LayerNorm2d
// -> GELU, the stages 1 -> 4 and 4 -> 16 channels; 16 -> 64 stays a GEMM: 9216 FMAs per output pixel are too many for a thread — that instantiation
// spilled 2.2 KB per lane and made the clip slower): ONE pass, a thread per output pixel with all COUT channels in registers — the conv's 9 CIN inputs against
// weights broadcast from LDS, then the channel LayerNorm and the GELU on the fp32 sums.  As im2col + GEMM + norm + activation (four launches on
// [N x 512^2, 4]-shaped operands: K = 9 padded to 16, N = 4 ...) these three stages were im2col 29 + norm 34 + GEMM / GELU ~20 ms of C4's 907 ms
// video-branch clip.  x [B, H, W, CIN] -> y [B, H/2, W/2, COUT] channels-last; w [COUT, ldw] in the tensors' dtype, column (ky * 3 + kx) * CIN + c.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3s2_ln_gelu_kernel(const void* __restrict__ x, const void* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                              const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, void* __restrict__ y,
                                                              int B, int H, int W, int dt) {
  __shared__ __attribute__((aligned(16))) float ws[9 * CIN * COUT];      // [k = tap * CIN + c][COUT]: a thread reads the COUT weights of input k as float4s
  for (int idx = threadIdx.x; idx < 9 * CIN * COUT; idx += 256) {
    const int o = idx % COUT, k = idx / COUT;
    ws[idx] = ld_any(w, (int64_t)o * ldw + k, dt);
  }
  __syncthreads();
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n = (int64_t)B * Ho * Wo;
  SP_LOOP(i, n) {
    const int ox = (int)(i % Wo);
    int64_t r = i / Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll 1
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const int64_t base = (((int64_t)b * H + iy) * W + ix) * CIN;
        const float* wk = ws + (ky * 3 + kx) * CIN * COUT;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float xv = ld_any(x, base + c, dt);
#pragma unroll
          for (int o4 = 0; o4 < COUT / 4; ++o4) {
            const f32x4_t wv = *(const f32x4_t*)(wk + c * COUT + o4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o4 * 4 + e] = fmaf(xv, wv[e], acc[o4 * 4 + e]);
          }
        }
      }
    }
    float mean = 0.f;
#pragma unroll
    for (int o = 0; o < COUT; ++o) mean += acc[o];
    mean *= 1.0f / COUT;
    float var = 0.f;
#pragma unroll
    for (int o = 0; o < COUT; ++o) { const float d = acc[o] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(var * (1.0f / COUT) + eps);
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = vg_act((acc[o] - mean) * rstd * lnw[o] + lnb[o], VG_ACT_GELU);
    if (dt == VG_BF16) {
      bf16_t* yo = (bf16_t*)y + i * COUT;
      if constexpr (COUT >= 8) {
#pragma unroll
        for (int q = 0; q < COUT / 8; ++q) {
          u32x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = f2bf2(acc[q * 8 + 2 * e], acc[q * 8 + 2 * e + 1]);
          *(u32x4_t*)(yo + q * 8) = v;
        }
      } else {
        uint2 v;
        v.x = f2bf2(acc[0], acc[1]);
        v.y = f2bf2(acc[2], acc[3]);
        *(uint2*)yo = v;
      }
    } else {
      float* yo = (float*)y + i * COUT;
#pragma unroll
      for (int q = 0; q < COUT / 4; ++q) {
        const f32x4_t v = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
        *(f32x4_t*)(yo + q * 4) = v;
      }
    }
  }
}
extern "C" int vg_conv3s2_ln_gelu(const void* x, const void* w, int64_t ldw, const float* bias, const float* ln_w, const float* ln_b, float eps, void* y,
                                  int B, int H, int W, int Cin, int Cout, int dtype, vg_stream_t stream) {
  VG_CHECK(x && w && ln_w && ln_b && y && B > 0 && H > 0 && W > 0, VG_ERR_ARG, "vg_conv3s2_ln_gelu: bad args");
  VG_CHECK(dtype == VG_F32 || dtype == VG_BF16, VG_ERR_ARG, "vg_conv3s2_ln_gelu: bad dtype %d", dtype);
  VG_CHECK(((uintptr_t)y & 15) == 0, VG_ERR_ARG, "vg_conv3s2_ln_gelu: y must be 16-byte aligned");
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 1 && Cout == 4) conv3s2_ln_gelu_kernel<1, 4><<<sp_grid(n), 256, 0, st>>>(x, w, (int)ldw, bias, ln_w, ln_b, eps, y, B, H, W, dtype);
  else if (Cin == 4 && Cout == 16) conv3s2_ln_gelu_kernel<4, 16><<<sp_grid(n), 256, 0, st>>>(x, w, (int)ldw, bias, ln_w, ln_b, eps, y, B, H, W, dtype);
  else {
    vg_set_error("vg_conv3s2_ln_gelu: built for the channel pairs 1->4 and 4->16 (got %d->%d)", Cin, Cout);
    return VG_ERR_UNSUPPORTED;
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const void* g, const float* bias, void* y, int B, int H, int W,
                                                             int C, int dt) {
  const int64_t n = (int64_t)B * 2 * H * 2 * W * C;
  SP_LOOP(i, n) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % (2 * W)); r /= (2 * W);
    const int oy = (int)(r % (2 * H));
    const int b = (int)(r / (2 * H));
    const int tap = (oy & 1) * 2 + (ox & 1);
    const int64_t src = ((((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * 4 + tap) * C + c;
    st_any(y, i, dt, ld_any(g, src, dt) + (bias ? bias[c] : 0.f));
  }
}
extern "C" int vg_pixel_shuffle2(const void* g, const float* bias, void* y, int B, int H, int W, int C, int dtype,
                                 vg_stream_t stream) {
  VG_CHECK(g && y && B > 0 && H > 0 && W > 0 && C > 0, VG_ERR_ARG, "vg_pixel_shuffle2: bad args");
  pixel_shuffle2_kernel<<<sp_grid((int64_t)B * 4 * H * W * C), 256, 0, (hipStream_t)stream>>>(g, bias, y, B, H, W, C, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void pool2_kernel(const void* x, void* y, int B, int H, int W, int C, int64_t xps,
                                                    int is_max, int dt) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t n = (int64_t)B * Ho * Wo * C;
  SP_LOOP(i, n) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int64_t p00 = ((int64_t)b * H + 2 * oy) * W + 2 * ox;
    const float v0 = ld_any(x, p00 * xps + c, dt), v1 = ld_any(x, (p00 + 1) * xps + c, dt);
    const float v2 = ld_any(x, (p00 + W) * xps + c, dt), v3 = ld_any(x, (p00 + W + 1) * xps + c, dt);
    const float o = is_max ? fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) : (v0 + v1 + v2 + v3) * 0.25f;
    st_any(y, i, dt, o);
  }
}
// bf16, 8 channels per thread: four 16-byte loads, one 16-byte store (the scalar kernel moved 2 bytes per load: 1.9 TB/s of the
// 755 MB a stage transition's shortcut pool touches at 16 frames).  Same arithmetic per element: max, or (v0 + v1 + v2 + v3) * 0.25.
__global__ __launch_bounds__(256) void pool2_vec8_kernel(const bf16_t* x, bf16_t* y, int B, int H, int W, int C8, int64_t xps, int is_max) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t n = (int64_t)B * Ho * Wo * C8;
  SP_LOOP(i, n) {
    const int c = (int)(i % C8) * 8;
    int64_t r = i / C8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int64_t p00 = ((int64_t)b * H + 2 * oy) * W + 2 * ox;
    const u32x4_t a0 = *(const u32x4_t*)(x + p00 * xps + c), a1 = *(const u32x4_t*)(x + (p00 + 1) * xps + c);
    const u32x4_t a2 = *(const u32x4_t*)(x + (p00 + W) * xps + c), a3 = *(const u32x4_t*)(x + (p00 + W + 1) * xps + c);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo[4] = {__uint_as_float(a0[e] << 16), __uint_as_float(a1[e] << 16), __uint_as_float(a2[e] << 16), __uint_as_float(a3[e] << 16)};
      float hi[4] = {__uint_as_float(a0[e] & 0xffff0000u), __uint_as_float(a1[e] & 0xffff0000u), __uint_as_float(a2[e] & 0xffff0000u),
                     __uint_as_float(a3[e] & 0xffff0000u)};
      const float l = is_max ? fmaxf(fmaxf(lo[0], lo[1]), fmaxf(lo[2], lo[3])) : (lo[0] + lo[1] + lo[2] + lo[3]) * 0.25f;
      const float h = is_max ? fmaxf(fmaxf(hi[0], hi[1]), fmaxf(hi[2], hi[3])) : (hi[0] + hi[1] + hi[2] + hi[3]) * 0.25f;
      o[e] = f2bf2(l, h);
    }
    *(u32x4_t*)(y + i * 8) = o;
  }
}
extern "C" int vg_pool2(const void* x, void* y, int B, int H, int W, int C, int64_t x_pix_stride, int is_max, int dtype,
                        vg_stream_t stream) {
  VG_CHECK(x && y && B > 0 && H > 1 && W > 1 && C > 0 && (H % 2 == 0) && (W % 2 == 0) && x_pix_stride >= C, VG_ERR_ARG,
           "vg_pool2: bad args (H, W must be even)");
  if (dtype == VG_BF16 && C % 8 == 0 && x_pix_stride % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    pool2_vec8_kernel<<<sp_grid((int64_t)B * (H / 2) * (W / 2) * (C / 8)), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, B, H, W, C / 8,
                                                                                                    x_pix_stride, is_max);
    VG_LAUNCH_CHECK();
    return VG_OK;
  }
  pool2_kernel<<<sp_grid((int64_t)B * (H / 2) * (W / 2) * C), 256, 0, (hipStream_t)stream>>>(x, y, B, H, W, C, x_pix_stride, is_max, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// windows: [B*nH*nW, ws*ws, C], window order (b, wy, wx), token order (r, c) row-major — the layout
// window_partition(...).view(-1, ws*ws, C) produces (backbones/utils.py:16-38).
__global__ __launch_bounds__(256) void window_part_kernel(const void* src_, void* dst_, int B, int H, int W, int C, int ws,
                                                          int nH, int nW, int dt, int reverse) {
  if (!reverse) {
    const int64_t n = (int64_t)B * nH * nW * ws * ws * C;
    SP_LOOP(i, n) {
      const int c = (int)(i % C);
      int64_t r = i / C;
      const int cc = (int)(r % ws); r /= ws;
      const int rr = (int)(r % ws); r /= ws;
      const int wx = (int)(r % nW); r /= nW;
      const int wy = (int)(r % nH);
      const int b = (int)(r / nH);
      const int yy = wy * ws + rr, xx = wx * ws + cc;
      const bool in = yy < H && xx < W;
      const int64_t si = (((int64_t)b * H + yy) * W + xx) * C + c;
      if (dt == 2) {  // 16-byte chunk mode: "C" counts chunks
        u32x4_t z = {0u, 0u, 0u, 0u};
        ((u32x4_t*)dst_)[i] = in ? ((const u32x4_t*)src_)[si] : z;
      } else {
        st_any(dst_, i, dt, in ? ld_any(src_, si, dt) : 0.f);
      }
    }
  } else {
    const int64_t n = (int64_t)B * H * W * C;
    SP_LOOP(i, n) {
      const int c = (int)(i % C);
      int64_t r = i / C;
      const int xx = (int)(r % W); r /= W;
      const int yy = (int)(r % H);
      const int b = (int)(r / H);
      const int wy = yy / ws, rr = yy % ws, wx = xx / ws, cc = xx % ws;
      const int64_t src = (((((int64_t)b * nH + wy) * nW + wx) * ws + rr) * ws + cc) * C + c;
      if (dt == 2) ((u32x4_t*)dst_)[i] = ((const u32x4_t*)src_)[src];
      else st_any(dst_, i, dt, ld_any(src_, src, dt));
    }
  }
}
extern "C" int vg_window_partition(const void* x, void* win, int B, int H, int W, int C, int ws, int dtype, vg_stream_t stream) {
  VG_CHECK(x && win && B > 0 && H > 0 && W > 0 && C > 0 && ws > 0, VG_ERR_ARG, "vg_window_partition: bad args");
  const int nH = (H + ws - 1) / ws, nW = (W + ws - 1) / ws;
  const int es = dtype == VG_BF16 ? 2 : 4;
  if ((C * es) % 16 == 0 && (((uintptr_t)x | (uintptr_t)win) & 15) == 0) { C = C * es / 16; dtype = 2; }  // move whole 16-byte chunks
  window_part_kernel<<<sp_grid((int64_t)B * nH * nW * ws * ws * C), 256, 0, (hipStream_t)stream>>>(x, win, B, H, W, C, ws, nH, nW, dtype, 0);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
extern "C" int vg_window_unpartition(const void* win, void* x, int B, int H, int W, int C, int ws, int dtype, vg_stream_t stream) {
  VG_CHECK(x && win && B > 0 && H > 0 && W > 0 && C > 0 && ws > 0, VG_ERR_ARG, "vg_window_unpartition: bad args");
  const int nH = (H + ws - 1) / ws, nW = (W + ws - 1) / ws;
  const int es = dtype == VG_BF16 ? 2 : 4;
  if ((C * es) % 16 == 0 && (((uintptr_t)x | (uintptr_t)win) & 15) == 0) { C = C * es / 16; dtype = 2; }
  window_part_kernel<<<sp_grid((int64_t)B * H * W * C), 256, 0, (hipStream_t)stream>>>(win, x, B, H, W, C, ws, nH, nW, dtype, 1);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// PyTorch upsample_bilinear2d, align_corners=False: src = max((dst+0.5)*scale-0.5, 0), scale = in/out.
// One body for the logits and for the fused "logits > 0" variant, so both evaluate the identical instruction sequence.
__device__ __forceinline__ float bilinear_at(const float* p, int Hi, int Wi, int oy, int ox, float sh, float sw) {
  float fy = ((float)oy + 0.5f) * sh - 0.5f; if (fy < 0.f) fy = 0.f;
  float fx = ((float)ox + 0.5f) * sw - 0.5f; if (fx < 0.f) fx = 0.f;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  return hy * (hx * p[(int64_t)y0 * Wi + x0] + lx * p[(int64_t)y0 * Wi + x1]) +
         ly * (hx * p[(int64_t)y1 * Wi + x0] + lx * p[(int64_t)y1 * Wi + x1]);
}
__global__ __launch_bounds__(256) void bilinear_kernel(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo) {
  const int64_t n = (int64_t)N * Ho * Wo;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  SP_LOOP(i, n) {
    const int ox = (int)(i % Wo);
    int64_t r = i / Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    out[i] = bilinear_at(in + (int64_t)b * Hi * Wi, Hi, Wi, oy, ox, sh, sw);
  }
}
extern "C" int vg_bilinear(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, vg_stream_t stream) {
  VG_CHECK(in && out && N >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, VG_ERR_ARG, "vg_bilinear: bad args");
  if (N == 0) return VG_OK;
  bilinear_kernel<<<sp_grid((int64_t)N * Ho * Wo), 256, 0, (hipStream_t)stream>>>(in, out, N, Hi, Wi, Ho, Wo);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
// vg_bilinear followed by vg_threshold in one pass: the fp32 logits at output resolution (4 B written + 4 B read per pixel)
// never reach HBM; a thread produces four neighbouring pixels = one 32-bit store.
// (r06: separable form for source rows that fit LDS — a workgroup takes 4 output rows, blends each row's two source rows ONCE into LDS (coalesced loads,
// the vertical weights per row), then every pixel is two LDS reads and one lerp; the per-pixel version ran a whole bilinear_at — four scattered global loads —
// per pixel: 0.7 TB/s of output at C4's clip size.  Same arithmetic, same order: hy * (hx a + lx b) + ly * (hx c + lx d) regrouped as
// hx (hy a + ly c) + lx (hy b + ly d) would round differently, so the blend keeps the two rows apart in LDS and the expression as bilinear_at writes it.)
constexpr int BM_ROWS = 4, BM_MAXW = 2048;
__global__ __launch_bounds__(256) void bilinear_mask_rows_kernel(const float* in, uint8_t* out, int N, int Hi, int Wi, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char bm_smem[];      // 2 x BM_ROWS source rows of Wi floats: sized by the launch, so that small sources leave the CU full of workgroups
  float* R0 = (float*)bm_smem;
  float* R1 = R0 + BM_ROWS * Wi;
  __shared__ float wy[BM_ROWS][2];
  const int gpr = (Ho + BM_ROWS - 1) / BM_ROWS;
  const int b = blockIdx.x / gpr, oy0 = (blockIdx.x % gpr) * BM_ROWS;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const int tid = threadIdx.x;
#pragma unroll
  for (int r = 0; r < BM_ROWS; ++r) {
    const int oy = min(oy0 + r, Ho - 1);
    float fy = ((float)oy + 0.5f) * sh - 0.5f;
    if (fy < 0.f) fy = 0.f;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
    const float* p0 = in + ((int64_t)b * Hi + y0) * Wi;
    const float* p1 = in + ((int64_t)b * Hi + y1) * Wi;
    for (int x = tid; x < Wi; x += 256) { R0[r * Wi + x] = p0[x]; R1[r * Wi + x] = p1[x]; }
    if (tid == 0) { wy[r][0] = fy - (float)y0; wy[r][1] = 1.f - (fy - (float)y0); }
  }
  __syncthreads();
  const int Wq = (Wo + 3) / 4;
  for (int xq = tid; xq < Wq; xq += 256) {      // the column terms of a thread's four pixels serve all the rows of the group
    int x0[4], x1[4];
    float lx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float fx = ((float)(xq * 4 + e) + 0.5f) * sw - 0.5f;
      if (fx < 0.f) fx = 0.f;
      x0[e] = min((int)fx, Wi - 1);
      x1[e] = x0[e] + (x0[e] < Wi - 1 ? 1 : 0);
      lx[e] = fx - (float)x0[e];
    }
    for (int r = 0; r < BM_ROWS && oy0 + r < Ho; ++r) {
      const float ly = wy[r][0], hy = wy[r][1];
      uint32_t bits = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float hx = 1.f - lx[e];
        const float v = hy * (hx * R0[r * Wi + x0[e]] + lx[e] * R0[r * Wi + x1[e]]) + ly * (hx * R1[r * Wi + x0[e]] + lx[e] * R1[r * Wi + x1[e]]);
        bits |= (xq * 4 + e < Wo && v > 0.f ? 1u : 0u) << (8 * e);
      }
      uint8_t* orow = out + ((int64_t)b * Ho + oy0 + r) * Wo;
      if ((Wo & 3) == 0) *(uint32_t*)(orow + xq * 4) = bits;
      else
        for (int e = 0; e < 4 && xq * 4 + e < Wo; ++e) orow[xq * 4 + e] = (uint8_t)((bits >> (8 * e)) & 1u);
    }
  }
}
__global__ __launch_bounds__(256) void bilinear_mask_kernel(const float* in, uint8_t* out, int N, int Hi, int Wi, int Ho, int Wo) {
  const int Wq = (Wo + 3) / 4;
  const int64_t n = (int64_t)N * Ho * Wq;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  SP_LOOP(i, n) {
    const int xq = (int)(i % Wq);
    int64_t r = i / Wq;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const float* p = in + (int64_t)b * Hi * Wi;
    uint8_t* o = out + ((int64_t)b * Ho + oy) * Wo + xq * 4;
    uint32_t bits = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ox = xq * 4 + e;
      const float v = ox < Wo ? bilinear_at(p, Hi, Wi, oy, ox, sh, sw) : 0.f;
      bits |= (v > 0.f ? 1u : 0u) << (8 * e);
    }
    if ((Wo & 3) == 0) *(uint32_t*)o = bits;
    else
      for (int e = 0; e < 4 && xq * 4 + e < Wo; ++e) o[e] = (uint8_t)((bits >> (8 * e)) & 1u);
  }
}
extern "C" int vg_bilinear_mask(const float* in, uint8_t* out, int N, int Hi, int Wi, int Ho, int Wo, vg_stream_t stream) {
  VG_CHECK(in && out && N >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, VG_ERR_ARG, "vg_bilinear_mask: bad args");
  VG_CHECK((Wo & 3) || (((uintptr_t)out) & 3) == 0, VG_ERR_ARG, "vg_bilinear_mask: out must be 4-byte aligned");
  if (N == 0) return VG_OK;
  const int64_t groups = (int64_t)N * ((Ho + BM_ROWS - 1) / BM_ROWS);
  if (Wi <= BM_MAXW && groups < (1ll << 31))
    bilinear_mask_rows_kernel<<<(unsigned)groups, 256, (size_t)2 * BM_ROWS * Wi * sizeof(float), (hipStream_t)stream>>>(in, out, N, Hi, Wi, Ho, Wo);
  else
    bilinear_mask_kernel<<<sp_grid((int64_t)N * Ho * ((Wo + 3) / 4)), 256, 0, (hipStream_t)stream>>>(in, out, N, Hi, Wi, Ho, Wo);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void upsample2_add_kernel(const void* lat, const void* top, void* y, int B, int H, int W,
                                                            int C, int dt) {
  const int64_t n = (int64_t)B * 2 * H * 2 * W * C;
  SP_LOOP(i, n) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int ox = (int)(r % (2 * W)); r /= (2 * W);
    const int oy = (int)(r % (2 * H));
    const int b = (int)(r / (2 * H));
    const int64_t src = (((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + c;
    st_any(y, i, dt, ld_any(lat, i, dt) + ld_any(top, src, dt));
  }
}
extern "C" int vg_upsample2_add(const void* lateral, const void* top, void* y, int B, int H, int W, int C, int dtype,
                                vg_stream_t stream) {
  VG_CHECK(lateral && top && y && B > 0 && H > 0 && W > 0 && C > 0, VG_ERR_ARG, "vg_upsample2_add: bad args");
  upsample2_add_kernel<<<sp_grid((int64_t)B * 4 * H * W * C), 256, 0, (hipStream_t)stream>>>(lateral, top, y, B, H, W, C, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
