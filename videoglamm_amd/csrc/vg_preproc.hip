// Image pre-processing on the device: the decoded uint8 frames are uploaded once (0.75 MB per
// 512^2 frame instead of 12.6 MB of fp32 per 1024^2 SAM input plus the encoder copies) and the three model inputs are
// produced in HBM.  Byte / integer work, HBM-bound.
//
//   resample_u8_kernel : one 8-bit pass of Pillow's separable resampler (Resample.c ImagingResampleHorizontal_8bpc /
//                        Vertical_8bpc): out = clip8((2^21 + sum_k in[xmin + k] * coeff[k]) >> 22), integer
//                        coefficients from the host (videoglamm_amd/preproc.py).  Bit-exact with Image.resize.
//   resize_cv_linear_kernel: OpenCV's 8-bit INTER_LINEAR resize (the InternVideo2 stream: cv2.resize in the reference)
//   normalize_u8_kernel: uint8 HWC (optional crop) -> planar CHW: SAM's (x - mean) / std in fp32 on 0..255 values
//                        (R/utils/sam_transforms.py:50-55), or (x / 255 - mean) / std evaluated in fp64 (the numpy
//                        arithmetic of the encoder processors, R/utils/enc_preprocessors.py:120-166).
#include "vg_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// horizontal: one thread per output PIXEL (all C <= 4 channels); vertical: one thread per output BYTE (x, c flattened),
// so that a wave reads 64 consecutive bytes per tap in both directions
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* in, uint8_t* out, int64_t rows, int W, int C, int Wo,
                                                         const int32_t* bounds, const int32_t* coeffs, int ksize) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * Wo) return;
  const int64_t row = i / Wo;
  const int xo = (int)(i % Wo);
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const int32_t* k = coeffs + (int64_t)xo * ksize;
  const uint8_t* src = in + (row * W + xmin) * C;
  int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
  for (int t = 0; t < n; ++t) {
    const int kv = k[t];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) acc[c] += (int)src[t * C + c] * kv;
  }
  uint8_t* dst = out + i * C;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < C) {
      const int v = acc[c] >> PRECISION_BITS;
      dst[c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* in, uint8_t* out, int H, int64_t rowbytes, int Ho,
                                                         const int32_t* bounds, const int32_t* coeffs, int ksize) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int yo = blockIdx.y;
  if (x >= rowbytes) return;
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  const int32_t* k = coeffs + (int64_t)yo * ksize;
  const uint8_t* src = in + ((int64_t)blockIdx.z * H + ymin) * rowbytes + x;
  int acc = 1 << (PRECISION_BITS - 1);
  for (int t = 0; t < n; ++t) acc += (int)src[t * rowbytes] * k[t];
  const int v = acc >> PRECISION_BITS;
  out[((int64_t)blockIdx.z * Ho + yo) * rowbytes + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// OpenCV's 8-bit INTER_LINEAR resize (imgproc/src/resize.cpp): 2 taps per axis in 11-bit fixed point, both passes in one
// kernel — one thread per output byte (x, c flattened: a wave reads runs of neighbouring bytes of two source rows).
//   S(row) = in[row, x0] * a0 + in[row, x1] * a1                                    (HResizeLinear, int32)
//   out    = (((b0 * (S(y0) >> 4)) >> 16) + ((b1 * (S(y1) >> 4)) >> 16) + 2) >> 2   (VResizeLinear, FixedPtCast<.., 22>)
// area2 != 0: the exact 2x down-scale, which cv::resize re-routes to the INTER_AREA fast path: (a + b + c + d + 2) >> 2.
__global__ __launch_bounds__(256) void resize_cv_linear_kernel(const uint8_t* in, uint8_t* out, int H, int W, int C, int Ho, int Wo,
                                                               const int32_t* xi, const int32_t* xa, const int32_t* yi, const int32_t* yb, int area2) {
  const int64_t rowb = (int64_t)Wo * C;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rowb) return;
  const int yo = blockIdx.y, n = blockIdx.z;
  const int xo = (int)(i / C), c = (int)(i % C);
  const uint8_t* img = in + (int64_t)n * H * W * C;
  int v;
  if (area2) {
    const uint8_t* r0 = img + ((int64_t)(2 * yo) * W + 2 * xo) * C + c;
    const uint8_t* r1 = r0 + (int64_t)W * C;
    v = ((int)r0[0] + (int)r0[C] + (int)r1[0] + (int)r1[C] + 2) >> 2;
  } else {
    const int x0 = xi[2 * xo] * C + c, x1 = xi[2 * xo + 1] * C + c, a0 = xa[2 * xo], a1 = xa[2 * xo + 1];
    const uint8_t* r0 = img + (int64_t)yi[2 * yo] * W * C;
    const uint8_t* r1 = img + (int64_t)yi[2 * yo + 1] * W * C;
    const int s0 = (int)r0[x0] * a0 + (int)r0[x1] * a1, s1 = (int)r1[x0] * a0 + (int)r1[x1] * a1;
    v = (((yb[2 * yo] * (s0 >> 4)) >> 16) + ((yb[2 * yo + 1] * (s1 >> 4)) >> 16) + 2) >> 2;
  }
  out[((int64_t)n * Ho + yo) * rowb + i] = (uint8_t)v;
}

struct NormArgs {
  const uint8_t* in;
  void* out;
  int N, H, W, top, left, h, w, mode, out_dtype;
  double mean[3], std[3];
};

__global__ __launch_bounds__(256) void normalize_u8_kernel(NormArgs p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // output pixel (n, y, x)
  const int64_t plane = (int64_t)p.h * p.w;
  if (i >= plane * p.N) return;
  const int n = (int)(i / plane);
  const int y = (int)((i % plane) / p.w), x = (int)(i % p.w);
  const uint8_t* src = p.in + (((int64_t)n * p.H + p.top + y) * p.W + p.left + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v;
    if (p.mode == 0) v = __fdiv_rn(__fsub_rn((float)src[c], (float)p.mean[c]), (float)p.std[c]);
    else v = (float)(((double)src[c] / 255.0 - p.mean[c]) / p.std[c]);
    st_any(p.out, ((int64_t)n * 3 + c) * plane + (int64_t)y * p.w + x, p.out_dtype, v);
  }
}

}  // namespace

extern "C" int vg_resample_u8(const uint8_t* in, uint8_t* out, int N, int H, int W, int C, int out_size, int axis,
                              const int32_t* bounds, const int32_t* coeffs, int ksize, vg_stream_t stream) {
  VG_CHECK(in && out && bounds && coeffs, VG_ERR_ARG, "vg_resample_u8: null pointer");
  VG_CHECK(N > 0 && H > 0 && W > 0 && C >= 1 && C <= 4 && out_size > 0 && ksize > 0, VG_ERR_ARG,
           "vg_resample_u8: bad shape N=%d H=%d W=%d C=%d out=%d ksize=%d", N, H, W, C, out_size, ksize);
  VG_CHECK(axis == 0 || axis == 1, VG_ERR_ARG, "vg_resample_u8: axis %d not in {0 (vertical), 1 (horizontal)}", axis);
  hipStream_t st = (hipStream_t)stream;
  if (axis == 1) {
    const int64_t total = (int64_t)N * H * out_size;
    resample_h_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(in, out, (int64_t)N * H, W, C, out_size, bounds, coeffs, ksize);
  } else {
    VG_CHECK(out_size <= 65535 && N <= 65535, VG_ERR_UNSUPPORTED, "vg_resample_u8: out_size / N above 65535");
    const int64_t rowbytes = (int64_t)W * C;
    resample_v_kernel<<<dim3((unsigned)((rowbytes + 255) / 256), out_size, N), 256, 0, st>>>(in, out, H, rowbytes, out_size, bounds, coeffs, ksize);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_resize_cv_linear_u8(const uint8_t* in, uint8_t* out, int N, int H, int W, int C, int Ho, int Wo, const int32_t* xi,
                                      const int32_t* xa, const int32_t* yi, const int32_t* yb, vg_stream_t stream) {
  VG_CHECK(in && out, VG_ERR_ARG, "vg_resize_cv_linear_u8: null pointer");
  VG_CHECK(N > 0 && H > 0 && W > 0 && C >= 1 && C <= 4 && Ho > 0 && Wo > 0 && Ho <= 65535 && N <= 65535, VG_ERR_ARG,
           "vg_resize_cv_linear_u8: bad shape N=%d H=%d W=%d C=%d -> %dx%d", N, H, W, C, Ho, Wo);
  const int area2 = (H == 2 * Ho && W == 2 * Wo) ? 1 : 0;
  VG_CHECK(area2 || (xi && xa && yi && yb), VG_ERR_ARG, "vg_resize_cv_linear_u8: tap tables missing");
  VG_CHECK(!(H == Ho && W == Wo), VG_ERR_ARG, "vg_resize_cv_linear_u8: same size (cv::resize copies; so should the caller)");
  const int64_t rowb = (int64_t)Wo * C;
  resize_cv_linear_kernel<<<dim3((unsigned)((rowb + 255) / 256), Ho, N), 256, 0, (hipStream_t)stream>>>(in, out, H, W, C, Ho, Wo, xi, xa, yi, yb, area2);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_normalize_u8(const uint8_t* in, void* out, int N, int H, int W, int top, int left, int h, int w,
                               const double* mean, const double* std, int mode, int out_dtype, vg_stream_t stream) {
  VG_CHECK(in && out && mean && std, VG_ERR_ARG, "vg_normalize_u8: null pointer");
  VG_CHECK(N > 0 && h > 0 && w > 0 && top >= 0 && left >= 0 && top + h <= H && left + w <= W, VG_ERR_ARG,
           "vg_normalize_u8: crop (%d,%d,%d,%d) outside %dx%d", top, left, h, w, H, W);
  VG_CHECK(mode == 0 || mode == 1, VG_ERR_ARG, "vg_normalize_u8: mode %d not in {0, 1}", mode);
  VG_CHECK(out_dtype == VG_F32 || out_dtype == VG_BF16, VG_ERR_ARG, "vg_normalize_u8: bad dtype %d", out_dtype);
  NormArgs p{in, out, N, H, W, top, left, h, w, mode, out_dtype, {mean[0], mean[1], mean[2]}, {std[0], std[1], std[2]}};
  const int64_t total = (int64_t)N * h * w;
  normalize_u8_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
