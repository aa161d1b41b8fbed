// Shared device/host helpers for the VideoGLaMM MI355X (gfx950) kernel library.
// wave = 64 lanes, MFMA 32x32 tiles, fp32 accumulate everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/vg_kernels.h"

typedef uint16_t bf16_t;  // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

#define VG_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer emulation it
// replaces was ~5 VALU ops per value — a fifth of the flash-attention inner loop, which is VALU-bound)
typedef __bf16 vg_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float vg_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
  const vg_f32x2_t x = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, vg_bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

template <typename T> struct vg_elt;
template <> struct vg_elt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct vg_elt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// dtype-erased scalar load/store (dt: VG_F32 / VG_BF16), index in elements
__device__ __forceinline__ float ld_any(const void* p, int64_t i, int dt) {
  return dt == VG_BF16 ? bf2f(((const bf16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int dt, float v) {
  if (dt == VG_BF16) ((bf16_t*)p)[i] = f2bf(v); else ((float*)p)[i] = v;
}

// Wave reductions on the VALU alone (r06): four DPP steps inside the 16-lane rows, then v_permlane16_swap / v_permlane32_swap across them; the result is in
// every lane.  __shfl_xor compiles to ds_bpermute_b32 — an LDS-pipe round trip per step, six per reduction — which the latency-bound kernels (the skinny GEMM,
// the decode GEMVs' norm prologue and row sums, the short-row norms) paid in full: 0.3 us per reduced value at the end of a dependent chain.
template <int CTRL>
__device__ __forceinline__ float vg_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over aligned groups of W lanes (W = 2 ... 64), result in every lane of the group.  The steps run from the widest to the narrowest — the association
// of the __shfl_xor butterfly (32, 16, 8, 4, 2, 1) this replaces, so every kernel's sums keep their bits, and a short row reduced by 16 lanes equals the same
// row reduced by a whole wave whose other lanes hold zeros (fused kernels are tested bit-equal to the stand-alone ones).  row_ror:8 IS xor 8 inside a
// 16-lane row; row_ror:4 equals xor 4 in VALUE once the row is 8-periodic (after the xor-8 step) — for a group of 8 on its own it is row_half_mirror.
template <int W>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (W >= 64) {
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  }
  if constexpr (W >= 32) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  }
  if constexpr (W >= 16) v += vg_dpp<0x128>(v);    // row_ror:8
  if constexpr (W >= 16) v += vg_dpp<0x124>(v);    // row_ror:4
  else if constexpr (W == 8) v += vg_dpp<0x141>(v);    // row_half_mirror
  if constexpr (W >= 4) v += vg_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  if constexpr (W >= 2) v += vg_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
// single butterfly steps across the 16-lane rows / the wave halves (the __shfl_xor(v, 16 | 32) they replace pairs the same lanes: same bits)
__device__ __forceinline__ float xor16_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float wave_max(float v) {
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  v = fmaxf(v, vg_dpp<0x128>(v));
  v = fmaxf(v, vg_dpp<0x124>(v));
  v = fmaxf(v, vg_dpp<0x4E>(v));
  return fmaxf(v, vg_dpp<0xB1>(v));
}

// Exact (erf) GELU: x * Phi(x), Phi(x) = 0.5 * (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|error of erf| <= 1.5e-7, i.e. fp32 rounding
// level) instead of libm erff's ~40 instructions — the fc1 epilogues of Hiera / InternVideo2 apply it to 22e9 elements per C2 clip, in kernels that
// are bound by instruction issue.  r03, in two steps measured on the fc1 GEMMs:
//   (1) v_rcp_f32 / v_exp_f32 directly (1 ulp each): the correctly rounded __frcp_rn expands to the IEEE division sequence, 10 of ~22 instructions;
//   (2) everything in terms of x (the 1/sqrt 2 folded into the constants, 0.5 folded into the polynomial), exp2 argument c * x * x, and the branch
//       "x < 0 ? q : 1 - q" as 0.5 + copysign(0.5 - q, x): one v_bfi instead of compare + select, and straight-line code the compiler packs into
//       v_pk_fma_f32 / v_pk_mul_f32 pairs.  (0.5 - q loses the RELATIVE accuracy of Phi deep in the negative tail — absolute error of x * Phi(x)
//       <= |x| * 6e-8 there, the rounding level of every other term.)
__device__ __forceinline__ float vg_gelu_erf(float x) {
  const float a = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, a, 1.0f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * x * x);   // 0.5 * erfc(|x| / sqrt 2)
  const float phi = 0.5f + __builtin_copysignf(0.5f - q, x);
  return x * phi;
}

// x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions): ONE definition for every SwiGLU site (GEMM
// epilogues, vg_swiglu, the decode GEMV) — the fused and unfused paths are compared bit for bit by the tests
__device__ __forceinline__ float vg_silu(float g) { return g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)); }

// activation codes (vg_kernels.h): 0 none, 1 gelu(erf), 2 quick_gelu, 3 relu, 4 silu, 5 sigmoid
__device__ __forceinline__ float vg_act(float x, int act) {
  switch (act) {
    case VG_ACT_GELU: return vg_gelu_erf(x);
    case VG_ACT_QUICK_GELU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    case VG_ACT_RELU: return x > 0.f ? x : 0.f;
    case VG_ACT_SILU: return vg_silu(x);
    case VG_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    default: return x;
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences GLOBAL memory: the compiler waits for vmcnt(0),
// i.e. for every global store the wave has issued to be acknowledged — inside a multi-pass GEMM epilogue that exposes the full
// store latency once per pass (measured r02, 256x256-tile kernel on M = 32768, N = 2304, K = 576: 168 -> 105 us without the waits).
__device__ __forceinline__ void vg_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// row index of accumulator register r for lane-half h in the 32x32 MFMA C/D layout
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// host-side error plumbing (vg_api.cpp)
extern "C" void vg_set_error(const char* fmt, ...);
#define VG_CHECK(cond, code, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      vg_set_error(__VA_ARGS__);             \
      return (code);                         \
    }                                        \
  } while (0)
#define VG_LAUNCH_CHECK()                                              \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      vg_set_error("launch failed: %s", hipGetErrorString(e__));       \
      return VG_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)
