// vg_twoway_image_update: the image side of a two-way block's image -> token cross-attention (SAM2 mask decoder, SURVEY.md rows S7 / S8;
// R/modeling/sam/transformer.py:160-193, Attention :236-260) for a 32-row slice of the 4096 x 256 image per wave, in ONE pass over the rows:
//
//   s[r, (h,t)] = (x_r + pe_r) . u2[(h,t)] + c2[(h,t)]        GEMM 1   [rows x 256] x [256 x 8 TP]      (u2, c2: the token side's k-projection folded
//   a[r, h, :]  = softmax over t < nt                                                                    into the q-projection weights — sam2.py: _i2t_fused)
//   y_r         = a[r, :] . w2 + bo                            GEMM 2   [rows x 8 TP] x [8 TP x 256]      (w2[(h,t)] = Wo[:, head h] v_h[t])
//   x'_r        = LayerNorm(x_r + y_r),   also written:  x'_r + pe_r   (the keys of the next pass)
//
// The unfused path runs q-projection GEMM (4096 x 256 x 128), attention (4096 queries x nt keys x 8 heads of 16), out-projection GEMM (+ residual),
// LayerNorm and the + pe add as five launches with five round trips of the [N, 4096, 256] tensor through HBM; this kernel reads x + pe and x once and
// writes x' and x' + pe once.  (The token -> image direction is one call of the head-dim-256 attention kernel: sam2.py.)
//
// Layout: both GEMMs run "swapped" (D^T = B^T A^T): a lane owns ONE image row (l31) and, per 32-column fragment, four groups of four consecutive
// columns 8 g + 4 h + {0..3}.  One v_permlane32_swap per register pairs the groups so that a lane holds 8 consecutive score columns = the 8 tokens of
// one head (TP = 8; TP = 16: half a head, the other half in lane ^ 32) — the softmax is lane-local — and those 8 probabilities, packed to bf16, ARE the
// k = 8 h .. 8 h + 7 slots of GEMM 2's B operand for the 16-column step: no shuffle between the two GEMMs.  GEMM 2's A operand (w2, transposed:
// [256 channels][8 TP]) and GEMM 1's A operand (u2: [8 TP][256]) are staged once per workgroup in LDS with XOR-swizzled 16-byte chunks; x + pe rows go
// from global memory straight into GEMM 1's B operand (a lane reads its own row: 16 x 16 bytes); y leaves through a wave-private 16 KB LDS slab into
// row-major order (a half-wave = one row of 256 channels), where the residual, the two-pass LayerNorm (32-lane reductions) and the + pe run on
// coalesced 16-byte accesses.  After the one barrier behind the staging, waves never synchronise.
#include "vg_common.h"

namespace {

struct TwoWayArgs {
  const bf16_t* xpe; const bf16_t* x; const bf16_t* u2; const float* c2; const bf16_t* w2t; const float* bo; const float* lnw; const float* lnb;
  const bf16_t* pe; bf16_t* xo; bf16_t* xpo;
  float eps;
  int N, P, nt, iters, xmod;
};

template <int TP>
__global__ __launch_bounds__(256) void twoway_image_update_kernel(TwoWayArgs p) {
  constexpr int NC = 8 * TP, NJ = NC / 32, NQ = NC / 16, NW = 4, NT = NW * 64;
  // TP = 16 (nine tokens: the video branch): u2 + w2t + four slabs would be 192 KB — u2's fragments then come straight from global memory
  // (16 KB per 4 k-steps, shared by the four waves: L1-resident)
  constexpr bool U2LDS = TP == 8;
  constexpr int U2B = U2LDS ? NC * 512 : 0, W2RB = NC * 2, W2B = 256 * W2RB, CPR = NC / 8;      // bytes; 16-byte chunks per w2t row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* u2s = smem;
  char* w2s = smem + U2B;
  float* c2s = (float*)(smem + U2B + W2B);
  char* slab = smem + U2B + W2B + NC * 4 + (threadIdx.x >> 6) * 16384;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const int n = blockIdx.y, P = p.P;
  const int nx = n % p.xmod;          // the instance's INPUT rows: x / xpe may be shared by several instances (objects of one frame, first block)

  // ---- stage u2 [NC][256], w2t [256][NC], c2 [NC] of this instance (swizzled 16-byte chunks: see the fragment reads below)
  {
    const u32x4_t* g = (const u32x4_t*)(p.u2 + (int64_t)n * NC * 256);
    if constexpr (U2LDS) {
      for (int i = tid; i < NC * 32; i += NT) {
        const int c = i >> 5, ch = i & 31;
        *(u32x4_t*)(u2s + c * 512 + ((ch ^ (c & 15)) << 4)) = g[i];
      }
    }
    const u32x4_t* gw = (const u32x4_t*)(p.w2t + (int64_t)n * 256 * NC);
    for (int i = tid; i < 256 * CPR; i += NT) {
      const int d = i / CPR, ch = i % CPR;
      const int key = TP == 8 ? (d >> 1) & 7 : d & 15;
      *(u32x4_t*)(w2s + d * W2RB + ((ch ^ key) << 4)) = gw[i];
    }
    for (int i = tid; i < NC; i += NT) c2s[i] = p.c2[(int64_t)n * NC + i];
  }
  // per-lane constants of the row-major side: this lane's 8 channels (chunk lane & 31) of bo / LayerNorm weight / bias
  float bo[8], lw[8], lb[8];
  {
    const int c8 = (lane & 31) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { bo[e] = p.bo ? p.bo[c8 + e] : 0.f; lw[e] = p.lnw[c8 + e]; lb[e] = p.lnb[c8 + e]; }
  }
  __syncthreads();

  for (int it = 0; it < p.iters; ++it) {
    const int r0 = ((blockIdx.x * p.iters + it) * NW + wave) * 32;       // this wave's 32 image rows
    if (r0 >= P) break;
    const int row = min(r0 + l31, P - 1);                                // (P is a multiple of 32 on the model's paths; clamped rows are not stored)
    // ---- GEMM 1 (swapped): S^T[c][r] = u2[c] . xpe[r]; B operand: this lane's row, 16 bytes per 16-channel step
    u32x4_t xb[16];
    {
      const u32x4_t* xr = (const u32x4_t*)(p.xpe + ((int64_t)nx * P + row) * 256) + h;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) xb[ks] = xr[2 * ks];
    }
    f32x16_t acc1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = 32 * j + l31;
        u32x4_t a;
        if constexpr (U2LDS) a = *(const u32x4_t*)(u2s + c * 512 + (((2 * ks + h) ^ (c & 15)) << 4));
        else a = ((const u32x4_t*)(p.u2 + ((int64_t)n * NC + c) * 256))[2 * ks + h];
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, xb[ks]), acc1[j], 0, 0, 0);
      }
    // ---- per-head softmax over the tokens; the probabilities packed as GEMM 2's B operand (16-column step q = 2 j + gp, slots 8 h + e)
    u32x4_t pf[NQ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float X = acc1[j][8 * gp + jj] + c2s[32 * j + 16 * gp + 4 * h + jj];
          const float Y = acc1[j][8 * gp + 4 + jj] + c2s[32 * j + 16 * gp + 8 + 4 * h + jj];
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X), __float_as_uint(Y), false, false);
          v[jj] = __uint_as_float(sw[0]);         // column 16 gp + 8 h + jj
          v[4 + jj] = __uint_as_float(sw[1]);     // column 16 gp + 8 h + 4 + jj
        }
        const int t0 = TP == 8 ? 0 : 8 * h;       // token index of v[0]
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (t0 + e >= p.nt) v[e] = -INFINITY;
          m = fmaxf(m, v[e]);
        }
        if (TP == 16) m = xor32_max(m);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = __expf(v[e] - m);
          sum += v[e];
        }
        if (TP == 16) sum = xor32_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[2 * j + gp][e] = f2bf2(v[2 * e] * inv, v[2 * e + 1] * inv);
      }
    // ---- GEMM 2 (swapped): Y^T[d][r] = w2t[d] . a[r]
    f32x16_t acc2[8];
#pragma unroll
    for (int jd = 0; jd < 8; ++jd)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[jd][r] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int jd = 0; jd < 8; ++jd) {
        const int d = 32 * jd + l31;
        const int key = TP == 8 ? (d >> 1) & 7 : d & 15;
        const u32x4_t a = *(const u32x4_t*)(w2s + d * W2RB + (((2 * q + h) ^ key) << 4));
        acc2[jd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, pf[q]), acc2[jd], 0, 0, 0);
      }
    // ---- y (rounded to bf16: the out-projection's output in the activation dtype) through the wave's slab into row-major order:
    //      row l31, 8-byte unit 8 jd + 2 g + h  ->  16-byte chunk 4 jd + g at slot chunk ^ (row & 7), half h
#pragma unroll
    for (int jd = 0; jd < 8; ++jd)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 d;
        d.x = f2bf2(acc2[jd][4 * g], acc2[jd][4 * g + 1]);
        d.y = f2bf2(acc2[jd][4 * g + 2], acc2[jd][4 * g + 3]);
        *(uint2*)(slab + l31 * 512 + (((4 * jd + g) ^ (l31 & 7)) << 4) + 8 * h) = d;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- row-major side: a half-wave = one row (32 lanes x 8 channels); residual, bias, two-pass LayerNorm, + pe
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      const int lr = 2 * rr + h, ch = lane & 31;
      const int gr = r0 + lr;
      const bool ok = gr < P;
      const int64_t off = ((int64_t)n * P + min(gr, P - 1)) * 256 + ch * 8;
      const u32x4_t yv = *(const u32x4_t*)(slab + lr * 512 + ((ch ^ (lr & 7)) << 4));
      const u32x4_t xv = *(const u32x4_t*)(p.x + ((int64_t)nx * P + min(gr, P - 1)) * 256 + ch * 8);
      const u32x4_t pv = *(const u32x4_t*)(p.pe + (int64_t)min(gr, P - 1) * 256 + ch * 8);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(xv[e] << 16) + (__uint_as_float(yv[e] << 16) + bo[2 * e]);
        v[2 * e + 1] = __uint_as_float(xv[e] & 0xffff0000u) + (__uint_as_float(yv[e] & 0xffff0000u) + bo[2 * e + 1]);
      }
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e];
      s = group_sum<32>(s);
      const float mean = s * (1.0f / 256.0f);
      float q2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] -= mean; q2 = fmaf(v[e], v[e], q2); }
      q2 = group_sum<32>(q2);
      const float rstd = __builtin_amdgcn_rsqf(q2 * (1.0f / 256.0f) + p.eps);
      u32x4_t xo, xpo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = v[2 * e] * rstd * lw[2 * e] + lb[2 * e], b = v[2 * e + 1] * rstd * lw[2 * e + 1] + lb[2 * e + 1];
        xo[e] = f2bf2(a, b);
        const float ar = __uint_as_float(xo[e] << 16), br = __uint_as_float(xo[e] & 0xffff0000u);
        xpo[e] = f2bf2(ar + __uint_as_float(pv[e] << 16), br + __uint_as_float(pv[e] & 0xffff0000u));
      }
      if (ok) {
        *(u32x4_t*)(p.xo + off) = xo;
        *(u32x4_t*)(p.xpo + off) = xpo;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slab reads have returned before the next block's y overwrites it
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// vg_mask_upscale: the mask decoder's output upscaling + hypernetwork product (R/modeling/sam/mask_decoder.py:225-245) for 32 image rows per wave:
//   g0 = ConvT2x2(x) (as a GEMM: 256 -> 4 taps x 64) + b0 + s1  ->  LayerNorm2d(64) -> GELU  ->  ConvT2x2 (64 -> 4 taps x 32) + b1 + s0 -> GELU
//   masks[k] = hyper[k] . (the 32 channels of each of the 16 output pixels),  k = 0..3
// i.e. what the unfused path runs as GEMM, pixel shuffle, add, LayerNorm, GELU, GEMM, pixel shuffle, add + GELU, batched GEMM — nine launches that
// write and re-read [N, 16384, 64] and [N, 65536, 32] tensors (3 GB at 512 instances) — with the row's 16 output pixels never leaving registers.
// All three products run swapped (a lane owns one image row = one input pixel; accumulator group g = four consecutive output channels), so the
// per-pixel LayerNorm is lane-local + one half-wave exchange, and v_permlane32_swap turns accumulator groups into the next product's B operand
// (as in vg_twoway_image_update).  The ConvT weights (128 KB + 16 KB) are staged once per workgroup (one workgroup per CU); s1 / s0 (shared by the
// objects of a frame: image i = n % Bi) are read as 8-byte pieces of the pixel's own 128 / 64-byte line; only lanes h = 0 hold the four mask values
// (the hypernetwork operand has 4 real rows of 32) and store them: 16 bytes = 4 consecutive output pixels, 512 contiguous bytes per half-wave.
struct UpscaleArgs {
  const bf16_t* x; const bf16_t* w0; const float* b0; const bf16_t* s1; const float* lnw; const float* lnb; const bf16_t* w1; const float* b1;
  const bf16_t* s0; const bf16_t* hyper; float* out;
  float eps;
  int N, Bi, es, iters;
};

__device__ __forceinline__ float tw_gelu(float x) { return vg_gelu_erf(x); }

__global__ __launch_bounds__(256, 1) void mask_upscale_kernel(UpscaleArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w0s = smem;                          // [256 outputs (tap * 64 + co)][256 in] bf16, 16-byte chunk c of row o at slot c ^ (o & 15)
  char* w1s = smem + 256 * 512;              // [128 outputs (tap2 * 32 + c2)][64 in] bf16, chunk c at slot c ^ ((o >> 1) & 7)
  char* hys = w1s + 128 * 128;               // [32][32] bf16: rows 0..3 = hyper[n], rows 4..31 zero (the A operand of the mask product)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const int n = blockIdx.y, es = p.es, P = es * es, img = n % p.Bi;
  {
    const u32x4_t* g = (const u32x4_t*)p.w0;
    for (int i = tid; i < 256 * 32; i += 256) {
      const int o = i >> 5, ch = i & 31;
      *(u32x4_t*)(w0s + o * 512 + ((ch ^ (o & 15)) << 4)) = g[i];
    }
    const u32x4_t* g1 = (const u32x4_t*)p.w1;
    for (int i = tid; i < 128 * 8; i += 256) {
      const int o = i >> 3, ch = i & 7;
      *(u32x4_t*)(w1s + o * 128 + ((ch ^ ((o >> 1) & 7)) << 4)) = g1[i];
    }
    for (int i = tid; i < 32 * 32 / 8; i += 256) {          // 128 chunks of 8 bf16
      const int r = i >> 2;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      *(u32x4_t*)(hys + i * 16) = r < 4 ? ((const u32x4_t*)(p.hyper + (int64_t)n * 128))[i] : z;
    }
  }
  // per-lane channel constants in the accumulator layout: channel 32 jf + 8 g + 4 h + jj (first product, 64 channels), 8 g + 4 h + jj (second, 32)
  float b0v[2][4][4], lwv[2][4][4], lbv[2][4][4], b1v[4][4];
#pragma unroll
  for (int jf = 0; jf < 2; ++jf)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int co = 32 * jf + 8 * g + 4 * h + jj;
        b0v[jf][g][jj] = p.b0[co];
        lwv[jf][g][jj] = p.lnw[co];
        lbv[jf][g][jj] = p.lnb[co];
      }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) b1v[g][jj] = p.b1[8 * g + 4 * h + jj];
  __syncthreads();
  u32x4_t hyf[2];
#pragma unroll
  for (int gp = 0; gp < 2; ++gp) hyf[gp] = *(const u32x4_t*)(hys + l31 * 64 + (2 * gp + h) * 16);

  for (int it = 0; it < p.iters; ++it) {
    const int r0 = ((blockIdx.x * p.iters + it) * 4 + wave) * 32;
    if (r0 >= P) break;
    const int row = min(r0 + l31, P - 1);
    const int y = row / es, x = row - y * es;
    // ---- product 1 (swapped): G^T[(tap, co)][r] = w0[(tap, co)] . x[r]
    u32x4_t xb[16];
    {
      const u32x4_t* xr = (const u32x4_t*)(p.x + ((int64_t)n * P + row) * 256) + h;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) xb[ks] = xr[2 * ks];
    }
    f32x16_t acc1[8];
#pragma unroll
    for (int jo = 0; jo < 8; ++jo)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[jo][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int jo = 0; jo < 8; ++jo) {
        const int o = 32 * jo + l31;
        const u32x4_t a = *(const u32x4_t*)(w0s + o * 512 + (((2 * ks + h) ^ (o & 15)) << 4));
        acc1[jo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, xb[ks]), acc1[jo], 0, 0, 0);
      }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      float res[2][4][4];          // [dx][tap2][k]
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int tap = dy * 2 + dx;
        // + b0 + s1, LayerNorm2d over the pixel's 64 channels (two-pass), GELU
        const int64_t pix1 = (int64_t)(2 * y + dy) * (2 * es) + 2 * x + dx;
        const uint2* s1p = (const uint2*)(p.s1 + ((int64_t)img * 4 * P + pix1) * 64) + h;       // 8-byte piece index = co / 4 = 8 jf + 2 g + h
        float v[2][4][4];
        float sum = 0.f;
#pragma unroll
        for (int jf = 0; jf < 2; ++jf)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint2 sv = s1p[8 * jf + 2 * g];
            const float s4[4] = {__uint_as_float(sv.x << 16), __uint_as_float(sv.x & 0xffff0000u), __uint_as_float(sv.y << 16), __uint_as_float(sv.y & 0xffff0000u)};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              v[jf][g][jj] = acc1[2 * tap + jf][4 * g + jj] + b0v[jf][g][jj] + s4[jj];
              sum += v[jf][g][jj];
            }
          }
        sum = xor32_sum(sum);
        const float mean = sum * (1.0f / 64.0f);
        float q2 = 0.f;
#pragma unroll
        for (int jf = 0; jf < 2; ++jf)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { v[jf][g][jj] -= mean; q2 = fmaf(v[jf][g][jj], v[jf][g][jj], q2); }
        q2 = xor32_sum(q2);
        const float rstd = __builtin_amdgcn_rsqf(q2 * (1.0f / 64.0f) + p.eps);
        // the 64 activations as product 2's B operand: 16-channel step q = 2 jf + gp holds channels 32 jf + 16 gp + 8 h + e in slot 8 h + e
        u32x4_t uf[4];
#pragma unroll
        for (int jf = 0; jf < 2; ++jf)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float e8[8];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const float X = tw_gelu(v[jf][2 * gp][jj] * rstd * lwv[jf][2 * gp][jj] + lbv[jf][2 * gp][jj]);
              const float Y = tw_gelu(v[jf][2 * gp + 1][jj] * rstd * lwv[jf][2 * gp + 1][jj] + lbv[jf][2 * gp + 1][jj]);
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X), __float_as_uint(Y), false, false);
              e8[jj] = __uint_as_float(sw[0]);
              e8[4 + jj] = __uint_as_float(sw[1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) uf[2 * jf + gp][e] = f2bf2(e8[2 * e], e8[2 * e + 1]);
          }
        // ---- product 2 (swapped): [(tap2, c2)][r] = w1[(tap2, c2)] . u[r]   (K = 64)
        f32x16_t acc2[4];
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[t2][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t2 = 0; t2 < 4; ++t2) {
            const int o = 32 * t2 + l31;
            const u32x4_t a = *(const u32x4_t*)(w1s + o * 128 + (((2 * q + h) ^ ((o >> 1) & 7)) << 4));
            acc2[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, uf[q]), acc2[t2], 0, 0, 0);
          }
        // ---- + b1 + s0, GELU, and the hypernetwork product: masks[k] = hyper[k] . u2   (K = 32, rows k >= 4 of the operand are zero)
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
          const int dy2 = t2 >> 1, dx2 = t2 & 1;
          const int64_t pix2 = (int64_t)(4 * y + 2 * dy + dy2) * (4 * es) + 4 * x + 2 * dx + dx2;
          const uint2* s0p = (const uint2*)(p.s0 + ((int64_t)img * 16 * P + pix2) * 32) + h;      // piece index = c2 / 4 = 2 g + h
          u32x4_t u2f[2];
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float e8[8];
            const uint2 sa = s0p[2 * (2 * gp)], sb = s0p[2 * (2 * gp + 1)];
            const float a4[4] = {__uint_as_float(sa.x << 16), __uint_as_float(sa.x & 0xffff0000u), __uint_as_float(sa.y << 16), __uint_as_float(sa.y & 0xffff0000u)};
            const float c4[4] = {__uint_as_float(sb.x << 16), __uint_as_float(sb.x & 0xffff0000u), __uint_as_float(sb.y << 16), __uint_as_float(sb.y & 0xffff0000u)};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const float X = tw_gelu(acc2[t2][8 * gp + jj] + b1v[2 * gp][jj] + a4[jj]);
              const float Y = tw_gelu(acc2[t2][8 * gp + 4 + jj] + b1v[2 * gp + 1][jj] + c4[jj]);
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X), __float_as_uint(Y), false, false);
              e8[jj] = __uint_as_float(sw[0]);
              e8[4 + jj] = __uint_as_float(sw[1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) u2f[gp][e] = f2bf2(e8[2 * e], e8[2 * e + 1]);
          }
          f32x16_t acc3;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp)
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, hyf[gp]), __builtin_bit_cast(bf16x8_t, u2f[gp]), acc3, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < 4; ++k) res[dx][t2][k] = acc3[k];          // (lanes h = 0: accumulator rows 0..3 = k)
        }
      }
      // ---- store: for each mask k and output row 4 y + 2 dy + dy2, the four pixels 4 x + {0..3} = (dx, dx2) in {0,1}^2: 16 bytes per lane (h = 0)
      if (h == 0 && r0 + l31 < P) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int dy2 = 0; dy2 < 2; ++dy2) {
            const f32x4_t o = {res[0][dy2 * 2 + 0][k], res[0][dy2 * 2 + 1][k], res[1][dy2 * 2 + 0][k], res[1][dy2 * 2 + 1][k]};
            *(f32x4_t*)(p.out + ((int64_t)n * 4 + k) * 16 * P + (int64_t)(4 * y + 2 * dy + dy2) * (4 * es) + 4 * x) = o;
          }
      }
    }
  }
}

}  // namespace

extern "C" int vg_twoway_image_update(const void* xpe, const void* x, const void* u2, const float* c2, const void* w2t, const float* bo, const float* ln_w,
                                      const float* ln_b, float eps, const void* pe, void* x_out, void* xpe_out, int N, int x_instances, int P, int nt, int TP, int dtype,
                                      vg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VG_CHECK(xpe && x && u2 && c2 && w2t && ln_w && ln_b && pe && x_out && xpe_out, VG_ERR_ARG, "vg_twoway_image_update: null pointer");
  VG_CHECK(dtype == VG_BF16, VG_ERR_ARG, "vg_twoway_image_update: bf16 only (the fp32 parity mode keeps the unfused order)");
  VG_CHECK(N > 0 && P > 0 && (TP == 8 || TP == 16) && nt > 0 && nt <= TP && x_instances > 0 && N % x_instances == 0, VG_ERR_ARG,
           "vg_twoway_image_update: bad sizes N=%d (inputs %d) P=%d nt=%d TP=%d", N, x_instances, P, nt, TP);
  TwoWayArgs a{(const bf16_t*)xpe, (const bf16_t*)x, (const bf16_t*)u2, c2, (const bf16_t*)w2t, bo, ln_w, ln_b, (const bf16_t*)pe, (bf16_t*)x_out,
               (bf16_t*)xpe_out, eps, N, P, nt, 2, x_instances};
  const int NC = 8 * TP, NW = 4;
  const int lds = (TP == 8 ? NC * 512 : 0) + 256 * NC * 2 + NC * 4 + NW * 16384;
  const int rows_wg = 32 * NW * a.iters;
  dim3 grid((P + rows_wg - 1) / rows_wg, N);
  static bool attr8 = false, attr16 = false;
  if (TP == 8) {
    if (!attr8) { (void)hipFuncSetAttribute((const void*)twoway_image_update_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr8 = true; }
    twoway_image_update_kernel<8><<<grid, NW * 64, lds, stream>>>(a);
  } else {
    if (!attr16) { (void)hipFuncSetAttribute((const void*)twoway_image_update_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr16 = true; }
    twoway_image_update_kernel<16><<<grid, NW * 64, lds, stream>>>(a);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_mask_upscale(const void* x, const void* w0, const float* b0, const void* s1, const float* ln_w, const float* ln_b, float eps,
                               const void* w1, const float* b1, const void* s0, const void* hyper, float* masks, int N, int images, int es, int dtype,
                               vg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VG_CHECK(x && w0 && b0 && s1 && ln_w && ln_b && w1 && b1 && s0 && hyper && masks, VG_ERR_ARG, "vg_mask_upscale: null pointer");
  VG_CHECK(dtype == VG_BF16, VG_ERR_ARG, "vg_mask_upscale: bf16 only (the fp32 parity mode keeps the unfused order)");
  VG_CHECK(N > 0 && images > 0 && N % images == 0 && es > 0, VG_ERR_ARG, "vg_mask_upscale: bad sizes N=%d images=%d es=%d", N, images, es);
  const int P = es * es;
  UpscaleArgs a{(const bf16_t*)x, (const bf16_t*)w0, b0, (const bf16_t*)s1, ln_w, ln_b, (const bf16_t*)w1, b1, (const bf16_t*)s0, (const bf16_t*)hyper,
                masks, eps, N, images, es, N >= 256 ? 8 : (N >= 64 ? 4 : 2)};
  const int lds = 256 * 512 + 128 * 128 + 32 * 64;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)mask_upscale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  const int rows_wg = 128 * a.iters;
  dim3 grid((P + rows_wg - 1) / rows_wg, N);
  mask_upscale_kernel<<<grid, 256, lds, stream>>>(a);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---- token-side layout helpers of the fused two-way path (r05): the per-head block-diagonal form the small GEMMs work on
// (videoglamm_amd/sam2.py:_heads_bd) and its inverse, one launch each (torch.zeros + a strided copy / a strided copy + reshape before)
namespace {
__global__ __launch_bounds__(256) void heads_bd_kernel(const void* x, void* out, int64_t n, int nt, int TP, int dt) {
  // out [N, 8, TP, 128]: row (h, t) holds head h's 16 channels of token t (x [N, nt, 128]), zeros elsewhere
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i & 127);
    int64_t r = i >> 7;
    const int t = (int)(r % TP); r /= TP;
    const int h = (int)(r & 7);
    const int64_t nn = r >> 3;
    const float v = ((c >> 4) == h && t < nt) ? ld_any(x, (nn * nt + t) * 128 + c, dt) : 0.f;
    st_any(out, i, dt, v);
  }
}
__global__ __launch_bounds__(256) void heads_bd_gather_kernel(const void* full, void* out, int64_t n, int nt, int TP, int dt) {
  // out [N, nt, 128]: channel c of token t = full [N, 8, TP, 128] at row (c / 16, t), channel c
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i & 127);
    const int64_t r = i >> 7;
    const int t = (int)(r % nt);
    const int64_t nn = r / nt;
    st_any(out, i, dt, ld_any(full, ((nn * 8 + (c >> 4)) * TP + t) * 128 + c, dt));
  }
}
}  // namespace

extern "C" int vg_heads_blockdiag(const void* x, void* out, int N, int nt, int TP, int gather, int dtype, vg_stream_t stream) {
  VG_CHECK(x && out && N > 0 && nt > 0 && nt <= TP && (dtype == VG_BF16 || dtype == VG_F32), VG_ERR_ARG, "vg_heads_blockdiag: bad args");
  const int64_t n = gather ? (int64_t)N * nt * 128 : (int64_t)N * 8 * TP * 128;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  if (gather) heads_bd_gather_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, out, n, nt, TP, dtype);
  else heads_bd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, out, n, nt, TP, dtype);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
