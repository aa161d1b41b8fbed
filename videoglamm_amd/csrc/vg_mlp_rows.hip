// vg_mlp_rows: y = x + W2 . gelu(W1 . LayerNorm(x) + b1) + b2 for narrow rows (C = 144 / 288: Hiera stages 1 and 2) in ONE launch (r05).
//
// What it replaces (R/.../sam2/modeling/backbones/hieradet.py:160-168: x = x + drop_path(self.mlp(self.norm2(x))), MLP = fc1, GELU, fc2 of
// sam2_utils.py:108-132): vg_layernorm, vg_gemm (+ GELU) and vg_gemm (+ residual) — at these widths three HBM-bound passes: the 4 C wide hidden
// activation alone is written and read back once per block (1.2 GB per 16-frame launch of stage 1, where the fc1 GEMM ran at 2.3 TB/s and its
// exact-erf GELU epilogue at the VALU's rate).  Here a workgroup keeps 128 rows on chip from the LayerNorm to the residual add: x is read
// twice (36 / 72 KB per workgroup, the second time from L2), y written once.
//
// 256 threads = 4 waves x 32 rows.  Prologue: the x tile goes to LDS, two threads per row normalise it in place (two-pass fp32 statistics, the
// result rounded to bf16 like vg_layernorm's output), every wave takes the K fragments of its 32 rows into registers (the b-operand of all fc1
// MFMAs), and the tile's LDS is handed to the weight chunks.  Then, per chunk of 32 hidden units: W1's 32 rows and the matching 32 columns of W2
// are staged in LDS (register prefetch one chunk ahead at C = 144); fc1 runs as D = W1c . xn^T, so a lane holds 16 hidden values of ONE row —
// + bias, GELU and the bf16 rounding happen in place and the packed values ARE the b-operand of the fc2 MFMAs (contraction index = the
// accumulator's own register order, the a-operand read from W2c in that order: the trick of vg_attention's P V step); fc2 accumulates the
// 32 x C output rows over the chunks.  Epilogue: + bias, + x, bf16, through a wave-private fp32 slab (row-major 16-byte stores).
#include "vg_gemm_common.h"

namespace {

struct MlpArgs {
  const bf16_t* x; bf16_t* y; int64_t ldx, ldy;
  const float* ln_w; const float* ln_b; float eps;
  const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
  int M, H;
};

template <int C, int N1, int N2>
__device__ __forceinline__ void mlp_fetch(const MlpArgs& p, int j, int tid, u32x4_t (&r1)[N1], u32x4_t (&r2)[N2]) {
  constexpr int CPR = C / 8;
#pragma unroll
  for (int i = 0; i < N1; ++i) {
    const int idx = min(tid + i * 256, 32 * CPR - 1);
    r1[i] = *(const u32x4_t*)(p.w1 + (int64_t)j * 32 * C + idx * 8);                     // 32 consecutive rows of W1 [H, C]: one contiguous block
  }
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const int idx = min(tid + i * 256, C * 4 - 1), row = idx >> 2, qd = idx & 3;
    r2[i] = *(const u32x4_t*)(p.w2 + (int64_t)row * p.H + j * 32 + qd * 8);              // columns [32 j, 32 j + 32) of W2 [C, H]
  }
}
template <int C, int N1, int N2>
__device__ __forceinline__ void mlp_stage(char* W1c, char* W2c, int tid, const u32x4_t (&r1)[N1], const u32x4_t (&r2)[N2]) {
  constexpr int CPR = C / 8, RSX = C * 2 + 16, RSW2 = 72;
#pragma unroll
  for (int i = 0; i < N1; ++i) {
    const int idx = tid + i * 256;
    if (idx < 32 * CPR) {
      const int row = idx / CPR, c = idx - row * CPR;
      *(u32x4_t*)(W1c + row * RSX + c * 16) = r1[i];
    }
  }
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const int idx = tid + i * 256;
    if (idx < C * 4) {
      const int row = idx >> 2, qd = idx & 3;
      const uint2 lo = {r2[i][0], r2[i][1]}, hi = {r2[i][2], r2[i][3]};
      *(uint2*)(W2c + row * RSW2 + qd * 16) = lo;
      *(uint2*)(W2c + row * RSW2 + qd * 16 + 8) = hi;
    }
  }
}

template <int C>
__global__ __launch_bounds__(256, 2) void mlp_rows_kernel(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = C / 16;             // fc1 k-steps
  constexpr int NF = (C + 31) / 32;      // 32-column fragments of the output
  constexpr int CPR = C / 8;             // 16-byte chunks per row of x / W1
  constexpr int RSX = C * 2 + 16;        // LDS row stride of the x tile and of a W1 chunk (conflict-free 16-byte fragment reads)
  constexpr int RSW2 = 72;               // LDS row stride of a W2 chunk (32 hidden = 64 bytes + 8: conflict-free 8-byte reads)
  constexpr bool PF = C <= 144;          // the next chunk's weights wait in registers (no registers to spare at C = 288: two workgroups per CU overlap instead)
  constexpr int NW1 = (32 * CPR + 255) / 256, NW2 = (C * 4 + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int M = p.M, H = p.H, nch = H / 32;
  const int m0 = blockIdx.x * 128;
  char* X = smem;                        // [128][RSX] during the prologue and the epilogue
  char* W1c = smem;                      // [32][RSX] | [C][RSW2] while the chunks run
  char* W2c = smem + 32 * RSX;

  // ---- prologue: x tile -> LDS
  for (int idx = tid; idx < 128 * CPR; idx += 256) {
    const int row = idx / CPR, c = idx - row * CPR;
    const int m = min(m0 + row, M - 1);
    *(u32x4_t*)(X + row * RSX + c * 16) = *(const u32x4_t*)(p.x + (int64_t)m * p.ldx + c * 8);
  }
  __syncthreads();
  {
    // LayerNorm in place: thread (row, half) owns every second 16-byte chunk of its row
    const int row = tid >> 1, half = tid & 1;
    constexpr int NC = (CPR + 1) / 2;
    u32x4_t v[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = half + 2 * i;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      v[i] = c < CPR ? *(const u32x4_t*)(X + row * RSX + c * 16) : z;
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __uint_as_float(v[i][e] << 16) + __uint_as_float(v[i][e] & 0xffff0000u);
    }
    s += __shfl_xor(s, 1, 64);
    const float mean = s * (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      if (half + 2 * i < CPR) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = __uint_as_float(v[i][e] << 16) - mean, d1 = __uint_as_float(v[i][e] & 0xffff0000u) - mean;
          q += d0 * d0 + d1 * d1;
        }
      }
    }
    q += __shfl_xor(q, 1, 64);
    const float rstd = rsqrtf(q * (1.0f / (float)C) + p.eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = half + 2 * i;
      if (c < CPR) {
        const f32x4_t w0 = *(const f32x4_t*)(p.ln_w + c * 8), w1 = *(const f32x4_t*)(p.ln_w + c * 8 + 4);
        const f32x4_t b0 = *(const f32x4_t*)(p.ln_b + c * 8), b1 = *(const f32x4_t*)(p.ln_b + c * 8 + 4);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float n0 = (__uint_as_float(v[i][e] << 16) - mean) * rstd, n1 = (__uint_as_float(v[i][e] & 0xffff0000u) - mean) * rstd;
          const float g0 = e < 2 ? w0[2 * e] : w1[2 * e - 4], g1 = e < 2 ? w0[2 * e + 1] : w1[2 * e - 3];
          const float c0 = e < 2 ? b0[2 * e] : b1[2 * e - 4], c1 = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
          o[e] = f2bf2(n0 * g0 + c0, n1 * g1 + c1);
        }
        *(u32x4_t*)(X + row * RSX + c * 16) = o;
      }
    }
  }
  __syncthreads();
  u32x4_t xq[KS];                         // the wave's 32 normalised rows as fc1 b-operand fragments: row l31, k = 16 s + 8 h + [0, 8)
#pragma unroll
  for (int s = 0; s < KS; ++s) xq[s] = *(const u32x4_t*)(X + (wave * 32 + l31) * RSX + s * 32 + h * 16);

  // ---- weight chunks
  u32x4_t pw1[NW1], pw2[NW2];            // (C = 288: dead — the chunk loop stages through locals)
  f32x16_t out[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[f][r] = 0.f;
  if constexpr (PF) mlp_fetch<C>(p, 0, tid, pw1, pw2);
  for (int j = 0; j < nch; ++j) {
    __syncthreads();                       // the previous chunk's fragments (first pass: the x tile's) have been read
    if constexpr (PF) {
      mlp_stage<C>(W1c, W2c, tid, pw1, pw2);
    } else {
      // no registers for a whole chunk beside 144 accumulator and 72 fragment registers: W1's and W2's pieces go through the same few registers
      // one after the other (the CU's second workgroup covers the round trips)
      constexpr int CPRc = C / 8;
#pragma unroll
      for (int i = 0; i < NW1; ++i) {
        const int idx = tid + i * 256;
        if (idx < 32 * CPRc) {
          const int row = idx / CPRc, c = idx - row * CPRc;
          *(u32x4_t*)(W1c + row * RSX + c * 16) = *(const u32x4_t*)(p.w1 + (int64_t)j * 32 * C + idx * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < NW2; ++i) {
        const int idx = tid + i * 256;
        if (idx < C * 4) {
          const int row = idx >> 2, qd = idx & 3;
          const u32x4_t r = *(const u32x4_t*)(p.w2 + (int64_t)row * H + j * 32 + qd * 8);
          const uint2 lo = {r[0], r[1]}, hi = {r[2], r[3]};
          *(uint2*)(W2c + row * RSW2 + qd * 16) = lo;
          *(uint2*)(W2c + row * RSW2 + qd * 16 + 8) = hi;
        }
      }
    }
    __syncthreads();
    if constexpr (PF) {
      if (j + 1 < nch) mlp_fetch<C>(p, j + 1, tid, pw1, pw2);
    }
    // fc1: hidden[n, m] = W1c[n, :] . xn[m, :]  (a-operand rows n = l31, b-operand rows m = l31): lane = row m, register r = hidden 8 g + 4 h + jj
    f32x16_t a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a1[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4_t wf = *(const u32x4_t*)(W1c + l31 * RSX + s * 32 + h * 16);
      MmaOp<bf16_t>::run(wf, xq[s], a1);
    }
    u32x4_t pb[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4_t bb = *(const f32x4_t*)(p.b1 + j * 32 + 8 * g + 4 * h);
      const float v0 = vg_gelu_erf(a1[4 * g] + bb[0]), v1 = vg_gelu_erf(a1[4 * g + 1] + bb[1]);
      const float v2 = vg_gelu_erf(a1[4 * g + 2] + bb[2]), v3 = vg_gelu_erf(a1[4 * g + 3] + bb[3]);
      pb[g >> 1][(g & 1) * 2] = f2bf2(v0, v1);
      pb[g >> 1][(g & 1) * 2 + 1] = f2bf2(v2, v3);
    }
    // fc2: out[c, m] += W2c[c, n] . hidden[m, n] over the chunk's 32 n, in the accumulator's register order: k-step s2, lane half h holds
    // n = 16 s2 + 4 h + {0..3} and 16 s2 + 8 + 4 h + {0..3}
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int crow = min(f * 32 + l31, C - 1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint2 lo = *(const uint2*)(W2c + crow * RSW2 + (16 * s2 + 4 * h) * 2);
        const uint2 hi = *(const uint2*)(W2c + crow * RSW2 + (16 * s2 + 8 + 4 * h) * 2);
        const u32x4_t wf = {lo.x, lo.y, hi.x, hi.y};
        MmaOp<bf16_t>::run(wf, pb[s2], out[f]);
      }
    }
  }
  __syncthreads();                         // every wave is done with the last chunk: the LDS becomes the epilogue's slabs

  // ---- epilogue: + b2, + x, bf16.  Per 32-column fragment a wave-private 32 x 32 fp32 slab turns the accumulator layout (lane = row, four
  // consecutive columns per register group) into row-major pieces of eight columns: one 16-byte residual load and one 16-byte store per lane
  char* slab = smem + wave * 4096;
  const int wkey = (l31 >> 1) & 7;
  const int mw = m0 + wave * 32;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int c0w = f * 32;
    u32x4_t rv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m = mw + k * 16 + (lane >> 2), col = c0w + (lane & 3) * 8;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      rv[k] = (m < M && col < C) ? *(const u32x4_t*)(p.x + (int64_t)m * p.ldx + col) : z;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = c0w + 8 * g + 4 * h;
      f32x4_t bb = {0.f, 0.f, 0.f, 0.f};
      if (c < C) bb = *(const f32x4_t*)(p.b2 + c);
      const f32x4_t v = {out[f][4 * g] + bb[0], out[f][4 * g + 1] + bb[1], out[f][4 * g + 2] + bb[2], out[f][4 * g + 3] + bb[3]};
      *(f32x4_t*)(slab + l31 * 128 + (((2 * g + h) ^ wkey) << 4)) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = k * 16 + (lane >> 2), c8 = lane & 3, rkey = (row >> 1) & 7;
      const f32x4_t x0 = *(const f32x4_t*)(slab + row * 128 + (((2 * c8) ^ rkey) << 4));
      const f32x4_t x1 = *(const f32x4_t*)(slab + row * 128 + (((2 * c8 + 1) ^ rkey) << 4));
      const int m = mw + row, col = c0w + c8 * 8;
      if (m < M && col < C) {
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = f2bf2(v[2 * e] + __uint_as_float(rv[k][e] << 16), v[2 * e + 1] + __uint_as_float(rv[k][e] & 0xffff0000u));
        *(u32x4_t*)(p.y + (int64_t)m * p.ldy + col) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int C>
int launch_mlp(const MlpArgs& a, hipStream_t st) {
  constexpr int RSX = C * 2 + 16;
  constexpr int lds_x = 128 * RSX, lds_w = 32 * RSX + C * 72, lds_e = 4 * 4096;
  constexpr int lds = lds_x > lds_w ? (lds_x > lds_e ? lds_x : lds_e) : (lds_w > lds_e ? lds_w : lds_e);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)mlp_rows_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  mlp_rows_kernel<C><<<dim3((a.M + 127) / 128), 256, lds, st>>>(a);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

}  // namespace

// C = 288 (Hiera stage 2) has an instantiation and passes the same tests, but 144 accumulator + 72 fragment registers per lane leave nothing for the
// weight staging (36 spilled registers at two workgroups per CU): measured r05 at M = 262144: 867 us against 717 us for the three launches — not
// routed.  C = 144 (stage 1, M = 1048576): 842 us against 1214 us.
extern "C" int vg_mlp_rows_supported(int C, int H) { return C == 144 && H > 0 && H % 32 == 0; }

extern "C" int vg_mlp_rows(const void* x, int64_t ldx, void* y, int64_t ldy, const float* ln_w, const float* ln_b, float eps, const void* w1,
                           const float* b1, const void* w2, const float* b2, int64_t M, int C, int H, int dtype, vg_stream_t stream) {
  VG_CHECK(x && y && ln_w && ln_b && w1 && b1 && w2 && b2, VG_ERR_ARG, "vg_mlp_rows: null pointer");
  VG_CHECK(dtype == VG_BF16, VG_ERR_UNSUPPORTED, "vg_mlp_rows: bf16 only (the fp32 parity mode runs vg_layernorm + vg_gemm x 2)");
  VG_CHECK((C == 144 || C == 288) && H > 0 && H % 32 == 0, VG_ERR_UNSUPPORTED, "vg_mlp_rows: C in {144, 288} and H %% 32 == 0 (C=%d H=%d)", C, H);
  VG_CHECK(M >= 0 && M < ((int64_t)1 << 31) && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
               ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)ln_w & 15) == 0 && ((uintptr_t)ln_b & 15) == 0 &&
               ((uintptr_t)b1 & 15) == 0 && ((uintptr_t)b2 & 15) == 0,
           VG_ERR_ARG, "vg_mlp_rows: rows / vectors must be 16-byte aligned");
  if (M == 0) return VG_OK;
  MlpArgs a{(const bf16_t*)x, (bf16_t*)y, ldx, ldy, ln_w, ln_b, eps, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (int)M, H};
  return C == 144 ? launch_mlp<144>(a, (hipStream_t)stream) : launch_mlp<288>(a, (hipStream_t)stream);
}
