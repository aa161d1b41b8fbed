// vg_gemm_rows: C = act(pro(A) . W^T + bias) [axial RoPE on leading columns] [+ R] for short rows (K = 64 / 128 / 192 / 256, bf16) — the
// whole K of a 64-row operand tile sits in LDS (gemm_small64_kernel's staging, vg_gemm.hip), so the row-wise producer of A and the
// column-pair consumer of C ride in the same launch:
//   pro = LayerNorm over the K columns (two-pass fp32 statistics, the result rounded to bf16: the arithmetic of vg_layernorm followed by
//         vg_gemm), or A + A2 (A2 broadcast over row blocks: memory + positional encoding), rounded to bf16 like vg_axpby's output;
//   RoPE = vg_rope_axial_heads on the product's first `rope_cols` columns (heads of `rope_ch` channels, rows [r0, r1) of every block of `rpb`
//          rows, token = (row - r0) % grid), applied to the bf16-rounded value like the separate launch.
// What it replaces on the SAM2 video path (one launch instead of two or three, r05): norm1 -> q|k|v projection -> RoPE of q | k, norm2 -> q
// projection -> RoPE, (memory + memory_pos) -> the four layers' k projections -> RoPE of the non-pointer rows, norm3 -> linear1 -> ReLU
// (R/.../sam2/modeling/memory_attention.py:60-99, sam/transformer.py:289-327), LayerNorm -> pwconv1 -> GELU of the memory fuser
// (memory_encoder.py:96-118).
#include "vg_gemm_common.h"

namespace {

struct RowsArgs {
  GemmArgs g;
  const float* ln_w; const float* ln_b; float ln_eps;
  const void* A2; int64_t lda2; int a2_rows;
  int64_t a_block_stride;       // != 0: row m of A lives at block (m / rpb) * a_block_stride + (m % rpb) * lda (a strided [B, rows, K] view)
  const float* cs; const float* sn;
  int rope_cols, rope_ch, rpb, r0, r1, grid;
  int ntile;                    // 64-column tiles per workgroup: the A tile is staged (and normalised) once and walks `ntile` W tiles
};

enum { PRO_NONE = 0, PRO_LN = 1, PRO_ADD = 2 };

template <int PRO, bool ROPE>
__global__ __launch_bounds__(256) void gemm_rows64_kernel(RowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  typedef bf16_t TO;
  const GemmArgs& p = a.g;
  constexpr int KPC = 8, SEG = 64 * 128;          // one 64-element K segment of a 64-row operand tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int bm = blockIdx.y;
  const int M = p.M, N = p.N, nseg = p.K / 64;
  const T* A = (const T*)p.A;
  const T* W = (const T*)p.W;
  // LDS: A tile | W tile | fp32 staging of the epilogue (separate: the A tile lives across the workgroup's W tiles)
  char* smW = smem + nseg * SEG;
  float* stg = (float*)(smem + 2 * nseg * SEG);
  auto load_w = [&](int bn) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wave * 16 + i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int gn = bn * 64 + row;
      gn = gn < N ? gn : N - 1;
      const T* wsrc = W + (int64_t)gn * p.ldw + chunk * KPC;
      char* dw = smW + wave * 16 * 128 + i * 1024;
      for (int sg = 0; sg < nseg; ++sg)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + sg * 64),
                                         (__attribute__((address_space(3))) void*)(dw + sg * SEG), 16, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 16 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gm = bm * 64 + row;
    gm = gm < M ? gm : M - 1;
    const T* as = A + (a.a_block_stride ? (int64_t)(gm / a.rpb) * a.a_block_stride + (int64_t)(gm % a.rpb) * p.lda : (int64_t)gm * p.lda) + chunk * KPC;
    char* da = smem + wave * 16 * 128 + i * 1024;
    for (int sg = 0; sg < nseg; ++sg)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(as + sg * 64),
                                       (__attribute__((address_space(3))) void*)(da + sg * SEG), 16, 0, 0);
  }
  const int bn0 = blockIdx.x * a.ntile;
  load_w(bn0);
  // the prologue's own operands are requested while the tiles are in flight: thread (row, part) owns 16-byte chunks 2 part, 2 part + 1 of every
  // K segment of its row (the four threads of a row hit different LDS slots)
  const int prow = tid >> 2, part = tid & 3, pkey = (prow >> 1) & 7;
  u32x4_t add2[4][2];
  if constexpr (PRO == PRO_ADD) {
    int gm = bm * 64 + prow;
    gm = gm < M ? gm : M - 1;
    const T* a2 = (const T*)a.A2 + (int64_t)(gm % a.a2_rows) * a.lda2;
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
      if (sg < nseg) {
#pragma unroll
        for (int j = 0; j < 2; ++j) add2[sg][j] = *(const u32x4_t*)(a2 + sg * 64 + (2 * part + j) * KPC);
      }
  }
  const int ra = wm * 32 + l31, rb = wn * 32 + l31;
  const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (PRO != PRO_NONE) {
    u32x4_t x[4][2];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
      if (sg < nseg) {
#pragma unroll
        for (int j = 0; j < 2; ++j) x[sg][j] = *(const u32x4_t*)(smem + sg * SEG + prow * 128 + (((2 * part + j) ^ pkey) << 4));
      }
    if constexpr (PRO == PRO_LN) {
      auto red4 = [](float v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); return v; };
      const float invK = 1.0f / (float)p.K;
      float s = 0.f;
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
        if (sg < nseg) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __uint_as_float(x[sg][j][e] << 16) + __uint_as_float(x[sg][j][e] & 0xffff0000u);
        }
      const float mean = red4(s) * invK;
      float q = 0.f;
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
        if (sg < nseg) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d0 = __uint_as_float(x[sg][j][e] << 16) - mean, d1 = __uint_as_float(x[sg][j][e] & 0xffff0000u) - mean;
              q += d0 * d0 + d1 * d1;
            }
        }
      const float rstd = rsqrtf(red4(q) * invK + a.ln_eps);
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
        if (sg < nseg) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int k0 = sg * 64 + (2 * part + j) * KPC;
            const f32x4_t w0 = *(const f32x4_t*)(a.ln_w + k0), w1 = *(const f32x4_t*)(a.ln_w + k0 + 4);
            const f32x4_t b0 = *(const f32x4_t*)(a.ln_b + k0), b1 = *(const f32x4_t*)(a.ln_b + k0 + 4);
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float n0 = (__uint_as_float(x[sg][j][e] << 16) - mean) * rstd, n1 = (__uint_as_float(x[sg][j][e] & 0xffff0000u) - mean) * rstd;
              const float g0 = e < 2 ? w0[2 * e] : w1[2 * e - 4], g1 = e < 2 ? w0[2 * e + 1] : w1[2 * e - 3];
              const float c0 = e < 2 ? b0[2 * e] : b1[2 * e - 4], c1 = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
              o[e] = f2bf2(n0 * g0 + c0, n1 * g1 + c1);
            }
            *(u32x4_t*)(smem + sg * SEG + prow * 128 + (((2 * part + j) ^ pkey) << 4)) = o;
          }
        }
    } else {
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
        if (sg < nseg) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              o[e] = f2bf2(__uint_as_float(x[sg][j][e] << 16) + __uint_as_float(add2[sg][j][e] << 16),
                           __uint_as_float(x[sg][j][e] & 0xffff0000u) + __uint_as_float(add2[sg][j][e] & 0xffff0000u));
            *(u32x4_t*)(smem + sg * SEG + prow * 128 + (((2 * part + j) ^ pkey) << 4)) = o;
          }
        }
    }
    __syncthreads();
  }
  TO* C = (TO*)p.C;
  const TO* R = (const TO*)p.R;
  constexpr int ES = 36;
  float* ws = stg + wave * 32 * ES;
  const int cg = lane & 3, rsub = lane >> 2;          // 4 column groups x 16 rows per pass
  const int m0w = bm * 64 + wm * 32;
  for (int jt = 0; jt < a.ntile; ++jt) {
    const int bn = bn0 + jt;
    if (bn * 64 >= N) break;                          // (workgroup-uniform)
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int sg = 0; sg < nseg; ++sg) {
      const char* sa = smem + sg * SEG + ra * 128;
      const char* sb = smW + sg * SEG + rb * 128;
      u32x4_t fa[4], fb[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 2 * g + h;
        fa[g] = *(const u32x4_t*)(sa + ((c ^ swa) << 4));
        fb[g] = *(const u32x4_t*)(sb + ((c ^ swb) << 4));
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) MmaOp<T>::run(fa[g], fb[g], acc);
    }
    __syncthreads();                                  // every wave has read this W tile: the next one may land under the epilogue
    const bool more = jt + 1 < a.ntile && (bn + 1) * 64 < N;
    if (more) load_w(bn + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) ws[mfma32_row(r, h) * ES + l31] = acc[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the staging slab is wave-private)
    const int n0w = bn * 64 + wn * 32;
    const int n0 = n0w + cg * 8;
    float bv[8], gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bv[e] = p.bias ? p.bias[n0 + e] : 0.f;
      gv[e] = 1.f;
    }
    if constexpr (ROPE) {
      if (n0 < a.rope_cols) {          // the lane's eight columns = four rotation pairs of one head (rope_cols, rope_ch multiples of 8)
        const int hc = a.rope_ch >> 1, j0 = (n0 % a.rope_ch) >> 1;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          float* rowp = ws + (ps * 16 + rsub) * ES + cg * 8;
          const int r = (m0w + ps * 16 + rsub) % a.rpb;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(rowp[e] + bv[e]));      // the projection's bf16 output: what the separate RoPE launch reads
          if (r >= a.r0 && r < a.r1) {
            const int tok = (r - a.r0) % a.grid;
            const f32x4_t c = *(const f32x4_t*)(a.cs + (int64_t)tok * hc + j0), sv = *(const f32x4_t*)(a.sn + (int64_t)tok * hc + j0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x0 = v[2 * e], x1 = v[2 * e + 1];
              v[2 * e] = x0 * c[e] - x1 * sv[e];
              v[2 * e + 1] = x0 * sv[e] + x1 * c[e];
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) rowp[e] = v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
      }
    }
    epi_dispatch(p.act, R != nullptr, false, [&](auto act, auto res, auto gam) {
      epi_rows_fast<TO, decltype(act)::value, decltype(res)::value != 0, 2, ES, 16, false>(p, ws, m0w, n0, cg, rsub, bv, gv, C, R);
    });
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

template <int PRO, bool ROPE>
int launch_rows(const RowsArgs& a, hipStream_t st) {
  const int nseg = a.g.K / 64;
  const int lds = nseg * 2 * 64 * 128 + 4 * 32 * 36 * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_rows64_kernel<PRO, ROPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  // column tiles per workgroup: as many as keep every resident workgroup slot (1 per CU at K = 256, 4 at K = 64) busy — the A tile (and its LayerNorm) is then staged once per
  // `ntile` outputs tiles instead of once per tile (measured r05, norm3 -> linear1 -> ReLU at M = 4096, N = 2048: 33.8 us with one tile each)
  const int mt = (a.g.M + 63) / 64, nt = a.g.N / 64;
  const int per_cu = 160 * 1024 / lds < 1 ? 1 : (160 * 1024 / lds > 4 ? 4 : 160 * 1024 / lds);      // resident workgroups per CU at this K
  int ntile = (int)(((int64_t)mt * nt) / (256 * per_cu));
  ntile = ntile < 1 ? 1 : (ntile > nt ? nt : (ntile > 16 ? 16 : ntile));
  RowsArgs b = a;
  b.ntile = ntile;
  dim3 grid((nt + ntile - 1) / ntile, mt, 1);
  gemm_rows64_kernel<PRO, ROPE><<<grid, 256, lds, st>>>(b);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

}  // namespace

extern "C" int vg_gemm_rows(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, const void* R, int64_t ldr,
                            int64_t M, int N, int K, int act, const float* ln_w, const float* ln_b, float ln_eps, const void* A2, int64_t lda2,
                            int a2_rows, const float* rope_cos, const float* rope_sin, int rope_cols, int rope_ch, int rows_per_block, int64_t a_block_stride,
                            int rope_r0, int rope_r1, int rope_grid, int dtype, vg_stream_t stream) {
  VG_CHECK(A && W && C, VG_ERR_ARG, "vg_gemm_rows: null pointer");
  VG_CHECK(dtype == VG_BF16, VG_ERR_UNSUPPORTED, "vg_gemm_rows: bf16 only (the fp32 parity mode runs the separate launches)");
  VG_CHECK(M >= 0 && M < ((int64_t)1 << 31) && N > 0 && N % 64 == 0 && (K == 64 || K == 128 || K == 192 || K == 256), VG_ERR_ARG,
           "vg_gemm_rows: needs N %% 64 == 0 and K in {64, 128, 192, 256} (M=%lld N=%d K=%d)", (long long)M, N, K);
  if (M == 0) return VG_OK;
  VG_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && (!R || ldr % 8 == 0) && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
               ((uintptr_t)C & 15) == 0 && (!R || ((uintptr_t)R & 15) == 0),
           VG_ERR_ARG, "vg_gemm_rows: rows must be 16-byte aligned");
  VG_CHECK(!(ln_w || ln_b) || !A2, VG_ERR_ARG, "vg_gemm_rows: one prologue at a time (LayerNorm or A + A2)");
  VG_CHECK(!(ln_w || ln_b) || (ln_w && ln_b), VG_ERR_ARG, "vg_gemm_rows: LayerNorm needs weight and bias");
  VG_CHECK(!A2 || (a2_rows > 0 && lda2 % 8 == 0 && ((uintptr_t)A2 & 15) == 0), VG_ERR_ARG, "vg_gemm_rows: bad A2");
  VG_CHECK(a_block_stride == 0 || (rows_per_block > 0 && a_block_stride % 8 == 0 && M % rows_per_block == 0), VG_ERR_ARG, "vg_gemm_rows: bad A block geometry");
  const bool rope = rope_cos != nullptr;
  VG_CHECK(!rope || (rope_sin && rope_cols > 0 && rope_cols <= N && rope_cols % 8 == 0 && rope_ch >= 8 && rope_ch % 8 == 0 && rope_cols % rope_ch == 0 &&
                     rows_per_block > 0 && rope_r0 >= 0 && rope_r0 <= rope_r1 && rope_r1 <= rows_per_block && rope_grid > 0),
           VG_ERR_ARG, "vg_gemm_rows: bad RoPE geometry");
  VG_CHECK(act == VG_ACT_NONE || (!R && (act == VG_ACT_GELU || act == VG_ACT_RELU || act == VG_ACT_QUICK_GELU)), VG_ERR_UNSUPPORTED,
           "vg_gemm_rows: epilogues are none (+ residual), GELU, quick-GELU, ReLU");
  VG_CHECK(!rope || (act == VG_ACT_NONE && !R), VG_ERR_UNSUPPORTED, "vg_gemm_rows: RoPE comes without activation / residual");
  RowsArgs a{};
  a.g = GemmArgs{A, W, C, bias, nullptr, R, lda, ldw, ldc, ldr, 0, 0, 0, 0, (int)M, N, K, act, 1, 0, 1, 0, 0, 0, 0, 0, 0, -1, nullptr};
  a.ln_w = ln_w; a.ln_b = ln_b; a.ln_eps = ln_eps;
  a.A2 = A2; a.lda2 = lda2; a.a2_rows = a2_rows; a.a_block_stride = a_block_stride;
  a.cs = rope_cos; a.sn = rope_sin; a.rope_cols = rope_cols; a.rope_ch = rope_ch; a.rpb = rows_per_block; a.r0 = rope_r0; a.r1 = rope_r1; a.grid = rope_grid;
  hipStream_t st = (hipStream_t)stream;
  if (ln_w) return rope ? launch_rows<PRO_LN, true>(a, st) : launch_rows<PRO_LN, false>(a, st);
  if (A2) return rope ? launch_rows<PRO_ADD, true>(a, st) : launch_rows<PRO_ADD, false>(a, st);
  return rope ? launch_rows<PRO_NONE, true>(a, st) : launch_rows<PRO_NONE, false>(a, st);
}
