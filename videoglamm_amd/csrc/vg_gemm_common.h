// Pieces shared by the GEMM translation units (vg_gemm.hip, vg_gemm_p8.hip): the launch record, the XCD-aware tile walk, the MFMA wrappers and
// the straight-line row-major side of the LDS-staged epilogues.  See vg_gemm.hip's header for the kernel list.
#pragma once
#include "vg_common.h"
#include <stdlib.h>

struct GemmArgs {
  const void* A; const void* W; void* C; const float* bias; const float* gamma; const void* R;
  int64_t lda, ldw, ldc, ldr, sA, sW, sC, sR;
  int M, N, K, act;
  int vec_out;  // C/R rows are 16-byte aligned: the epilogue stores whole 16-byte chunks
  int a_op;     // 1: W holds gate|up rows ([2N,K]); output column n = silu(A.gate_n)*(A.up_n) (skinny path only)
  int gn;       // tile order: column groups of gn N-tiles, M-tiles fastest-but-one inside a group (see gemm_tile_of)
  // window-order <-> image-order row maps (vg_gemm_window): wmode 1 = A rows are gathered from an image-order tensor
  // (window_partition fused into the load), 2 = C and R rows are scattered to / read from image order (window_unpartition
  // + residual add fused into the epilogue).  GEMM row m is always the window-order index.
  int wmode, wH, wW, wws, wnH, wnW;
  int wsh;      // >= 0: ws, wnW, wnH are powers of two, their log2 packed as ws | nW << 8 | nH << 16 (gemm_window_row's shift path); -1: divide
  const void* zrow;   // K zeros: source of the padded window rows (the LDS-DMA cannot zero-fill)
  // fp8 operands (vg_gemm_f8): C = (A8 . W8^T) * sa[m] * sw[n] — one fp32 scale per A row (token) and per W row (output)
  const float* sa; const float* sw;
  // split-K (vg_gemm_splitk, 128x128 LDS-DMA kernel only): blockIdx.z = K slice of kchunk elements; raw fp32 partial tiles
  // go to part[z][M][N], a second kernel sums them and applies the epilogue
  int ksplit, kchunk; float* part;
  int nbatch;   // persistent 256x256 kernel: batch count (tiles of all batch entries form one queue)
  int nt;       // output tiles leave with non-temporal (streaming) stores: large outputs whose rows are whole 64-byte sectors (launch_gemm)
  // LayerNorm over K in front of the contraction (vg_gemm_ln; the row-register kernel only): A rows are normalised as they arrive in registers
  const float* ln_w; const float* ln_b; float ln_eps;
};

// window-order row m -> image-order row, or -1 for a padding row (backbones/utils.py:16-38 window_partition).
// r03: when the window side and the window counts are powers of two (every Hiera stage of a 1024^2 input: 8 / 4 / 16 / 8-token windows on
// 256 / 128 / 64 / 32-token grids) the five integer divisions — ~25 instructions each on this ISA, per ROW: two to four rows per lane in the
// gathering prologue, eight per lane and tile in the scattering epilogue, on kernels that are bound by instruction issue — are shifts and
// masks (p.wsh: log2 of ws | nW << 8 | nH << 16, or -1).
__device__ __forceinline__ int64_t gemm_window_row(const GemmArgs& p, int m) {
  if (p.wsh >= 0) {
    const int ws_sh = p.wsh & 0xff, nw_sh = (p.wsh >> 8) & 0xff, nh_sh = (p.wsh >> 16) & 0xff;
    const int win = m >> (2 * ws_sh), tok = m & ((1 << (2 * ws_sh)) - 1);
    const int rr = tok >> ws_sh, cc = tok & ((1 << ws_sh) - 1);
    const int wx = win & ((1 << nw_sh) - 1), t = win >> nw_sh;
    const int wy = t & ((1 << nh_sh) - 1), b = t >> nh_sh;
    const int y = (wy << ws_sh) + rr, x = (wx << ws_sh) + cc;
    return (y < p.wH && x < p.wW) ? ((int64_t)b * p.wH + y) * p.wW + x : -1;
  }
  const int per = p.wws * p.wws;
  const int win = m / per, tok = m - win * per;
  const int rr = tok / p.wws, cc = tok - rr * p.wws;
  const int wx = win % p.wnW, t = win / p.wnW;
  const int wy = t % p.wnH, b = t / p.wnH;
  const int y = wy * p.wws + rr, x = wx * p.wws + cc;
  return (y < p.wH && x < p.wW) ? ((int64_t)b * p.wH + y) * p.wW + x : -1;
}

// Linear tile id (already XCD-remapped: every XCD owns a contiguous run) -> (bm, bn).  Tiles are walked in column
// groups `gn` N-tiles wide, row by row inside a group, so that the ~64 tiles an XCD runs concurrently form a roughly
// square gn x (64/gn) patch and share their A row-panels and W column-panels through that XCD's L2.  A plain row-major
// walk makes those 64 tiles ONE row of 64 N-tiles: every W panel is then fetched once per M-tile (measured r01: the
// Llama gate|up GEMM pulled 3.2 GB through the fabric for 250 MB of operands).
__device__ __forceinline__ void gemm_tile_of(int wgid, int mt, int nt, int gn, int& bm, int& bn) {
  const int per = gn * mt;
  const int g = wgid / per, r = wgid - g * per;
  const int w = min(gn, nt - g * gn);
  bm = r / w;
  bn = g * gn + (r - bm * w);
}

template <typename T> struct MmaOp;
template <> struct MmaOp<bf16_t> {
  static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaOp<float> {
  static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
__device__ __forceinline__ i32x8_t f8_operand(const u32x4_t& lo, const u32x4_t& hi) {
  i32x8_t v = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  return v;
}

// Output-tile store flavour of the 256x256 kernel's epilogue (A/B macro): 0 plain, 1 nontemporal (nt), 2 write-through (sc0 sc1: the
// line is not kept in the XCD's L2), 3 sc1 only
__device__ __forceinline__ void epi_store16(void* ptr, const u32x4_t& v, int nt = 0) {
  // nt (wave-uniform): a streaming store.  Measured r03 (tools/bench_gemm.py with VG_BENCH_ACT, same-box A/B against the r02 library) on
  // Hiera's stage-2 GEMMs, whose outputs (150-600 MB) are read by the next kernel long after they have left the caches: the 128x128
  // kernels are bound by their output stores (a build without them: -35...50 %), and with nt stores qkv runs 389 -> 252 us, fc1 487 -> 338,
  // proj 132 -> 112; stage 1's fc1 773 -> 646.  Rows that are not whole 64-byte sectors (N = 432, 144: partial sectors want the cache to
  // merge them) lose 3-7 %, hence the rule in launch_gemm.  Inline asm: hipcc merges "if (nt) __builtin_nontemporal_store(...) else
  // plain store" into one plain store — the hint is dropped.
  if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(v) : "memory");
  else *(u32x4_t*)ptr = v;
}

// Row-major side of the LDS-staged epilogues, fast path: the lane's 8 columns are whole inside N, 16-byte aligned rows, no fp8 scales.
// NP passes of RPP rows (rows m0 + RPP pass + rsub; RPP = 64 / lanes per row).  Measured r02 on the 256x256 kernel (Hiera stage-3 fc1, M = 65536): the previous
// per-element form (ds_read_b32 + the activation's branch chain per element, a conditional residual load per pass that made every
// pass wait for the previous pass's stores — loads and stores share vmcnt on gfx9) cost 90 of the GEMM's 287 us.  Here: every
// residual load of the block is requested before its first store, the staged tile is read with two ds_read_b128 per row, the
// activation and the presence of a residual are compile-time constants (one switch per block, epi_dispatch).
template <typename TO, int ACT, bool RES, int NP, int ES, int RPP = 8, bool GAM = true>
__device__ __forceinline__ void epi_rows_fast(const GemmArgs& p, const float* ws, int m0, int n0, int cg, int rsub,
                                              const float (&bv)[8], const float (&gv)[8], TO* C, const TO* R) {
  int64_t mo[NP];
  bool ok[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int m = m0 + ps * RPP + rsub;
    ok[ps] = m < p.M;
    mo[ps] = m;
    if (p.wmode == 2) {
      mo[ps] = ok[ps] ? gemm_window_row(p, m) : -1;
      ok[ps] = mo[ps] >= 0;
    }
  }
  constexpr int RW = RES ? (sizeof(TO) == 2 ? 1 : 2) : 1;
  u32x4_t rv[NP][RW];
  if constexpr (RES) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
      for (int w = 0; w < RW; ++w) {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        rv[ps][w] = ok[ps] ? *(const u32x4_t*)((const char*)(R + mo[ps] * p.ldr + n0) + 16 * w) : z;
      }
  }
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const float* row = ws + (ps * RPP + rsub) * ES + cg * 8;
    const f32x4_t x0 = *(const f32x4_t*)row, x1 = *(const f32x4_t*)(row + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = vg_act(x0[e] + bv[e], ACT);
      v[4 + e] = vg_act(x1[e] + bv[4 + e], ACT);
      if constexpr (GAM) { v[e] *= gv[e]; v[4 + e] *= gv[4 + e]; }        // LayerScale: only the (none, residual) variant carries it
    }
    TO* cp = C + mo[ps] * p.ldc + n0;
    if constexpr (sizeof(TO) == 2) {
      if constexpr (RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(rv[ps][0][e] << 16); v[2 * e + 1] += __uint_as_float(rv[ps][0][e] & 0xffff0000u); }
      }
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf2(v[2 * e], v[2 * e + 1]);
      if (ok[ps]) epi_store16(cp, o, p.nt);
    } else {
      if constexpr (RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(rv[ps][0][e]); v[4 + e] += __uint_as_float(rv[ps][1][e]); }
      }
      const f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
      if (ok[ps]) {
        *(f32x4_t*)cp = o0;
        *(f32x4_t*)(cp + 4) = o1;
      }
    }
  }
}

template <int V> struct epi_ic { static constexpr int value = V; };
// one switch per block: f(activation constant, has-residual constant) -> true; false = no straight-line variant for this combination, the
// caller's general path takes the block.  r03: only the combinations the path runs are instantiated — none (+- residual), GELU, quick-GELU, ReLU
// without residual (SiLU / sigmoid / activation + residual epilogues: the mask decoder's IoU head on > 16 rows, nothing else) — 5 variants per
// kernel instead of 12.  Speed-neutral on C2 (Hiera 110.2 / 109.7 -> 109.9 / 110.1 ms same-box), vg_gemm.hip compiles in 1 min instead of 2.5.
template <typename F>
__device__ __forceinline__ bool epi_dispatch(int act, bool res, bool gam, F&& f) {
  // LayerScale (gamma) comes with "no activation + residual" only (InternVideo2's ls1 / ls2, the memory encoder's fuser): that variant alone
  // multiplies by it; the others do not carry the 8 multiplies per 8 outputs
  if (gam) {
    if (act != VG_ACT_NONE || !res) return false;
    f(epi_ic<0>{}, epi_ic<1>{}, epi_ic<1>{});
    return true;
  }
#define VG_EPI_CASE(A) \
  case A:              \
    if (res) return false; \
    f(epi_ic<A>{}, epi_ic<0>{}, epi_ic<0>{}); \
    return true;
  switch (act) {
    VG_EPI_CASE(VG_ACT_GELU)
    VG_EPI_CASE(VG_ACT_QUICK_GELU)
    VG_EPI_CASE(VG_ACT_RELU)
    case VG_ACT_NONE:
      if (res) f(epi_ic<0>{}, epi_ic<1>{}, epi_ic<0>{}); else f(epi_ic<0>{}, epi_ic<0>{}, epi_ic<0>{});
      return true;
    default:
      return false;
  }
#undef VG_EPI_CASE
}

// vg_gemm_p8.hip: the phase-split 256x256-tile bf16 kernel (launched from launch_gemm's 256x256 route)
bool vg_gemm_p8_eligible(const GemmArgs& p, int batch);
bool vg_gemm_p8_window_ok(int wmode, int wsh, int wH, int wW, int wws);
int vg_gemm_p8_launch(const GemmArgs& q, int out_is_bf16, int wgs, hipStream_t st);
// vg_gemm_p8n.hip: the same pipeline on 256 x 192 tiles (launch_gemm's route_p8n decides)
int vg_gemm_p8n_launch(const GemmArgs& q, int out_is_bf16, int wgs, hipStream_t st);
// vg_gemm_rr.hip: the row-register kernel for K = 144 / 288 over very many rows (A rows in registers, W chunks in LDS; launch_gemm's short-K route)
bool vg_gemm_rr_eligible(const GemmArgs& p, int batch, bool out_is_bf16);
int vg_gemm_rr_launch(const GemmArgs& q, int ncu, hipStream_t st);
