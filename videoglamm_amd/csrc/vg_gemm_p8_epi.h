// Pieces of the phase-split GEMM kernels' wave-private epilogue shared by vg_gemm_p8.hip (256x256 tile) and vg_gemm_p8n.hip (256x192 tile):
// the packed-bf16 flush of a 32-row slab, the window-scatter row map, small loaders.  See vg_gemm_p8.hip for the epilogue's design notes.
#pragma once
#include "vg_gemm_common.h"

#define P8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

namespace {

__device__ __forceinline__ void p8_load4(const float* src, int c0, int N, float dflt, float (&o)[4]) {
  if (src && c0 + 4 <= N) {
    const f32x4_t x = *(const f32x4_t*)(src + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = x[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (src && c0 + e < N) ? src[c0 + e] : dflt;
  }
}
template <typename TO>
__device__ __forceinline__ void p8_store_tail(TO* cp, const float* v, int nvalid) {
  for (int e = 0; e < nvalid; ++e) vg_elt<TO>::st(cp + e, v[e]);
}
__device__ __forceinline__ void p8_wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// wmode 2 (vg_gemm_window's scatter: window_unpartition + residual add in the epilogue) for power-of-two windows that tile the image: GEMM row
// m = [b, wy, wx | rr | cc] goes to image row [b, wy, rr | wx | cc] — a bit-field swap (see setup()).  Wave-uniform branch.
__device__ __forceinline__ int p8_out_row(const GemmArgs& p, int m) {
  if (p.wmode == 2) {
    const int a = p.wsh & 0xff, nw = (p.wsh >> 8) & 0xff;
    const int rr = (m >> a) & ((1 << a) - 1), wx = (m >> (2 * a)) & ((1 << nw) - 1);
    m = (m & ~((((1 << (a + nw)) - 1)) << a)) | (wx << a) | (rr << (a + nw));
  }
  return m;
}

// bf16 output without a residual: packed staging, one pass per 32-row fragment (NC = 64 columns per wave; the SwiGLU form: 32).
// INTERIOR (wave-uniform, chosen once per tile): the wave's 128 x NC block lies inside M x N — no per-lane bounds code at all.
template <int NC, bool INTERIOR>
__device__ __forceinline__ void p8_flush_packed(const GemmArgs& p, const char* slab, char* cbase, int64_t rstride, int m0, int n0w, int lane) {
  // slab: 32 rows x NC bf16, 16-byte chunk c of row r at slot c ^ key(r); cbase: this lane's (row lane / CPR, chunk lane % CPR) of the pass's first rows
  constexpr int RB = NC * 2, CPR = NC / 8, RPI = 64 / CPR;      // row bytes, 16-byte chunks per row, rows per read instruction
  p8_wave_lds_fence();
  u32x4_t d[32 / RPI];
#pragma unroll
  for (int k = 0; k < 32 / RPI; ++k) {
    const int row = k * RPI + lane / CPR, c = lane % CPR;
    const int key = NC == 64 ? (row >> 1) & 7 : (row >> 1) & 3;
    d[k] = *(const u32x4_t*)(slab + row * RB + ((c ^ key) << 4));
  }
#pragma unroll
  for (int k = 0; k < 32 / RPI; ++k) {
    char* cp = cbase + k * RPI * rstride;
    if (p.wmode == 2) {        // scattered rows: cbase belongs to row m0 + lane / CPR (the caller's), this pass's row is k * RPI further in WINDOW order
      const int mr = m0 + lane / CPR;
      cp = cbase + ((int64_t)p8_out_row(p, mr + k * RPI) - mr) * rstride;
    }
    if constexpr (INTERIOR) {
      epi_store16(cp, d[k], p.nt);
    } else {
      const int m = m0 + k * RPI + lane / CPR, col = n0w + (lane % CPR) * 8;
      if (m < p.M) {
        if (col + 8 <= p.N) epi_store16(cp, d[k], p.nt);
        else
          for (int e = 0; e < 8 && col + e < p.N; ++e) ((bf16_t*)cp)[e] = (bf16_t)(d[k][e >> 1] >> (16 * (e & 1)));
      }
    }
  }
  // (the reads have returned — their data was stored — before the next pass overwrites the slab)
}

}  // namespace
