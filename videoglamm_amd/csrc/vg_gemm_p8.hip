// gemm_tile_p8_kernel: the 256x256-tile bf16 GEMM as a phase-split ("ping-pong") pipeline — round 4's replacement for the lock-step
// gemm_tile_w128x8_kernel on every shape that kernel took (LLM o / gate|up / down, Hiera's K >= 1152 shapes, InternVideo2).
//
// What was wrong with the lock-step form: all eight waves read fragments, multiply and wait for the next stage in the same
// rhythm — one K step drained the LDS-DMA queue (vmcnt(0)) and met at one barrier, 1.8 us per 128-byte K step against 0.85 us of MFMA time.
// Here (cdna_hip_programming.md section 5, "8-phase" structure, rebuilt on this library's 32x32x16 fragments and swizzled 128-byte rows):
//   * a K step (64 bf16 = one 128-byte line per row) is FOUR phases; in a phase a wave multiplies one 64x32 quadrant of its 128x64 output over the
//     whole K step (8 MFMAs = 256 matrix-pipe cycles) from fragments it read in the same phase;
//   * the waves of tile rows 128..255 (group 1) run ONE barrier behind those of rows 0..127 (group 0): on every SIMD one wave is in its MFMA
//     section while its partner issues its ds_reads and LDS-DMAs — matrix beside memory, never matrix beside matrix (MI355X_MICROARCH.md,
//     "Two waves per SIMD");
//   * the operands are staged as four 16 KB half-tiles per K step (A0 / A1 = the rows a wave multiplies in phases 1-2 / 3-4, B0 / B1 = the
//     W rows of quadrant columns 0 / 1), one half-tile per phase, double-buffered per half-tile: a buffer is refilled two phases after its last
//     read, so every half-tile has five to six phases (>= 1300 matrix cycles) to land and the loop never waits for vmcnt(0) — the waits are
//     counted (four half-tiles = 8 DMA instructions stay in flight across every barrier);
//   * s_setprio 1 around the MFMA sections (the matrix wave wins the issue arbitration against its partner's address arithmetic).
// A workgroup's LDS: A(q, buf) at q * 32 KB + buf * 16 KB, B(q, buf) at 64 KB + q * 32 KB + buf * 16 KB (every ds_read offset fits the 16-bit
// immediate of ONE base register per operand and k-group); rows are 128-byte lines, 16-byte chunk c of local row r sits at slot c ^ ((r >> 1) & 7)
// (the swizzle of gemm_tile_glds_kernel, applied to the per-lane SOURCE address: the DMA writes LDS lane-linearly).
// Half-tile <-> tile rows: A-half q holds tile rows {wr * 128 + q * 64 + [0, 64)}, B-half q tile columns {wc * 64 + q * 32 + [0, 32)}: a wave's
// output stays one contiguous 128 x 64 block, so the epilogue (LDS-staged, straight-line, SwiGLU by wave pairs) is the lock-step kernel's.
// Persistent: one workgroup per CU walks the tile queue in the XCD-aware order of gemm_tile_of.
#include "vg_gemm_p8_epi.h"

namespace {

constexpr int P8_HALF = 16 * 1024;

// Epilogue: wave-private LDS transposes, no workgroup barrier.  The MFMAs run as D^T = W . A^T (operands swapped), so a lane owns ONE output row
// (l31) and, per 32-column fragment, four groups g of four CONSECUTIVE columns 8 g + 4 h + {0..3}: bias / activation / LayerScale are applied in
// that layout (the lane's 32 bias values are loaded once per tile), a group is packed to 8 bytes of bf16 (16 bytes of fp32 where a residual or an
// fp32 output needs the unrounded value) and written with ONE ds_write_b64 / b128 into the wave's own 4 KB slab above the operand ring; the slab is
// read back row-major — 8 lanes x 16 bytes = a whole 128-byte line of one row — and leaves as 16-byte global stores.  Nothing is shared between
// waves, so there is no barrier, group 0 starts while group 1 still multiplies, and the next tile's first DMAs (issued before the epilogue) land
// under it.  Measured r04 on Hiera's stage-3 fc1 (M = 65536, N = 2304, K = 576) with the previous epilogue (the lock-step kernel's: fp32 tile
// through LDS by 128 ds_write_b32 per lane at the LDS's 64 B/clk store rate, sixteen workgroup barriers, row-major re-read): 8.7 us of a tile's
// 23.5 us (build without the epilogue: 212 -> 134 us), the global stores themselves 2.2 us.  A store straight from the swapped accumulators
// (v_permlane32_swap -> 16 bytes per lane, 32 bytes per row and instruction) was built and measured too: 380 us — partial-line stores are far
// worse than the LDS round trip.
constexpr int P8_SLAB = 8 * P8_HALF;      // + wave * 4096

template <typename TO, int ACT, bool RES, bool GAM, bool INTERIOR>      // ACT < 0: the activation code is read at run time (rare combinations)
__device__ __forceinline__ void p8_epi_plain_body(const GemmArgs& p, f32x16_t (&acc)[4][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const TO* R = RES ? (const TO*)p.R + (int64_t)bz * p.sR : nullptr;
  const int n0w = bn * 256 + wn * 64, m0w = bm * 256 + wm * 128;
  char* slab = smem + P8_SLAB + wave * 4096;
  const int wkey = (l31 >> 1) & 7;
  float bv[2][4][4], gv[4][4];       // (LayerScale: the current column fragment's only — 32 more registers beside the accumulators spill)
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = n0w + j * 32 + 8 * g + 4 * h;
        if constexpr (INTERIOR) {
          const f32x4_t x = *(const f32x4_t*)(p.bias + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[j][g][e] = x[e];
        } else {
          p8_load4(p.bias, c0, N, 0.f, bv[j][g]);
        }
      }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][g][e] = 0.f;
  }
  auto value = [&](int i, int j, int g, int jj) {
    float v = vg_act(acc[i][j][4 * g + jj] + bv[j][g][jj], ACT < 0 ? p.act : ACT);
    if constexpr (GAM) v *= gv[g][jj];
    return v;
  };
  if constexpr (!RES && !GAM && sizeof(TO) == 2) {
    char* wbase = slab + l31 * 128;
    char* cbase = (char*)(C + (int64_t)(m0w + (lane >> 3)) * p.ldc + n0w + (lane & 7) * 8);
    const int64_t rstride = (int64_t)p.ldc * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int unit = j * 8 + 2 * g + h;                                   // 8-byte unit of the row; the key keeps 16-byte chunks whole
          uint2 d;
          d.x = f2bf2(value(i, j, g, 0), value(i, j, g, 1));
          d.y = f2bf2(value(i, j, g, 2), value(i, j, g, 3));
          *(uint2*)(wbase + ((unit ^ (2 * wkey)) << 3)) = d;
        }
      p8_flush_packed<64, INTERIOR>(p, slab, cbase + i * 32 * rstride, rstride, m0w + i * 32, n0w, lane);
    }
  } else {
    // fp32 staging, one pass per (row fragment, column fragment): 32 rows x 32 fp32 columns; a lane then finishes 8 columns of one row
    const int64_t rs = RES ? p.ldr : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m0 = m0w + i * 32, c0w = n0w + j * 32;
        if constexpr (GAM) {
#pragma unroll
          for (int g = 0; g < 4; ++g) p8_load4(p.gamma, c0w + 8 * g + 4 * h, N, 1.f, gv[g]);
        }
        // residual pieces first (row-major side: rows k * 16 + lane / 4, columns (lane & 3) * 8)
        u32x4_t rv[2][sizeof(TO) == 2 ? 1 : 2];
        if constexpr (RES) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int m = m0 + k * 16 + (lane >> 2), col = c0w + (lane & 3) * 8;
            const u32x4_t z = {0u, 0u, 0u, 0u};
            const bool ok = INTERIOR || (m < M && col + 8 <= N);
            const int mo = p8_out_row(p, m);
#pragma unroll
            for (int w = 0; w < (sizeof(TO) == 2 ? 1 : 2); ++w) rv[k][w] = ok ? *(const u32x4_t*)((const char*)(R + (int64_t)mo * rs + col) + 16 * w) : z;
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t x = {value(i, j, g, 0), value(i, j, g, 1), value(i, j, g, 2), value(i, j, g, 3)};
          *(f32x4_t*)(slab + l31 * 128 + (((2 * g + h) ^ wkey) << 4)) = x;
        }
        p8_wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = k * 16 + (lane >> 2), c8 = lane & 3, rkey = (row >> 1) & 7;
          const f32x4_t x0 = *(const f32x4_t*)(slab + row * 128 + (((2 * c8) ^ rkey) << 4));
          const f32x4_t x1 = *(const f32x4_t*)(slab + row * 128 + (((2 * c8 + 1) ^ rkey) << 4));
          float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          const int m = m0 + row, col = c0w + c8 * 8, nvalid = INTERIOR ? 8 : N - col;
          if (!INTERIOR && (m >= M || nvalid <= 0)) continue;
          const int mo = p8_out_row(p, m);
          TO* cp = C + (int64_t)mo * p.ldc + col;
          if (nvalid >= 8) {
            if constexpr (sizeof(TO) == 2) {
              if constexpr (RES) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(rv[k][0][e] << 16); v[2 * e + 1] += __uint_as_float(rv[k][0][e] & 0xffff0000u); }
              }
              u32x4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = f2bf2(v[2 * e], v[2 * e + 1]);
              epi_store16(cp, o, p.nt);
            } else {
              if constexpr (RES) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(rv[k][0][e]); v[4 + e] += __uint_as_float(rv[k][1][e]); }
              }
              const f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
              *(f32x4_t*)cp = o0;
              *(f32x4_t*)(cp + 4) = o1;
            }
          } else {
            if constexpr (RES)
              for (int e = 0; e < nvalid; ++e) v[e] += vg_elt<TO>::ld(R + (int64_t)mo * rs + col + e);
            p8_store_tail<TO>(cp, v, nvalid);
          }
        }
        p8_wave_lds_fence();
      }
  }
}
template <typename TO, int ACT, bool RES, bool GAM>
__device__ __forceinline__ void p8_epi_plain(const GemmArgs& p, f32x16_t (&acc)[4][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  const bool interior = bm * 256 + (wave >> 2) * 128 + 128 <= p.M && bn * 256 + (wave & 3) * 64 + 64 <= p.N;      // wave-uniform
  if (interior) p8_epi_plain_body<TO, ACT, RES, GAM, true>(p, acc, smem, bm, bn, bz, wave, lane);
  else p8_epi_plain_body<TO, ACT, RES, GAM, false>(p, acc, smem, bm, bn, bz, wave, lane);
}

// SwiGLU: fragment column 0 of a wave holds the gate rows, column 1 the up rows of the SAME 32 outputs (setup's W-row map), so
// y = round(silu(round(gate + b_g))) * round(up + b_u)  (HF LlamaMLP in the activation dtype; the arithmetic of vg_swiglu) is lane-local
template <typename TO>
__device__ __forceinline__ void p8_epi_glu(const GemmArgs& p, f32x16_t (&acc)[4][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const int n0w = bn * 128 + wn * 32;
  char* slab = smem + P8_SLAB + wave * 4096;
  const bool interior = bm * 256 + wm * 128 + 128 <= M && n0w + 32 <= N;      // wave-uniform
  float bg[4][4], bu[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = n0w + 8 * g + 4 * h;
    p8_load4(p.bias, c0, N, 0.f, bg[g]);
    p8_load4(p.bias ? p.bias + N : nullptr, c0, N, 0.f, bu[g]);
  }
  auto value = [&](int i, int g, int jj) {
    float gg = acc[i][0][4 * g + jj] + bg[g][jj], uu = acc[i][1][4 * g + jj] + bu[g][jj];
    if (sizeof(TO) == 2) { gg = bf2f(f2bf(gg)); uu = bf2f(f2bf(uu)); }
    gg = vg_silu(gg);
    if (sizeof(TO) == 2) gg = bf2f(f2bf(gg));
    return gg * uu;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m0 = bm * 256 + wm * 128 + i * 32;
    if constexpr (sizeof(TO) == 2) {
      const int wkey = (l31 >> 1) & 3;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 d;
        d.x = f2bf2(value(i, g, 0), value(i, g, 1));
        d.y = f2bf2(value(i, g, 2), value(i, g, 3));
        *(uint2*)(slab + l31 * 64 + (((2 * g + h) ^ (2 * wkey)) << 3)) = d;
      }
      {
        char* cbase = (char*)((bf16_t*)C + (int64_t)(m0 + (lane >> 2)) * p.ldc + n0w + (lane & 3) * 8);
        if (interior) p8_flush_packed<32, true>(p, slab, cbase, (int64_t)p.ldc * 2, m0, n0w, lane);
        else p8_flush_packed<32, false>(p, slab, cbase, (int64_t)p.ldc * 2, m0, n0w, lane);
      }
    } else {
      // fp32 output (tests / diagnostics): 32 rows x 32 fp32 columns, as the plain form's fp32 pass
      const int wkey = (l31 >> 1) & 7;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t x = {value(i, g, 0), value(i, g, 1), value(i, g, 2), value(i, g, 3)};
        *(f32x4_t*)(slab + l31 * 128 + (((2 * g + h) ^ wkey) << 4)) = x;
      }
      p8_wave_lds_fence();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = k * 16 + (lane >> 2), c8 = lane & 3, rkey = (row >> 1) & 7;
        const f32x4_t x0 = *(const f32x4_t*)(slab + row * 128 + (((2 * c8) ^ rkey) << 4));
        const f32x4_t x1 = *(const f32x4_t*)(slab + row * 128 + (((2 * c8 + 1) ^ rkey) << 4));
        const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        const int m = m0 + row, col = n0w + c8 * 8, nvalid = N - col;
        if (m >= M || nvalid <= 0) continue;
        TO* cp = C + (int64_t)m * p.ldc + col;
        if (nvalid >= 8) {
          *(f32x4_t*)cp = x0;
          *(f32x4_t*)(cp + 4) = x1;
        } else {
          p8_store_tail<TO>(cp, v, nvalid);
        }
      }
      p8_wave_lds_fence();
    }
  }
}

template <typename TO>
__device__ __forceinline__ void p8_epilogue(const GemmArgs& p, f32x16_t (&acc)[4][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  if (p.a_op == 1) {
    p8_epi_glu<TO>(p, acc, smem, bm, bn, bz, wave, lane);
    return;
  }
  if (epi_dispatch(p.act, p.R != nullptr, p.gamma != nullptr, [&](auto act, auto res, auto gam) {
        p8_epi_plain<TO, decltype(act)::value, decltype(res)::value != 0, decltype(gam)::value != 0>(p, acc, smem, bm, bn, bz, wave, lane);
      }))
    return;
  // the combinations without a straight-line variant (SiLU / sigmoid, activation + residual, LayerScale elsewhere): run-time activation code
  if (p.R) p8_epi_plain<TO, -1, true, true>(p, acc, smem, bm, bn, bz, wave, lane);
  else p8_epi_plain<TO, -1, false, true>(p, acc, smem, bm, bn, bz, wave, lane);
}

enum { P8_FULL = 0, P8_PRELAST = 1, P8_LAST = 2 };

template <typename TO>
__global__ __launch_bounds__(512, 2) void gemm_tile_p8_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N, K = p.K;
  const int mt = (M + 255) / 256, nt = p.a_op == 1 ? (N + 127) / 128 : (N + 255) / 256;
  const int per = mt * nt, total = per * p.nbatch;
  const int xq = total >> 3, xr = total & 7;
  const int nk = K >> 6;

  // fragment read offsets: local row r = (wave's first row) + l31 (+ 32 per row fragment: the key (r >> 1) & 7 does not change), k-group s
  // (16 elements = chunks 2s, 2s + 1; lane half h takes chunk 2s + h)
  uint32_t aoff[4], boff[4];
  {
    const int key = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int slot = ((2 * s + h) ^ key) << 4;
      aoff[s] = (wr * 64 + l31) * 128 + slot;
      boff[s] = 4 * P8_HALF + (wc * 32 + l31) * 128 + slot;
    }
  }

  // staging: half-tile kind k (0 A0, 1 A1, 2 B0, 3 B1), DMA instruction i (0, 1): this wave writes local rows (i * 8 + wave) * 8 + [0, 8)
  uint32_t soff[4][2];
  const char* Ab;
  const char* Wb;
  int bm, bn, bz;
  auto setup = [&](int lin, int lane) {       // (lane: laundered per tile by the caller — hipcc must not hoist these terms out of the tile loop)
    const int xcd = lin & 7;
    const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    bz = wgid / per;
    gemm_tile_of(wgid - bz * per, mt, nt, p.gn, bm, bn);
    Ab = (const char*)((const T*)p.A + (int64_t)bz * p.sA);
    Wb = (const char*)((const T*)p.W + (int64_t)bz * p.sW);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (i * 8 + wave) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int tr = (lr >> 6) * 128 + q * 64 + (lr & 63);
        int gm = bm * 256 + tr;
        gm = gm < M ? gm : M - 1;
        // wmode 1 (r04, vg_gemm_window's gather on this kernel): GEMM row = window-order index, the source row is its image-order row — only for
        // power-of-two windows that tile the image exactly (every Hiera stage of a 1024^2 input: no padding rows, shifts and masks only)
        // — there the map is a bit-field swap: window order m = [b, wy, wx | rr | cc], image order = [b, wy, rr | wx | cc]
        if (p.wmode == 1) {
          const int a = p.wsh & 0xff, nw = (p.wsh >> 8) & 0xff;          // log2 of the window side / of the windows per image row
          const int rr = (gm >> a) & ((1 << a) - 1), wx = (gm >> (2 * a)) & ((1 << nw) - 1);
          gm = (gm & ~((((1 << (a + nw)) - 1)) << a)) | (wx << a) | (rr << (a + nw));
        }
        soff[q][i] = (uint32_t)gm * (uint32_t)(p.lda * 2) + chunk * 16;
        const int tc = (lr >> 5) * 64 + q * 32 + (lr & 31);
        int gn;
        if (p.a_op == 1) {       // fused SwiGLU: a wave's 64 tile columns = 32 gate rows then the 32 up rows of the SAME 32 outputs
          const int o = bn * 128 + (tc >> 6) * 32 + (tc & 31);
          gn = (o < N ? o : N - 1) + ((tc & 32) ? N : 0);
        } else {
          gn = bn * 256 + tc;
          gn = gn < N ? gn : N - 1;
        }
        soff[2 + q][i] = (uint32_t)gn * (uint32_t)(p.ldw * 2) + chunk * 16;
      }
    }
  };
  // LDS byte offset of half-tile (kind, buf)
  auto region = [](int kind, int buf) { return (kind >> 1) * 4 * P8_HALF + (kind & 1) * 2 * P8_HALF + buf * P8_HALF; };
  // LDS-DMA through inline asm: 32-bit lane offset + SGPR base (no 64-bit per-lane pointers for hipcc to hoist and spill), M0 written in the
  // statement that reads it; s_nop 4 covers "SALU wrote the base / M0 -> VMEM reads it" (cdna_hip_programming.md section 5.7 item 2)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto stage = [&](int kind, int buf, int kt) {
    const char* base = (kind < 2 ? Ab : Wb) + (int64_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t dst = lds0 + region(kind, buf) + (i * 8 + wave) * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(soff[kind][i]), "s"(base), "s"(dst) : "memory");
    }
  };

  f32x16_t acc[4][2];
  u32x4_t fa[2][4], fb[2][4];
  auto readA = [&](int buf, int q) {
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int s = 0; s < 4; ++s) fa[i2][s] = *(const u32x4_t*)(smem + aoff[s] + region(q, buf) + i2 * 4096);
  };
  // end of a phase's read / stage section: the barrier the partner group's MFMA section ends at, then this wave's fragments
  auto enter_mma = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto leave_mma = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // One K step in buffer `buf` (static: even steps live in buffer 0, odd steps in buffer 1).  `kind` (wave-uniform, run time): FULL stages steps
  // kt + 1 and kt + 2, PRELAST (kt == nk - 2) only kt + 1, LAST nothing; a counted wait = the number of DMA instructions issued AFTER the
  // half-tile the NEXT phase reads.
  // Balanced reads (8 / 4 / 8 / 4 ds_read_b128 per phase): the B0 fragments of K step t + 1 are read in phase 4 of step t, into the register set
  // B1 left free after phase 3 — the two B sets swap roles every K step (static, like the buffer).
  //   phase 1: read A0(t)            stage B1(t+1) -> other buffer      MFMA (A0, B0)
  //   phase 2: read B1(t)            stage A1(t+1) -> other buffer      MFMA (A0, B1)
  //   phase 3: read A1(t)            stage B0(t+2) -> this buffer       MFMA (A1, B1)
  //   phase 4: read B0(t+1)          stage A0(t+2) -> this buffer       MFMA (A1, B0)
  // Steady state: every wait is vmcnt(8) (four half-tiles issued after the one needed next).
  auto readBinto = [&](int set, int buf, int q) {
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[set][s] = *(const u32x4_t*)(smem + boff[s] + (region(2 + q, buf) - 4 * P8_HALF));
  };
  auto mma2 = [&](int qa, int qb, int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) MmaOp<T>::run(fb[set][s], fa[i2][s], acc[qa * 2 + i2][qb]);      // D^T = W . A^T: a lane owns one output row
    __builtin_amdgcn_s_setprio(0);
  };
  auto kstep = [&](auto buf_c, int kind_in, int kt) {
    constexpr int buf = decltype(buf_c)::value;
    constexpr int P = buf, Q = buf ^ 1;        // register set holding B0(t) on entry / the free one
    const int kind = __builtin_amdgcn_readfirstlane(kind_in);      // an SGPR integer: s_cmp + s_cbranch_scc, not lane-mask booleans
    readA(buf, 0);
    if (kind != P8_LAST) { stage(3, buf ^ 1, kt + 1); P8_VMCNT(8); } else { P8_VMCNT(2); }
    enter_mma();
    mma2(0, 0, P);
    leave_mma();
    readBinto(Q, buf, 1);
    if (kind != P8_LAST) { stage(1, buf ^ 1, kt + 1); P8_VMCNT(8); } else { P8_VMCNT(0); }
    enter_mma();
    mma2(0, 1, Q);
    leave_mma();
    readA(buf, 1);
    if (kind == P8_FULL) { stage(2, buf, kt + 2); P8_VMCNT(8); } else if (kind == P8_PRELAST) { P8_VMCNT(6); }
    enter_mma();
    mma2(1, 1, Q);
    leave_mma();
    if (kind == P8_FULL) { readBinto(Q, buf ^ 1, 0); stage(0, buf, kt + 2); P8_VMCNT(8); }
    else if (kind == P8_PRELAST) { readBinto(Q, buf ^ 1, 0); P8_VMCNT(4); }
    enter_mma();
    mma2(1, 0, P);
    leave_mma();
  };

  auto prologue = [&]() {     // the steady-state issue order: B0, A0, B1, A1 of step 0, then B0, A0 of step 1
    stage(2, 0, 0);
    stage(0, 0, 0);
    stage(3, 0, 0);
    stage(1, 0, 0);
    stage(2, 1, 1);
    stage(0, 1, 1);
  };
  int t = blockIdx.x;
  if (t >= total) return;
  int lane_t = lane;
  asm volatile("" : "+v"(lane_t));            // tile-invariant address terms stay inside the tile (hoisted, they are live across the K loop: spills)
  setup(t, lane_t);
  prologue();
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // B0(0), A0(0) landed: at most the eight youngest operations may be pending — the previous tile's output stores (younger than this tile's
    // prologue DMAs: they only make the counted waits more conservative) or the last four half-tiles of the prologue
    P8_VMCNT(8);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
    readBinto(0, 0, 0);                             // "phase 4 of step -1": B0(0)
    // any nk >= 2: even steps in buffer 0, odd steps in buffer 1; the kind of a step is a run-time (wave-uniform) value so that the loop is
    // two bodies and one conditional — a three-way tail of templated bodies made hipcc copy 16-register accumulator tuples between paths (500 spills)
    for (int kt = 0; kt < nk; kt += 2) {
      kstep(epi_ic<0>{}, kt + 2 < nk ? P8_FULL : (kt + 1 < nk ? P8_PRELAST : P8_LAST), kt);
      if (kt + 1 < nk) kstep(epi_ic<1>{}, kt + 3 < nk ? P8_FULL : (kt + 2 < nk ? P8_PRELAST : P8_LAST), kt + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // group 0 waits for group 1's last MFMA section: every fragment read of the tile is done
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's first six half-tiles go out BEFORE this tile's epilogue: they land under it
    const int cbm = bm, cbn = bn, cbz = bz;
    const int tn = t + gridDim.x;
    const bool more = tn < total;
    asm volatile("" : "+v"(lane_t));
    if (more) {
      setup(tn, lane_t);
      prologue();
    }
    asm volatile("" : "+v"(lane_t));
    p8_epilogue<TO>(p, acc, smem, cbm, cbn, cbz, wave, lane_t);
    if (!more) break;
    t = tn;
    // the source offsets are recomputed here instead of living across the epilogue (whose bias / LayerScale registers next to the 128 accumulator
    // registers made hipcc spill them — and reload them inside the K loop behind a vmcnt(0))
    asm volatile("" : "+v"(lane_t));
    setup(t, lane_t);
  }
}

}  // namespace

// the window forms this kernel takes: A rows gathered (mode 1) or C / R rows scattered (mode 2), windows a power of two that tile the image exactly (no padding rows: the 32-bit
// source offsets cannot reach the zero row)
bool vg_gemm_p8_window_ok(int wmode, int wsh, int wH, int wW, int wws) {
  return (wmode == 1 || wmode == 2) && wsh >= 0 && wws > 0 && wH % wws == 0 && wW % wws == 0;
}

// Launcher (vg_gemm.hip's route_w128 decides; this only checks what the 32-bit source offsets need)
bool vg_gemm_p8_eligible(const GemmArgs& p, int batch) {
  const int64_t arows = p.M, wrows = p.a_op == 1 ? 2 * (int64_t)p.N : p.N;
  const bool win_ok = p.wmode == 0 || (vg_gemm_p8_window_ok(p.wmode, p.wsh, p.wH, p.wW, p.wws) && !(p.wmode == 2 && p.a_op == 1));
  return p.K % 64 == 0 && p.K >= 128 && arows * p.lda * 2 < (int64_t)1 << 32 && wrows * p.ldw * 2 < (int64_t)1 << 32 && !p.sa && win_ok && p.vec_out;
}

template <typename TO>
static int p8_launch(const GemmArgs& q, int wgs, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tile_p8_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  gemm_tile_p8_kernel<TO><<<wgs, 512, 160 * 1024, st>>>(q);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

int vg_gemm_p8_launch(const GemmArgs& q, int out_is_bf16, int wgs, hipStream_t st) {
  return out_is_bf16 ? p8_launch<bf16_t>(q, wgs, st) : p8_launch<float>(q, wgs, st);
}
