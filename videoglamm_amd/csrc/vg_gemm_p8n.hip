// gemm_tile_p8n_kernel: the phase-split bf16 GEMM of vg_gemm_p8.hip on a 256 x 192 output tile (r05).
//
// Why a second tile shape: the 256 x 256 kernel pays for whole tiles and whole rounds of 256 workgroups.  N = 576 (Hiera stage 3's fc2 and proj:
// 144 + 132 launches per 32-frame clip) fills 2.25 of its 3 column tiles — a quarter of the MFMAs multiply padding —, Llama's q|k|v at
// M = 3361 is 312 tiles = 1.22 rounds (the second round runs 56 workgroups), CLIP's fc2 148 tiles = 0.58 of a round; the vendor library was
// 1.3-1.6x ahead on exactly these shapes (tools/gemm_vs_lib.py, r04).  With 192 columns N = 576 is three exact tiles, q|k|v 416 tiles of 0.75
// the work, CLIP's fc2 222 tiles in one round: launch_gemm's cost model (rounds x tile width) picks the shape per problem.
//
// Same pipeline as the 256 x 256 kernel — four phases per 64-element K step, two wave groups one barrier apart, half-tiles double-buffered
// with counted vmcnt waits, LDS-DMA through inline asm, the next tile's first loads issued in front of the epilogue, wave-private slab
// epilogue — with the operand ROLES swapped: the side that is split over the two wave groups and the two row halves (A0 / A1 there) holds
// the W rows here, 2 x (64 + 32) = 192 output columns, and the side split over the four wave columns (B0 / B1) holds the activation rows,
// 4 x (32 + 32) = 256 output rows.  A wave owns a 64-row x 96-column block: the MFMAs run as D = W . X^T (a-operand = W fragment), so a lane
// still owns ONE output row and four consecutive columns per register group.  Phases: (A0, B0) and (A0, B1) are 8 MFMAs, (A1, B1) and (A1, B0)
// 4 — 24 per K step, three quarters of the square tile's.  Staging per K step: A0 16 KB (2 DMA instructions per wave), A1 8 KB (1), B0 / B1
// 16 KB each (2): every steady-state wait is vmcnt(7).
// LDS: A0(buf) at buf * 16 KB, A1(buf) at 32 KB + buf * 8 KB, B(q, buf) at 48 KB + q * 32 KB + buf * 16 KB, slabs at 112 KB + wave * 4 KB.
#include "vg_gemm_p8_epi.h"

namespace {

constexpr int N8_A0 = 0, N8_A1 = 32 * 1024, N8_B = 48 * 1024, N8_SLAB = 112 * 1024;
constexpr int N8_TN = 192;

// bf16 output, no residual / LayerScale: packed staging; per 32-row fragment one 64-column pass (column fragments 0, 1) and one 32-column pass
template <typename TO, int ACT, bool RES, bool GAM, bool INTERIOR>      // ACT < 0: the activation code is read at run time (rare combinations)
__device__ __forceinline__ void p8n_epi_body(const GemmArgs& p, f32x16_t (&acc)[3][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  const int wr = wave >> 2, wc = wave & 3, l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const TO* R = RES ? (const TO*)p.R + (int64_t)bz * p.sR : nullptr;
  const int n0w = bn * N8_TN + wr * 96, m0w = bm * 256 + wc * 64;
  char* slab = smem + N8_SLAB + wave * 4096;
  const int wkey = (l31 >> 1) & 7;
  float bv[3][4][4], gv[4][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = n0w + i * 32 + 8 * g + 4 * h;
      if (p.bias) {
        if constexpr (INTERIOR) {
          const f32x4_t x = *(const f32x4_t*)(p.bias + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[i][g][e] = x[e];
        } else {
          p8_load4(p.bias, c0, N, 0.f, bv[i][g]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[i][g][e] = 0.f;
      }
    }
  auto value = [&](int i, int j, int g, int jj) {
    float v = vg_act(acc[i][j][4 * g + jj] + bv[i][g][jj], ACT < 0 ? p.act : ACT);
    if constexpr (GAM) v *= gv[g][jj];
    return v;
  };
  if constexpr (!RES && !GAM && sizeof(TO) == 2) {
    const int64_t rstride = (int64_t)p.ldc * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m0 = m0w + j * 32;
      {   // columns [0, 64) of the wave's block
        char* wbase = slab + l31 * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int unit = i * 8 + 2 * g + h;
            uint2 d;
            d.x = f2bf2(value(i, j, g, 0), value(i, j, g, 1));
            d.y = f2bf2(value(i, j, g, 2), value(i, j, g, 3));
            *(uint2*)(wbase + ((unit ^ (2 * wkey)) << 3)) = d;
          }
        char* cbase = (char*)(C + (int64_t)(m0 + (lane >> 3)) * p.ldc + n0w + (lane & 7) * 8);
        p8_flush_packed<64, INTERIOR>(p, slab, cbase, rstride, m0, n0w, lane);
      }
      {   // columns [64, 96)
        const int wkey4 = (l31 >> 1) & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 d;
          d.x = f2bf2(value(2, j, g, 0), value(2, j, g, 1));
          d.y = f2bf2(value(2, j, g, 2), value(2, j, g, 3));
          *(uint2*)(slab + l31 * 64 + (((2 * g + h) ^ (2 * wkey4)) << 3)) = d;
        }
        char* cbase = (char*)(C + (int64_t)(m0 + (lane >> 2)) * p.ldc + n0w + 64 + (lane & 3) * 8);
        p8_flush_packed<32, INTERIOR>(p, slab, cbase, rstride, m0, n0w + 64, lane);
      }
    }
  } else {
    // fp32 staging, one pass per (row fragment, column fragment): 32 rows x 32 fp32 columns; a lane then finishes 8 columns of one row
    const int64_t rs = RES ? p.ldr : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int m0 = m0w + j * 32, c0w = n0w + i * 32;
        if constexpr (GAM) {
#pragma unroll
          for (int g = 0; g < 4; ++g) p8_load4(p.gamma, c0w + 8 * g + 4 * h, N, 1.f, gv[g]);
        }
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
__device__ __forceinline__ void p8n_epi(const GemmArgs& p, f32x16_t (&acc)[3][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  const bool interior = bm * 256 + (wave & 3) * 64 + 64 <= p.M && bn * N8_TN + (wave >> 2) * 96 + 96 <= p.N;      // wave-uniform
  if (interior) p8n_epi_body<TO, ACT, RES, GAM, true>(p, acc, smem, bm, bn, bz, wave, lane);
  else p8n_epi_body<TO, ACT, RES, GAM, false>(p, acc, smem, bm, bn, bz, wave, lane);
}

template <typename TO>
__device__ __forceinline__ void p8n_epilogue(const GemmArgs& p, f32x16_t (&acc)[3][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  if (epi_dispatch(p.act, p.R != nullptr, p.gamma != nullptr, [&](auto act, auto res, auto gam) {
        p8n_epi<TO, decltype(act)::value, decltype(res)::value != 0, decltype(gam)::value != 0>(p, acc, smem, bm, bn, bz, wave, lane);
      }))
    return;
  if (p.R) p8n_epi<TO, -1, true, true>(p, acc, smem, bm, bn, bz, wave, lane);
  else p8n_epi<TO, -1, false, true>(p, acc, smem, bm, bn, bz, wave, lane);
}

enum { N8_FULL = 0, N8_PRELAST = 1, N8_LAST = 2 };

template <typename TO>
__global__ __launch_bounds__(512, 2) void gemm_tile_p8n_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N, K = p.K;
  const int mt = (M + 255) / 256, nt = (N + N8_TN - 1) / N8_TN;
  const int per = mt * nt, total = per * p.nbatch;
  const int xq = total >> 3, xr = total & 7;
  const int nk = K >> 6;

  // fragment read offsets inside a half-tile: local row = (wave's first row) + l31 (+ 32 for A0's second fragment: same key), k-group s
  uint32_t a0off[4], a1off[4], boff[4];
  {
    const int key = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int slot = ((2 * s + h) ^ key) << 4;
      a0off[s] = N8_A0 + (wr * 64 + l31) * 128 + slot;
      a1off[s] = N8_A1 + (wr * 32 + l31) * 128 + slot;
      boff[s] = N8_B + (wc * 32 + l31) * 128 + slot;
    }
  }
  // staging sources: A0 (W rows) two instructions, A1 one, B0 / B1 (activation rows) two each; a wave writes 8 rows x 128 bytes per instruction
  uint32_t sA0[2], sA1, sB[2][2];
  const char* Xb;
  const char* Wb;
  int bm, bn, bz;
  auto setup = [&](int lin, int lane) {       // (lane: laundered per tile by the caller — hipcc must not hoist these terms out of the tile loop)
    const int xcd = lin & 7;
    const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    bz = wgid / per;
    gemm_tile_of(wgid - bz * per, mt, nt, p.gn, bm, bn);
    Xb = (const char*)((const T*)p.A + (int64_t)bz * p.sA);
    Wb = (const char*)((const T*)p.W + (int64_t)bz * p.sW);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (i * 8 + wave) * 8 + (lane >> 3);            // local row of a 128-row half-tile
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
      int gn = bn * N8_TN + (lr >> 6) * 96 + (lr & 63);           // A0: W rows of tile columns wr * 96 + [0, 64)
      gn = gn < N ? gn : N - 1;
      sA0[i] = (uint32_t)gn * (uint32_t)(p.ldw * 2) + chunk * 16;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int gm = bm * 256 + (lr >> 5) * 64 + q * 32 + (lr & 31);  // B(q): activation rows wc * 64 + q * 32 + [0, 32)
        gm = gm < M ? gm : M - 1;
        if (p.wmode == 1) {        // vg_gemm_window's gather for power-of-two windows that tile the image: a bit-field swap (see vg_gemm_p8.hip)
          const int a = p.wsh & 0xff, nw = (p.wsh >> 8) & 0xff;
          const int rr = (gm >> a) & ((1 << a) - 1), wx = (gm >> (2 * a)) & ((1 << nw) - 1);
          gm = (gm & ~((((1 << (a + nw)) - 1)) << a)) | (wx << a) | (rr << (a + nw));
        }
        sB[q][i] = (uint32_t)gm * (uint32_t)(p.lda * 2) + chunk * 16;
      }
    }
    {
      const int lr = wave * 8 + (lane >> 3);                       // A1: 64 rows = W rows of tile columns wr * 96 + 64 + [0, 32)
      const int chunk = (lane & 7) ^ ((lr >> 1) & 7);
      int gn = bn * N8_TN + (lr >> 5) * 96 + 64 + (lr & 31);
      gn = gn < N ? gn : N - 1;
      sA1 = (uint32_t)gn * (uint32_t)(p.ldw * 2) + chunk * 16;
    }
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](uint32_t off, const char* base, uint32_t dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
  };
  auto stageA0 = [&](int buf, int kt) {
    const char* base = Wb + (int64_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(sA0[i], base, lds0 + N8_A0 + buf * 16384 + (i * 8 + wave) * 1024);
  };
  auto stageA1 = [&](int buf, int kt) { dma(sA1, Wb + (int64_t)kt * 128, lds0 + N8_A1 + buf * 8192 + wave * 1024); };
  auto stageB = [&](int q, int buf, int kt) {
    const char* base = Xb + (int64_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(sB[q][i], base, lds0 + N8_B + q * 32768 + buf * 16384 + (i * 8 + wave) * 1024);
  };

  f32x16_t acc[3][2];
  u32x4_t fa[2][4], fb[2][4];
  auto readA0 = [&](int buf) {
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int s = 0; s < 4; ++s) fa[i2][s] = *(const u32x4_t*)(smem + a0off[s] + buf * 16384 + i2 * 4096);
  };
  auto readA1 = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 4; ++s) fa[0][s] = *(const u32x4_t*)(smem + a1off[s] + buf * 8192);
  };
  auto readBinto = [&](int set, int buf, int q) {
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[set][s] = *(const u32x4_t*)(smem + boff[s] + q * 32768 + buf * 16384);
  };
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
  // D = W . X^T: the a-operand is the W fragment (its rows become the accumulator's register index = output columns), the b-operand the
  // activation fragment (its rows become the lane index = output rows)
  auto mmaA0 = [&](int qb, int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) MmaOp<T>::run(fa[i2][s], fb[set][s], acc[i2][qb]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mmaA1 = [&](int qb, int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) MmaOp<T>::run(fa[0][s], fb[set][s], acc[2][qb]);
    __builtin_amdgcn_s_setprio(0);
  };
  // One K step in buffer `buf`; on entry register set P = buf holds B0(t), the other set is free.
  //   phase 1: read A0(t)            stage B1(t+1) -> other buffer      MFMA (A0, B0)   8
  //   phase 2: read B1(t)            stage A1(t+1) -> other buffer      MFMA (A0, B1)   8
  //   phase 3: read A1(t)            stage B0(t+2) -> this buffer       MFMA (A1, B1)   4
  //   phase 4: read B0(t+1)          stage A0(t+2) -> this buffer       MFMA (A1, B0)   4
  // DMA instructions per half-tile: A0 2, A1 1, B0 2, B1 2 — four half-tiles issued after the one the next phase reads = 7 in steady state.
  auto kstep = [&](auto buf_c, int kind_in, int kt) {
    constexpr int buf = decltype(buf_c)::value;
    constexpr int P = buf, Q = buf ^ 1;
    const int kind = __builtin_amdgcn_readfirstlane(kind_in);
    readA0(buf);
    if (kind != N8_LAST) { stageB(1, buf ^ 1, kt + 1); P8_VMCNT(7); } else { P8_VMCNT(1); }
    enter_mma();
    mmaA0(0, P);
    leave_mma();
    readBinto(Q, buf, 1);
    if (kind != N8_LAST) { stageA1(buf ^ 1, kt + 1); P8_VMCNT(7); } else { P8_VMCNT(0); }
    enter_mma();
    mmaA0(1, Q);
    leave_mma();
    readA1(buf);
    if (kind == N8_FULL) { stageB(0, buf, kt + 2); P8_VMCNT(7); } else if (kind == N8_PRELAST) { P8_VMCNT(5); }
    enter_mma();
    mmaA1(1, Q);
    leave_mma();
    if (kind == N8_FULL) { readBinto(Q, buf ^ 1, 0); stageA0(buf, kt + 2); P8_VMCNT(7); }
    else if (kind == N8_PRELAST) { readBinto(Q, buf ^ 1, 0); P8_VMCNT(3); }
    enter_mma();
    mmaA1(0, P);
    leave_mma();
  };
  auto prologue = [&]() {     // the steady-state issue order: B0, A0, B1, A1 of step 0, then B0, A0 of step 1
    stageB(0, 0, 0);
    stageA0(0, 0);
    stageB(1, 0, 0);
    stageA1(0, 0);
    stageB(0, 1, 1);
    stageA0(1, 1);
  };
  int t = blockIdx.x;
  if (t >= total) return;
  int lane_t = lane;
  asm volatile("" : "+v"(lane_t));
  setup(t, lane_t);
  prologue();
  while (true) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // B0(0), A0(0) landed: at most the seven youngest operations may be pending (B1, A1 of step 0, B0, A0 of step 1 — or younger stores of the
    // previous tile's epilogue, which only make the wait more conservative)
    P8_VMCNT(7);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
    readBinto(0, 0, 0);                             // "phase 4 of step -1": B0(0)
    for (int kt = 0; kt < nk; kt += 2) {
      kstep(epi_ic<0>{}, kt + 2 < nk ? N8_FULL : (kt + 1 < nk ? N8_PRELAST : N8_LAST), kt);
      if (kt + 1 < nk) kstep(epi_ic<1>{}, kt + 3 < nk ? N8_FULL : (kt + 2 < nk ? N8_PRELAST : N8_LAST), kt + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // group 0 waits for group 1's last MFMA section: every fragment read of the tile is done
    __builtin_amdgcn_sched_barrier(0);
    const int cbm = bm, cbn = bn, cbz = bz;
    const int tn = t + gridDim.x;
    const bool more = tn < total;
    asm volatile("" : "+v"(lane_t));
    if (more) {
      setup(tn, lane_t);
      prologue();
    }
    asm volatile("" : "+v"(lane_t));
    p8n_epilogue<TO>(p, acc, smem, cbm, cbn, cbz, wave, lane_t);
    if (!more) break;
    t = tn;
    asm volatile("" : "+v"(lane_t));
    setup(t, lane_t);
  }
}

template <typename TO>
int p8n_launch(const GemmArgs& q, int wgs, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tile_p8n_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    attr = true;
  }
  gemm_tile_p8n_kernel<TO><<<wgs, 512, 144 * 1024, st>>>(q);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

}  // namespace

// (eligibility = vg_gemm_p8_eligible's without the SwiGLU form: launch_gemm checks it)
int vg_gemm_p8n_launch(const GemmArgs& q, int out_is_bf16, int wgs, hipStream_t st) {
  return out_is_bf16 ? p8n_launch<bf16_t>(q, wgs, st) : p8n_launch<float>(q, wgs, st);
}
