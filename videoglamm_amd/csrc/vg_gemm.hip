// vg_gemm: C = ((act(A @ W^T + bias)) * gamma) + R      (see include/vg_kernels.h; DESIGN.md section 5 has the measurements)
//
// Kernels (all bf16: v_mfma_f32_32x32x16_bf16; fp32 parity mode: v_mfma_f32_32x32x2_f32, exact fp32 FMA):
//   gemm_tile_glds_kernel    128x128 tile, 4 waves, 128-byte K steps staged by LDS-DMA into swizzled unpadded rows, two stages,
//                            two workgroups per CU.  Carries the window gather/scatter (vg_gemm_window), the SwiGLU epilogue,
//                            split-K (vg_gemm_splitk) and, instantiated on bytes, the fp8 x fp8 GEMM (vg_gemm_f8).
//   gemm_tile_p8_kernel      (vg_gemm_p8.hip) 256x256 tile, 8 waves, phase-split pipeline, persistent: bf16 grids that fill the chip.  (The lock-step
//                            256x256 kernels of r01-r03 it replaced were removed in r05; DESIGN_HISTORY.md has their measurements.)
//   gemm_tile_s128_kernel    128x128 tile, ONE 128-byte-row stage, four workgroups per CU: 1024 <= K*es <= 3072 bytes.
//   gemm_tile_k64b_kernel    128x128 tile, 64-byte K steps, four workgroups per CU: K*es < 1024 bytes.
//   (r01's register-staged and 3-stage-ring A/B twins were removed at the end of r02; their measurements are in DESIGN.md section 5)
//   gemm_small64_kernel      64x64 tile, the whole K (<= 256) in one DMA burst: problems of fewer than 256 128x128 tiles (memory attention,
//                            mask decoder).
//   gemm_skinny_kernel       M <= 16 rows (LLM decode lm_head, mask-decoder token MLPs): one wave per output column streams its
//                            W row once with 16-byte loads; HBM-bound by construction.  gemm_skinny_shortk_kernel: <= 4 rows against
//                            short W rows (the mask product), a lane per output column.
//   epi_rows_fast / epi_dispatch: the straight-line row-major side shared by every LDS-staged epilogue.
// Routing: launch_gemm / route_* below (and vg_gemm_route for the bench).
#include "vg_gemm_common.h"

constexpr int GBM = 128, GBN = 128;

template <typename TO>
__device__ __forceinline__ void gemm_epilogue128(const GemmArgs& p, f32x16_t (&acc)[2][2], char* smem, int m0w, int n0w, int bz,
                                                 int wave, int lane);


// Epilogue shared by the 128x128-tile kernels: bias/activation/LayerScale, optional residual, store.
// The wave's 64x64 fp32 accumulator tile is staged RAW through LDS (the K loop is done with it) so that every lane then
// owns 8 consecutive columns of one row: residual loads and C stores become 16-byte accesses and 8 lanes write a full
// 128-byte line, instead of 2-byte stores scattered over the MFMA accumulator layout.  Bias / activation / LayerScale
// run on that row-major side (per-lane bias / gamma registers: a lane's 8 columns are the same in all 8 passes).
template <typename TO>
__device__ __forceinline__ void gemm_epilogue128(const GemmArgs& p, f32x16_t (&acc)[2][2], char* smem, int m0w, int n0w, int bz,
                                                 int wave, int lane) {
  const int M = p.M, N = p.N, l31 = lane & 31, h = lane >> 5;
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const TO* R = p.R ? (const TO*)p.R + (int64_t)bz * p.sR : nullptr;
  if (p.vec_out) {
    constexpr int ES = 68;                       // fp32 row stride (64 + 4 pad)
    float* ws = (float*)smem + wave * 64 * ES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nl = j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) ws[(i * 32 + mfma32_row(r, h)) * ES + nl] = acc[i][j][r];
      }
    __syncthreads();
    const int cg = lane & 7, rsub = lane >> 3;    // 8 column groups x 8 rows per pass
    const int n0 = n0w + cg * 8;
    if (p.a_op == 1) {
      // SwiGLU: waves (wm,0) / (wm,1) staged the gate / up halves of the same 64 outputs; each of the two finishes 32
      // of the 64 rows: y = round(silu(round(gate + b_g))) * round(up + b_u)  (HF LlamaMLP in the activation dtype,
      // the arithmetic of vg_swiglu on the rounded GEMM outputs)
      const int wm = wave >> 1, wn = wave & 1;
      const float* wg = (const float*)smem + (wm * 2) * 64 * ES;
      const float* wu = wg + 64 * ES;
      float bg[8], bu[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bg[e] = (p.bias && n0 + e < N) ? p.bias[n0 + e] : 0.f;
        bu[e] = (p.bias && n0 + e < N) ? p.bias[N + n0 + e] : 0.f;
      }
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int ml = wn * 32 + pass * 8 + rsub;
        const int m = m0w + ml;                    // m0w = bm*128 + wm*64: both waves of the pair share the row band
        if (m >= M || n0 >= N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float g = wg[ml * ES + cg * 8 + e], u = wu[ml * ES + cg * 8 + e];
          if (p.sa) {
            const int nn = min(n0 + e, N - 1);
            g *= p.sa[m] * p.sw[nn];
            u *= p.sa[m] * p.sw[N + nn];
          }
          g += bg[e];
          u += bu[e];
          if (sizeof(TO) == 2) { g = bf2f(f2bf(g)); u = bf2f(f2bf(u)); }
          g = vg_silu(g);
          if (sizeof(TO) == 2) g = bf2f(f2bf(g));
          v[e] = g * u;
        }
        TO* cp = C + (int64_t)m * p.ldc + n0;
        if (n0 + 8 <= N) {
          if constexpr (sizeof(TO) == 2) {
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf2(v[2 * e], v[2 * e + 1]);
            *(u32x4_t*)cp = o;
          } else {
            f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            *(f32x4_t*)cp = o0;
            *(f32x4_t*)(cp + 4) = o1;
          }
        } else {
          for (int e = 0; e < 8 && n0 + e < N; ++e) vg_elt<TO>::st(cp + e, v[e]);
        }
      }
      return;
    }
    float bv[8], gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bv[e] = (p.bias && n0 + e < N) ? p.bias[n0 + e] : 0.f;
      gv[e] = (p.gamma && n0 + e < N) ? p.gamma[n0 + e] : 1.f;
    }
    if (!p.sa && n0w + 64 <= N &&     // (wave-uniform) whole 16-byte groups, no fp8 scales: the straight-line form
        epi_dispatch(p.act, R != nullptr, p.gamma != nullptr, [&](auto act, auto res, auto gam) {
          epi_rows_fast<TO, decltype(act)::value, decltype(res)::value != 0, 8, ES, 8, decltype(gam)::value != 0>(p, ws, m0w, n0, cg, rsub, bv, gv, C, R);
        }))
      return;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int ml = pass * 8 + rsub;
      const int m = m0w + ml;
      if (m >= M || n0 >= N) continue;
      int64_t mo = m;
      if (p.wmode == 2) {
        mo = gemm_window_row(p, m);
        if (mo < 0) continue;
      }
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ws[ml * ES + cg * 8 + e];
      if (p.sa) {           // fp8 operands: the token and output-channel scales leave the integer-like dot products
        const float sr = p.sa[m];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= sr * p.sw[min(n0 + e, N - 1)];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = vg_act(v[e] + bv[e], p.act) * gv[e];
      TO* cp = C + mo * p.ldc + n0;
      const TO* rp = R ? R + mo * p.ldr + n0 : nullptr;
      if (n0 + 8 <= N) {
        if constexpr (sizeof(TO) == 2) {
          if (rp) {
            const u32x4_t rv = *(const u32x4_t*)rp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(rv[e] << 16); v[2 * e + 1] += __uint_as_float(rv[e] & 0xffff0000u); }
          }
          u32x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf2(v[2 * e], v[2 * e + 1]);
          *(u32x4_t*)cp = o;
        } else {
          if (rp) {
            const f32x4_t r0 = *(const f32x4_t*)rp, r1 = *(const f32x4_t*)(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          }
          f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
          *(f32x4_t*)cp = o0;
          *(f32x4_t*)(cp + 4) = o1;
        }
      } else {
        for (int e = 0; e < 8 && n0 + e < N; ++e) {
          float o = v[e];
          if (rp) o += vg_elt<TO>::ld(rp + e);
          vg_elt<TO>::st(cp + e, o);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0w + j * 32 + l31;
      if (n >= N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
      const float gv = p.gamma ? p.gamma[n] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0w + i * 32 + mfma32_row(r, h);
        if (m >= M) continue;
        float v = vg_act(acc[i][j][r] + bv, p.act) * gv;   // (window scatter takes the 16-byte path only: vg_gemm_window checks)
        if (R) v += vg_elt<TO>::ld(R + (int64_t)m * p.ldr + n);
        vg_elt<TO>::st(C + (int64_t)m * p.ldc + n, v);
      }
    }
}

// 128x128 tile with LDS-DMA staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  The DMA writes
// LDS lane-linearly (wave-uniform base + lane*16), so rows are unpadded 128-byte lines and bank conflicts are avoided
// by an XOR swizzle applied to the per-lane SOURCE chunk and again to the fragment reads (slot = chunk ^ ((row>>1)&7):
// within a ds_read_b128 lane group the 16 rows then hit 16 distinct 4-bank slots).  Out-of-range rows are clamped
// and discarded by the epilogue; a partial last K step goes through registers (issue_tail).
template <typename T, typename TO, bool FRAG_ALL = false>
__global__ __launch_bounds__(256) void gemm_tile_glds_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KPC = 16 / sizeof(T);
  constexpr int BK = 128 / sizeof(T);
  constexpr int TILEB = 128 * 128;        // bytes per operand per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  int bm, bn;
  gemm_tile_of(wgid, gridDim.y, gridDim.x, p.gn, bm, bn);
  const int bz = p.ksplit > 1 ? 0 : blockIdx.z;
  const int kz = p.ksplit > 1 ? blockIdx.z : 0, k0 = kz * p.kchunk;
  const int M = p.M, N = p.N, K = p.ksplit > 1 ? min(p.K - k0, p.kchunk) : p.K;      // K = this workgroup's slice
  const T* A = (const T*)p.A + (int64_t)bz * p.sA + k0;
  const T* W = (const T*)p.W + (int64_t)bz * p.sW + k0;

  // this lane's 4 source rows per operand (wave w stages rows [32w, 32w+32) in 4 DMA instructions of 8 rows)
  const T* asrc[4];
  const T* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gm = bm * GBM + row, gn = bn * GBN + row;
    gm = gm < M ? gm : M - 1;
    gn = gn < N ? gn : N - 1;
    if (p.wmode == 1) {
      const int64_t r = gemm_window_row(p, gm);
      asrc[i] = (r >= 0 ? A + r * p.lda : (const T*)p.zrow) + chunk * KPC;
    } else {
      asrc[i] = A + (int64_t)gm * p.lda + chunk * KPC;
    }
    if (p.a_op == 1) {
      // fused SwiGLU (M > 16): the 128 W rows of a tile are 64 gate rows followed by the 64 up rows of the SAME 64 outputs
      // (W = [gate; up], 2N rows): the interleave exists only in these row pointers, the checkpoint layout is untouched
      const int o = bn * 64 + (row & 63);
      const int oc = o < N ? o : N - 1;
      wsrc[i] = W + (int64_t)(row < 64 ? oc : N + oc) * p.ldw + chunk * KPC;
    } else {
      wsrc[i] = W + (int64_t)gn * p.ldw + chunk * KPC;
    }
  }
  auto issue = [&](int kt, int buf) {
    char* sa = smem + buf * 2 * TILEB + wave * 32 * 128;
    char* sb = sa + TILEB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (int64_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (int64_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(sb + i * 1024), 16, 0, 0);
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment row bases and swizzle keys (rows r and r+32 share ((r>>1)&7) because 32>>1 = 16 ≡ 0 mod 8)
  const int ra = wm * 64 + l31, rb = wn * 64 + l31;
  const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;
  // A partial last K step (K % BK != 0) cannot go through the DMA (no zero fill): that one step is staged through
  // registers into the same lane-linear / source-swizzled layout, with chunks past K written as zeros.
  auto issue_tail = [&](int kt, int buf) {
    char* sa = smem + buf * 2 * TILEB + wave * 32 * 128 + lane * 16;
    char* sb = sa + TILEB;
    u32x4_t va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const bool ok = kt * BK + chunk * KPC < K;
      u32x4_t z = {0u, 0u, 0u, 0u};
      va[i] = ok ? *(const u32x4_t*)(asrc[i] + (int64_t)kt * BK) : z;
      vb[i] = ok ? *(const u32x4_t*)(wsrc[i] + (int64_t)kt * BK) : z;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4_t*)(sa + i * 1024) = va[i];
      *(u32x4_t*)(sb + i * 1024) = vb[i];
    }
  };
  const int nkf = K / BK, nk = (K + BK - 1) / BK;
  if (nkf > 0) issue(0, 0);
  else issue_tail(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkf) issue(kt + 1, buf ^ 1);
    else if (kt + 1 < nk) issue_tail(kt + 1, buf ^ 1);
    const char* sa = smem + buf * 2 * TILEB + ra * 128;
    const char* sb = smem + buf * 2 * TILEB + TILEB + rb * 128;
    if constexpr (FRAG_ALL) {
      // all 16 fragment reads of the K-step are requested first; the MFMAs then start as the first ones land
      u32x4_t fa0[4], fa1[4], fb0[4], fb1[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 2 * g + h;
        fa0[g] = *(const u32x4_t*)(sa + ((c ^ swa) << 4));
        fb0[g] = *(const u32x4_t*)(sb + ((c ^ swb) << 4));
        fa1[g] = *(const u32x4_t*)(sa + 32 * 128 + ((c ^ swa) << 4));
        fb1[g] = *(const u32x4_t*)(sb + 32 * 128 + ((c ^ swb) << 4));
      }
      if constexpr (sizeof(T) == 1) {
        // fp8 (e4m3) operands: a K step is 128 elements = two v_mfma_scale_f32_32x32x64_f8f6f4 groups (unit scales); a lane
        // half holds 32 of a group's 64 K bytes — chunks 4G+h and 4G+2+h, the same pair for A and W, which is all that the
        // dot products need
#pragma unroll
        for (int G = 0; G < 2; ++G) {
          const i32x8_t a0 = f8_operand(fa0[2 * G], fa0[2 * G + 1]), a1 = f8_operand(fa1[2 * G], fa1[2 * G + 1]);
          const i32x8_t b0 = f8_operand(fb0[2 * G], fb0[2 * G + 1]), b1 = f8_operand(fb1[2 * G], fb1[2 * G + 1]);
          acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b0, acc[0][0], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          acc[0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b1, acc[0][1], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b0, acc[1][0], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          acc[1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b1, acc[1][1], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          MmaOp<T>::run(fa0[g], fb0[g], acc[0][0]);
          MmaOp<T>::run(fa0[g], fb1[g], acc[0][1]);
          MmaOp<T>::run(fa1[g], fb0[g], acc[1][0]);
          MmaOp<T>::run(fa1[g], fb1[g], acc[1][1]);
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 2 * g + h;
        u32x4_t a0 = *(const u32x4_t*)(sa + ((c ^ swa) << 4));
        u32x4_t a1 = *(const u32x4_t*)(sa + 32 * 128 + ((c ^ swa) << 4));
        u32x4_t b0 = *(const u32x4_t*)(sb + ((c ^ swb) << 4));
        u32x4_t b1 = *(const u32x4_t*)(sb + 32 * 128 + ((c ^ swb) << 4));
        MmaOp<T>::run(a0, b0, acc[0][0]);
        MmaOp<T>::run(a0, b1, acc[0][1]);
        MmaOp<T>::run(a1, b0, acc[1][0]);
        MmaOp<T>::run(a1, b1, acc[1][1]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (p.ksplit > 1) {        // raw partial tile: lanes 0..31 of an accumulator register are 32 consecutive columns of one row
    float* part = p.part + (int64_t)kz * M * N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = bn * GBN + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = bm * GBM + wm * 64 + i * 32 + mfma32_row(r, h);
          if (row < M && col < N) part[(int64_t)row * N + col] = acc[i][j][r];
        }
      }
    return;
  }
  gemm_epilogue128<TO>(p, acc, smem, bm * GBM + wm * 64, p.a_op == 1 ? bn * 64 : bn * GBN + wn * 64, bz, wave, lane);
}

// Epilogue of the four-workgroups-per-CU 128x128 kernels (64-byte-step and single-stage): the wave's 64x64 tile goes out 32 rows at a
// time through 4 x 32 x 68 floats of staging (34.8 KB with the stage buffers aliased: what keeps four workgroups on a CU).
template <typename TO>
__device__ __forceinline__ void gemm_epilogue64x32(const GemmArgs& p, f32x16_t (&acc)[2][2], char* smem, int m0w, int n0w, int bz,
                                                   int wave, int lane) {
  const int M = p.M, N = p.N, l31 = lane & 31, h = lane >> 5;
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const TO* R = p.R ? (const TO*)p.R + (int64_t)bz * p.sR : nullptr;
  constexpr int ES = 68;
  float* ws = (float*)smem + wave * 32 * ES;
  const int cg = lane & 7, rsub = lane >> 3;
  const int n0 = n0w + cg * 8;
  float bv[8], gv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bv[e] = (p.bias && n0 + e < N) ? p.bias[n0 + e] : 0.f;
    gv[e] = (p.gamma && n0 + e < N) ? p.gamma[n0 + e] : 1.f;
  }
  const bool fast = p.vec_out && !p.sa && n0w + 64 <= N;     // wave-uniform: whole 16-byte groups -> the straight-line form
  auto pass32 = [&](auto ic) {           // (instantiated by hand, like the 256x256 kernel's epilogue)
    constexpr int i = decltype(ic)::value;
    if (i) vg_lds_barrier();      // (NOT __syncthreads: its fence would wait for the previous pass's global stores to be acknowledged)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[mfma32_row(r, h) * ES + j * 32 + l31] = acc[i][j][r];
    vg_lds_barrier();
    if (fast && epi_dispatch(p.act, R != nullptr, p.gamma != nullptr, [&](auto act, auto res, auto gam) {
          epi_rows_fast<TO, decltype(act)::value, decltype(res)::value != 0, 4, ES, 8, decltype(gam)::value != 0>(p, ws, m0w + i * 32, n0, cg, rsub, bv, gv, C, R);
        }))
      return;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int ml = pass * 8 + rsub;
      const int m = m0w + i * 32 + ml;
      if (m >= M || n0 >= N) continue;
      int64_t mo = m;
      if (p.wmode == 2) {
        mo = gemm_window_row(p, m);
        if (mo < 0) continue;
      }
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = vg_act(ws[ml * ES + cg * 8 + e] + bv[e], p.act) * gv[e];
      TO* cp = C + mo * p.ldc + n0;
      const TO* rp = R ? R + mo * p.ldr + n0 : nullptr;
      if (n0 + 8 <= N && p.vec_out) {
        if constexpr (sizeof(TO) == 2) {
          if (rp) {
            const u32x4_t rv = *(const u32x4_t*)rp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(rv[e] << 16); v[2 * e + 1] += __uint_as_float(rv[e] & 0xffff0000u); }
          }
          u32x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2bf2(v[2 * e], v[2 * e + 1]);
          *(u32x4_t*)cp = o;
        } else {
          if (rp) {
            const f32x4_t r0 = *(const f32x4_t*)rp, r1 = *(const f32x4_t*)(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          }
          f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
          *(f32x4_t*)cp = o0;
          *(f32x4_t*)(cp + 4) = o1;
        }
      } else {
        for (int e = 0; e < 8 && n0 + e < N; ++e) {
          float o = v[e];
          if (rp) o += vg_elt<TO>::ld(rp + e);
          vg_elt<TO>::st(cp + e, o);
        }
      }
    }
  };
  pass32(epi_ic<0>{});
  pass32(epi_ic<1>{});
}

// 128x128 tile with 64-BYTE K steps and a 35 KB LDS footprint: FOUR workgroups (16 waves) per CU instead of two.
// The SQ counters of the 128-byte-step kernel show its waves parked on the DMA wait / barrier a third of the time with
// only two waves per SIMD to cover for each other (profiles/r01_pmc_gemm_sq_stalls.json); this variant trades half the
// MFMAs per barrier for twice the resident waves.  LDS rows are 64 B (4 chunks): slot = chunk ^ ((row >> 2) & 3) keeps
// the 16 rows of a ds_read_b128 lane group on distinct bank slots.  The epilogue stages 32 rows per wave at a time.
template <typename T, typename TO>
__global__ __launch_bounds__(256, 4) void gemm_tile_k64b_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KPC = 16 / sizeof(T);
  constexpr int BK = 64 / sizeof(T);
  constexpr int TILEB = 128 * 64;         // bytes per operand per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  int bm, bn;
  gemm_tile_of(wgid, gridDim.y, gridDim.x, p.gn, bm, bn);
  const int bz = blockIdx.z;
  const int M = p.M, N = p.N, K = p.K;
  const T* A = (const T*)p.A + (int64_t)bz * p.sA;
  const T* W = (const T*)p.W + (int64_t)bz * p.sW;

  // wave w stages rows [32w, 32w+32) of each operand in 2 DMA instructions of 16 rows (4 lanes per 64-byte row)
  const T* asrc[2];
  const T* wsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 32 + i * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    int gm = bm * GBM + row, gn = bn * GBN + row;
    gm = gm < M ? gm : M - 1;
    gn = gn < N ? gn : N - 1;
    if (p.wmode == 1) {
      const int64_t r = gemm_window_row(p, gm);
      asrc[i] = (r >= 0 ? A + r * p.lda : (const T*)p.zrow) + chunk * KPC;
    } else {
      asrc[i] = A + (int64_t)gm * p.lda + chunk * KPC;
    }
    wsrc[i] = W + (int64_t)gn * p.ldw + chunk * KPC;
  }
  auto issue = [&](int kt, int buf) {
    char* sa = smem + buf * 2 * TILEB + wave * 32 * 64;
    char* sb = sa + TILEB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (int64_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (int64_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(sb + i * 1024), 16, 0, 0);
    }
  };
  auto issue_tail = [&](int kt, int buf) {
    char* sa = smem + buf * 2 * TILEB + wave * 32 * 64 + lane * 16;
    char* sb = sa + TILEB;
    u32x4_t va[2], vb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wave * 32 + i * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const bool ok = kt * BK + chunk * KPC < K;
      u32x4_t z = {0u, 0u, 0u, 0u};
      va[i] = ok ? *(const u32x4_t*)(asrc[i] + (int64_t)kt * BK) : z;
      vb[i] = ok ? *(const u32x4_t*)(wsrc[i] + (int64_t)kt * BK) : z;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(u32x4_t*)(sa + i * 1024) = va[i];
      *(u32x4_t*)(sb + i * 1024) = vb[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int ra = wm * 64 + l31, rb = wn * 64 + l31;
  const int swa = (ra >> 2) & 3, swb = (rb >> 2) & 3;       // rows r and r+32 share the key: 32 >> 2 = 8 = 0 mod 4
  const int nkf = K / BK, nk = (K + BK - 1) / BK;
  if (nkf > 0) issue(0, 0);
  else issue_tail(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkf) issue(kt + 1, buf ^ 1);
    else if (kt + 1 < nk) issue_tail(kt + 1, buf ^ 1);
    const char* sa = smem + buf * 2 * TILEB + ra * 64;
    const char* sb = smem + buf * 2 * TILEB + TILEB + rb * 64;
    u32x4_t fa0[2], fa1[2], fb0[2], fb1[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int c = 2 * g + h;
      fa0[g] = *(const u32x4_t*)(sa + ((c ^ swa) << 4));
      fb0[g] = *(const u32x4_t*)(sb + ((c ^ swb) << 4));
      fa1[g] = *(const u32x4_t*)(sa + 32 * 64 + ((c ^ swa) << 4));
      fb1[g] = *(const u32x4_t*)(sb + 32 * 64 + ((c ^ swb) << 4));
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      MmaOp<T>::run(fa0[g], fb0[g], acc[0][0]);
      MmaOp<T>::run(fa0[g], fb1[g], acc[0][1]);
      MmaOp<T>::run(fa1[g], fb0[g], acc[1][0]);
      MmaOp<T>::run(fa1[g], fb1[g], acc[1][1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue, 32 rows of the wave's 64x64 tile at a time (4 x 32 x 68 floats = 34.8 KB of staging)
  gemm_epilogue64x32<TO>(p, acc, smem, bm * GBM + wm * 64, bn * GBN + wn * 64, bz, wave, lane);
}

// 128x128 tile, 128-BYTE K steps, ONE 32 KB stage (the epilogue staging aliases it: 35 KB in all): four workgroups per
// CU like the 64-byte-step kernel, but every DMA lane group fetches a whole 128-byte line — the CU's global->LDS path
// moves about one line per two clocks whether half of it is used or all of it (tools/dma_peak.hip).  A workgroup does
// not overlap its own loads with its MFMAs (load, barrier, 16 MFMAs per wave, barrier); the other three do.
template <typename T, typename TO>
__global__ __launch_bounds__(256, 4) void gemm_tile_s128_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KPC = 16 / sizeof(T);
  constexpr int BK = 128 / sizeof(T);
  constexpr int TILEB = 128 * 128;        // bytes per operand (single stage)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  int bm, bn;
  gemm_tile_of(wgid, gridDim.y, gridDim.x, p.gn, bm, bn);
  const int bz = blockIdx.z;
  const int M = p.M, N = p.N, K = p.K;
  const T* A = (const T*)p.A + (int64_t)bz * p.sA;
  const T* W = (const T*)p.W + (int64_t)bz * p.sW;

  // wave w stages rows [32w, 32w+32) of each operand in 4 DMA instructions of 8 rows (8 lanes per 128-byte row)
  const T* asrc[4];
  const T* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gm = bm * GBM + row, gn = bn * GBN + row;
    gm = gm < M ? gm : M - 1;
    gn = gn < N ? gn : N - 1;
    if (p.wmode == 1) {
      const int64_t r = gemm_window_row(p, gm);
      asrc[i] = (r >= 0 ? A + r * p.lda : (const T*)p.zrow) + chunk * KPC;
    } else {
      asrc[i] = A + (int64_t)gm * p.lda + chunk * KPC;
    }
    wsrc[i] = W + (int64_t)gn * p.ldw + chunk * KPC;
  }
  auto issue = [&](int kt) {
    char* sa = smem + wave * 32 * 128;
    char* sb = sa + TILEB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (int64_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (int64_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(sb + i * 1024), 16, 0, 0);
    }
  };
  auto issue_tail = [&](int kt) {
    char* sa = smem + wave * 32 * 128 + lane * 16;
    char* sb = sa + TILEB;
    u32x4_t va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const bool ok = kt * BK + chunk * KPC < K;
      u32x4_t z = {0u, 0u, 0u, 0u};
      va[i] = ok ? *(const u32x4_t*)(asrc[i] + (int64_t)kt * BK) : z;
      vb[i] = ok ? *(const u32x4_t*)(wsrc[i] + (int64_t)kt * BK) : z;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4_t*)(sa + i * 1024) = va[i];
      *(u32x4_t*)(sb + i * 1024) = vb[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int ra = wm * 64 + l31, rb = wn * 64 + l31;
  const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;       // rows r and r+32 share the key
  const int nkf = K / BK, nk = (K + BK - 1) / BK;
  const char* sa = smem + ra * 128;
  const char* sb = smem + TILEB + rb * 128;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt < nkf) issue(kt);
    else issue_tail(kt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4_t fa0[4], fa1[4], fb0[4], fb1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 2 * g + h;
      fa0[g] = *(const u32x4_t*)(sa + ((c ^ swa) << 4));
      fb0[g] = *(const u32x4_t*)(sb + ((c ^ swb) << 4));
      fa1[g] = *(const u32x4_t*)(sa + 32 * 128 + ((c ^ swa) << 4));
      fb1[g] = *(const u32x4_t*)(sb + 32 * 128 + ((c ^ swb) << 4));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      MmaOp<T>::run(fa0[g], fb0[g], acc[0][0]);
      MmaOp<T>::run(fa0[g], fb1[g], acc[0][1]);
      MmaOp<T>::run(fa1[g], fb0[g], acc[1][0]);
      MmaOp<T>::run(fa1[g], fb1[g], acc[1][1]);
    }
    __syncthreads();           // everyone has its fragments in registers: the stage may be overwritten
  }

  // epilogue, 32 rows of the wave's 64x64 tile at a time (4 x 32 x 68 floats = 34.8 KB of staging)
  gemm_epilogue64x32<TO>(p, acc, smem, bm * GBM + wm * 64, bn * GBN + wn * 64, bz, wave, lane);
}

template <typename T> __device__ __forceinline__ float dot16(const u32x4_t& a, const u32x4_t& b);
template <> __device__ __forceinline__ float dot16<float>(const u32x4_t& a, const u32x4_t& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s = fmaf(__uint_as_float(a[e]), __uint_as_float(b[e]), s);
  return s;
}
template <> __device__ __forceinline__ float dot16<bf16_t>(const u32x4_t& a, const u32x4_t& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16), s);
    s = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u), s);
  }
  return s;
}

template <typename T> __device__ __forceinline__ void unpack16(const u32x4_t& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4_t& v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(v[e]);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4_t& v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(v[e] << 16); f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
}

// Small GEMMs: 64x64 tile, the WHOLE K (64, 128, 192 or 256 bf16 elements) staged by one burst of LDS DMAs — SAM2's memory-attention and
// mask-decoder projections ([4096, 256] x [256, 256] and relatives: R/.../sam2/modeling/memory_attention.py:23-101,
// sam/transformer.py:118-193).  On the 128x128 kernels such a problem is 64 workgroups walking K in serialised 64-byte steps, one DMA
// round trip (~2 us) each: 18-25 us per launch, 840 + 256 of them per C2 clip in the video branch (r02 trace).  Here it is 256
// workgroups (the whole chip, one round), one round trip, 4-16 MFMAs per wave and the straight-line epilogue.
template <typename TO>
__global__ __launch_bounds__(256) void gemm_small64_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int KPC = 8, SEG = 64 * 128;          // one 64-element K segment of a 64-row operand tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int bn = blockIdx.x, bm = blockIdx.y, bz = blockIdx.z;
  const int M = p.M, N = p.N, nseg = p.K / 64;
  const T* A = (const T*)p.A + (int64_t)bz * p.sA;
  const T* W = (const T*)p.W + (int64_t)bz * p.sW;
  // wave w stages rows [16w, 16w+16) of both operands: per K segment 2 DMA instructions of 8 rows (8 lanes per 128-byte line) each
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 16 + i * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gm = bm * 64 + row, gn = bn * 64 + row;
    gm = gm < M ? gm : M - 1;
    gn = gn < N ? gn : N - 1;
    const T* as = A + (int64_t)gm * p.lda + chunk * KPC;
    const T* wsrc = W + (int64_t)gn * p.ldw + chunk * KPC;
    char* da = smem + wave * 16 * 128 + i * 1024;
    for (int sg = 0; sg < nseg; ++sg) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(as + sg * 64),
                                       (__attribute__((address_space(3))) void*)(da + sg * SEG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + sg * 64),
                                       (__attribute__((address_space(3))) void*)(da + (nseg + sg) * SEG), 16, 0, 0);
    }
  }
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ra = wm * 32 + l31, rb = wn * 32 + l31;
  const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int sg = 0; sg < nseg; ++sg) {
    const char* sa = smem + sg * SEG + ra * 128;
    const char* sb = smem + (nseg + sg) * SEG + rb * 128;
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
  __syncthreads();   // the fp32 staging (4 waves x 32 rows x 36 floats) aliases the operand buffers

  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const TO* R = p.R ? (const TO*)p.R + (int64_t)bz * p.sR : nullptr;
  constexpr int ES = 36;
  float* ws = (float*)smem + wave * 32 * ES;
#pragma unroll
  for (int r = 0; r < 16; ++r) ws[mfma32_row(r, h) * ES + l31] = acc[r];
  vg_lds_barrier();
  const int cg = lane & 3, rsub = lane >> 2;          // 4 column groups x 16 rows per pass
  const int n0w = bn * 64 + wn * 32, m0w = bm * 64 + wm * 32;
  const int n0 = n0w + cg * 8;
  float bv[8], gv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bv[e] = (p.bias && n0 + e < N) ? p.bias[n0 + e] : 0.f;
    gv[e] = (p.gamma && n0 + e < N) ? p.gamma[n0 + e] : 1.f;
  }
  if (n0w + 32 <= N && epi_dispatch(p.act, R != nullptr, p.gamma != nullptr, [&](auto act, auto res, auto gam) {
        epi_rows_fast<TO, decltype(act)::value, decltype(res)::value != 0, 2, ES, 16, decltype(gam)::value != 0>(p, ws, m0w, n0, cg, rsub, bv, gv, C, R);
      }))
    return;
  for (int ps = 0; ps < 2; ++ps) {                     // the last N tile of a ragged N (N % 8 == 0 still holds: vec_out)
    const int ml = ps * 16 + rsub, m = m0w + ml;
    if (m >= M || n0 >= N) continue;
    for (int e = 0; e < 8 && n0 + e < N; ++e) {
      float o = vg_act(ws[ml * ES + cg * 8 + e] + bv[e], p.act) * gv[e];
      if (R) o += vg_elt<TO>::ld(R + (int64_t)m * p.ldr + n0 + e);
      vg_elt<TO>::st(C + (int64_t)m * p.ldc + n0 + e, o);
    }
  }
}

// M <= 16 rows.  One wave per output column n streams W[n,:] once (16-byte loads, 4 independent loads in flight per
// lane) against the L1/L2-resident A rows.  a_op == 1: W is [2N, K] = gate rows | up rows and column n of the
// output is silu(A.gate_n) * (A.up_n) (HF LlamaMLP act(gate(x)) * up(x)) — the SwiGLU never touches HBM.
template <typename T, typename TO, int MT, bool GLU>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
  constexpr int KPC = 16 / sizeof(T);
  constexpr int UNR = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave, bz = blockIdx.z;
  if (n >= p.N) return;
  const T* A = (const T*)p.A + (int64_t)bz * p.sA;
  const T* Wr = (const T*)p.W + (int64_t)bz * p.sW + (int64_t)n * p.ldw;
  const T* Wu = Wr + (int64_t)p.N * p.ldw;   // GLU only
  // the epilogue's operands are requested BEFORE the K loop (r06): bias / LayerScale by every lane, the residual of row m by lane m — they used to be
  // dependent loads of lane 0 behind the reduction, two more memory round trips on a kernel whose whole time is latency (K = 256: one load per lane)
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const float bv = p.bias ? p.bias[n] : 0.f;
  const float bu = (GLU && p.bias) ? p.bias[p.N + n] : 0.f;
  const float gv = p.gamma ? p.gamma[n] : 1.f;
  const int row = lane;      // lane m finishes row m (every lane holds every row's sum after the butterfly)
  float rv = 0.f;
  if (p.R && row < p.M) rv = vg_elt<TO>::ld((const TO*)p.R + (int64_t)bz * p.sR + (int64_t)row * p.ldr + n);
  float acc[MT], accu[GLU ? MT : 1];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
#pragma unroll
  for (int m = 0; m < (GLU ? MT : 1); ++m) accu[m] = 0.f;
  const int step = 64 * KPC;
  int k = lane * KPC;
  for (; k + (UNR - 1) * step < p.K; k += UNR * step) {
    u32x4_t wv[UNR], uv[GLU ? UNR : 1];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      wv[u] = *(const u32x4_t*)(Wr + k + u * step);
      if constexpr (GLU) uv[u] = *(const u32x4_t*)(Wu + k + u * step);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < p.M) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const u32x4_t av = *(const u32x4_t*)(A + (int64_t)m * p.lda + k + u * step);
          acc[m] += dot16<T>(wv[u], av);
          if constexpr (GLU) accu[m] += dot16<T>(uv[u], av);
        }
      }
    }
  }
  for (; k < p.K; k += step) {
    const u32x4_t wv = *(const u32x4_t*)(Wr + k);
    u32x4_t uv = wv;
    if constexpr (GLU) uv = *(const u32x4_t*)(Wu + k);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < p.M) {
        const u32x4_t av = *(const u32x4_t*)(A + (int64_t)m * p.lda + k);
        acc[m] += dot16<T>(wv, av);
        if constexpr (GLU) accu[m] += dot16<T>(uv, av);
      }
    }
  }
  float mine = 0.f, mineu = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float t = wave_sum(acc[m]);
    mine = lane == m ? t : mine;
    if constexpr (GLU) {
      const float tu = wave_sum(accu[m]);
      mineu = lane == m ? tu : mineu;
    }
  }
  if (row < MT && row < p.M) {      // M stores in parallel
    float v;
    if constexpr (GLU) {
      float g = mine + bv, u = mineu + bu;
      if (sizeof(T) == 2) { g = bf2f(f2bf(g)); u = bf2f(f2bf(u)); }   // gate/up projections materialise in bf16 in HF
      g = vg_silu(g);
      if (sizeof(T) == 2) g = bf2f(f2bf(g));
      v = g * u * gv;
    } else {
      v = vg_act(mine + bv, p.act) * gv;
    }
    if (p.R) v += rv;
    vg_elt<TO>::st(C + (int64_t)row * p.ldc + n, v);
  }
}

// M <= 4 rows against SHORT rows of W (K x element size <= 256 bytes: SAM2's mask product, 4 hypernetwork rows x 32 channels against
// 65536 upscaled pixels per frame — R/.../sam2/modeling/sam/mask_decoder.py:222-234).  The wave-per-column kernel above keeps 4 of 64
// lanes busy there (998 us per C2 clip, r02 trace); here a LANE owns an output column: 64 consecutive W rows per wave are one
// contiguous block, the A rows sit in LDS and are read as broadcasts, the outputs of a wave are 64 consecutive elements per row.
template <typename T, typename TO, int MT>
__global__ __launch_bounds__(256) void gemm_skinny_shortk_kernel(GemmArgs p) {
  constexpr int KPC = 16 / sizeof(T);
  __shared__ __attribute__((aligned(16))) char as[MT * 256];
  const int bz = blockIdx.z, nch = p.K / KPC;
  const T* A = (const T*)p.A + (int64_t)bz * p.sA;
  for (int i = threadIdx.x; i < p.M * nch; i += 256) {
    const int m = i / nch, c = i - m * nch;
    *(u32x4_t*)(as + m * 256 + c * 16) = *(const u32x4_t*)(A + (int64_t)m * p.lda + c * KPC);
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= p.N) return;
  const T* Wr = (const T*)p.W + (int64_t)bz * p.sW + (int64_t)n * p.ldw;
  float acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
  for (int c = 0; c < nch; ++c) {
    const u32x4_t wv = *(const u32x4_t*)(Wr + c * KPC);
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (m < p.M) acc[m] += dot16<T>(wv, *(const u32x4_t*)(as + m * 256 + c * 16));
  }
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const TO* R = p.R ? (const TO*)p.R + (int64_t)bz * p.sR : nullptr;
  const float bv = p.bias ? p.bias[n] : 0.f;
  const float gv = p.gamma ? p.gamma[n] : 1.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (m < p.M) {
      float v = vg_act(acc[m] + bv, p.act) * gv;
      if (R) v += vg_elt<TO>::ld(R + (int64_t)m * p.ldr + n);
      vg_elt<TO>::st(C + (int64_t)m * p.ldc + n, v);
    }
  }
}

template <typename T, typename TO, bool GLU>
static void launch_skinny(const GemmArgs& p, int batch, hipStream_t st) {
  if constexpr (!GLU) {
    constexpr int KPC = 16 / sizeof(T);
    if (p.M <= 4 && (int64_t)p.K * sizeof(T) <= 256 && p.K % KPC == 0 && p.N >= 4096 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 &&
        p.lda % KPC == 0 && p.ldw % KPC == 0 && p.sA % KPC == 0 && p.sW % KPC == 0) {
      dim3 gridk((p.N + 255) / 256, 1, batch);
      gemm_skinny_shortk_kernel<T, TO, 4><<<gridk, 256, 0, st>>>(p);
      return;
    }
  }
  dim3 grid((p.N + 3) / 4, 1, batch);
  if (p.M <= 1) gemm_skinny_kernel<T, TO, 1, GLU><<<grid, 256, 0, st>>>(p);
  else if (p.M <= 4) gemm_skinny_kernel<T, TO, 4, GLU><<<grid, 256, 0, st>>>(p);
  else if (p.M <= 8) gemm_skinny_kernel<T, TO, 8, GLU><<<grid, 256, 0, st>>>(p);
  else gemm_skinny_kernel<T, TO, 16, GLU><<<grid, 256, 0, st>>>(p);
}

// ---- routing of an M > 16 GEMM to a tile kernel (shared by the launcher and vg_gemm_route, which bench.py uses to
// attribute launches to kernels).  Knobs are read once.
static int env_knob(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static int knob_rr() { static const int v = env_knob("VG_GEMM_RR", 1); return v; }    // row-register kernel (K = 144 / 288): 0 off (A/B), 1 rule (M >= 65536 rows), 2 every eligible shape (tests)
static int knob_p8() { static const int v = env_knob("VG_GEMM_P8", 1); return v; }    // 0: no 256-row tile routes (A/B), 2 / 3: the 256x256 / 256x192 kernel on every eligible bf16 shape
// small problems with a short K: fewer than 256 tiles of 128x128 (the chip is not filled), K = 64 / 128 / 192 / 256 bf16 -> gemm_small64_kernel
static bool route_small64(int64_t M, int64_t N, int64_t K, int es, int a_op, int wmode, int vec_out, int batch) {
  if (es != 2 || a_op || wmode || !vec_out || M <= 16 || K % 64 != 0 || K > 256) return false;
  return ((M + 127) / 128) * ((N + 127) / 128) * batch < 256;
}
// few K-steps per tile (K x element size <= 3 KB): the 64-byte-step kernel with four workgroups per CU
// (measured r01, tools/bench_gemm.py: +20...50 % on the Hiera / tower shapes up to K = 1408, -10...15 % from K = 2304 up)
static bool route_small_k(int64_t K, int es, int a_op) {
  return K * es <= 3072 && a_op == 0;
}
// single-stage 128-byte-row kernel: the small-K shapes with at least four whole-line K steps (measured r01: +16...18 % on
// Hiera stage 3 / 4 and +7...14 % on the tower shapes over the 64-byte-row kernel; K = 144 / 288 stay there)
static bool route_s128(int64_t K, int es, int a_op) {
  return route_small_k(K, es, a_op) && K * es >= 1024;
}
// 256x256 tile, 128x128 per wave: bf16, whole 128-byte K steps, and the 256-tiles must use the chip well: (useful
// fraction of the tiles' area) x (fill of the rounds of 256 workgroups) >= 0.7 — Hiera's N = 576 outputs or 168-tile
// grids stay on the 128x128 kernels (measured r01, tools/bench_gemm.py).  a_op == 1: a tile is 256 rows x 128 outputs
static bool route_p8(int64_t M, int64_t N, int64_t K, int es, int a_op, int wmode, int vec_out, int batch, int* ntw_out, int* mtw_out) {
  const int ntw = (int)(a_op == 1 ? (N + 127) / 128 : (N + 255) / 256), mtw = (int)((M + 255) / 256);
  if (ntw_out) { *ntw_out = ntw; *mtw_out = mtw; }
  const int w = knob_p8();
  // (r02, persistent kernel + straight-line epilogue: from K x es = 2304 B — Hiera stage 4, InternVideo2 — the 256x256 kernel beats the
  // single-stage 128x128 one by 2...9 %; K = 1024 / 576 stay there)
  constexpr int smallk_min = 1152;     // r04: with the phase-split kernel also K = 576 (Hiera stage 3): C2 268.5 -> 266.7 ms same-box
  if (!w || es != 2 || (route_small_k(K, es, a_op) && K * es < smallk_min) || wmode || K % (128 / es) != 0 || !vec_out) return false;
  if (w == 2) return true;       // 2: force the 256x256 kernel on every eligible shape (4 = the shape rule, without the 256x192 tile)
  const int64_t t256 = (int64_t)ntw * mtw * batch;
  const double useful = (double)M * N / ((double)mtw * 256 * ntw * (a_op == 1 ? 128 : 256));
  const double fill = (double)t256 / (double)(((t256 + 255) / 256) * 256);
  return t256 >= 128 && useful * fill >= 0.7;
}

// 256 x 192 tiles (vg_gemm_p8n.hip, r05) instead of 256 x 256 when whole rounds x tile cost say so: cost = rounds of 256 workgroups x
// (0.87 | 1) — measured r05 (tools/lab/p8n_ab.sh, same box): a narrow tile costs 0.83-0.90 of a wide one at equal rounds (it stages 7/8 of the
// bytes for 3/4 of the MFMAs).  N = 576 is three exact 192-tiles against 2.25 of three 256-tiles (Hiera stage 3 fc2 195 -> 162 us, proj
// 86 -> 55), Llama's q|k|v at M = 3361 two rounds of the narrow tile against two of the wide one (189 -> 170 us; r04's 128x128 route: 213).  Without a valid 256-route for the shape (under-filled or badly quantised grids) the narrow tile is judged by the
// same fill rule.
static bool route_p8n(int64_t M, int64_t N, int64_t K, int es, int a_op, int wmode, int vec_out, int batch, bool p8_ok, int* nt_out, int* mt_out) {
  const int nt = (int)((N + 191) / 192), mt = (int)((M + 255) / 256);
  if (nt_out) { *nt_out = nt; *mt_out = mt; }
  if (es != 2 || a_op || wmode || K % 64 != 0 || K < 128 || !vec_out) return false;
  if (knob_p8() == 3) return true;       // 3: force the 256x192 kernel on every eligible shape (tests: minimal K, tiny grids)
  if (knob_p8() != 1 || (route_small_k(K, es, a_op) && K * es < 1152)) return false;      // (4: the shape rule without the narrow tile — A/B)
  const int64_t t192 = (int64_t)nt * mt * batch, t256 = (int64_t)((N + 255) / 256) * mt * batch;
  if (t192 < 128) return false;
  const int64_t r192 = (t192 + 255) / 256, r256 = (t256 + 255) / 256;
  if (!p8_ok) {
    const double useful = (double)M * N / ((double)mt * 256 * nt * 192), fill = (double)t192 / (double)(r192 * 256);
    return useful * fill >= 0.7;
  }
  return (double)r192 * 0.87 < 0.97 * (double)r256;
}

template <typename T, typename TO>
static int launch_gemm(const GemmArgs& p, int batch, hipStream_t st) {
  if (p.M <= 16) {
    if (p.a_op == 1) launch_skinny<T, TO, true>(p, batch, st);
    else launch_skinny<T, TO, false>(p, batch, st);
  } else {
    // tile walk (gemm_tile_of): column groups of 4 N-tiles, i.e. a 4 x 16 patch of concurrent tiles per XCD
    // (measured r01, tools/bench_gemm.py: 4 >= 8 > 16 > row-major on every C1 shape)
    auto pick_gn = [&](int mt, int nt) { return nt < 4 ? nt : 4; };
    GemmArgs q = p;
    q.gn = pick_gn((p.M + GBM - 1) / GBM, (p.N + GBN - 1) / GBN);
    // streaming stores for big outputs (>= 128 MB, r03) with sector-aligned rows (epi_store16)
    q.nt = p.vec_out && (p.ldc * (int64_t)sizeof(TO)) % 64 == 0 && (int64_t)p.M * p.N * (int64_t)sizeof(TO) * batch >= (int64_t)128 << 20;
    dim3 grid((p.N + GBN - 1) / GBN, (p.M + GBM - 1) / GBM, batch);
    // the fp32 epilogue staging needs 4 x 64 x 68 floats = 69632 B of LDS whatever the K step
    const int lds128 = 4 * 128 * 144;
    static bool glds_attr = false;
    if (!glds_attr) {
      (void)hipFuncSetAttribute((const void*)gemm_tile_glds_kernel<T, TO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds128);
      glds_attr = true;
    }
    const bool small_k = route_small_k(p.K, (int)sizeof(T), p.a_op);
    int ntw, mtw;
    // (r04: a padding-free power-of-two window GATHER rides on the phase-split kernel — Hiera stage 3's windowed qkv; everything else windowed
    // stays on the 128x128 kernels)
    const int wroute = ((knob_p8() == 1 || knob_p8() >= 3) && sizeof(T) == 2 && vg_gemm_p8_window_ok(p.wmode, p.wsh, p.wH, p.wW, p.wws)) ? 0 : p.wmode;
    const bool big = route_p8(p.M, p.N, p.K, (int)sizeof(T), p.a_op, wroute, p.vec_out, batch, &ntw, &mtw) && vg_gemm_p8_eligible(p, batch);
    int ntn, mtn;
    const bool narrow = sizeof(T) == 2 && route_p8n(p.M, p.N, p.K, (int)sizeof(T), p.a_op, wroute, p.vec_out, batch, big, &ntn, &mtn) && vg_gemm_p8_eligible(p, batch);
    if constexpr (sizeof(T) == 2) {
      if (route_small64(p.M, p.N, p.K, 2, p.a_op, p.wmode, p.vec_out, batch) && !p.sa) {
        const int nseg = p.K / 64;
        const int lds = nseg * 2 * 64 * 128 > 4 * 32 * 36 * 4 ? nseg * 2 * 64 * 128 : 4 * 32 * 36 * 4;
        dim3 grids((p.N + 63) / 64, (p.M + 63) / 64, batch);
        static bool small_attr = false;
        if (!small_attr) {
          (void)hipFuncSetAttribute((const void*)gemm_small64_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
          small_attr = true;
        }
        gemm_small64_kernel<TO><<<grids, 256, lds, st>>>(q);
        VG_LAUNCH_CHECK();
        return VG_OK;
      }
    }
    if constexpr (sizeof(T) == 2) {
      // short K over very many rows (Hiera stages 1-2, the FPN laterals of those levels): A rows in registers, no tiles (vg_gemm_rr.hip)
      if (knob_rr() && vg_gemm_rr_eligible(p, batch, sizeof(TO) == 2) && (knob_rr() == 2 || p.M >= 65536)) {
        static const int ncu_rr = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        return vg_gemm_rr_launch(q, ncu_rr, st);
      }
    }
    if (narrow) {
      if constexpr (sizeof(T) == 2) {
        q.gn = pick_gn(mtn, ntn);
        q.nbatch = batch;
        static const int ncu = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const int64_t tot = (int64_t)ntn * mtn * batch;
        return vg_gemm_p8n_launch(q, sizeof(TO) == 2, (int)(tot > ncu ? ncu : tot), st);
      }
    }
    if (big) {
      if constexpr (sizeof(T) == 2) {
        q.gn = pick_gn(mtw, ntw);
        q.nbatch = batch;
        // persistent: one workgroup per CU walks the tile queue
        static const int ncu = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const int64_t tot = (int64_t)ntw * mtw * batch;
        return vg_gemm_p8_launch(q, sizeof(TO) == 2, (int)(tot > ncu ? ncu : tot), st);
      }
    } else if (route_s128(p.K, (int)sizeof(T), p.a_op)) {
      gemm_tile_s128_kernel<T, TO><<<grid, 256, 4 * 32 * 68 * 4, st>>>(q);
    } else if (small_k) {
      gemm_tile_k64b_kernel<T, TO><<<grid, 256, 4 * 32 * 68 * 4, st>>>(q);
    } else if (p.a_op == 1) {       // SwiGLU in the epilogue: a tile = 128 rows x 64 outputs (gate | up halves)
      dim3 gridg((p.N + 63) / 64, (p.M + GBM - 1) / GBM, batch);
      q.gn = pick_gn((p.M + GBM - 1) / GBM, (p.N + 63) / 64);
      gemm_tile_glds_kernel<T, TO, true><<<gridg, 256, lds128, st>>>(q);
    } else if (p.wmode) {
      gemm_tile_glds_kernel<T, TO, true><<<grid, 256, lds128, st>>>(q);
    } else gemm_tile_glds_kernel<T, TO, true><<<grid, 256, lds128, st>>>(q);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}


// vg_gemm_window passes its geometry to the shared body through this thread-local (the body is vg_gemm's)
struct GemmWindow { int mode, H, W, ws; const void* zrow; };
static thread_local GemmWindow g_window{0, 0, 0, 0, nullptr};
// vg_gemm_ln: the LayerNorm of the A rows rides the same way
struct GemmLn { const float* w; const float* b; float eps; bool on; };
static thread_local GemmLn g_ln{nullptr, nullptr, 0.f, false};

// ---- split-K for grids that leave most of the chip idle (few 128x128 tiles, long K): K slices on blockIdx.z, fp32 partials,
// one pass that sums them and applies the usual epilogue
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* part, int ksplit, TO* C, int64_t ldc, const float* bias,
                                                            const float* gamma, const TO* R, int64_t ldr, int M, int N, int act) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread = 8 consecutive columns of one row
  const int ng = N / 8;
  if (i >= (int64_t)M * ng) return;
  const int m = (int)(i / ng), n0 = (int)(i % ng) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < ksplit; ++z) {
    const f32x4_t* pp = (const f32x4_t*)(part + ((int64_t)z * M + m) * N + n0);
    const f32x4_t a = pp[0], b = pp[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = vg_act(v[e] + (bias ? bias[n0 + e] : 0.f), act) * (gamma ? gamma[n0 + e] : 1.f);
    if (R) v[e] += vg_elt<TO>::ld(R + (int64_t)m * ldr + n0 + e);
    vg_elt<TO>::st(C + (int64_t)m * ldc + n0 + e, v[e]);
  }
}

template <typename T, typename TO>
static int launch_gemm_splitk(GemmArgs p, hipStream_t st) {
  static bool attr = false;
  const int lds128 = 4 * 128 * 144;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tile_glds_kernel<T, TO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds128);
    attr = true;
  }
  constexpr int BK = 128 / (int)sizeof(T);
  const int nk = (p.K + BK - 1) / BK;
  p.kchunk = ((nk + p.ksplit - 1) / p.ksplit) * BK;
  p.ksplit = (p.K + p.kchunk - 1) / p.kchunk;         // no empty slices
  const int nt = (p.N + GBN - 1) / GBN, mt = (p.M + GBM - 1) / GBM;
  p.gn = nt < 4 ? nt : 4;
  if (p.ksplit > 1) {
    gemm_tile_glds_kernel<T, TO, true><<<dim3(nt, mt, p.ksplit), 256, lds128, st>>>(p);
    const int64_t groups = (int64_t)p.M * (p.N / 8);
    splitk_reduce_kernel<TO><<<(unsigned)((groups + 255) / 256), 256, 0, st>>>(p.part, p.ksplit, (TO*)p.C, p.ldc, p.bias, p.gamma, (const TO*)p.R,
                                                                             p.ldr, p.M, p.N, p.act);
  } else {
    gemm_tile_glds_kernel<T, TO, true><<<dim3(nt, mt, 1), 256, lds128, st>>>(p);
  }
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_gemm_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                              const float* gamma, const void* R, int64_t ldr, int M, int N, int K, int in_dtype, int out_dtype, int act,
                              int ksplit, float* workspace, int64_t ws_floats, vg_stream_t stream) {
  VG_CHECK(A && W && C && workspace, VG_ERR_ARG, "vg_gemm_splitk: null pointer");
  VG_CHECK(M > 16 && N > 0 && K > 0 && N % 8 == 0 && ksplit >= 2 && ksplit <= 16, VG_ERR_ARG, "vg_gemm_splitk: bad shape M=%d N=%d K=%d ksplit=%d", M, N, K, ksplit);
  VG_CHECK(in_dtype == VG_BF16 || in_dtype == VG_F32, VG_ERR_ARG, "vg_gemm_splitk: bad in_dtype %d", in_dtype);
  const int kpc = in_dtype == VG_BF16 ? 8 : 4, oes = out_dtype == VG_BF16 ? 2 : 4;
  VG_CHECK(K % kpc == 0 && lda % kpc == 0 && ldw % kpc == 0 && (((uintptr_t)A | (uintptr_t)W | (uintptr_t)workspace) & 15) == 0, VG_ERR_ARG,
           "vg_gemm_splitk: K / lda / ldw must be multiples of %d and A / W / workspace 16-byte aligned", kpc);
  VG_CHECK((ldc * oes) % 16 == 0 && ((uintptr_t)C & 15) == 0, VG_ERR_ARG, "vg_gemm_splitk: C rows must be 16-byte aligned");
  VG_CHECK(ws_floats >= (int64_t)ksplit * M * N, VG_ERR_ARG, "vg_gemm_splitk: workspace %lld < %lld floats", (long long)ws_floats, (long long)ksplit * M * N);
  GemmArgs p{};
  p.wsh = -1;
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.gamma = gamma; p.R = R;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.act = act; p.vec_out = 1; p.gn = 4;
  p.ksplit = ksplit; p.part = workspace;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == VG_BF16 && out_dtype == VG_BF16) return launch_gemm_splitk<bf16_t, bf16_t>(p, st);
  if (in_dtype == VG_BF16 && out_dtype == VG_F32) return launch_gemm_splitk<bf16_t, float>(p, st);
  if (in_dtype == VG_F32 && out_dtype == VG_F32) return launch_gemm_splitk<float, float>(p, st);
  vg_set_error("vg_gemm_splitk: unsupported dtype combination %d -> %d", in_dtype, out_dtype);
  return VG_ERR_UNSUPPORTED;
}

// ---- fp8 path (config C4's LLM prefill): per-row quantisation of the activations and the fp8 x fp8 tile GEMM
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const T* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int K) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* xr = x + (int64_t)blockIdx.x * ldx;
  float mx = 0.f;
  for (int k = tid; k < K; k += 256) mx = fmaxf(mx, fabsf(vg_elt<T>::ld(xr + k)));
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = fmaxf(mx, 1e-12f) / 448.0f;
  if (tid == 0) scale[blockIdx.x] = sc;
  uint8_t* qr = q + (int64_t)blockIdx.x * ldq;
  for (int k = tid * 4; k < K; k += 1024) {         // K % 4 == 0: four values -> one dword of e4m3 codes (RNE, as torch rounds)
    const float a = __fdiv_rn(vg_elt<T>::ld(xr + k), sc), b = __fdiv_rn(vg_elt<T>::ld(xr + k + 1), sc);
    const float c = __fdiv_rn(vg_elt<T>::ld(xr + k + 2), sc), d = __fdiv_rn(vg_elt<T>::ld(xr + k + 3), sc);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    *(int*)(qr + k) = w;
  }
}

extern "C" int vg_quantize_fp8_rows(const void* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int64_t M, int K, int dtype,
                                    vg_stream_t stream) {
  VG_CHECK(x && q && scale && M > 0 && K > 0 && K % 4 == 0 && ldq % 4 == 0, VG_ERR_ARG, "vg_quantize_fp8_rows: bad args M=%lld K=%d", (long long)M, K);
  VG_CHECK(dtype == VG_BF16 || dtype == VG_F32, VG_ERR_ARG, "vg_quantize_fp8_rows: bad dtype %d", dtype);
  VG_CHECK(M <= 0x7fffffff, VG_ERR_UNSUPPORTED, "vg_quantize_fp8_rows: too many rows");
  if (dtype == VG_BF16) quantize_fp8_rows_kernel<bf16_t><<<(unsigned)M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, q, ldq, scale, K);
  else quantize_fp8_rows_kernel<float><<<(unsigned)M, 256, 0, (hipStream_t)stream>>>((const float*)x, ldx, q, ldq, scale, K);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

template <typename TO>
static int launch_gemm_f8(const GemmArgs& p, hipStream_t st) {
  static bool attr = false;
  const int lds128 = 4 * 128 * 144;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tile_glds_kernel<uint8_t, TO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds128);
    attr = true;
  }
  GemmArgs q = p;
  const int nt = p.a_op == 1 ? (p.N + 63) / 64 : (p.N + GBN - 1) / GBN, mt = (p.M + GBM - 1) / GBM;
  q.gn = nt < 4 ? nt : 4;
  gemm_tile_glds_kernel<uint8_t, TO, true><<<dim3(nt, mt, 1), 256, lds128, st>>>(q);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_gemm_f8(const uint8_t* A8, int64_t lda, const float* a_scale, const uint8_t* W8, int64_t ldw, const float* w_scale,
                          void* C, int64_t ldc, const float* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K,
                          int out_dtype, int a_op, vg_stream_t stream) {
  VG_CHECK(A8 && W8 && C && a_scale && w_scale && M > 16 && N > 0 && K > 0, VG_ERR_ARG, "vg_gemm_f8: bad args M=%lld N=%lld K=%lld (M > 16)",
           (long long)M, (long long)N, (long long)K);
  VG_CHECK(K % 16 == 0 && lda % 16 == 0 && ldw % 16 == 0 && (((uintptr_t)A8 | (uintptr_t)W8) & 15) == 0, VG_ERR_ARG,
           "vg_gemm_f8: K / lda / ldw must be multiples of 16 and the operands 16-byte aligned");
  VG_CHECK(out_dtype == VG_BF16 || out_dtype == VG_F32, VG_ERR_ARG, "vg_gemm_f8: bad out_dtype %d", out_dtype);
  VG_CHECK(a_op == 0 || (a_op == 1 && !R), VG_ERR_ARG, "vg_gemm_f8: a_op %d (0, or 1 = SwiGLU without residual)", a_op);
  const int oes = out_dtype == VG_BF16 ? 2 : 4;
  VG_CHECK(N % 8 == 0 && (ldc * oes) % 16 == 0 && ((uintptr_t)C & 15) == 0 && (!R || ((ldr * oes) % 16 == 0 && ((uintptr_t)R & 15) == 0)),
           VG_ERR_UNSUPPORTED, "vg_gemm_f8: C / R rows must be 16-byte aligned and N a multiple of 8");
  GemmArgs p{};
  p.wsh = -1;
  p.A = A8; p.W = W8; p.C = C; p.bias = bias; p.gamma = nullptr; p.R = R;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = VG_ACT_NONE; p.vec_out = 1; p.a_op = a_op; p.gn = 4;
  p.sa = a_scale; p.sw = w_scale;
  if (out_dtype == VG_BF16) return launch_gemm_f8<bf16_t>(p, (hipStream_t)stream);
  return launch_gemm_f8<float>(p, (hipStream_t)stream);
}

extern "C" int vg_gemm_route(int64_t M, int64_t N, int64_t K, int in_dtype, int a_op, int windowed) {
  if (M <= 16) return 0;
  const int es = in_dtype == VG_BF16 ? 2 : 4;
  // 7: the row-register kernel (bf16 output, no LayerScale, activation none | GELU assumed — what the path runs at K = 144 / 288)
  if (es == 2 && !a_op && knob_rr() && (K == 144 || K == 288) && N % 16 == 0 && (knob_rr() == 2 || M >= 65536)) return 7;
  if (route_small64(M, N, K, es, a_op, windowed, 1, 1)) return 5;
  const bool p8ok = route_p8(M, N, K, es, a_op, windowed, 1, 1, nullptr, nullptr);
  if (route_p8n(M, N, K, es, a_op, windowed, 1, 1, p8ok, nullptr, nullptr)) return 6;
  if (p8ok) return 3;
  if (route_s128(K, es, a_op)) return 4;
  if (route_small_k(K, es, a_op)) return 2;
  return 1;
}

extern "C" int vg_gemm(const void* A, int64_t lda, int64_t sA, const void* W, int64_t ldw, int64_t sW,
                       void* C, int64_t ldc, int64_t sC, const float* bias, const float* gamma,
                       const void* R, int64_t ldr, int64_t sR, int M, int N, int K, int batch,
                       int in_dtype, int out_dtype, int act, int a_op, vg_stream_t stream) {
  VG_CHECK(A && W && C, VG_ERR_ARG, "vg_gemm: null pointer");
  VG_CHECK(M >= 0 && N > 0 && K > 0 && batch >= 1, VG_ERR_ARG, "vg_gemm: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  if (M == 0) return VG_OK;
  VG_CHECK(a_op == 0 || a_op == 1, VG_ERR_ARG, "vg_gemm: bad a_op %d", a_op);
  const int kpc = in_dtype == VG_BF16 ? 8 : 4;
  VG_CHECK(in_dtype == VG_BF16 || in_dtype == VG_F32, VG_ERR_ARG, "vg_gemm: bad in_dtype %d", in_dtype);
  VG_CHECK(K % kpc == 0 && lda % kpc == 0 && ldw % kpc == 0 && sA % kpc == 0 && sW % kpc == 0, VG_ERR_ARG,
           "vg_gemm: K/lda/ldw/strides must be multiples of %d (K=%d lda=%lld ldw=%lld)", kpc, K, (long long)lda, (long long)ldw);
  VG_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, VG_ERR_ARG, "vg_gemm: A/W must be 16-byte aligned");
  const int ovec = out_dtype == VG_BF16 ? 8 : 4;
  const int vec_out = (ldc % ovec == 0) && (sC % ovec == 0) && (((uintptr_t)C & 15) == 0) &&
                      (!R || ((ldr % ovec == 0) && (sR % ovec == 0) && (((uintptr_t)R & 15) == 0)));
  VG_CHECK(a_op == 0 || M <= 16 || (vec_out && !R && !gamma && batch == 1 && act == VG_ACT_NONE), VG_ERR_UNSUPPORTED,
           "vg_gemm: a_op=1 with M > 16 needs 16-byte aligned C rows and no residual / LayerScale / activation / batch");
  GemmArgs p{A, W, C, bias, gamma, R, lda, ldw, ldc, ldr, sA, sW, sC, sR, M, N, K, act, vec_out, a_op, 1, 0, 0, 0, 0, 0, 0, -1, nullptr};
  if (g_window.mode) {
    VG_CHECK(g_window.mode != 2 || vec_out, VG_ERR_UNSUPPORTED, "vg_gemm_window: scattered C/R rows must be 16-byte aligned (ldc=%lld ldr=%lld)",
             (long long)ldc, (long long)ldr);
    p.wmode = g_window.mode; p.wH = g_window.H; p.wW = g_window.W; p.wws = g_window.ws;
    p.wnH = (g_window.H + g_window.ws - 1) / g_window.ws; p.wnW = (g_window.W + g_window.ws - 1) / g_window.ws;
    p.zrow = g_window.zrow;
    auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
    const int a = lg2(p.wws), b = lg2(p.wnW), c = lg2(p.wnH);
    p.wsh = (a >= 0 && b >= 0 && c >= 0) ? (a | (b << 8) | (c << 16)) : -1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (g_ln.on) {
    p.ln_w = g_ln.w; p.ln_b = g_ln.b; p.ln_eps = g_ln.eps;
    // only the row-register kernel normalises its rows: anything else is the caller's vg_layernorm + vg_gemm
    VG_CHECK(in_dtype == VG_BF16 && out_dtype == VG_BF16 && knob_rr() && vg_gemm_rr_eligible(p, batch, true) && (knob_rr() == 2 || M >= 65536),
             VG_ERR_UNSUPPORTED, "vg_gemm_ln: M=%d N=%d K=%d is not a row-register shape (bf16, K in {144, 288}, N %% 16 == 0, >= 65536 rows, no residual)", M, N, K);
  }
  if (in_dtype == VG_BF16 && out_dtype == VG_BF16) return launch_gemm<bf16_t, bf16_t>(p, batch, st);
  if (in_dtype == VG_BF16 && out_dtype == VG_F32) return launch_gemm<bf16_t, float>(p, batch, st);
  if (in_dtype == VG_F32 && out_dtype == VG_F32) return launch_gemm<float, float>(p, batch, st);
  if (in_dtype == VG_F32 && out_dtype == VG_BF16) return launch_gemm<float, bf16_t>(p, batch, st);
  vg_set_error("vg_gemm: unsupported dtype combination %d -> %d", in_dtype, out_dtype);
  return VG_ERR_UNSUPPORTED;
}

extern "C" int vg_gemm_ln(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                          const float* ln_w, const float* ln_b, float ln_eps, int64_t M, int N, int K, int act,
                          int window, int B, int H, int Wd, int ws, const void* zero_row, int dtype, vg_stream_t stream);

extern "C" int vg_gemm_window(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                              const float* gamma, const void* R, int64_t ldr, int N, int K, int in_dtype, int out_dtype, int act,
                              int mode, int B, int H, int Wd, int ws, const void* zero_row, vg_stream_t stream) {
  VG_CHECK((mode == 1 || mode == 2) && B > 0 && H > 0 && Wd > 0 && ws > 0, VG_ERR_ARG, "vg_gemm_window: bad window mode=%d B=%d H=%d W=%d ws=%d", mode, B, H, Wd, ws);
  VG_CHECK(mode != 1 || zero_row, VG_ERR_ARG, "vg_gemm_window: mode 1 needs a row of K zeros");
  const int nH = (H + ws - 1) / ws, nW = (Wd + ws - 1) / ws;
  const int64_t M = (int64_t)B * nH * nW * ws * ws;
  VG_CHECK(M > 16 && M < (1ll << 31), VG_ERR_UNSUPPORTED, "vg_gemm_window: %lld rows", (long long)M);
  g_window = GemmWindow{mode, H, Wd, ws, zero_row};
  const int rc = vg_gemm(A, lda, 0, W, ldw, 0, C, ldc, 0, bias, gamma, R, ldr, 0, (int)M, N, K, 1, in_dtype, out_dtype, act, 0, stream);
  g_window.mode = 0;
  return rc;
}

// y = act(LayerNorm_K(A) W^T + bias): the LayerNorm in front of Hiera's q|k|v and fc1 projections at the widths of stages 1 and 2 (hieradet.py:117-123,
// 160-166), normalised in the registers of the row-register kernel (vg_gemm_rr.hip).  window = 1: the rows are gathered window by window from an image-order
// tensor (vg_gemm_window's mode 1; M is then derived from B, H, Wd, ws and padding rows are zero BEHIND the norm, like window_partition's F.pad).
extern "C" int vg_gemm_ln(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                          const float* ln_w, const float* ln_b, float ln_eps, int64_t M, int N, int K, int act,
                          int window, int B, int H, int Wd, int ws, const void* zero_row, int dtype, vg_stream_t stream) {
  VG_CHECK(ln_w && ln_b, VG_ERR_ARG, "vg_gemm_ln: LayerNorm weight / bias missing");
  VG_CHECK(dtype == VG_BF16, VG_ERR_UNSUPPORTED, "vg_gemm_ln: bf16 only (dtype %d)", dtype);
  g_ln = GemmLn{ln_w, ln_b, ln_eps, true};
  int rc;
  if (window) rc = vg_gemm_window(A, lda, W, ldw, C, ldc, bias, nullptr, nullptr, 0, N, K, dtype, dtype, act, 1, B, H, Wd, ws, zero_row, stream);
  else if (M <= 0 || M >= (1ll << 31)) { vg_set_error("vg_gemm_ln: %lld rows", (long long)M); rc = VG_ERR_ARG; }
  else rc = vg_gemm(A, lda, 0, W, ldw, 0, C, ldc, 0, bias, nullptr, nullptr, 0, 0, (int)M, N, K, 1, dtype, dtype, act, 0, stream);
  g_ln.on = false;
  return rc;
}
