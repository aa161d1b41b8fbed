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
#include "../vg_gemm_common.h"

namespace {

constexpr int P8_HALF = 16 * 1024;

#define P8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <typename TO>
__device__ __forceinline__ void p8_epilogue(const GemmArgs& pa, f32x16_t (&acc)[4][2], char* smem, int bm, int bn, int bz, int wave, int lane) {
  // this kernel never runs the window scatter or the fp8 scales (vg_gemm_p8_eligible): as constants they fold out of every epilogue variant
  GemmArgs p = pa;
  p.wmode = 0;
  p.sa = nullptr;
  p.sw = nullptr;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  TO* C = (TO*)p.C + (int64_t)bz * p.sC;
  const TO* R = p.R ? (const TO*)p.R + (int64_t)bz * p.sR : nullptr;
  constexpr int ES = 68;
  float* ws = (float*)smem + wave * 32 * ES;
  const int cg = lane & 7, rsub = lane >> 3;
  const int n0w = p.a_op == 1 ? bn * 128 + (wn & 1) * 64 : bn * 256 + wn * 64;
  const int n0 = n0w + cg * 8;
  const bool fast = n0w + 64 <= N;               // wave-uniform: whole 16-byte groups -> the straight-line forms
  float bv[8], gv[8], bu[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bv[e] = (p.bias && n0 + e < N) ? p.bias[n0 + e] : 0.f;
    gv[e] = (p.gamma && n0 + e < N) ? p.gamma[n0 + e] : 1.f;
    bu[e] = (p.a_op == 1 && p.bias && n0 + e < N) ? p.bias[N + n0 + e] : 0.f;
  }
  auto pass32 = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if (i) vg_lds_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[mfma32_row(r, h) * ES + j * 32 + l31] = acc[i][j][r];
    vg_lds_barrier();
    const int mrow = bm * 256 + wm * 128 + i * 32;
    if (p.a_op == 1) {
      // SwiGLU: waves (wm, c) / (wm, c + 2) staged the gate / up halves of the same 32 rows x 64 outputs; each finishes 16 rows:
      // y = round(silu(round(gate + b_g))) * round(up + b_u)  (HF LlamaMLP in the activation dtype; the arithmetic of vg_swiglu)
      const float* wg = (const float*)smem + (wm * 4 + (wn & 1)) * 32 * ES;
      const float* wu = wg + 2 * 32 * ES;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int ml = (wn >> 1) * 16 + pass * 8 + rsub;
        const int m = mrow + ml;
        if (m >= M || n0 >= N) continue;
        float gx[8], ux[8];
        if (fast) {
          const f32x4_t g0 = *(const f32x4_t*)(wg + ml * ES + cg * 8), g1 = *(const f32x4_t*)(wg + ml * ES + cg * 8 + 4);
          const f32x4_t u0 = *(const f32x4_t*)(wu + ml * ES + cg * 8), u1 = *(const f32x4_t*)(wu + ml * ES + cg * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { gx[e] = g0[e]; gx[4 + e] = g1[e]; ux[e] = u0[e]; ux[4 + e] = u1[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { gx[e] = wg[ml * ES + cg * 8 + e]; ux[e] = wu[ml * ES + cg * 8 + e]; }
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float g = gx[e] + bv[e];
          float u = ux[e] + bu[e];
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
            epi_store16(cp, o, p.nt);
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
    if (fast && epi_dispatch(p.act, R != nullptr, p.gamma != nullptr, [&](auto act, auto res, auto gam) {
          epi_rows_fast<TO, decltype(act)::value, decltype(res)::value != 0, 4, ES, 8, decltype(gam)::value != 0>(p, ws, mrow, n0, cg, rsub, bv, gv, C, R);
        }))
      return;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {     // edge tiles (N not a whole 16-byte group here) and the rarely used epilogue combinations
      const int ml = pass * 8 + rsub;
      const int m = mrow + ml;
      if (m >= M || n0 >= N) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = vg_act(ws[ml * ES + cg * 8 + e] + bv[e], p.act) * gv[e];
      TO* cp = C + (int64_t)m * p.ldc + n0;
      const TO* rp = R ? R + (int64_t)m * p.ldr + n0 : nullptr;
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
          epi_store16(cp, o, p.nt);
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
  pass32(epi_ic<2>{});
  pass32(epi_ic<3>{});
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
        soff[q][i] = (uint32_t)gm * (uint32_t)(p.lda * 2) + chunk * 16;
        const int tc = (lr >> 5) * 64 + q * 32 + (lr & 31);
        int gn;
        if (p.a_op == 1) {       // fused SwiGLU: tile columns 0..127 = gate rows, 128..255 = up rows of the SAME 128 outputs
          const int o = bn * 128 + (tc & 127);
          gn = (o < N ? o : N - 1) + (tc < 128 ? 0 : N);
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
  auto readB = [&](int buf, int q) {
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[q][s] = *(const u32x4_t*)(smem + boff[s] + (region(2 + q, buf) - 4 * P8_HALF));
  };
  auto mma = [&](int qa, int qb) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) MmaOp<T>::run(fa[i2][s], fb[qb][s], acc[qa * 2 + i2][qb]);
    __builtin_amdgcn_s_setprio(0);
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

  // one K step in buffer `buf` (tile kinds: FULL stages steps kt + 1 and kt + 2, PRELAST only kt + 1, LAST nothing; the counted waits are
  // the number of DMA instructions issued AFTER the half-tile the NEXT phase reads — see the schedule in the header)
  auto kstep = [&](auto kind_c, auto buf_c, int kt) {
    constexpr int KIND = decltype(kind_c)::value, buf = decltype(buf_c)::value;
    // phase 1: quadrant (0, 0)
    readA(buf, 0);
    readB(buf, 0);
    if constexpr (KIND != P8_LAST) { stage(3, buf ^ 1, kt + 1); P8_VMCNT(8); } else { P8_VMCNT(2); }
    enter_mma();
    mma(0, 0);
    leave_mma();
    // phase 2: quadrant (0, 1)
    readB(buf, 1);
    if constexpr (KIND != P8_LAST) { stage(1, buf ^ 1, kt + 1); P8_VMCNT(8); } else { P8_VMCNT(0); }
    enter_mma();
    mma(0, 1);
    leave_mma();
    // phase 3: quadrant (1, 1)
    readA(buf, 1);
    if constexpr (KIND == P8_FULL) stage(0, buf, kt + 2);
    enter_mma();
    mma(1, 1);
    leave_mma();
    // phase 4: quadrant (1, 0): B0 is still in registers
    if constexpr (KIND == P8_FULL) { stage(2, buf, kt + 2); P8_VMCNT(8); } else if constexpr (KIND == P8_PRELAST) { P8_VMCNT(4); }
    enter_mma();
    mma(1, 0);
    leave_mma();
  };

  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));          // tile-invariant address terms stay inside the tile (hoisted, they are live across the K loop: spills)
    setup(t, lane_t);
    stage(0, 0, 0);
    stage(2, 0, 0);
    stage(3, 0, 0);
    stage(1, 0, 0);
    stage(0, 1, 1);
    stage(2, 1, 1);
    P8_VMCNT(8);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
    // nk is even (vg_gemm_p8_eligible: K % 128 == 0): pairs of K steps in buffers 0 / 1, the last pair peeled with its shorter waits — a straight
    // loop + tail keeps the accumulators in one register web (a three-way tail made hipcc copy 16-register tuples between paths: 500 spills)
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
      kstep(epi_ic<P8_FULL>{}, epi_ic<0>{}, kt);
      kstep(epi_ic<P8_FULL>{}, epi_ic<1>{}, kt + 1);
    }
    kstep(epi_ic<P8_PRELAST>{}, epi_ic<0>{}, kt);
    kstep(epi_ic<P8_LAST>{}, epi_ic<1>{}, kt + 1);
    if (wr == 0) __builtin_amdgcn_s_barrier();      // group 0 waits for group 1's last MFMA section: the staging below overwrites the operands
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(lane_t));
    p8_epilogue<TO>(p, acc, smem, bm, bn, bz, wave, lane_t);
    vg_lds_barrier();                               // staging reads done before the next tile's DMAs land
  }
}

}  // namespace

// Launcher (vg_gemm.hip's route_w128 decides; this only checks what the 32-bit source offsets need)
bool vg_gemm_p8_eligible(const GemmArgs& p, int batch) {
  const int64_t arows = p.M, wrows = p.a_op == 1 ? 2 * (int64_t)p.N : p.N;
  return p.K % 128 == 0 && p.K >= 128 && arows * p.lda * 2 < (int64_t)1 << 32 && wrows * p.ldw * 2 < (int64_t)1 << 32 && !p.sa && !p.wmode && p.vec_out;
}

template <typename TO>
static int p8_launch(const GemmArgs& q, int wgs, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tile_p8_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * P8_HALF);
    attr = true;
  }
  gemm_tile_p8_kernel<TO><<<wgs, 512, 8 * P8_HALF, st>>>(q);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

int vg_gemm_p8_launch(const GemmArgs& q, int out_is_bf16, int wgs, hipStream_t st) {
  return out_is_bf16 ? p8_launch<bf16_t>(q, wgs, st) : p8_launch<float>(q, wgs, st);
}
