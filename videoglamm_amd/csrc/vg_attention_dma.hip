// Flash attention with LDS-DMA staging (r06).  bf16, head dim <= 128 (LLM prefill d = 128, Hiera's global blocks d = 72, the towers d = 64 / 88).
//
// attn_kernel (vg_attention.hip) stages a K / V tile through registers: every thread computes addresses for, loads, and ds_write-s its chunks of the tile —
// ~90 of the ~230 VALU instructions a wave spends per tile — and pays two barriers per tile; the counters say the loop is VALU- and barrier-bound (9.6 VALU
// instructions per MFMA, 34 % of the wave cycles waiting), not LDS-bound.  Here the tiles arrive by LDS-DMA (global_load ... lds: no staging registers, no
// address arithmetic in the loop beyond two pointer bumps, no ds_write) into a three-stage ring with ONE barrier per tile and a prefetch distance of two
// tiles; rows are 256 bytes with the 16-byte chunk index XOR-ed by a bit-swapped row index, which keeps both the ds_read_b128 K fragments (16-lane groups
// over 16 rows) and the ds_read_b64_tr_b16 V^T fragments (32-lane groups over 4 rows x 64 bytes) free of bank conflicts (SQ_LDS_BANK_CONFLICT = 0).
// QB = 32-row query blocks per wave: 1 (8 waves = two per SIMD, the hardware overlaps one wave's softmax with the other's MFMAs — the form that is routed)
// or 2 (4 waves = one per SIMD with 512 registers, every fragment feeding two MFMAs: tools/lab/attn64/README.md — at parity at best, kept for the record).
// Arithmetic = attn_kernel's: swapped product S^T = K Q^T (a lane owns one query column: row statistics are lane-local + one half-wave exchange), softmax in
// the exp2 domain with an fp32 reference that moves only when a row outgrew it by 2^64, P rounded to bf16 for the PV product, O accumulated in fp32.
#include "vg_attn_args.h"
#include <type_traits>

namespace {

__device__ __forceinline__ int adma_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int DP, bool CAUSAL, int QB, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_dma_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NG = DP / 16;       // 16-element k-groups of the head dim (one MFMA K step each)
  constexpr int NDT = DP / 32;      // 32-row output tiles along the head dim
  constexpr int BKV = 64, RW = 32 * QB, BQ = NW * RW;      // rows per wave, per workgroup
  constexpr int PPW = 16 / NW;                                // one-KiB DMA pieces of a K (or V) tile per wave
  constexpr int STAGE = 2 * BKV * 256;      // K tile + V tile, 256-byte rows; three stages
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
  int bx = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  if (p.xcd) {      // (batch, head) pairs in blocks per XCD: the query tiles of a pair walk the same K / V through ONE L2 (attn_kernel's order)
    const int nx = gridDim.x, BH = gridDim.y * gridDim.z;
    const int L = (blockIdx.z * gridDim.y + blockIdx.y) * nx + blockIdx.x, j = L >> 3, per = BH >> 3;
    // causal: the query tiles' lengths differ by up to nx x — inside an XCD the workgroups go out longest-first over ALL its heads (tile rank major), so the
    // short ones fill the tail; head-major order started a head's longest tile when the CUs were already busy (C2 prefill: 80 tile times of makespan
    // against 56 for this order, 52.5 for a perfect balance)
    const int bh = (L & 7) * per + (CAUSAL ? j % per : j / nx);
    bx = CAUSAL ? j / per : j % nx;
    head = bh % (int)gridDim.y;
    b = bh / (int)gridDim.y;
  }
  const int qtile = CAUSAL ? (int)gridDim.x - 1 - bx : bx;      // causal: the long tiles first
  const int q0 = qtile * BQ;
  const int kvh = head / (p.Hq / p.Hkv);
  const int D = p.D, Sq = p.Sq, Skv = p.Skv;
  const bf16_t* Qg = (const bf16_t*)p.Q + (int64_t)b * p.q_sb + (int64_t)head * p.q_sh;
  const bf16_t* Kg = (const bf16_t*)p.K + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh;
  const bf16_t* Vg = (const bf16_t*)p.V + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh;
  const int off = Skv - Sq;
  int kv_end = Skv;
  if (CAUSAL) {
    const int lim = q0 + BQ + off;
    kv_end = lim < Skv ? (lim > 0 ? lim : 0) : Skv;
  }
  const int ntile = (kv_end + BKV - 1) / BKV;

  // ---- K / V tile t -> ring stage t & 1: 16 + 16 one-KiB pieces, four of each per wave; lane L of a piece fills (row 4 piece + L / 16, slot L % 16),
  //      whose LOGICAL chunk is slot ^ swizzle(row).  Rows past the sequence re-read its last row (their scores are masked), chunks past the head dim
  //      re-read its last chunk (finite values against zero Q columns / unstored O columns).
  const int cmax = D / 8 - 1;
  // source addresses of this wave's pieces: computed once (row, swizzled chunk), then bumped by 64 rows per tile — only the LAST tile of a ragged sequence needs
  // the row clamp, and takes the slow form
  const bf16_t* kp[PPW];
  const bf16_t* vp[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int piece = wave * PPW + j, row = piece * 4 + (lane >> 4);
    const int c = min((lane & 15) ^ adma_sw(row), cmax);
    kp[j] = Kg + (int64_t)row * p.k_ss + c * 8;
    vp[j] = Vg + (int64_t)row * p.v_ss + c * 8;
  }
  const int64_t kstep = (int64_t)BKV * p.k_ss, vstep = (int64_t)BKV * p.v_ss;
  auto issue = [&](int t) {
    char* st = smem + (t % 3) * STAGE;
    const int kv0 = t * BKV;
    if (__builtin_expect(kv0 + BKV > Skv, 0)) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int piece = wave * PPW + j, row = piece * 4 + (lane >> 4);
        const int c = min((lane & 15) ^ adma_sw(row), cmax);
        const int key = min(kv0 + row, Skv - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(Kg + (int64_t)key * p.k_ss + c * 8), (lptr_t)(st + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(Vg + (int64_t)key * p.v_ss + c * 8), (lptr_t)(st + BKV * 256 + piece * 1024), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave * PPW + j;
      __builtin_amdgcn_global_load_lds((gptr_t)kp[j], (lptr_t)(st + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)vp[j], (lptr_t)(st + BKV * 256 + piece * 1024), 16, 0, 0);
      kp[j] += kstep;
      vp[j] += vstep;
    }
  };
  if (ntile > 0) issue(0);

  // ---- Q: the wave's 2 x 32 rows as MFMA B fragments, straight from global memory into registers (zero past the head dim / the sequence)
  u32x4_t q[QB][NG];
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qr = q0 + wave * RW + qb * 32 + l31;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c = 2 * g + h;
      q[qb][g] = (qr < Sq && c * 8 < D) ? *(const u32x4_t*)(Qg + (int64_t)qr * p.q_ss + c * 8) : zero4;
    }
  }
  f32x16_t o[QB][NDT];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][dt][r] = 0.f;
  float m_i[QB], l_i[QB];      // softmax reference (log2 units), row sum
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { m_i[qb] = -INFINITY; l_i[qb] = 0.f; }
  const float sl2 = p.scale * 1.4426950408889634f;
  const int wq0 = q0 + wave * RW;                                  // the wave's first query row
  int kofs[NG];      // byte offset of this lane's K fragment of k-group g inside a 32-key sub-tile (row l31, logical chunk 2 g + h, swizzled)
#pragma unroll
  for (int g = 0; g < NG; ++g) kofs[g] = l31 * 256 + (((2 * g + h) ^ adma_sw(l31)) << 4);
  // V^T fragment (ds_read_b64_tr_b16): lane i of a 16-lane group supplies the 8-byte piece (key row i >> 2, column quad i & 3) of a 4-key x 16-column
  // block; the A operand of 16-key step t wants keys 16 t + 4 h + {0..3} (lo) and 16 t + 8 + 4 h + {0..3} (hi) — attn_kernel's map on the swizzled rows
  int kofsB[NG];     // the same + two stages: a ds_read's immediate offset reaches 65535 bytes, so stages 0 / 1 are immediates on kofs, stage 2 on kofsB
#pragma unroll
  for (int g = 0; g < NG; ++g) kofsB[g] = kofs[g] + 2 * STAGE;
  int vofs[NDT][2], vofsB[NDT][2];
  {
    const int vrow = 4 * h + ((lane & 15) >> 2);
    const int vchunk = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1), vin = (lane & 1) * 8;
    const int vsw_lo = ((((lane & 15) >> 2) & 3) << 2) | (h & 3), vsw_hi = ((((lane & 15) >> 2) & 3) << 2) | ((h + 2) & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      vofs[dt][0] = vrow * 256 + vin + (((dt * 4 + vchunk) ^ vsw_lo) << 4);
      vofs[dt][1] = (vrow + 8) * 256 + vin + (((dt * 4 + vchunk) ^ vsw_hi) << 4);
      vofsB[dt][0] = vofs[dt][0] + 2 * STAGE;
      vofsB[dt][1] = vofs[dt][1] + 2 * STAGE;
    }
  }
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lds4_t;

  auto xor32 = [&](float x, bool mx) -> float {      // combine with the other half-wave's value (v_permlane32_swap: no LDS round trip)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float a = __uint_as_float(r[0]), b2 = __uint_as_float(r[1]);
    return mx ? fmaxf(a, b2) : a + b2;
  };
  const f32x16_t z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // inline-constant C operand: no accumulator initialisation
  // tiles 0 and 1 in flight (three-stage ring, prefetch distance two: a tile has two tiles of arithmetic to arrive)
  if (ntile > 1) issue(1);
  // one tile; the ring stage S is a compile-time constant (the loop below is unrolled by three), so every fragment address is a lane register + an immediate
  auto tile = [&](const int t, auto stage_c) {
    constexpr int S = decltype(stage_c)::value;
    if (t + 1 < ntile) {      // a wave issues 2 PPW pieces per tile: everything but tile t + 1's has landed
      if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                      // everybody's pieces of tile t are in, and everybody is done with tile t - 1's stage
    if (t + 2 < ntile) issue(t + 2);
    const int kv0 = t * BKV;
    if (CAUSAL && kv0 > wq0 + RW - 1 + off) return;         // the whole tile lies behind this wave's diagonal (the wave keeps the workgroup's barriers)
    // stages 0 / 1: smem + immediate + kofs; stage 2: smem + immediate + kofsB (= kofs + 2 STAGE)
    const char* Ks = smem + (S == 1 ? STAGE : 0);
    const char* Vs = Ks + BKV * 256;
    const int (&ko)[NG] = S == 2 ? kofsB : kofs;
    const int (&vo)[NDT][2] = S == 2 ? vofsB : vofs;
    // ---- S^T = K Q^T: one K fragment read, two MFMAs (the wave's two query blocks); fragments one step ahead
    f32x16_t sc[QB][2];
    {
      u32x4_t kf[2];
      kf[0] = *(const u32x4_t*)(Ks + ko[0]);
#pragma unroll
      for (int i = 0; i < 2 * NG; ++i) {      // i = 2 g + kt
        if (i + 1 < 2 * NG) kf[(i + 1) & 1] = *(const u32x4_t*)(Ks + ((i + 1) & 1) * 8192 + ko[(i + 1) >> 1]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          sc[qb][i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf[i & 1]), __builtin_bit_cast(bf16x8_t, q[qb][i >> 1]),
                                                                  i < 2 ? z16 : sc[qb][i & 1], 0, 0, 0);
      }
    }
    // ---- online softmax (exp2 domain), P packed to bf16 B fragments
    u32x4_t pb[QB][4];
    const bool need_mask = kv0 + BKV > Skv || (CAUSAL && kv0 + BKV - 1 > wq0 + off);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = -INFINITY;
      if (__builtin_expect(need_mask, 0)) {
        const int q_idx = wq0 + qb * 32 + l31;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + kt * 32 + mfma32_row(r, h);
            const bool ok = key < Skv && (!CAUSAL || key <= q_idx + off);
            sc[qb][kt][r] = ok ? sc[qb][kt][r] : -INFINITY;
          }
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[qb][kt][r]);
      mx = xor32(mx * sl2, true);
      // Deferred maximum: the reference m_i only moves when some row's maximum outgrew it by more than 2^64 — P <= 2^64 is as exact in fp32 / bf16 as
      // P <= 1 (same mantissa, exponent far from the range's ends), and terms that flush to zero are below 2^-62 of the row's largest.  The pass over
      // the O accumulators (VALU on MFMA registers: copies both ways) then runs in the first tile of a row at most, instead of in nearly every tile.
      m_i[qb] = (m_i[qb] == -INFINITY) ? mx : m_i[qb];      // first visible keys of the row: O and l are still zero, nothing to rescale (a select, not a branch)
      if (__builtin_expect(__any(mx > m_i[qb] + 64.0f), 0)) {
        const float m_new = fmaxf(m_i[qb], mx);
        const float alpha = exp2f(m_i[qb] - m_new);
        l_i[qb] *= alpha;
        m_i[qb] = m_new;
        // on the accumulator registers IN PLACE (inline asm with accumulator-class operands): written as "o *= alpha" the compiler keeps a VGPR copy of
        // all of O alive across the loop for this cold block — 96-128 v_accvgpr_read per tile in the hot path
        if constexpr (QB == 2) {
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float tmp;
              asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1" : "+a"(o[qb][dt][r]), "=&v"(tmp) : "v"(alpha));
            }
          asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // accumulator writes -> the next MFMA that reads them
        } else {      // two waves per SIMD: <= 256 registers, the MFMAs take the VGPR form and O is an ordinary operand
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][dt][r] *= alpha;
        }
      }
      const float nm = (m_i[qb] == -INFINITY) ? 0.f : -m_i[qb];
      vg_f32x2_t rs2 = {0.f, 0.f};      // (pairs: v_pk_add_f32 halves the row-sum instructions)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sc[qb][kt][r], sl2, nm));      // (-inf scores: exp2(-inf) = 0)
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sc[qb][kt][r + 1], sl2, nm));
          sc[qb][kt][r] = p0;
          sc[qb][kt][r + 1] = p1;
          rs2 += vg_f32x2_t{p0, p1};
        }
      l_i[qb] += xor32(rs2[0] + rs2[1], false);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) pb[qb][tt][jj] = f2bf2(sc[qb][tt >> 1][(tt & 1) * 8 + jj * 2], sc[qb][tt >> 1][(tt & 1) * 8 + jj * 2 + 1]);
    }
    // ---- O^T += V^T P^T: one V^T fragment (two transposing reads), two MFMAs; fragments one step ahead
    {
      auto vread = [&](int tt, int dt) -> u32x4_t {
        const char* a = Vs + tt * 16 * 256;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + vo[dt][0]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + vo[dt][1]));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        const u32x4_t v = {l2.x, l2.y, h2.x, h2.y};
        return v;
      };
      u32x4_t vf[2];
      vf[0] = vread(0, 0);
#pragma unroll
      for (int i = 0; i < 4 * NDT; ++i) {      // i = tt NDT + dt
        if (i + 1 < 4 * NDT) vf[(i + 1) & 1] = vread((i + 1) / NDT, (i + 1) % NDT);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          o[qb][i % NDT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf[i & 1]), __builtin_bit_cast(bf16x8_t, pb[qb][i / NDT]), o[qb][i % NDT], 0, 0, 0);
      }
    }
  };
  {
    int t = 0;
    for (; t + 3 <= ntile; t += 3) {
      tile(t, std::integral_constant<int, 0>{});
      tile(t + 1, std::integral_constant<int, 1>{});
      tile(t + 2, std::integral_constant<int, 2>{});
    }
    if (t < ntile) tile(t, std::integral_constant<int, 0>{});
    if (t + 1 < ntile) tile(t + 1, std::integral_constant<int, 1>{});
  }
  // ---- epilogue: a lane holds 4 consecutive head-dim values per register quad of its query row: 8-byte stores
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qr = wq0 + qb * 32 + l31;
    if (qr >= Sq) continue;
    const float inv = l_i[qb] > 0.f ? 1.0f / l_i[qb] : 0.f;
    bf16_t* Og = (bf16_t*)p.O + (int64_t)b * p.o_sb + (int64_t)head * p.o_sh + (int64_t)qr * p.o_ss;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        if (d0 < D) {
          uint2 v;
          v.x = f2bf2(o[qb][dt][4 * g] * inv, o[qb][dt][4 * g + 1] * inv);
          v.y = f2bf2(o[qb][dt][4 * g + 2] * inv, o[qb][dt][4 * g + 3] * inv);
          *(uint2*)(Og + d0) = v;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same loop for SAM2's memory cross-attention (vg_attention_dv): head dim 256 keys, 64-wide values, no mask, split-KV partials for attn_combine_kernel.
// K rows are 512 bytes (two 256-byte bank lines: the swizzle acts on the 16-byte slot inside a line), V rows 128 bytes kept in 256-byte LDS rows (the V^T reads
// are then exactly the d <= 128 kernel's); a stage = K tile 32 KB + V tile 16 KB, three stages = 144 KB (one 8-wave workgroup per CU, two waves per SIMD);
// a wave issues 4 + 2 one-KiB pieces per tile.  r05's kernel for this shape splits every 64-key tile over TWO waves of the same 32 query rows (8 waves on
// 128 rows) and stages through registers; here a wave owns its 32 rows for all 64 keys and 8 waves share a tile over 256 rows.
__global__ __launch_bounds__(512, 1) void attn_dma_d256v64_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NG = 16, NDT = 2, BKV = 64, BQ = 256;
  constexpr int KT = BKV * 512, STAGE = KT + BKV * 256;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lds4_t;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
  const int split = blockIdx.x % p.nsplit, qtile = blockIdx.x / p.nsplit, head = blockIdx.y, b = blockIdx.z;
  const int q0 = qtile * BQ, Sq = p.Sq, Skv = p.Skv;
  const bf16_t* Qg = (const bf16_t*)p.Q + (int64_t)b * p.q_sb + (int64_t)head * p.q_sh;
  const bf16_t* Kg = (const bf16_t*)p.K + (int64_t)b * p.k_sb + (int64_t)head * p.k_sh;
  const bf16_t* Vg = (const bf16_t*)p.V + (int64_t)b * p.v_sb + (int64_t)head * p.v_sh;
  int kv_begin = 0, kv_end = Skv;
  if (p.nsplit > 1) {
    kv_begin = split * p.split_len;                   // (split_len is a whole number of 64-key tiles)
    kv_end = min(kv_begin + p.split_len, Skv);
  }
  const int t0 = kv_begin / BKV, ntile = kv_end > kv_begin ? (kv_end - kv_begin + BKV - 1) / BKV : 0;

  // K pieces: 32 per tile, 4 per wave, lane L -> (row 2 piece + L / 32, line (L / 16) & 1, slot L % 16); V pieces: 16 per tile, 2 per wave, as in attn_dma_kernel
  auto issue = [&](int t) {      // t: tile index inside this split's range
    char* st = smem + (t % 3) * STAGE;
    const int kv0 = (t0 + t) * BKV;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j, row = piece * 2 + (lane >> 5);
      const int c = ((lane >> 4) & 1) * 16 + ((lane & 15) ^ adma_sw(row));
      const int key = min(kv0 + row, Skv - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Kg + (int64_t)key * p.k_ss + c * 8), (lptr_t)(st + piece * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = wave * 2 + j, row = piece * 4 + (lane >> 4);
      const int c = min((lane & 15) ^ adma_sw(row), 7);      // 64-wide values: chunks 0..7 (the slots of chunks 8..15 re-read chunk 7: never used)
      const int key = min(kv0 + row, Skv - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Vg + (int64_t)key * p.v_ss + c * 8), (lptr_t)(st + KT + piece * 1024), 16, 0, 0);
    }
  };
  if (ntile > 0) issue(0);
  u32x4_t q[NG];
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  const int qr = q0 + wave * 32 + l31;
#pragma unroll
  for (int g = 0; g < NG; ++g) q[g] = qr < Sq ? *(const u32x4_t*)(Qg + (int64_t)qr * p.q_ss + (2 * g + h) * 8) : zero4;
  f32x16_t o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_i = -INFINITY, l_i = 0.f;
  const float sl2 = p.scale * 1.4426950408889634f;
  int kofs[8];      // k-groups g and g + 8 sit 256 bytes apart (the row's second bank line): an immediate
#pragma unroll
  for (int g = 0; g < 8; ++g) kofs[g] = l31 * 512 + (((2 * g + h) ^ adma_sw(l31)) << 4);
  int vofs[NDT][2];
  {
    const int vrow = 4 * h + ((lane & 15) >> 2);
    const int vchunk = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1), vin = (lane & 1) * 8;
    const int vsw_lo = ((((lane & 15) >> 2) & 3) << 2) | (h & 3), vsw_hi = ((((lane & 15) >> 2) & 3) << 2) | ((h + 2) & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      vofs[dt][0] = vrow * 256 + vin + (((dt * 4 + vchunk) ^ vsw_lo) << 4);
      vofs[dt][1] = (vrow + 8) * 256 + vin + (((dt * 4 + vchunk) ^ vsw_hi) << 4);
    }
  }
  auto xor32 = [&](float x, bool mx) -> float {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float a = __uint_as_float(r[0]), b2 = __uint_as_float(r[1]);
    return mx ? fmaxf(a, b2) : a + b2;
  };
  const f32x16_t z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ntile > 1) issue(1);
  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // a wave issues 6 pieces per tile: everything but tile t + 1's has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 2 < ntile) issue(t + 2);
    const int kv0 = (t0 + t) * BKV;
    const char* Ks = smem + (t % 3) * STAGE;
    const char* Vs = Ks + KT;
    f32x16_t sc[2];
    {
      u32x4_t kf[2];
      kf[0] = *(const u32x4_t*)(Ks + kofs[0]);
#pragma unroll
      for (int i = 0; i < 2 * NG; ++i) {      // i = 2 g + kt
        if (i + 1 < 2 * NG) {
          const int g1 = (i + 1) >> 1;
          kf[(i + 1) & 1] = *(const u32x4_t*)(Ks + ((i + 1) & 1) * 32 * 512 + (g1 >> 3) * 256 + kofs[g1 & 7]);
        }
        sc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf[i & 1]), __builtin_bit_cast(bf16x8_t, q[i >> 1]), i < 2 ? z16 : sc[i & 1], 0, 0, 0);
      }
    }
    if (__builtin_expect(kv0 + BKV > kv_end, 0)) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kt][r] = kv0 + kt * 32 + mfma32_row(r, h) < kv_end ? sc[kt][r] : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
    mx = xor32(mx * sl2, true);
    m_i = (m_i == -INFINITY) ? mx : m_i;      // (deferred reference, as in attn_dma_kernel)
    if (__builtin_expect(__any(mx > m_i + 64.0f), 0)) {
      const float m_new = fmaxf(m_i, mx);
      const float alpha = exp2f(m_i - m_new);
      l_i *= alpha;
      m_i = m_new;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float nm = (m_i == -INFINITY) ? 0.f : -m_i;
    vg_f32x2_t rs2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], sl2, nm));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(sc[kt][r + 1], sl2, nm));
        sc[kt][r] = p0;
        sc[kt][r + 1] = p1;
        rs2 += vg_f32x2_t{p0, p1};
      }
    l_i += xor32(rs2[0] + rs2[1], false);
    u32x4_t pb[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) pb[tt][jj] = f2bf2(sc[tt >> 1][(tt & 1) * 8 + jj * 2], sc[tt >> 1][(tt & 1) * 8 + jj * 2 + 1]);
    {
      auto vread = [&](int tt, int dt) -> u32x4_t {
        const char* a = Vs + tt * 16 * 256;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + vofs[dt][0]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + vofs[dt][1]));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        const u32x4_t v = {l2.x, l2.y, h2.x, h2.y};
        return v;
      };
      u32x4_t vf[2];
      vf[0] = vread(0, 0);
#pragma unroll
      for (int i = 0; i < 4 * NDT; ++i) {
        if (i + 1 < 4 * NDT) vf[(i + 1) & 1] = vread((i + 1) / NDT, (i + 1) % NDT);
        o[i % NDT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf[i & 1]), __builtin_bit_cast(bf16x8_t, pb[i / NDT]), o[i % NDT], 0, 0, 0);
      }
    }
  }
  if (qr >= Sq) return;
  const int DV = p.DV;
  if (p.nsplit > 1) {      // unnormalised O, reference (natural-log units) and row sum for attn_combine_kernel
    float* pp = p.part + ((((int64_t)b * p.Hq + head) * p.nsplit + split) * Sq + qr) * (DV + 2);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        if (d0 < DV) {
          *(float2*)(pp + d0) = float2{o[dt][4 * g], o[dt][4 * g + 1]};
          *(float2*)(pp + d0 + 2) = float2{o[dt][4 * g + 2], o[dt][4 * g + 3]};
        }
      }
    if (h == 0) { pp[DV] = m_i * 0.6931471805599453f; pp[DV + 1] = l_i; }
    return;
  }
  const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
  bf16_t* Og = (bf16_t*)p.O + (int64_t)b * p.o_sb + (int64_t)head * p.o_sh + (int64_t)qr * p.o_ss;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = dt * 32 + 8 * g + 4 * h;
      if (d0 < DV) {
        uint2 v;
        v.x = f2bf2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
        v.y = f2bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *(uint2*)(Og + d0) = v;
      }
    }
}

template <int DP, bool CAUSAL, int QB, int NW>
int launch_dma(const AttnArgs& p, hipStream_t st) {
  constexpr int lds = 3 * 2 * 64 * 256;      // three stages of (K tile + V tile)
  constexpr int BQ = NW * 32 * QB;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)attn_dma_kernel<DP, CAUSAL, QB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    once = true;
  }
  dim3 grid((p.Sq + BQ - 1) / BQ, p.Hq, p.B);
  attn_dma_kernel<DP, CAUSAL, QB, NW><<<grid, NW * 64, lds, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

}  // namespace

bool attn_dma_dv_eligible(const AttnArgs& p) {
  return p.D == 256 && p.DV == 64 && p.Hq == p.Hkv && p.causal == 0 && !p.fold && !p.skv_dev && p.Sq >= 256 && (p.nsplit == 1 || p.split_len % 64 == 0) &&
         ((p.o_ss | p.o_sh | p.o_sb) & 3) == 0 && ((uintptr_t)p.O & 7) == 0;
}

int attn_dma_dv_launch(const AttnArgs& p, hipStream_t st) {
  constexpr int lds = 3 * (64 * 512 + 64 * 256);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)attn_dma_d256v64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    once = true;
  }
  dim3 grid(((p.Sq + 255) / 256) * p.nsplit, p.Hq, p.B);
  attn_dma_d256v64_kernel<<<grid, 512, lds, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

bool attn_dma_eligible(const AttnArgs& p) {
  const int padded = (p.Sq + 255) / 256 * 256;      // 256-row query tiles: below 1024 queries at most 1/8 of the rows may idle (CLIP's 577 -> 768 stays on attn_kernel)
  return p.DV == p.D && p.D % 8 == 0 && p.D > 32 && p.D <= 128 && p.nsplit == 1 && !p.fold && !p.skv_dev && (p.causal == 0 || p.causal == 1) && p.Sq >= 512 &&
         (p.Sq >= 1024 || (padded - p.Sq) * 8 <= padded) && ((p.o_ss | p.o_sh | p.o_sb) & 3) == 0 && ((uintptr_t)p.O & 7) == 0;
}

int attn_dma_launch(const AttnArgs& p, hipStream_t st) {
  const bool c = p.causal == 1;
  if (p.D <= 64) return c ? launch_dma<64, true, 1, 8>(p, st) : launch_dma<64, false, 1, 8>(p, st);
  if (p.D <= 96) return c ? launch_dma<96, true, 1, 8>(p, st) : launch_dma<96, false, 1, 8>(p, st);
  return c ? launch_dma<128, true, 1, 8>(p, st) : launch_dma<128, false, 1, 8>(p, st);
}
