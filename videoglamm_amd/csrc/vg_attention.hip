// vg_attention: flash-style attention for gfx950 (see include/vg_kernels.h).
//
// Workgroup = NW waves, each owning 32 query rows; K/V tiles of BKV keys staged in LDS once per
// workgroup and shared by all waves; Q tile staged once.  Per wave and 32-key sub-tile:
//   S^T[key,q] = K·Q^T   (MFMA 32x32, "swapped" product: a lane owns ONE query column q = lane&31 and
//                         16 key rows, so the softmax row statistics are lane-local + one xor-32 shuffle)
//   online softmax in registers (running max m, running sum l per lane/query)
//   O^T[d,q]  += V^T·P^T (P^T stays in the accumulator registers it was produced in and is fed to
//                         the MFMA B operand directly; the key order inside a k-step is the MFMA
//                         row map, applied identically to the V gather, so no cross-lane traffic)
// LDS rows are padded by 16 B so the 16-byte fragment reads are bank-conflict free.
#include "vg_common.h"
#include <math.h>
#include <stdlib.h>

#include "vg_attn_args.h"

template <typename T> struct AMma;
template <> struct AMma<bf16_t> {
  static __device__ __forceinline__ void qk(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct AMma<float> {
  static __device__ __forceinline__ void qk(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

// O^T[dt] += V^T P^T for one 32-key sub-tile; vs = LDS pointer to the sub-tile's first key row.
template <typename T, int RS>
__device__ __forceinline__ void pv_step(const char* vs, int col, int h, const f32x16_t& p, f32x16_t& o);

// bf16 PV step.  V is staged TRANSPOSED: Vt[d][key], 128-byte rows (64 keys), keys permuted inside every 16-key block
// ([0-3, 8-11, 4-7, 12-15]) so that the 8 keys an MFMA A operand needs — rows {4h..4h+3, 8+4h..8+4h+3} of the P
// accumulator (the 32x32 C-layout row map) — are one 16-byte group; groups are XOR-swizzled with (d>>1)&7 so the 32
// rows of a ds_read_b128 hit distinct bank slots.  One ds_read_b128 per MFMA instead of sixteen ds_read_u16.
__device__ __forceinline__ void pv_step_bf16t(const char* vt, int d, int kt, int h, const f32x16_t& p, f32x16_t& o) {
  const char* row = vt + d * 128;
  const int sw = (d >> 1) & 7;
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const u32x4_t a = *(const u32x4_t*)(row + (((kt * 4 + st * 2 + h) ^ sw) << 4));
    u32x4_t b;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int r0 = st * 8 + jj * 2;
      b[jj] = f2bf2(p[r0], p[r0 + 1]);
    }
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), o, 0, 0, 0);
  }
}
template <int RS>
__device__ __forceinline__ void pv_step_f32(const char* vs, int col, int h, const f32x16_t& p, f32x16_t& o) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float a = *(const float*)(vs + mfma32_row(r, h) * RS + col * 4);
    o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p[r], o, 0, 0, 0);
  }
}

// bytes of padding behind a bf16 V row in LDS so that row r + 1 starts 64 bytes (mod 256) behind row r: 256 -> 320, 192 -> 192, 128 -> 192, 64 -> 64
#define ATTN_VPAD(DP) ((DP) == 128 ? 64 : (DP) == 96 ? 0 : (DP) == 64 ? 64 : (DP) == 32 ? 0 : 16)
// measured constants of the tile loop (DESIGN_HISTORY.md section 5d has the A/B of each): K/V tile prefetch, Q fragments in registers, V through
// ds_read_b64_tr_b16, two K/V tile buffers in the DV = 64 form, minimum workgroups per CU in __launch_bounds__
#define VG_ATTN_PREFETCH 1
#define VG_ATTN_QREG 1
#define VG_ATTN_VTR 1
#define VG_ATTN_DB 1
#define VG_ATTN_MINW 2
#define VG_ATTN_MINW96 2

// combine a value with the other half-wave's (lane ^ 32): v_permlane32_swap instead of __shfl_xor's ds_bpermute — no LDS instruction, no lgkmcnt wait in a loop
// that is bound by its LDS traffic
__device__ __forceinline__ float attn_xor32(float x, bool mx) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
  return mx ? fmaxf(a, b) : a + b;
}

// KS = 2 (head dim 256, SAM2's memory attention): the two 32-key halves of every KV tile go to two different waves of the same
// 32 query rows — 8 waves on the LDS footprint of 4, each with its own (O, m, l) over its half of the keys, merged through LDS
// once after the last tile.  At head dim 256 the 128 accumulator registers of a wave leave room for one wave per SIMD only
// when a wave owns whole tiles; two waves per SIMD let one wave's softmax / staging VALU run under the other's MFMAs.
// DVP (default DP): padded head dim of V and of the output when it differs from Q / K's — SAM2's memory cross-attention keeps its values in the
// memory's own 64 dims (softmax rows sum to one, so P (M Wv^T + b) = (P M) Wv^T + b: the v-projection moves behind the attention, onto 4096 rows
// instead of 28 000, and the PV half of the kernel shrinks four-fold: vg_attention_dv, r04).  Only on the paths without the transpose read.
template <typename T, int DP, int BKV, int NW, int KS = 1, int DVP = DP>
__global__ __launch_bounds__(NW * 64, (NW > 4 ? 1 : (DP <= 96 ? VG_ATTN_MINW96 : (DP <= 128 ? VG_ATTN_MINW : 1)))) void attn_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T);
  constexpr int KPC = 16 / ES;
  constexpr int RS = DP * ES + 16;  // LDS row stride in bytes
  constexpr int CPR = DP * ES / 16; // 16-byte chunks per row
  constexpr int NQW = NW / KS;      // waves along the query rows
  constexpr int BQ = NQW * 32;
  constexpr int NT = NW * 64;
  constexpr int NG = DP * ES / 32;  // k-groups (two 16-byte chunks each) along the head dim
  constexpr int NDT = DVP / 32;
  constexpr int RSVF = DVP * ES + 16;   // V row stride where V is staged as rows (fp32)
  constexpr int CPRV = DVP * ES / 16;
  static_assert(DVP == DP || DP > 128, "a separate value dim only on the head-dim-256 paths");
  constexpr int NKT = BKV / 32 / KS;   // 32-key sub-tiles of a KV tile this wave multiplies
  static_assert(KS == 1 || (KS == 2 && BKV == 64 && sizeof(T) == 2), "key split: two waves per 64-key bf16 tile");
  // VTR (bf16, head dim <= 128 = the PIPE path): V goes to LDS as it comes (padded rows, like K) and the PV step's V^T fragments are read with
  // ds_read_b64_tr_b16 — the transposing staging pass (≈ 80 bit-shuffle VALU instructions per thread and tile in a VALU-bound loop) is gone.
  // Row stride: the 16 lanes of a transpose read touch 4 rows x 4 column quads of 8 bytes; rows must land 8 banks apart
  constexpr bool VTR = VG_ATTN_VTR && sizeof(T) == 2 && DP <= 128;
  // (r06: a ds_read_b64_tr_b16 is serviced in two 32-lane groups — 4 key rows x 64 bytes each, MI355X_MICROARCH.md LDS table — so consecutive rows must sit 16
  // banks = 64 bytes apart modulo 256, not 8: SQ_LDS_BANK_CONFLICT was 32 % of the LDS cycles of the Hiera global blocks with rows 52 / 72 dwords apart)
  constexpr int RSV = DP * ES + ATTN_VPAD(DP);
  // DB (key-split kernel with 64-wide values, vg_attention_dv): TWO K / V tile buffers — a wave writes tile t + 1 into the other buffer as soon as
  // it is done multiplying tile t, so the loop has one barrier per tile instead of two and the staging writes of the early waves run under the MFMAs
  // of the late ones (K 33 KB + V^T 8 KB per buffer: 148 KB with the Q tile; the head-dim-256 values of the self-attention form do not fit twice)
  constexpr bool DB = VG_ATTN_DB && sizeof(T) == 2 && KS == 2 && DVP <= 64;
  char* Qs = smem;
  char* Ks = Qs + BQ * RS;
  char* Vs = Ks + BKV * RS;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = (tid >> 6) % NQW, kh = (tid >> 6) / NQW;   // query-row group, key half (KS == 2)
  // XCD-aware order (VG_ATTN_XCD, p.xcd): workgroup L of the launch runs on XCD L % 8.  Every query tile of a (batch, head) walks the same
  // K / V, and the G query heads of a KV head share them too: XCD x takes a contiguous block of (batch, head) pairs and runs each
  // pair's query tiles back to back, so K / V are fetched into ONE L2 instead of eight
  int bx = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  if (p.xcd) {
    const int nx = gridDim.x, BH = gridDim.y * gridDim.z;
    const int L = (blockIdx.z * gridDim.y + blockIdx.y) * nx + blockIdx.x, j = L >> 3;
    const int bh = (L & 7) * (BH >> 3) + j / nx;
    bx = j % nx;
    head = bh % (int)gridDim.y;
    b = bh / (int)gridDim.y;
  }
  const int split = bx % p.nsplit;
  // causal: the LAST query tiles walk the most keys — hand them out first, so the short ones fill the tail of the launch
  const int qtile = (p.causal > 0 && !p.fold) ? ((int)gridDim.x / p.nsplit - 1 - bx / p.nsplit) : bx / p.nsplit;
  const int q0 = qtile * BQ;
  const int G = p.Hq / p.Hkv;
  const int kvh = p.fold ? head : head / G;
  const int D = p.D, DV = p.DV, Sq = p.Sq, Skv = p.skv_dev ? (*p.skv_dev + p.Sq) : p.Skv;
  const int nrow = p.fold ? G * Sq : Sq;   // valid rows of the query tile space
  const T* Qg = (const T*)p.Q + (int64_t)b * p.q_sb + (p.fold ? 0 : (int64_t)head * p.q_sh);
  const T* Kg = (const T*)p.K + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh;
  const T* Vg = (const T*)p.V + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  for (int idx = tid; idx < BQ * CPR; idx += NT) {
    const int row = idx / CPR, c = idx - row * CPR;
    const int qr = q0 + row;
    u32x4_t v = zero4;
    if (qr < nrow && c * KPC < D) {
      const int64_t qo = p.fold ? (int64_t)(head * G + qr / Sq) * p.q_sh + (int64_t)(qr % Sq) * p.q_ss : (int64_t)qr * p.q_ss;
      v = *(const u32x4_t*)(Qg + qo + c * KPC);
    }
    *(u32x4_t*)(Qs + row * RS + c * 16) = v;
  }

  const int off = Skv - Sq;
  int kv_end = Skv;
  if (p.causal > 0) {
    const int lim = (p.fold ? Sq : q0 + BQ) + off;  // keys >= lim are invisible to every query of this block
    kv_end = lim < Skv ? (lim > 0 ? lim : 0) : Skv;
  }
  // causal < 0: block-diagonal mask, windows of wtok = -causal tokens packed back to back along the sequence (Hiera's
  // 16- and 64-token windows: 8 or 2 of them fill one 128-query tile instead of one padded tile each)
  const int wtok = p.causal < 0 ? -p.causal : 0;
  // causal >= 2: causal mask with a sliding window of `causal` visible keys (the query's own position included):
  // key j visible to query i iff i + off - causal < j <= i + off  (HF Phi-3 / Mistral sliding_window)
  const int win = p.causal > 1 ? p.causal : 0;
  int kv_first = 0;
  if (win) {
    const int lo = (p.fold ? 0 : q0) + off - win + 1;      // first key any query of this block sees
    kv_first = lo > 0 ? lo / BKV * BKV : 0;
  }
  if (wtok) {
    kv_first = q0 / wtok * wtok;
    const int e = (q0 + BQ + wtok - 1) / wtok * wtok;
    kv_end = e < Skv ? e : Skv;
  }

  float m_i = -INFINITY, l_i = 0.f;                       // running max in log2 units (see the softmax below)
  const float sl2 = p.scale * 1.4426950408889634f;
  f32x16_t o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

  const int q_local = wave * 32 + l31;
  const int q_row = q0 + q_local;                        // row in the tile space
  const int q_idx = p.fold ? q_row % Sq : q_row;         // query position (causal mask)
  const int q_head = p.fold ? head * G + q_row / Sq : head;
  const char* qrow = Qs + q_local * RS + h * 16;

  const int wlo = wtok ? q_idx / wtok * wtok : 0, whi = wtok ? wlo + wtok : 0x7fffffff;
  int kv_begin = kv_first;
  if (p.nsplit > 1) {
    kv_begin = split * p.split_len;
    const int e = kv_begin + p.split_len;
    kv_end = e < kv_end ? e : kv_end;
    if (kv_begin < kv_first) kv_begin = kv_first;
  }
  // K/V staging: a tile is fetched into registers one iteration ahead (the loads of tile t+1 are in flight while
  // tile t is multiplied), then written to LDS — K as padded rows, V (bf16) transposed for pv_step_bf16t
  constexpr bool VT = sizeof(T) == 2 && !VTR;
  static_assert(!VT || BKV == 64, "the transposed V image assumes 64-key tiles");
  constexpr int NKI = (BKV * CPR + NT - 1) / NT;                 // K chunks per thread per tile
  constexpr int NVI = VT ? (16 * CPRV + NT - 1) / NT : (BKV * CPRV + NT - 1) / NT;       // V items per thread: (key quad, chunk) | chunks
  u32x4_t kreg[NKI], vreg[VT ? NVI * 4 : NVI];
  auto fetch = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int idx = tid + i * NT, row = idx / CPR, c = idx - row * CPR, key = kv0 + row;
      kreg[i] = (idx < BKV * CPR && key < Skv && c * KPC < D) ? *(const u32x4_t*)(Kg + (int64_t)key * p.k_ss + c * KPC) : zero4;
    }
    if constexpr (!VT) {
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        // rows past the end of the sequence read the LAST row instead of being zero-filled: their probabilities are exactly zero (masked scores), so
        // any finite value does — and the load is unconditional (r04: with the 64-wide value rows of vg_attention_dv the zero-filling select went
        // wrong in fp32 exactly when one wave's four row groups were all valid and the other's last one was not: Skv % 32 in [28, 31])
        const int idx = tid + i * NT, row = idx / CPRV, c = idx - row * CPRV, key = min(kv0 + row, Skv - 1);
        vreg[i] = (idx < BKV * CPRV && c * KPC < DV) ? *(const u32x4_t*)(Vg + (int64_t)key * p.v_ss + c * KPC) : zero4;
      }
    }
    if constexpr (VT) {
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int item = tid + i * NT, kq = item & 15, c = item >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = kv0 + kq * 4 + j;
          vreg[i * 4 + j] = (item < 16 * CPRV && key < Skv && c * KPC < DV) ? *(const u32x4_t*)(Vg + (int64_t)key * p.v_ss + c * KPC) : zero4;
        }
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int idx = tid + i * NT, row = idx / CPR, c = idx - row * CPR;
      if (idx < BKV * CPR) *(u32x4_t*)(Ks + row * RS + c * 16) = kreg[i];
    }
    if constexpr (!VT) {
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int idx = tid + i * NT, row = idx / CPRV, c = idx - row * CPRV;
        if (idx < BKV * CPRV) *(u32x4_t*)(Vs + row * (VTR ? RSV : RSVF) + c * 16) = vreg[i];
      }
    }
    if constexpr (VT) {
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int item = tid + i * NT, kq = item & 15, c = item >> 4;
        if (item < 16 * CPRV) {
          const int b4 = (kq * 4) & 15;
          const int pos = ((kq * 4) & ~15) + (b4 == 4 ? 8 : (b4 == 8 ? 4 : b4));   // key permutation inside a 16-block
          const int grp = pos >> 3, half = (pos >> 2) & 1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int d = c * 8 + e, w = e >> 1, sh = (e & 1) * 16;
            const uint32_t lo = ((vreg[i * 4 + 0][w] >> sh) & 0xffffu) | (((vreg[i * 4 + 1][w] >> sh) & 0xffffu) << 16);
            const uint32_t hi = ((vreg[i * 4 + 2][w] >> sh) & 0xffffu) | (((vreg[i * 4 + 3][w] >> sh) & 0xffffu) << 16);
            uint2 val;
            val.x = lo;
            val.y = hi;
            *(uint2*)(Vs + d * 128 + ((grp ^ ((d >> 1) & 7)) << 4) + half * 8) = val;
          }
        }
      }
    }
  };
  constexpr bool PF = VG_ATTN_PREFETCH;
  if (PF && kv_begin < kv_end) fetch(kv_begin);
  // QREG (bf16, head dim <= 128): the wave's Q fragments stay in registers for the whole KV walk — one of every five LDS reads of a tile
  // (NG of 5 NG per 64 keys) was the same Q bytes again, and the loop is LDS-read-bound at eight waves per CU
  // (r04: also the head-dim-256 kernel when its values are 64 wide — vg_attention_dv: 2 instead of 8 output accumulators leave room for the 64 Q registers)
  constexpr bool QREG = VG_ATTN_QREG && sizeof(T) == 2 && (DP <= 128 || DVP <= 64);
  u32x4_t qreg[QREG ? NG : 1];
  if constexpr (QREG) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) qreg[g] = *(const u32x4_t*)(qrow + g * 32);
  }
  static_assert(!DB || (PF && QREG), "the double-buffered loop prefetches and keeps Q in registers");
  // (r05, measured and dropped: running the kh = 1 waves half an iteration behind their SIMD partners — softmax / PV of tile t - 1 beside the partner's
  // QK^T of tile t, V^T tiles in a ring of three — changed nothing: 131 vs 126 us on the C2 shape, 3 spilled registers; tools/lab/attn_db_ab.sh)
  constexpr bool PIPE = sizeof(T) == 2 && (DP <= 128 || DVP <= 64);
  f32x16_t s[NKT];
  auto qk_tile = [&]() {
    // PIPE (bf16, head dim <= 128): the K / Q fragments of k-group g + 1 are requested before the MFMAs of group g issue, and the
    // sub-tiles' independent accumulators alternate — left to itself the compiler emits read, wait, MFMA per fragment (measured r02:
    // every MFMA of the loop then pays a full LDS round trip); head dim 256 has no registers to spare for the second fragment set
    if constexpr (PIPE) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      const char* krow0 = Ks + (kh * NKT * 32 + l31) * RS + h * 16;
      u32x4_t kf[2][NKT], qf[2];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) kf[0][kt] = *(const u32x4_t*)(krow0 + kt * 32 * RS);
      if constexpr (!QREG) qf[0] = *(const u32x4_t*)qrow;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < NG) {
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt) kf[nxt][kt] = *(const u32x4_t*)(krow0 + kt * 32 * RS + (g + 1) * 32);
          if constexpr (!QREG) qf[nxt] = *(const u32x4_t*)(qrow + (g + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) AMma<T>::qk(kf[cur][kt], QREG ? qreg[g] : qf[cur], s[kt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
        const char* krow = Ks + ((kh * NKT + kt) * 32 + l31) * RS + h * 16;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const u32x4_t a = *(const u32x4_t*)(krow + g * 32);
          const u32x4_t bq = *(const u32x4_t*)(qrow + g * 32);
          AMma<T>::qk(a, bq, s[kt]);
        }
      }
    }

  };
  auto smpv_tile = [&](const int kv0, const char* Vt) {
    // softmax in the exp2 domain: t = s * (scale * log2 e), so every score costs one fma + one v_exp_f32; m_i and the
    // partials keep natural-log units (m = max(t) / log2 e) for the split-KV merge.  The mask arithmetic only runs on
    // tiles that need it (sequence end, causal diagonal, window edges): the inner loop is VALU-bound, not MFMA-bound.
    const bool need_mask = kv0 + BKV > Skv || p.causal < 0 || (p.causal > 0 && kv0 + BKV - 1 > (p.fold ? 0 : q0) + off) ||
                           (win && kv0 <= (p.fold ? Sq - 1 : q0 + BQ - 1) + off - win);
    float mx = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + (kh * NKT + kt) * 32 + mfma32_row(r, h);
          const bool ok = key < Skv && (p.causal <= 0 || key <= q_idx + off) && key >= wlo && key < whi && (!win || key > q_idx + off - win);
          const float v = ok ? s[kt][r] * sl2 : -INFINITY;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
    } else {
      // no mask: the scale is positive, so the maximum is taken over the raw scores (v_max3) and scale and shift ride in ONE fma per score in front of
      // the exponential below — 16 + 32 VALU instructions per 32 scores instead of 32 + 32 + 32 (multiply, maximum, subtract)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
      mx *= sl2;
    }
    mx = attn_xor32(mx, true);
    // Deferred maximum (r06): the reference m_i only moves when some row of the wave has no reference yet or its maximum outgrew the reference by more than
    // 2^24 — P <= 2^24 is as exact in fp32 / bf16 as P <= 1 (same mantissa, exponents far from the range's ends; terms that flush to zero are far below the
    // row's largest).  With a reference that tracks every new maximum the rescale of O and l ran in nearly every tile of a 64-lane wave (a new row maximum
    // somewhere is the rule for the first hundred tiles of random data); now it runs in a row's first tile and then almost never.
    const bool move = __any(mx > m_i + 24.0f || (m_i == -INFINITY && mx != -INFINITY));
    const float m_new = move ? fmaxf(m_i, mx) : m_i;          // log2 units
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = move ? exp2f(m_i - m_safe) : 1.0f;
    float rs = 0.f;
    if (need_mask) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[kt][r] - m_safe);
          s[kt][r] = pv;
          rs += pv;
        }
    } else {
      const float nm = -m_safe;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sl2, nm));
          s[kt][r] = pv;
          rs += pv;
        }
    }
    rs = attn_xor32(rs, false);
    l_i = l_i * alpha + rs;
    m_i = m_new;
    if (move) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }

    if constexpr (PIPE) {
      // P packed once per 16-key step (the same B operand for every d-tile); V^T fragments of step t + 1 requested before the
      // MFMAs of step t; the d-tiles' accumulators are independent chains (see pv_step_bf16t for the V^T image)
      u32x4_t pb[NKT * 2];
#pragma unroll
      for (int t = 0; t < NKT * 2; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) pb[t][jj] = f2bf2(s[t >> 1][(t & 1) * 8 + jj * 2], s[t >> 1][(t & 1) * 8 + jj * 2 + 1]);
      u32x4_t vf[2][NDT];
      // VTR: lane i of a 16-lane group hands ds_read_b64_tr_b16 the 8-byte piece (key row i >> 2, column quad i & 3) of a 4-key x 16-column
      // block and receives column i of it = 4 keys of ONE head-dim column (tools/lab/tr_probe.hip); the MFMA A operand of 16-key step t wants
      // the keys whose P values this lane's accumulator registers hold: 16 t + 4 h + {0..3} and 16 t + 8 + 4 h + {0..3} -> two reads
      const char* vtr = Vt + (4 * h + ((lane & 15) >> 2)) * RSV + ((((lane >> 4) & 1) * 16 + 4 * (lane & 3)) << 1);
      auto vread = [&](int t, int dt) -> u32x4_t {
        if constexpr (VTR) {
          typedef short s16x4_t __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) s16x4_t* lds4_t;
          const char* a = vtr + ((kh * NKT + (t >> 1)) * 32 + (t & 1) * 16) * RSV + dt * 64;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)a);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + 8 * RSV));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          const u32x4_t v = {l2.x, l2.y, h2.x, h2.y};
          return v;
        } else {
          const int d = dt * 32 + l31;
          return *(const u32x4_t*)(Vt + d * 128 + ((((kh * NKT + (t >> 1)) * 4 + (t & 1) * 2 + h) ^ ((d >> 1) & 7)) << 4));
        }
      };
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) vf[0][dt] = vread(0, dt);
#pragma unroll
      for (int t = 0; t < NKT * 2; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (t + 1 < NKT * 2) {
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) vf[nxt][dt] = vread(t + 1, dt);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf[cur][dt]), __builtin_bit_cast(bf16x8_t, pb[t]), o[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const char* vs = Vt + kt * 32 * RSVF;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          if constexpr (sizeof(T) == 2) pv_step_bf16t(Vt, dt * 32 + l31, kh * NKT + kt, h, s[kt], o[dt]);
          else pv_step_f32<RSVF>(vs, dt * 32 + l31, h, s[kt], o[dt]);
        }
      }
    }
  };
  char* const Kb0 = Qs + BQ * RS;
  char* const Vb0 = Kb0 + (DB ? 2 : 1) * BKV * RS;
  constexpr int VBY = DVP * 128;
  if constexpr (DB) {
    Vs = Vb0;
    if (kv_begin < kv_end) {
      stage();
      __syncthreads();
      if (kv_begin + BKV < kv_end) fetch(kv_begin + BKV);
    }
  }
  int ti = 0;                                   // tile index of the walk (DB: K / V buffer ti & 1)
  for (int kv0 = kv_begin; kv0 < kv_end; kv0 += BKV, ++ti) {
    if constexpr (!DB) {
      if (!PF) fetch(kv0);
      __syncthreads();
      stage();
      __syncthreads();
      if (PF && kv0 + BKV < kv_end) fetch(kv0 + BKV);
      qk_tile();
      smpv_tile(kv0, Vs);
    } else {
      qk_tile();
      smpv_tile(kv0, Vb0 + (ti & 1) * VBY);
      if (kv0 + BKV < kv_end) {                 // tile t + 1 (in registers since the last barrier) goes to the buffers nobody reads any more
        Ks = Kb0 + ((ti + 1) & 1) * (BKV * RS);
        Vs = Vb0 + ((ti + 1) & 1) * VBY;
        stage();
      }
      __syncthreads();
      if (kv0 + 2 * BKV < kv_end) fetch(kv0 + 2 * BKV);
    }
  }

  if constexpr (KS == 2) {
    // merge the two key halves: the kh = 1 waves park (O, m, l) in LDS (lane-major: conflict-free), their kh = 0 partners fold them in
    float* xo = (float*)smem + (int64_t)wave * (NDT * 16 + 2) * 64 + lane;
    __syncthreads();                                          // every wave is done with the last K / V tile
    if (kh == 1) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) xo[(dt * 16 + r) * 64] = o[dt][r];
      xo[NDT * 16 * 64] = m_i;
      xo[(NDT * 16 + 1) * 64] = l_i;
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = xo[NDT * 16 * 64], l1 = xo[(NDT * 16 + 1) * 64];
    const float m_new = fmaxf(m_i, m1);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float a0 = exp2f(m_i - m_safe), a1 = exp2f(m1 - m_safe);
    l_i = l_i * a0 + l1 * a1;
    m_i = m_new;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = o[dt][r] * a0 + xo[(dt * 16 + r) * 64] * a1;
  }
  if (p.nsplit > 1) {
    if (q_row < nrow) {
      float* pp = p.part + ((((int64_t)b * p.Hq + q_head) * p.nsplit + split) * Sq + q_idx) * (DV + 2);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {       // 4 consecutive head-dim values per register quad; a partial row starts 8-byte aligned (D + 2 is even)
          const int d0 = dt * 32 + 8 * g + 4 * h;
          if (d0 < DV) {
            const float2 lo = {o[dt][4 * g], o[dt][4 * g + 1]}, hi = {o[dt][4 * g + 2], o[dt][4 * g + 3]};
            *(float2*)(pp + d0) = lo;
            *(float2*)(pp + d0 + 2) = hi;
          }
        }
      if (h == 0) { pp[DV] = m_i * 0.6931471805599453f; pp[DV + 1] = l_i; }
    }
    return;
  }
  if (q_row < nrow) {
    const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
    T* Og = (T*)p.O + (int64_t)b * p.o_sb + (int64_t)q_head * p.o_sh + (int64_t)q_idx * p.o_ss;
    // a lane holds 4 consecutive head-dim values per register quad (rows 8 g + 4 h + 0..3 of the 32x32 C layout): one 8-byte (bf16) /
    // 16-byte (fp32) store per quad instead of four scalar ones when the output rows keep that alignment (D % 8 == 0 always holds)
    const bool ovec = ((p.o_ss | p.o_sh | p.o_sb) & 3) == 0 && ((uintptr_t)p.O & (4 * sizeof(T) - 1)) == 0;
    if (ovec) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = dt * 32 + 8 * g + 4 * h;
          if (d0 < DV) {
            if constexpr (sizeof(T) == 2) {
              uint2 v;
              v.x = f2bf2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
              v.y = f2bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
              *(uint2*)(Og + d0) = v;
            } else {
              const f32x4_t v = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
              *(f32x4_t*)(Og + d0) = v;
            }
          }
        }
    } else {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = dt * 32 + mfma32_row(r, h);
          if (d < DV) vg_elt<T>::st(Og + d, o[dt][r] * inv);
        }
    }
  }
}

template <typename T, int DP, int BKV, int NW, int KS = 1, int DVP = DP>
static int launch_attn(const AttnArgs& p, hipStream_t st) {
  constexpr int RS = DP * sizeof(T) + 16;
  constexpr int RSVF = DVP * sizeof(T) + 16;
  constexpr bool VTR = VG_ATTN_VTR && sizeof(T) == 2 && DP <= 128;
  constexpr int RSV = DP * (int)sizeof(T) + ATTN_VPAD(DP);
  constexpr int vbytes = VTR ? BKV * RSV : (sizeof(T) == 2 ? DVP * 128 : BKV * RSVF);   // bf16, head dim 256: transposed V image, DP rows of 64 keys
  constexpr int BQ = NW / KS * 32;
  constexpr bool DB = VG_ATTN_DB && sizeof(T) == 2 && KS == 2 && DVP <= 64;
  constexpr int lds = (BQ + BKV) * RS + vbytes + (DB ? BKV * RS + vbytes : 0);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(KS == 1 || lds >= (NW / KS) * (DVP / 32 * 16 + 2) * 64 * 4, "the key-half merge reuses the tile buffers");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_kernel<T, DP, BKV, NW, KS, DVP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  dim3 grid(((p.Sq + BQ - 1) / BQ) * p.nsplit, p.Hq, p.B);
  if (p.fold) grid = dim3(p.nsplit, p.Hkv, p.B);
  attn_kernel<T, DP, BKV, NW, KS, DVP><<<grid, NW * 64, lds, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Hiera's 256-token windows (stage 3 of Hiera-B+/L: 16x16 windows, head_dim 72; R/.../backbones/hieradet.py:37-83 on windows
// made by backbones/utils.py:16-38) — one workgroup per (window, head), the WHOLE window's K and V staged once:
//   * every global load of the workgroup (K rows, V rows, the waves' Q fragments) is requested up front, one barrier, then
//     no further synchronisation: a wave owns 64 query rows (two 32-row passes) against all 256 keys, so the softmax is a
//     single pass over 128 score registers per lane (no running max / rescale), P feeds the PV MFMA from registers;
//   * K rows sit at an odd number of 16-byte slots (conflict-free ds_read_b128 without padding bytes to zero), the k-step
//     that straddles the end of a 72-wide row reads the next row's first chunk against a ZERO Q operand;
//   * V is staged transposed / key-permuted / swizzled exactly like attn_kernel's (pv_step_bf16t), four 64-key tiles;
//   * 73.7 KB of LDS and <= 256 registers: two workgroups per CU, one staging while the other multiplies.
// The generic kernel walks such a window as 2 query tiles x 4 KV tiles with two barriers and a register->LDS transpose per
// tile: ~90 TFLOP/s on this shape (r02 C2 profile: 28 ms per 32-frame clip); this one is bounded by its MFMAs.
struct WinAttnArgs {
  const void* Q; const void* K; const void* V; void* O;
  int Bw, H;
  int64_t q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh;
  float scale;
};

template <int D>
__global__ __launch_bounds__(256, 2) void win256_attn_kernel(WinAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int W = 256;                        // tokens per window
  constexpr int CH = D / 8;                     // 16-byte chunks per row
  constexpr int NG = (D + 15) / 16;             // QK^T k-steps
  constexpr int NDT = (D + 31) / 32;            // output d-tiles
  constexpr int RS = (CH | 1) * 16;             // K row stride: an odd number of 16-byte slots
  constexpr int VT_BYTES = 4 * D * 128;         // four 64-key tiles of [D][64 keys]
  char* Vs = smem;
  char* Ks = smem + VT_BYTES;                   // (reads past the last V^T row / the last K chunk land in finite data)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  // XCD-aware (window, head) order: workgroup L lands on XCD L % 8; the heads of one window read neighbouring 144-byte pieces of the same
  // rows of the fused q|k|v tensor, so a window's heads run back to back on ONE XCD and share those lines in its L2 (with head = blockIdx.x
  // the eight heads sat on eight XCDs: PMC traffic 540 MB per launch for 301 MB algorithmic, r02)
  int head = blockIdx.x, win = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, j = L >> 3;
    win = (L & 7) + 8 * (j / (int)gridDim.x);
    head = j % (int)gridDim.x;
  }
  const bf16_t* Qg = (const bf16_t*)p.Q + (int64_t)win * p.q_sb + (int64_t)head * p.q_sh;
  const bf16_t* Kg = (const bf16_t*)p.K + (int64_t)win * p.k_sb + (int64_t)head * p.k_sh;
  const bf16_t* Vg = (const bf16_t*)p.V + (int64_t)win * p.v_sb + (int64_t)head * p.v_sh;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  // ---- every global load up front
  u32x4_t kreg[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int idx = tid + i * 256, row = idx / CH, c = idx - row * CH;
    kreg[i] = *(const u32x4_t*)(Kg + (int64_t)row * p.k_ss + c * 8);
  }
  constexpr int NVI = (64 * CH + 255) / 256;
  u32x4_t vreg[NVI * 4];
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int item = tid + i * 256, kq = item & 63, c = item >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      vreg[i * 4 + j] = item < 64 * CH ? *(const u32x4_t*)(Vg + (int64_t)(kq * 4 + j) * p.v_ss + c * 8) : zero4;
  }
  u32x4_t qf[NG];                                // Q fragments (MFMA B operand) of the wave's current 32-row pass
  auto load_q = [&](int sub) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c = 2 * g + h;
      qf[g] = c < CH ? *(const u32x4_t*)(Qg + (int64_t)(wave * 64 + sub * 32 + l31) * p.q_ss + c * 8) : zero4;
    }
  };
  load_q(0);
  // ---- stage K (rows) and V (transposed image)
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int idx = tid + i * 256, row = idx / CH, c = idx - row * CH;
    *(u32x4_t*)(Ks + row * RS + c * 16) = kreg[i];
  }
  if (tid < 4) *(uint32_t*)(Ks + W * RS + tid * 4) = 0u;      // the chunk after the last row: read against a zero Q operand
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int item = tid + i * 256, kq64 = item & 63, c = item >> 6;
    if (item < 64 * CH) {
      const int tile = kq64 >> 4, kq = kq64 & 15;
      const int b4 = (kq * 4) & 15;
      const int pos = ((kq * 4) & ~15) + (b4 == 4 ? 8 : (b4 == 8 ? 4 : b4));   // key permutation inside a 16-block (pv_step_bf16t)
      const int grp = pos >> 3, half = (pos >> 2) & 1;
      char* vt = Vs + tile * D * 128;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = c * 8 + e, w = e >> 1, sh = (e & 1) * 16;
        uint2 val;
        val.x = ((vreg[i * 4 + 0][w] >> sh) & 0xffffu) | (((vreg[i * 4 + 1][w] >> sh) & 0xffffu) << 16);
        val.y = ((vreg[i * 4 + 2][w] >> sh) & 0xffffu) | (((vreg[i * 4 + 3][w] >> sh) & 0xffffu) << 16);
        *(uint2*)(vt + d * 128 + ((grp ^ ((d >> 1) & 7)) << 4) + half * 8) = val;
      }
    }
  }
  __syncthreads();

  const float sl2 = p.scale * 1.4426950408889634f;
  bf16_t* Og = (bf16_t*)p.O + (int64_t)win * p.o_sb + (int64_t)head * p.o_sh;
#pragma unroll 1
  for (int sub = 0; sub < 2; ++sub) {
    // S^T[key, q] for all 256 keys: 8 key tiles x NG k-steps; the K fragments of tile kt+1 are requested before the MFMAs of
    // tile kt issue (the compiler otherwise hoists all 8 x NG reads: 160 registers it does not have)
    f32x16_t s[8];
    u32x4_t ka[2][NG];
    auto load_k = [&](int kt, int buf) {
      const char* krow = Ks + (kt * 32 + l31) * RS + h * 16;
#pragma unroll
      for (int g = 0; g < NG; ++g) ka[buf][g] = *(const u32x4_t*)(krow + g * 32);
    };
    load_k(0, 0);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt + 1 < 8) load_k(kt + 1, (kt + 1) & 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka[kt & 1][g]), __builtin_bit_cast(bf16x8_t, qf[g]), s[kt], 0, 0, 0);
      asm volatile("" ::: "memory");
    }
    if (sub == 0) load_q(1);                     // the next pass's Q rows arrive under this pass's softmax and PV
    // softmax over the lane's 128 scores + the other half-wave's (one query per lane pair), single pass
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = xor32_max(mx);
    const float mo = mx * sl2;
    float rs = 0.f;
    uint32_t pb[8][8];                            // P^T as packed bf16 pairs: the PV MFMA's B operands
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sl2, -mo));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[kt][r + 1], sl2, -mo));
        rs += p0 + p1;
        pb[kt][r >> 1] = f2bf2(p0, p1);
      }
    rs = xor32_sum(rs);
    // O^T[d, q] = V^T P^T: per 32-key tile two 16-key MFMA steps per d-tile (operand layout of pv_step_bf16t)
    f32x16_t o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const char* vt = Vs + (kt >> 1) * D * 128;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const u32x4_t b = {pb[kt][st * 4], pb[kt][st * 4 + 1], pb[kt][st * 4 + 2], pb[kt][st * 4 + 3]};
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int d = dt * 32 + l31;
          const u32x4_t a = *(const u32x4_t*)(vt + d * 128 + ((((kt & 1) * 4 + st * 2 + h) ^ ((d >> 1) & 7)) << 4));
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), o[dt], 0, 0, 0);
        }
      }
      asm volatile("" ::: "memory");
    }
    const float inv = 1.0f / rs;
    bf16_t* orow = Og + (int64_t)(wave * 64 + sub * 32 + l31) * p.o_ss;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = dt * 32 + r4 * 8 + h * 4;          // accumulator registers 4 r4 .. 4 r4 + 3 = four consecutive d
        if (d0 < D) {
          uint2 v;
          v.x = f2bf2(o[dt][4 * r4] * inv, o[dt][4 * r4 + 1] * inv);
          v.y = f2bf2(o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv);
          *(uint2*)(orow + d0) = v;
        }
      }
  }
}

template <int D>
static int launch_win256(const WinAttnArgs& p, hipStream_t st) {
  constexpr int CH = D / 8, RS = (CH | 1) * 16;
  constexpr int lds = 4 * D * 128 + 256 * RS + 16;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)win256_attn_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  win256_attn_kernel<D><<<dim3(p.H, p.Bw), 256, lds, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Hiera's SMALL windows (stage 1 / 4: 8x8 = 64 tokens, stage 2: 4x4 = 16 tokens, and the q-pooled first block of a stage:
// 16 queries x 64 keys, 4 x 16): ONE WAVE per (window, head), 16x16x32 MFMA tiles, no workgroup-level staging of Q / K.
//   * Q and K rows go straight from global memory into the MFMA operand layout (lane = row l&15, 16-byte chunk l>>4 of a
//     32-wide k-step): a window's rows are contiguous in the fused projection, neighbouring waves of a workgroup take
//     neighbouring heads, so whole lines are consumed together;
//   * S^T = K Q^T leaves a lane with 4 keys of ONE query: the softmax needs two xor-shuffles (lanes 16 / 32 apart);
//   * P feeds the PV MFMA from the registers it was made in: contraction slot 8g + j of the 32-deep step is key 4g + j of
//     the 16-key tile for j < 4 and an explicit zero for j >= 4 (half of the step is padding: MFMA time is nothing here);
//     the matching V^T operand is gathered from a wave-private row image of V in LDS (four 2-byte reads per operand);
//   * these launches are bound by the fused projection's bytes (453 MB per 8-frame launch at stage 1): the generic kernel
//     ran them at 1.1-1.5 TB/s (one padded 128-query tile per few windows, two barriers per KV tile).
struct TinyAttnArgs {
  const void* Q; const void* K; const void* V; void* O;
  int Bw, H;
  int64_t q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh;
  float scale;
};
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

template <int D, int WQ, int WK>
__global__ __launch_bounds__(256) void tiny_win_attn_kernel(TinyAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = D / 8;                 // 16-byte chunks per row
  constexpr int NKS = (D + 31) / 32;        // 32-deep k-steps of QK^T
  constexpr int NDT = (D + 15) / 16;        // 16-wide d tiles of the output
  constexpr int QT = (WQ + 15) / 16, KT = WK / 16;
  constexpr int RSV = CH * 16;              // V row image: [key][D] bf16
  constexpr int VBYTES = WK * RSV + 64;     // + slack: d-tiles past D read (finite) bytes after the last row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
  const int64_t items = (int64_t)p.Bw * p.H;
  int64_t item = (int64_t)blockIdx.x * 4 + wave;
  const bool live = item < items;
  if (!live) item = items - 1;              // (every wave reaches the barrier; a surplus wave recomputes the last item, stores nothing)
  const int win = (int)(item / p.H), head = (int)(item % p.H);
  const bf16_t* Qg = (const bf16_t*)p.Q + (int64_t)win * p.q_sb + (int64_t)head * p.q_sh;
  const bf16_t* Kg = (const bf16_t*)p.K + (int64_t)win * p.k_sb + (int64_t)head * p.k_sh;
  const bf16_t* Vg = (const bf16_t*)p.V + (int64_t)win * p.v_sb + (int64_t)head * p.v_sh;
  char* Vw = smem + wave * VBYTES;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  u32x4_t qf[QT][NKS], kf[KT][NKS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int c = ks * 4 + g, row = qt * 16 + i16;
      qf[qt][ks] = (c < CH && row < WQ) ? *(const u32x4_t*)(Qg + (int64_t)row * p.q_ss + c * 8) : zero4;
    }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int c = ks * 4 + g;
      kf[kt][ks] = c < CH ? *(const u32x4_t*)(Kg + (int64_t)(kt * 16 + i16) * p.k_ss + c * 8) : zero4;
    }
  constexpr int NVI = (WK * CH + 63) / 64;
#pragma unroll
  for (int i = 0; i < NVI; ++i) {
    const int idx = lane + i * 64, row = idx / CH, c = idx - row * CH;
    if (idx < WK * CH) *(u32x4_t*)(Vw + row * RSV + c * 16) = *(const u32x4_t*)(Vg + (int64_t)row * p.v_ss + c * 8);
  }
  if (lane < 16) *(uint32_t*)(Vw + WK * RSV + lane * 4) = 0u;
  __syncthreads();

  // S^T[key, q]: lane (q = i16 of tile qt, g) holds keys kt*16 + 4g + r
  f32x4v_t s[KT][QT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      s[kt][qt] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[kt][ks]), __builtin_bit_cast(bf16x8_t, qf[qt][ks]),
                                                            s[kt][qt], 0, 0, 0);
    }
  const float sl2 = p.scale * 1.4426950408889634f;
  uint32_t pb[KT][QT][2];
  float inv[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
    mx = xor32_max(xor16_max(mx));
    const float mo = mx * sl2;
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], sl2, -mo)); rs += e[r]; }
      pb[kt][qt][0] = f2bf2(e[0], e[1]);
      pb[kt][qt][1] = f2bf2(e[2], e[3]);
    }
    rs = xor32_sum(xor16_sum(rs));
    inv[qt] = 1.0f / rs;
  }
  // O^T[d, q] = V^T P^T per 16-key tile; the contraction's slots 8g + 4 .. 8g + 7 are zeros on the P side
  f32x4v_t o[NDT][QT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) o[dt][qt] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const char* vp = Vw + (kt * 16 + 4 * g) * RSV + (dt * 16 + i16) * 2;
      const uint32_t v0 = *(const uint16_t*)(vp), v1 = *(const uint16_t*)(vp + RSV), v2 = *(const uint16_t*)(vp + 2 * RSV),
                     v3 = *(const uint16_t*)(vp + 3 * RSV);
      const u32x4_t a = {v0 | (v1 << 16), v2 | (v3 << 16), 0u, 0u};
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const u32x4_t b = {pb[kt][qt][0], pb[kt][qt][1], 0u, 0u};
        o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), o[dt][qt], 0, 0, 0);
      }
    }
  if (!live) return;
  bf16_t* Og = (bf16_t*)p.O + (int64_t)win * p.o_sb + (int64_t)head * p.o_sh;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qt * 16 + i16;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d0 = dt * 16 + 4 * g;
      if (q < WQ && d0 < D) {
        uint2 v;
        v.x = f2bf2(o[dt][qt][0] * inv[qt], o[dt][qt][1] * inv[qt]);
        v.y = f2bf2(o[dt][qt][2] * inv[qt], o[dt][qt][3] * inv[qt]);
        *(uint2*)(Og + (int64_t)q * p.o_ss + d0) = v;
      }
    }
  }
}

template <int D, int WQ, int WK>
static int launch_tiny_win(const TinyAttnArgs& p, hipStream_t st) {
  constexpr int lds = 4 * (WK * (D / 8) * 16 + 64);
  const int64_t items = (int64_t)p.Bw * p.H;
  tiny_win_attn_kernel<D, WQ, WK><<<dim3((unsigned)((items + 3) / 4)), 256, lds, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_window_attention(const void* Q, const void* K, const void* V, void* O, int Bw, int H, int wq, int wtok, int D,
                                   int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                                   int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                                   float scale, int dtype, vg_stream_t stream) {
  VG_CHECK(Q && K && V && O, VG_ERR_ARG, "vg_window_attention: null pointer");
  VG_CHECK(dtype == VG_BF16, VG_ERR_UNSUPPORTED, "vg_window_attention: bf16 only (fp32 windows go through vg_attention)");
  const bool big = wq == 256 && wtok == 256 && (D == 64 || D == 72 || D == 80);
  const bool tiny = D == 72 && ((wq == 16 && wtok == 16) || (wq == 64 && wtok == 64) || (wq == 4 && wtok == 16) || (wq == 16 && wtok == 64));
  VG_CHECK(big || tiny, VG_ERR_UNSUPPORTED,
           "vg_window_attention: %d queries x %d keys per window with head_dim %d unsupported (256 x 256 with head_dim 64 / 72 / 80; "
           "16 x 16, 64 x 64, 4 x 16, 16 x 64 with head_dim 72)", wq, wtok, D);
  VG_CHECK(Bw > 0 && (Bw <= 65535 || tiny) && H > 0 && scale > 0.f, VG_ERR_ARG, "vg_window_attention: bad shape Bw=%d H=%d", Bw, H);
  VG_CHECK(q_ss % 8 == 0 && q_sh % 8 == 0 && q_sb % 8 == 0 && k_ss % 8 == 0 && k_sh % 8 == 0 && k_sb % 8 == 0 && v_ss % 8 == 0 &&
               v_sh % 8 == 0 && v_sb % 8 == 0 && o_ss % 4 == 0 && o_sh % 4 == 0 && o_sb % 4 == 0,
           VG_ERR_ARG, "vg_window_attention: strides must keep 16-byte (q/k/v) / 8-byte (o) alignment");
  VG_CHECK((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0 && (((uintptr_t)O) & 7) == 0, VG_ERR_ARG,
           "vg_window_attention: q/k/v must be 16-byte aligned, o 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (tiny) {
    TinyAttnArgs t{Q, K, V, O, Bw, H, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh, scale};
    if (wq == 16 && wtok == 16) return launch_tiny_win<72, 16, 16>(t, st);
    if (wq == 64) return launch_tiny_win<72, 64, 64>(t, st);
    if (wq == 4) return launch_tiny_win<72, 4, 16>(t, st);
    return launch_tiny_win<72, 16, 64>(t, st);
  }
  WinAttnArgs p{Q, K, V, O, Bw, H, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh, scale};
  if (D == 72) return launch_win256<72>(p, st);
  if (D == 64) return launch_win256<64>(p, st);
  return launch_win256<80>(p, st);
}

// merge the split-KV partials: O = sum_s O_s e^{m_s - m} / sum_s l_s e^{m_s - m}.  One workgroup per (q, head, b):
// the per-split weights are formed once in LDS, then the D columns are accumulated with independent (pipelined) loads.
template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs p) {
  __shared__ float wgt[64];
  __shared__ float s_l;
  const int q = blockIdx.x, head = blockIdx.y, b = blockIdx.z, D = p.DV, ns = p.nsplit;      // (rows of the VALUE dim)
  const float* base = p.part + (((int64_t)b * p.Hq + head) * ns * p.Sq + q) * (D + 2);
  const int64_t sstride = (int64_t)p.Sq * (D + 2);
  if (threadIdx.x < 64) {
    const int s = threadIdx.x;
    const float ms = s < ns ? base[s * sstride + D] : -INFINITY;
    const float ls = s < ns ? base[s * sstride + D + 1] : 0.f;
    const float m = wave_max(ms);
    const float msafe = (m == -INFINITY) ? 0.f : m;
    const float w = __expf(ms - msafe);
    wgt[s] = w;
    const float l = wave_sum(ls * w);
    if (s == 0) s_l = l;
  }
  __syncthreads();
  const float inv = s_l > 0.f ? 1.0f / s_l : 0.f;
  T* Og = (T*)p.O + (int64_t)b * p.o_sb + (int64_t)head * p.o_sh + (int64_t)q * p.o_ss;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < ns; ++s) acc += base[s * sstride + d] * wgt[s];
    vg_elt<T>::st(Og + d, acc * inv);
  }
}

template <typename T, int BKV, int NW>
static int dispatch_dp(const AttnArgs& p, hipStream_t st) {
  const int D = p.D;
  if (D <= 32) return launch_attn<T, 32, BKV, NW>(p, st);
  if (D <= 64) return launch_attn<T, 64, BKV, NW>(p, st);
  if (D <= 96) return launch_attn<T, 96, BKV, NW>(p, st);
  if (D <= 128) return launch_attn<T, 128, BKV, NW>(p, st);
  return launch_attn<T, 256, BKV, NW>(p, st);
}

static int attention_impl(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv,
                          int Sq, int Skv, int D, int DV, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                          int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss,
                          int64_t v_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float scale,
                          int causal, int dtype, float* workspace, int64_t ws_floats, int nsplit,
                          const int* skv_dev, vg_stream_t stream) {
  // (bf16 only: the fp32 instantiation <float, 256, 32, 2, 1, 64> mis-scored ONE key — row 27 of a 32-key tile — whenever the last tile held 28..31
  // keys (tools/lab/dv_debug2.py: one-hot values; the <.., 256> value width of the same source is exact) and was not worth an ISA-level hunt: the
  // fp32 parity mode zero-pads the values to the key width on the host instead (ops.attention_dv), same arithmetic on the exact-fp32 kernel)
  VG_CHECK(DV == D || (DV == 64 && D > 128 && Hq == Hkv && causal == 0 && dtype == VG_BF16), VG_ERR_UNSUPPORTED,
           "vg_attention_dv: a value head dim other than the query's is built for bf16, D in (128, 256], DV = 64, no GQA, no mask (D=%d DV=%d dtype=%d)", D, DV, dtype);
  VG_CHECK(nsplit <= 64, VG_ERR_ARG, "vg_attention_splitkv: nsplit must be <= 64");
  VG_CHECK(Q && K && V && O, VG_ERR_ARG, "vg_attention: null pointer");
  VG_CHECK(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Sq >= 0 && Skv > 0, VG_ERR_ARG,
           "vg_attention: bad shape B=%d Hq=%d Hkv=%d Sq=%d Skv=%d", B, Hq, Hkv, Sq, Skv);
  VG_CHECK(D > 0 && D <= 256 && D % 8 == 0, VG_ERR_ARG, "vg_attention: head dim %d unsupported (multiple of 8, <= 256)", D);
  VG_CHECK(dtype == VG_F32 || dtype == VG_BF16, VG_ERR_ARG, "vg_attention: bad dtype %d", dtype);
  VG_CHECK(scale > 0.f, VG_ERR_ARG, "vg_attention: scale %g must be positive (the running maximum is taken over the raw scores)", (double)scale);
  const int kpc = dtype == VG_BF16 ? 8 : 4;
  VG_CHECK(q_ss % kpc == 0 && q_sh % kpc == 0 && q_sb % kpc == 0 && k_ss % kpc == 0 && k_sh % kpc == 0 &&
               k_sb % kpc == 0 && v_ss % kpc == 0 && v_sh % kpc == 0 && v_sb % kpc == 0,
           VG_ERR_ARG, "vg_attention: q/k/v strides must keep 16-byte alignment");
  VG_CHECK((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0, VG_ERR_ARG, "vg_attention: q/k/v must be 16-byte aligned");
  if (Sq == 0) return VG_OK;
  VG_CHECK(causal >= 0 || (Sq == Skv && nsplit == 1 && Hq == Hkv), VG_ERR_ARG,
           "vg_attention: the block-diagonal window mask (causal = -tokens_per_window) needs Sq == Skv, no KV split, no GQA");
  VG_CHECK(nsplit >= 1, VG_ERR_ARG, "vg_attention: nsplit must be >= 1");
  int split_len = 0;
  if (nsplit > 1) {
    split_len = (((Skv + nsplit - 1) / nsplit) + 63) / 64 * 64;   // whole KV tiles per split
    VG_CHECK(workspace && ws_floats >= (int64_t)B * Hq * nsplit * Sq * (DV + 2), VG_ERR_ARG,
             "vg_attention_splitkv: workspace too small (need B*Hq*nsplit*Sq*(DV+2) floats)");
  }
  AttnArgs p{Q, K, V, O, B, Hq, Hkv, Sq, Skv, D, causal, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh,
             v_sb, v_ss, v_sh, o_sb, o_ss, o_sh, scale, nsplit, split_len, 0, workspace, skv_dev, DV, 0};
  // fold the G query heads of a KV head into one query tile when they all fit (decode: G*Sq = 4 rows): K/V staged
  // once per KV head instead of once per query head
  const int tile_rows = dtype == VG_BF16 ? 128 : 64;
  if (Hq > Hkv && (Hq / Hkv) * Sq <= tile_rows) p.fold = 1;
  static const int xcd_on = getenv("VG_ATTN_XCD") ? atoi(getenv("VG_ATTN_XCD")) : 1;
  p.xcd = xcd_on && !p.fold && ((int64_t)Hq * B) % 8 == 0;
  hipStream_t st = (hipStream_t)stream;
  // (measured: a 1-wave workgroup for <= 32 query rows is SLOWER — 50 vs 31 us per decode launch — because the
  // K/V tile staging, not the MFMA work, dominates a few-row block and 64 threads stage 4x slower than 256)
  // long bf16 sequences (LLM prefill, Hiera's global blocks): 8 waves = 256 query rows share every staged K/V tile
  // measured r02: +5 % on Hiera's global blocks (4096^2, d = 72), -6 % on the causal LLM prefill (coarser diagonal), -12 % at S = 1025
  constexpr int nw8 = 1;
  // head dim 256 (SAM2 memory attention): key-split waves, two per SIMD
  constexpr int ks2 = 1;
  int rc;
  // r06: the LDS-DMA-staged kernel for the long bf16 sequences (vg_attention_dma.hip); VG_ATTN_DMA=0: attn_kernel everywhere (A/B knob)
  static const int dma_on = getenv("VG_ATTN_DMA") ? atoi(getenv("VG_ATTN_DMA")) : 1;
  if (dma_on && dtype == VG_BF16 && attn_dma_eligible(p)) {
    rc = attn_dma_launch(p, st);
  } else if (DV != D && dma_on && attn_dma_dv_eligible(p)) {
    rc = attn_dma_dv_launch(p, st);
  } else if (DV != D) {
    rc = launch_attn<bf16_t, 256, 64, 8, 2, 64>(p, st);
  } else if (dtype == VG_BF16 && ks2 && !p.fold && D > 128) {
    rc = launch_attn<bf16_t, 256, 64, 8, 2>(p, st);
  } else if (dtype == VG_BF16 && nw8 && !p.fold && nsplit == 1 && Sq >= 2048 && D > 64 && D <= 128 && causal == 0) {
    rc = D <= 96 ? launch_attn<bf16_t, 96, 64, 8>(p, st) : launch_attn<bf16_t, 128, 64, 8>(p, st);
  } else {
    rc = (dtype == VG_BF16) ? dispatch_dp<bf16_t, 64, 4>(p, st) : dispatch_dp<float, 32, 2>(p, st);
  }
  if (rc != VG_OK || nsplit == 1) return rc;
  dim3 grid(Sq, Hq, B);
  if (dtype == VG_BF16) attn_combine_kernel<bf16_t><<<grid, 256, 0, st>>>(p);
  else attn_combine_kernel<float><<<grid, 256, 0, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_attention_splitkv(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv,
                                    int Sq, int Skv, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                                    int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss,
                                    int64_t v_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float scale,
                                    int causal, int dtype, float* workspace, int64_t ws_floats, int nsplit,
                                    const int* skv_dev, vg_stream_t stream) {
  return attention_impl(Q, K, V, O, B, Hq, Hkv, Sq, Skv, D, D, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh, scale, causal,
                        dtype, workspace, ws_floats, nsplit, skv_dev, stream);
}

// Attention whose values (and output) live in DV dims while queries and keys have D: SAM2's memory cross-attention with the v-projection moved
// behind the attention (R/modeling/sam/transformer.py:289-327 computes softmax(q k^T) (M Wv^T + b); rows of the softmax sum to one, so that is
// (softmax(q k^T) M) Wv^T + b).  V / O rows hold DV elements; the split-KV workspace holds B*Hq*nsplit*Sq*(DV+2) floats.
extern "C" int vg_attention_dv(const void* Q, const void* K, const void* V, void* O, int B, int H, int Sq, int Skv, int D, int DV,
                               int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                               int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float scale,
                               int dtype, float* workspace, int64_t ws_floats, int nsplit, vg_stream_t stream) {
  VG_CHECK(DV > 0 && DV % 8 == 0, VG_ERR_ARG, "vg_attention_dv: bad DV %d", DV);
  return attention_impl(Q, K, V, O, B, H, H, Sq, Skv, D, DV, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh, scale, 0,
                        dtype, workspace, ws_floats, nsplit ? nsplit : 1, nullptr, stream);
}

extern "C" int vg_attention(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv,
                            int Sq, int Skv, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                            int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss,
                            int64_t v_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float scale,
                            int causal, int dtype, vg_stream_t stream) {
  return vg_attention_splitkv(Q, K, V, O, B, Hq, Hkv, Sq, Skv, D, q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh,
                              o_sb, o_ss, o_sh, scale, causal, dtype, nullptr, 0, 1, nullptr, stream);
}
