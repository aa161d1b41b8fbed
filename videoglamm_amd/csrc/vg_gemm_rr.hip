// Row-register GEMM (round 5): C = act(A W^T + bias) [+ R] for SHORT K (144 / 288 — Hiera stages 1 and 2, the FPN's lateral convs of those levels:
// R/modeling/backbones/hieradet.py:37-168, image_encoder.py:101-133) over very many rows (M = 1 M / 262 K per 16-frame chunk).
//
// Why another kernel.  The tile kernels walk such a problem as tens of thousands of 128x128 (or 256x192) tiles whose K loop is 3-9 steps long: the
// pipeline of a tile never fills, its epilogue is as long as its MFMAs, and every step exposes a memory latency — the 64-byte-step kernel and the
// phase-split kernels land on the same 2.2-3.5x of the shapes' HBM roof (tools/lab/notes: shortk_p8_probe.log).  Here nothing of A goes through LDS
// and no tile has a prologue:
//   * a workgroup (8 waves) owns panels of 256 rows, a wave 32 of them; the wave's A rows — ALL K columns — sit in its registers as the MFMA's
//     b-operand fragments (K / 16 x 4 VGPRs), loaded straight from global memory: lane (row, half) reads 16 bytes at k = 16 j + 8 half.  The
//     window gather of Hiera's partition (wmode 1) is just another row address; the next panel's rows are requested behind the last MFMA of the
//     current one;
//   * W is the a-operand: 64-row chunks in LDS (row stride K * 2 + 16 bytes: conflict-free ds_read_b128).  RESIDENT: all of W fits (K = 144, N <= 512)
//     and is staged once per workgroup — the panel loop then has NO barrier; STREAM: two chunk slots, chunk c + 1 is fetched into registers before
//     and written to LDS behind the MFMAs of chunk c, one LDS-only barrier per chunk (W comes from L2: every workgroup of an XCD streams the same chunks);
//   * D^T = W A^T: a lane owns ONE output row and 16 of a block's 32 columns; bias (from LDS) / GELU / residual in registers, v_permlane32_swap turns the
//     (4 + 4)-column pieces of the two lane halves into 16-byte row pieces, a wave-private 4 KB slab turns those into whole 128-byte row segments
//     (eight lanes per row, eight rows per store instruction; streaming stores for big outputs, vg_gemm_common.h);
//   * optional LayerNorm over K of the rows as they sit in registers (vg_gemm_ln: norm1 -> q|k|v, norm2 -> fc1).
// HBM traffic = A once + C once (+ R); per workgroup and panel the LDS carries one W read per wave.
// Measured (tools/lab/rr_bench.py, us, this kernel | the tile kernels): stage-1 q|k|v 400 | 610, stage-1 -> 2 projection 210 | 369, FPN level 0 165 | 262 (5.1 TB/s:
// the practical HBM rate of a read + write stream), stage-2 q|k|v 180 | 218, stage-2 projection + residual 128 | 162, stage-2 fc1 + GELU 321 | 331 (VALU-bound
// on the exact-erf GELU: 105 us of it).  Ablations: without its stores the stage-1 q|k|v runs 156 us (W resident) — the first version's scattered 16-byte stores
// cost 220 us; bias loads in the epilogue (global, behind the stores in the vmcnt queue) cost 6 %.  Tried and dropped: the second wave of each SIMD running its
// epilogue one chunk late (beside its partner's MFMAs): no gain, as on the attention kernel.
#include "vg_gemm_common.h"

namespace {

template <int KS, bool RESIDENT, int ACT, bool RES, bool LNP>
__global__ __launch_bounds__(512, 1) void gemm_rr_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char rr_smem[];
  constexpr int K = KS * 16, RSW = K * 2 + 16, SLOT = 64 * RSW, PPR = K / 8;   // 16-byte pieces per W row
  constexpr int NPF = (64 * PPR + 511) / 512;                                   // W pieces per thread and chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int nch = (p.N + 63) >> 6;
  const bf16_t* Wg = (const bf16_t*)p.W;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  u32x4_t wreg[NPF];
  auto wfetch = [&](int c) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int q = tid + i * 512, row = q / PPR, pc = q - row * PPR, n = c * 64 + row;
      wreg[i] = (q < 64 * PPR && n < p.N) ? *(const u32x4_t*)(Wg + (int64_t)n * p.ldw + pc * 8) : zero4;
    }
  };
  auto wstage = [&](char* slot) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int q = tid + i * 512, row = q / PPR, pc = q - row * PPR;
      if (q < 64 * PPR) *(u32x4_t*)(slot + row * RSW + pc * 16) = wreg[i];
    }
  };

  // bias in LDS: a global load in the epilogue would sit behind the previous block's stores in the in-order vmcnt queue — every 32-column block then
  // waits out a full store round trip (measured: 25 us per 256-row panel instead of 6)
  float* bias_s = (float*)rr_smem;
  const int nbias = nch * 64;
  float* ln_s = bias_s + nbias;                            // LNP: gamma[K] | beta[K]
  constexpr int LNB = LNP ? 2 * K * 4 : 0;
  char* slab = rr_smem + nbias * 4 + LNB + wave * 4096;   // wave-private: 32 rows x 64 bf16 of the chunk being stored (16-byte pieces XOR-swizzled by row)
  char* slots = rr_smem + nbias * 4 + LNB + 8 * 4096;
  for (int i = tid; i < nbias; i += 512) bias_s[i] = (p.bias && i < p.N) ? p.bias[i] : 0.f;
  if constexpr (LNP) {
    for (int i = tid; i < K; i += 512) { ln_s[i] = p.ln_w ? p.ln_w[i] : 1.f; ln_s[K + i] = p.ln_b ? p.ln_b[i] : 0.f; }
  }
  const int npanel = (p.M + 255) >> 8;
  const int mine = ((int)blockIdx.x < npanel) ? (npanel - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // panels blockIdx.x, + gridDim.x, ...
  if (mine == 0) return;

  if constexpr (RESIDENT) {
    for (int c = 0; c < nch; ++c) {
      wfetch(c);
      wstage(slots + c * SLOT);
    }
    __syncthreads();
  } else {
    wfetch(0);
    wstage(slots);
    __syncthreads();
  }

  u32x4_t areg[KS];
  bool avalid = false, nvalid = false;      // LNP: the lane's row holds data (a padding row of the window gather stays zero BEHIND the LayerNorm, as F.pad does)
  // LayerNorm of the wave's 32 rows in place: lane (row, half) holds K / 2 of the row's values, its partner lane the rest; two-pass fp32 statistics and the
  // bf16-rounded output of vg_layernorm (norm_short_kernel: same expressions)
  auto ln_apply = [&](u32x4_t (&a)[KS], bool valid) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += __uint_as_float(a[j][e] << 16) + __uint_as_float(a[j][e] & 0xffff0000u);
    sum = xor32_sum(sum);
    const float invK = 1.0f / (float)K, mean = sum * invK;
    // (the packed rows are re-unpacked in every pass: left to itself the compiler keeps all K / 2 unpacked values of the lane alive across the three passes —
    //  72 / 144 more registers, 150-330 of them spilled)
    auto opaque = [&]() {
#pragma unroll
      for (int j = 0; j < KS; ++j) asm volatile("" : "+v"(a[j][0]), "+v"(a[j][1]), "+v"(a[j][2]), "+v"(a[j][3]));
    };
    opaque();
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d0 = __uint_as_float(a[j][e] << 16) - mean, d1 = __uint_as_float(a[j][e] & 0xffff0000u) - mean;
        q += d0 * d0 + d1 * d1;
      }
    q = xor32_sum(q);
    const float rstd = rsqrtf(q * invK + p.ln_eps);
    opaque();
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      asm volatile("" ::: "memory");          // gamma / beta are read here, per panel: hoisted out of the panel loop they are 2 K / 2 registers per lane (spills)
      const int k = j * 16 + h * 8;
      const f32x4_t w0 = *(const f32x4_t*)(ln_s + k), w1 = *(const f32x4_t*)(ln_s + k + 4);
      const f32x4_t b0 = *(const f32x4_t*)(ln_s + K + k), b1 = *(const f32x4_t*)(ln_s + K + k + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float n0 = (__uint_as_float(a[j][e] << 16) - mean) * rstd, n1 = (__uint_as_float(a[j][e] & 0xffff0000u) - mean) * rstd;
        const float g0 = e < 2 ? w0[2 * e] : w1[2 * e - 4], g1 = e < 2 ? w0[2 * e + 1] : w1[2 * e - 3];
        const float c0 = e < 2 ? b0[2 * e] : b1[2 * e - 4], c1 = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
        a[j][e] = valid ? f2bf2(n0 * g0 + c0, n1 * g1 + c1) : 0u;
      }
    }
  };
  int64_t orow = -1;            // output (and residual) row of this lane's GEMM row; -1: nothing to store
  int64_t orow_s[4];            // output rows of the store layout: pass s writes rows 8 s + lane / 8 of the wave's 32, eight lanes per row
  auto aload = [&](int panel) {
    const int m = panel * 256 + wave * 32 + l31;
    int64_t src = -1;
    if (m < p.M) src = p.wmode == 1 ? gemm_window_row(p, m) : (int64_t)m;
    const bf16_t* ap = (const bf16_t*)p.A + (src >= 0 ? src : 0) * p.lda + h * 8;
#pragma unroll
    for (int j = 0; j < KS; ++j) areg[j] = src >= 0 ? *(const u32x4_t*)(ap + j * 16) : zero4;
    avalid = src >= 0;
  };
  auto out_row = [&](int panel) {
    const int m = panel * 256 + wave * 32 + l31;
    orow = -1;
    if (m < p.M) orow = p.wmode == 2 ? gemm_window_row(p, m) : (int64_t)m;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int ms = panel * 256 + wave * 32 + 8 * s4 + (lane >> 3);
      orow_s[s4] = -1;
      if (ms < p.M) orow_s[s4] = p.wmode == 2 ? gemm_window_row(p, ms) : (int64_t)ms;
    }
  };

  // one 32-column block of the lane's row: registers r -> column nb0 + 8 (r >> 2) + 4 h + (r & 3)
  // residual pieces of a block in the accumulator layout (4 columns = 8 bytes per group), requested BEFORE the chunk's MFMAs (same queue argument as the bias)
  auto rload = [&](int nb0, uint2 (&rv)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = nb0 + 8 * g + 4 * h;
      const uint2 z = {0u, 0u};
      rv[g] = (n + 4 <= p.N && orow >= 0) ? *(const uint2*)((const bf16_t*)p.R + orow * p.ldr + n) : z;
    }
  };
  auto epilogue = [&](const f32x16_t& acc, int nb0, const uint2 (&rv)[4]) {
    uint32_t pk[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = nb0 + 8 * g + 4 * h;
      float v[4];
      const f32x4_t b = *(const f32x4_t*)(bias_s + n);         // (columns past N: zeros)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = vg_act(acc[4 * g + e] + b[e], ACT);
      if constexpr (RES) {
        const uint2 r = rv[g];
        v[0] += __uint_as_float(r.x << 16);
        v[1] += __uint_as_float(r.x & 0xffff0000u);
        v[2] += __uint_as_float(r.y << 16);
        v[3] += __uint_as_float(r.y & 0xffff0000u);
      }
      pk[2 * g] = f2bf2(v[0], v[1]);
      pk[2 * g + 1] = f2bf2(v[2], v[3]);
    }
    // groups (0, 1) and (2, 3): the low half keeps its own group a and takes the partner's group a (columns +4..+7), the high half gets both group-b pieces
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      u32x4_t o;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const auto s = __builtin_amdgcn_permlane32_swap(pk[4 * t + w], pk[4 * t + 2 + w], false, false);
        o[w] = s[0];
        o[2 + w] = s[1];
      }
      const int q = ((nb0 >> 3) & 7) + 2 * t + h;          // 16-byte piece of the row's 128-byte chunk segment (nb0 is the chunk's column 0 or 32)
      *(u32x4_t*)(slab + l31 * 128 + ((q ^ (l31 & 7)) << 4)) = o;
    }
  };
  // the chunk's 64 columns leave as whole 128-byte row segments: eight lanes per row, eight rows per store instruction.  (First version: every lane stored
  // its own two 16-byte pieces per block — 32 rows x 32 bytes per instruction: 378 us on Hiera's stage-1 q|k|v of which 220 us were the stores.)
  auto flush = [&](int n0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int qq = lane & 7, n = n0 + 8 * qq;
    u32x4_t d[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int row = 8 * s4 + (lane >> 3);
      d[s4] = *(const u32x4_t*)(slab + row * 128 + ((qq ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (orow_s[s4] >= 0 && n + 8 <= p.N) epi_store16((bf16_t*)p.C + orow_s[s4] * p.ldc + n, d[s4], p.nt);
    }
    asm volatile("" ::: "memory");
  };

  // K = 144, W resident: the next panel's rows have their own registers and are requested at the TOP of the current panel — by the time they are needed every store that
  // was issued before them has long been acknowledged (vmcnt retires in order: a load behind fresh stores waits out their round trip).  K = 288 has no room
  // for a second set (72 registers): its rows are requested behind the last MFMA of the panel.
  constexpr bool ADB = KS <= 9 && RESIDENT;       // (the streamed form at K = 144 has the W-chunk registers on top: a second row set spills)
  u32x4_t anext[ADB ? KS : 1];
  auto aload_next = [&](int panel) {
    const int m = panel * 256 + wave * 32 + l31;
    int64_t src = -1;
    if (m < p.M) src = p.wmode == 1 ? gemm_window_row(p, m) : (int64_t)m;
    const bf16_t* ap = (const bf16_t*)p.A + (src >= 0 ? src : 0) * p.lda + h * 8;
#pragma unroll
    for (int j = 0; j < (ADB ? KS : 1); ++j) anext[j] = src >= 0 ? *(const u32x4_t*)(ap + j * 16) : zero4;
    nvalid = src >= 0;
  };

  // one 64-column chunk of the panel: MFMAs, then (STREAM) the next chunk's W goes to its slot and the one after is requested, then the epilogue
  auto chunk = [&](const char* slot, int c, int next_panel, char* stage_to, int fetch_c) {
    const int n0 = c * 64;
    const bool two = n0 + 32 < p.N;             // (wave-uniform) the chunk's second 32-column block holds columns
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const char* w0 = slot + l31 * RSW + h * 16;
    const char* w1 = w0 + 32 * RSW;
    uint2 r0[4], r1[4];
    if constexpr (RES) {
      rload(n0, r0);
      if (two) rload(n0 + 32, r1);
    }
    u32x4_t wa[2], wb[2];
    wa[0] = *(const u32x4_t*)w0;
    wb[0] = two ? *(const u32x4_t*)w1 : zero4;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int cur = j & 1, nxt = cur ^ 1;
      if (j + 1 < KS) {
        wa[nxt] = *(const u32x4_t*)(w0 + (j + 1) * 32);
        wb[nxt] = two ? *(const u32x4_t*)(w1 + (j + 1) * 32) : zero4;
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wa[cur]), __builtin_bit_cast(bf16x8_t, areg[j]), acc0, 0, 0, 0);
      if (two) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wb[cur]), __builtin_bit_cast(bf16x8_t, areg[j]), acc1, 0, 0, 0);
    }
    if constexpr (!RESIDENT) {
      if (stage_to) wstage(stage_to);           // chunk g + 1 (requested one step ago, behind that step's MFMAs) -> the slot step g - 1 read
      if (fetch_c >= 0) wfetch(fetch_c);        // chunk g + 2
    }
    if constexpr (!ADB) {
      if (next_panel >= 0) aload(next_panel);   // the panel's last chunk: the next panel's rows are requested behind the last MFMA that reads these
    }
    epilogue(acc0, n0, r0);
    if (two) epilogue(acc1, n0 + 32, r1);
    flush(n0);
  };

  aload((int)blockIdx.x);
  const int total = mine * nch;
  if constexpr (!RESIDENT) {
    if (total > 1) wfetch(nch > 1 ? 1 : 0);     // chunk of step 1
  }
  for (int it = 0; it < mine; ++it) {
    const int panel = (int)blockIdx.x + it * (int)gridDim.x;
    const int nextp = it + 1 < mine ? panel + (int)gridDim.x : -1;
    out_row(panel);
    if constexpr (LNP) ln_apply(areg, avalid);
    if constexpr (ADB) {
      if (nextp >= 0) aload_next(nextp);
    }
    for (int c = 0; c < nch; ++c) {
      const int np = (c == nch - 1) ? nextp : -1;
      if constexpr (RESIDENT) {
        chunk(slots + c * SLOT, c, np, nullptr, -1);
      } else {
        const int g = it * nch + c;
        chunk(slots + (g & 1) * SLOT, c, np, g + 1 < total ? slots + ((g + 1) & 1) * SLOT : nullptr, g + 2 < total ? (c + 2) % nch : -1);
        vg_lds_barrier();
      }
    }
    if constexpr (ADB) {
      if (nextp >= 0) {
#pragma unroll
        for (int j = 0; j < KS; ++j) areg[j] = anext[j];
        avalid = nvalid;
      }
    }
  }
}

template <int KS, bool RESIDENT>
static int rr_launch_v(const GemmArgs& q, int wgs, size_t lds, hipStream_t st) {
#define VG_RR_GO(A, R, L)                                                                                                                       \
  do {                                                                                                                                          \
    static bool attr = false;                                                                                                                   \
    if (!attr) {                                                                                                                                \
      (void)hipFuncSetAttribute((const void*)gemm_rr_kernel<KS, RESIDENT, A, R, L>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
      attr = true;                                                                                                                              \
    }                                                                                                                                           \
    gemm_rr_kernel<KS, RESIDENT, A, R, L><<<wgs, 512, lds, st>>>(q);                                                                            \
  } while (0)
  const bool ln = q.ln_w || q.ln_b;
  if (ln) {
    if (q.act == VG_ACT_GELU) VG_RR_GO(VG_ACT_GELU, false, true);
    else VG_RR_GO(VG_ACT_NONE, false, true);
  } else if (q.R) VG_RR_GO(VG_ACT_NONE, true, false);
  else if (q.act == VG_ACT_GELU) VG_RR_GO(VG_ACT_GELU, false, false);
  else VG_RR_GO(VG_ACT_NONE, false, false);
#undef VG_RR_GO
  VG_LAUNCH_CHECK();
  return VG_OK;
}

}  // namespace

// K in {144, 288}, bf16 in and out, N a multiple of 16, 16-byte aligned C / R rows, act none | GELU (GELU without a residual), no LayerScale / fp8 scales / batch
bool vg_gemm_rr_eligible(const GemmArgs& p, int batch, bool out_is_bf16) {
  return (p.K == 144 || p.K == 288) && out_is_bf16 && batch == 1 && !p.a_op && !p.sa && !p.gamma && p.vec_out && p.N % 16 == 0 && p.N >= 16 &&
         (p.act == VG_ACT_NONE || (p.act == VG_ACT_GELU && !p.R)) && p.ksplit <= 1 && !((p.ln_w || p.ln_b) && p.R) &&
         // the streamed form's LDS: bias per 64-column chunk + gamma | beta + the epilogue slabs + two weight slots (vg_gemm_rr_launch) within 158 KB
         (size_t)((p.N + 63) / 64) * 256 + 8 * (size_t)p.K + 8 * 4096 + 2 * 64 * (size_t)(p.K * 2 + 16) <= 158 * 1024;
}

int vg_gemm_rr_launch(const GemmArgs& q, int ncu, hipStream_t st) {
  const int npanel = (q.M + 255) / 256, wgs = npanel < ncu ? npanel : ncu, nch = (q.N + 63) / 64;
  const size_t slot = 64 * (size_t)(q.K * 2 + 16), fixed = (size_t)nch * 256 + ((q.ln_w || q.ln_b) ? 8 * (size_t)q.K : 0) + 8 * 4096;   // bias | gamma, beta | slabs
  if (q.K == 144) {
    if (fixed + nch * slot <= 158 * 1024) return rr_launch_v<9, true>(q, wgs, fixed + nch * slot, st);
    return rr_launch_v<9, false>(q, wgs, fixed + 2 * slot, st);
  }
  return rr_launch_v<18, false>(q, wgs, fixed + 2 * slot, st);
}
