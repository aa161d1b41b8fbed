// Decode step, round 6: the attention of the one new token as a WAVE-PRIVATE flash pass, and the step's bookkeeping on the device.
//
// vg_decode_attention2 (bf16, head_dim 128): q arrives rotated and the new key / value rows are already in the cache (vg_decode_qkv_rope), so a
// wave starts straight on its loads: 32 keys x (K row + V row) = 16 x 16 bytes per lane, every one requested before anything is waited for,
// addresses independent of the position word (rows past the position are masked, never skipped: the caches are zero-initialised, so a masked
// row is finite).  Lane (r, c) = (lane / 16, lane % 16) holds the 16-byte chunk c of rows 4 i + r: a load instruction covers four whole 256-byte
// rows.  q.k = v_dot2c_f32_bf16 on the packed operands + a 4-step DPP sum over the 16 lanes of a row; softmax over the wave's 32 keys in
// registers; p.v accumulates per lane — no LDS and no barrier until the 4 / 8 waves of the workgroup (128 / 256 keys) merge their
// (max, sum, acc) through LDS, ONE barrier.  The split's partial leaves in 16-byte write-through stores; the last workgroup of a KV head to
// arrive (agent-scope ticket, as in vg_decode_attention) merges <= 16 partials per fabric round trip with 16-byte loads.
// r05 timeline of the kernel this replaces: RoPE phase -> K/V fetch -> q.k (LDS) -> softmax (LDS) -> p.v (LDS) -> publish (4-byte stores)
// -> ticket -> merge (4-byte loads of 27 partials): 18.8 us per layer for 14 MB of K / V.
//
// vg_decode_advance: what the host did between two steps — apply a forced token, record the emitted and the raw token, bump the position —
// so that step k + 1 can be enqueued before token k has been read back; and the current position's cos / sin row for vg_decode_qkv_rope.
#include "vg_common.h"

template <int CTRL> __device__ __forceinline__ float dec_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dec_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dec_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dec_dpp<0x141>(v);    // row_half_mirror
  v += dec_dpp<0x140>(v);    // row_mirror
  return v;
}
// 8 bf16 products into one fp32 sum.  Inline asm, not __builtin_amdgcn_fdot2_f32_bf16: hipcc (ROCm 7.2) folds bit_cast<bf16x2>(v[e]) of a 4 x u32 vector to
// element 0 for every e (the ISA showed one global_load_dword and four identical v_dot2c).  The trailing s_nop covers what the compiler cannot see: the
// result is next read by a DPP move (VALU write -> DPP read: 2 wait states; the compiler itself pads 3 after its own v_dot2c).
__device__ __forceinline__ float dot8_bf16(const u32x4_t& a, const u32x4_t& b, float acc) {
  asm("v_dot2c_f32_bf16 %0, %1, %5\n\tv_dot2c_f32_bf16 %0, %2, %6\n\tv_dot2c_f32_bf16 %0, %3, %7\n\tv_dot2c_f32_bf16 %0, %4, %8\n\ts_nop 2"
      : "+v"(acc)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  return acc;
}

struct DecAttn2Args {
  const bf16_t* q; const bf16_t* kc; const bf16_t* vc; bf16_t* o;
  float* ws; int* cnt; const int* pos_dev;
  int Hkv, max_len, nsplit, window; float scale; int64_t ws_bytes;
};

constexpr int A2_D = 128, A2_PS = 132;      // partial row: 128 accumulators, max, sum, 2 pad (16-byte rows)

template <int G, int NW>
__global__ __launch_bounds__(NW * 64) void decode_attn2_kernel(DecAttn2Args p) {
  extern __shared__ __attribute__((aligned(16))) char a2_smem[];
  __shared__ int ticket;
  constexpr int KW = NW * 32;                // keys per workgroup
  float* red = (float*)a2_smem;              // [NW][4 r][G][128]
  float* ml = red + NW * 4 * G * A2_D;       // [NW][G][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane >> 4, c = lane & 15;
  const int s = blockIdx.x, kvh = blockIdx.y;
  const int j0 = s * KW + wave * 32;

  // ---- 1. every load up front; none of the addresses needs the position
  u32x4_t qv[G], kreg[8], vreg[8];
#pragma unroll
  for (int g = 0; g < G; ++g) qv[g] = *(const u32x4_t*)(p.q + (int64_t)(kvh * G + g) * A2_D + c * 8);
  const int64_t rs = (int64_t)p.Hkv * A2_D;
  const bf16_t* kb = p.kc + (int64_t)kvh * A2_D + c * 8;
  const bf16_t* vb = p.vc + (int64_t)kvh * A2_D + c * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) kreg[i] = *(const u32x4_t*)(kb + (int64_t)min(j0 + 4 * i + r, p.max_len - 1) * rs);
#pragma unroll
  for (int i = 0; i < 8; ++i) vreg[i] = *(const u32x4_t*)(vb + (int64_t)min(j0 + 4 * i + r, p.max_len - 1) * rs);
  const int pos = *p.pos_dev;
  const int lo = p.window > 0 ? max(0, pos + 1 - p.window) : 0;
  const int active = pos / KW + 1, first = lo / KW;
  if (s >= active || s < first) return;
  const int nact = active - first;

  // ---- 2. scores of the wave's 32 keys x G heads, softmax in registers
  float sc[G][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g) sc[g][i] = row16_sum(dot8_bf16(kreg[i], qv[g], 0.f)) * p.scale;
  float mw[G], lw[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = j0 + 4 * i + r;
      if (j > pos || j < lo) sc[g][i] = -INFINITY;
      mx = fmaxf(mx, sc[g][i]);
    }
    mx = xor32_max(xor16_max(mx));      // across the wave's four 16-lane rows on the VALU (__shfl_xor is an LDS-pipe round trip per step)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e = mx == -INFINITY ? 0.f : __expf(sc[g][i] - mx);      // a wave without a visible key contributes nothing
      sc[g][i] = e;
      sum += e;
    }
    sum = xor32_sum(xor16_sum(sum));
    mw[g] = mx;
    lw[g] = sum;
  }
  // ---- 3. p.v per lane: chunk c of rows 4 i + r
  float acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float vf[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { vf[2 * e] = __uint_as_float(vreg[i][e] << 16); vf[2 * e + 1] = __uint_as_float(vreg[i][e] & 0xffff0000u); }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(sc[g][i], vf[e], acc[g][e]);
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    f32x4_t* dst = (f32x4_t*)(red + ((wave * 4 + r) * G + g) * A2_D + c * 8);
    dst[0] = f32x4_t{acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
    dst[1] = f32x4_t{acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
  }
  if (lane == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) { ml[(wave * G + g) * 2] = mw[g]; ml[(wave * G + g) * 2 + 1] = lw[g]; }
  }
  __syncthreads();
  // ---- 4. the workgroup's partial: thread -> (head g, 4 columns); 16-byte WRITE-THROUGH (sc1) stores through a buffer descriptor, every
  //         storing wave drains, then ONE lane takes the ticket (MI355X_MICROARCH.md, valid hand-off forms; the reader uses sc1 loads)
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)min(p.ws_bytes, (int64_t)0x7fffffff), 0x00020000);
  const int pbase = (int)((((int64_t)kvh * p.nsplit + s) * G) * A2_PS * 4);
  if (tid < G * 32) {
    const int g = tid >> 5, d4 = tid & 31;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, ml[(w * G + g) * 2]);
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float mwv = ml[(w * G + g) * 2];
      const float f = mwv == -INFINITY ? 0.f : __expf(mwv - M);
      L = fmaf(f, ml[(w * G + g) * 2 + 1], L);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4_t v = *(const f32x4_t*)(red + ((w * 4 + rr) * G + g) * A2_D + d4 * 4);
        o += v * f;
      }
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rsrc, pbase + (g * A2_PS + d4 * 4) * 4, 0, 16);
    if (d4 == 0) {
      const f32x4_t t = {M, L, 0.f, 0.f};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t), rsrc, pbase + (g * A2_PS + A2_D) * 4, 0, 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) ticket = __hip_atomic_fetch_add(&p.cnt[kvh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != nact - 1) return;
  // ---- 5. merge by the last workgroup of this KV head to arrive.  ALL its threads load: thread group q (128 threads = one (head, 4 columns) map)
  //         takes the partials q, q + NG, ... — 8 per thread and pass: up to 32 (NW = 8) / 16 (NW = 4) splits at G <= 4 are requested in ONE
  //         fabric round trip (sc1 loads) — and the groups' (max, sum, acc) meet in LDS (the arrays of step 3 are free again)
  {
    constexpr int GS = G * 32 <= 128 ? 128 : G * 32;      // threads of one (head, 4 columns) map
    constexpr int NG = NW * 64 / GS;
    const int grp = tid / GS, t7 = tid % GS, g = (t7 >> 5) % G, d4 = t7 & 31;
    const bool live = t7 < G * 32;
    const int mbase = (int)((((int64_t)kvh * p.nsplit + first) * G) * A2_PS * 4);
    float M = -INFINITY, L = 0.f;
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < nact; b0 += 8 * NG) {
      f32x4_t pv[8], pm[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int off = mbase + ((min(b0 + u * NG + grp, nact - 1) * G + g) * A2_PS) * 4;
        pv[u] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + d4 * 16, 0, 16));
        pm[u] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + A2_D * 4, 0, 16));
      }
      float Mn = M;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b0 + u * NG + grp < nact) Mn = fmaxf(Mn, pm[u][0]);
      const float fo = M == -INFINITY ? 0.f : __expf(M - Mn);
      o *= fo;
      L *= fo;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b0 + u * NG + grp < nact) {
          const float f = pm[u][0] == -INFINITY ? 0.f : __expf(pm[u][0] - Mn);
          L = fmaf(f, pm[u][1], L);
          o += pv[u] * f;
        }
      M = Mn;
    }
    float* gml = red;                       // [NG][GS][2]
    float* gacc = red + NG * GS * 2;        // [NG][GS][4]
    if (live) {
      gml[(grp * GS + t7) * 2] = M;
      gml[(grp * GS + t7) * 2 + 1] = L;
      *(f32x4_t*)(gacc + (grp * GS + t7) * 4) = o;
    }
    __syncthreads();
    if (grp == 0 && live) {
      float Mt = -INFINITY;
#pragma unroll
      for (int q = 0; q < NG; ++q) Mt = fmaxf(Mt, gml[(q * GS + t7) * 2]);
      float Lt = 0.f;
      f32x4_t ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const float mq = gml[(q * GS + t7) * 2];
        const float f = mq == -INFINITY ? 0.f : __expf(mq - Mt);
        Lt = fmaf(f, gml[(q * GS + t7) * 2 + 1], Lt);
        ot += *(const f32x4_t*)(gacc + (q * GS + t7) * 4) * f;
      }
      const float inv = 1.0f / Lt;
      uint2 ov = {f2bf2(ot[0] * inv, ot[1] * inv), f2bf2(ot[2] * inv, ot[3] * inv)};
      *(uint2*)(p.o + (int64_t)(kvh * G + g) * A2_D + d4 * 4) = ov;
    }
  }
  if (tid == 0) __hip_atomic_store(&p.cnt[kvh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G, int NW>
static int launch_attn2(const DecAttn2Args& p, hipStream_t st) {
  const size_t lds = sizeof(float) * ((size_t)NW * 4 * G * A2_D + NW * G * 2);
  static size_t cap = 64 * 1024;
  if (lds > cap) {
    (void)hipFuncSetAttribute((const void*)decode_attn2_kernel<G, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    cap = lds;
  }
  decode_attn2_kernel<G, NW><<<dim3(p.nsplit, p.Hkv), NW * 64, lds, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_decode_attention2_supported(int H, int Hkv, int D, int dtype) {
  if (H <= 0 || Hkv <= 0 || H % Hkv || D != 128 || dtype != VG_BF16) return 0;
  const int G = H / Hkv;
  return G == 1 || G == 2 || G == 4 || G == 8;
}

extern "C" int vg_decode_attention2(const void* q, const void* k_cache, const void* v_cache, void* out, int H, int Hkv, int D, int max_len,
                                    int window, float scale, const int* pos_dev, float* workspace, int64_t ws_floats, int keys_per_wg, int dtype,
                                    vg_stream_t stream) {
  VG_CHECK(q && k_cache && v_cache && out && pos_dev && workspace, VG_ERR_ARG, "vg_decode_attention2: null pointer");
  VG_CHECK(vg_decode_attention2_supported(H, Hkv, D, dtype), VG_ERR_UNSUPPORTED, "vg_decode_attention2: H=%d Hkv=%d D=%d dtype=%d not covered", H, Hkv, D, dtype);
  VG_CHECK(keys_per_wg == 128 || keys_per_wg == 256, VG_ERR_ARG, "vg_decode_attention2: keys_per_wg %d not in {128, 256}", keys_per_wg);
  VG_CHECK(max_len > 0 && window >= 0, VG_ERR_ARG, "vg_decode_attention2: bad max_len / window");
  VG_CHECK((((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out | (uintptr_t)workspace) & 15) == 0, VG_ERR_ARG,
           "vg_decode_attention2: 16-byte alignment");
  const int G = H / Hkv;
  const int nsplit = (max_len + keys_per_wg - 1) / keys_per_wg;
  const int64_t need = (int64_t)Hkv * nsplit * G * A2_PS + Hkv;
  VG_CHECK(ws_floats >= need, VG_ERR_ARG, "vg_decode_attention2: workspace %lld < %lld floats", (long long)ws_floats, (long long)need);
  DecAttn2Args p{(const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, (bf16_t*)out, workspace, (int*)(workspace + (ws_floats - Hkv)), pos_dev,
                 Hkv, max_len, nsplit, window, scale, ws_floats * 4};
  hipStream_t st = (hipStream_t)stream;
#define VG_A2(GG) return keys_per_wg == 256 ? launch_attn2<GG, 8>(p, st) : launch_attn2<GG, 4>(p, st)
  switch (G) {
    case 1: VG_A2(1);
    case 2: VG_A2(2);
    case 4: VG_A2(4);
    default: VG_A2(8);
  }
#undef VG_A2
}

// ---------------------------------------------------------------------------------------------------------------
// Bookkeeping of the decode loop on the device, one workgroup.  With tok / step given (end of a step: inc = 1; end of the prefill: inc = 0):
//   k = *step;  raw[k] = *tok;  *tok = forced[k] >= 0 ? forced[k] : *tok;  hist[k] = *tok;  *step = k + 1;  *pos += inc
// with rope_cs given:  rope_cs = [cos[*pos] | sin[*pos]]  (the row vg_decode_qkv_rope rotates with; *pos AFTER the increment).
// forced == NULL: no forcing.  hist / raw may be NULL.  tok: int64[1] (vg_argmax's output); the next step's embedding reads it.
__global__ __launch_bounds__(256) void decode_advance_kernel(int64_t* tok, int* pos, int* step, const int64_t* forced, int n_forced, int64_t* hist,
                                                             int64_t* raw, int cap, const float* cosT, const float* sinT, float* rope_cs, int hd,
                                                             int inc) {
  const int np = *pos + inc;
  if (rope_cs)
    for (int d = threadIdx.x; d < 2 * hd; d += 256) rope_cs[d] = d < hd ? cosT[(int64_t)np * hd + d] : sinT[(int64_t)np * hd + d - hd];
  if (threadIdx.x == 0) {
    if (tok && step) {
      const int k = *step;
      const int64_t t0 = *tok;
      int64_t t = t0;
      if (forced && k < n_forced && forced[k] >= 0) t = forced[k];
      if (raw && k < cap) raw[k] = t0;
      if (hist && k < cap) hist[k] = t;
      *tok = t;
      *step = k + 1;
    }
    if (inc) *pos = np;
  }
}

extern "C" int vg_decode_advance(int64_t* tok, int* pos, int* step, const int64_t* forced, int n_forced, int64_t* hist, int64_t* raw, int cap,
                                 const float* cos, const float* sin, float* rope_cs, int half_dim, int inc, vg_stream_t stream) {
  VG_CHECK(pos && (inc == 0 || inc == 1) && (!tok == !step), VG_ERR_ARG, "vg_decode_advance: bad args");
  VG_CHECK(!rope_cs || (cos && sin && half_dim > 0), VG_ERR_ARG, "vg_decode_advance: rope_cs needs the cos / sin tables");
  decode_advance_kernel<<<1, 256, 0, (hipStream_t)stream>>>(tok, pos, step, forced, n_forced, hist, raw, cap, cos, sin, rope_cs, half_dim, inc);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The head and the tail of a captured decode step as ONE small launch each (r06: per token the step carried embed, the cos / sin row, the final-norm
// row store, a memset + two argmax launches and the bookkeeping = 8 launches of ~5 us on the critical path; now 3: begin, argmax partials, end).
//   vg_decode_step_begin: x = table[*tok] (the embedding row of the token the previous step emitted) and rope_cs = [cos[*pos] | sin[*pos]].
//   vg_argmax_partial:    vg_argmax's first stage only — packed (value, ~index) keys atomicMax-ed into acc[row] (acc must be zero: step_end leaves it so).
//   vg_decode_step_end:   *tok = index decoded from acc[0] (acc[0] = 0 again); hid_all[*pos] = row; then vg_decode_advance's bookkeeping with inc = 1.
__global__ __launch_bounds__(256) void decode_step_begin_kernel(const int64_t* tok, const void* table, void* x, int D, int es, const int* pos, const float* cosT,
                                                                const float* sinT, float* rope_cs, int hd) {
  const int64_t t = *tok;
  const int nb = D * es / 16;
  const u32x4_t* src = (const u32x4_t*)((const char*)table + t * (int64_t)D * es);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) ((u32x4_t*)x)[i] = src[i];
  if (blockIdx.x == 0 && rope_cs) {
    const int np = *pos;
    for (int d = threadIdx.x; d < 2 * hd; d += 256) rope_cs[d] = d < hd ? cosT[(int64_t)np * hd + d] : sinT[(int64_t)np * hd + d - hd];
  }
}

extern "C" int vg_decode_step_begin(const int64_t* tok, const void* table, void* x, int D, int dtype, const int* pos, const float* cos, const float* sin,
                                    float* rope_cs, int half_dim, vg_stream_t stream) {
  VG_CHECK(tok && table && x && pos && D > 0 && (dtype == VG_BF16 || dtype == VG_F32), VG_ERR_ARG, "vg_decode_step_begin: bad args");
  const int es = dtype == VG_BF16 ? 2 : 4;
  VG_CHECK((D * es) % 16 == 0 && (((uintptr_t)table | (uintptr_t)x) & 15) == 0, VG_ERR_ARG, "vg_decode_step_begin: rows must be whole 16-byte chunks, 16-byte aligned");
  VG_CHECK(!rope_cs || (cos && sin && half_dim > 0), VG_ERR_ARG, "vg_decode_step_begin: rope_cs needs the cos / sin tables");
  decode_step_begin_kernel<<<(D * es / 16 + 255) / 256, 256, 0, (hipStream_t)stream>>>(tok, table, x, D, es, pos, cos, sin, rope_cs, half_dim);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

__global__ __launch_bounds__(256) void decode_step_end_kernel(unsigned long long* acc, int64_t* tok, int* pos, int* step, const int64_t* forced, int n_forced,
                                                              int64_t* hist, int64_t* raw, int cap, const void* row, void* hid_all, int D, int es) {
  const int p0 = *pos;
  const int nb = D * es / 16;
  const u32x4_t* src = (const u32x4_t*)row;
  u32x4_t* dst = (u32x4_t*)((char*)hid_all + (int64_t)p0 * D * es);
  for (int i = threadIdx.x; i < nb; i += 256) dst[i] = src[i];
  if (threadIdx.x == 0) {
    const int k = *step;
    const int64_t t0 = (int64_t)(0xffffffffu - (uint32_t)(acc[0] & 0xffffffffull));      // vg_argmax's key: (monotone value bits) << 32 | ~index
    acc[0] = 0;
    int64_t t = t0;
    if (forced && k < n_forced && forced[k] >= 0) t = forced[k];
    if (raw && k < cap) raw[k] = t0;
    if (hist && k < cap) hist[k] = t;
    *tok = t;
    *step = k + 1;
    *pos = p0 + 1;
  }
}

extern "C" int vg_decode_step_end(uint64_t* acc, int64_t* tok, int* pos, int* step, const int64_t* forced, int n_forced, int64_t* hist, int64_t* raw, int cap,
                                  const void* row, void* hid_all, int D, int dtype, vg_stream_t stream) {
  VG_CHECK(acc && tok && pos && step && row && hid_all && D > 0 && (dtype == VG_BF16 || dtype == VG_F32), VG_ERR_ARG, "vg_decode_step_end: bad args");
  const int es = dtype == VG_BF16 ? 2 : 4;
  VG_CHECK((D * es) % 16 == 0 && (((uintptr_t)row | (uintptr_t)hid_all) & 15) == 0, VG_ERR_ARG, "vg_decode_step_end: rows must be whole 16-byte chunks, 16-byte aligned");
  decode_step_end_kernel<<<1, 256, 0, (hipStream_t)stream>>>((unsigned long long*)acc, tok, pos, step, forced, n_forced, hist, raw, cap, row, hid_all, D, es);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
