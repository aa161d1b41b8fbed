// vg_mlp3_grouped: G independent three-layer MLPs (Linear, ReLU, Linear, ReLU, Linear [, sigmoid]) on a handful of rows each, in ONE launch (r06).
//
// What it replaces: SAM2's mask decoder ends in small MLP heads on the output tokens — the four output_hypernetworks_mlps (one per mask token,
// R/modeling/sam/mask_decoder.py:232-236), iou_prediction_head and pred_obj_score_head (mask_decoder.py:239-245), and the tracker's obj_ptr_proj
// (R/modeling/sam2_base.py:425-431); MLP = R/modeling/sam2_utils.py:108-132.  On the video branch they see N <= 16 rows per frame: 21 launches of 4-5 us
// (vg_gemm's skinny route, one per layer and head).  Here a workgroup = (head g, 32 rows): the rows go to LDS once, every
// layer is D = W . X^T on the MFMA with the weight rows read straight from global memory into the a-operand (each is used once: no staging) and X^T as the
// b-operand from LDS; a lane then holds 16 output columns of ONE row — + bias, ReLU, bf16 (the rounding vg_gemm's bf16 output applies between the layers)
// and back to LDS as the next layer's rows.  Weights of the G heads are stacked and packed in fragment order (below; fp32 biases [G, out]); the heads' input
// rows are x + g * x_gs (+ row * x_rs): the token rows of the decoder's output, no gather.
#include "vg_gemm_common.h"

namespace {

struct Mlp3Args {
  const bf16_t* x; int64_t x_rs, x_gs;
  const bf16_t* w0; const float* b0; const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
  void* out; int64_t o_rs, o_gs; int out_f32;
  int R, K, Hd, No; unsigned sig_mask;
};

constexpr int M3_WAVES = 8;    // a 256-wide layer is eight 32-column tiles: one per wave
constexpr int M3_LD = 264;      // LDS row stride in bf16 (256 + 8: the 16-lane groups of a ds_read_b128 land on distinct banks)

// The kernel's time is memory latency, not bytes (a workgroup streams its head's 393 KB alone), so EVERYTHING it reads from global memory is requested in the
// prologue: the rows, the three bias vectors (to LDS) and — a 256-wide layer being eight 32-column tiles, one per wave — each wave's weight fragments of all
// three layers (3 x 16 x 16 bytes per lane = 192 registers; two waves per SIMD).  Three layers then cost one round trip, three LDS passes and two barriers.
struct Mlp3Frag { u32x4_t a[16]; };

// Weights arrive PACKED in fragment order (ops.mlp3_pack; the layout is stated in include/vg_kernels.h): [head][tile of 32 outputs][k-step of 16][lane][8 bf16], lane (l31, h) =
// W[32 tile + l31][16 step + 8 h ...] — a wave's load is one contiguous KiB.  Read row-major, a lane's 16 bytes sit 512 bytes from its neighbour's: 32 cache
// lines per load instruction, every line touched by four instructions, and the kernel spent 12 of its 14 us in the L1's tag pipe (measured r06).
__device__ __forceinline__ void mlp3_fetch(Mlp3Frag& f, const bf16_t* __restrict__ Wp, int n_in, int n_out, int j0, int lane) {
  const int nk = n_in >> 4;
  const bf16_t* w = Wp + ((int64_t)(j0 >> 5) * nk * 64 + lane) * 8;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) f.a[ks] = (ks < nk && j0 < n_out) ? *(const u32x4_t*)(w + (int64_t)ks * 512) : u32x4_t{0u, 0u, 0u, 0u};
}

// one layer: rows X [32][M3_LD] (bf16, LDS) -> Y = act(W X^T + b) for the wave's tile (columns j0 ..): LAST: to global memory, else to Xn (LDS)
template <bool LAST>
__device__ __forceinline__ void mlp3_layer(const Mlp3Args& p, const Mlp3Frag& f, const float* bias_lds, int n_in, int n_out, int j0, const bf16_t* X, bf16_t* Xn,
                                           int row0, int g, bool sigmoid, int l31, int h) {
  if (j0 >= n_out) return;
  const int nk = n_in >> 4;
  const bf16_t* xrow = X + l31 * M3_LD + h * 8;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    if (ks < nk) {
      const u32x4_t b = *(const u32x4_t*)(xrow + ks * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.a[ks]), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
  }
  // lane (l31, h): row l31, output columns j0 + mfma32_row(r, h)
  const int row = row0 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int col = j0 + mfma32_row(r, h);
    if (col >= n_out) continue;
    float v = acc[r] + bias_lds[col];
    if (LAST) {
      if (sigmoid) v = 1.0f / (1.0f + __expf(-v));
      if (row < p.R) {
        if (p.out_f32) ((float*)p.out)[(int64_t)row * p.o_rs + (int64_t)g * p.o_gs + col] = v;
        else ((bf16_t*)p.out)[(int64_t)row * p.o_rs + (int64_t)g * p.o_gs + col] = f2bf(v);
      }
    } else {
      Xn[l31 * M3_LD + col] = f2bf(fmaxf(v, 0.f));
    }
  }
}

__global__ __launch_bounds__(M3_WAVES * 64, 1) void mlp3_grouped_kernel(Mlp3Args p) {
  __shared__ __attribute__((aligned(16))) bf16_t xs[2][32 * M3_LD];
  __shared__ float bs[3][256];
  const int g = blockIdx.x, row0 = blockIdx.y * 32, tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5, j0 = (tid >> 6) * 32;
  // ---- every global read of the kernel, requested together
  const int cpr = p.K >> 3;      // 16-byte pieces per row (<= 32): 32 rows = <= 1024 pieces = <= 2 per thread
  u32x4_t xv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = tid + t * M3_WAVES * 64, r = i / cpr, c = i - r * cpr;
    xv[t] = (i < 32 * cpr && row0 + r < p.R) ? *(const u32x4_t*)(p.x + (int64_t)(row0 + r) * p.x_rs + (int64_t)g * p.x_gs + c * 8) : u32x4_t{0u, 0u, 0u, 0u};
  }
  float bv = 0.f;
  if (tid < 256) bv = tid < p.Hd ? p.b0[g * p.Hd + tid] : 0.f;
  else bv = tid - 256 < p.Hd ? p.b1[g * p.Hd + tid - 256] : 0.f;
  const float bv2 = tid < p.No ? p.b2[g * p.No + tid] : 0.f;
  Mlp3Frag f0, f1, f2;
  const int64_t th = (p.Hd + 31) >> 5, to = (p.No + 31) >> 5;      // 32-output tiles of a hidden / the output layer: a packed tile is 32 x n_in elements
  mlp3_fetch(f0, p.w0 + g * th * 32 * p.K, p.K, p.Hd, j0, lane);
  mlp3_fetch(f1, p.w1 + g * th * 32 * p.Hd, p.Hd, p.Hd, j0, lane);
  mlp3_fetch(f2, p.w2 + g * to * 32 * p.Hd, p.Hd, p.No, j0, lane);
  // ---- rows and biases to LDS
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = tid + t * M3_WAVES * 64, r = i / cpr, c = i - r * cpr;
    if (i < 32 * cpr) *(u32x4_t*)(&xs[0][r * M3_LD + c * 8]) = xv[t];
  }
  bs[tid >> 8][tid & 255] = bv;
  if (tid < 256) bs[2][tid] = bv2;
  __syncthreads();
  mlp3_layer<false>(p, f0, bs[0], p.K, p.Hd, j0, xs[0], xs[1], row0, g, false, l31, h);
  __syncthreads();
  mlp3_layer<false>(p, f1, bs[1], p.Hd, p.Hd, j0, xs[1], xs[0], row0, g, false, l31, h);
  __syncthreads();
  mlp3_layer<true>(p, f2, bs[2], p.Hd, p.No, j0, xs[0], nullptr, row0, g, (p.sig_mask >> g) & 1u, l31, h);
}

}  // namespace

extern "C" int vg_mlp3_grouped(const void* x, int64_t x_rs, int64_t x_gs, const void* w0, const float* b0, const void* w1, const float* b1, const void* w2,
                               const float* b2, void* out, int64_t o_rs, int64_t o_gs, int out_dtype, int G, int R, int K, int Hd, int No, unsigned sig_mask,
                               vg_stream_t stream) {
  VG_CHECK(x && w0 && b0 && w1 && b1 && w2 && b2 && out, VG_ERR_ARG, "vg_mlp3_grouped: null pointer");
  VG_CHECK(G >= 1 && G <= 32 && R >= 0, VG_ERR_ARG, "vg_mlp3_grouped: G = %d (1 .. 32), R = %d", G, R);
  VG_CHECK(K >= 16 && K <= 256 && K % 16 == 0 && Hd >= 16 && Hd <= 256 && Hd % 16 == 0 && No >= 1 && No <= 256, VG_ERR_UNSUPPORTED,
           "vg_mlp3_grouped: K = %d, hidden = %d must be multiples of 16 in [16, 256], outputs = %d in [1, 256]", K, Hd, No);
  VG_CHECK(out_dtype == VG_BF16 || out_dtype == VG_F32, VG_ERR_ARG, "vg_mlp3_grouped: bad output dtype %d", out_dtype);
  VG_CHECK(x_rs % 8 == 0 && x_gs % 8 == 0 && (((uintptr_t)x | (uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0, VG_ERR_ARG,
           "vg_mlp3_grouped: x / weights must be 16-byte aligned with row and group strides that are multiples of 8 elements");
  if (R == 0) return VG_OK;
  Mlp3Args p{(const bf16_t*)x, x_rs, x_gs, (const bf16_t*)w0, b0, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, out, o_rs, o_gs, out_dtype == VG_F32,
             R, K, Hd, No, sig_mask};
  mlp3_grouped_kernel<<<dim3(G, (R + 31) / 32), M3_WAVES * 64, 0, (hipStream_t)stream>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
