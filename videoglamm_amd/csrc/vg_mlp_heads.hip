// vg_mlp3_grouped: G independent three-layer MLPs (Linear, ReLU, Linear, ReLU, Linear [, sigmoid]) on a handful of rows each, in ONE launch (r06).
//
// What it replaces: SAM2's mask decoder ends in small MLP heads on the output tokens — the four output_hypernetworks_mlps (one per mask token,
// R/modeling/sam/mask_decoder.py:232-236), iou_prediction_head and pred_obj_score_head (mask_decoder.py:239-245), and the tracker's obj_ptr_proj
// (R/modeling/sam2_base.py:425-431); MLP = R/modeling/sam2_utils.py:108-132.  On the video branch they see N <= 16 rows per frame: 21 launches of 4-5 us
// (vg_gemm's skinny route, one per layer and head) whose time is the launch floor.  Here a workgroup = (head g, 32 rows): the rows go to LDS once, every
// layer is D = W . X^T on the MFMA with the weight rows read straight from global memory into the a-operand (each is used once: no staging) and X^T as the
// b-operand from LDS; a lane then holds 16 output columns of ONE row — + bias, ReLU, bf16 (the rounding vg_gemm's bf16 output applies between the layers)
// and back to LDS as the next layer's rows.  Weights of the G heads are stacked ([G, out, in], fp32 biases [G, out]); the heads' input rows are
// x + g * x_gs (+ row * x_rs): the token rows of the decoder's output, no gather.
#include "vg_gemm_common.h"

namespace {

struct Mlp3Args {
  const bf16_t* x; int64_t x_rs, x_gs;
  const bf16_t* w0; const float* b0; const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
  void* out; int64_t o_rs, o_gs; int out_f32;
  int R, K, Hd, No; unsigned sig_mask;
};

constexpr int M3_LD = 264;      // LDS row stride in bf16 (256 + 8: the 16-lane groups of a ds_read_b128 land on distinct banks)

// one layer: rows X [32][M3_LD] (bf16, LDS) -> Y = act(W X^T + b); W [n_out, n_in] row-major in global memory.  LAST: to global memory, else to Xn (LDS).
template <bool LAST>
__device__ __forceinline__ void mlp3_layer(const Mlp3Args& p, const bf16_t* __restrict__ W, const float* __restrict__ bias, int n_in, int n_out, const bf16_t* X,
                                           bf16_t* Xn, int row0, int g, bool sigmoid) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const int nk = n_in >> 4;      // MFMA k-steps of 16
  for (int j0 = wave * 32; j0 < n_out; j0 += 128) {
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wr = min(j0 + l31, n_out - 1);      // (rows past n_out re-read the last one: their outputs are not stored)
    const bf16_t* wrow = W + (int64_t)wr * n_in + h * 8;
    const bf16_t* xrow = X + l31 * M3_LD + h * 8;
    u32x4_t a = *(const u32x4_t*)wrow;
    for (int ks = 0; ks < nk; ++ks) {
      const u32x4_t an = ks + 1 < nk ? *(const u32x4_t*)(wrow + (ks + 1) * 16) : a;
      const u32x4_t b = *(const u32x4_t*)(xrow + ks * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
      a = an;
    }
    // lane (l31, h): row l31, output columns j0 + mfma32_row(r, h)
    const int row = row0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = j0 + mfma32_row(r, h);
      if (col >= n_out) continue;
      float v = acc[r] + bias[col];
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
}

__global__ __launch_bounds__(256) void mlp3_grouped_kernel(Mlp3Args p) {
  __shared__ __attribute__((aligned(16))) bf16_t xs[2][32 * M3_LD];
  const int g = blockIdx.x, row0 = blockIdx.y * 32, tid = threadIdx.x;
  // the group's 32 rows -> LDS (16-byte pieces; rows past R: zeros)
  const int cpr = p.K >> 3;
  for (int i = tid; i < 32 * cpr; i += 256) {
    const int r = i / cpr, c = i - r * cpr;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row0 + r < p.R) v = *(const u32x4_t*)(p.x + (int64_t)(row0 + r) * p.x_rs + (int64_t)g * p.x_gs + c * 8);
    *(u32x4_t*)(&xs[0][r * M3_LD + c * 8]) = v;
  }
  __syncthreads();
  mlp3_layer<false>(p, p.w0 + (int64_t)g * p.Hd * p.K, p.b0 + g * p.Hd, p.K, p.Hd, xs[0], xs[1], row0, g, false);
  __syncthreads();
  mlp3_layer<false>(p, p.w1 + (int64_t)g * p.Hd * p.Hd, p.b1 + g * p.Hd, p.Hd, p.Hd, xs[1], xs[0], row0, g, false);
  __syncthreads();
  mlp3_layer<true>(p, p.w2 + (int64_t)g * p.No * p.Hd, p.b2 + g * p.No, p.Hd, p.No, xs[0], nullptr, row0, g, (p.sig_mask >> g) & 1u);
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
  mlp3_grouped_kernel<<<dim3(G, (R + 31) / 32), 256, 0, (hipStream_t)stream>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
