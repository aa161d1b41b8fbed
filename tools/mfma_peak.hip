// MFMA ceilings of the part as it actually clocks under load.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
//  reg      : independent v_mfma_f32_32x32x16_bf16 on registers only (4 / 16 accumulators per wave)
//  reg+lds  : the GEMM inner loop without global memory: 16 accumulators (a 128x128 wave tile), the 8 operand fragments
//             of each group of 16 MFMAs re-read from LDS with ds_read_b128 interleaved between the MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int NACC>
__global__ __launch_bounds__(256, 1) void mfma_reg(float* out, int iters) {
  f32x16_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

// LDS: 2 operands x 128 rows x 64 B per wave, same swizzled layout as gemm_tile_w128_kernel
template <int READS>
__global__ __launch_bounds__(256, 1) void mfma_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 4 * 2 * 128 * 64];   // two copies: the address changes every iteration
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < (int)sizeof(smem) / 4; i += 256) ((unsigned*)smem)[i] = 0x3f803f80u;
  __syncthreads();
  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const char* sa = smem + wave * 2 * 128 * 64 + l31 * 64;
  const char* sb = sa + 128 * 64;
  const int sw = (l31 >> 2) & 3;
  u32x4_t fa[2][4], fb[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[g][i] = *(const u32x4_t*)(sa + i * 32 * 64 + (((2 * g + h) ^ sw) << 4));
      fb[g][i] = *(const u32x4_t*)(sb + i * 32 * 64 + (((2 * g + h) ^ sw) << 4));
    }
  for (int it = 0; it < iters; ++it) {
    const int flip = (it & 1) * (4 * 2 * 128 * 64);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      if (READS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[g ^ 1][i] = *(const u32x4_t*)(sa + flip + i * 32 * 64 + (((2 * (g ^ 1) + h) ^ sw) << 4));
          fb[g ^ 1][i] = *(const u32x4_t*)(sb + flip + i * 32 * 64 + (((2 * (g ^ 1) + h) ^ sw) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[g][i]), __builtin_bit_cast(bf16x8_t, fb[g][j]), acc[i][j], 0, 0, 0);
      if (READS == 1) {          // read, MFMA, MFMA, read, ...
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
      } else if (READS == 2) {   // all 8 reads first, then the 16 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      } else if (READS == 3) {   // 2 MFMAs, then read / MFMA alternating, the last 6 MFMAs without reads behind them
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
void run(const char* name, F launch, double mfma_per_wave_iter, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch(blocks, 50);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  launch(blocks, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * mfma_per_wave_iter * iters * (blocks * 4.0);
  printf("%-34s blocks %4d : %8.1f TFLOP/s  (%.2f ms)\n", name, blocks, flops / ms / 1e9, ms);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  for (int blocks : {256, 512}) {
    run("reg, 4 accumulators", [&](int b, int it) { mfma_reg<4><<<b, 256>>>(out, it * 8); }, 4 * 8, blocks);
    run("reg, 16 accumulators", [&](int b, int it) { mfma_reg<16><<<b, 256>>>(out, it * 2); }, 16 * 2, blocks);
  }
  run("reg+lds frags, no re-read", [&](int b, int it) { mfma_lds<0><<<b, 256>>>(out, it); }, 32, 256);
  run("reg+lds, 8 reads interleaved", [&](int b, int it) { mfma_lds<1><<<b, 256>>>(out, it); }, 32, 256);
  run("reg+lds, 8 reads up front", [&](int b, int it) { mfma_lds<2><<<b, 256>>>(out, it); }, 32, 256);
  run("reg+lds, reads early-interleaved", [&](int b, int it) { mfma_lds<3><<<b, 256>>>(out, it); }, 32, 256);
  return 0;
}
