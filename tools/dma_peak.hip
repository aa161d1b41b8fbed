// LDS-DMA (global_load_lds_dwordx4) fill rate of a CU, as the GEMM kernels use it: every wave requests 8 x 1 KB per
// step (16 rows x 64 B per instruction, row stride = ld bytes) into a 4-stage LDS ring and waits with a counted vmcnt
// so that 3 steps stay in flight.  No MFMA, no LDS reads.  Reports GB/s per CU for 4 / 8 / 16 waves per CU and for a
// small (L2-resident) and a large footprint.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_peak.hip -o /tmp/dma_peak && /tmp/dma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BARRIER>
__global__ __launch_bounds__(256) void dma_loop(const char* src, int64_t ld, int rows_total, int ksteps, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // a block owns 512 "rows" (256 of A + 256 of W in the GEMM); block b starts at row (b * 512) % rows_total
  const char* p[8];
  for (int i = 0; i < 8; ++i) {
    const int row = (int)(((int64_t)blockIdx.x * 512 + wave * 128 + i * 16 + (lane >> 2)) % rows_total);
    p[i] = src + (int64_t)row * ld + ((lane & 3) ^ ((row >> 2) & 3)) * 16;
  }
  auto issue = [&](int kt, int buf) {
    char* dst = smem + buf * 32768 + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[i] + (int64_t)(kt % ksteps) * 64),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  int buf = 0;
  for (int kt = 0; kt < iters; ++kt) {
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (BARRIER) __builtin_amdgcn_s_barrier();
    issue(kt + 3, buf == 0 ? 3 : buf - 1);
    buf = buf == 3 ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// MODE 1: the same requests as plain global_load_dwordx4 into registers (no LDS);  MODE 2: half DMA, half plain
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
template <int MODE>
__global__ __launch_bounds__(256) void vec_loop(const char* src, int64_t ld, int rows_total, int ksteps, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* p[8];
  for (int i = 0; i < 8; ++i) {
    const int row = (int)(((int64_t)blockIdx.x * 512 + wave * 128 + i * 16 + (lane >> 2)) % rows_total);
    p[i] = src + (int64_t)row * ld + ((lane & 3) ^ ((row >> 2) & 3)) * 16;
  }
  u32x4_t acc = {0, 0, 0, 0};
  constexpr int NV = MODE == 1 ? 8 : 4;
  u32x4_t r[3][NV];
  auto issue = [&](int kt, int slot, int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r[slot][i] = __builtin_nontemporal_load((const u32x4_t*)(p[i] + (int64_t)(kt % ksteps) * 64));
    if (MODE == 2) {
      char* dst = smem + buf * 32768 + wave * 8192;
#pragma unroll
      for (int i = 4; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[i] + (int64_t)(kt % ksteps) * 64),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    }
  };
  issue(0, 0, 0);
  issue(1, 1, 1);
  issue(2, 2, 2);
  int buf = 0;
  for (int kt = 0; kt + 2 < iters; kt += 3) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NV; ++i) acc ^= r[s][i];
      issue(kt + s + 3, s, buf == 0 ? 3 : buf - 1);
      buf = buf == 3 ? 0 : buf + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

// 128-byte rows: 8 lanes per row, 8 rows per instruction, 16 instructions (16 KB) per wave per step, 2-stage ring
__global__ __launch_bounds__(256) void dma_loop128(const char* src, int64_t ld, int rows_total, int ksteps, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* p[16];
  for (int i = 0; i < 16; ++i) {
    const int row = (int)(((int64_t)blockIdx.x * 512 + wave * 128 + i * 8 + (lane >> 3)) % rows_total);
    p[i] = src + (int64_t)row * ld + ((lane & 7) ^ ((row >> 1) & 7)) * 16;
  }
  auto issue = [&](int kt, int buf) {
    char* dst = smem + buf * 65536 + wave * 16384;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[i] + (int64_t)(kt % ksteps) * 128),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };
  issue(0, 0);
  for (int kt = 0; kt < iters; ++kt) {
    issue(kt + 1, (kt + 1) & 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
  const int64_t ld = 16384;                    // K = 8192 bf16
  const int rows_big = 16384;                  // 256 MB
  char* src;
  (void)hipMalloc(&src, ld * rows_big);
  (void)hipMemset(src, 1, ld * rows_big);
  (void)hipFuncSetAttribute((const void*)dma_loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipFuncSetAttribute((const void*)dma_loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 2048;
  for (int barrier = 0; barrier < 2; ++barrier)
    for (int rows : {2048, 16384})             // all blocks share 2048 rows x 16 KB = 32 MB (L2 + MALL) / 256 MB
      for (int ksteps : {16, 256})             // 16 steps x 64 B = 1 KB per row revisited (cache-resident) / the whole row
        for (int bpc : {1, 2, 4}) {
          if (bpc > 1 && 131072 * bpc > 160 * 1024) continue;
          const int blocks = 256 * bpc;
          auto launch = [&](int it) {
            if (barrier) dma_loop<1><<<blocks, 256, 131072>>>(src, ld, rows, ksteps, it);
            else dma_loop<0><<<blocks, 256, 131072>>>(src, ld, rows, ksteps, it);
          };
          launch(64);
          (void)hipDeviceSynchronize();
          (void)hipEventRecord(e0);
          launch(iters);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms = 0.f;
          (void)hipEventElapsedTime(&ms, e0, e1);
          const double bytes = 32768.0 * iters * blocks;
          printf("barrier %d rows %5d ksteps %3d blocks/CU %d : %7.1f GB/s per CU  %6.2f TB/s total  (%.0f ns per step)\n", barrier, rows, ksteps, bpc,
                 bytes / ms / 1e6 / 256, bytes / ms / 1e9, ms * 1e6 / iters);
        }
  (void)hipFuncSetAttribute((const void*)dma_loop128, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int ksteps : {8, 128}) {
    dma_loop128<<<256, 256, 131072>>>(src, ld, 16384, ksteps, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    dma_loop128<<<256, 256, 131072>>>(src, ld, 16384, ksteps, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 65536.0 * iters * 256;
    printf("LDS-DMA, 128-byte rows, ksteps %3d : %7.1f GB/s per CU  %6.2f TB/s total  (%.0f ns per 64 KB step)\n", ksteps, bytes / ms / 1e6 / 256, bytes / ms / 1e9,
           ms * 1e6 / iters);
  }
  unsigned* sink;
  (void)hipMalloc(&sink, 4);
  (void)hipFuncSetAttribute((const void*)vec_loop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int mode = 1; mode <= 2; ++mode)
    for (int ksteps : {16, 256}) {
      auto launch = [&](int it) {
        if (mode == 1) vec_loop<1><<<256, 256, 0>>>(src, ld, 16384, ksteps, it, sink);
        else vec_loop<2><<<256, 256, 131072>>>(src, ld, 16384, ksteps, it, sink);
      };
      launch(66);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      launch(iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 32768.0 * iters * 256;
      printf("%s ksteps %3d : %7.1f GB/s per CU  %6.2f TB/s total  (%.0f ns per step)\n", mode == 1 ? "plain global_load_dwordx4 x8" : "4 plain + 4 LDS-DMA        ", ksteps,
             bytes / ms / 1e6 / 256, bytes / ms / 1e9, ms * 1e6 / iters);
    }
  return 0;
}
