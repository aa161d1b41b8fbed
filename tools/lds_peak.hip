// LDS read throughput of a CU for the fragment-read pattern of the GEMM kernels: every lane reads 16 bytes (ds_read_b128) /
// 8 bytes (ds_read_b64) of a swizzled 128-byte-row tile, 32 rows x 2 k-halves per wave-instruction, NW waves per workgroup,
// one workgroup per CU.  No MFMA, no global memory.  Reports bytes per clock per CU (clock from wall time at the measured MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/lds_peak.hip -o /tmp/lds_peak && /tmp/lds_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// PAT: 0 = the GEMM fragment pattern (row = lane & 31, 16-byte chunk (2g + h) ^ ((row >> 1) & 7) of a 128-byte row)
//      1 = linear (lane i reads bytes [16 i, 16 i + 16) of a 1 KB block)   2 = rows without the swizzle (chunk 2g + h)
//      3 = swizzle key row & 7   4 = 64 lanes on 64 different rows, chunk g ^ (row & 7)
template <int WIDTH, int PAT>
__global__ __launch_bounds__(1024) void lds_read(int iters, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((uint32_t*)smem)[i] = i;    // 64 KB
  __syncthreads();
  int row = (wave * 32 + l31) & 255, sw = (row >> 1) & 7, hh = h;
  if (PAT == 2) sw = 0;
  if (PAT == 3) sw = row & 7;
  if (PAT == 4) { row = (wave * 64 + lane) & 255; sw = row & 7; hh = 0; }
  const char* base = PAT == 1 ? smem + wave * 4096 + lane * 16 : smem + row * 128;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += 4) {       // 16 independent reads in flight per wave before the first use
    u32x4 v[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = PAT == 1 ? 0 : (PAT == 4 ? g : 2 * g + hh);
        const int lin = PAT == 1 ? g * 1024 : 0;
        const char* a = base + lin + ((c ^ sw) << 4) + (u << 13);
        if (WIDTH == 16) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(v[u * 4 + g]) : "v"((uint32_t)(uintptr_t)a));
        } else {
          u32x2 lo, hi;
          asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"((uint32_t)(uintptr_t)a));
          asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(hi) : "v"((uint32_t)(uintptr_t)a));
          v[u * 4 + g][0] = lo[0]; v[u * 4 + g][1] = lo[1]; v[u * 4 + g][2] = hi[0]; v[u * 4 + g][3] = hi[1];
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) out[0] = 1;
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 4);
  int mhz = 0;
  hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)lds_read<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)lds_read<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)lds_read<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)lds_read<16, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)lds_read<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)lds_read<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)lds_read<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int cfg = 0; cfg < 7; ++cfg)
    for (int nw : {8, 16}) {
      const int iters = 20000;
      const int width = cfg < 5 ? 16 : 8, pat = cfg < 5 ? cfg : cfg - 5;
      auto launch = [&]() {
        switch (cfg) {
          case 0: lds_read<16, 0><<<256, nw * 64, 65536, 0>>>(iters, out); break;
          case 1: lds_read<16, 1><<<256, nw * 64, 65536, 0>>>(iters, out); break;
          case 2: lds_read<16, 2><<<256, nw * 64, 65536, 0>>>(iters, out); break;
          case 3: lds_read<16, 3><<<256, nw * 64, 65536, 0>>>(iters, out); break;
          case 4: lds_read<16, 4><<<256, nw * 64, 65536, 0>>>(iters, out); break;
          case 5: lds_read<8, 0><<<256, nw * 64, 65536, 0>>>(iters, out); break;
          default: lds_read<8, 1><<<256, nw * 64, 65536, 0>>>(iters, out); break;
        }
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)iters * 4 * 64 * 16 * nw;      // per CU
      printf("ds_read_b%-3d pattern %d %2d waves/CU: %7.1f GB/s per CU = %6.1f B/clk at %d MHz\n", width * 8, pat, nw, bytes / ms / 1e6, bytes / (ms * 1e-3) / (mhz * 1e3), mhz / 1000);
    }
  return 0;
}
