// Operand layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) on gfx950:
// C[32x32] = A[32x64] * B[32x64]^T with lane l holding row l%32 of A (and of B).  Which 32 K-bytes does lane half
// h = l/32 hold?  H1: K [32h, 32h+32) contiguous.  H2: VGPR 0-3 = K [16h, 16h+16), VGPR 4-7 = K [32+16h, 32+16h+16).
//   hipcc -O3 --offload-arch=gfx950 tools/f8_mfma_layout.hip -o /tmp/f8l && /tmp/f8l
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int hyp) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  i32x8_t va, vb;
  for (int v = 0; v < 8; ++v) {
    int k0;
    if (hyp == 1) k0 = 32 * h + 4 * v;
    else k0 = (v < 4 ? 16 * h : 32 + 16 * h) + 4 * (v & 3);
    va[v] = *(const int*)(A + r * 64 + k0);
    vb[v] = *(const int*)(B + r * 64 + k0);
  }
  f32x16_t acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;     // 32x32 C layout of the other 32x32 MFMAs
    C[row * 32 + r] = acc[i];
  }
}

int main() {
  const uint8_t codes[4] = {0x00, 0x38, 0x40, 0xB8};   // 0, 1, 2, -1 in e4m3
  const float vals[4] = {0.f, 1.f, 2.f, -1.f};
  uint8_t hA[32 * 64], hB[32 * 64];
  float fA[32 * 64], fB[32 * 64], ref[32 * 32];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) { int a = rand() & 3, b = rand() & 3; hA[i] = codes[a]; fA[i] = vals[a]; hB[i] = codes[b]; fB[i] = vals[b]; }
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      float s = 0.f;
      for (int kk = 0; kk < 64; ++kk) s += fA[m * 64 + kk] * fB[n * 64 + kk];
      ref[m * 32 + n] = s;
    }
  uint8_t *dA, *dB;
  float* dC;
  (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dC, 32 * 32 * 4);
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int hyp = 1; hyp <= 2; ++hyp) {
    k<<<1, 64>>>(dA, dB, dC, hyp);
    float out[32 * 32];
    (void)hipMemcpy(out, dC, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) { bad += out[m * 32 + n] != ref[m * 32 + n]; badT += out[n * 32 + m] != ref[m * 32 + n]; }
    printf("hypothesis %d: %d mismatches (transposed C: %d)   out[0..3] = %g %g %g %g   ref = %g %g %g %g\n", hyp, bad, badT, out[0], out[1], out[2], out[3],
           ref[0], ref[1], ref[2], ref[3]);
  }
  return 0;
}
