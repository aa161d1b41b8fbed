// lab: semantics of ds_read_b64_tr_b16 (gfx950) — which LDS elements lane l receives when lane i of a 16-lane group supplies the address of the
// 8-byte piece (row i >> 2, column quad i & 3) of a 4 x 16 bf16 block.  hipcc --offload-arch=gfx950 tools/lab/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const short* in, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (i >> 2) * 64 + g * 16 + (i & 3) * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (short)i;
  short *d, *e;
  hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, e);
  hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) { printf(" (r%d,c%2d)", o[l * 4 + j] / 64, o[l * 4 + j] % 64); ok &= o[l * 4 + j] == j * 64 + (l >> 4) * 16 + (l & 15); }
    printf("\n");
  }
  printf("model 'lane l gets rows 0..3 of column 16 (l >> 4) + (l & 15)': %s\n", ok ? "HOLDS" : "does NOT hold");
  return 0;
}
