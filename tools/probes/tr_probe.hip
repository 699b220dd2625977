// Probe of ds_read_b64_tr_b16 semantics on gfx950: prints, for each lane, which LDS element indices it receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int off;
  if (mode == 0) off = l * 4;                                   // consecutive 8-byte chunks
  else if (mode == 1) off = (l & 15) * 32 + (l >> 4) * 4;       // 16 lanes of a group -> 16 different rows (pitch 32), groups -> column blocks
  else off = (l & 3) * 4 + ((l >> 2) & 3) * 64 + (l >> 4) * 256; // 4x4 arrangement: lanes 0-3 along a row, then 4 rows of pitch 64
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 3; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; l++) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
