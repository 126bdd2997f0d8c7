// diagnostic: which SIMD / CU does each wave of a 512-thread workgroup land on (HW_REG_HW_ID, gfx9 layout)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(4), dim3(512), 100 * 1024, 0, d);
  unsigned h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf("  w%d simd=%u cu=%u wave=%u", w, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15, h[b*8+w] & 15); printf("\n"); }
  return 0;
}
