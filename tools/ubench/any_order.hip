// Does hipExtAnyOrderLaunch let two INDEPENDENT kernels of one stream overlap on gfx950 (hip_ext.h says "not supported on GFX9xx")?
// Two launches of 384 one-workgroup-per-CU blocks (1.5 rounds of 256 CUs each, every block spins T cycles): back to back in order they
// cost 4 rounds, overlapped 3.  hipcc --offload-arch=gfx950 tools/ubench/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(unsigned long long cycles, unsigned* sink) {
  extern __shared__ char lds[];                       // 100 KB: one workgroup per CU
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned x = threadIdx.x;
  while (__builtin_readcyclecounter() - t0 < cycles) x = x * 1664525u + 1013904223u;
  if (x == 12345u) sink[0] = x + lds[threadIdx.x];
}
int main() {
  unsigned* sink; hipMalloc(&sink, 4);
  hipStream_t st; hipStreamCreate(&st);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned long long T = 200000;                // ~100 us at 2 GHz (s_memtime ticks at 100 MHz on some parts: the ratio is what matters)
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      for (int it = 0; it < 10; ++it) {
        hipLaunchKernelGGL(spin, dim3(384), dim3(256), 100 * 1024, st, T, sink);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(384), dim3(256), 100 * 1024, st, T, sink);
        else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(384), dim3(256), 100 * 1024, st, nullptr, nullptr, hipExtAnyOrderLaunch, T, sink);
        else { /* one launch of 768 blocks: the ideal */ hipLaunchKernelGGL(spin, dim3(384), dim3(256), 100 * 1024, st, T, sink); }
      }
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s rep %d: %.3f ms per pair\n", mode == 0 ? "in order      " : (mode == 1 ? "any-order 2nd " : "(pair of identical in-order launches again)"), rep, ms / 10);
    }
  }
  hipEventRecord(e0, st);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(spin, dim3(768), dim3(256), 100 * 1024, st, T, sink);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("one launch of 768 blocks (3 rounds): %.3f ms\n", ms / 10);
  hipEventRecord(e0, st);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 100 * 1024, st, T, sink);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("one launch of 256 blocks (1 round): %.3f ms\n", ms / 10);
  return 0;
}
