// What does one memory instruction cost a wave that owns its SIMD (one wave per SIMD, 16 MFMA accumulators, back-to-back
// v_mfma_f32_32x32x16_f16 = 32 cycles each) when it is placed between the MFMAs?  Not part of the product: it prices the choices of
// gemm_sp_kernel's main loop (LDS-DMA pieces vs register loads, with and without the fragment reads beside them).
//   filler 0: nothing                                   (floor: 16 MFMAs = 512 cycles per iteration)
//   filler 1: buffer_load_dwordx4 ... lds               (LDS-DMA piece, 1 KiB per wave-instruction, L2-resident source)
//   filler 2: global_load_dwordx4 -> VGPRs, lane-contiguous (1 KiB contiguous per wave-instruction)
//   filler 3: global_load_dwordx4 -> VGPRs, 32 rows x 32 bytes (lane l: row l % 32 with a 2560-byte pitch, 16-byte slot l / 32)
//   filler 4: ds_read_b128
// N fillers per iteration, evenly spaced behind MFMAs (sched_barrier pins the order); READS extra ds_read_b128 per iteration beside
// them (the sp loop carries 8 per 15 MFMAs).  Output: cycles per iteration (s_memtime) and the cost per filler over the floor.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int FILL, int N, int READS>
__global__ __launch_bounds__(256, 1) void issue_kernel(const half_t* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* __restrict__ clk) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  half8_t a[4], b[4];
  const half8_t* s8 = reinterpret_cast<const half8_t*>(src);
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = s8[i * 64 + lane]; b[i] = s8[(8 + i) * 64 + lane]; }
  half8_t* l8 = reinterpret_cast<half8_t*>(smem) + wave * 16 * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) l8[i * 64 + lane] = s8[i * 64 + lane];
  __syncthreads();
  floatx16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(src), 0, 0x80000000u, 0x00020000);
  // every wave walks its own 64-KiB window of the (L2-resident) source, 1 KiB per filler
  const unsigned base = ((blockIdx.x * 4 + wave) & 63) * 65536u;
  const unsigned voff_lin = base + lane * 16;
  const unsigned voff_row = base + (lane & 31) * 2560 + (lane >> 5) * 16;       // 32 rows x 32 bytes, 2560-byte pitch (80 KiB span)
  const char* gsrc = reinterpret_cast<const char*>(src);
  char* dma_dst = smem + 65536 + wave * 8192;
  half8_t ld[2][N > 0 ? N : 1];
  half8_t rd[READS > 0 ? READS : 1];
#pragma unroll
  for (int q = 0; q < (N > 0 ? N : 1); ++q) ld[0][q] = ld[1][q] = a[0];
  unsigned long long t0 = 0;
  if (lane == 0) t0 = __builtin_readcyclecounter();
  unsigned soff = 0;
  auto body = [&](half8_t (&cur)[N > 0 ? N : 1], half8_t (&prev)[N > 0 ? N : 1]) {
    constexpr int STRIDE = N > 0 ? 16 / N : 16;
    constexpr int RSTRIDE = READS > 0 ? 16 / READS : 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = k >> 2, j = k & 3;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (READS > 0 && k % RSTRIDE == 0 && k / RSTRIDE < READS) {
        unsigned off = wave * 16 * 1024 + (k / RSTRIDE) * 1024 + lane * 16;
        asm volatile("" : "+v"(off));
        rd[k / RSTRIDE < READS ? k / RSTRIDE : 0] = *reinterpret_cast<const half8_t*>(smem + off);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (N > 0 && k % STRIDE == (STRIDE > 1 ? 1 : 0) && k / STRIDE < N) {
        const int q = k / STRIDE < N ? k / STRIDE : 0;
        if constexpr (FILL == 1) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dma_dst + (q & 7) * 1024), 16, voff_lin, soff + q * 1024, 0, 0);
        } else if constexpr (FILL == 2) {
          cur[q] = *reinterpret_cast<const half8_t*>(gsrc + voff_lin + soff + q * 1024);
        } else if constexpr (FILL == 3) {
          cur[q] = *reinterpret_cast<const half8_t*>(gsrc + voff_row + ((soff >> 10) & 7) * 32 + q * 256);
        } else if constexpr (FILL == 4) {
          unsigned off = wave * 16 * 1024 + (8 + (q & 7)) * 1024 + lane * 16;
          asm volatile("" : "+v"(off));
          cur[q] = *reinterpret_cast<const half8_t*>(smem + off);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    soff = (soff + N * 1024) & 0xffff;
    // the loads of the PREVIOUS iteration are consumed here: one iteration stays in flight (the sp loop waits for the pieces
    // issued a K tile ago)
    if constexpr (FILL == 1 && N > 0) wait_vmcnt<N>();
    if constexpr (FILL >= 2 && N > 0) {
#pragma unroll
      for (int q = 0; q < N; ++q) asm volatile("" ::"v"(prev[q]));
    }
    if constexpr (READS > 0) {
#pragma unroll
      for (int q = 0; q < READS; ++q) asm volatile("" ::"v"(rd[q]));
    }
  };
  for (int it = 0; it < iters; it += 2) {
    body(ld[0], ld[1]);
    body(ld[1], ld[0]);
  }
  if (lane == 0 && wave == 0) clk[blockIdx.x] = __builtin_readcyclecounter() - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
#endif
}

template <int FILL, int N, int READS>
static double run(const char* name, const half_t* src, float* out, unsigned long long* clk, int iters, double floor_cyc) {
  const size_t smem = 65536 + 4 * 8192;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&issue_kernel<FILL, N, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int ncu = 256;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((issue_kernel<FILL, N, READS>), dim3(ncu), dim3(256), smem, 0, src, out, iters, clk);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(ncu);
  CHECK(hipMemcpy(h.data(), clk, ncu * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double cyc = sum / ncu / iters;
  if (floor_cyc > 0 && N > 0) printf("%-58s %8.1f cycles / 16 MFMAs   %+7.1f per filler\n", name, cyc, (cyc - floor_cyc) / N);
  else printf("%-58s %8.1f cycles / 16 MFMAs\n", name, cyc);
  return cyc;
}

int main() {
  const size_t bytes = 8u << 20;
  std::vector<half_t> h(bytes / 2);
  srand(1);
  for (auto& v : h) v = (half_t)((rand() % 2001 - 1000) / 500.0f);
  half_t* src; float* out; unsigned long long* clk;
  CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&out, 256 * 256 * 4)); CHECK(hipMalloc(&clk, 256 * 8));
  CHECK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
  const int iters = 2000;
  const double f0 = run<0, 0, 0>("MFMAs only", src, out, clk, iters, 0);
  const double f8 = run<0, 0, 8>("MFMAs + 8 ds_read_b128", src, out, clk, iters, 0);
  run<4, 4, 0>("+ 4 ds_read_b128", src, out, clk, iters, f0);
  run<1, 2, 0>("+ 2 LDS-DMA pieces", src, out, clk, iters, f0);
  run<1, 4, 0>("+ 4 LDS-DMA pieces", src, out, clk, iters, f0);
  run<1, 8, 0>("+ 8 LDS-DMA pieces", src, out, clk, iters, f0);
  run<2, 2, 0>("+ 2 global_load_dwordx4 (contiguous)", src, out, clk, iters, f0);
  run<2, 4, 0>("+ 4 global_load_dwordx4 (contiguous)", src, out, clk, iters, f0);
  run<2, 8, 0>("+ 8 global_load_dwordx4 (contiguous)", src, out, clk, iters, f0);
  run<3, 4, 0>("+ 4 global_load_dwordx4 (32 rows x 32 B)", src, out, clk, iters, f0);
  run<3, 8, 0>("+ 8 global_load_dwordx4 (32 rows x 32 B)", src, out, clk, iters, f0);
  run<1, 4, 8>("8 ds_read_b128 + 4 LDS-DMA pieces", src, out, clk, iters, f8);
  run<2, 4, 8>("8 ds_read_b128 + 4 global_load_dwordx4 (contiguous)", src, out, clk, iters, f8);
  run<2, 8, 8>("8 ds_read_b128 + 8 global_load_dwordx4 (contiguous)", src, out, clk, iters, f8);
  run<3, 8, 8>("8 ds_read_b128 + 8 global_load_dwordx4 (32 rows x 32 B)", src, out, clk, iters, f8);
  run<2, 8, 4>("4 ds_read_b128 + 8 global_load_dwordx4 (contiguous)", src, out, clk, iters, f8);
  return 0;
}
