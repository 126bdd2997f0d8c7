// Sustained MFMA rate / shader clock under the chip's power management, by operand data and instruction shape (gfx950).
// Not part of the product: a measurement that prices what a GEMM main loop can reach on THIS workload's operands
// (fp16, N(0,1) activations), so that "MFMA duty" and "clock" can be told apart.
//   mode 0: v_mfma_f32_32x32x16_f16, operands in registers (4 A x 4 B fragments, 16 accumulators), 1 wave / SIMD
//   mode 1: v_mfma_f32_16x16x32_f16, same
//   mode 2: v_mfma_f32_32x32x16_bf16, same (bit patterns reinterpreted)
//   mode 3: mode 0 + fragments re-read from LDS every step: 8 ds_read_b128 per 16 MFMAs (128x128 per-wave tile ratio)
//   mode 4: mode 0 with 2 A x 5 B: 7 ds_read_b128 per 10 MFMAs (the 64x160 per-wave tile of gemm_pp_kernel)
//   mode 5: mode 0, 2 waves / SIMD (8 accumulators each)
// Output per run: TFLOP/s, shader clock from s_memtime / s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(MODE == 5 ? 512 : 256) void mfma_kernel(const half_t* __restrict__ src, float* __restrict__ out, int iters,
                                                                      unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int NA = (MODE == 4) ? 2 : (MODE == 5 ? 2 : 4);
  constexpr int NB = (MODE == 4) ? 5 : 4;
  half8_t a[NA], b[NB];
  const half8_t* s8 = reinterpret_cast<const half8_t*>(src) + (size_t)(blockIdx.x * 8 + wave) * 64 * 16;
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = s8[(i * 64 + lane)];
#pragma unroll
  for (int j = 0; j < NB; ++j) b[j] = s8[((8 + j) * 64 + lane)];
  if (MODE == 3 || MODE == 4) {
    // per-wave private LDS image: fragment f at f*1024 + lane*16 (conflict-free ds_read_b128)
    half8_t* l8 = reinterpret_cast<half8_t*>(smem) + wave * 16 * 64;
#pragma unroll
    for (int i = 0; i < NA; ++i) l8[i * 64 + lane] = a[i];
#pragma unroll
    for (int j = 0; j < NB; ++j) l8[(8 + j) * 64 + lane] = b[j];
    __syncthreads();
  }
  unsigned long long t0 = 0, r0 = 0;
  if (lane == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float acc_sum = 0.f;
  if constexpr (MODE == 1) {
    floatx4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = floatx4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_sum += acc[i][j][0] + acc[i][j][3];
  } else {
    floatx16 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mm = [&](half8_t (&aa)[NA], half8_t (&bb)[NB]) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if constexpr (MODE == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, aa[i]), __builtin_bit_cast(bf16x8_t, bb[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa[i], bb[j], acc[i][j], 0, 0, 0);
          }
        }
    };
    if constexpr (MODE == 3 || MODE == 4) {
      // software pipelined: the fragments of step s+1 are read (into the other register set) before the MFMAs of step s; the
      // pointer is laundered so that the reads stay in the loop
      half8_t a2[NA], b2[NB];
      auto rd = [&](half8_t (&aa)[NA], half8_t (&bb)[NB]) {
        unsigned off = wave * 16 * 1024;
        asm volatile("" : "+v"(off));
        const half8_t* p = reinterpret_cast<const half8_t*>(smem + off);
#pragma unroll
        for (int i = 0; i < NA; ++i) aa[i] = p[i * 64 + lane];
#pragma unroll
        for (int j = 0; j < NB; ++j) bb[j] = p[(8 + j) * 64 + lane];
      };
      auto interleave = [&]() {
        // one fragment read behind each of the first NA+NB MFMAs of the step, the rest of the MFMAs back to back
#pragma unroll
        for (int k = 0; k < NA + NB; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NA * NB - (NA + NB), 0);
      };
      for (int it = 0; it < iters; it += 2) {
        rd(a2, b2);
        mm(a, b);
        interleave();
        rd(a, b);
        mm(a2, b2);
        interleave();
      }
    } else {
      for (int it = 0; it < iters; ++it) mm(a, b);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_sum += acc[i][j][0] + acc[i][j][15];
  }
  if (lane == 0) {
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (wave == 0) {
      clk[blockIdx.x * 2] = t1 - t0;
      clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc_sum;
}

template <int MODE>
static void run(const char* name, const half_t* src, float* out, unsigned long long* clk, int iters, int ncu, double flop_per_iter_wave, const char* data) {
  const int threads = MODE == 5 ? 512 : 256;
  const size_t smem = 8 * 16 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 0, last = 0;
  double ghz = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_kernel<MODE>, dim3(ncu), dim3(threads), smem, 0, src, out, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    last = ms;
    if (rep == 0 || ms < best) best = ms;
    std::vector<unsigned long long> h(ncu * 2);
    CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double c = 0, r = 0;
    for (int i = 0; i < ncu; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    ghz = c / r * 0.1;
  }
  const double waves = (double)ncu * threads / 64;
  const double tf = flop_per_iter_wave * iters * waves / (last * 1e-3) / 1e12;
  printf("%-58s data=%-6s  %8.2f ms  %8.1f TFLOP/s  shader clock %.3f GHz (s_memtime / s_memrealtime)\n", name, data, last, tf, ghz);
  fflush(stdout);
}

int main(int argc, char** argv) {
  int dev = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, ncu, prop.clockRate);
  const size_t n = (size_t)ncu * 8 * 64 * 16 * 8;     // halves
  std::vector<half_t> h(n);
  half_t* src; float* out; unsigned long long* clk;
  CHECK(hipMalloc(&src, n * 2));
  CHECK(hipMalloc(&out, (size_t)ncu * 512 * 4));
  CHECK(hipMalloc(&clk, (size_t)ncu * 16));
  const int iters = argc > 1 ? atoi(argv[1]) : 60000;
  for (int pass = 0; pass < 3; ++pass) {
    const char* data = pass == 0 ? "randn" : (pass == 1 ? "zeros" : "unif");
    srand(1234);
    for (size_t i = 0; i < n; ++i) {
      float v = 0.f;
      if (pass == 0) {  // Box-Muller N(0,1)
        const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
        v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
      } else if (pass == 2) {
        v = 2.f * rand() / (float)RAND_MAX - 1.f;
      }
      h[i] = (half_t)v;
    }
    CHECK(hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice));
    run<0>("f16 32x32x16  regs  16 acc  1 wave/SIMD", src, out, clk, iters, ncu, 16 * 2.0 * 32 * 32 * 16, data);
    run<1>("f16 16x16x32  regs  16 acc x4  1 wave/SIMD", src, out, clk, iters / 2, ncu, 64 * 2.0 * 16 * 16 * 32, data);
    run<2>("bf16 32x32x16 regs  16 acc  1 wave/SIMD", src, out, clk, iters, ncu, 16 * 2.0 * 32 * 32 * 16, data);
    run<3>("f16 32x32x16  + 8 ds_read_b128 / 16 MFMA", src, out, clk, iters, ncu, 16 * 2.0 * 32 * 32 * 16, data);
    run<4>("f16 32x32x16  + 7 ds_read_b128 / 10 MFMA", src, out, clk, iters, ncu, 10 * 2.0 * 32 * 32 * 16, data);
    run<5>("f16 32x32x16  regs  8 acc  2 waves/SIMD", src, out, clk, iters, ncu, 8 * 2.0 * 32 * 32 * 16, data);
  }
  return 0;
}
