// Shared device/host helpers for libmdance_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define MD_OK 0
#define MD_ERR_ARG (-1)
#define MD_ERR_LAUNCH (-2)

void md_set_error(const char* fmt, ...);

#define MD_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      md_set_error(__VA_ARGS__);           \
      return MD_ERR_ARG;                   \
    }                                      \
  } while (0)

#define MD_CHECK_LAUNCH(name)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      md_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return MD_ERR_LAUNCH;                                                       \
    }                                                                             \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// One-time, PER-DEVICE opt-in to more than 64 KiB of dynamic LDS for kernel `Kernel` (hipFuncSetAttribute is a per-device
// property of the function), safe to call from any number of host threads: a bit per device in an atomic mask; two threads
// racing on the same device both set the same value, which is harmless.  `Kernel` is a template VALUE parameter so that every
// kernel instantiation owns its own mask.
#include <atomic>
template <auto Kernel>
static inline void md_ensure_dynamic_lds(int bytes) {
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}

// Tuning knobs are read from the environment once per process (function-local statics: thread-safe initialisation).
static inline int md_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
