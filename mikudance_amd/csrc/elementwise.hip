// Small HBM-bound kernels around the two UNets: layout packing at the API boundary, channel concat for the decoder
// skip connections, window accumulation + classifier-free guidance + DDIM step (reference
// src/pipelines/pipeline_mikudance.py:577-589, 662-678 and diffusers DDIMScheduler.step, v-prediction, eta = 0).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void md_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* md_last_error(void) { return g_err; }
extern "C" int md_version(void) { return 100; }

// ---- strided gather -> NHWC fp16 with zero channel padding and optional nearest sub-sampling --------------------------
// dst[n][y][x][c] = c < c_count ? src[(n / F)*sB + (n % F)*sF + (c_begin + c)*sC + ny(y)*sY + nx(x)*sX] : 0
// with ny(y) = min(floor(y * Hin / Ho), Hin - 1)  (PyTorch 'nearest' rule; identity when Hin == Ho)
template <typename T>
__global__ void pack_nhwc_kernel(const T* __restrict__ src, half_t* __restrict__ dst, long total, int F, long sB, long sF, long sC, long sY, long sX,
                                 int c_begin, int c_count, int Cpad, int Ho, int Wo, int Hin, int Win) {
  const float fy = (float)Hin / (float)Ho, fx = (float)Win / (float)Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cpad);
    long r = idx / Cpad;
    const int x = (int)(r % Wo);
    r /= Wo;
    const int y = (int)(r % Ho);
    const long n = r / Ho;
    float v = 0.f;
    if (c < c_count) {
      const int sy = min((int)floorf(y * fy), Hin - 1), sx = min((int)floorf(x * fx), Win - 1);
      v = (float)src[(n / F) * sB + (n % F) * sF + (long)(c_begin + c) * sC + (long)sy * sY + (long)sx * sX];
    }
    dst[idx] = (half_t)v;
  }
}

extern "C" int md_pack_nhwc_f16(const void* src, int src_is_f32, void* dst, int N, int F, long sB, long sF, long sC, long sY, long sX, int c_begin,
                                int c_count, int Cpad, int Ho, int Wo, int Hin, int Win, void* stream) {
  MD_CHECK_ARG(N > 0 && F > 0 && c_count <= Cpad && Hin >= 1 && Win >= 1, "md_pack_nhwc: bad arguments");
  const long total = (long)N * Ho * Wo * Cpad;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (src_is_f32)
    hipLaunchKernelGGL(pack_nhwc_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, (half_t*)dst, total, F, sB, sF, sC, sY, sX,
                       c_begin, c_count, Cpad, Ho, Wo, Hin, Win);
  else
    hipLaunchKernelGGL(pack_nhwc_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)src, (half_t*)dst, total, F, sB, sF, sC, sY, sX,
                       c_begin, c_count, Cpad, Ho, Wo, Hin, Win);
  MD_CHECK_LAUNCH("md_pack_nhwc");
  return MD_OK;
}

// ---- channel concat of two token-major matrices ------------------------------------------------------------------------
__global__ void concat_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, half_t* __restrict__ o, long M, int ca8, int cb8) {
  const int ct = ca8 + cb8;
  const long total = M * ct;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / ct;
    const int c = (int)(idx - m * ct);
    const half8_t v = c < ca8 ? reinterpret_cast<const half8_t*>(a)[m * ca8 + c] : reinterpret_cast<const half8_t*>(b)[m * cb8 + (c - ca8)];
    reinterpret_cast<half8_t*>(o)[idx] = v;
  }
}

extern "C" int md_concat_channels_f16(const void* a, int Ca, const void* b, int Cb, void* out, long M, void* stream) {
  MD_CHECK_ARG(Ca % 8 == 0 && Cb % 8 == 0, "md_concat_channels: channel counts must be multiples of 8");
  const long total = M * ((Ca + Cb) / 8);
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(concat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b, (half_t*)out, M, Ca / 8, Cb / 8);
  MD_CHECK_LAUNCH("md_concat_channels");
  return MD_OK;
}

// ---- window accumulate: noise_sum[half][win[i]] += pred[half*f + i], counter[win[i]] += 1 -------------------------------
// Frame slots of one window must be unique or -1 (skipped): blocks of different slots update disjoint rows without atomics.
// pred: [(2 f) HW][4] fp16 (conv_out output, NHWC with 4 channels); noise_sum: [2][Ftot][HW][4] fp32; counter [Ftot] fp32
__global__ void window_accumulate_kernel(const half_t* __restrict__ pred, float* __restrict__ noise_sum, float* __restrict__ counter,
                                         const int* __restrict__ win, int f, int Ftot, int HW4, int halves) {
  const int i = blockIdx.y;  // frame slot inside the window
  const int fr = win[i];
  if (fr < 0) return;  // an earlier duplicate of a frame named twice by this window (host marks it: last occurrence wins)
  if (blockIdx.x == 0 && threadIdx.x == 0) counter[fr] += 1.f;
  for (int h = 0; h < halves; ++h) {
    const half_t* src = pred + (size_t)(h * f + i) * HW4;
    float* dst = noise_sum + ((size_t)h * Ftot + fr) * HW4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < HW4; e += gridDim.x * blockDim.x) dst[e] += (float)src[e];
  }
}

extern "C" int md_window_accumulate(const void* pred, void* noise_sum, void* counter, const int* window, int f, int Ftot, int HW, int halves, void* stream) {
  MD_CHECK_ARG(f > 0 && Ftot >= f && (halves == 1 || halves == 2), "md_window_accumulate: bad arguments");
  hipLaunchKernelGGL(window_accumulate_kernel, dim3(cdiv(HW * 4, 256 * 4), f), dim3(256), 0, (hipStream_t)stream, (const half_t*)pred, (float*)noise_sum,
                     (float*)counter, window, f, Ftot, HW * 4, halves);
  MD_CHECK_LAUNCH("md_window_accumulate");
  return MD_OK;
}

// ---- CFG combine + DDIM v-prediction step (eta = 0) --------------------------------------------------------------------
//   v   = u + s (c - u),  u = sum_u / cnt, c = sum_c / cnt                         pipeline_mikudance.py:670-674
//   x0  = sqrt(a_t) x - sqrt(1-a_t) v ;  eps = sqrt(a_t) v + sqrt(1-a_t) x
//   x'  = sqrt(a_prev) x0 + sqrt(1-a_prev) eps                                      DDIMScheduler.step
// latents: [Ftot][HW][4] fp16, updated in place (fp32 arithmetic, one rounding).
__global__ void cfg_ddim_kernel(half_t* __restrict__ lat, const float* __restrict__ noise_sum, const float* __restrict__ counter,
                                const half_t* __restrict__ variance_noise, int Ftot, int HW4, int halves, float guidance, float sa, float sb, float sap,
                                float sdir, float sigma) {
  const long total = (long)Ftot * HW4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int fr = (int)(idx / HW4);
    // without guidance the reference takes the window SUM as it is: its division by the counter sits inside
    // `if do_classifier_free_guidance:` (src/pipelines/pipeline_mikudance.py:670-674)
    const float inv = halves == 2 ? 1.f / counter[fr] : 1.f;
    float v = noise_sum[idx] * inv;
    if (halves == 2) {
      const float c = noise_sum[total + idx] * inv;
      v = v + guidance * (c - v);
    }
    const float x = (float)lat[idx];
    const float x0 = sa * x - sb * v;
    const float ep = sa * v + sb * x;
    float out = sap * x0 + sdir * ep;                              // sdir = sqrt(1 - alpha_prev - sigma^2)
    if (variance_noise) out += sigma * (float)variance_noise[idx];  // eta > 0: + sigma_t * z
    lat[idx] = (half_t)out;
  }
}

static int cfg_ddim_launch(void* latents, const void* noise_sum, const void* counter, const void* variance_noise, int Ftot, int HW, int halves,
                           float guidance, float alpha_t, float alpha_prev, float eta, void* stream, const char* who) {
  MD_CHECK_ARG(Ftot > 0 && HW > 0 && (halves == 1 || halves == 2) && eta >= 0.f && (eta == 0.f || variance_noise), "md_cfg_ddim_step: bad arguments");
  // diffusers DDIMScheduler._get_variance: sigma_t^2 = eta^2 (1 - a_prev) / (1 - a_t) (1 - a_t / a_prev); a_t == 1 never occurs (t >= 0 of a
  // zero-terminal-SNR table has a_t < 1)
  const float var = eta > 0.f ? (1.f - alpha_prev) / (1.f - alpha_t) * (1.f - alpha_t / alpha_prev) : 0.f;
  const float sigma = eta * sqrtf(var > 0.f ? var : 0.f);
  const float dir2 = 1.f - alpha_prev - sigma * sigma;
  const long total = (long)Ftot * HW * 4;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (half_t*)latents, (const float*)noise_sum, (const float*)counter,
                     eta > 0.f ? (const half_t*)variance_noise : nullptr, Ftot, HW * 4, halves, guidance, sqrtf(alpha_t), sqrtf(1.f - alpha_t),
                     sqrtf(alpha_prev), sqrtf(dir2 > 0.f ? dir2 : 0.f), sigma);
  MD_CHECK_LAUNCH(who);
  return MD_OK;
}

extern "C" int md_cfg_ddim_step(void* latents, const void* noise_sum, const void* counter, int Ftot, int HW, int halves, float guidance, float alpha_t,
                                float alpha_prev, void* stream) {
  return cfg_ddim_launch(latents, noise_sum, counter, nullptr, Ftot, HW, halves, guidance, alpha_t, alpha_prev, 0.f, stream, "md_cfg_ddim_step");
}

extern "C" int md_cfg_ddim_step_eta(void* latents, const void* noise_sum, const void* counter, const void* variance_noise, int Ftot, int HW, int halves,
                                    float guidance, float alpha_t, float alpha_prev, float eta, void* stream) {
  return cfg_ddim_launch(latents, noise_sum, counter, variance_noise, Ftot, HW, halves, guidance, alpha_t, alpha_prev, eta, stream,
                         "md_cfg_ddim_step_eta");
}

// ---- generic strided scatter of NHWC fp16 -> any layout/dtype (API boundary: UNet.forward returns NCFHW) ---------------
template <typename T>
__global__ void unpack_nhwc_kernel(const half_t* __restrict__ src, T* __restrict__ dst, long total, int F, long sB, long sF, long sC, long sY, long sX,
                                   int C, int ldc, int Ho, int Wo) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int x = (int)(r % Wo);
    r /= Wo;
    const int y = (int)(r % Ho);
    const long n = r / Ho;
    dst[(n / F) * sB + (n % F) * sF + (long)c * sC + (long)y * sY + (long)x * sX] = (T)(float)src[((n * Ho + y) * Wo + x) * ldc + c];
  }
}

extern "C" int md_unpack_nhwc_f16(const void* src, int ldc, void* dst, int dst_is_f32, int N, int F, long sB, long sF, long sC, long sY, long sX, int C,
                                  int Ho, int Wo, void* stream) {
  MD_CHECK_ARG(N > 0 && F > 0 && C <= ldc, "md_unpack_nhwc: bad arguments");
  const long total = (long)N * Ho * Wo * C;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dst_is_f32)
    hipLaunchKernelGGL(unpack_nhwc_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)src, (float*)dst, total, F, sB, sF, sC, sY,
                       sX, C, ldc, Ho, Wo);
  else
    hipLaunchKernelGGL(unpack_nhwc_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)src, (half_t*)dst, total, F, sB, sF, sC, sY,
                       sX, C, ldc, Ho, Wo);
  MD_CHECK_LAUNCH("md_unpack_nhwc");
  return MD_OK;
}
