// HBM-bound normalisation kernels on NHWC / token-major fp16 activations (wavefront reductions, 16-B loads).
//   md_groupnorm_nhwc_f16 : GroupNorm(G, eps) [+ SiLU]  -- reference src/models/resnet.py:20-28,220-221,231,237
//                           (InflatedGroupNorm) and the eps=1e-6 norms of transformer_3d.py:60-62 / motion_module.py:121-123
//   md_layernorm_f16      : LayerNorm(C) with optional second output y + bank (mutual_mix_attention.py:169-170) or
//                           y + positional encoding of the row's frame (motion_module.py:416-417)
//   md_instnorm_spade_f16 : InstanceNorm2d(x) * (1 + gamma) + beta   -- src/models/man_module.py:25-31
#include "common.h"
#ifndef GN_UNROLL
#define GN_UNROLL 4
#endif
#ifndef GN_THREADS
#define GN_THREADS 512
#endif
// Plain stores on purpose: non-temporal stores for the streamed outputs measured +2..16 % on these kernels alone and -0.6 % end to
// end (the consumer of a normalised tensor is the very next kernel; profiles/r03_ab_norm_nt_stores.log).
template <typename T>
__device__ __forceinline__ void norm_store(T* p, const T& v) { *p = v; }
#ifndef GN_SLAB
#define GN_SLAB 32      // rows per row-lane and slab (same-box sweep on MI355X: 8: -35 %, 16: baseline, 24-32: +9 ... +20 %, 64: -8 %)
#endif

// ------------------------------------------------------------------------------------------------ GroupNorm
// Statistics are the one-sweep sums of (x - k) and (x - k)^2 in fp32, k = the value of the group's FIRST channel at PIXEL 0 of
// the image: a pilot of the group mean (a sample of the very distribution being normalised), so the sums stay O(n * sigma) and
// var = E[(x-k)^2] - E[x-k]^2 does not cancel when |mean| >> sigma (the plain E[x^2] - mean^2 loses the variance at
// |mean| / sigma ~ 100 in fp32).  The statistics sweep reads k from x (two 2-byte loads per thread, 8 when a thread's 8 channels can
// span more than two groups) and its slab-0 workgroups hand it to the apply sweep through the workspace (bit-identical on both sides,
// and safe when the apply sweep runs in place).
// Thread (cc, r): channel chunk cc (8 channels), row lane r.  A block owns `slab` consecutive pixels of one image.
__global__ void gn_stats_kernel(const half_t* __restrict__ x, float* __restrict__ part, float* __restrict__ pilot, int HW, int C, int ldx, int G, int R,
                                int slab, int nslab) {
  extern __shared__ float sh[];  // [R][C] sums, then [R][C] sumsq
  const int cch = C >> 3;
  const int cc = threadIdx.x % cch, r = threadIdx.x / cch;
  const int b = blockIdx.y, sl = blockIdx.x;
  const int p0 = sl * slab, p1 = min(p0 + slab, HW);
  const int cpg = C / G;
  const half_t* ximg = x + (size_t)b * HW * ldx;          // ldx: pixel pitch (C, or wider when x is a channel slice of a wider tensor)
  float s[8], q[8], k[8];
  if (cpg >= 8) {                // the 8 channels of a thread lie in at most two groups
    const int g0 = (cc * 8) / cpg, g1 = (cc * 8 + 7) / cpg;
    const float k0 = (float)ximg[g0 * cpg], k1 = (float)ximg[g1 * cpg];
#pragma unroll
    for (int e = 0; e < 8; ++e) k[e] = (cc * 8 + e) / cpg == g0 ? k0 : k1;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) k[e] = (float)ximg[((cc * 8 + e) / cpg) * cpg];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  const half_t* base = ximg + cc * 8;
#pragma unroll GN_UNROLL
  for (int p = p0 + r; p < p1; p += R) {
    const half8_t v = *reinterpret_cast<const half8_t*>(base + (size_t)p * ldx);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e] - k[e];
      s[e] += f;
      q[e] += f * f;
    }
  }
  float* ss = sh;
  float* sq = sh + R * C;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ss[r * C + cc * 8 + e] = s[e];
    sq[r * C + cc * 8 + e] = q[e];
  }
  __syncthreads();
#ifndef GN_STATS_SERIAL_TAIL
  // two short steps instead of G threads walking R * cpg LDS words each (120-960 dependent reads at the END of every workgroup, with 32 of
  // its ~500 threads active): one thread per channel sums its R row lanes into row 0, then one thread per group sums its cpg channels
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = ss[c], c2 = sq[c];
    for (int rr = 1; rr < R; ++rr) {
      a += ss[rr * C + c];
      c2 += sq[rr * C + c];
    }
    ss[c] = a;                                  // row 0 of column c is read and written by this thread only
    sq[c] = c2;
  }
  __syncthreads();
#endif
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, c2 = 0.f;
#ifndef GN_STATS_SERIAL_TAIL
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      a += ss[c];
      c2 += sq[c];
    }
#else
    for (int rr = 0; rr < R; ++rr)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        a += ss[rr * C + c];
        c2 += sq[rr * C + c];
      }
#endif
    float* o = part + (((size_t)b * nslab + sl) * G + g) * 2;
    o[0] = a;
    o[1] = c2;
    // the pilot travels through the workspace: the apply sweep may run IN PLACE (y == x) and overwrite pixel 0 before the other
    // workgroups of the image have read it
    if (sl == 0) pilot[(size_t)b * G + g] = (float)ximg[g * cpg];
  }
}

// Sum of the (sum, sum of squares) partials pp[k * stride], k = k0, k0 + step, ... < n, IN THAT ORDER (every caller that must agree bit for
// bit -- gn_apply_kernel, gn_table_kernel -- walks the same sequence), with eight loads in flight: the plain loop compiles to one L2
// round trip per partial (s_waitcnt vmcnt(0) inside the loop), i.e. 24-72 serial latencies in front of every workgroup's sweep.
__device__ __forceinline__ void gn_sum_partials(const float* __restrict__ pp, size_t stride, int k0, int step, int n, float& a, float& c2) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  int k = k0;
#ifndef GN_SERIAL_PARTIALS                     // A/B knob: the one-load-per-iteration form of rounds 1-4
  for (; k + 7 * step < n; k += 8 * step) {
    f2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f2*>(pp + (size_t)(k + u * step) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a += v[u].x;
      c2 += v[u].y;
    }
  }
#endif
  for (; k < n; k += step) {
    a += pp[(size_t)k * stride];
    c2 += pp[(size_t)k * stride + 1];
  }
}

// Images of many slabs (the AutoencoderKL at 768 x 768: 576 slabs per image; the temporal decoder's clip-wide GroupNorm: ~9000): every
// workgroup of the apply sweep re-reducing all partial sums of its image is O(slabs) work per workgroup and O(slabs^2) per image -- 290 of
// the 421 ms of a temporal-decoder chunk in round 4's form (profiles/r05_vae_temporal.json).  Above GN_FIN_SLABS slabs ONE small launch
// reduces them (8 lanes per group, k = j, j + 8, ..., three exchanges) and leaves (mean, rstd) per (image, group) for the apply sweep.
#ifndef GN_FIN_SLABS
#define GN_FIN_SLABS 128
#endif
__global__ void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ pilot, float* __restrict__ fin, int HW, int C, int G, int nslab,
                                   float eps) {
  const int b = blockIdx.x, g = threadIdx.x >> 3, j = threadIdx.x & 7;
  if (g >= G) return;
  float a = 0.f, c2 = 0.f;
  const float* pp = part + ((size_t)b * nslab * G + g) * 2;
  gn_sum_partials(pp, (size_t)G * 2, j, 8, nslab, a, c2);
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    a += __shfl_xor(a, o, 64);
    c2 += __shfl_xor(c2, o, 64);
  }
  if (j == 0) {
    const float n = (float)HW * (float)(C / G);
    const float mu = a / n;
    fin[((size_t)b * G + g) * 2] = pilot[(size_t)b * G + g] + mu;
    fin[((size_t)b * G + g) * 2 + 1] = rsqrtf(fmaxf(c2 / n - mu * mu, 0.f) + eps);
  }
}

__global__ void gn_apply_kernel(const half_t* x, half_t* y, const float* __restrict__ part, const float* __restrict__ pilot, const half_t* __restrict__ gamma,
                                const half_t* __restrict__ beta, int HW, int C, int ldx, int G, int R, int slab, int nslab, float eps, int silu,
                                int zigzag, const float* __restrict__ fin) {
  __shared__ float mean_s[64], rstd_s[64];
  const int cch = C >> 3;
  const int cc = threadIdx.x % cch, r = threadIdx.x / cch;
  // zigzag: walk the slabs in the reverse of the statistics sweep's order, so that the most recently read (still cached) ones
  // are re-read first
  const int b = zigzag ? gridDim.y - 1 - blockIdx.y : blockIdx.y, sl = zigzag ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const int cpg = C / G;
  const int p0 = sl * slab, p1 = min(p0 + slab, HW);
  const size_t base = (size_t)b * HW * C + cc * 8, xbase = (size_t)b * HW * ldx + cc * 8;
  // (Requesting the thread's first rows before the statistics are assembled was measured in round 5 and is neutral -- GroupNorm family
  // 90.0 vs 89.8-90.2 ms per clip, profiles/r05_ab_groupnorm_apply_prefetch.log: the other workgroups of the CU cover that latency.)
  const half8_t gm = *reinterpret_cast<const half8_t*>(gamma + cc * 8), bt = *reinterpret_cast<const half8_t*>(beta + cc * 8);
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    if (fin) {                                            // many slabs: reduced once by gn_finalize_kernel
      mean_s[g] = fin[((size_t)b * G + g) * 2];
      rstd_s[g] = fin[((size_t)b * G + g) * 2 + 1];
      continue;
    }
    float a = 0.f, c2 = 0.f;
    gn_sum_partials(part + ((size_t)b * nslab * G + g) * 2, (size_t)G * 2, 0, 1, nslab, a, c2);
    const float n = (float)HW * (float)cpg;
    const float mu = a / n;                               // mean of x - k
    const float var = fmaxf(c2 / n - mu * mu, 0.f);
    mean_s[g] = pilot[(size_t)b * G + g] + mu;                   // + the pilot (as the statistics sweep saw it)
    rstd_s[g] = rsqrtf(var + eps);
  }
  __syncthreads();
  float sc[8], sf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cc * 8 + e;
    const int g = c / cpg;
    sc[e] = rstd_s[g] * (float)gm[e];
    sf[e] = __builtin_fmaf(-mean_s[g], sc[e], (float)bt[e]);          // explicit fma here, in gn_table_kernel and in the in-LDS apply of
  }                                                                    // wsgemm_kernel<PRO_AFF>: the three must agree bit for bit
  auto finish = [&](const half8_t& v, int p) {
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = __builtin_fmaf((float)v[e], sc[e], sf[e]);
      if (silu) f = silu_f(f);
      o[e] = (half_t)f;
    }
    norm_store(reinterpret_cast<half8_t*>(y + base + (size_t)p * C), o);
  };
#pragma unroll GN_UNROLL
  for (int p = p0 + r; p < p1; p += R) finish(*reinterpret_cast<const half8_t*>(x + xbase + (size_t)p * ldx), p);
}

// ------------------------------------------------------------------------------------------------ GroupNorm, small images: ONE sweep
// The 24 x 24 and 12 x 12 levels (HW = 576 / 144, C = 1280 ... 2560) are 12-47 MB per launch: the two-kernel form above spends them on
// launch latency and on a re-read (1.4 TB/s at HW = 144, 3.3 at HW = 576 against 4.4-4.7 on the large levels).  Here ONE 256-thread
// workgroup owns a whole (image, group): HW x cpg values (<= 46 080) sit in its REGISTERS as 8-byte pieces, so the tensor is read once,
// the statistics are the exact two-pass ones (mean, then sum of (x - mean)^2: no pilot needed, nothing to cancel), and the normalised
// values are written from the registers: 4 bytes per element instead of 6, one launch instead of two, no workspace.  Thread (tc, tp):
// piece tc of the group's cpg / 4 pieces of a pixel, pixel lane tp; it visits pixels tp, tp + R, ...  (at most NIT of them).  The 80-byte
// (cpg = 40) slices of a 2560-byte pixel row share cache lines with the neighbouring groups' workgroups: at these sizes the tensor
// lives in L2 / the memory-side cache, so that costs requests, not HBM bytes; the large levels stay on the two-kernel form.
template <int NIT, int NT>
__global__ __launch_bounds__(NT) void gn_small_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const half_t* __restrict__ gamma,
                                                       const half_t* __restrict__ beta, int HW, int C, int ldx, int cpg, int R, float eps, int silu, int B) {
  constexpr int NWV = NT / 64;
  __shared__ float red[2][NWV];
  const int cpr = cpg >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tc = tid % cpr, tp = tid / cpr;
  // workgroup L runs on XCD L % 8 (observed dispatch order, speed only): give each XCD G / 8 CONSECUTIVE groups of every image, so that
  // the cache lines two neighbouring groups share (a group's slice of a pixel row is 40-160 bytes) are fetched into ONE L2
  const int G = gridDim.x / B, L = blockIdx.x;                  // 1-D grid of B * G workgroups
  const int b = L / G, l = L - b * G;
  const int g = (G & 7) ? l : (l & 7) * (G >> 3) + (l >> 3);
  const bool live = tp < R;
  const half_t* xb = x + (size_t)b * HW * ldx + g * cpg + tc * 4;
  half4_t v[NIT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int p = tp + i * R;
    v[i] = half4_t{0, 0, 0, 0};
    if (live && p < HW) v[i] = *reinterpret_cast<const half4_t*>(xb + (size_t)p * ldx);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) s += ((float)v[i][0] + (float)v[i][1]) + ((float)v[i][2] + (float)v[i][3]);
  s = wave_sum(s);
  if (lane == 0) red[0][wave] = s;
  __syncthreads();
  const float n = (float)HW * (float)cpg;
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) tot += red[0][w];
  const float mean = tot / n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int p = tp + i * R;
    if (live && p < HW) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = (float)v[i][e] - mean;
        q += d * d;
      }
    }
  }
  q = wave_sum(q);
  if (lane == 0) red[1][wave] = q;
  __syncthreads();
  float qt = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) qt += red[1][w];
  const float rstd = rsqrtf(qt / n + eps);
  if (!live) return;
  const half4_t gm = *reinterpret_cast<const half4_t*>(gamma + g * cpg + tc * 4), bt = *reinterpret_cast<const half4_t*>(beta + g * cpg + tc * 4);
  float sc[4], sf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = rstd * (float)gm[e];
    sf[e] = (float)bt[e] - mean * sc[e];
  }
  half_t* yb = y + (size_t)b * HW * C + g * cpg + tc * 4;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int p = tp + i * R;
    if (p < HW) {
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float f = (float)v[i][e] * sc[e] + sf[e];
        if (silu) f = silu_f(f);
        o[e] = (half_t)f;
      }
      *reinterpret_cast<half4_t*>(yb + (size_t)p * C) = o;
    }
  }
}

// the one-sweep form applies when a group's channels are whole 8-byte pieces, an (image, group) fits the registers of 256 threads and
// the grid fills the chip; returns the sweeps per thread (0: use the two-kernel form)
static int gn_small_iters(int B, int HW, int C, int G, int* R, int* NT) {
  const int cpg = C / G;
  if (cpg % 4 || cpg / 4 > 64 || (long)B * G < 256 || HW > 1024) return 0;      // HW = 2304, C = 640 (94 MB: HBM-bound) measured 10 % SLOWER: 40-byte slices
  *NT = cdiv(HW, 256 / (cpg / 4)) > 24 ? 512 : 256;            // <= 24 sweeps per thread (139 registers: 3 waves per SIMD)
  *R = *NT / (cpg / 4);
  const int it = cdiv(HW, *R);
  return it <= 24 ? it : 0;
}

// Launch geometry: R row lanes per 8-channel chunk (~512 threads), slabs of GN_SLAB rows per lane -- halved while the grid would
// not give every CU two workgroups (small images).
static void gn_geometry(int B, int HW, int C, int& R, int& slab, int& nslab) {
  const int cch = C / 8;
  R = GN_THREADS / cch;
  if (R < 1) R = 1;
  int rows = GN_SLAB;
  while (rows > 8 && (long)B * cdiv(HW, R * rows) < 512) rows >>= 1;
  slab = R * rows;
  nslab = cdiv(HW, slab);
}

extern "C" size_t md_groupnorm_workspace_bytes(int B, int HW, int C, int G) {
  int R, slab, nslab;
  gn_geometry(B, HW, C, R, slab, nslab);
  return ((size_t)B * nslab * G * 2 + (size_t)B * G * 3) * sizeof(float);   // partial sums + one pilot + (mean, rstd) per (image, group)
}

extern "C" int md_groupnorm_ld_nhwc_f16(const void* x, int ldx, void* y, const void* gamma, const void* beta, int B, int HW, int C, int G, float eps,
                                        int silu, void* workspace, size_t ws_bytes, void* stream) {
  MD_CHECK_ARG(C % 8 == 0 && G > 0 && G <= 64 && C % G == 0, "md_groupnorm: need C %% 8 == 0, G <= 64, C %% G == 0 (C=%d G=%d)", C, G);
  MD_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (ldx == C || x != y),
               "md_groupnorm: ldx=%d must be a multiple of 8 and >= C=%d, x 16-byte aligned; in place only with ldx == C", ldx, C);
  MD_CHECK_ARG(C / 8 <= 1024, "md_groupnorm: C=%d too large", C);
  MD_CHECK_ARG(((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0, "md_groupnorm: gamma / beta must be 16-byte aligned");
  MD_CHECK_ARG(ws_bytes >= md_groupnorm_workspace_bytes(B, HW, C, G), "md_groupnorm: workspace too small");
  {
    int Rs, NTs;
    const int it = (ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0) ? gn_small_iters(B, HW, C, G, &Rs, &NTs) : 0;
    if (it) {
#define GN_SMALL(N, T)                                                                                                                       \
  hipLaunchKernelGGL((gn_small_kernel<N, T>), dim3(B * G), dim3(T), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, (const half_t*)gamma, \
                     (const half_t*)beta, HW, C, ldx, C / G, Rs, eps, silu, B)
      if (NTs == 512) {
        if (it <= 12) GN_SMALL(12, 512);
        else GN_SMALL(24, 512);
      } else if (it <= 6) GN_SMALL(6, 256);
      else if (it <= 12) GN_SMALL(12, 256);
      else GN_SMALL(24, 256);
#undef GN_SMALL
      MD_CHECK_LAUNCH("md_groupnorm");
      return MD_OK;
    }
  }
  const int cch = C / 8;
  int R, slab, nslab;
  gn_geometry(B, HW, C, R, slab, nslab);
  // The apply sweep re-reads what the statistics sweep has just read, in reverse order (most recently read slabs first): +3 % at
  // C = 640, +-0 elsewhere.  Splitting the batch into chunks so that the re-read would be served from the 256-MB memory-side cache
  // was measured and is a loss at every chunk size (24 MB: 2.7x slower, 96 MB: -12 %): launch gaps and tails, no visible hit-rate gain.
  const int zigzag = 1;
  const dim3 grid(nslab, B), block(cch * R);
  const size_t sh = (size_t)2 * R * C * sizeof(float);
  float* pilot = (float*)workspace + (size_t)B * nslab * G * 2;
  float* fin = nslab > GN_FIN_SLABS ? pilot + (size_t)B * G : nullptr;
  hipLaunchKernelGGL(gn_stats_kernel, grid, block, sh, (hipStream_t)stream, (const half_t*)x, (float*)workspace, pilot, HW, C, ldx, G, R, slab, nslab);
  if (fin)
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(8 * G), 0, (hipStream_t)stream, (const float*)workspace, (const float*)pilot, fin, HW, C, G, nslab,
                       eps);
  hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, (const float*)workspace, (const float*)pilot,
                     (const half_t*)gamma, (const half_t*)beta, HW, C, ldx, G, R, slab, nslab, eps, silu, zigzag, (const float*)fin);
  MD_CHECK_LAUNCH("md_groupnorm");
  return MD_OK;
}

// GroupNorm as a TABLE: the statistics sweep above, then scale[b][c] = rstd * gamma[c], shift[b][c] = beta[c] - mean * scale (exactly what
// gn_apply_kernel derives per workgroup, same partial-sum order) written as fp32 [B][2][C].  The consumer -- md_gemm_affine_f16, the
// proj_in of Transformer3DModel / the motion module's temporal transformer -- applies x * scale + shift to the rows it has just
// streamed into LDS, so the normalised tensor never exists in HBM: 2 bytes per element (this sweep) instead of 6.
__global__ void gn_table_kernel(const float* __restrict__ part, const float* __restrict__ pilot, const half_t* __restrict__ gamma,
                                const half_t* __restrict__ beta, float* __restrict__ table, int HW, int C, int G, int nslab, float eps,
                                const float* __restrict__ fin) {
  const int b = blockIdx.x, cpg = C / G;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / cpg;
    float mean, rstd;
    if (fin) {                                            // many slabs: the same (mean, rstd) the apply sweep would use
      mean = fin[((size_t)b * G + g) * 2];
      rstd = fin[((size_t)b * G + g) * 2 + 1];
    } else {
      float a = 0.f, c2 = 0.f;
      gn_sum_partials(part + ((size_t)b * nslab * G + g) * 2, (size_t)G * 2, 0, 1, nslab, a, c2);
      const float n = (float)HW * (float)cpg;
      const float mu = a / n;                             // mean of x - k
      mean = pilot[(size_t)b * G + g] + mu;
      rstd = rsqrtf(fmaxf(c2 / n - mu * mu, 0.f) + eps);
    }
    const float sc = rstd * (float)gamma[c];
    table[(size_t)b * 2 * C + c] = sc;
    table[(size_t)b * 2 * C + C + c] = __builtin_fmaf(-mean, sc, (float)beta[c]);
  }
}

extern "C" int md_groupnorm_table_f16(const void* x, int ldx, const void* gamma, const void* beta, int B, int HW, int C, int G, float eps, float* table,
                                      void* workspace, size_t ws_bytes, void* stream) {
  MD_CHECK_ARG(C % 8 == 0 && G > 0 && G <= 64 && C % G == 0, "md_groupnorm_table: need C %% 8 == 0, G <= 64, C %% G == 0 (C=%d G=%d)", C, G);
  MD_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "md_groupnorm_table: ldx=%d must be a multiple of 8 and >= C=%d, x 16-byte aligned", ldx, C);
  MD_CHECK_ARG(C / 8 <= 1024 && (reinterpret_cast<uintptr_t>(table) & 15) == 0, "md_groupnorm_table: C=%d too large or table not 16-byte aligned", C);
  MD_CHECK_ARG(ws_bytes >= md_groupnorm_workspace_bytes(B, HW, C, G), "md_groupnorm_table: workspace too small");
  const int cch = C / 8;
  int R, slab, nslab;
  gn_geometry(B, HW, C, R, slab, nslab);
  const dim3 grid(nslab, B), block(cch * R);
  const size_t sh = (size_t)2 * R * C * sizeof(float);
  float* pilot = (float*)workspace + (size_t)B * nslab * G * 2;
  hipLaunchKernelGGL(gn_stats_kernel, grid, block, sh, (hipStream_t)stream, (const half_t*)x, (float*)workspace, pilot, HW, C, ldx, G, R, slab, nslab);
  float* fin = nslab > GN_FIN_SLABS ? pilot + (size_t)B * G : nullptr;
  if (fin)
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(8 * G), 0, (hipStream_t)stream, (const float*)workspace, (const float*)pilot, fin, HW, C, G, nslab,
                       eps);
  hipLaunchKernelGGL(gn_table_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (const float*)pilot, (const half_t*)gamma,
                     (const half_t*)beta, table, HW, C, G, nslab, eps, (const float*)fin);
  MD_CHECK_LAUNCH("md_groupnorm_table");
  return MD_OK;
}

extern "C" int md_groupnorm_nhwc_f16(const void* x, void* y, const void* gamma, const void* beta, int B, int HW, int C, int G, float eps, int silu,
                                     void* workspace, size_t ws_bytes, void* stream) {
  return md_groupnorm_ld_nhwc_f16(x, C, y, gamma, beta, B, HW, C, G, eps, silu, workspace, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave owns LN_R consecutive rows held in registers (C <= 8*64*MAXC): the loads of all LN_R rows are issued before
// the first reduction, which quadruples the bytes in flight per wave (a single 640-byte row per wave left the kernel
// latency bound at ~3.3 TB/s).
#define LN_R 4
#define LN_MAXC 4
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, half_t* __restrict__ y2,
                                                        const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                        const half_t* __restrict__ add, int M, int C, float eps, int add_mode, int add_row_begin,
                                                        int rows_per_frame, int frames) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_R;
  if (row0 >= M) return;
  const int cch = C >> 3;
  float v[LN_R][MAXC][8];
#pragma unroll
  for (int r = 0; r < LN_R; ++r) {
    const int row = min(row0 + r, M - 1);
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + k * 64;
      half8_t h = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < cch) h = *reinterpret_cast<const half8_t*>(x + (size_t)row * C + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[r][k][e] = (float)h[e];
    }
  }
  half8_t g[MAXC], bt[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + k * 64;
    if (c < cch) {
      g[k] = *reinterpret_cast<const half8_t*>(gamma + c * 8);
      bt[k] = *reinterpret_cast<const half8_t*>(beta + c * 8);
    }
  }
  float mu[LN_R], rstd[LN_R];
#pragma unroll
  for (int r = 0; r < LN_R; ++r) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[r][k][e];      // lanes beyond the row hold zeros
    mu[r] = sum;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < LN_R; ++r) mu[r] += __shfl_xor(mu[r], o, 64);
#pragma unroll
  for (int r = 0; r < LN_R; ++r) {
    mu[r] /= (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      if (lane + k * 64 < cch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[r][k][e] - mu[r];
          sq += d * d;
        }
      }
    }
    rstd[r] = sq;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < LN_R; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
  for (int r = 0; r < LN_R; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    const float rs = rsqrtf(rstd[r] / (float)C + eps);
    const half_t* addp = nullptr;
    if (add_mode == 1 && row >= add_row_begin) addp = add + (size_t)(row - add_row_begin) * C;  // bank rows
    if (add_mode == 2) addp = add + (size_t)((row / rows_per_frame) % frames) * C;            // positional encoding of the frame
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + k * 64;
      if (c < cch) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((v[r][k][e] - mu[r]) * rs * (float)g[k][e] + (float)bt[k][e]);
        norm_store(reinterpret_cast<half8_t*>(y + (size_t)row * C + c * 8), o);
        if (y2) {
          half8_t o2 = o;
          if (addp) {
            const half8_t a = *reinterpret_cast<const half8_t*>(addp + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o2[e] = (half_t)((float)o[e] + (float)a[e]);  // fp16 n + fp16 bank, one rounding
          }
          norm_store(reinterpret_cast<half8_t*>(y2 + (size_t)row * C + c * 8), o2);
        }
      }
    }
  }
}

// C = 320 (the 96 x 96 level: half of the clip's LayerNorm time).  A 640-byte row fills 40 of a wave's 64 lanes.  Here a wave owns EIGHT
// consecutive rows = 320 16-byte chunks = five full-wave loads of one contiguous 5-KiB block: chunk q = 64 i + lane belongs to row
// q / 40.  Row statistics are accumulated per lane into the (at most three) rows a load index can touch and reduced across the wave
// as above; same arithmetic per element.  Same box: 0.081-0.083 -> 0.077-0.080 ms at M = 294912 (4.6 -> 4.8 TB/s), 0.034 -> 0.032 at
// M = 147456; a persistent form that requests a wave's next block before reducing the current one needs 168 registers and runs at
// 0.097-0.100 ms (profiles/r04_ab_layernorm_c320.log).
__global__ __launch_bounds__(256) void layernorm320_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, half_t* __restrict__ y2,
                                                           const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                           const half_t* __restrict__ add, int M, float eps, int add_mode, int add_row_begin,
                                                           int rows_per_frame, int frames) {
  constexpr int C = 320, CCH = 40, R = 8, NI = 5;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  int ri[NI], ci[NI];
  float v[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = i * 64 + lane;
    ri[i] = q / CCH;
    ci[i] = q - ri[i] * CCH;
    const int row = min(row0 + ri[i], M - 1);
    const half8_t h = *reinterpret_cast<const half8_t*>(x + (size_t)row * C + ci[i] * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[i][e] = (float)h[e];
  }
  float mu[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mu[r] = rstd[r] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[i][e];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r >= (i * 64) / CCH && r <= (i * 64 + 63) / CCH) mu[r] += ri[i] == r ? s : 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < R; ++r) mu[r] += __shfl_xor(mu[r], o, 64);
#pragma unroll
  for (int r = 0; r < R; ++r) mu[r] /= (float)C;
  float mi[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r >= (i * 64) / CCH && r <= (i * 64 + 63) / CCH) m = ri[i] == r ? mu[r] : m;
    mi[i] = m;
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[i][e] - m;
      sq += d * d;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r >= (i * 64) / CCH && r <= (i * 64 + 63) / CCH) rstd[r] += ri[i] == r ? sq : 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(rstd[r] / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = row0 + ri[i];
    if (row >= M) continue;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r >= (i * 64) / CCH && r <= (i * 64 + 63) / CCH) rs = ri[i] == r ? rstd[r] : rs;
    const half8_t g = *reinterpret_cast<const half8_t*>(gamma + ci[i] * 8);
    const half8_t bt = *reinterpret_cast<const half8_t*>(beta + ci[i] * 8);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((v[i][e] - mi[i]) * rs * (float)g[e] + (float)bt[e]);
    norm_store(reinterpret_cast<half8_t*>(y + (size_t)row * C + ci[i] * 8), o);
    if (y2) {
      const half_t* addp = nullptr;
      if (add_mode == 1 && row >= add_row_begin) addp = add + (size_t)(row - add_row_begin) * C;  // bank rows
      if (add_mode == 2) addp = add + (size_t)((row / rows_per_frame) % frames) * C;            // positional encoding of the frame
      half8_t o2 = o;
      if (addp) {
        const half8_t a = *reinterpret_cast<const half8_t*>(addp + ci[i] * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o2[e] = (half_t)((float)o[e] + (float)a[e]);  // fp16 n + fp16 bank, one rounding
      }
      norm_store(reinterpret_cast<half8_t*>(y2 + (size_t)row * C + ci[i] * 8), o2);
    }
  }
}

extern "C" int md_layernorm_f16(const void* x, void* y, void* y2, const void* gamma, const void* beta, const void* add, int M, int C, float eps,
                                int add_mode, int add_row_begin, int rows_per_frame, int frames, void* stream) {
  MD_CHECK_ARG(C % 8 == 0 && C <= 8 * 64 * LN_MAXC, "md_layernorm: C=%d must be a multiple of 8 and <= %d", C, 8 * 64 * LN_MAXC);
  MD_CHECK_ARG(add_mode == 0 || (add != nullptr && y2 != nullptr), "md_layernorm: add_mode needs add and y2");
  MD_CHECK_ARG(add_mode != 2 || (rows_per_frame > 0 && frames > 0), "md_layernorm: add_mode 2 needs rows_per_frame and frames");
  const dim3 grid(cdiv(M, 4 * LN_R)), block(256);
  hipStream_t st = (hipStream_t)stream;
#ifndef LN_NO_320
  if (C == 320) {
    hipLaunchKernelGGL(layernorm320_kernel, dim3(cdiv(M, 32)), block, 0, st, (const half_t*)x, (half_t*)y, (half_t*)y2, (const half_t*)gamma,
                       (const half_t*)beta, (const half_t*)add, M, eps, add_mode, add_row_begin, rows_per_frame, frames);
    MD_CHECK_LAUNCH("md_layernorm");
    return MD_OK;
  }
#endif
#define LN_LAUNCH(MC)                                                                                                                           \
  hipLaunchKernelGGL(layernorm_kernel<MC>, grid, block, 0, st, (const half_t*)x, (half_t*)y, (half_t*)y2, (const half_t*)gamma, (const half_t*)beta, \
                     (const half_t*)add, M, C, eps, add_mode, add_row_begin, rows_per_frame, frames)
  if (C <= 512) LN_LAUNCH(1);
  else if (C <= 1024) LN_LAUNCH(2);
  else if (C <= 1536) LN_LAUNCH(3);
  else LN_LAUNCH(4);
#undef LN_LAUNCH
  MD_CHECK_LAUNCH("md_layernorm");
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------ InstanceNorm + SPADE
// Block = 64 channels (8 chunks) x 32 row lanes of one image; two passes over HW (second pass is L2 resident).
__global__ __launch_bounds__(256) void instnorm_spade_kernel(const half_t* __restrict__ x, const half_t* __restrict__ gb, half_t* __restrict__ y, int HW,
                                                             int C, int ldx, float eps) {
  __shared__ float ss[32][64], sq[32][64], mean_s[64], rstd_s[64];
  const int cc = threadIdx.x & 7, r = threadIdx.x >> 3;
  const int b = blockIdx.y, c0 = blockIdx.x * 64 + cc * 8;
  const half_t* xb = x + (size_t)b * HW * ldx + c0;
  // sums of (x - k), (x - k)^2 with k = the channel's value at pixel 0 (a pilot of its mean: no cancellation when |mean| >> sigma)
  const half8_t k8 = *reinterpret_cast<const half8_t*>(xb);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  for (int p = r; p < HW; p += 32) {
    const half8_t v = *reinterpret_cast<const half8_t*>(xb + (size_t)p * ldx);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e] - (float)k8[e];
      s[e] += f;
      q[e] += f * f;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ss[r][cc * 8 + e] = s[e];
    sq[r][cc * 8 + e] = q[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float a = 0.f, c2 = 0.f;
    for (int rr = 0; rr < 32; ++rr) {
      a += ss[rr][threadIdx.x];
      c2 += sq[rr][threadIdx.x];
    }
    const float mu = a / (float)HW;                     // mean of x - k
    mean_s[threadIdx.x] = (float)x[(size_t)b * HW * ldx + blockIdx.x * 64 + threadIdx.x] + mu;
    rstd_s[threadIdx.x] = rsqrtf(fmaxf(c2 / (float)HW - mu * mu, 0.f) + eps);
  }
  __syncthreads();
  const half_t* gbb = gb + (size_t)b * HW * 2 * C + c0;
  half_t* yb = y + (size_t)b * HW * C + c0;
  for (int p = r; p < HW; p += 32) {
    const half8_t v = *reinterpret_cast<const half8_t*>(xb + (size_t)p * ldx);
    const half8_t ga = *reinterpret_cast<const half8_t*>(gbb + (size_t)p * 2 * C);
    const half8_t be = *reinterpret_cast<const half8_t*>(gbb + (size_t)p * 2 * C + C);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float n = ((float)v[e] - mean_s[cc * 8 + e]) * rstd_s[cc * 8 + e];
      o[e] = (half_t)(n * (1.f + (float)ga[e]) + (float)be[e]);
    }
    *reinterpret_cast<half8_t*>(yb + (size_t)p * C) = o;
  }
}

extern "C" int md_instnorm_spade_ld_f16(const void* x, int ldx, const void* gamma_beta, void* y, int B, int HW, int C, float eps, void* stream) {
  MD_CHECK_ARG(C % 64 == 0, "md_instnorm_spade: C=%d must be a multiple of 64", C);
  MD_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "md_instnorm_spade: ldx=%d must be a multiple of 8 and >= C=%d", ldx, C);
  hipLaunchKernelGGL(instnorm_spade_kernel, dim3(C / 64, B), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)gamma_beta, (half_t*)y, HW,
                     C, ldx, eps);
  MD_CHECK_LAUNCH("md_instnorm_spade");
  return MD_OK;
}

extern "C" int md_instnorm_spade_f16(const void* x, const void* gamma_beta, void* y, int B, int HW, int C, float eps, void* stream) {
  return md_instnorm_spade_ld_f16(x, C, gamma_beta, y, B, HW, C, eps, stream);
}

// ------------------------------------------------------------------------------------------------ row softmax
// In-place softmax(scale * x) over the rows of a row-major fp16 matrix: the score matrix of the VAE mid-block attention
// (ONE head of 512 channels: too wide for the flash kernel's register tile, so that attention runs as QK^T GEMM ->
// this kernel -> PV GEMM).  One workgroup per row; the row (<= 32 KiB) stays in L2 between the three sweeps; statistics
// in fp32.  HBM-bound: 4 bytes per element (read + write once).
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* x, long ldx, int cols, float scale_log2) {
  half_t* row = x + (size_t)blockIdx.x * ldx;
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = cols >> 3;                         // cols % 8 == 0 and 16-B aligned rows (checked by the launcher)
  float m = -1.0e30f;
  for (int i = tid; i < nvec; i += 256) {
    const half8_t v = *reinterpret_cast<const half8_t*>(row + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, (float)v[j]);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2;
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < nvec; i += 256) {
    const half8_t v = *reinterpret_cast<const half8_t*>(row + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += __builtin_amdgcn_exp2f((float)v[j] * scale_log2 - m);
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = tid; i < nvec; i += 256) {
    const half8_t v = *reinterpret_cast<const half8_t*>(row + i * 8);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(__builtin_amdgcn_exp2f((float)v[j] * scale_log2 - m) * inv);
    *reinterpret_cast<half8_t*>(row + i * 8) = o;
  }
}

extern "C" int md_softmax_rows_f16(void* x, int ldx, int rows, int cols, float scale, void* stream) {
  MD_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldx >= cols, "md_softmax_rows: rows=%d cols=%d ldx=%d (cols, ldx multiples of 8)", rows, cols, ldx);
  MD_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "md_softmax_rows: x must be 16-byte aligned");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (half_t*)x, (long)ldx, cols, scale * 1.4426950408889634f);
  MD_CHECK_LAUNCH("md_softmax_rows");
  return MD_OK;
}
