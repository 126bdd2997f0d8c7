// Temporal self-attention core of the AnimateDiff motion module (reference src/models/motion_module.py:364-439):
// for every (clip-half b, pixel, head) an f x f attention over FRAMES, f <= 32.  HBM-bound (4 bytes moved per 2*f
// MACs), so no MFMA: one lane owns one (pixel, head, query frame), keeps its f scores in registers, and reads K/V
// rows with 16-byte loads that the 8 head-lanes of a frame coalesce into whole token rows.  The '(b f) d c <-> (b d) f c'
// transposes of the reference (:404-406, :437) are folded into the addressing: inputs and output stay in the
// token-major [(b f) HW][C] layout.  The positional encoding is added to the Q input only (quirk 6) by the LayerNorm
// kernel's second output, before the to_q GEMM.
#include "common.h"

struct TemporalParams {
  const half_t* Q;
  const half_t* K;
  const half_t* V;
  half_t* O;
  int ldq, ldk, ldv, ldo;
  int NB, F, HW, H, D;  // NB clip-halves, F frames, D head dim
  int PB, HG;           // pixels and heads per workgroup
  float scale_log2;
};

// Workgroup = PB pixels x HG heads x F query frames (one lane each).  The K and V rows of those pixels/heads for ALL F
// frames are first staged in LDS with fully coalesced 16-byte loads (every K/V byte is read from HBM exactly once and
// then re-used by the F query lanes from LDS: ds_read_b128, conflict free because the HG head-lanes sit D*2 bytes
// apart and the F frame-lanes broadcast).
template <int FMAX>
__global__ __launch_bounds__(256) void temporal_attn_kernel(TemporalParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int CW = p.HG * p.D;                       // staged columns per row
  const int cw8 = CW >> 3;
  const int ngrp = p.H / p.HG;
  const int grp = blockIdx.x % ngrp;
  const long pg = (long)(blockIdx.x / ngrp) * p.PB;  // first global (b, pixel) of this workgroup
  const long npix = (long)p.NB * p.HW;
  const int col0 = grp * CW;
  // ---- stage K and V by direct-to-LDS DMA: chunk id c -> (frame j, pixel pl, 16-B column chunk) lands at LDS
  // offset 16*c, i.e. the LDS image is lane linear, so every wave issues all its DMAs back to back (no VGPR staging,
  // maximum memory-level parallelism) and waits once.  Each region is padded to a whole number of 1-KiB DMA rows.
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nchunk = p.F * p.PB * cw8;
  const int region = ((nchunk * 16 + 1023) >> 10) << 10;
  half_t* Ks = reinterpret_cast<half_t*>(smem);
  half_t* Vs = reinterpret_cast<half_t*>(smem + region);
  for (int c = t; c < (region >> 4); c += blockDim.x) {
    const int cl = min(c, nchunk - 1);
    const int cc = cl % cw8;
    const int pl = (cl / cw8) % p.PB;
    const int j = cl / (cw8 * p.PB);
    const long gp = min(pg + pl, npix - 1);
    const int b = (int)(gp / p.HW), pix = (int)(gp % p.HW);
    const size_t row = ((size_t)b * p.F + j) * p.HW + pix;
    const int cbase = __builtin_amdgcn_readfirstlane(c) << 4;    // lane 0 of the wave: c - lane
    __builtin_amdgcn_global_load_lds((gptr_t)(p.K + row * p.ldk + col0 + cc * 8), (lptr_t)(smem + cbase), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(p.V + row * p.ldv + col0 + cc * 8), (lptr_t)(smem + region + cbase), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int hl = t % p.HG;
  const int i = (t / p.HG) % p.F;
  const int pl = t / (p.HG * p.F);
  const long gp = pg + pl;
  if (pl >= p.PB || gp >= npix) return;
  const int b = (int)(gp / p.HW), pix = (int)(gp % p.HW);
  const size_t qrow = ((size_t)b * p.F + i) * p.HW + pix;
  const half_t* qp = p.Q + qrow * p.ldq + col0 + hl * p.D;
  half_t* op = p.O + qrow * p.ldo + col0 + hl * p.D;
  const half_t* kp = Ks + (size_t)pl * CW + hl * p.D;   // frame j at + j * PB * CW
  const half_t* vp = Vs + (size_t)pl * CW + hl * p.D;
  const int fstride = p.PB * CW;
  const int nch = p.D >> 3;

  // No per-frame branches: frames beyond F are clamped to F-1 for the loads and masked arithmetically afterwards, so the
  // compiler can batch all FMAX ds_read_b128 of a chunk ahead of the dot products (a branch per frame serialised one
  // LDS round trip per frame).
  int joff[FMAX];
#pragma unroll
  for (int j = 0; j < FMAX; ++j) joff[j] = min(j, p.F - 1) * fstride;
  float s[FMAX];
#pragma unroll
  for (int j = 0; j < FMAX; ++j) s[j] = 0.f;
  for (int c = 0; c < nch; ++c) {
    const half8_t q8 = *reinterpret_cast<const half8_t*>(qp + c * 8);
    half8_t k8[FMAX];
#pragma unroll
    for (int j = 0; j < FMAX; ++j) k8[j] = *reinterpret_cast<const half8_t*>(kp + joff[j] + c * 8);
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      float a = s[j];
#pragma unroll
      for (int e = 0; e < 8; e += 2) a = __builtin_amdgcn_fdot2(half2_t{q8[e], q8[e + 1]}, half2_t{k8[j][e], k8[j][e + 1]}, a, false);
      s[j] = a;
    }
  }
  float m = -1.0e30f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    s[j] = j < p.F ? s[j] : -1.0e30f;
    m = fmaxf(m, s[j]);
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    s[j] = __builtin_amdgcn_exp2f((s[j] - m) * p.scale_log2);   // masked frames: exp2(-huge) = 0
    l += s[j];
  }
  const float inv = 1.f / l;
  for (int c = 0; c < nch; ++c) {
    half8_t v8[FMAX];
#pragma unroll
    for (int j = 0; j < FMAX; ++j) v8[j] = *reinterpret_cast<const half8_t*>(vp + joff[j] + c * 8);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += s[j] * (float)v8[j][e];
    }
    half8_t ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = (half_t)(o[e] * inv);
    *reinterpret_cast<half8_t*>(op + c * 8) = ov;
  }
}

template <int FMAX>
static void launch_temporal(const TemporalParams& p, int grid, int threads, size_t smem, hipStream_t st) {
  md_ensure_dynamic_lds<temporal_attn_kernel<FMAX>>(96 * 1024);
  hipLaunchKernelGGL(temporal_attn_kernel<FMAX>, dim3(grid), dim3(threads), smem, st, p);
}

extern "C" int md_temporal_attention_fwd_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int NB, int F, int HW,
                                             int H, int D, float scale, void* stream) {
  MD_CHECK_ARG(F >= 1 && F <= 32, "md_temporal_attention_fwd: F=%d frames, the positional-encoding table holds 32", F);
  MD_CHECK_ARG(D % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "md_temporal_attention_fwd: D and strides must be multiples of 8");
  MD_CHECK_ARG(H >= 1 && H <= 8 && (H & (H - 1)) == 0, "md_temporal_attention_fwd: H=%d heads must be a power of two <= 8", H);
  TemporalParams p;
  p.Q = (const half_t*)Q; p.K = (const half_t*)K; p.V = (const half_t*)V; p.O = (half_t*)O;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.NB = NB; p.F = F; p.HW = HW; p.H = H; p.D = D;
  // heads per workgroup: as many as keep one pixel's K+V (4*F*HG*D bytes) within 48 KiB and HG*F lanes within 256
  int HG = H;
  while (HG > 1 && ((size_t)4 * F * HG * D > 48 * 1024 || HG * F > 256)) HG >>= 1;
  MD_CHECK_ARG((size_t)4 * F * HG * D <= 96 * 1024 && HG * F <= 256, "md_temporal_attention_fwd: F*D too large for LDS");
  int PB = 256 / (HG * F);
  const int pb_lds = (int)((48 * 1024) / ((size_t)4 * F * HG * D));
  if (PB > pb_lds) PB = pb_lds;
  if (PB < 1) PB = 1;
  p.PB = PB; p.HG = HG;
  p.scale_log2 = scale * 1.4426950408889634f;
  const int threads = ((PB * HG * F + 63) / 64) * 64;
  const size_t region = (((size_t)2 * F * PB * HG * D + 1023) >> 10) << 10;   // K (and V) image, whole 1-KiB DMA rows
  const size_t smem = 2 * region;
  const int grid = cdiv((long)NB * HW, PB) * (H / HG);
  hipStream_t st = (hipStream_t)stream;
  if (F <= 4) launch_temporal<4>(p, grid, threads, smem, st);
  else if (F <= 8) launch_temporal<8>(p, grid, threads, smem, st);
  else if (F <= 16) launch_temporal<16>(p, grid, threads, smem, st);
  else launch_temporal<32>(p, grid, threads, smem, st);
  MD_CHECK_LAUNCH("md_temporal_attention_fwd");
  return MD_OK;
}
