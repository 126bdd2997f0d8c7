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
  int pix_per_block;
  float scale_log2;
};

template <int FMAX>
__global__ __launch_bounds__(256) void temporal_attn_kernel(TemporalParams p) {
  const int t = threadIdx.x;
  const int h = t % p.H;
  const int i = (t / p.H) % p.F;
  const int pl = t / (p.H * p.F);
  if (pl >= p.pix_per_block) return;
  const long gp = (long)blockIdx.x * p.pix_per_block + pl;  // global (b, pixel)
  if (gp >= (long)p.NB * p.HW) return;
  const int b = (int)(gp / p.HW), pix = (int)(gp % p.HW);
  const size_t row0 = (size_t)b * p.F * p.HW + pix;  // row of frame 0; frame j is row0 + j*HW
  const int col = h * p.D;
  const half_t* qp = p.Q + (row0 + (size_t)i * p.HW) * p.ldq + col;
  const half_t* kp = p.K + row0 * p.ldk + col;
  const half_t* vp = p.V + row0 * p.ldv + col;
  half_t* op = p.O + (row0 + (size_t)i * p.HW) * p.ldo + col;
  const int nch = p.D >> 3;

  float s[FMAX];
#pragma unroll
  for (int j = 0; j < FMAX; ++j) s[j] = 0.f;
  for (int c = 0; c < nch; ++c) {
    const half8_t q8 = *reinterpret_cast<const half8_t*>(qp + c * 8);
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      if (j < p.F) {
        const half8_t k8 = *reinterpret_cast<const half8_t*>(kp + (size_t)j * p.HW * p.ldk + c * 8);
        float a = s[j];
#pragma unroll
        for (int e = 0; e < 8; e += 2) a = __builtin_amdgcn_fdot2(half2_t{q8[e], q8[e + 1]}, half2_t{k8[e], k8[e + 1]}, a, false);
        s[j] = a;
      }
    }
  }
  float m = -1.0e30f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j)
    if (j < p.F) m = fmaxf(m, s[j]);
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    if (j < p.F) {
      s[j] = __builtin_amdgcn_exp2f((s[j] - m) * p.scale_log2);
      l += s[j];
    } else {
      s[j] = 0.f;
    }
  }
  const float inv = 1.f / l;
  for (int c = 0; c < nch; ++c) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      if (j < p.F) {
        const half8_t v8 = *reinterpret_cast<const half8_t*>(vp + (size_t)j * p.HW * p.ldv + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += s[j] * (float)v8[e];
      }
    }
    half8_t ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = (half_t)(o[e] * inv);
    *reinterpret_cast<half8_t*>(op + c * 8) = ov;
  }
}

extern "C" int md_temporal_attention_fwd_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int NB, int F, int HW,
                                             int H, int D, float scale, void* stream) {
  MD_CHECK_ARG(F >= 1 && F <= 32, "md_temporal_attention_fwd: F=%d frames, the positional-encoding table holds 32", F);
  MD_CHECK_ARG(D % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "md_temporal_attention_fwd: D and strides must be multiples of 8");
  MD_CHECK_ARG(H * F <= 256, "md_temporal_attention_fwd: H*F=%d exceeds the 256-thread workgroup", H * F);
  TemporalParams p;
  p.Q = (const half_t*)Q; p.K = (const half_t*)K; p.V = (const half_t*)V; p.O = (half_t*)O;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.NB = NB; p.F = F; p.HW = HW; p.H = H; p.D = D;
  p.pix_per_block = 256 / (H * F);
  p.scale_log2 = scale * 1.4426950408889634f;
  const int grid = cdiv((long)NB * HW, p.pix_per_block);
  hipStream_t st = (hipStream_t)stream;
  if (F <= 4) hipLaunchKernelGGL(temporal_attn_kernel<4>, dim3(grid), dim3(256), 0, st, p);
  else if (F <= 8) hipLaunchKernelGGL(temporal_attn_kernel<8>, dim3(grid), dim3(256), 0, st, p);
  else if (F <= 16) hipLaunchKernelGGL(temporal_attn_kernel<16>, dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(temporal_attn_kernel<32>, dim3(grid), dim3(256), 0, st, p);
  MD_CHECK_LAUNCH("md_temporal_attention_fwd");
  return MD_OK;
}
