// Fused (flash-style) multi-head attention forward for gfx950: O = softmax(Q K^T * d^-1/2) V, no mask.
// Replaces diffusers Attention/AttnProcessor2_0 -> F.scaled_dot_product_attention as invoked by the reference at
// src/models/mutual_mix_attention.py:141-148 (write), :173-200 (mixed read), :213-220 (cross) and
// src/models/attention.py:263-269.  Head dims 40 / 80 / 160 (SD-1.5) plus 8 / 16 / 32 / 64 for reduced test geometry.
//
// Layouts (all fp16):  Q [B*Lq][ldq], K [nkv*kv_stride][ldk]  (head h at columns h*D..h*D+D-1),
//                      Vt [H*D][ldvt]  = V TRANSPOSED (row h*D+c, column kb*kv_stride + j) as written by md_gemm_f16
//                      with transpose_out=1, so a 64-key slice of one V column is 128 contiguous bytes,
//                      O [B*Lq][ldo].   kv_index (optional) maps the query batch b to its K/V batch kb.
//
// Workgroup = 4 waves x 32 query rows; K/V tiles of 64 keys are register-staged into a 2-deep LDS ring (one barrier
// per tile).  Per wave and tile:
//   S^T[key][q] = K . Q^T        v_mfma_f32_32x32x16_f16, A = K fragment from LDS, B = Q fragment held in VGPRs
//   online softmax               the 32x32 C layout puts one query column in lanes (q, q+32): max / sum are lane-local
//                                plus ONE cross-half exchange; O^T rescale is lane-local too
//   O^T[c][q] += V^T . P^T       A = V^T fragment (two ds_read_b64), B = P in registers: the C layout of S^T already
//                                is a valid B-fragment once V^T uses the same key permutation
//                                key(slot j, half hi) = 16t + (j&3) + 8(j>>2) + 4hi  -> no cross-lane traffic for P.
#include "common.h"

struct AttnParams {
  const half_t* Q;
  const half_t* K;
  const half_t* Vt;
  half_t* O;
  const int* kv_index;
  int ldq, ldk, ldvt, ldo;
  int B, H, Lq, Lk, kv_stride;
  float scale_log2;
};

#define KT 64
#define NEG_BIG (-1.0e30f)

template <int D>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
  constexpr int KS = (D + 15) / 16;  // k-steps of Q K^T
  constexpr int DQ = KS * 16;
  constexpr int DVT = (D + 31) / 32;  // 32-row tiles of O^T
  constexpr int KLD = DQ + 8;         // halfs per K row in LDS (16-B pad: conflict-free ds_read_b128)
  constexpr int VLD = KT + 4;         // halfs per V^T row in LDS (136 B: conflict-free ds_read_b64)
  constexpr int KBYTES = KT * KLD * 2;
  constexpr int VBYTES = DVT * 32 * VLD * 2;
  constexpr int CPK = D / 8;  // 16-B chunks per key row
  constexpr int NCHUNK = KT * CPK;
  constexpr int R = (NCHUNK + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kb = p.kv_index ? p.kv_index[b] : b;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = min(q0 + ql, p.Lq - 1);

  const half_t* Qp = p.Q + ((size_t)b * p.Lq + qrow) * p.ldq + h * D;
  const half_t* Kb = p.K + (size_t)kb * p.kv_stride * p.ldk + h * D;
  const half_t* Vb = p.Vt + (size_t)h * D * p.ldvt + (size_t)kb * p.kv_stride;

  // zero both LDS stages once: pad columns of K (D..DQ) and pad rows of V^T (D..32*DVT) must not hold NaN patterns
  for (int i = tid; i < 2 * (KBYTES + VBYTES) / 16; i += 256) reinterpret_cast<floatx4*>(smem)[i] = floatx4{0, 0, 0, 0};

  half8_t qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = s * 16 + hi * 8;
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < D) v = *reinterpret_cast<const half8_t*>(Qp + c);
    qf[s] = v;
  }

  half8_t rk[R], rv[R];
  auto load_tile = [&](int j0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int c = tid + r * 256;
      if (c < NCHUNK) {
        const int key = c / CPK, dc = c - key * CPK;
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (j0 + key < p.Lk) v = *reinterpret_cast<const half8_t*>(Kb + (size_t)(j0 + key) * p.ldk + dc * 8);
        rk[r] = v;
        const int dv = c >> 3, kc = c & 7, j = j0 + kc * 8;
        const half_t* src = Vb + (size_t)dv * p.ldvt + j;
        half8_t w = {0, 0, 0, 0, 0, 0, 0, 0};
        if (j + 8 <= p.Lk && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
          w = *reinterpret_cast<const half8_t*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (j + e < p.Lk) w[e] = src[e];
        }
        rv[r] = w;
      }
    }
  };
  auto store_tile = [&](int stage) {
    char* ks = smem + stage * (KBYTES + VBYTES);
    char* vs = ks + KBYTES;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int c = tid + r * 256;
      if (c < NCHUNK) {
        const int key = c / CPK, dc = c - key * CPK;
        *reinterpret_cast<half8_t*>(ks + key * (KLD * 2) + dc * 16) = rk[r];
        const int dv = c >> 3, kc = c & 7;
        half4_t lo = {rv[r][0], rv[r][1], rv[r][2], rv[r][3]}, hi4 = {rv[r][4], rv[r][5], rv[r][6], rv[r][7]};
        char* d = vs + dv * (VLD * 2) + kc * 16;
        *reinterpret_cast<half4_t*>(d) = lo;
        *reinterpret_cast<half4_t*>(d + 8) = hi4;
      }
    }
  };

  floatx16 o[DVT];
#pragma unroll
  for (int t = 0; t < DVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const float sc = p.scale_log2;

  const int ntiles = (p.Lk + KT - 1) / KT;
  load_tile(0);
  __syncthreads();  // zero-fill done before the first tile lands
  store_tile(0);
  __syncthreads();

  for (int it = 0; it < ntiles; ++it) {
    const int stage = it & 1;
    const int j0 = it * KT;
    if (it + 1 < ntiles) load_tile(j0 + KT);
    const char* ks = smem + stage * (KBYTES + VBYTES);
    const char* vs = ks + KBYTES;

    // ---- S^T = K Q^T  (2 sub-tiles of 32 keys)
    floatx16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(ks + (sub * 32 + ql) * (KLD * 2) + (k * 16 + hi * 8) * 2);
        s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[k], s[sub], 0, 0, 0);
      }
    }
    if (j0 + KT > p.Lk) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= p.Lk) s[sub][r] = NEG_BIG;
        }
    }
    // ---- online softmax (scores scaled by d^-1/2 * log2 e inside exp2)
    float mloc = NEG_BIG;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[sub][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc * sc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float lsum = 0.f;
    half8_t pf[4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(s[sub][r] * sc - m_new);
        lsum += pv;
        pf[sub * 2 + (r >> 3)][r & 7] = (half_t)pv;
      }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < DVT; ++t) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const char* a = vs + (t * 32 + ql) * (VLD * 2) + (k * 16 + 4 * hi) * 2;
        const half4_t lo = *reinterpret_cast<const half4_t*>(a);
        const half4_t hi4 = *reinterpret_cast<const half4_t*>(a + 16);
        const half8_t vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[k], o[t], 0, 0, 0);
      }
    }
    if (it + 1 < ntiles) store_tile(stage ^ 1);
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q0 + ql < p.Lq) {
    half_t* Op = p.O + ((size_t)b * p.Lq + q0 + ql) * p.ldo + h * D;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = t * 32 + 8 * g + 4 * hi;
        if (dv < D) {
          half4_t ov = {(half_t)(o[t][4 * g] * inv), (half_t)(o[t][4 * g + 1] * inv), (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
          *reinterpret_cast<half4_t*>(Op + dv) = ov;
        }
      }
  }
}

#include "attention_v2.h"
#include "attention_v2s.h"
#include <stdlib.h>

template <int D>
static int launch_attn(const AttnParams& p, hipStream_t stream) {
  // fast path: 16-byte aligned K / V^T key chunks (DMA granularity) and a readable pad up to the next multiple of 8 keys
  const bool fast = p.ldvt % 8 == 0 && p.kv_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(p.Vt) & 15) == 0 &&
                    ((p.Lk + 7) & ~7) <= p.kv_stride;
  if constexpr (D == 40) {
    // cross-attention at d = 40 (257 CLIP tokens = a handful of key tiles, Lq >= 2048): K / V^T resident in LDS, persistent walk
    // over the q-blocks (attention_v2s.h).  Same-box A/B on MI355X (profiles/r03_ab_attention_small.log): Lq = 9216 0.33-0.34 ->
    // 0.25-0.27 ms; d = 80 / 160 measured 2 % slower than the ring kernel and stay there.
    if (fast && attn2s_eligible<D>(p)) return launch_attn2s<D>(p, stream);
  }
  if (fast) return launch_attn2<D>(p, stream);
  constexpr int KS = (D + 15) / 16, DQ = KS * 16, DVT = (D + 31) / 32;
  constexpr int smem = 2 * (KT * (DQ + 8) * 2 + DVT * 32 * (KT + 4) * 2);
  md_ensure_dynamic_lds<attn_kernel<D>>(smem);
  dim3 grid(cdiv(p.Lq, 128), p.H, p.B);
  hipLaunchKernelGGL(attn_kernel<D>, grid, dim3(256), smem, stream, p);
  MD_CHECK_LAUNCH("md_attention_fwd");
  return MD_OK;
}

extern "C" int md_attention_fwd_f16(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo, const int* kv_index,
                                    int B, int H, int D, int Lq, int Lk, int kv_stride, float scale, void* stream) {
  MD_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "md_attention_fwd: empty problem");
  MD_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "md_attention_fwd: ldq/ldk must be multiples of 8, ldo of 4");
  MD_CHECK_ARG((reinterpret_cast<uintptr_t>(Q) & 15) == 0 && (reinterpret_cast<uintptr_t>(K) & 15) == 0 && (reinterpret_cast<uintptr_t>(O) & 7) == 0,
               "md_attention_fwd: Q/K must be 16-byte aligned, O 8-byte aligned");
  MD_CHECK_ARG(kv_stride >= Lk, "md_attention_fwd: kv_stride < Lk");
  AttnParams p;
  p.Q = (const half_t*)Q; p.K = (const half_t*)K; p.Vt = (const half_t*)Vt; p.O = (half_t*)O; p.kv_index = kv_index;
  p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.kv_stride = kv_stride;
  p.scale_log2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 8: return launch_attn<8>(p, st);
    case 16: return launch_attn<16>(p, st);
    case 32: return launch_attn<32>(p, st);
    case 40: return launch_attn<40>(p, st);
    case 64: return launch_attn<64>(p, st);
    case 80: return launch_attn<80>(p, st);
    case 160: return launch_attn<160>(p, st);
    default: md_set_error("md_attention_fwd: unsupported head dim %d (8,16,32,40,64,80,160)", D); return MD_ERR_ARG;
  }
}
