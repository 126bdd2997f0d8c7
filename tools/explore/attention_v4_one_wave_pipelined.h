// NOT PART OF THE BUILD -- kept as the record of a measured dead end (profiles/r04_ab_attention_one_wave_pipelined.log): this kernel
// was included from attention.hip for one GPU call in round 4, ran at 735-750 TFLOP/s against attn2_kernel's 813-815 on the same box,
// and still failed parity (6 of 39 attention tests) when the timing settled the question.  Do not ship it; read it for the structure.
// attn4_kernel<40>: the d = 40 self-attention of the 96 x 96 / 128 x 128 level (Lq = Lk = 9216 / 16384: 20 % of the clip), round 4.
//
// Same arithmetic, layouts and LDS images as attn2_kernel<40> (attention_v2.h: kappa-permuted K rows, ones row in V^T, softmax
// reference folded into the spare k-slots of the last Q K^T step, lazy rescale on "some P >= 2").  What changes is WHO overlaps
// the exponentials with the matrix pipe.  attn2 runs four waves per SIMD, each one a straight line QK^T -> 32 exp + 16 pack -> PV
// (the compiler emits the 32 v_exp_f32 back to back, no MFMA between them), and relies on the hardware to interleave the waves;
// the counters say it barely does: per 64-key tile and wave 448 matrix-pipe cycles + ~385 VALU cycles = 833 against 829 measured
//  -- the two pipes of a SIMD take turns.  Here ONE wave per SIMD owns TWO 32-query tiles and is software pipelined
// against itself, issue order pinned by hand (sched_barrier(0) lets nothing cross):
//
//   iteration t:   PV of tile t-1 (16 MFMAs, P(t-1))  ||  exp2 + pack of tile t (64 v_exp_f32, 32 v_cvt_pk: P(t))
//                  QK^T of tile t+1 (12 MFMAs, S(t+1) overwrites S(t) sub-tile by sub-tile as its exponentials retire)
//
// in four phases  [PV k-steps 0,1 | exp of sub-tile 0] [QK^T sub-tile 0 | exp of sub-tile 1, first half] [PV k-steps 2,3 | second
// half] [QK^T sub-tile 1], so that every MFMA has 2-6 independent VALU instructions of the OTHER stage behind it and the S
// accumulators need no second copy.  Two q-tiles per wave halve the fragment reads per MFMA (K / V^T fragments are shared), the
// fragments of a phase are read one phase ahead, and the DMA rings are three deep so that the tile pair a barrier certifies is the
// NEXT iteration's: no load latency is exposed after the barrier.  The rare rescale keeps attn2's semantics (reference 4 octaves
// above the running maximum, fp16-representable, P recomputed from the scores it still holds); it is checked once per 32-key
// sub-tile, before that sub-tile's S registers are handed to the next QK^T.
//
// Eligibility (launcher): D == 40, Lk % 64 == 0, Lk >= 1024, Lq % 256 == 0; everything else stays on attn2_kernel.
#pragma once
#include "common.h"

template <int N>
__device__ __forceinline__ void a4_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int D>
__global__ __launch_bounds__(256, 1) void attn4_kernel(AttnParams p) {
  static_assert(D == 40, "attn4_kernel is the d = 40 flavour (KS = 3 with 8 spare k-slots, two 32-row O^T tiles)");
  constexpr int KS = 3;
  constexpr int KROWB = D * 2;                   // 80 bytes per K row
  constexpr int KBYTES = 64 * KROWB;             // 5 KiB: five 1-KiB DMA pieces
  constexpr int VBYTES = 64 * 128;               // 64 rows (40 channels, the ones row, zeros) x 64 keys
  constexpr int VOFF = 3 * KBYTES;               // rings: K slots 0..2, then V^T slots 0..2
  constexpr int CONST_OFF = VOFF + 3 * VBYTES;   // {1, 0, .., 0}: the k-slots 40..47 of every K row
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  // XCD-aware placement as in attn2: all q-blocks of a (batch, head) pair on one XCD (its K / V^T stay in that L2)
  const int nqb = p.Lq >> 8;
  int pair, qblk;
  {
    const int L = blockIdx.x, npair = p.B * p.H;
    if ((npair & 7) == 0) {
      const int xcd = L & 7, slot = L >> 3;
      pair = xcd + 8 * (slot / nqb);
      qblk = slot - (slot / nqb) * nqb;
    } else {
      pair = L / nqb;
      qblk = L - pair * nqb;
    }
  }
  const int b = pair / p.H, h = pair - b * p.H;
  const int kb = p.kv_index ? p.kv_index[b] : b;
  const int q0 = qblk * 256 + wave * 64;
  const half_t* Kb = p.K + (size_t)kb * p.kv_stride * p.ldk + h * D;
  const half_t* Vb = p.Vt + (size_t)h * D * p.ldvt + (size_t)kb * p.kv_stride;
  const int ntiles = p.Lk >> 6;

  // ---- LDS images that no DMA writes: rows 40..63 of every V^T slot (row 40 = ones: softmax denominator), the whole of slot 2
  // (it stands for "tile -1": P(-1) = 0 times finite numbers), the constant K block
  for (int i = tid; i < 3 * 24 * 8; i += 256) {
    const int st = i / (24 * 8), rem = i - st * (24 * 8);
    const int row = 40 + rem / 8, slot = rem & 7;
    const half_t v = row == 40 ? (half_t)1.0f : (half_t)0.0f;
    const half8_t w = {v, v, v, v, v, v, v, v};
    *reinterpret_cast<half8_t*>(smem + VOFF + st * VBYTES + row * 128 + slot * 16) = w;
  }
  for (int i = tid; i < 40 * 8; i += 256) {
    const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<half8_t*>(smem + VOFF + 2 * VBYTES + i * 16) = z;
  }
  if (tid == 0) {
    const half8_t w = {(half_t)1.0f, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<half8_t*>(smem + CONST_OFF) = w;
  }

  // ---- DMA pieces of this wave.  K tile: pieces 0..4 (wave w: w and, for w == 0, 4); V^T tile: pieces 0..4 (wave 3 - w: w ... so
  // that every wave issues 2 or 3 pieces per iteration).  Per-lane sources of tile 0 + a per-tile byte step.
  const int nk_mine = wave == 0 ? 2 : 1, nv_mine = wave == 3 ? 2 : 1;
  const char* ksrc[2];
  const char* vsrc[2];
  int kdst[2], vdst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = i == 0 ? wave : 4;
    const int o = q * 1024 + lane * 16;
    const int row = o / KROWB, cb = o - row * KROWB;
    ksrc[i] = reinterpret_cast<const char*>(Kb + (size_t)row * p.ldk) + cb;
    kdst[i] = q * 1024;
    const int qv = i == 0 ? 3 - wave : 4;
    const int ov = qv * 1024 + lane * 16;
    const int dv = ov >> 7, ps = (ov & 127) >> 4;
    vsrc[i] = reinterpret_cast<const char*>(Vb + (size_t)dv * p.ldvt + ((ps ^ ((dv >> 1) & 7)) << 3));
    vdst[i] = qv * 1024;
  }
  const long kstep = (long)64 * p.ldk * 2;
  auto issue_k = [&](int tile, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < nk_mine) __builtin_amdgcn_global_load_lds((gptr_t)(ksrc[i] + tile * kstep), (lptr_t)(smem + slot * KBYTES + kdst[i]), 16, 0, 0);
  };
  auto issue_v = [&](int tile, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < nv_mine) __builtin_amdgcn_global_load_lds((gptr_t)(vsrc[i] + tile * 128), (lptr_t)(smem + VOFF + slot * VBYTES + vdst[i]), 16, 0, 0);
  };
  issue_k(0, 0);
  issue_k(1, 1);
  issue_k(2, 2);
  issue_v(0, 0);

  // ---- Q fragments (pre-multiplied by scale * log2 e); slot 40 of a row (lane half 1, k-step 2, element 0) will carry -m
  const float sc = p.scale_log2;
  half8_t qf[2][KS];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const half_t* Qp = p.Q + ((size_t)b * p.Lq + q0 + u * 32 + ql) * p.ldq + h * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = s * 16 + hi * 8;
      half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < D) v = *reinterpret_cast<const half8_t*>(Qp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * sc);
      qf[u][s] = v;
    }
  }

  // per-lane LDS offsets of the fragment reads (within a slot)
  const int krow = (ql & ~12) | ((ql & 4) << 1) | ((ql & 8) >> 1);       // K row read by MFMA row ql: key kappa(ql)
  const int vsw = (ql >> 1) & 7;
  int koff[2][KS], voff[2][4];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int k = 0; k < KS; ++k)
      koff[sub][k] = (k == KS - 1 && hi) ? CONST_OFF : (sub * 32 + krow) * KROWB + (k * 16 + hi * 8) * 2;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) voff[t][k] = VOFF + (t * 32 + ql) * 128 + (((k * 2 + hi) ^ vsw) << 4);
  auto kread = [&](int slot, int sub, int k) {
    return *reinterpret_cast<const half8_t*>(smem + ((k == KS - 1 && hi) ? 0 : slot * KBYTES) + koff[sub][k]);
  };
  auto vread = [&](int slot, int t, int k) { return *reinterpret_cast<const half8_t*>(smem + slot * VBYTES + voff[t][k]); };

  floatx16 s[2][2], o[2][2];
  half8_t pa[2][4], pb[2][4];
  float m_run[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][t][r] = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < 4; ++c) pa[u][c] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};

  a4_wait<0>();
  __syncthreads();

  // ---- prologue: S(0), and the reference of every row = its maximum over tile 0 + 4 octaves
  {
    half8_t kf[2][KS];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int k = 0; k < KS; ++k) kf[sub][k] = kread(0, sub, k);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][sub][r] = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) s[u][sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][k], qf[u][k], s[u][sub], 0, 0, 0);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float mloc = fmaxf(s[u][0][0], s[u][1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[u][0][r]), s[u][1][r]);
      mloc = a2_xhalf_max(mloc) + 4.0f;
      const float m_new = (float)(half_t)fminf(fmaxf(mloc, -60000.f), 60000.f);
      m_run[u] = m_new;
      if (hi) qf[u][KS - 1][0] = (half_t)(-m_new);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][sub][r] -= m_new;
    }
  }

  // The S accumulators must live in the VGPR half (the exponentials read them): with 512 registers per lane the compiler selects the
  // AGPR form for every MFMA and copies the 64 scores of a tile back with v_accvgpr_read (+50 % VALU).  These two are the VGPR form,
  // written out; their results are first read a whole phase (>= 6 MFMAs) later, far beyond the matrix pipe's write-back hazard.
#define A4_MFMA_S0(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "v"(B));
#define A4_MFMA_S(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B));
  // exp2 + pack of registers r0 .. r0+3 of S sub-tile `sub` of q-tile u into the P fragment set pc
#define A4_EXP4(PC, U, SUB, R0)                                                                                            \
  {                                                                                                                        \
    /* packed at once (asm: the compiler otherwise sinks all conversions of a phase behind its last MFMA and keeps the 32-bit  \
       exponentials alive until then: 30 registers that push the S accumulators out of the VGPR half) */                  \
    const float e0 = __builtin_amdgcn_exp2f(s[U][SUB][(R0)]), e1 = __builtin_amdgcn_exp2f(s[U][SUB][(R0) + 1]);            \
    const float e2 = __builtin_amdgcn_exp2f(s[U][SUB][(R0) + 2]), e3 = __builtin_amdgcn_exp2f(s[U][SUB][(R0) + 3]);        \
    unsigned w0, w1;                                                                                                       \
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w0) : "v"(e0), "v"(e1));                                             \
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w1) : "v"(e2), "v"(e3));                                             \
    uint4v pw = __builtin_bit_cast(uint4v, PC[U][(SUB) * 2 + ((R0) >> 3)]);                                                \
    pw[((R0) & 7) >> 1] = w0;                                                                                              \
    pw[(((R0) & 7) >> 1) + 1] = w1;                                                                                        \
    PC[U][(SUB) * 2 + ((R0) >> 3)] = __builtin_bit_cast(half8_t, pw);                                                      \
  }
  // "some P >= 2 (or inf / nan)" over the two fragments of a sub-tile, both q-tiles: bit 14 of an fp16
  auto trig = [&](const half8_t (&pc)[2][4], int sub) {
    const uint4v x = __builtin_bit_cast(uint4v, pc[0][sub * 2]) | __builtin_bit_cast(uint4v, pc[0][sub * 2 + 1]) |
                     __builtin_bit_cast(uint4v, pc[1][sub * 2]) | __builtin_bit_cast(uint4v, pc[1][sub * 2 + 1]);
    return __any(((x[0] | x[1] | x[2] | x[3]) & 0x40004000u) != 0);
  };

  // One iteration.  PP = P(t-1) (read by PV), PC = P(t) (written).  K(t+1) in K slot kuse, V^T(t-1) in V slot vuse.
  int kuse = 1, vuse = 2, kfill = 0, vfill = 1;        // slots: in use this iteration / refilled behind this iteration's barrier
  half8_t vf[2][2], kf[KS];
  // fragments of the first phase of iteration 0 (certified by the prologue's barrier)
#pragma unroll
  for (int t = 0; t < 2; ++t) vf[0][t] = vread(vuse, t, 0);

#define A4_BODY(PP, PC, T)                                                                                                    \
  {                                                                                                                           \
    a4_wait<0>();                            /* this wave's pieces of K(t+2), V(t): issued one whole iteration ago */         \
    __builtin_amdgcn_s_barrier();                                                                                             \
    if ((T) + 3 < ntiles) issue_k((T) + 3, kfill);                                                                            \
    if ((T) + 1 < ntiles) issue_v((T) + 1, vfill);                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    /* ---- phase 1: PV k-steps 0, 1  ||  exp of S sub-tile 0 (8 groups of 4 registers) */                                    \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                                           \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int t = 0; t < 2; ++t) {                           \
        o[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[k & 1][t], PP[u][k], o[u][t], 0, 0, 0);                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        const int g = k * 4 + u * 2 + t;                                                                                      \
        if (g < 2) { vf[(k + 1) & 1][g] = vread(vuse, g, k + 1); }                  /* V^T fragments of k-step 1 */           \
        if (g >= 4 && g < 6) { vf[(k + 1) & 1][g - 4] = vread(vuse, g - 4, k + 1); } /* ... of k-step 2 (phase 3) */          \
        if (g == 2 || g == 3 || g == 6) { kf[g == 6 ? 2 : g - 2] = kread(kuse, 0, g == 6 ? 2 : g - 2); }                      \
        A4_EXP4(PC, g >> 2, 0, (g & 3) * 4)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
      }                                                                                                                       \
    }                                                                                                                         \
    if (__builtin_expect(trig(PC, 0), 0)) {                                                                                                     \
      /* rare: some score of tile t sits >= 5 octaves above its row's reference.  S(t) is still whole: move the references,   \
         rescale what was accumulated under the old ones (O, and the half of P(t-1) that PV has not consumed yet) */          \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                         \
        float mloc = fmaxf(s[u][0][0], s[u][1][0]);                                                                           \
        _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[u][0][r]), s[u][1][r]);                     \
        mloc = a2_xhalf_max(mloc) + 4.0f;                                                                                     \
        const float m_new = (float)(half_t)fminf(fmaxf(m_run[u] + fmaxf(mloc, 0.f), -60000.f), 60000.f);                      \
        const float d = m_new - m_run[u];                                                                                     \
        m_run[u] = m_new;                                                                                                     \
        if (hi) qf[u][KS - 1][0] = (half_t)(-m_new);                                                                          \
        const float alpha = __builtin_amdgcn_exp2f(-d);                                                                       \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int r = 0; r < 16; ++r) o[u][t][r] *= alpha;      \
        _Pragma("unroll") for (int c = 2; c < 4; ++c) _Pragma("unroll") for (int e = 0; e < 8; ++e)                           \
            PP[u][c][e] = (half_t)((float)PP[u][c][e] * alpha);                                                               \
        _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) _Pragma("unroll") for (int r = 0; r < 16; ++r) s[u][sub][r] -= d;  \
        _Pragma("unroll") for (int r0 = 0; r0 < 16; r0 += 4) A4_EXP4(PC, u, 0, r0)                                            \
      }                                                                                                                       \
    }                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    /* ---- phase 2: QK^T of tile t+1, sub-tile 0 (its S registers are free now)  ||  exp of sub-tile 1, groups 0..3 */       \
    _Pragma("unroll") for (int k = 0; k < KS; ++k) _Pragma("unroll") for (int u = 0; u < 2; ++u) {                            \
      if (k == 0) { A4_MFMA_S0(s[u][0], kf[k], qf[u][k]) } else { A4_MFMA_S(s[u][0], kf[k], qf[u][k]) }                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                                      \
      const int g = k * 2 + u;                                                                                                \
      if (g == 1 || g == 2) { vf[1][g - 1] = vread(vuse, g - 1, 3); }               /* V^T fragments of k-step 3 */           \
      if (g < 4) A4_EXP4(PC, 0, 1, g * 4)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                                      \
    }                                                                                                                         \
    /* ---- phase 3: PV k-steps 2, 3  ||  exp of sub-tile 1, q-tile 1; K fragments of sub-tile 1 */                           \
    _Pragma("unroll") for (int k = 2; k < 4; ++k) {                                                                           \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int t = 0; t < 2; ++t) {                           \
        o[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[k & 1][t], PP[u][k], o[u][t], 0, 0, 0);                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        const int g = (k - 2) * 4 + u * 2 + t;                                                                                \
        if (g >= 4 && g < 4 + KS) { kf[g - 4] = kread(kuse, 1, g - 4); }                                                      \
        if (g < 4) A4_EXP4(PC, 1, 1, g * 4)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
      }                                                                                                                       \
    }                                                                                                                         \
    if (__builtin_expect(trig(PC, 1), 0)) {                                                                                                     \
      /* rare: as above for sub-tile 1.  Sub-tile 0's registers already hold S(t+1) (computed under the old references) and   \
         P(t) sub-tile 0 is final: both move with the reference; PV of tile t-1 is complete */                               \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                         \
        float mloc = s[u][1][0];                                                                                              \
        _Pragma("unroll") for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[u][1][r]);                                        \
        mloc = a2_xhalf_max(mloc) + 4.0f;                                                                                     \
        const float m_new = (float)(half_t)fminf(fmaxf(m_run[u] + fmaxf(mloc, 0.f), -60000.f), 60000.f);                      \
        const float d = m_new - m_run[u];                                                                                     \
        m_run[u] = m_new;                                                                                                     \
        if (hi) qf[u][KS - 1][0] = (half_t)(-m_new);                                                                          \
        const float alpha = __builtin_amdgcn_exp2f(-d);                                                                       \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int r = 0; r < 16; ++r) o[u][t][r] *= alpha;      \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) _Pragma("unroll") for (int e = 0; e < 8; ++e)                           \
            PC[u][c][e] = (half_t)((float)PC[u][c][e] * alpha);                                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { s[u][0][r] -= d; s[u][1][r] -= d; }                                   \
        _Pragma("unroll") for (int r0 = 0; r0 < 16; r0 += 4) A4_EXP4(PC, u, 1, r0)                                            \
      }                                                                                                                       \
    }                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    /* slots of the next iteration; its first V^T fragments are read behind the MFMAs below (certified by THIS barrier) */    \
    kuse = kuse == 2 ? 0 : kuse + 1;                                                                                          \
    vuse = vuse == 2 ? 0 : vuse + 1;                                                                                          \
    kfill = kfill == 2 ? 0 : kfill + 1;                                                                                       \
    vfill = vfill == 2 ? 0 : vfill + 1;                                                                                       \
    /* ---- phase 4: QK^T of tile t+1, sub-tile 1 */                                                                          \
    _Pragma("unroll") for (int k = 0; k < KS; ++k) _Pragma("unroll") for (int u = 0; u < 2; ++u) {                            \
      if (k == 0) { A4_MFMA_S0(s[u][1], kf[k], qf[u][k]) } else { A4_MFMA_S(s[u][1], kf[k], qf[u][k]) }                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                                      \
      const int g = k * 2 + u;                                                                                                \
      if (g >= 4) { vf[0][g - 4] = vread(vuse, g - 4, 0); }                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                      \
    }                                                                                                                         \
  }

  int t = 0;
#pragma unroll 1
  for (; t + 1 < ntiles; t += 2) {
    A4_BODY(pa, pb, t)
    A4_BODY(pb, pa, t + 1)
  }
  if (t < ntiles) A4_BODY(pa, pb, t)
  // ---- PV of the last tile: its V^T slot is `vuse` (certified by the last barrier), vf[0] holds k-step 0
#define A4_TAIL(PP)                                                                                   \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                     \
    if (k + 1 < 4) { _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) vf[(k + 1) & 1][tt] = vread(vuse, tt, k + 1); } \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)      \
        o[u][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[k & 1][tt], PP[u][k], o[u][tt], 0, 0, 0); \
  }
  if (ntiles & 1) { A4_TAIL(pb) } else { A4_TAIL(pa) }
  a4_wait<0>();
#undef A4_TAIL
#undef A4_BODY
#undef A4_EXP4
#undef A4_MFMA_S
#undef A4_MFMA_S0

  // ---- normalise by the ones row (row 40 = row 8 of O^T tile 1) and store
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float inv = 1.0f / __shfl(o[u][1][4], ql, 64);     // row 8 of the tile: register (8 & 3) + 4 (8 >> 3) = 4, lane half 0
    half_t* Op = p.O + ((size_t)b * p.Lq + q0 + u * 32 + ql) * p.ldo + h * D;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = tt * 32 + 8 * g + 4 * hi;
        if (dv < D) {
          const half4_t ov = {(half_t)(o[u][tt][4 * g] * inv), (half_t)(o[u][tt][4 * g + 1] * inv), (half_t)(o[u][tt][4 * g + 2] * inv),
                              (half_t)(o[u][tt][4 * g + 3] * inv)};
          *reinterpret_cast<half4_t*>(Op + dv) = ov;
        }
      }
  }
}

template <int D>
static bool attn4_eligible(const AttnParams& p) {
  static const int on = md_env_int("MD_ATTN_V4", 1);
  return on && D == 40 && p.Lk % 64 == 0 && p.Lk >= 1024 && p.Lq % 256 == 0;
}

template <int D>
static int launch_attn4(const AttnParams& p, hipStream_t stream) {
  constexpr int smem = 3 * 64 * D * 2 + 3 * 64 * 128 + 16;
  md_ensure_dynamic_lds<attn4_kernel<D>>(smem);
  const dim3 grid((p.Lq >> 8) * p.H * p.B);
  hipLaunchKernelGGL((attn4_kernel<D>), grid, dim3(256), smem, stream, p);
  MD_CHECK_LAUNCH("md_attention_fwd");
  return MD_OK;
}
