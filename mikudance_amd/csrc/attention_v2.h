// attn2_kernel<D>: the fast path of md_attention_fwd_f16 (aligned K / V^T: every shape of the UNets at latent sizes that
// are multiples of 8; the cross-attention context is padded to a multiple of 8 tokens).  Same math and layouts as
// attn_kernel<D> (attention.hip), restructured around what limited that kernel on MI355X (VALU work and stalls, not MFMA):
//
//  * K and V^T tiles (64 keys) go HBM/L2 -> LDS by direct-to-LDS DMA into a 2-deep ring with counted vmcnt waits and one
//    raw s_barrier per tile (no VGPR staging, two tiles in flight).
//  * The K rows fed to MFMA row index i are the keys kappa(i) = i with bits 2 and 3 swapped.  With that choice the
//    32x32 C layout of S^T leaves every lane holding, per 16-key step, EIGHT CONSECUTIVE keys, so the P fragment pairs
//    with ONE ds_read_b128 of a plain row-major V^T tile (16-B slots XOR-swizzled on the DMA source side).
//  * The softmax denominator is produced by the matrix core: a constant row of ones appended to V^T (free whenever
//    D % 32 != 0: head dims 40 and 80 pad to 64 / 96 rows anyway) makes row D of O^T accumulate sum_k P[k] in fp32 from
//    the same fp16-rounded P as the numerator.
//  * The O^T rescale is skipped unless some row's running maximum grows by more than 2^5 (wave-uniform test, exact
//    arithmetic otherwise: P <= 32 in fp16 keeps its relative precision).
//  * P is packed with v_cvt_pk_f16_f32 (round to nearest).
//  * Head dims with D % 16 == 8 (8, 40) leave eight unused k-slots in the last Q K^T step.  The matrix core then also does
//    the "scale, subtract the running reference" of the softmax: Q is pre-multiplied by scale*log2(e), slot D of every Q row
//    holds -m (the row's running reference, kept fp16-representable so that the product is exact) and slot D of every K row
//    reads a constant 1.0 from LDS, so S' = log2e*scale*q.k - m comes straight out of the MFMA and the per-score work is
//    exp2 + pack only (measured on MI355X: this kernel is bound by VALU issue -- 72 % busy vs 47 % for the MFMA pipe --
//    and the fma was 32 of its ~125 VALU instructions per 64-key tile).  m moves only in the (rare) rescale branch.
//    Round 2: the reference sits 4 octaves above the running maximum and the "has the maximum grown by > 2^5" test is one OR
//    over the packed P registers (bit 14 of an fp16 = "P >= 2") instead of a 32-input maximum per tile (+3.5 % same-box; the
//    maximum is computed on the rare path only; build with -DA2_NO_ORCHECK for the previous form).
#pragma once
#include "common.h"
#include <stdlib.h>

#define A2_KT 64
#define A2_THR 5.0f

template <int N>
__device__ __forceinline__ void a2_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void a2_wait_dyn(int n) {
  switch (n) {
    case 0: a2_wait<0>(); break;
    case 1: a2_wait<1>(); break;
    case 2: a2_wait<2>(); break;
    case 3: a2_wait<3>(); break;
    case 4: a2_wait<4>(); break;
    case 5: a2_wait<5>(); break;
    case 6: a2_wait<6>(); break;
    case 8: a2_wait<8>(); break;
    case 10: a2_wait<10>(); break;
    case 12: a2_wait<12>(); break;
    case 16: a2_wait<16>(); break;
    case 20: a2_wait<20>(); break;
    default: a2_wait<0>(); break;
  }
}

__device__ __forceinline__ float a2_xhalf_max(float x) {
  // max of the two lane halves (lane l <-> l^32) without going through LDS: v_permlane32_swap
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int D, int NST, int QT, int WPS, int NW = 4>
__global__ __launch_bounds__(64 * NW, WPS) void attn2_kernel(AttnParams p) {
  // NW waves per workgroup, 32*QT queries each: every K / V^T tile a workgroup DMAs into LDS is shared by all NW waves, so a
  // 16-wave workgroup (one per CU at 4 waves per SIMD) fetches and issues a quarter of the DMA pieces of four 4-wave ones
  constexpr int KS = (D + 15) / 16;            // k-steps of Q K^T
  constexpr bool ONES = (D % 32) != 0;         // room for the ones row in the last O^T tile
  constexpr int DVT = (D + 31) / 32;           // 32-row tiles of O^T
  constexpr int KROWB = D * 2;                 // bytes per K row in LDS (unpadded: DMA image is lane linear)
  // K-row swizzle (round 4).  A `ds_read_b128` is served in four groups of 16 lanes, conflict-free when the 16 addresses fall
  // into 16 different 16-byte slots of the 256-byte bank row (MI355X guide, LDS).  The 16 rows a group reads are a complete
  // residue system mod 16 (kappa maps each group's row set onto itself) and a row is D/8 slots long: 5 at d = 40 (odd: 5r mod 16
  // already takes 16 values), 10 at d = 80 (2-way conflicts), 20 at d = 160 (4r mod 16: 4-way, a K fragment read costs 16 LDS
  // cycles instead of 4).  With D/8 = 2^E * odd, slot c of row r is stored at c ^ g(r), g(r) = the top E bits of r mod 16: the
  // DMA applies the XOR to its per-lane SOURCE address (the LDS image stays lane linear), the fragment read to its LDS address.
  // SQ_LDS_BANK_CONFLICT per launch: d = 160 1.1e7 -> 0, d = 80 2.7e7 -> 0 (profiles/r04_ab_attention_k_swizzle.log).
#ifdef A2_NO_KSWIZZLE
  constexpr int KSW_E = 0;
#else
  constexpr int KSW_E = ((D / 8) % 2) ? 0 : ((D / 8) % 4) ? 1 : ((D / 8) % 8) ? 2 : 3;
#endif
  static_assert(KSW_E == 0 || (D % 16) != 8, "a k-step shared by K columns and the folded constant: unswizzled flavours only");
  auto ksw = [](int row) { return KSW_E ? (row >> (4 - KSW_E)) & ((1 << KSW_E) - 1) : 0; };
  constexpr int KBYTES = A2_KT * KROWB;
  constexpr int VBYTES = DVT * 32 * 128;       // V^T rows of 64 keys = 128 B
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int NKI = KBYTES / 1024, NVI = (D * 128) / 1024;  // DMA instructions per tile (K, V^T)
  constexpr int NI = NKI + NVI;
#ifdef A2_NO_FOLD
  constexpr bool FOLD = false;
#else
  constexpr bool FOLD = (D % 16) == 8;         // softmax reference folded into the last k-step (see header)
#endif
  constexpr int CONST_OFF = NST * STAGE;       // FOLD: two 16-B blocks {1,0,...,0}, 32 K rows apart (one per 32-key sub-tile)
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  // XCD-aware mapping: workgroup id L runs on XCD L % 8 (observed dispatch order, speed only).  All q-blocks of one
  // (batch, head) pair go to ONE XCD so that its K / V^T tiles are fetched into a single private L2 instead of all
  // eight (measured: 8x the algorithmic K/V bytes at the fabric otherwise).
  const int nqb = (p.Lq + 32 * NW * QT - 1) / (32 * NW * QT);
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
  const int q0 = qblk * (32 * NW * QT) + wave * (32 * QT);
  const int lk8 = (p.Lk + 7) & ~7;

  // (token-major Q / K: an 80-byte head slice straddles 128-byte lines, 2.8x the algorithmic fetch at d = 40.  A head-major build was
  // measured in round 5 -- Q, K as [batch][head][token][D]: FETCH_SIZE 1.68 -> 0.87 GB per launch, time -1.1 % -- and not adopted:
  // profiles/r05_ab_attention_head_major.log)
  const half_t* Kb = p.K + (size_t)kb * p.kv_stride * p.ldk + h * D;
  const half_t* Vb = p.Vt + (size_t)h * D * p.ldvt + (size_t)kb * p.kv_stride;

  // constant rows of every V^T stage: ones in row D (softmax denominator), zeros in the rest of the padding
  if (D % 32 != 0) {
    for (int i = tid; i < NST * (DVT * 32 - D) * 8; i += 64 * NW) {
      const int st = i / ((DVT * 32 - D) * 8), rem = i % ((DVT * 32 - D) * 8);
      const int row = D + rem / 8, slot = rem % 8;
      const half_t v = row == D ? (half_t)1.0f : (half_t)0.0f;
      half8_t w = {v, v, v, v, v, v, v, v};
      *reinterpret_cast<half8_t*>(smem + st * STAGE + KBYTES + row * 128 + slot * 16) = w;
    }
  }

  if (FOLD && tid < 2) {
    half8_t w = {(half_t)1.0f, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<half8_t*>(smem + CONST_OFF + tid * 32 * KROWB) = w;
  }

  const float sc = p.scale_log2;
  half8_t qf[QT][KS];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qrow = min(q0 + u * 32 + ql, p.Lq - 1);
    const half_t* Qp = p.Q + ((size_t)b * p.Lq + qrow) * p.ldq + h * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = s * 16 + hi * 8;
      half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < D) v = *reinterpret_cast<const half8_t*>(Qp + c);
      if (FOLD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * sc);
      }
      qf[u][s] = v;
    }
  }

  // ---- DMA: instruction q of a tile (q < NKI: K image, else V^T image) is issued by wave q % NW
  const int n_mine = wave < NI ? (NI - wave + NW - 1) / NW : 0;
  // per-lane source of tile 0 and the per-tile byte step (K: 64 rows down, V^T: 64 keys = 128 bytes to the right), so that a
  // full tile costs one 64-bit add per DMA; only the ragged last tile recomputes clamped addresses
  constexpr int NQ = (NI + NW - 1) / NW;
  const char* src0[NQ];
  long step[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = qi * NW + wave;
    src0[qi] = reinterpret_cast<const char*>(Kb);
    step[qi] = 0;
    if (q < NKI) {
      const int o = q * 1024 + lane * 16;
      const int row = o / KROWB, cb = (o - row * KROWB) ^ (ksw(row) << 4);
      src0[qi] = reinterpret_cast<const char*>(Kb + (size_t)row * p.ldk) + cb;
      step[qi] = (long)A2_KT * p.ldk * 2;
    } else if (q < NI) {
      const int qv = q - NKI;
      const int o = qv * 1024 + lane * 16;
      const int dv = o >> 7, ps = (o & 127) >> 4;
      const int ls = ps ^ ((dv >> 1) & 7);
      src0[qi] = reinterpret_cast<const char*>(Vb + (size_t)dv * p.ldvt + ls * 8);
      step[qi] = A2_KT * 2;
    }
  }
  auto issue_tile = [&](int it_, int stage) {
    char* sb = smem + stage * STAGE;
    const int j0 = it_ * A2_KT;
    if (j0 + A2_KT <= p.Lk) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const int q = qi * NW + wave;
        if (q < NI)
          __builtin_amdgcn_global_load_lds((gptr_t)(src0[qi] + it_ * step[qi]), (lptr_t)(sb + (q < NKI ? q * 1024 : KBYTES + (q - NKI) * 1024)), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const int q = qi * NW + wave;
      if (q < NKI) {
        const int o = q * 1024 + lane * 16;
        const int row = o / KROWB, cb = (o - row * KROWB) ^ (ksw(row) << 4);
        const int key = min(j0 + row, p.Lk - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(Kb + (size_t)key * p.ldk) + cb), (lptr_t)(sb + q * 1024), 16, 0, 0);
      } else if (q < NI) {
        const int qv = q - NKI;
        const int o = qv * 1024 + lane * 16;
        const int dv = o >> 7, ps = (o & 127) >> 4;
        const int ls = ps ^ ((dv >> 1) & 7);
        const int j = min(j0 + ls * 8, lk8 - 8);
        __builtin_amdgcn_global_load_lds((gptr_t)(Vb + (size_t)dv * p.ldvt + j), (lptr_t)(sb + KBYTES + qv * 1024), 16, 0, 0);
      }
    }
  };

  floatx16 o[QT][DVT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m_run[u] = FOLD ? 0.f : NEG_BIG;
    l_run[u] = 0.f;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][t][r] = 0.f;
  }
  const int ntiles = (p.Lk + A2_KT - 1) / A2_KT;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < ntiles) issue_tile(s, s);

  // K row read by MFMA row index ql: key kappa(ql) = ql with bits 2 and 3 swapped
  const int krow = (ql & ~12) | ((ql & 4) << 1) | ((ql & 8) >> 1);
  const int vsw = (ql >> 1) & 7;  // V^T slot swizzle of row t*32 + ql: ((row >> 1) & 7), 32 | row offset keeps it
  const int kswz = ksw(krow);     // K slot swizzle of rows krow and 32 + krow
  int stage = 0;
  for (int it = 0; it < ntiles; ++it) {
    const int j0 = it * A2_KT;
    const int ahead = min(NST - 2, ntiles - 1 - it);
    a2_wait_dyn(ahead * n_mine);
    __builtin_amdgcn_s_barrier();
    if (it + NST - 1 < ntiles) {
      int st = stage + NST - 1;
      if (st >= NST) st -= NST;
      issue_tile(it + NST - 1, st);
    }
    const char* ks = smem + stage * STAGE;
    const char* vs = ks + KBYTES;

    // ---- S^T = K Q^T  (2 sub-tiles of 32 keys, MFMA row i <-> key kappa(i)); K fragments shared by the QT q-tiles
    floatx16 s[QT][2];
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][sub][r] = 0.f;
    if constexpr (D <= 40) {
      // all K fragments of the tile in flight before the first MFMA (one LDS latency per tile instead of three) and the two
      // 32-key accumulation chains alternate: +1 % at d = 40 on MI355X (same-box, with 8-wave workgroups); 24 more registers,
      // which the d = 80 / 160 flavours do not have
      half8_t kfr[2][KS];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const char* kp = ks + (sub * 32 + krow) * KROWB + (((k * 2 + hi) ^ kswz) << 4);
          if (FOLD && k == KS - 1) kp = hi ? smem + CONST_OFF + sub * 32 * KROWB : kp;
          kfr[sub][k] = *reinterpret_cast<const half8_t*>(kp);
        }
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int u = 0; u < QT; ++u) s[u][sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr[sub][k], qf[u][k], s[u][sub], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * KS, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * KS * QT, 0);
    } else {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const char* kp = ks + (sub * 32 + krow) * KROWB + (((k * 2 + hi) ^ kswz) << 4);
        if (FOLD && k == KS - 1) kp = hi ? smem + CONST_OFF + sub * 32 * KROWB : kp;   // k-slots D.. of every key: {1,0,..,0}
        const half8_t kf = *reinterpret_cast<const half8_t*>(kp);
#ifdef A2_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int u = 0; u < QT; ++u) s[u][sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[u][k], s[u][sub], 0, 0, 0);
#ifdef A2_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
    }
    }
    // register r of sub-tile `sub` in lane half `hi` holds key j0 + sub*32 + 16*(r>>3) + 8*hi + (r&7)
    if (j0 + A2_KT > p.Lk) {
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j0 + sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (key >= p.Lk) s[u][sub][r] = NEG_BIG;
          }
    }
    half8_t pf[QT][4];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
#ifndef A2_NO_ORCHECK
      if constexpr (FOLD) {
        // Reference r (slot D of Q) is kept A2_OFF = 4 octaves ABOVE the row's running maximum, so that P <= 2^-4 as long as the
        // maximum has not grown and "some P >= 2" -- bit 14 of an fp16, i.e. one OR over the packed P registers instead of a
        // 32-input maximum -- means "the maximum has grown by more than 2^5" (the lazy-rescale threshold; also catches inf / nan).
        // The maximum itself is only computed on the rare path, which then re-derives P from the scores it still holds.
        auto make_p = [&]() {
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              pf[u][sub * 2 + (r >> 3)][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[u][sub][r]);
              pf[u][sub * 2 + (r >> 3)][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(s[u][sub][r + 1]);
            }
#ifdef MD_DEGRADE_P8
          // DIAGNOSTIC build only (tools/build_ab.sh, never shipped): P keeps 8 of its 11 significant bits -- a deliberately degraded
          // kernel that the parity budgets of tests/parity_budget.py must catch (profiles/r06_parity_budget_degraded.log)
          typedef unsigned uint4d __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int t = 0; t < 4; ++t) pf[u][t] = __builtin_bit_cast(half8_t, __builtin_bit_cast(uint4d, pf[u][t]) & 0xfff8fff8u);
#endif
        };
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        const bool first = it == 0;
        bool trig = first;
        if (!first) {
          make_p();
          const uint4v ob = __builtin_bit_cast(uint4v, pf[u][0]) | __builtin_bit_cast(uint4v, pf[u][1]) |
                            __builtin_bit_cast(uint4v, pf[u][2]) | __builtin_bit_cast(uint4v, pf[u][3]);
          trig = __any(((ob[0] | ob[1] | ob[2] | ob[3]) & 0x40004000u) != 0);
        }
        if (trig) {
          float mloc = fmaxf(s[u][0][0], s[u][1][0]);
#pragma unroll
          for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[u][0][r]), s[u][1][r]);
          mloc = a2_xhalf_max(mloc) + 4.0f;                           // where the reference should sit, relative to the current one
          const float want = m_run[u] + (first ? mloc : fmaxf(mloc, 0.f));
          const float m_new = (float)(half_t)fminf(fmaxf(want, -60000.f), 60000.f);
          const float d = m_new - m_run[u];
          m_run[u] = m_new;
          if (hi) qf[u][KS - 1][0] = (half_t)(-m_new);
          if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[u][t][r] *= alpha;
          }
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][sub][r] -= d;
          make_p();
        }
        continue;
      }
#endif
      float mloc = fmaxf(s[u][0][0], s[u][1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[u][0][r]), s[u][1][r]);
      mloc = a2_xhalf_max(mloc);
      if constexpr (FOLD) {
        // s already is log2e*scale*q.k - m_run (m_run: fp16-representable reference sitting in slot D of the Q fragment)
        const bool first = it == 0;
        if (first || !__all(mloc <= A2_THR)) {
          const float want = m_run[u] + (first ? mloc : fmaxf(mloc, 0.f));
          const float m_new = (float)(half_t)fminf(fmaxf(want, -60000.f), 60000.f);
          const float d = m_new - m_run[u];
          m_run[u] = m_new;
          if (hi) qf[u][KS - 1][0] = (half_t)(-m_new);
          if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[u][t][r] *= alpha;
          }
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][sub][r] -= d;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            pf[u][sub * 2 + (r >> 3)][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[u][sub][r]);
            pf[u][sub * 2 + (r >> 3)][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(s[u][sub][r + 1]);
          }
        continue;
      }
      const float mt = mloc * sc;
      if (!__all(mt <= m_run[u] + A2_THR)) {
        const float m_new = fmaxf(m_run[u], mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
        m_run[u] = m_new;
        l_run[u] *= alpha;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[u][t][r] *= alpha;
      }
      float lsum = 0.f;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(s[u][sub][r] * sc - m_run[u]);
          const float p1 = __builtin_amdgcn_exp2f(s[u][sub][r + 1] * sc - m_run[u]);
          if (!ONES) lsum += p0 + p1;
          pf[u][sub * 2 + (r >> 3)][r & 7] = (half_t)p0;
          pf[u][sub * 2 + (r >> 3)][(r & 7) + 1] = (half_t)p1;
        }
      if (!ONES) l_run[u] += lsum;
    }
    // ---- O^T += V^T P^T : one ds_read_b128 per V^T fragment, shared by the QT q-tiles
#pragma unroll
    for (int t = 0; t < DVT; ++t) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const half8_t vf = *reinterpret_cast<const half8_t*>(vs + (t * 32 + ql) * 128 + (((k * 2 + hi) ^ vsw) << 4));
#ifdef A2_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int u = 0; u < QT; ++u) o[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u][k], o[u][t], 0, 0, 0);
#ifdef A2_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
    }
    if (++stage == NST) stage = 0;
  }

#pragma unroll
  for (int u = 0; u < QT; ++u) {
    float l_tot;
    if (ONES) {
      constexpr int rt = D % 32;                       // row of the ones inside the last tile; rt % 8 == 0 -> lane half 0
      constexpr int reg = (rt & 3) + 4 * (rt >> 3);
      l_tot = __shfl(o[u][DVT - 1][reg], ql, 64);
    } else {
      l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
    }
    const float inv = 1.0f / l_tot;
    const int qr = q0 + u * 32 + ql;
    if (qr < p.Lq) {
      half_t* Op = p.O + ((size_t)b * p.Lq + qr) * p.ldo + h * D;
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = t * 32 + 8 * g + 4 * hi;
          if (dv < D) {
            half4_t ov = {(half_t)(o[u][t][4 * g] * inv), (half_t)(o[u][t][4 * g + 1] * inv), (half_t)(o[u][t][4 * g + 2] * inv),
                          (half_t)(o[u][t][4 * g + 3] * inv)};
            *reinterpret_cast<half4_t*>(Op + dv) = ov;
          }
        }
    }
  }
}

template <int D, int QT, int NW = 4>
static int launch_attn2_qt(const AttnParams& p, hipStream_t stream) {
#ifdef A2_NST
  constexpr int NST = A2_NST;
#else
  constexpr int NST = 2;   // same-box A/B on MI355X: a 2-deep ring beats 3-deep by ~3 % at D = 40 (less LDS, same overlap)
#endif
  // occupancy targets that fit without spilling: D <= 40 -> 4 waves/SIMD (<= 128 registers), D <= 80 -> 3 (<= 168)
#if defined(A2_WPS6)
  constexpr int WPS = QT != 1 ? 1 : (D <= 40 ? 6 : (D <= 80 ? 3 : 1));
#elif defined(A2_WPS5)
  constexpr int WPS = QT != 1 ? 1 : (D <= 40 ? 5 : (D <= 80 ? 3 : 1));
#elif defined(A2_QT2_WPS2)
  constexpr int WPS = QT != 1 ? 2 : (D <= 40 ? 4 : (D <= 80 ? 3 : 1));
#else
  constexpr int WPS = QT != 1 ? 1 : (D <= 40 ? 4 : (D <= 80 ? 3 : 1));
#endif
  constexpr int DVT = (D + 31) / 32;
  constexpr int smem = NST * (A2_KT * D * 2 + DVT * 32 * 128) + ((D % 16) == 8 ? 32 * D * 2 + 16 : 0);   // + the FOLD constants
  md_ensure_dynamic_lds<attn2_kernel<D, NST, QT, WPS, NW>>(smem);
  dim3 grid(cdiv(p.Lq, 32 * NW * QT) * p.H * p.B);
  hipLaunchKernelGGL((attn2_kernel<D, NST, QT, WPS, NW>), grid, dim3(64 * NW), smem, stream, p);
  MD_CHECK_LAUNCH("md_attention_fwd");
  return MD_OK;
}

template <int D>
static int launch_attn2(const AttnParams& p, hipStream_t stream) {
  // Measured on MI355X (profiles/r02_ab_attention_*.log) and settled: ONE 32-row q-tile per wave (two tiles per wave leave one wave
  // per SIMD and the compiler does not interleave the two softmax chains: -4 %); at d <= 40 long self-attention runs 8 waves per
  // workgroup, which share each K / V^T tile (L = 9216: 8 waves 802-810 TFLOP/s, 16 waves 803, 4 waves 782).
  if constexpr (D <= 40) {
    if (p.Lq >= 1024) return launch_attn2_qt<D, 1, 8>(p, stream);
  }
  return launch_attn2_qt<D, 1>(p, stream);
}
