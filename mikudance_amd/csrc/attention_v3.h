// attn3_kernel<D>: "ping-pong" flavour of the flash attention fast path for head dims with D % 16 == 8 (d = 40: the 96x96 level
// of both UNets) on long keys (Lk % 64 == 0).  Same operand layouts, key permutation, ones-row denominator, lazy rescale and
// folded softmax reference as attn2_kernel (attention_v2.h) -- what changes is WHEN a wave does what.
//
// attn2 at d = 40 sits on both pipes at once (PMC: VALU ~ 69 % busy, MFMA ~ 57 %, only ~ 19 % of the time together): per 64-key
// tile a wave issues 14 MFMAs (448 cycles of the matrix pipe) and ~ 90 VALU instructions (max / exp2 / pack of the softmax, about
// the same issue time), and a wave cannot overlap its OWN two halves (the softmax needs the scores, P.V needs the softmax).
// With free-running waves the overlap between DIFFERENT waves of a SIMD is left to chance.  Here it is arranged:
//   * a workgroup is 8 waves = 2 groups of 4 (waves w and w + 4 share a SIMD); each wave owns 32 queries;
//   * time is cut into slots separated by ONE workgroup barrier; in every slot one group runs a MATRIX phase
//         M(t):  O^T += V^T(t-1) . P^T(t-1)   (8 MFMAs)   then   S^T(t) = K(t) . Q^T   (6 MFMAs)
//     while the other runs the VECTOR phase of its previous tile
//         V(t):  running reference / lazy rescale, P(t) = exp2(S'(t)) packed to fp16     (~ 90 VALU, no MFMA)
//     and they swap in the next slot, so every SIMD always holds one wave feeding the matrix pipe and one feeding the VALU:
//         slot        0      1      2      3      4    ...
//         group 0    M(0)   V(0)   M(1)   V(1)   M(2)
//         group 1     -     M(0)   V(0)   M(1)   V(1)
//   * LDS holds a 3-deep ring of PAIRS {K tile t, V^T tile t-1} (what M(t) reads), filled by direct-to-LDS DMA two pairs ahead;
//     pair t is read in slots 2t (group 0) and 2t+1 (group 1) and its stage refilled (pair t+3) after the barrier that opens
//     slot 2t+2.  Every wave issues the same number of DMA pieces for every pair (first / last pair: the missing V^T / K tile is
//     replaced by a clamped, unused one), so the counted vmcnt wait is a constant.
//
// MEASURED OUTCOME (MI355X, B = 32, H = 8, L = 9216, d = 40; profiles/r02_ab_attention_pingpong.log): correct (same tests as
// attn2), but SLOWER: 5.1 ms (680 TFLOP/s) against 4.4 ms (790) for attn2.  Phase-isolation builds show why: skeleton only
// (barriers + DMA) 1.15 ms, matrix phases only 3.08 ms, vector phases only 3.01 ms, both 5.1 ms -- i.e. the matrix phase of one
// wave and the vector phase of its SIMD partner do NOT overlap, they add up (pairing verified: the two wrong pairings cost
// another 15 %; s_setprio on either phase, AGPR accumulators and all-zero operands change nothing of that picture).  On this
// chip a wave's softmax VALU work is hidden only by interleaving it with that SAME wave's MFMAs (attn2's P.V block), not by a
// partner wave -- and attn2's counters (MFMA 57 % + VALU 69 % busy) say that even there the two pipes are together only ~ 20-25 %
// of the time: the floor is close to the SUM of MFMA and VALU time.  Kept as an opt-in flavour (MD_ATTN_PP=1) for the record;
// NOT on the default path.
#pragma once
#include <type_traits>

template <int D>
__global__ __launch_bounds__(512, 2) void attn3_kernel(AttnParams p) {
  static_assert(D % 16 == 8 && D % 32 != 0, "folded reference + ones row");
  constexpr int KS = (D + 15) / 16;            // k-steps of Q K^T
  constexpr int DVT = (D + 31) / 32;           // 32-row tiles of O^T
  constexpr int KROWB = D * 2;
  constexpr int KBYTES = A2_KT * KROWB;
  constexpr int VBYTES = DVT * 32 * 128;
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int NKI = KBYTES / 1024, NVI = (D * 128) / 1024, NI = NKI + NVI;
  static_assert(KBYTES % 1024 == 0 && (D * 128) % 1024 == 0, "whole DMA pieces");
  constexpr int NST = 3, NW = 8;
  constexpr int NQ = (NI + NW - 1) / NW;
  constexpr int CONST_OFF = NST * STAGE;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The two groups must pair the waves that SHARE a SIMD.  Where the hardware puts wave w of a workgroup is not architected (this
  // kernel at 154 VGPRs: consecutive waves share a SIMD; the 256-VGPR ping-pong GEMM: w and w + 4), so every wave reads its
  // SIMD id (HW_REG_HW_ID bits 5:4) and joins group 1 iff a lower-numbered wave of the workgroup sits on the same SIMD.
  // Any assignment is CORRECT (each group runs the same number of barriers); only the overlap depends on it.
  __shared__ int simd_of[NW];
  const int my_simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  if (lane == 0) simd_of[wave] = my_simd;
  __syncthreads();
  int grp_ = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w)
    if (w < wave && simd_of[w] == my_simd) grp_ = 1;
  if (p.dbg == 1) grp_ = wave >> 2;            // A/B only: wave >> 2 measured = SIMD-id pairing; wave & 1 and (wave >> 1) & 1 are 15 % slower
  else if (p.dbg == 2) grp_ = wave & 1;
  const int grp = __builtin_amdgcn_readfirstlane(grp_);
  const int ql = lane & 31, hi = lane >> 5;
  // XCD-aware mapping as in attn2: all q-blocks of one (batch, head) pair on ONE XCD
  const int nqb = (p.Lq + 32 * NW - 1) / (32 * NW);
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
  const int q0 = qblk * (32 * NW) + wave * 32;
  const half_t* Kb = p.K + (size_t)kb * p.kv_stride * p.ldk + h * D;
  const half_t* Vb = p.Vt + (size_t)h * D * p.ldvt + (size_t)kb * p.kv_stride;
  const int nt = p.Lk / A2_KT;                 // key tiles (Lk % 64 == 0: launcher); pairs 0 .. nt

  // constant rows of every V^T stage: ones in row D (softmax denominator), zeros in the rest of the padding
  for (int i = tid; i < NST * (DVT * 32 - D) * 8; i += 64 * NW) {
    const int st = i / ((DVT * 32 - D) * 8), rem = i % ((DVT * 32 - D) * 8);
    const int row = D + rem / 8, slot = rem % 8;
    const half_t v = row == D ? (half_t)1.0f : (half_t)0.0f;
    half8_t w = {v, v, v, v, v, v, v, v};
    *reinterpret_cast<half8_t*>(smem + st * STAGE + KBYTES + row * 128 + slot * 16) = w;
  }
  if (tid < 2) {                               // k-slots D.. of every key: {1, 0, .., 0}, one block per 32-key sub-tile
    half8_t w = {(half_t)1.0f, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<half8_t*>(smem + CONST_OFF + tid * 32 * KROWB) = w;
  }

  const float sc = p.scale_log2;
  half8_t qf[KS];
  {
    const int qrow = min(q0 + ql, p.Lq - 1);
    const half_t* Qp = p.Q + ((size_t)b * p.Lq + qrow) * p.ldq + h * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = s * 16 + hi * 8;
      half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < D) v = *reinterpret_cast<const half8_t*>(Qp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * sc);
      qf[s] = v;
    }
  }

  // ---- DMA: piece q of a pair (q < NKI: K image, else V^T image) is issued by wave q % NW
  const int n_mine = wave < NI ? (NI - wave + NW - 1) / NW : 0;
  const char* src0[NQ];
  long step[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = qi * NW + wave;
    src0[qi] = reinterpret_cast<const char*>(Kb);
    step[qi] = 0;
    if (q < NKI) {
      const int o = q * 1024 + lane * 16;
      const int row = o / KROWB, cb = o - row * KROWB;
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
  auto issue_pair = [&](int tp, int stage) {   // pair tp = {K tile tp, V^T tile tp - 1}, clamped to existing tiles at the two ends
    char* sb = smem + stage * STAGE;
    const int kt = min(tp, nt - 1), vt = max(tp - 1, 0);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const int q = qi * NW + wave;
      if (q < NKI)
        __builtin_amdgcn_global_load_lds((gptr_t)(src0[qi] + kt * step[qi]), (lptr_t)(sb + q * 1024), 16, 0, 0);
      else if (q < NI)
        __builtin_amdgcn_global_load_lds((gptr_t)(src0[qi] + vt * step[qi]), (lptr_t)(sb + KBYTES + (q - NKI) * 1024), 16, 0, 0);
    }
  };

  floatx16 o[DVT], s[2];
  half8_t pf[4];
  float m_run = 0.f;
#pragma unroll
  for (int t = 0; t < DVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) pf[k] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};

  // K row read by MFMA row index ql: key kappa(ql) = ql with bits 2 and 3 swapped; V^T slot swizzle of row t*32 + ql
  const int krow = (ql & ~12) | ((ql & 4) << 1) | ((ql & 8) >> 1);
  const int vsw = (ql >> 1) & 7;

  // ---- matrix phase of pair tp (stage st): P.V of tile tp-1 (PV) and / or the scores of tile tp (QK).  ALL operand fragments are
  // requested from LDS before the first MFMA (one LDS latency per phase: with two waves per SIMD nobody else hides a per-fragment
  // wait), then the MFMAs run back to back.
  auto mphase_impl = [&](auto pv_c, auto qk_c, int st) {
    constexpr bool PV = decltype(pv_c)::value, QK = decltype(qk_c)::value;
    const char* ks = smem + st * STAGE;
    const char* vs = ks + KBYTES;
    half8_t vfr[DVT][4], kfr[2][KS];
    if constexpr (PV) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < DVT; ++t) vfr[t][k] = *reinterpret_cast<const half8_t*>(vs + (t * 32 + ql) * 128 + (((k * 2 + hi) ^ vsw) << 4));
    }
    if constexpr (QK) {
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const char* kp = ks + (sub * 32 + krow) * KROWB + (k * 16 + hi * 8) * 2;
          if (k == KS - 1) kp = hi ? smem + CONST_OFF + sub * 32 * KROWB : kp;
          kfr[sub][k] = *reinterpret_cast<const half8_t*>(kp);
        }
    }
    if constexpr (PV) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < DVT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfr[t][k], pf[k], o[t], 0, 0, 0);
    }
    if constexpr (QK) {
      const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr[sub][k], qf[k], k == 0 ? zero : s[sub], 0, 0, 0);
    }
    // issue order: PF fragment reads ahead, then one read per MFMA (the four matrix-phase waves of a CU start together: asking
    // for all 14 KiB per wave up front makes the LDS, not the matrix pipe, pace the first half of the phase)
    constexpr int NRD = (PV ? DVT * 4 : 0) + (QK ? 2 * KS : 0), PF = 4;
    __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
    for (int i = 0; i < NRD - PF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
    // the MFMAs belong to THIS slot: without the pins the compiler sinks them past the closing barrier into the vector phase
    if constexpr (PV) {
#pragma unroll
      for (int t = 0; t < DVT; ++t) asm volatile("" : "+v"(o[t]) : : "memory");
    }
    if constexpr (QK) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) asm volatile("" : "+v"(s[sub]) : : "memory");
    }
  };
  // ---- vector phase of tile t: s already is log2e*scale*q.k - m_run (m_run: fp16-representable reference in slot D of Q)
  auto vphase = [&](auto first_c) {
    float mloc = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[0][r]), s[1][r]);
    mloc = a2_xhalf_max(mloc);
    constexpr bool first = decltype(first_c)::value;
    if (first || !__all(mloc <= A2_THR)) {
      const float want = m_run + (first ? mloc : fmaxf(mloc, 0.f));
      const float m_new = (float)(half_t)fminf(fmaxf(want, -60000.f), 60000.f);
      const float d = m_new - m_run;
      m_run = m_new;
      if (hi) qf[KS - 1][0] = (half_t)(-m_new);
      if (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[sub][r] -= d;
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        pf[sub * 2 + (r >> 3)][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[sub][r]);
        pf[sub * 2 + (r >> 3)][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(s[sub][r + 1]);
      }
    // P must EXIST before the barrier that ends this phase: left alone, the compiler sinks the exponentials across the
    // s_barrier to their first use (the P.V MFMAs of the next matrix phase) and the two phases collapse into one
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      typedef int int4v __attribute__((ext_vector_type(4)));
      int4v x = __builtin_bit_cast(int4v, pf[k]);
      asm volatile("" : "+v"(x) : : "memory");
      pf[k] = __builtin_bit_cast(half8_t, x);
    }
  };

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the constant rows above are WRITTEN before the first barrier hands them over
  issue_pair(0, 0);
  issue_pair(1, 1);                            // nt >= 1, so pair 1 exists
  // One loop per group (not one loop with per-slot group tests), first and last pair peeled: the steady-state body has ONE
  // matrix-phase flavour and keeps O^T, S^T and P in fixed registers (merging flavours costs a 64-register copy per phase).
  using T_ = std::true_type;
  using F_ = std::false_type;
  auto run = [&](auto grp_c) {
    constexpr int G = decltype(grp_c)::value;
    auto slot_a = [&](int tp, int st) {        // opens slot 2tp: pair tp has landed; this wave's pieces of pair tp+1 may stay in flight
      a2_wait_dyn(tp + 1 <= nt ? n_mine : 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_sched_barrier(0);
      if (tp + 2 <= nt) {
        int s2 = st + 2;
        if (s2 >= NST) s2 -= NST;
        issue_pair(tp + 2, s2);                // into the stage pair tp-1 has left (both groups are past its slots)
      }
    };
    // pair 0: scores only
    slot_a(0, 0);
    if constexpr (G == 0) mphase_impl(F_{}, T_{}, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 0) vphase(T_{});
    else mphase_impl(F_{}, T_{}, 0);
    int st = 1;
    for (int tp = 1; tp < nt; ++tp) {
      slot_a(tp, st);
      if constexpr (G == 0) mphase_impl(T_{}, T_{}, st);
      else if (tp == 1) vphase(T_{});
      else vphase(F_{});
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (G == 0) vphase(F_{});
      else mphase_impl(T_{}, T_{}, st);
      if (++st == NST) st = 0;
    }
    // pair nt: P.V of the last tile only
    slot_a(nt, st);
    if constexpr (G == 0) mphase_impl(T_{}, F_{}, st);
    else if (nt == 1) vphase(T_{});
    else vphase(F_{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 1) mphase_impl(T_{}, F_{}, st);
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  {
    constexpr int rt = D % 32;                 // row of the ones inside the last tile; rt % 8 == 0 -> lane half 0
    constexpr int reg = (rt & 3) + 4 * (rt >> 3);
    const float l_tot = __shfl(o[DVT - 1][reg], ql, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + ql;
    if (qr < p.Lq) {
      half_t* Op = p.O + ((size_t)b * p.Lq + qr) * p.ldo + h * D;
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = t * 32 + 8 * g + 4 * hi;
          if (dv < D) {
            half4_t ov = {(half_t)(o[t][4 * g] * inv), (half_t)(o[t][4 * g + 1] * inv), (half_t)(o[t][4 * g + 2] * inv),
                          (half_t)(o[t][4 * g + 3] * inv)};
            *reinterpret_cast<half4_t*>(Op + dv) = ov;
          }
        }
    }
  }
}

template <int D>
static int launch_attn3(const AttnParams& p_, hipStream_t stream) {
  AttnParams p = p_;
  static const int dbg = md_env_int("MD_ATTN3_GRP", 0);   // 0: group by SIMD id; 1: wave >> 2; 2: wave & 1 (A/B of the pairing)
  p.dbg = dbg;
  constexpr int DVT = (D + 31) / 32;
  constexpr int smem = 3 * (A2_KT * D * 2 + DVT * 32 * 128) + 32 * D * 2 + 16;
  md_ensure_dynamic_lds<attn3_kernel<D>>(smem);
  dim3 grid(cdiv(p.Lq, 256) * p.H * p.B);
  hipLaunchKernelGGL(attn3_kernel<D>, grid, dim3(512), smem, stream, p);
  MD_CHECK_LAUNCH("md_attention_fwd");
  return MD_OK;
}
