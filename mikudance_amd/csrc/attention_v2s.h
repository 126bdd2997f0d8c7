// attn2s_kernel<D>: short-key flavour of attn2_kernel (attention_v2.h) for the CROSS-attention of the UNets: Lq = H*W queries per
// frame against the 257 (padded 264) CLIP tokens, dispatched at d = 40 (the 96 x 96 level).
//
// attn2 gives such a problem one workgroup per 256 queries: constant rows, Q load, the first K / V^T DMA (~ 1-2 us before the
// first MFMA can start) and then only five key tiles (~ 2 us of work) -- the prologue latency is never hidden and the kernel runs
// at 400-490 TFLOP/s where the long self-attention reaches 790 (24 + 10 ms per clip at d = 40 / 80).  Here the whole K / V^T of a
// (batch, head) pair (<= 5 key tiles, 13-22 KiB each) is DMA'd into LDS ONCE per workgroup, and the workgroup then walks over
// many 256-query blocks of that pair: no barrier and no DMA inside the walk (LDS is read-only), waves run free, the only global
// traffic in the loop is the Q fragment load and the O store.  Math, layouts (kappa-permuted keys, swizzled V^T, ones row,
// folded reference with the OR-based lazy-rescale test at d = 40) are those of attn2, QT = 1.
//
// Validated in round 3 (every attention parity test with it switched on) and measured on MI355X against the ring kernel
// (profiles/r03_ab_attention_small.log): d = 40, Lq = 9216: 0.33-0.34 -> 0.25-0.27 ms (+22..36 %); d = 80 (Lq = 2304) and d = 160
// -2 %: the dispatcher uses it at d = 40 only (the template still compiles for 80).
#pragma once

template <int D>
__global__ __launch_bounds__(512, D <= 40 ? 4 : 2) void attn2s_kernel(AttnParams p, int qsplit) {
  constexpr int NW = 8;
  constexpr int KS = (D + 15) / 16;
  constexpr bool ONES = (D % 32) != 0;
  static_assert(ONES, "the denominator comes from the ones row (d = 40, 80)");
  constexpr int DVT = (D + 31) / 32;
  constexpr int KROWB = D * 2;
  constexpr int KBYTES = A2_KT * KROWB;
  constexpr int VBYTES = DVT * 32 * 128;
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int NKI = KBYTES / 1024, NVI = (D * 128) / 1024, NI = NKI + NVI;
  constexpr bool FOLD = (D % 16) == 8;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  const int ntiles = (p.Lk + A2_KT - 1) / A2_KT;
  const int const_off = ntiles * STAGE;        // FOLD: two 16-B blocks {1,0,...,0}, 32 K rows apart
  const int nqb = (p.Lq + 32 * NW - 1) / (32 * NW);
  // workgroup -> ((batch, head) pair, slice of its q-blocks): pair = blockIdx % npair keeps the qsplit workgroups of a pair apart
  const int npair = p.B * p.H;
  const int pair = blockIdx.x % npair, slice = blockIdx.x / npair;
  const int b = pair / p.H, h = pair - b * p.H;
  const int kb = p.kv_index ? p.kv_index[b] : b;
  const int lk8 = (p.Lk + 7) & ~7;
  const half_t* Kb = p.K + (size_t)kb * p.kv_stride * p.ldk + h * D;
  const half_t* Vb = p.Vt + (size_t)h * D * p.ldvt + (size_t)kb * p.kv_stride;

  // constant rows of every V^T stage: ones in row D (softmax denominator), zeros in the rest of the padding
  for (int i = tid; i < ntiles * (DVT * 32 - D) * 8; i += 64 * NW) {
    const int st = i / ((DVT * 32 - D) * 8), rem = i % ((DVT * 32 - D) * 8);
    const int row = D + rem / 8, slot = rem % 8;
    const half_t v = row == D ? (half_t)1.0f : (half_t)0.0f;
    half8_t w = {v, v, v, v, v, v, v, v};
    *reinterpret_cast<half8_t*>(smem + st * STAGE + KBYTES + row * 128 + slot * 16) = w;
  }
  if (FOLD && tid < 2) {
    half8_t w = {(half_t)1.0f, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<half8_t*>(smem + const_off + tid * 32 * KROWB) = w;
  }

  // ---- the whole K / V^T of the pair: piece q of tile it (q < NKI: K image, else V^T image) is issued by wave q % NW; the last
  // tile may be ragged (clamped source addresses, the scores of the keys past Lk are masked below)
  for (int it = 0; it < ntiles; ++it) {
    char* sb = smem + it * STAGE;
    const int j0 = it * A2_KT;
    for (int q = wave; q < NI; q += NW) {
      if (q < NKI) {
        const int o = q * 1024 + lane * 16;
        const int row = o / KROWB, cb = o - row * KROWB;
        const int key = min(j0 + row, p.Lk - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(Kb + (size_t)key * p.ldk) + cb), (lptr_t)(sb + q * 1024), 16, 0, 0);
      } else {
        const int qv = q - NKI;
        const int o = qv * 1024 + lane * 16;
        const int dv = o >> 7, ps = (o & 127) >> 4;
        const int ls = ps ^ ((dv >> 1) & 7);
        const int j = min(j0 + ls * 8, lk8 - 8);
        __builtin_amdgcn_global_load_lds((gptr_t)(Vb + (size_t)dv * p.ldvt + j), (lptr_t)(sb + KBYTES + qv * 1024), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA pieces landed, constant rows written
  __syncthreads();

  const float sc = p.scale_log2;
  const int krow = (ql & ~12) | ((ql & 4) << 1) | ((ql & 8) >> 1);
  const int vsw = (ql >> 1) & 7;

  for (int qb = slice; qb < nqb; qb += qsplit) {
    const int q0 = qb * (32 * NW) + wave * 32;
    if (q0 >= p.Lq) continue;                  // wave uniform; no barrier below
    half8_t qf[KS];
    {
      const int qrow = min(q0 + ql, p.Lq - 1);
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
        qf[s] = v;
      }
    }
    floatx16 o[DVT];
    float m_run = FOLD ? 0.f : NEG_BIG;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

    for (int it = 0; it < ntiles; ++it) {
      const int j0 = it * A2_KT;
      const char* ks = smem + it * STAGE;
      const char* vs = ks + KBYTES;
      // ---- S^T = K Q^T
      floatx16 s[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const char* kp = ks + (sub * 32 + krow) * KROWB + (k * 16 + hi * 8) * 2;
          if (FOLD && k == KS - 1) kp = hi ? smem + const_off + sub * 32 * KROWB : kp;
          const half8_t kf = *reinterpret_cast<const half8_t*>(kp);
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[k], s[sub], 0, 0, 0);
        }
      }
      // register r of sub-tile `sub` in lane half `hi` holds key j0 + sub*32 + 16*(r>>3) + 8*hi + (r&7)
      if (j0 + A2_KT > p.Lk) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j0 + sub * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (key >= p.Lk) s[sub][r] = NEG_BIG;
          }
      }
      half8_t pf[4];
      if constexpr (FOLD) {
        // s = log2e*scale*q.k - r, r = the reference in slot D of Q, kept 4 octaves above the running maximum (see attn2)
        auto make_p = [&]() {
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              pf[sub * 2 + (r >> 3)][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[sub][r]);
              pf[sub * 2 + (r >> 3)][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(s[sub][r + 1]);
            }
        };
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        const bool first = it == 0;
        bool trig = first;
        if (!first) {
          make_p();
          const uint4v ob = __builtin_bit_cast(uint4v, pf[0]) | __builtin_bit_cast(uint4v, pf[1]) | __builtin_bit_cast(uint4v, pf[2]) |
                            __builtin_bit_cast(uint4v, pf[3]);
          trig = __any(((ob[0] | ob[1] | ob[2] | ob[3]) & 0x40004000u) != 0);
        }
        if (trig) {
          float mloc = fmaxf(s[0][0], s[1][0]);
#pragma unroll
          for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[0][r]), s[1][r]);
          mloc = a2_xhalf_max(mloc) + 4.0f;
          const float want = m_run + (first ? mloc : fmaxf(mloc, 0.f));
          const float m_new = (float)(half_t)fminf(fmaxf(want, -60000.f), 60000.f);
          const float d = m_new - m_run;
          m_run = m_new;
          if (hi) qf[KS - 1][0] = (half_t)(-m_new);
          if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
          }
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] -= d;
          make_p();
        }
      } else {
        float mloc = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[0][r]), s[1][r]);
        mloc = a2_xhalf_max(mloc);
        const float mt = mloc * sc;
        if (!__all(mt <= m_run + A2_THR)) {
          const float m_new = fmaxf(m_run, mt);
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
#pragma unroll
          for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            pf[sub * 2 + (r >> 3)][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[sub][r] * sc - m_run);
            pf[sub * 2 + (r >> 3)][(r & 7) + 1] = (half_t)__builtin_amdgcn_exp2f(s[sub][r + 1] * sc - m_run);
          }
      }
      // ---- O^T += V^T P^T
#pragma unroll
      for (int t = 0; t < DVT; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const half8_t vf = *reinterpret_cast<const half8_t*>(vs + (t * 32 + ql) * 128 + (((k * 2 + hi) ^ vsw) << 4));
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[k], o[t], 0, 0, 0);
        }
      }
    }

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

// eligibility: d = 40 / 80, all key tiles of a pair resident in <= 64 KiB (two workgroups per CU at d = 40), long query side
template <int D>
static bool attn2s_eligible(const AttnParams& p) {
  if constexpr (D != 40) return false;
  constexpr int DVT = (D + 31) / 32;
  constexpr int STAGE = A2_KT * D * 2 + DVT * 32 * 128;
  const int ntiles = (p.Lk + A2_KT - 1) / A2_KT;
  return p.Lk >= 8 && (size_t)ntiles * STAGE + 32 * D * 2 + 16 <= (size_t)(D == 40 ? 72 : 120) * 1024 && p.Lq >= 2048;
}

template <int D>
static int launch_attn2s(const AttnParams& p, hipStream_t stream) {
  if constexpr (D == 40) {
    constexpr int DVT = (D + 31) / 32;
    constexpr int STAGE = A2_KT * D * 2 + DVT * 32 * 128;
    const int ntiles = (p.Lk + A2_KT - 1) / A2_KT;
    const int smem = ntiles * STAGE + 32 * D * 2 + 16;
    const int nqb = cdiv(p.Lq, 256), npair = p.B * p.H;
    // about two workgroups per CU in total, at least 4 q-blocks per workgroup so that the one-off K / V^T load is amortised
    int qsplit = cdiv(2 * 256, npair);
    if (qsplit > nqb / 4) qsplit = nqb / 4;
    if (qsplit < 1) qsplit = 1;
    md_ensure_dynamic_lds<attn2s_kernel<D>>(smem);
    hipLaunchKernelGGL(attn2s_kernel<D>, dim3(npair * qsplit), dim3(512), smem, stream, p, qsplit);
    MD_CHECK_LAUNCH("md_attention_fwd");
  }
  return MD_OK;
}
