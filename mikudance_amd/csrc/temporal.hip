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
// NCH = D / 8 when the head dim is one of the UNet's (40 / 80 / 160): the lane's whole query row is then fetched into registers
// BEFORE the K/V staging is issued, so its HBM latency hides behind the DMA instead of being paid once per 8-column chunk
// inside the score loop (NCH = 0: any D, query chunks loaded in the loop).
template <int FMAX, int NCH>
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
  // this lane's (pixel, head, query frame); lanes past the last pixel keep a valid address and are retired after the barrier
  const int hl = t % p.HG;
  const int i = (t / p.HG) % p.F;
  const int pl = t / (p.HG * p.F);
  const long gpq = pg + pl;
  const bool live = pl < p.PB && gpq < npix;
  const long gpc = min(gpq, npix - 1);
  const size_t qrow = ((size_t)(gpc / p.HW) * p.F + i) * p.HW + (size_t)(gpc % p.HW);
  const half_t* qp = p.Q + qrow * p.ldq + col0 + hl * p.D;
  half8_t qreg[NCH > 0 ? NCH : 1];
  if constexpr (NCH > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) qreg[c] = *reinterpret_cast<const half8_t*>(qp + c * 8);
  }
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

  if (!live) return;
  half_t* op = p.O + qrow * p.ldo + col0 + hl * p.D;
  const half_t* kp = Ks + (size_t)pl * CW + hl * p.D;   // frame j at + j * PB * CW
  const half_t* vp = Vs + (size_t)pl * CW + hl * p.D;
  const int fstride = p.PB * CW;
  const int nch = NCH > 0 ? NCH : (p.D >> 3);

  // No per-frame branches: frames beyond F are clamped to F-1 for the loads and masked arithmetically afterwards, so the
  // compiler can batch all FMAX ds_read_b128 of a chunk ahead of the dot products (a branch per frame serialised one
  // LDS round trip per frame).
  int joff[FMAX];
#pragma unroll
  for (int j = 0; j < FMAX; ++j) joff[j] = min(j, p.F - 1) * fstride;
  float s[FMAX];
#pragma unroll
  for (int j = 0; j < FMAX; ++j) s[j] = 0.f;
  auto score_chunk = [&](int c, const half8_t q8) {
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
  };
  if constexpr (NCH > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      score_chunk(c, qreg[c]);
      // one chunk's FMAX K rows in registers at a time: left alone, the NCH * FMAX LDS reads are all issued up front and the dot
      // products sunk behind them (350+ VGPRs); the empty asm pins every partial score before the next chunk's reads
#pragma unroll
      for (int j = 0; j < FMAX; ++j) asm volatile("" : "+v"(s[j]) : : "memory");
    }
  } else {
    for (int c = 0; c < nch; ++c) score_chunk(c, *reinterpret_cast<const half8_t*>(qp + c * 8));
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
#pragma nounroll
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


// ---------------------------------------------------------------------------------------------- matrix-core flavour, F <= 32
// The lane-per-query kernel above is VALU bound (PMC on MI355X, d = 40: VALU 90 % busy at 3.9 TB/s: ~580 VALU instructions
// per (pixel, head)).  Here ONE WAVE owns a (pixel, head) unit and the two small products run on the matrix core, in QB x QB
// blocks of 16 keys x 16 queries (QB = 1: F <= 16, QB = 2: F <= 32 -- the 30-frame windows of BASELINE configs[4]):
//   S^T[key][query] = K Q^T   v_mfma_f32_16x16x32_f16, A = K fragment, B = Q fragment (both LDS, ds_read_b128), ceil(D / 32) steps,
//                             k-slots past D zero filled
//   softmax over keys         a lane holds keys 16 kb + 4g .. 4g+3 of query 16 qb + m (g = lane / 16, m = lane % 16): in-lane steps
//                             + 2 lane exchanges (xor 16, xor 32) for the maximum and for the sum
//   O^T[d][query] = V^T P^T   v_mfma_f32_16x16x16_f16 (one per key block, accumulated): the S^T accumulator layout IS the B operand
//                             layout (keys 4g .. 4g+3 of query m), A = V^T tile gathered from the frame-major LDS image with 2-byte
//                             reads, ceil(D / 16) tiles
// Frames >= F are clamped for the loads, masked as keys (-inf) and not stored as queries.  The LDS image keeps the DMA's lane
// linear order but every frame row carries one extra 16-byte slot: the row pitch is an odd number of slots (41 / 81), so the 16 frame
// rows of one k-slice start in 16 different slots of the 256-byte bank row (unpadded rows of 640 / 1280 bytes would all start in the
// same one).  A ds_read_b128 lane group mixes two k-slices, which leaves 2-way conflicts (3 % of the cycles at d = 40; a pitch of
// 2 mod 4 slots would remove them: tests/test_lds_layouts_cpu.py, profiles/r04_pmc_temporal_final.md).
//
// Round 4, the memory side (stream-by-stream ablation on MI355X, profiles/r04_ab_temporal_q_and_o_through_lds.log: with the arithmetic
// stripped the kernel ran at the same 4.3 TB/s, K + V alone moved at 5.9 TB/s, Q alone at 2.0, the output at 2.7):
//   * Q comes in like K and V, as whole frame rows by direct-to-LDS DMA (the fragment loads straight from global memory touched 16
//     rows x 64 bytes per instruction), three images back to back, the lanes past the last 16-byte chunk switched off instead of
//     rounding each image to the DMA's 1-KiB rows;
//   * at d <= 80 the output leaves through the Q image (a unit's Q columns are dead once its scores exist): the accumulator layout
//     gives a lane 8 bytes and a wave 32-byte runs per row; from LDS every store is a 16-byte piece of a whole row;
//   * 48 KiB of LDS per workgroup at most -> one pixel x 8 heads at d = 40 (31 KiB, 5 workgroups per CU).
// Same box, F = 16: 0.183 -> 0.152 ms at 96 x 96 (5.0 TB/s), 0.094 -> 0.077 at 48 x 48, 0.045 -> 0.034 at 24 x 24; F = 30 at
// 128 x 128: 0.715 -> 0.594 ms.  -DTA_Q_REGISTERS builds the previous form (A/B only).
template <int D, int QB>
__global__ __launch_bounds__(256) void temporal_attn_mfma_kernel(TemporalParams p) {
  constexpr int NKS = (D + 31) / 32, NT = (D + 15) / 16, MAXU = QB == 1 ? 4 : 2;
#ifdef TA_Q_REGISTERS
  constexpr bool O_LDS = false;
#else
  constexpr bool O_LDS = D <= 80;                      // d = 160 workgroups hold two units: the extra barrier costs more than the 8-byte stores
#endif
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int CW = p.HG * D, cw8 = CW >> 3;
  const int ngrp = p.H / p.HG;
  const int grp = blockIdx.x % ngrp;
  const int pg = (blockIdx.x / ngrp) * p.PB;          // first global (b, pixel) of this workgroup (NB * HW < 2^31: launcher)
  const int npix = p.NB * p.HW;
  const int col0 = grp * CW;
  const int m = lane & 15, g = lane >> 4;
  int fm[QB];                                          // frame of this lane's query (B operand column) / key (A operand row) per block
#pragma unroll
  for (int b = 0; b < QB; ++b) fm[b] = min(16 * b + m, p.F - 1);
  const int nunit = p.PB * p.HG;                       // <= 4 * MAXU (launcher); wave w owns units w, w + 4, ...

#ifdef TA_Q_REGISTERS
  // ---- Q fragments of this wave's units: global -> registers, in flight while K / V are staged
  half8_t qf[MAXU][QB][NKS];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int unit = min(wave + 4 * u, nunit - 1);
    const int pl = unit / p.HG, hl = unit - pl * p.HG;
    const int gp = min(pg + pl, npix - 1);
    const int b = gp / p.HW, pix = gp - b * p.HW;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const size_t row = ((size_t)b * p.F + fm[qb]) * p.HW + pix;
      const half_t* qp = p.Q + row * p.ldq + col0 + hl * D + 8 * g;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (32 * s + 8 * g < D) v = *reinterpret_cast<const half8_t*>(qp + 32 * s);
        qf[u][qb][s] = v;
      }
    }
  }
#endif

  // ---- stage Q, K and V: chunk c -> (frame j, slot); slot < PB * cw8: (pixel, 16-byte column chunk), the last slot is padding
  const int spr = p.PB * cw8 + 1;                      // 16-byte slots per frame row
  const int RS = spr * 16;                             // frame row stride in bytes
  const int nchunk = p.F * spr;
#ifdef TA_Q_REGISTERS
  const int region = ((nchunk * 16 + 1023) >> 10) << 10;
  for (int c = t; c < (region >> 4); c += 256) {
#else
  // the three images back to back, no rounding to the DMA's 1-KiB rows: the lanes past the last chunk are switched off instead (a
  // DMA lane writes M0 + 16 * lane only if it is active), which is what lets a fifth workgroup fit on the CU at d = 40
  const int region = nchunk * 16;
  for (int c = t; c < nchunk; c += 256) {
#endif
#ifdef TA_Q_REGISTERS
    const int cl = min(c, nchunk - 1);                 // the rounded-up region has chunks past the last one
#else
    const int cl = c;
#endif
    const int j = cl / spr;
    int sl = cl - j * spr;
    if (sl == spr - 1) sl = 0;                         // padding slot: any valid address
    const int pl = sl / cw8, cc = sl - pl * cw8;
    const int gp = min(pg + pl, npix - 1);
    const int b = gp / p.HW, pix = gp - b * p.HW;
    const size_t row = ((size_t)b * p.F + j) * p.HW + pix;
    const int cbase = __builtin_amdgcn_readfirstlane(c) << 4;
    __builtin_amdgcn_global_load_lds((gptr_t)(p.K + row * p.ldk + col0 + cc * 8), (lptr_t)(smem + cbase), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(p.V + row * p.ldv + col0 + cc * 8), (lptr_t)(smem + region + cbase), 16, 0, 0);
#ifndef TA_Q_REGISTERS
    __builtin_amdgcn_global_load_lds((gptr_t)(p.Q + row * p.ldq + col0 + cc * 8), (lptr_t)(smem + 2 * region + cbase), 16, 0, 0);
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const char* Ks = smem;
  const char* Vs = smem + region;
#ifndef TA_Q_REGISTERS
  const char* Qs = smem + 2 * region;
#endif
  const int x16 = (lane ^ 16) << 2, x32 = (lane ^ 32) << 2;   // ds_bpermute addresses of the lanes holding the other keys of query m
  auto xch = [](int addr, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v))); };

#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int unit = wave + 4 * u;
    if (unit >= nunit) break;                          // wave uniform
    const int pl = unit / p.HG, hl = unit - pl * p.HG;
    const int gp = pg + pl;
    if (gp >= npix) break;
    const int ubase = (pl * CW + hl * D) * 2;          // byte offset of this unit's columns inside a frame row
#ifndef TA_Q_REGISTERS
    half8_t qfu[QB][NKS];                              // B operand: query 16 qb + m, k-slots 32 s + 8 g .. + 7 (zero past D)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int s2 = 0; s2 < NKS; ++s2) {
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (32 * s2 + 8 * g < D) v = *reinterpret_cast<const half8_t*>(Qs + fm[qb] * RS + ubase + (32 * s2 + 8 * g) * 2);
        qfu[qb][s2] = v;
      }
#endif
    // ---- S^T = K Q^T, QB x QB blocks
    floatx4 sacc[QB][QB];
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) sacc[kb][qb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
#pragma unroll
      for (int kb = 0; kb < QB; ++kb) {
        half8_t kf = {0, 0, 0, 0, 0, 0, 0, 0};
        if (32 * s + 8 * g < D) kf = *reinterpret_cast<const half8_t*>(Ks + fm[kb] * RS + ubase + (32 * s + 8 * g) * 2);
#pragma unroll
#ifdef TA_Q_REGISTERS
        for (int qb = 0; qb < QB; ++qb) sacc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[u][qb][s], sacc[kb][qb], 0, 0, 0);
#else
        for (int qb = 0; qb < QB; ++qb) sacc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qfu[qb][s], sacc[kb][qb], 0, 0, 0);
#endif
      }
    }
    // ---- softmax over the keys 16 kb + 4g + r of query 16 qb + m
    half4_t pb[QB][QB];                                 // [qb][kb]
    float inv[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float sv[QB][4];
      float mx = -1.0e30f;
#pragma unroll
      for (int kb = 0; kb < QB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[kb][r] = (16 * kb + 4 * g + r < p.F) ? sacc[kb][qb][r] * p.scale_log2 : -1.0e30f;
          mx = fmaxf(mx, sv[kb][r]);
        }
      mx = fmaxf(mx, xch(x16, mx));
      mx = fmaxf(mx, xch(x32, mx));
      float l = 0.f;
#pragma unroll
      for (int kb = 0; kb < QB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(sv[kb][r] - mx);   // masked keys: exp2(-huge) = 0
          l += e;
          pb[qb][kb][r] = (half_t)e;
        }
      l += xch(x16, l);
      l += xch(x32, l);
      inv[qb] = 1.f / l;
    }
    // ---- O^T = V^T P^T, 16 rows of d per tile
    const int ob = gp / p.HW, opix = gp - ob * p.HW;
    half_t* op[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) op[qb] = p.O + (((size_t)ob * p.F + fm[qb]) * p.HW + opix) * p.ldo + col0 + hl * D;
    const half_t* vcol = reinterpret_cast<const half_t*>(Vs + ubase);
    int vrow[QB][4];
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) vrow[kb][i] = min(16 * kb + 4 * g + i, p.F - 1) * (RS >> 1);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int dc = 16 * tt + m;
      half4_t vf[QB];
#pragma unroll
      for (int kb = 0; kb < QB; ++kb) {
        vf[kb] = half4_t{0, 0, 0, 0};
        if (dc < D) {
#pragma unroll
          for (int i = 0; i < 4; ++i) vf[kb][i] = vcol[vrow[kb][i] + dc];
        }
      }
      const int db = 16 * tt + 4 * g;                  // o[r] = O[query 16 qb + m][d = db + r]
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        floatx4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < QB; ++kb) o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[kb], pb[qb][kb], o, 0, 0, 0);
        if (16 * qb + m < p.F && db < D) {
          const half4_t ov = {(half_t)(o[0] * inv[qb]), (half_t)(o[1] * inv[qb]), (half_t)(o[2] * inv[qb]), (half_t)(o[3] * inv[qb])};
          if constexpr (O_LDS)   // into this unit's own (dead) columns of the Q image: the rows leave the workgroup as whole 16-byte pieces below
            *reinterpret_cast<half4_t*>(smem + 2 * region + fm[qb] * RS + ubase + db * 2) = ov;
          else
            *reinterpret_cast<half4_t*>(op[qb] + db) = ov;
        }
      }
    }
  }
  if constexpr (O_LDS) {
  // ---- O leaves through the Q image: 16-byte pieces of whole frame rows (the accumulator layout gives a lane 8 bytes and a wave
  // 32-byte runs; written straight from it the output moved at 2.7 TB/s, profiles/r04_ab_temporal_q_and_o_through_lds.log)
  __syncthreads();
  for (int c = t; c < nchunk; c += 256) {
    const int j = c / spr, sl = c - j * spr;
    if (sl == spr - 1) continue;                       // padding slot
    const int pl = sl / cw8, cc = sl - pl * cw8;
    const int gp = pg + pl;
    if (gp >= npix) continue;
    const int b = gp / p.HW, pix = gp - b * p.HW;
    const size_t row = ((size_t)b * p.F + j) * p.HW + pix;
    *reinterpret_cast<half8_t*>(p.O + row * p.ldo + col0 + cc * 8) = *reinterpret_cast<const half8_t*>(smem + 2 * region + (c << 4));
  }
  }
}

template <int D, int QB>
static void launch_temporal_mfma(TemporalParams p, hipStream_t st) {
  // heads per workgroup: all of them unless one pixel's Q + K + V images (3 * F rows of HG * D * 2 + 16 bytes) exceed the LDS budget
  // (48 KiB, swept 16 .. 128 at F = 16 and 48 / 64 / 100 at F = 30: profiles/r04_ab_temporal_q_and_o_through_lds.log); then as many
  // pixels as fit, at most 4 * MAXU (pixel, head) units (MAXU = 4 / 2 per wave).  d = 40 / 80 / 160 at F <= 16: 8 / 4 / 2 heads of one pixel
#ifdef TA_Q_REGISTERS
  constexpr int NIMG = 2;
  const size_t cap = (size_t)(QB == 1 ? 48 : 80) * 1024;
#else
  constexpr int NIMG = 3;                              // Q, K, V images
#ifndef TA_LDS_CAP
#define TA_LDS_CAP 48
#endif
#ifndef TA_LDS_CAP2
#define TA_LDS_CAP2 48
#endif
  const size_t cap = (size_t)(QB == 1 ? TA_LDS_CAP : TA_LDS_CAP2) * 1024;
#endif
  constexpr int UNITS = QB == 1 ? 16 : 8;
  int HG = p.H;
#ifdef TA_Q_REGISTERS
  auto lds = [&](int hg, int pb) { return (size_t)NIMG * ((((size_t)p.F * (pb * hg * D * 2 + 16)) + 1023) / 1024 * 1024); };
#else
  auto lds = [&](int hg, int pb) { return (size_t)NIMG * p.F * (pb * hg * D * 2 + 16); };
#endif
  while (HG > 1 && lds(HG, 1) > cap) HG >>= 1;
  int PB = UNITS / HG;
  while (PB > 1 && lds(HG, PB) > cap) --PB;
  p.HG = HG; p.PB = PB;
  const size_t smem = lds(HG, PB);
  const int grid = cdiv((long)p.NB * p.HW, PB) * (p.H / HG);
  md_ensure_dynamic_lds<temporal_attn_mfma_kernel<D, QB>>(128 * 1024);
  hipLaunchKernelGGL((temporal_attn_mfma_kernel<D, QB>), dim3(grid), dim3(256), smem, st, p);
}

template <int FMAX, int NCH>
static void launch_temporal_nch(const TemporalParams& p, int grid, int threads, size_t smem, hipStream_t st) {
  md_ensure_dynamic_lds<temporal_attn_kernel<FMAX, NCH>>(96 * 1024);
  hipLaunchKernelGGL((temporal_attn_kernel<FMAX, NCH>), dim3(grid), dim3(threads), smem, st, p);
}

template <int FMAX>
static void launch_temporal(const TemporalParams& p, int grid, int threads, size_t smem, hipStream_t st) {
  // head dims 40 / 80 / 160 run on the matrix-core kernel; this kernel serves every other D with query chunks loaded in the loop
  launch_temporal_nch<FMAX, 0>(p, grid, threads, smem, st);
}

extern "C" int md_temporal_attention_fwd_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo, int NB, int F, int HW,
                                             int H, int D, float scale, void* stream) {
  MD_CHECK_ARG(F >= 1 && F <= 32, "md_temporal_attention_fwd: F=%d frames, the positional-encoding table holds 32", F);
  MD_CHECK_ARG(D % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "md_temporal_attention_fwd: D and strides must be multiples of 8");
  MD_CHECK_ARG(H >= 1 && H <= 8 && (H & (H - 1)) == 0, "md_temporal_attention_fwd: H=%d heads must be a power of two <= 8", H);
  {
    // O must not overlap Q / K / V: a workgroup's output rows are other workgroups' inputs (every pixel's F frames are spread over the
    // token matrix), and the matrix-core kernel parks O in the LDS image of Q
    const size_t rows = (size_t)NB * F * HW;
    auto lo = [](const void* q) { return reinterpret_cast<uintptr_t>(q); };
    auto hi = [&](const void* q, int ld) { return reinterpret_cast<uintptr_t>(q) + ((rows - 1) * (size_t)ld + (size_t)H * D) * 2; };
    auto apart = [&](const void* q, int ld) { return hi(O, ldo) <= lo(q) || hi(q, ld) <= lo(O) || (ld == ldo && ((lo(q) > lo(O) ? lo(q) - lo(O) : lo(O) - lo(q)) / 2) % ldo >= (size_t)H * D && ((lo(q) > lo(O) ? lo(q) - lo(O) : lo(O) - lo(q)) / 2) % ldo + (size_t)H * D <= (size_t)ldo); };
    MD_CHECK_ARG(apart(Q, ldq) && apart(K, ldk) && apart(V, ldv), "md_temporal_attention_fwd: O must not overlap Q, K or V");
  }
  TemporalParams p;
  p.Q = (const half_t*)Q; p.K = (const half_t*)K; p.V = (const half_t*)V; p.O = (half_t*)O;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.NB = NB; p.F = F; p.HW = HW; p.H = H; p.D = D;
  p.scale_log2 = scale * 1.4426950408889634f;
#ifdef TA_Q_REGISTERS
  const bool o16 = true;
#else
  // the matrix-core kernel moves every operand and the output as 16-byte pieces
  const bool o16 = ((reinterpret_cast<uintptr_t>(O) | reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15) == 0;
#endif
  if ((D == 40 || D == 80 || D == 160) && (long)NB * HW < (1L << 31) && o16) {
    hipStream_t st = (hipStream_t)stream;
    if (F <= 16) {
      if (D == 40) launch_temporal_mfma<40, 1>(p, st);
      else if (D == 80) launch_temporal_mfma<80, 1>(p, st);
      else launch_temporal_mfma<160, 1>(p, st);
    } else {
      if (D == 40) launch_temporal_mfma<40, 2>(p, st);
      else if (D == 80) launch_temporal_mfma<80, 2>(p, st);
      else launch_temporal_mfma<160, 2>(p, st);
    }
    MD_CHECK_LAUNCH("md_temporal_attention_fwd");
    return MD_OK;
  }
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
