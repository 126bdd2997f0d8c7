// gemm_tw_kernel: TWO free-running waves per SIMD -- the software-pipelined wave of gemm_sp_kernel at the ping-pong kernel's tile
// geometry (included by gemm.hip; same operands, LDS image, swizzle, tile order and epilogue arithmetic as gemm_sp_kernel).
//
// What the round-3 ablations of gemm_sp_kernel said (profiles/r03_ab_gemm_sp_ablation.log): barriers and vmcnt waits cost nothing,
// the fragment reads 5-7 %, and ISSUING the direct-to-LDS DMA pieces from the compute waves costs 25-30 % (a piece blocks the
// in-order wave for ~60 cycles, two MFMA slots, and with one wave per SIMD nobody else feeds that SIMD's matrix pipe meanwhile;
// spreading the waves' pieces over different MFMA gaps did not help).  gemm_pp_kernel has the second wave but alternates the two
// between a load slot and an MFMA slot under a workgroup barrier, and its load slot is the longer one.  Here:
//   * 512 threads, 8 waves as 4 (M) x 2 (N), wave tile 64 x (32 NT): NT = 5 -> 256 x 320 tiles (plain / conv), NT = 4 -> 256 x 256
//     (GEGLU); 160 / 128 accumulator registers, 256 registers per wave;
//   * EVERY wave runs the whole software pipeline by itself -- MFMAs of k-step u, fragment reads of step u+1 and its share of
//     the DMA pieces of the K tile 3-4 ahead interleaved by hand -- and the two waves of a SIMD are NOT phase-locked: when one is
//     held up issuing a DMA piece or waiting for a fragment, the other one's MFMAs keep the matrix pipe busy;
//   * fragment registers: the A fragments (2) are double buffered, a W fragment is re-read for the next k-step right behind the
//     two MFMAs that consume it (36 fragment registers instead of 56);
//   * ring, barrier placement, persistence, peeled first K tile, grouped tile order, accumulator epilogue: as gemm_sp_kernel.
//     36 (32) DMA pieces per K tile: every wave 2 A + 2 W pieces, waves 0-3 one more W piece when NT = 5 (counted waits differ).
#pragma once

template <bool CONV, bool GEGLU, int NT>
__global__ __launch_bounds__(512, 2) void gemm_tw_kernel(GemmParams p) {
  constexpr int MT = 2, BK = 32;
  constexpr int BM = 256, BN = 64 * NT;
  constexpr int ROWB = BK * 2, RPI = 1024 / ROWB;
  constexpr int PA = 2, PB = (BN / RPI) / 8;             // DMA pieces of every wave per K tile: A rows, W rows
  constexpr int XB = BN / RPI - PB * 8;                  // ... plus one more W piece on waves 0 .. XB-1
  static_assert(PB == 2 && (XB == 0 || XB == 4), "piece schedule below");
  static_assert(!GEGLU || NT % 2 == 0, "GEGLU pairs 32-column sub-tiles (2q, 2q+1) of a wave");
  constexpr int OPA = BM * ROWB, STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool has_x = XB > 0 && wave < XB;
  const int lrow = lane >> 2, pslot = lane & 3;
  const int nk = p.K / BK;
  const int nwg = p.tiles_total;
  const int ntile = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // output tiles of this workgroup

  auto tile_origin = [&](int i, int& m0, int& n0) {       // see gemm_sp_kernel
    const int v = (int)blockIdx.x + i * (int)gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = v & 7, idx = v >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gsz = p.group_m * p.tiles_n;
    const int grp = t / gsz, first = grp * p.group_m;
    const int rows = min(p.group_m, p.tiles_m - first);
    const int l = t - grp * gsz;
    const int tn = l / rows;
    m0 = (first + l - tn * rows) * BM;
    n0 = tn * BN;
  };

  // ------------------------------------------------------------------ issue side (runs 3-4 K tiles ahead of the compute side)
  const half_t* a_src[PA];
  const half_t* w_src[PB + 1];
  int a_oy[PA], a_ox[PA];
  auto set_sources = [&](int i) {
    int m0, n0;
    tile_origin(i, m0, n0);
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int row = (wave * PA + j) * RPI + lrow;
      const int lslot = pslot ^ ((row >> 2) & 3);             // source-side swizzle (the DMA writes LDS lane-linearly)
      const int m = m0 + row;
      const int mm = m < p.M ? m : p.M - 1;
      if (CONV) {
        const int hw = p.Hout * p.Wout;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        a_oy[j] = oy * p.stride - p.pad;
        a_ox[j] = ox * p.stride - p.pad;
        a_src[j] = p.A + (size_t)b * p.Hin * p.Win * p.Cin + lslot * 8;
      } else {
        a_oy[j] = a_ox[j] = 0;
        a_src[j] = p.A + (size_t)mm * p.lda + lslot * 8;
      }
    }
#pragma unroll
    for (int j = 0; j < PB + 1; ++j) {
      const int row = (j < PB ? wave * PB + j : PB * 8 + (wave & 3)) * RPI + lrow;
      w_src[j] = p.W + (size_t)(n0 + row) * p.K + (pslot ^ ((row >> 2) & 3)) * 8;      // N % BN == 0 (launcher)
    }
  };
  const half_t* zero_src = g_zero_page + 0;
  int is_it = 0, is_kt = 0;
  int is_k0 = 0, is_c0 = 0, is_ky = 0, is_kx = 0;
  set_sources(0);
  auto issue_a = [&](int stage, int j) {
    char* dst = smem + stage * STAGE + (wave * PA + j) * 1024;
    if (CONV) {
      const unsigned hup = p.Hin << p.upsample, wup = p.Win << p.upsample;
      const int iy = a_oy[j] + is_ky, ix = a_ox[j] + is_kx;
      const bool ok = (unsigned)iy < hup && (unsigned)ix < wup;
      unsigned off = __umul24(__umul24((unsigned)(iy >> p.upsample), (unsigned)p.Win) + (unsigned)(ix >> p.upsample), (unsigned)p.Cin) + is_c0;
      asm volatile("" : "+v"(off));
      const half_t* src = a_src[j] + off;
      src = ok ? src : zero_src;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + is_k0), (lptr_t)dst, 16, 0, 0);
    }
  };
  auto issue_w = [&](int stage, int j) {                  // j = PB: the extra piece (waves 0 .. XB-1)
    const int piece = j < PB ? wave * PB + j : PB * 8 + (wave & 3);
    __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + is_k0), (lptr_t)(smem + stage * STAGE + OPA + piece * 1024), 16, 0, 0);
  };
  auto issue_advance = [&]() {
    is_k0 += BK;
    if (CONV) {
      is_c0 += BK;
      if (is_c0 == p.Cin) {
        is_c0 = 0;
        if (++is_kx == 3) { is_kx = 0; ++is_ky; }
      }
    }
    if (++is_kt == nk) {
      is_kt = 0;
      is_k0 = is_c0 = is_ky = is_kx = 0;
      if (++is_it < ntile) set_sources(is_it);
    }
  };

  // ------------------------------------------------------------------ compute side
  const int frow = lane & 31, fhi = lane >> 5;
  int a_rd[2][MT], b_rd[2][NT];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ra = wm * (32 * MT) + i * 32 + frow;
      a_rd[s][i] = ra * ROWB + (((s * 2 + fhi) ^ ((ra >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int rb = wn * (32 * NT) + j * 32 + frow;
      b_rd[s][j] = OPA + rb * ROWB + (((s * 2 + fhi) ^ ((rb >> 2) & 3)) << 4);
    }
  }
  floatx16 acc[MT][NT];
  half8_t fa0[MT], fa1[MT], fb[NT];

  // One k-step, issue order pinned by hand: the MFMAs of the current fragments (FAU, fb), j-major; behind the two MFMAs that consume
  // fb[j] its re-read for the NEXT k-step (ring stage SB, k-step S); behind MFMAs 0 and 2 the next step's A fragments (FAL); behind
  // MFMAs 4, 6 (, 8) the DMA pieces of this half.  PART 1: the two A pieces of tile g+4 (half 2); PART 2: the W pieces of tile g+3
  // (half 1; the third only on the waves that own one).
#define TW_STEP(FAU, FAL, SB, S, ZERO, DST, PART)                                                           \
  {                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < MT * NT; ++k) {                                                   \
      const int j = k / MT, i = k % MT;                                                                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], FAU[i], (ZERO) ? floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[i][j], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      if (i == MT - 1) {                                                                                    \
        fb[j] = *reinterpret_cast<const half8_t*>((SB) + b_rd[S][j]);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      } else if (j < MT) {                                                                                  \
        FAL[j < MT ? j : 0] = *reinterpret_cast<const half8_t*>((SB) + a_rd[S][j < MT ? j : 0]);            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      }                                                                                                     \
      if (k == 4 || k == 6) {                                                                               \
        if ((PART) == 1) issue_a(DST, (k - 4) / 2);                                                         \
        else issue_w(DST, (k - 4) / 2);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      }                                                                                                     \
      if (k == 8 && (PART) == 2 && XB > 0) {                                                                \
        if (has_x) issue_w(DST, PB);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      }                                                                                                     \
    }                                                                                                       \
  }

  // ------------------------------------------------------------------ prologue: tiles 0, 1, 2 and the A pieces of tile 3
#pragma unroll 1
  for (int s = 0; s < 3; ++s) {
    issue_a(s, 0);
    issue_a(s, 1);
    issue_w(s, 0);
    issue_w(s, 1);
    if (has_x) issue_w(s, PB);
    issue_advance();
  }
  issue_a(3, 0);
  issue_a(3, 1);
  if (has_x) wait_vmcnt<2 * (PA + PB + 1) + PA>();                    // this wave's pieces of tile 0
  else wait_vmcnt<2 * (PA + PB) + PA>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < MT; ++i) fa0[i] = *reinterpret_cast<const half8_t*>(smem + a_rd[0][i]);
#pragma unroll
  for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const half8_t*>(smem + b_rd[0][j]);

#define TW_BODY(ZERO)                                                                                       \
  {                                                                                                         \
    const char* sb = smem + (g & 3) * STAGE;                                                                \
    const char* sbn = smem + ((g + 1) & 3) * STAGE;                                                         \
    /* half 1: MFMA (g, k 0-15) || read (g, k 16-31) || W pieces of tile g+3 */                             \
    TW_STEP(fa0, fa1, sb, 1, ZERO, (g + 3) & 3, 2)                                                          \
    issue_advance();                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    if (has_x) wait_vmcnt<2 * (PA + PB + 1)>(); /* this wave's pieces of tile g+1 (tiles g+2, g+3 may fly) */ \
    else wait_vmcnt<2 * (PA + PB)>();                                                                       \
    __builtin_amdgcn_s_barrier(); /* B_{g+1} */                                                             \
    /* half 2: MFMA (g, k 16-31) || read (g+1, k 0-15) || A pieces of tile g+4 -> the slot tile g has left */ \
    TW_STEP(fa1, fa0, sbn, 0, false, g & 3, 1)                                                              \
    ++g;                                                                                                    \
  }

  int g = 0;
#pragma unroll 1
  for (int ct = 0; ct < ntile; ++ct) {
    TW_BODY(true)
#pragma unroll 1
    for (int kt = 1; kt < nk; ++kt) TW_BODY(false)
    {
      // ---- epilogue of output tile ct, straight from the accumulators (see gemm_sp_kernel)
      int m0, n0;
      tile_origin(ct, m0, n0);
      const int lc = lane & 31, hi = lane >> 5;
      if constexpr (GEGLU) {
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          const int nc = n0 + wn * (32 * NT) + q * 64;
          half4_t bh[4], bg[4];
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            bh[gi] = half4_t{0, 0, 0, 0};
            bg[gi] = half4_t{0, 0, 0, 0};
            if (p.bias) {
              bh[gi] = *reinterpret_cast<const half4_t*>(p.bias + nc + 8 * gi + 4 * hi);
              bg[gi] = *reinterpret_cast<const half4_t*>(p.bias + nc + 32 + 8 * gi + 4 * hi);
            }
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * (32 * MT) + i * 32 + lc;
            const int mc = m < p.M ? m : p.M - 1;
            geglu_store32(acc[i][2 * q], acc[i][2 * q + 1], bh, bg, p.C + (size_t)mc * p.ldc + (nc >> 1), hi, m < p.M);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        if (p.residual) sp_plain_epilogue<MT, NT, true, false>(p, acc, m0 + wm * (32 * MT), n0 + wn * (32 * NT), lc, hi);
        else sp_plain_epilogue<MT, NT, false, false>(p, acc, m0 + wm * (32 * MT), n0 + wn * (32 * NT), lc, hi);
      }
    }
  }
  wait_vmcnt<0>();                                                   // the DMA pieces issued past the last tile
#undef TW_BODY
#undef TW_STEP
}

template <bool CONV, bool GEGLU>
static void launch_tw(GemmParams& p, hipStream_t stream) {
  constexpr int NT = GEGLU ? 4 : 5;
  constexpr int BM = 256, BN = 64 * NT;
  constexpr size_t smem = (size_t)4 * (BM + BN) * 64;
  md_ensure_dynamic_lds<gemm_tw_kernel<CONV, GEGLU, NT>>((int)smem);
  p.tiles_n = p.N / BN;
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_total = p.tiles_m * p.tiles_n;
  p.group_m = 8;
  const int ncu = md_device_cus();
  const int grid = p.tiles_total < ncu ? p.tiles_total : ncu;
  hipLaunchKernelGGL((gemm_tw_kernel<CONV, GEGLU, NT>), dim3(grid), dim3(512), smem, stream, p);
}
