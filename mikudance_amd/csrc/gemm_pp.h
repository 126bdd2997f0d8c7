// gemm_pp_kernel: the deep-pipelined "ping-pong" flavour of the MFMA GEMM / implicit-GEMM 3x3 convolution (included by
// gemm.hip; same operands, layouts, swizzle and epilogue math as gemm_kernel).
//
// Why a second structure: gemm_kernel hides its load latency with OTHER workgroups on the CU (2-3 small-LDS workgroups, one
// barrier per K step); that tops out at ~35 % of the MFMA peak on MI355X because each workgroup still waits out the full
// HBM/L2 latency once per K step.  Here ONE 512-thread workgroup owns the CU:
//   * tile 256 x (64*NJ) x 32 (NJ = 5: 256x320, divides every channel count of the UNets; NJ = 4: 256x256 for GEGLU whose
//     h|g column pairing needs 64-column groups per wave), 8 waves as 4 (M) x 2 (N), 64 x 32*NJ outputs per wave;
//   * a 4-deep LDS ring of 36-KiB (32-KiB) K tiles filled by direct-to-LDS DMA: three tiles are in flight behind the one
//     being consumed, waits are counted (s_waitcnt vmcnt(2*G)), never 0 while tiles remain;
//   * the two waves that share a SIMD (wave w and w+4) run one barrier apart: while one streams its 2*NJ+4 fragments of
//     tile kt from LDS into registers (and issues the DMA of tile kt+3), the other issues its 4*NJ MFMAs of that tile from
//     registers, then they swap -- the matrix pipe of every SIMD always has a wave with operands ready.
// Synchronisation (B_i = i-th s_barrier, L/M = fragment-load / MFMA slot):
//   group 0 (waves 0-3): B0 | L(0) B1 | M(0) B2 | L(1) B3 | M(1) B4 | ...
//   group 1 (waves 4-7): B0 |  --  B1 | L(0) B2 | M(0) B3 | L(1) B4 | ...
//   RAW: every wave retires its own DMA share of tile kt+1 (counted vmcnt) before B(2kt+2); the first reader of that tile
//        is group 0 in the slot after B(2kt+2).
//   WAR: the ring slot of tile kt-1 is refilled (tile kt+3) by group 0 after B(2kt) and by group 1 after B(2kt+1); its last
//        reader, group 1 in L(kt-1), retires its ds_reads (lgkmcnt(0)) before B(2kt).
#pragma once

#ifdef PP_TRACE
// diagnostic build only (tools/): shader-clock stamps of workgroup 0, waves 0 and 4, K tiles 8..8+PP_TRACE_N
#define PP_TRACE_N 12
__device__ unsigned long long g_pp_trace[2][PP_TRACE_N][6];
#define PP_STAMP(k)                                                                                         \
  if (tr_on && kt >= 8 && kt < 8 + PP_TRACE_N && lane == 0) g_pp_trace[grp][kt - 8][k] = __builtin_readcyclecounter();
extern "C" int md_debug_pp_trace(void* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_trace), sizeof(g_pp_trace));
}
#else
#define PP_STAMP(k)
#endif

template <int N2, int N1>
__device__ __forceinline__ void pp_wait_tiles(int tiles) {
  // `tiles` K tiles have been (partly) issued behind the one that must have landed: N2 / N1 / 0 DMA instructions of this
  // wave may stay in flight for tiles >= 2 / == 1 / <= 0
  if (tiles >= 2) wait_vmcnt<N2>();
  else if (tiles == 1) wait_vmcnt<N1>();
  else wait_vmcnt<0>();
}

template <bool CONV, bool GEGLU, int NJ, int PM>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmParams p) {
  constexpr int BK = 32, NST = 4, MI = 2;
  constexpr int BM = 256, BN = 64 * NJ;
  constexpr int NW = 8, T = 512;
  constexpr int ROWB = BK * 2;             // 64-byte tile rows
  constexpr int RPI = 1024 / ROWB;         // 16 rows per wave-wide DMA instruction
  constexpr int IPA = (BM / RPI) / NW;     // 2 A instructions per wave per tile
  constexpr int IPB = (BN / RPI) / NW;     // 2 W instructions per wave per tile ...
  constexpr int XB = BN / RPI - IPB * NW;  // ... plus one more on waves 0..XB-1 (NJ = 5: rows 256..319)
  static_assert(XB == 0 || XB == 4, "the extra W rows must fall on exactly the group-0 waves");
  constexpr int OPA = BM * ROWB;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int G0 = IPA + IPB + (XB ? 1 : 0), G1 = IPA + IPB;
  // DMA pieces 0..NP-1 of a tile (A rows, W rows, extra W rows); the first NP-PM go out in the fragment-load slot, the last
  // PM in the shadow of the MFMAs (the extra piece only exists on group 0, so PM = 1 gives both groups 4 pieces per L slot)
  constexpr int NP = G0, NL = NP - PM;
  constexpr int L1 = NL < G1 ? NL : G1;    // pieces group 1 issues in its L slot
  static_assert(PM >= 0 && PM <= NP, "PM");
  constexpr int CS_LD = BN + 4;
  static_assert(!GEGLU || NJ == 4, "GEGLU pairs 32-column sub-tiles (2q, 2q+1) of a wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Cs = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // the two staggered groups pair the waves that share a SIMD: wave w and w + 4 (checked with HW_REG_HW_ID on MI355X;
  // pairing w with w ^ 1 or w ^ 2 instead measured 20 % slower); xw = rank of this wave inside its group
  const int grp = wave >> 2;
  const int xw = wave & 3;
#ifdef PP_TRACE
  const bool tr_on = blockIdx.x == 0 && (wave & 3) == 0;
#endif

  int bid = blockIdx.x;
  {
    const int nwg = p.tiles_total;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-lane DMA sources (16-B slot of row r swizzled by (r >> 2) & 3 on the source side)
  const int lrow = lane >> 2, pslot = lane & 3;
  const half_t* a_src[IPA];
  const half_t* w_src[IPB + 1];
  int a_oy[IPA], a_ox[IPA];
#pragma unroll
  for (int j = 0; j < IPA; ++j) {
    const int row = (wave * IPA + j) * RPI + lrow;
    const int lslot = pslot ^ ((row >> 2) & 3);
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
  for (int j = 0; j < IPB + 1; ++j) {
    const int row = (j < IPB ? (wave * IPB + j) : (IPB * NW + xw)) * RPI + lrow;
    const int lslot = pslot ^ ((row >> 2) & 3);
    w_src[j] = p.W + (size_t)(n0 + row) * p.K + lslot * 8;     // N % BN == 0 (launcher), rows always valid for j < IPB
  }
  const half_t* zero_src = g_zero_page + 0;

  // filter tap of the next tile to issue (tiles are issued in K order): k0 = tap * Cin + c0
  int is_k0 = 0, is_c0 = 0, is_ky = 0, is_kx = 0;
  // one DMA instruction (piece) of the next tile: pieces 0..IPA-1 are A rows, IPA..IPA+IPB-1 W rows, IPA+IPB the extra W rows
  auto issue_piece = [&](int stage, int pc) {
    char* sa = smem + stage * STAGE + (wave * IPA) * 1024;
    char* sw = smem + stage * STAGE + OPA;
    if (pc < IPA) {
      const int j = pc;
      if (CONV) {
        const unsigned hup = p.Hin << p.upsample, wup = p.Win << p.upsample;
        const int iy = a_oy[j] + is_ky, ix = a_ox[j] + is_kx;
        const bool ok = (unsigned)iy < hup && (unsigned)ix < wup;
        unsigned off = __umul24(__umul24((unsigned)(iy >> p.upsample), (unsigned)p.Win) + (unsigned)(ix >> p.upsample), (unsigned)p.Cin) + is_c0;
        asm volatile("" : "+v"(off));
        const half_t* src = a_src[j] + off;
        src = ok ? src : zero_src;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + j * 1024), 16, 0, 0);
      } else {
        __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + is_k0), (lptr_t)(sa + j * 1024), 16, 0, 0);
      }
    } else if (pc < IPA + IPB) {
      const int j = pc - IPA;
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + is_k0), (lptr_t)(sw + (wave * IPB + j) * 1024), 16, 0, 0);
    } else if (XB && grp == 0) {
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[IPB] + is_k0), (lptr_t)(sw + (IPB * NW + xw) * 1024), 16, 0, 0);
    }
  };
  auto issue_advance = [&]() {
    if (CONV) {
      is_c0 += BK;
      if (is_c0 == p.Cin) {
        is_c0 = 0;
        if (++is_kx == 3) { is_kx = 0; ++is_ky; }
      }
    }
    is_k0 += BK;
  };
  auto issue_tile = [&](int stage) {
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) issue_piece(stage, pc);
    issue_advance();
  };
  auto issue_l_part = [&](int stage) {
#pragma unroll
    for (int pc = 0; pc < NL; ++pc) issue_piece(stage, pc);
    if (PM == 0) issue_advance();
  };

  floatx16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[MI], b_off[NJ], a_sw[MI], b_sw[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ra = wm * 64 + i * 32 + frow;
    a_off[i] = ra * ROWB;
    a_sw[i] = (ra >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int rb = wn * (32 * NJ) + j * 32 + frow;
    b_off[j] = OPA + rb * ROWB;
    b_sw[j] = (rb >> 2) & 3;
  }

  half8_t af[MI][2], bf[NJ][2];
  auto load_frags = [&](int stage) {
    const char* sb = smem + stage * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i][s] = *reinterpret_cast<const half8_t*>(sb + a_off[i] + (((s * 2 + fhi) ^ a_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j][s] = *reinterpret_cast<const half8_t*>(sb + b_off[j] + (((s * 2 + fhi) ^ b_sw[j]) << 4));
    }
  };
  // the 4*NJ MFMAs of one K tile; the last PM DMA pieces of tile kt + 3 ride in the shadow of the matrix pipe, one after
  // every fourth MFMA.  Measured on MI355X (s_memtime stamps, 8192x10240x8192): an L slot costs ~400 cycles for the 14
  // fragment reads plus ~90 per DMA piece, an M slot 640 + ~60 per piece, and the slower of the two sets the pace.
  auto mfma_tile = [&](bool issue, int stage) {
    __builtin_amdgcn_s_setprio(1);
    int n = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = GEGLU ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][s], af[i][s], acc[i][j], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
          if (PM > 0 && (n & 3) == 1 && (n >> 2) < PM) {
            __builtin_amdgcn_sched_barrier(0);
            if (issue) issue_piece(stage, NL + (n >> 2));
            __builtin_amdgcn_sched_barrier(0);
          }
          ++n;
        }
    __builtin_amdgcn_s_setprio(0);
    if (PM > 0 && issue) issue_advance();
  };

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue_tile(s);
  {
    const int fl = nk - 1 < 2 ? nk - 1 : 2;
    if (grp == 0) pp_wait_tiles<2 * G0, G0>(fl);
    else pp_wait_tiles<2 * G1, G1>(fl);
  }
  __builtin_amdgcn_s_barrier();                                     // B0
  if (grp == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      PP_STAMP(0)
      load_frags(kt & 3);
      if (kt + 3 < nk) issue_l_part((kt + 3) & 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_STAMP(1)
      __builtin_amdgcn_s_barrier();                                 // B(2kt+1)
      PP_STAMP(2)
      mfma_tile(kt + 3 < nk, (kt + 3) & 3);
      PP_STAMP(3)
      const int fl = nk - 2 - kt;
      pp_wait_tiles<2 * G0, G0>(fl);                                // tile kt+1 has landed (this wave's share); tiles kt+2, kt+3 may fly
      PP_STAMP(4)
      __builtin_amdgcn_s_barrier();                                 // B(2kt+2)
    }
  } else {
    __builtin_amdgcn_s_barrier();                                   // B1
    for (int kt = 0; kt < nk; ++kt) {
      PP_STAMP(0)
      load_frags(kt & 3);
      if (kt + 3 < nk) issue_l_part((kt + 3) & 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_STAMP(1)
      // in flight behind tile kt+1: all of tile kt+2 and the L-slot part of tile kt+3 (its M-slot part goes out below)
      const int fl = nk - 2 - kt;
      pp_wait_tiles<G1 + L1, G1>(fl);
      PP_STAMP(4)
      __builtin_amdgcn_s_barrier();                                 // B(2kt+2)
      PP_STAMP(2)
      mfma_tile(kt + 3 < nk, (kt + 3) & 3);
      PP_STAMP(3)
      if (kt < nk - 1) __builtin_amdgcn_s_barrier();                // B(2kt+3)
    }
  }

  if (GEGLU) {
    // straight from the accumulators (see gemm_kernel): acc = mfma(W frag, A frag), a lane owns row m = ..+lane%32 and
    // columns 8g + 4*(lane/32) + {0..3} of each 32-column sub-tile; sub-tile 2q is h, 2q+1 is g of the same outputs
    const int lc = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int q = 0; q < NJ / 2; ++q) {
      const int nc = n0 + wn * (32 * NJ) + q * 64;                  // first packed weight row of this h|g group
      half4_t bh[4], bg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bh[g] = half4_t{0, 0, 0, 0};
        bg[g] = half4_t{0, 0, 0, 0};
        if (p.bias) {
          bh[g] = *reinterpret_cast<const half4_t*>(p.bias + nc + 8 * g + 4 * hi);
          bg[g] = *reinterpret_cast<const half4_t*>(p.bias + nc + 32 + 8 * g + 4 * hi);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * 64 + i * 32 + lc;
        const int mc = m < p.M ? m : p.M - 1;
        geglu_store32(acc[i][2 * q], acc[i][2 * q + 1], bh, bg, p.C + (size_t)mc * p.ldc + (nc >> 1), hi, m < p.M);
      }
    }
    return;
  }

  // ---- plain epilogue: accumulators -> LDS (fp32, 64 rows per pass = the rows of wave row wm == pass) -> 16-byte pieces of
  // coalesced rows.  The launcher guarantees N % BN == 0 and 16-byte aligned C / residual / rowadd / bias rows.
  constexpr int TPR = BN / 8;              // 16-byte pieces per row
  constexpr int CH = 64 * TPR / T;         // pieces per thread per pass
  static_assert(64 * TPR % T == 0, "pass does not divide over the threads");
#pragma unroll 1
  for (int pass = 0; pass < BM / 64; ++pass) {
    const int mp = m0 + pass * 64;
    half8_t rv[CH];
    if (p.residual) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int id = tid + T * i;
        const int row = id / TPR, c8 = (id - row * TPR) * 8;
        const int m = mp + row;
        if (m < p.M) rv[i] = *reinterpret_cast<const half8_t*>(p.residual + (size_t)m * p.ldr + n0 + c8);
      }
    }
    __syncthreads();
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
            const int col = wn * (32 * NJ) + j * 32 + frow;
            Cs[row * CS_LD + col] = acc[i][j][r];
          }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int id = tid + T * i;
      const int row = id / TPR, c8 = (id - row * TPR) * 8;
      const int m = mp + row, n = n0 + c8;
      if (m >= p.M) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = Cs[row * CS_LD + c8 + j];
      if (p.bias) {
        const half8_t bv = *reinterpret_cast<const half8_t*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)bv[j];
      }
      if (p.rowadd) {
        const half8_t av = *reinterpret_cast<const half8_t*>(p.rowadd + (size_t)(m / p.rows_per_group) * p.ldra + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)av[j];
      }
      if (p.act == ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
      } else if (p.act == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if (p.act == ACT_QUICKGELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.0f + __expf(-1.702f * v[j]));
      }
      if (p.residual) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rv[i][j];
      }
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
      *reinterpret_cast<half8_t*>(p.C + (size_t)m * p.ldc + n) = o;
    }
  }
}

template <bool CONV, bool GEGLU>
static bool pp_eligible(const GemmParams& p) {
  constexpr int BN = GEGLU ? 256 : 320;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.transpose_out || p.N % BN != 0 || p.K % 32 != 0 || p.K < 128) return false;
  if (CONV && p.Cin % 32 != 0) return false;
  if (!al16(p.C) || p.ldc % 8 != 0) return false;
  if (p.bias && !al16(p.bias)) return false;
  if (p.residual && (!al16(p.residual) || p.ldr % 8 != 0)) return false;
  if (p.rowadd && (!al16(p.rowadd) || p.ldra % 8 != 0)) return false;
  return true;
}

template <bool CONV, bool GEGLU, int PM>
static void launch_pp_g(GemmParams& p, hipStream_t stream) {
  constexpr int NJ = GEGLU ? 4 : 5;
  constexpr int BN = 64 * NJ;
  constexpr size_t ring = (size_t)4 * (256 + BN) * 64;
  constexpr size_t cs = (size_t)64 * (BN + 4) * 4;
  constexpr size_t smem = ring > cs ? ring : cs;
  md_ensure_dynamic_lds<gemm_pp_kernel<CONV, GEGLU, NJ, PM>>((int)smem);
  p.tiles_n = p.N / BN;
  p.tiles_total = cdiv(p.M, 256) * p.tiles_n;
  hipLaunchKernelGGL((gemm_pp_kernel<CONV, GEGLU, NJ, PM>), dim3(p.tiles_total), dim3(512), smem, stream, p);
}

template <bool CONV, bool GEGLU>
static void launch_pp(GemmParams& p, hipStream_t stream) {
  launch_pp_g<CONV, GEGLU, 1>(p, stream);     // one DMA piece in the MFMA slot: 0-3 measured within +-2 %
}

// ------------------------------------------------------------------------------------------------ persistent GEGLU flavour
// gemm_ppg_kernel: the ping-pong structure above for the GEGLU GEMMs (256x256 tiles, plain A addressing), with ONE workgroup
// per CU that walks over its tiles and keeps the DMA ring running ACROSS tile boundaries: the first three K tiles of the next
// output tile are issued during the last three fragment-load slots of the current one, so the prologue latency (no other
// workgroup covers it here) disappears, and the GEGLU epilogues of the two wave groups -- straight from the accumulators,
// no LDS -- run concurrently in one slot while those DMAs are in flight.  K tiles are numbered g = 0 .. T-1 over all tiles
// of the workgroup; ring slot g & 3; barrier / vmcnt bookkeeping exactly as in gemm_pp_kernel with T in place of nk.
__global__ __launch_bounds__(512, 2) void gemm_ppg_kernel(GemmParams p) {
  constexpr int BK = 32, MI = 2, NJ = 4;
  constexpr int BM = 256, BN = 256;
  constexpr int ROWB = 64, RPI = 16;
  constexpr int IPA = 2, IPB = 2, G = IPA + IPB;
  constexpr int OPA = BM * ROWB, STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2;
  const int lrow = lane >> 2, pslot = lane & 3;
  const int nk = p.K / BK;
  const int nwg = p.tiles_total;
  const int ntile = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup
  const int T = ntile * nk;

  auto tile_origin = [&](int i, int& m0, int& n0) {
    // virtual workgroup id -> tile, same XCD-aware order as gemm_pp_kernel (gridDim.x % 8 == 0 or gridDim.x == nwg)
    const int v = (int)blockIdx.x + i * (int)gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = v & 7, idx = v >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    m0 = (t / p.tiles_n) * BM;
    n0 = (t % p.tiles_n) * BN;
  };

  // ---- issue side: tile `it`, K tile `ikt` of it, global K-tile counter gi
  const half_t* a_src[IPA];
  const half_t* w_src[IPB];
  auto set_sources = [&](int i) {
    int m0, n0;
    tile_origin(i, m0, n0);
#pragma unroll
    for (int j = 0; j < IPA; ++j) {
      const int row = (wave * IPA + j) * RPI + lrow;
      const int m = m0 + row;
      a_src[j] = p.A + (size_t)(m < p.M ? m : p.M - 1) * p.lda + (pslot ^ ((row >> 2) & 3)) * 8;
    }
#pragma unroll
    for (int j = 0; j < IPB; ++j) {
      const int row = (wave * IPB + j) * RPI + lrow;
      w_src[j] = p.W + (size_t)(n0 + row) * p.K + (pslot ^ ((row >> 2) & 3)) * 8;
    }
  };
  int it = 0, ikt = 0, gi = 0;
  set_sources(0);
  auto issue_next = [&]() {
    char* sa = smem + (gi & 3) * STAGE + (wave * IPA) * 1024;
    char* sw = smem + (gi & 3) * STAGE + OPA + (wave * IPB) * 1024;
    const int k0 = ikt * BK;
#pragma unroll
    for (int j = 0; j < IPA; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + k0), (lptr_t)(sa + j * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IPB; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + k0), (lptr_t)(sw + j * 1024), 16, 0, 0);
    ++gi;
    if (++ikt == nk) {
      ikt = 0;
      if (++it < ntile) set_sources(it);
    }
  };

  floatx16 acc[MI][NJ];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[MI], b_off[NJ], a_sw[MI], b_sw[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ra = wm * 64 + i * 32 + frow;
    a_off[i] = ra * ROWB;
    a_sw[i] = (ra >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int rb = wn * (32 * NJ) + j * 32 + frow;
    b_off[j] = OPA + rb * ROWB;
    b_sw[j] = (rb >> 2) & 3;
  }
  half8_t af[MI][2], bf[NJ][2];
  auto load_frags = [&](int stage) {
    const char* sb = smem + stage * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i][s] = *reinterpret_cast<const half8_t*>(sb + a_off[i] + (((s * 2 + fhi) ^ a_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j][s] = *reinterpret_cast<const half8_t*>(sb + b_off[j] + (((s * 2 + fhi) ^ b_sw[j]) << 4));
    }
  };
  auto mfma_tile = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // GEGLU epilogue of tile i, straight from the accumulators (see gemm_pp_kernel)
  auto epilogue = [&](int i_tile) {
    int m0, n0;
    tile_origin(i_tile, m0, n0);
    const int lc = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int q = 0; q < NJ / 2; ++q) {
      const int nc = n0 + wn * (32 * NJ) + q * 64;
      half4_t bh[4], bg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bh[g] = half4_t{0, 0, 0, 0};
        bg[g] = half4_t{0, 0, 0, 0};
        if (p.bias) {
          bh[g] = *reinterpret_cast<const half4_t*>(p.bias + nc + 8 * g + 4 * hi);
          bg[g] = *reinterpret_cast<const half4_t*>(p.bias + nc + 32 + 8 * g + 4 * hi);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * 64 + i * 32 + lc;
        const int mc = m < p.M ? m : p.M - 1;
        geglu_store32(acc[i][2 * q], acc[i][2 * q + 1], bh, bg, p.C + (size_t)mc * p.ldc + (nc >> 1), hi, m < p.M);
      }
    }
  };

  // The stores of an epilogue also count in vmcnt and are not ordered against loads, so the FIRST wait after an epilogue
  // drains everything (vmcnt(0): the stores are a slot old by then, the two youngest K tiles are made to land early --
  // once per output tile); every other wait is counted as in gemm_pp_kernel.
  bool drain = false;
  auto wait_landed = [&](int tiles_behind) {
    if (drain) {
      wait_vmcnt<0>();
      drain = false;
    } else {
      pp_wait_tiles<2 * G, G>(tiles_behind);
    }
  };
#pragma unroll 1
  for (int s = 0; s < 3; ++s)
    if (s < T) issue_next();
  pp_wait_tiles<2 * G, G>(T - 1);
  __builtin_amdgcn_s_barrier();                                     // B0
  int kt = 0, ct = 0;
  if (grp == 0) {
#pragma unroll 1
    for (int g = 0; g < T; ++g) {
      if (kt == 0) {
        if (g > 0) {
          epilogue(ct++);                                           // previous tile, concurrent with group 1's (slot 2g)
          drain = true;
        }
        zero_acc();
      }
      load_frags(g & 3);
      if (g + 3 < T) issue_next();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                 // B(2g+1)
      mfma_tile();
      wait_landed(T - 2 - g);
      __builtin_amdgcn_s_barrier();                                 // B(2g+2)
      if (++kt == nk) kt = 0;
    }
    epilogue(ct);
  } else {
    __builtin_amdgcn_s_barrier();                                   // B1
#pragma unroll 1
    for (int g = 0; g < T; ++g) {
      if (kt == 0) zero_acc();
      load_frags(g & 3);
      if (g + 3 < T) issue_next();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wait_landed(T - 2 - g);
      __builtin_amdgcn_s_barrier();                                 // B(2g+2)
      mfma_tile();
      if (++kt == nk) {
        kt = 0;
        epilogue(ct++);
        drain = true;
      }
      if (g < T - 1) __builtin_amdgcn_s_barrier();                  // B(2g+3)
    }
  }
}

static void launch_ppg(GemmParams& p, hipStream_t stream) {
  constexpr size_t smem = (size_t)4 * (256 + 256) * 64;
  md_ensure_dynamic_lds<gemm_ppg_kernel>((int)smem);
  const int ncu = md_device_cus();                                // MI355X: 256 CUs (one persistent workgroup each), per device
  p.tiles_n = p.N / 256;
  p.tiles_total = cdiv(p.M, 256) * p.tiles_n;
  const int grid = p.tiles_total < ncu ? p.tiles_total : ncu;
  hipLaunchKernelGGL(gemm_ppg_kernel, dim3(grid), dim3(512), smem, stream, p);
}
