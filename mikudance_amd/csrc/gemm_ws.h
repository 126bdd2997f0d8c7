// W-stationary streaming GEMM for the HBM-bound short-K projections (included by gemm.hip).
//
//   C[M, N] = epi( A[M, K] . W[N, K]^T ),   K = 320 or 640, N a multiple of the column group (320 / 128), M large, M % 16 == 0.
//
// The level-0 / level-1 Linear layers of both UNets (to_q / to_k / to_out, proj_in / proj_out, the motion module's qkv and
// out projections, FeedForward net.0 at C = 320: reference src/models/attention.py:109-157,323-364,
// src/models/transformer_3d.py:66-98, src/models/motion_module.py:124-146,293-317) have K = N = C = 320 / 640 on M = 294 912 /
// 73 728 rows: 160 FLOP per byte of A + C, half the machine balance, i.e. they are HBM streams (377 MB per launch at C = 320)
// with a small GEMM attached.  The tiled kernel reloads the weight tile with every output tile and waits on a 5-step K loop per
// tile (2.4-3.6 TB/s).  Here the roles are turned around:
//   * the WEIGHTS are stationary in REGISTERS: one persistent 512-thread workgroup per CU, its four compute waves (one per
//     SIMD) each own 16*CB output columns x all of K as MFMA operand fragments (<= 200 VGPRs), loaded once and drained with the
//     s_waitcnt builtin the compiler's waitcnt pass understands (left alone it waits lazily at each fragment's first use, i.e.
//     it plants ~50 `s_waitcnt vmcnt(N)` between the MFMAs of the tile loop; every extra issue slot between two MFMAs of a
//     one-wave-per-SIMD stream costs far more than its own cycle);
//   * A streams HBM -> LDS through a ring of ROUND stages (TPR 16-row tiles each) filled by direct-to-LDS DMA
//     (global_load_lds_dwordx4), 40-100 KiB in flight per CU;
//   * waves are specialised so that every counted s_waitcnt sees ONE kind of memory operation:
//       waves 0-3  compute : LDS fragment reads (kept PD deep with sched_group_barrier, two per-lane base addresses + immediate
//                            offsets: no address VALU between MFMAs) + v_mfma_f32_16x16x32_f16 (C^T = W.A^T, so a lane ends up
//                            with 4 consecutive output columns of one row), fp32 results to an LDS staging tile
//       waves 4-5  loaders : issue the DMA of round r+NR-1 (tile-invariant 64-bit source pointers: one add per piece) and wait --
//                            their vmcnt counts nothing but their own DMAs, so the counted wait is exact -- for round r+1
//       waves 6-7  stores  : round r-1: staging tiles + bias (+ row-broadcast term) (+ residual, prefetched RD tiles ahead)
//                            -> one rounding -> coalesced 16-byte global stores
//     ONE s_barrier per round hands the stages round.  A raw s_barrier does not wait for a wave's LDS stores (gfx950 has the
//     back-off barrier, the compiler adds no s_waitcnt), so the compute waves drain lgkmcnt before it.
//   * TPR = 1 (deepest DMA ring) for the single-group launches (N = 320: streams A from HBM, sits on the store path) and for
//     GEGLU (bound by the GELU VALU work that compute and store waves do on the same SIMDs); TPR = 2 for the multi-group plain
//     launches, whose A comes out of L2: two tiles per barrier round let the staging stores of the first tile and the fragment
//     latency of the second overlap the MFMAs (K = 640: +6-10 %, K = 320 N = 960: +2 %).
//   * a launch with G = N / group column groups runs G workgroups side by side on the same row stream inside one XCD, so
//     the G-1 re-reads of an A stage hit that XCD's L2 (PMC: 189 MB fetched at N = 960, exactly A).
// LDS bank conflicts: the DMA writes a stage lane-linearly, so the swizzle is applied to the per-lane SOURCE address and to
// the fragment read (same involution): 16-byte slot c of row r sits at c ^ s(r), s(r) = (r >> 1) & 7 for 40 slots per row
// (K = 320: consecutive rows already shift by 8 slots) and r & 15 for 80 (K = 640): every ds_read_b128 of a 16-row x 32-k
// fragment touches 16 distinct slots per lane group.
#pragma once
#ifndef WS_RES_DEPTH
#define WS_RES_DEPTH 4         // residual tiles in flight per store wave
#endif

struct WsParams {
  const half_t* A;
  const half_t* W;
  half_t* C;
  const half_t* bias;
  const half_t* residual;
  const half_t* rowadd;
  int lda, ldc, ldr, ldra;
  int M, N;
  int rows_per_group;
  int groups;        // column groups G
  int streams;       // row streams (workgroups per column group)
  int spx;           // row streams per XCD
};

template <int KS, int CB, int TPR>
struct WsCfg {
  static constexpr int K = 32 * KS;
  static constexpr int CPR = K / 8;               // 16-byte slots per A row
  static constexpr int GC = 64 * CB;              // output (GEGLU: packed weight) columns per workgroup
  static constexpr int TR = 16;                   // rows per tile
  static constexpr int STAGE = TR * K * 2;        // bytes per A tile
  static constexpr int RSTAGE = TPR * STAGE;      // ... per round
  static constexpr bool GEGLU_CFG = CB == 4;      // the GEGLU flavour is the only one with 4 column blocks per wave
  static constexpr int CS_LD = (GEGLU_CFG ? GC / 2 : GC) + 4;   // fp32 staging pitch (floats): rows shift by 4 banks; GEGLU stages
                                                  // only column pair 1 (h1 | g1: 32 floats per compute wave)
  static constexpr int CSTAGE = TR * CS_LD * 4;   // one staging tile; 2 * TPR of them (double buffered rounds)
  static constexpr int OSTAGE = GEGLU_CFG ? 2 * TPR * 2048 : 0;   // GEGLU: fp16 outputs finished by the compute waves (2 KiB per tile)
  static constexpr int NR_FIT = (160 * 1024 - 2 * TPR * CSTAGE - OSTAGE) / RSTAGE;
  static constexpr int NR = NR_FIT > 12 ? 12 : NR_FIT;       // ring depth in rounds
  static constexpr int DPT = STAGE / 1024;        // DMA wave-instructions per tile
  static constexpr int PER = DPT / 2;             // ... per loader wave
  static constexpr int SMEM = NR * RSTAGE + 2 * TPR * CSTAGE + OSTAGE;
  static constexpr int CHUNKS = TR * GC / 8;      // 16-byte output pieces per tile
  static constexpr int SPL = CHUNKS / 128;        // ... per lane of the two store waves
  static constexpr int PD = KS <= 10 ? 3 : 6;     // A fragments in flight per compute wave (register budget: 256 per wave)
  static_assert(DPT % 2 == 0 && CHUNKS % 128 == 0, "tile must split evenly over the loader / store waves");
  static_assert(NR >= 3 && (NR - 2) * TPR * PER <= 63, "ring depth / vmcnt immediate");
  static_assert(WS_RES_DEPTH % TPR == 0, "residual buffers are indexed statically per unrolled round");
};

template <int CPR>
__device__ __forceinline__ int ws_swz(int row) {
  return CPR == 40 ? ((row >> 1) & 7) : (row & 15);
}

template <int KS, int CB, int TPR, bool RES, bool RA, bool GEGLU = false>
__global__ __launch_bounds__(512, 1) void wsgemm_kernel(WsParams p) {
  static_assert(!GEGLU || (CB == 4 && !RES && !RA), "GEGLU: 2 h + 2 g column blocks per compute wave, bias only");
  using Cfg = WsCfg<KS, CB, TPR>;
  constexpr int K = Cfg::K, CPR = Cfg::CPR, GC = Cfg::GC, TR = Cfg::TR, STAGE = Cfg::STAGE, RSTAGE = Cfg::RSTAGE, NR = Cfg::NR,
                CS_LD = Cfg::CS_LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + NR * RSTAGE);       // staging tile (round parity, tile u): index (r & 1) * TPR + u
  char* ost = smem + NR * RSTAGE + 2 * TPR * Cfg::CSTAGE;          // GEGLU: fp16 pieces, same indexing, 2 KiB each
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // workgroup -> (XCD, column group, row stream): the G groups of one row stream share an XCD (blockIdx % 8)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = slot % p.groups, sl = slot / p.groups;
  if (sl >= p.spx) return;
  const int stream = xcd * p.spx + sl;
  const int ntiles = (p.M + TR - 1) / TR;
  const int my_tiles = stream < ntiles ? (ntiles - stream + p.streams - 1) / p.streams : 0;   // tiles stream, stream+S, ...
  if (my_tiles == 0) return;
  const int rounds = (my_tiles + TPR - 1) / TPR;                  // barriers b_0 .. b_rounds
  const int n0 = grp * GC;

  if (wave < 4) {
    // ------------------------------------------------------------------------------------------------ compute waves
    half8_t wf[CB][KS];
    {
      const half_t* wp = p.W + (size_t)(n0 + wave * 16 * CB + (lane & 15)) * K + 8 * (lane >> 4);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[cb][ks] = *reinterpret_cast<const half8_t*>(wp + (size_t)cb * 16 * K + 32 * ks);
    }
    const int row = lane & 15, kq = lane >> 4;
    const int rbase = row * CPR, sw = ws_swz<CPR>(row);
    // GEGLU: the exact-erf GELU is split between the waves that have slack: every compute wave finishes column pair 0 (its h
    // block 0 and g block 0: 4 outputs per lane, fp16 straight into LDS), the two store waves finish pair 1 and ship both
    float gb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (GEGLU) {
      if (p.bias) {
        const half4_t b0 = *reinterpret_cast<const half4_t*>(p.bias + n0 + wave * 64 + 4 * kq);
        const half4_t b1 = *reinterpret_cast<const half4_t*>(p.bias + n0 + wave * 64 + 32 + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) gb[r] = (float)b0[r], gb[4 + r] = (float)b1[r];
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): weights (and bias) are in registers before the loop
    constexpr int PD = Cfg::PD;
    constexpr int SWB = CPR == 40 ? 3 : 4, P = (1 << SWB) / 4;
    auto compute_tile = [&](const char* st, int buf) {
      floatx4 acc[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
      // slot (4ks + kq) ^ sw: the swizzle only touches the low SWB bits, so there are P = 2^SWB / 4 distinct per-lane base
      // addresses per tile and every fragment read is base[ks % P] + a compile-time offset
      half8_t af[PD];
      const char* fb[P];
#pragma unroll
      for (int j = 0; j < P; ++j) fb[j] = st + (rbase + ((4 * j + kq) ^ sw)) * 16;
      auto frag = [&](int ks) { return *reinterpret_cast<const half8_t*>(fb[ks % P] + (ks / P) * (16 << SWB)); };
#pragma unroll
      for (int j = 0; j < PD; ++j) af[j] = frag(j);
      __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);                         // PD ds_reads first ...
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8_t cur = af[ks % PD];
        if (ks + PD < KS) af[ks % PD] = frag(ks + PD);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cb][ks], cur, acc[cb], 0, 0, 0);
        if (ks + PD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // ... then one refill per k step, issued
        __builtin_amdgcn_sched_group_barrier(0x008, CB, 0);                       //     ahead of that step's CB MFMAs
      }
      // acc[cb][r] = C[m = row][n = wave*16CB + cb*16 + 4*kq + r]
      if constexpr (GEGLU) {
        // blocks 0, 1 = h columns 0-15, 16-31 of this wave's 32 outputs, blocks 2, 3 = the matching g columns
        half4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)((acc[0][r] + gb[r]) * gelu_fast(acc[2][r] + gb[4 + r]));
        *reinterpret_cast<half4_t*>(ost + buf * 2048 + row * 128 + (wave * 16 + 4 * kq) * 2) = o;
        float* cs = cst + buf * (TR * CS_LD) + row * CS_LD + wave * 32 + 4 * kq;
        *reinterpret_cast<floatx4*>(cs) = acc[1];
        *reinterpret_cast<floatx4*>(cs + 16) = acc[3];
      } else {
        float* cs = cst + buf * (TR * CS_LD) + row * CS_LD + wave * 16 * CB + 4 * kq;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<floatx4*>(cs + cb * 16) = acc[cb];
      }
    };
    for (int r = 0; r < rounds; ++r) {
      __builtin_amdgcn_s_barrier();                 // b_r: round r has landed; the staging tiles of parity r & 1 are free
      const char* st = ring + (r % NR) * RSTAGE;
#pragma unroll
      for (int u = 0; u < TPR; ++u)
        if (r * TPR + u < my_tiles) compute_tile(st + u * STAGE, (r & 1) * TPR + u);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging stores WRITTEN before the hand-over barrier
    }
    __builtin_amdgcn_s_barrier();                   // b_rounds
  } else if (wave < 6) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    const half_t* sp[Cfg::PER];
    const size_t astep = (size_t)p.streams * TR * p.lda;
#pragma unroll
    for (int i = 0; i < Cfg::PER; ++i) {
      const int pidx = (lw * Cfg::PER + i) * 64 + lane;
      const int r = pidx / CPR, c = pidx % CPR;
      sp[i] = p.A + (size_t)(stream * TR + r) * p.lda + ((c ^ ws_swz<CPR>(r)) << 3);
    }
    auto issue_round = [&](int q) {                 // called with q = 0, 1, 2, ... in order; tiles are issued in order too
      char* st = ring + (q % NR) * RSTAGE;
#pragma unroll
      for (int u = 0; u < TPR; ++u) {
        if (q * TPR + u < my_tiles) {
#pragma unroll
          for (int i = 0; i < Cfg::PER; ++i) {
            const int base = __builtin_amdgcn_readfirstlane((lw * Cfg::PER + i) * 64);   // first 16-byte slot of this instruction
            __builtin_amdgcn_global_load_lds((gptr_t)sp[i], (lptr_t)(st + u * STAGE + base * 16), 16, 0, 0);
            sp[i] += astep;
          }
        }
      }
    };
    const int pre = rounds < NR - 1 ? rounds : NR - 1;
    for (int q = 0; q < pre; ++q) issue_round(q);
    for (int r = 0; r < rounds; ++r) {
      // round r must have landed before the barrier.  Issued so far: rounds <= r+NR-2; all of r+1 .. r+NR-2 are FULL rounds
      // (TPR tiles, TPR*PER pieces each) as long as none of them is the last one -- otherwise simply drain.
      if (r + NR - 2 < rounds - 1) wait_vmcnt<(NR - 2) * TPR * Cfg::PER>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                 // also: the compute waves are done with round r-1 -> its stage is free
      if (r + NR - 1 < rounds) issue_round(r + NR - 1);   // into stage (r - 1) % NR
    }
    __builtin_amdgcn_s_barrier();                   // b_rounds
  } else {
    // ------------------------------------------------------------------------------------------------ store waves
    if constexpr (GEGLU) {
      // FeedForward net.0 (GEGLU, reference src/models/attention.py:152-157 / diffusers FeedForward): the weight rows are packed
      // in blocks of 32 h rows then 32 g rows (packing.geglu_weight), so compute wave w owns exactly one block = output columns
      // [32w, 32w+32) of this workgroup's 128.  out = (h + b_h) * gelu_erf(g + b_g), one rounding, 16-byte stores into the
      // [M][N/2] output.  Per tile this lane ships TWO pieces of its row: columns 32w + 8*half (pair 0, finished by compute wave w,
      // fp16 in LDS) and 32w + 16 + 8*half (pair 1: from the fp32 staging tile, GELU done here).
      constexpr int OC = GC / 2;
      const int sid = (wave - 6) * 64 + lane;                         // 0..127 = 16 rows x 4 waves x 2 halves
      const int prow = sid >> 3, pw = (sid >> 1) & 3, half = sid & 1;
      float bh[8], bg[8];
      {
        const half8_t b0 = p.bias ? *reinterpret_cast<const half8_t*>(p.bias + n0 + pw * 64 + 16 + 8 * half) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        const half8_t b1 = p.bias ? *reinterpret_cast<const half8_t*>(p.bias + n0 + pw * 64 + 48 + 8 * half) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) bh[j] = (float)b0[j], bg[j] = (float)b1[j];
      }
      const size_t cstep = (size_t)p.streams * TR * p.ldc;
      half_t* cp0 = p.C + (size_t)(stream * TR + prow) * p.ldc + grp * OC + pw * 32 + 8 * half;
      for (int r = 0; r <= rounds; ++r) {
        __builtin_amdgcn_s_barrier();                                   // b_r
        if (r >= 1) {
#pragma unroll
          for (int u = 0; u < TPR; ++u) {
            if ((r - 1) * TPR + u < my_tiles) {
              const int buf = ((r - 1) & 1) * TPR + u;
              const float* q = cst + buf * (TR * CS_LD) + prow * CS_LD + pw * 32 + 8 * half;
              const floatx4 h0 = *reinterpret_cast<const floatx4*>(q), h1 = *reinterpret_cast<const floatx4*>(q + 4);
              const floatx4 g0 = *reinterpret_cast<const floatx4*>(q + 16), g1 = *reinterpret_cast<const floatx4*>(q + 20);
              const half8_t first = *reinterpret_cast<const half8_t*>(ost + buf * 2048 + prow * 128 + (pw * 16 + 8 * half) * 2);
              *reinterpret_cast<half8_t*>(cp0) = first;
              half8_t o;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                o[j] = (half_t)((h0[j] + bh[j]) * gelu_fast(g0[j] + bg[j]));
                o[j + 4] = (half_t)((h1[j] + bh[j + 4]) * gelu_fast(g1[j] + bg[j + 4]));
              }
              *reinterpret_cast<half8_t*>(cp0 + 16) = o;
              cp0 += cstep;
            }
          }
        }
      }
    } else {
      // Lean on purpose: two waves move every output byte of the workgroup, so per 16-byte piece the loop is 2 LDS reads, 8 fp32
      // adds per epilogue term, 4 packed converts, a 64-bit pointer bump and the store.  Pointers advance by a constant per tile;
      // bias / row-broadcast terms live in registers as floats.
      const int sid = (wave - 6) * 64 + lane;         // 0..127
      constexpr int CPRO = GC / 8;                    // 16-byte pieces per output row
      constexpr int SPL = Cfg::SPL;
      int prow[SPL], pcol[SPL];
      float biasf[SPL][8];
      half_t* cp[SPL];
      const half_t* rp[SPL];
      const size_t cstep = (size_t)p.streams * TR * p.ldc, rstep = (size_t)p.streams * TR * p.ldr;
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        const int id = i * 128 + sid;
        prow[i] = id / CPRO;
        pcol[i] = (id % CPRO) * 8;
        const half8_t bv = p.bias ? *reinterpret_cast<const half8_t*>(p.bias + n0 + pcol[i]) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) biasf[i][j] = (float)bv[j];
        cp[i] = p.C + (size_t)(stream * TR + prow[i]) * p.ldc + n0 + pcol[i];
        rp[i] = RES ? p.residual + (size_t)(stream * TR + prow[i]) * p.ldr + n0 + pcol[i] : nullptr;
      }
      constexpr int RD = RA ? 2 : WS_RES_DEPTH;       // residual tiles in flight per store wave (fewer when the row-broadcast term
                                                      // also lives in registers: 256 VGPRs per wave)
      static_assert(RD % TPR == 0, "residual buffers are indexed statically per unrolled round");
      constexpr int UNR = RD / TPR;                   // rounds per unrolled loop body
      half8_t res[RES ? RD : 1][SPL];
      auto fetch_res = [&](half8_t (&dst)[SPL]) {     // the next tile of this stream in order; rp[] advances
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          dst[i] = *reinterpret_cast<const half8_t*>(rp[i]);
          rp[i] += rstep;
        }
      };
      int cur_group = -1;
      float raf[RA ? SPL : 1][8];
      auto store_tile = [&](int tile, int buf, const half8_t (&rs)[SPL]) {
        const int m0 = (stream + tile * p.streams) * TR;
        const float* cs = cst + buf * (TR * CS_LD);
        if constexpr (RA) {
          // row-broadcast term: one table row per `rows_per_group` output rows (a frame); reloaded when the tile enters a new
          // group.  Tiles that straddle two groups take the per-piece path.
          const int g0 = m0 / p.rows_per_group, g1 = (m0 + TR - 1) / p.rows_per_group;
          if (g0 != cur_group || g1 != g0) {
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
              const int g = g1 == g0 ? g0 : (m0 + prow[i]) / p.rows_per_group;
              const half8_t ra = *reinterpret_cast<const half8_t*>(p.rowadd + (size_t)g * p.ldra + n0 + pcol[i]);
#pragma unroll
              for (int j = 0; j < 8; ++j) raf[i][j] = (float)ra[j];
            }
            cur_group = g1 == g0 ? g0 : -1;
          }
        }
        // all LDS reads first (one latency per tile, not one per piece), then the arithmetic
        floatx4 ca[SPL], cb2[SPL];
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          ca[i] = *reinterpret_cast<const floatx4*>(cs + prow[i] * CS_LD + pcol[i]);
          cb2[i] = *reinterpret_cast<const floatx4*>(cs + prow[i] * CS_LD + pcol[i] + 4);
        }
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          float v[8] = {ca[i][0], ca[i][1], ca[i][2], ca[i][3], cb2[i][0], cb2[i][1], cb2[i][2], cb2[i][3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += biasf[i][j];
          if constexpr (RA) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += raf[i][j];
          }
          if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rs[i][j];
          }
          half8_t o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
          *reinterpret_cast<half8_t*>(cp[i]) = o;
          cp[i] += cstep;
        }
      };
      if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < RD; ++j)
          if (j < my_tiles) fetch_res(res[j]);        // tiles 0 .. RD-1
      }
      // barrier b_r (r = 0 .. rounds): afterwards the compute waves work on round r and this wave stores round r-1, then prefetches
      // the residual of the tile RD places further into the buffer it has just freed (RD tile periods ahead: the HBM latency
      // under load is several tile periods).  Unrolled by UNR rounds so that the residual buffers are addressed statically:
      // tile t uses res[t % RD], and (r - 1) % UNR == (j + UNR - 1) % UNR because base % UNR == 0.
      for (int base = 0; base <= rounds; base += UNR) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
          const int r = base + j;
          if (r <= rounds) {
            __builtin_amdgcn_s_barrier();                             // b_r
            if (r >= 1) {
#pragma unroll
              for (int u = 0; u < TPR; ++u) {
                const int tile = (r - 1) * TPR + u;
                const int slot_ = ((j + UNR - 1) % UNR) * TPR + u;    // == tile % RD, a compile-time value after unrolling
                if (tile < my_tiles) {
                  store_tile(tile, ((r - 1) & 1) * TPR + u, res[RES ? slot_ : 0]);
                  if constexpr (RES) {
                    if (tile + RD < my_tiles) fetch_res(res[slot_]);
                  }
                }
              }
            }
          }
        }
      }
    }
  }
}
