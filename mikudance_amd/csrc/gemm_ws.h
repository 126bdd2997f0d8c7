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
//     GEGLU (bound by the GELU VALU work, which the four MEMORY waves share, one per SIMD, so that the MFMA-issuing waves
//     carry none of it); TPR = 2 for the multi-group plain
//     launches, whose A comes out of L2: two tiles per barrier round let the staging stores of the first tile and the fragment
//     latency of the second overlap the MFMAs (K = 640: +6-10 %, K = 320 N = 960: +2 %).
//   * a launch with G = N / group column groups runs G workgroups side by side on the same row stream inside one XCD, so
//     the G-1 re-reads of an A stage hit that XCD's L2 (PMC: 189 MB fetched at N = 960, exactly A).
// LDS bank conflicts: the DMA writes a stage lane-linearly, so the swizzle is applied to the per-lane SOURCE address and to
// the fragment read (same involution): 16-byte slot c of row r sits at c ^ s(r), s(r) = (r >> 1) & 7 for 40 slots per row
// (K = 320: consecutive rows already shift by 8 slots) and r & 15 for 80 (K = 640): every ds_read_b128 of a 16-row x 32-k
// fragment touches 16 distinct slots per lane group.
#pragma once
#ifndef WS_ABL
#define WS_ABL 0               // DIAGNOSTIC builds only (tools/build_ab.sh, wrong results): 1 no MFMAs | 2 no fragment reads | 4 no global stores | 8 store waves idle |
#endif                         // 16 no DMA after the first ring fill | 32 no staging writes   (profiles/r06_ws_ablation.log)
#ifndef WS_RES_DEPTH
#define WS_RES_DEPTH 4         // residual tiles in flight per store wave
#endif
#ifndef WS_RES_DEPTH_640
#define WS_RES_DEPTH_640 8              // ... of the K = 640 flavour (2 pieces per lane and tile instead of 5: the registers allow more)
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
  int xstreams;      // extra row streams made of the CUs that G column groups x spx streams leave idle on every XCD (round 6, ws_streams)
  // prologue flavours (PRO != 0): a normalisation of the A rows that never exists in HBM
  const float* lnf;  // PRO_LNF: [2][N] fp32 -- s[n] = sum_k W'[n][k] and c[n] = sum_k beta[k] W[n][k] + bias[n] of the folded LayerNorm
  float eps;         // PRO_LNF: LayerNorm epsilon
  const float* aff;  // PRO_AFF: [M / rows_per_image][2][K] fp32 -- per (image, input channel) scale then shift (GroupNorm apply)
  int rows_per_image;
};

// Prologues.  Both rest on the fact that a stage of this kernel holds WHOLE rows of A (all of K) in LDS before the matrix core reads it.
//   PRO_LNF  LayerNorm folded into the Linear that consumes it (norm2 -> attn2.to_q, the motion module's norms -> q|k|v: reference
//            src/models/attention.py:131-141,339-347, src/models/motion_module.py:245-268).  With
//            W'[n][k] = fp16(gamma[k] W[n][k]),  s[n] = sum_k W'[n][k],  c[n] = sum_k beta[k] W[n][k] + bias[n]:
//                LN(x) . W^T + bias = rstd * (x . W'^T - mu * s) + c
//            so the GEMM runs on the RAW rows and the normalised tensor is never written or read (4 bytes per element and a launch
//            saved per LayerNorm).  The loader waves, which own eight rows of every tile they have just DMA'd, compute the exact two-pass
//            (mu, rstd) of their rows from the landed stage (sum on v_dot2, 8-lane reductions on DPP) and leave a = rstd, b = -mu * rstd
//            in a small LDS ring; the store waves apply a * acc + (b * s + c).  s is summed from the ROUNDED W', so x . W'^T - mu * s is
//            exactly sum_k (x_k - mu) W'[n][k]: the fold is as insensitive to the row mean as the two-pass LayerNorm it replaces.
//   PRO_AFF  per-(image, channel) affine x * scale + shift applied to the landed stage IN PLACE, one rounding to fp16 -- GroupNorm's
//            apply sweep in front of proj_in (reference src/models/transformer_3d.py:60-68,121-137, src/models/motion_module.py:121-124,
//            159-170) with scale = rstd * gamma, shift = beta - mean * scale from md_groupnorm_table_f16: bit-identical to
//            gn_apply_kernel followed by this GEMM, minus the 4 bytes per element the normalised tensor cost.  The table of the lane's
//            columns lives in registers and changes with the image, so PRO_AFF streams walk CONTIGUOUS row blocks (a stream meets at
//            most ceil(rows per stream / rows per image) + 1 images) instead of the interleaved tiles of the other flavours.
// Measured in round 5 and REMOVED again (profiles/r05_ab_fused_norms_statistics_chain.log): row statistics emitted by the producing
// out-projection's store waves (a row's pieces in 8 adjacent lanes, exact two-pass, 8 bytes per row to a side buffer) and consumed by the
// fold (16 pairs DMA'd beside the A tile), which let the GEGLU flavour take the fold.  The consumers became free (q|k|v 0.246 vs 0.261 ms,
// GEGLU 0.626 vs 0.618 + a 0.082-ms LayerNorm), the producers paid +0.02 ms each for ~110 VALU per tile and lane in the store waves, and in
// the denoising loop the folded GEGLU lost the memory-side-cache warmth the LayerNorm pass used to leave behind (0.653 vs 0.561 ms per
// launch): +0.5 % end to end where in-kernel statistics on the plain consumers alone are worth more.  Every K = 640 form and GEGLU with
// in-kernel statistics lost outright (the loader waves' work per 20-KiB tile, redone by each of 5-15 column groups, doubles the tile time).
enum { PRO_NONE = 0, PRO_LNF = 1, PRO_AFF = 2 };

template <int KS, int CB, int TPR>
struct WsCfg {
  static constexpr int K = 32 * KS;
  static constexpr int CPR = K / 8;               // 16-byte slots per A row
  static constexpr int GC = 64 * CB;              // output (GEGLU: packed weight) columns per workgroup
  static constexpr int TR = 16;                   // rows per tile
  static constexpr int STAGE = TR * K * 2;        // bytes per A tile
  static constexpr int RSTAGE = TPR * STAGE;      // ... per round
  static constexpr bool GEGLU_CFG = CB == 4;      // the GEGLU flavour is the only one with 4 column blocks per wave
  static constexpr int CS_LD = GC + 4;            // fp32 staging pitch (floats): rows shift by 4 banks
  static constexpr int CSTAGE = TR * CS_LD * 4;   // one staging tile; 2 * TPR of them (double buffered rounds)
  static constexpr int OSTAGE = GEGLU_CFG ? 2 * 2048 : 0;   // GEGLU: fp16 pieces finished by the loader waves, shipped by the store waves
  static constexpr int SSLOT = TR;                          // float2 per (round, tile) slot of the statistics ring: one per row
  static constexpr int SSTAGE = 4 * TPR * SSLOT * 8;        // PRO_LNF: (a, b) per row, 4 rounds deep (written in round r-1, read in r and r+1)
  static constexpr int NR_FIT = (160 * 1024 - 2 * TPR * CSTAGE - OSTAGE - SSTAGE) / RSTAGE;
  static constexpr int NR = NR_FIT > 12 ? 12 : NR_FIT;       // ring depth in rounds
  static constexpr int DPT = STAGE / 1024;        // DMA wave-instructions per tile
  static constexpr int PER = DPT / 2;             // ... per loader wave
  static constexpr int SMEM = NR * RSTAGE + 2 * TPR * CSTAGE + OSTAGE + SSTAGE;
  static constexpr int CHUNKS = TR * GC / 8;      // 16-byte output pieces per tile
  static constexpr int SPL = CHUNKS / 128;        // ... per lane of the two store waves
  static constexpr int PD = KS <= 10 ? 3 : 6;     // A fragments in flight per compute wave (register budget: 256 per wave)
  static_assert(DPT % 2 == 0 && CHUNKS % 128 == 0, "tile must split evenly over the loader / store waves");
  static_assert(NR >= 3 && (NR - 2) * TPR * PER <= 63, "ring depth / vmcnt immediate");
  static constexpr int RES_DEPTH = KS == 20 ? WS_RES_DEPTH_640 : WS_RES_DEPTH;
  static_assert(RES_DEPTH % TPR == 0, "residual buffers are indexed statically per unrolled round");
};

// Sum over the 8 adjacent lanes that share a row (lane ^ 1, lane ^ 2, then the mirrored half-row = lane ^ 4 once the quads are uniform), on
// the DPP path: three VALU operations, every lane gets the total (the __shfl_xor form goes through ds_bpermute: two chains of three
// dependent ~100-cycle LDS exchanges per tile).
__device__ __forceinline__ float ws_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}

template <int CPR>
__device__ __forceinline__ int ws_swz(int row) {
  return CPR == 40 ? ((row >> 1) & 7) : (row & 15);
}

// GEGLU flavour: one 16-byte output piece = 8 columns of (h + b_h) * gelu_erf(g + b_g) from the fp32 staging tile.
struct WsGegluPiece {
  int row, hcol;            // staging row, staging column of h (g sits 32 columns further)
  float bh[8], bg[8];
};
__device__ __forceinline__ half8_t ws_geglu_piece(const float* cs, int cs_ld, const WsGegluPiece& q) {
  const float* s = cs + q.row * cs_ld + q.hcol;
  const floatx4 h0 = *reinterpret_cast<const floatx4*>(s), h1 = *reinterpret_cast<const floatx4*>(s + 4);
  const floatx4 g0 = *reinterpret_cast<const floatx4*>(s + 32), g1 = *reinterpret_cast<const floatx4*>(s + 36);
  half8_t o;
  float2_t gl[4];                         // pairs: the GELU polynomial runs on packed fp32, the four chains of a piece side by side
  // pairs of ADJACENT columns: the LDS reads deliver them in consecutive registers and v_cvt_pk_f16_f32 wants them so (pairs (j, j + 4), until round 6, cost
  // ten v_mov per piece; same arithmetic per element, same bits; no measurable difference: profiles/r06_ws_ablation.log)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    gl[j] = float2_t{g0[2 * j] + q.bg[2 * j], g0[2 * j + 1] + q.bg[2 * j + 1]};
    gl[j + 2] = float2_t{g1[2 * j] + q.bg[2 * j + 4], g1[2 * j + 1] + q.bg[2 * j + 5]};
  }
  gelu_fast2_x<4>(gl);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    o[2 * j] = (half_t)((h0[2 * j] + q.bh[2 * j]) * gl[j].x);
    o[2 * j + 1] = (half_t)((h0[2 * j + 1] + q.bh[2 * j + 1]) * gl[j].y);
    o[2 * j + 4] = (half_t)((h1[2 * j] + q.bh[2 * j + 4]) * gl[j + 2].x);
    o[2 * j + 5] = (half_t)((h1[2 * j + 1] + q.bh[2 * j + 5]) * gl[j + 2].y);
  }
  return o;
}

template <int KS, int CB, int TPR, bool RES, bool RA, bool GEGLU = false, int PRO = PRO_NONE>
__global__ __launch_bounds__(512, 1) void wsgemm_kernel(WsParams p) {
  constexpr bool LN = PRO == PRO_LNF;                             // the folded LayerNorm is applied in the epilogue (store waves)
  static_assert(PRO == PRO_NONE || !GEGLU, "GEGLU has no prologue flavour: its memory waves are VALU bound by the GELU");
  static_assert(!GEGLU || (CB == 4 && TPR == 1 && !RES && !RA), "GEGLU: 2 h + 2 g column blocks per compute wave, one tile per round, bias only");
  static_assert(PRO == PRO_NONE || !RES, "prologue flavours: no residual (their consumers have none)");

  static_assert(PRO != PRO_AFF || (!RA && !GEGLU), "PRO_AFF: bias only");
  using Cfg = WsCfg<KS, CB, TPR>;
  auto lslot_of = [](int r, int u) { return ((r & 3) * TPR + u) * Cfg::SSLOT; };      // statistics slot of (round r, tile u)
  constexpr int K = Cfg::K, CPR = Cfg::CPR, GC = Cfg::GC, TR = Cfg::TR, STAGE = Cfg::STAGE, RSTAGE = Cfg::RSTAGE, NR = Cfg::NR,
                CS_LD = Cfg::CS_LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + NR * RSTAGE);       // staging tile (round parity, tile u): index (r & 1) * TPR + u
  char* ost = smem + NR * RSTAGE + 2 * TPR * Cfg::CSTAGE;          // GEGLU: 2 x 2 KiB of finished fp16 pieces, loader -> store waves
  float2_t* lst = reinterpret_cast<float2_t*>(smem + NR * RSTAGE + 2 * TPR * Cfg::CSTAGE + Cfg::OSTAGE);   // PRO_LNF: (a, b) of round r, tile u, row: lslot_of(r, u) + row
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // workgroup -> (XCD, column group, row stream): the G groups of one row stream share an XCD (blockIdx % 8)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int grp = slot % p.groups, stream = xcd * p.spx + slot / p.groups;
  if (slot >= p.groups * p.spx) {
    // Leftover CUs (round 6).  G groups x spx streams fill G spx of the 32 CUs of an XCD: 30 for G = 5 (K = N = 640), 15 (N = 1920), 3 (N = 960)
    // and 10 (GEGLU N = 2560) -- two CUs per XCD, 16 on the chip, used to idle.  They are numbered XCD-major (a stream's groups stay on as few
    // XCDs as possible; their re-reads of the stream's A rows then cross XCDs, a few percent of one stream's bytes) and form xstreams more
    // row streams of G workgroups each.
    const int left = 32 - p.groups * p.spx, e = xcd * left + (slot - p.groups * p.spx);
    if (e >= p.xstreams * p.groups) return;
    grp = e % p.groups;
    stream = 8 * p.spx + e / p.groups;
  }
  const int ntiles = (p.M + TR - 1) / TR;
  // local tile t of this stream is tile tile0 + t * tstep: interleaved (stream, stream + S, ...), or one contiguous block per stream (PRO_AFF)
  const int tps = (ntiles + p.streams - 1) / p.streams;
  const int tile0 = PRO == PRO_AFF ? stream * tps : stream;
  const int tstep = PRO == PRO_AFF ? 1 : p.streams;
  const int my_tiles = PRO == PRO_AFF ? max(0, min(tps, ntiles - tile0)) : (stream < ntiles ? (ntiles - stream + p.streams - 1) / p.streams : 0);
  if (my_tiles == 0) return;
  const int rounds = (my_tiles + TPR - 1) / TPR;                  // barriers b_0 .. b_rounds
  const int n0 = grp * GC;
  // GEGLU (FeedForward net.0, reference src/models/attention.py:152-157 / diffusers FeedForward): the weight rows are packed in
  // blocks of 32 h rows then 32 g rows (packing.geglu_weight), so staging columns [64w, 64w+32) hold h and [64w+32, 64w+64) hold
  // g of output columns [32w, 32w+32) of this workgroup; out = (h + b_h) * gelu_erf(g + b_g), one rounding, 16-byte stores into
  // the [M][N/2] output.  The GELU arithmetic (two transcendentals + ~14 VALU per output) is the longest job of a tile, so all
  // FOUR memory waves (one per SIMD) share it: piece id = (wave - 4) * 64 + lane of the tile's 256 pieces.  The store waves write
  // theirs to global memory; the loader waves (whose vmcnt must see nothing but their DMAs) park theirs in LDS and the store
  // waves ship them one round later.  One extra barrier at the end drains that pipeline stage.  (Same-box, in the denoising
  // loop: 109 ms per clip against 118 ms with half of the GELU in the compute waves, although the two tie on random inputs.)
  WsGegluPiece gp = {};
  if constexpr (GEGLU) {
    if (wave >= 4) {
      const int id = (wave - 4) * 64 + lane;
      gp.row = id / (GC / 16);
      const int oc = (id % (GC / 16)) * 8;
      gp.hcol = 64 * (oc >> 5) + (oc & 31);
      {
        const half8_t b0 = p.bias ? *reinterpret_cast<const half8_t*>(p.bias + n0 + gp.hcol) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        const half8_t b1 = p.bias ? *reinterpret_cast<const half8_t*>(p.bias + n0 + gp.hcol + 32) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) gp.bh[j] = (float)b0[j], gp.bg[j] = (float)b1[j];
      }
    }
  }

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
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): weights are in registers before the loop
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
#if WS_ABL & 2
      auto frag = [&](int ks) { return wf[0][ks]; };
#else
      auto frag = [&](int ks) { return *reinterpret_cast<const half8_t*>(fb[ks % P] + (ks / P) * (16 << SWB)); };
#endif
#pragma unroll
      for (int j = 0; j < PD; ++j) af[j] = frag(j);
      __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);                         // PD ds_reads first ...
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8_t cur = af[ks % PD];
        if (ks + PD < KS) af[ks % PD] = frag(ks + PD);
#if WS_ABL & 1
        asm volatile("" ::"v"(cur));
#else
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cb][ks], cur, acc[cb], 0, 0, 0);
#endif
        if (ks + PD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // ... then one refill per k step, issued
        __builtin_amdgcn_sched_group_barrier(0x008, CB, 0);                       //     ahead of that step's CB MFMAs
      }
      // acc[cb][r] = C[m = row][n = wave*16CB + cb*16 + 4*kq + r]
      float* cs = cst + buf * (TR * CS_LD) + row * CS_LD + wave * 16 * CB + 4 * kq;
#if WS_ABL & 32
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) asm volatile("" ::"v"(acc[cb]));
      (void)cs;
#else
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<floatx4*>(cs + cb * 16) = acc[cb];
#endif
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
    if constexpr (GEGLU) __builtin_amdgcn_s_barrier();   // b_{rounds + 1}
  } else if (wave < 6) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    const half_t* sp[Cfg::PER];
    const size_t astep = (size_t)tstep * TR * p.lda;
#pragma unroll
    for (int i = 0; i < Cfg::PER; ++i) {
      const int pidx = (lw * Cfg::PER + i) * 64 + lane;
      const int r = pidx / CPR, c = pidx % CPR;
      sp[i] = p.A + (size_t)(tile0 * TR + r) * p.lda + ((c ^ ws_swz<CPR>(r)) << 3);
    }
    // Prologues.  Loader wave lw DMA'd rows 8 lw .. 8 lw + 7 of every tile itself (PER * 64 slots = 8 rows), so its own counted vmcnt
    // wait is all it needs before touching them.  Lane (prow, psub): row prow, 16-byte chunks 8 j + psub, j < PJ (the swizzle only
    // permutes chunks inside a group of 8 / 16, so chunk c of the row sits in LDS slot c ^ swz(row)).
    static_assert(Cfg::PER * 64 == 8 * CPR, "a loader wave owns whole rows");
    constexpr int PJ = CPR / 8;
    const int prow = lw * 8 + (lane >> 3), psub = lane & 7;
    const int pswz = ws_swz<CPR>(prow);
    auto ln_stats = [&](const char* st, int lslot) {            // exact two-pass (mu, rstd) of the landed row -> (a, b) = (rstd, -mu rstd)
      half8_t h[PJ];
#pragma unroll
      for (int j = 0; j < PJ; ++j) h[j] = *reinterpret_cast<const half8_t*>(st + (prow * CPR + ((8 * j + psub) ^ pswz)) * 16);
      float sum = 0.f;                                          // v_dot2_f32_f16 against (1, 1): two elements per instruction, fp32 accumulate
#pragma unroll
      for (int j = 0; j < PJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; e += 2) sum = __builtin_amdgcn_fdot2(half2_t{h[j][e], h[j][e + 1]}, half2_t{(half_t)1.0f, (half_t)1.0f}, sum, false);
      sum = ws_sum8(sum);
      const float mu = sum * (1.0f / K);
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < PJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)h[j][e] - mu;
          sq += d * d;
        }
      sq = ws_sum8(sq);
      const float a = rsqrtf(sq * (1.0f / K) + p.eps);
      if (psub == 0) lst[lslot + prow] = float2_t{a, -mu * a};
    };
    float asc[PRO == PRO_AFF ? PJ : 1][8], asf[PRO == PRO_AFF ? PJ : 1][8];
    int cur_img = -1;
    auto load_table = [&](int img) {                            // scale / shift of this lane's columns for image `img`; drains vmcnt
      const float* t = p.aff + (size_t)img * 2 * K;
#pragma unroll
      for (int j = 0; j < (PRO == PRO_AFF ? PJ : 1); ++j) {
        const floatx4 s0 = *reinterpret_cast<const floatx4*>(t + (8 * j + psub) * 8), s1 = *reinterpret_cast<const floatx4*>(t + (8 * j + psub) * 8 + 4);
        const floatx4 f0 = *reinterpret_cast<const floatx4*>(t + K + (8 * j + psub) * 8), f1 = *reinterpret_cast<const floatx4*>(t + K + (8 * j + psub) * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) asc[j][e] = s0[e], asc[j][e + 4] = s1[e], asf[j][e] = f0[e], asf[j][e + 4] = f1[e];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): everything landed (the DMAs in flight too); the counted waits
      cur_img = img;                                            // below stay valid -- fewer operations are outstanding than they assume
    };
    auto aff_apply = [&](char* st) {                            // x * scale + shift, one rounding: gn_apply_kernel's arithmetic
#pragma unroll
      for (int j = 0; j < (PRO == PRO_AFF ? PJ : 1); ++j) {
        char* a = st + (prow * CPR + ((8 * j + psub) ^ pswz)) * 16;
        const half8_t h = *reinterpret_cast<const half8_t*>(a);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)__builtin_fmaf((float)h[e], asc[j][e], asf[j][e]);
        *reinterpret_cast<half8_t*>(a) = o;
      }
    };
    if constexpr (PRO == PRO_AFF) load_table((tile0 * TR) / p.rows_per_image);      // before the first DMA is issued
    auto issue_round = [&](int q) {                 // called with q = 0, 1, 2, ... in order; tiles are issued in order too
      char* st = ring + (q % NR) * RSTAGE;
#if WS_ABL & 16
      if (q >= NR - 1) return;
#endif
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
    if constexpr (GEGLU) __builtin_amdgcn_s_waitcnt(0x0F70);      // the bias loads above: nothing but DMAs may be counted below
    const int pre = rounds < NR - 1 ? rounds : NR - 1;
    for (int q = 0; q < pre; ++q) issue_round(q);
    for (int r = 0; r < rounds; ++r) {
      // round r must have landed before the barrier.  Issued so far: rounds <= r+NR-2; all of r+1 .. r+NR-2 are FULL rounds
      // (TPR tiles, TPR*PER pieces each) as long as none of them is the last one -- otherwise simply drain.
      if (r + NR - 2 < rounds - 1) wait_vmcnt<(NR - 2) * TPR * Cfg::PER>();
      else wait_vmcnt<0>();
      if constexpr (PRO != PRO_NONE) {              // round r has landed (this wave's rows): normalise / take the statistics before the hand-over
        char* st = ring + (r % NR) * RSTAGE;
#pragma unroll
        for (int u = 0; u < TPR; ++u) {
          const int t = r * TPR + u;
          if (t < my_tiles) {
            if constexpr (PRO == PRO_LNF) ln_stats(st + u * STAGE, lslot_of(r, u));
            if constexpr (PRO == PRO_AFF) {
              const int img = ((tile0 + t * tstep) * TR) / p.rows_per_image;
              if (img != cur_img) load_table(img);
              aff_apply(st + u * STAGE);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                 // also: the compute waves are done with round r-1 -> its stage is free
      if (r + NR - 1 < rounds) issue_round(r + NR - 1);   // into stage (r - 1) % NR
      if constexpr (GEGLU) {
        if (r >= 1) {                               // this wave's piece of tile r-1 -> LDS (shipped by a store wave after b_{r+1})
          const half8_t o = ws_geglu_piece(cst + ((r - 1) & 1) * (TR * CS_LD), CS_LD, gp);
          *reinterpret_cast<half8_t*>(ost + ((r - 1) & 1) * 2048 + (lw * 64 + lane) * 16) = o;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
    __builtin_amdgcn_s_barrier();                   // b_rounds
    if constexpr (GEGLU) {
      const half8_t o = ws_geglu_piece(cst + ((rounds - 1) & 1) * (TR * CS_LD), CS_LD, gp);
      *reinterpret_cast<half8_t*>(ost + ((rounds - 1) & 1) * 2048 + (lw * 64 + lane) * 16) = o;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // b_{rounds + 1}
    }
  } else {
    // ------------------------------------------------------------------------------------------------ store waves
    if constexpr (GEGLU) {
      constexpr int OC = GC / 2;
      const int sw2 = wave - 6;                                          // 0 / 1: ships the pieces of loader wave sw2 as well
      const int id_own = (wave - 4) * 64 + lane, id_ld = sw2 * 64 + lane;
      const size_t cstep = (size_t)tstep * TR * p.ldc;
      half_t* cp_own = p.C + (size_t)(tile0 * TR + id_own / (OC / 8)) * p.ldc + grp * OC + (id_own % (OC / 8)) * 8;
      half_t* cp_ld = p.C + (size_t)(tile0 * TR + id_ld / (OC / 8)) * p.ldc + grp * OC + (id_ld % (OC / 8)) * 8;
      for (int r = 0; r <= rounds + 1; ++r) {
        __builtin_amdgcn_s_barrier();                                   // b_r
        if (r >= 2) {                                                   // loader pieces of tile r-2, parked in LDS during round r-1
          *reinterpret_cast<half8_t*>(cp_ld) = *reinterpret_cast<const half8_t*>(ost + ((r - 2) & 1) * 2048 + id_ld * 16);
          cp_ld += cstep;
        }
        if (r >= 1 && r <= rounds) {                                    // own piece of tile r-1
          *reinterpret_cast<half8_t*>(cp_own) = ws_geglu_piece(cst + ((r - 1) & 1) * (TR * CS_LD), CS_LD, gp);
          cp_own += cstep;
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
      const size_t cstep = (size_t)tstep * TR * p.ldc, rstep = (size_t)tstep * TR * p.ldr;
      float lsf[LN ? SPL : 1][8], lcf[LN ? SPL : 1][8];      // folded LayerNorm: s[n], c[n] of this lane's columns
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        const int id = i * 128 + sid;
        prow[i] = id / CPRO;
        pcol[i] = (id % CPRO) * 8;
        const half8_t bv = p.bias ? *reinterpret_cast<const half8_t*>(p.bias + n0 + pcol[i]) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) biasf[i][j] = (float)bv[j];
        cp[i] = p.C + (size_t)(tile0 * TR + prow[i]) * p.ldc + n0 + pcol[i];
        rp[i] = RES ? p.residual + (size_t)(tile0 * TR + prow[i]) * p.ldr + n0 + pcol[i] : nullptr;
        if constexpr (LN) {
          const floatx4 s0 = *reinterpret_cast<const floatx4*>(p.lnf + n0 + pcol[i]), s1 = *reinterpret_cast<const floatx4*>(p.lnf + n0 + pcol[i] + 4);
          const floatx4 c0 = *reinterpret_cast<const floatx4*>(p.lnf + p.N + n0 + pcol[i]), c1 = *reinterpret_cast<const floatx4*>(p.lnf + p.N + n0 + pcol[i] + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) lsf[i][j] = s0[j], lsf[i][j + 4] = s1[j], lcf[i][j] = c0[j], lcf[i][j + 4] = c1[j];
        }
      }
#ifndef WS_PRE_WAIT
#define WS_PRE_WAIT 1
#endif
      // The tables above are loop invariants requested ONCE, and the wait-count pass must know that they have landed before the tile loop:
      // left to itself it covers their first use INSIDE the loop (flavours whose loop body starts with the row-term branch get no peeled
      // first iteration), with the counts that are right on the entry path -- vmcnt(16) .. vmcnt(4) in front of the pieces of EVERY tile of
      // the LayerNorm + row-term flavour (the motion module's q|k|v) -- and on every later tile those counts wait for the tile's own
      // stores to be acknowledged: one store round trip per tile (profiles/r06_ab_ws_preamble_wait.log).
      if constexpr (WS_PRE_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);
      constexpr int RD = RA ? (KS == 20 ? 4 : 2) : Cfg::RES_DEPTH;   // residual tiles in flight per store wave (fewer when the row-broadcast term
                                                                // also lives in registers at K = 320: 5 pieces per lane and tile)
      static_assert(RD % TPR == 0, "residual buffers are indexed statically per unrolled round");
      constexpr int UNR = RD / TPR;                   // rounds per unrolled loop body
      half8_t res[RES ? RD : 1][SPL];
      int cur_group = -1;
      float raf[RA ? SPL : 1][8];
      auto store_tile = [&](int tile, int buf, int lslot, const half8_t (&rs)[SPL]) {
        const int m0 = (tile0 + tile * tstep) * TR;
        const float* cs = cst + buf * (TR * CS_LD);
#if WS_ABL & 8
        return;
#endif
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
            // the reload's latency is paid HERE, once per group change, with the wait the compiler's pass understands: left to the pass, the
            // wait lands behind the merge of the two paths, i.e. as vmcnt(0) in front of EVERY tile's epilogue (residual fetches included)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            cur_group = g1 == g0 ? g0 : -1;
          }
        }
        // all LDS reads first (one latency per tile, not one per piece), then the arithmetic
        floatx4 ca[SPL], cb2[SPL];
        float2_t ab[LN ? SPL : 1];
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          ca[i] = *reinterpret_cast<const floatx4*>(cs + prow[i] * CS_LD + pcol[i]);
          cb2[i] = *reinterpret_cast<const floatx4*>(cs + prow[i] * CS_LD + pcol[i] + 4);
          if constexpr (LN) ab[i] = lst[lslot + prow[i]];
        }
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          float v[8] = {ca[i][0], ca[i][1], ca[i][2], ca[i][3], cb2[i][0], cb2[i][1], cb2[i][2], cb2[i][3]};
          if constexpr (LN) {                       // LN(x) . W^T + bias = rstd (x . W'^T - mu s) + c
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ab[i].x * v[j] + (ab[i].y * lsf[i][j] + lcf[i][j]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += biasf[i][j];
          }
          if constexpr (RA) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += raf[i][j];
          }
          if constexpr (RES) {
#if defined(WS_ABL_RES) && WS_ABL_RES == 1              // diagnostic build: the loads without the adds
            asm volatile("" ::"v"(rs[i]));
#else
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rs[i][j];
#endif
          }
          half8_t o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
#if WS_ABL & 4
          asm volatile("" ::"v"(o));
#else
          *reinterpret_cast<half8_t*>(cp[i]) = o;
#endif
          cp[i] += cstep;
        }
      };
      // K = 640 only.  At K = 320 the counted loop is +4.5 % on the isolated launch (0.120 -> 0.115 ms) and +-0 inside the denoising loop, where those
      // launches move A + C + R at the HBM rate either way (profiles/r06_ab_ws_store_loop.log); with two tiles per round it would also spill (256 VGPRs).
      constexpr bool COUNTED = RES && KS == 20;
      if constexpr (COUNTED) {
        // Residual flavours (restructured in round 6).  Until then this loop fetched and stored under per-tile conditions (`tile < my_tiles`,
        // `tile + RD < my_tiles`), and the compiler's wait-count pass, which has to be right on every path through such a loop, answered with
        // `s_waitcnt vmcnt(1)` / `vmcnt(0)` in front of every tile's residual: the wave waited for ALL its outstanding operations -- the stores
        // it had just issued and the residual tile it had requested for RD tiles later -- i.e. it exposed one HBM latency per tile, whatever the
        // prefetch depth (profiles/r05_ab_ws_residual_stream.log: depth 4 -> 8 -> 16 changed nothing; the K = 640 flavour paid 0.022 ms per launch
        // for its residual).  Now the steady state is free of conditions on memory operations: the fetch is unconditional (past the last tile it
        // re-reads the last one: the pointer stops advancing), full rounds run in an unrolled sequence that is LEFT by a jump (never skipped
        // into), and a partial last round is handled at the exit it belongs to, with its own static slot.  The pass then counts: in front of a
        // tile's residual it leaves the (RD - 1) SPL younger fetches (plus the stores between them) in flight.
        int nf = 0;                                   // tiles fetched so far (the next fetch is tile min(nf, my_tiles - 1))
        auto fetch_next = [&](half8_t (&dst)[SPL]) {
          const size_t step = nf + 1 < my_tiles ? rstep : 0;
#pragma unroll
          for (int i = 0; i < SPL; ++i) {
            dst[i] = *reinterpret_cast<const half8_t*>(rp[i]);
            rp[i] += step;
          }
          ++nf;
        };
#pragma unroll
        for (int j = 0; j < RD; ++j) fetch_next(res[j]);                // tiles 0 .. RD-1
        const int full = my_tiles / TPR;                                // rounds whose TPR tiles all exist
        __builtin_amdgcn_s_barrier();                                   // b_0
        int r = 1;                                                      // barrier b_r is followed by the stores of round r - 1
        // body J of the unrolled sequence (J spelled out: the slots must be compile-time indices whatever the loop transformations do --
        // with `#pragma unroll` over a loop that is left by goto the residual buffers ended up dynamically indexed, in scratch memory)
#define WS_STORE_BODY(J)                                                                                    \
        if constexpr ((J) < UNR) {                                                                          \
          if (r > full) {                                                                                   \
            if (r <= rounds) {       /* the partial last round (TPR = 2, odd tile count): its first tile only */ \
              __builtin_amdgcn_s_barrier();                             /* b_rounds */                      \
              store_tile((r - 1) * TPR, ((r - 1) & 1) * TPR, lslot_of(r - 1, 0), res[(J) * TPR]);           \
            }                                                                                               \
            goto ws_store_done;                                                                             \
          }                                                                                                 \
          __builtin_amdgcn_s_barrier();                                 /* b_r */                           \
          _Pragma("unroll") for (int u = 0; u < TPR; ++u) {                                                 \
            /* slot == tile % RD: r - 1 == J (mod UNR) */                                                   \
            store_tile((r - 1) * TPR + u, ((r - 1) & 1) * TPR + u, lslot_of(r - 1, u), res[(J) * TPR + u]); \
            fetch_next(res[(J) * TPR + u]);                                                                 \
          }                                                                                                 \
          ++r;                                                                                              \
        }
        static_assert(UNR <= 8, "WS_STORE_BODY is spelled out eight times");
        for (;;) {
          WS_STORE_BODY(0) WS_STORE_BODY(1) WS_STORE_BODY(2) WS_STORE_BODY(3) WS_STORE_BODY(4) WS_STORE_BODY(5) WS_STORE_BODY(6) WS_STORE_BODY(7)
        }
#undef WS_STORE_BODY
      ws_store_done:;
      } else {
        // barrier b_r (r = 0 .. rounds): afterwards the compute waves work on round r and this wave stores round r-1, then (K = 320 residual
        // flavour with two tiles per round) prefetches the residual of the tile RD places further into the buffer it has just freed.  Unrolled
        // by UNR rounds so that the residual buffers are addressed statically: tile t uses res[t % RD].
        auto fetch_res = [&](half8_t (&dst)[SPL]) {
#pragma unroll
          for (int i = 0; i < SPL; ++i) {
            dst[i] = *reinterpret_cast<const half8_t*>(rp[i]);
            rp[i] += rstep;
          }
        };
        if constexpr (RES) {
#pragma unroll
          for (int j = 0; j < RD; ++j)
            if (j < my_tiles) fetch_res(res[j]);        // tiles 0 .. RD-1
        }
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
                    store_tile(tile, ((r - 1) & 1) * TPR + u, lslot_of(r - 1, u), res[RES ? slot_ : 0]);
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
}
