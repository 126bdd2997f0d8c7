// gemm_sp_kernel: the "one wave per SIMD" flavour of the MFMA GEMM / implicit-GEMM 3x3 convolution (included by gemm.hip; same
// operands, swizzle idea and epilogue arithmetic as gemm_kernel).
//
// Why this structure.  The "ping-pong" kernel of rounds 1-2 (gemm_pp.h, removed in round 3 after this one beat it on every shape)
// kept two waves per SIMD and alternated them between a fragment-load slot and an MFMA
// slot with a workgroup barrier in between; its load slot (14 ds_read_b128 + 4-5 DMA pieces, ~830 cycles) is longer than its
// MFMA slot (20 MFMAs, ~700 cycles), so the matrix pipe of a SIMD is busy ~64 % of the time at best, and with 256 registers
// per wave the wave tile is 64 x 160 (0.7 fragment reads per MFMA).  Here a 256-thread workgroup owns the CU with ONE wave per
// SIMD and the whole 512-register file per lane:
//   * wave tile (32 MT) x (32 NT), wave grid 2 x 2: MT = 3, NT = 5 -> 192 x 320 tiles (240 accumulator registers, 8 fragment
//     reads per 15 MFMAs; 192 divides every token count of the UNets and gives 294 912 x 320 six and 73 728 x 640 three exact
//     rounds of 256 CUs); MT = NT = 4 -> 256 x 256 for GEGLU (h | g column pairing needs 64-column groups per wave);
//   * the wave is software pipelined against ITSELF: the fragments of k-step u+1 are read into the other register set and
//     the DMA pieces of the K tiles ahead are issued in the issue slots BETWEEN the MFMAs of step u (sched_barrier pins one
//     ds_read_b128 behind each of the first MT + NT MFMAs and the step's DMA pieces behind evenly spaced MFMAs; giving each wave
//     its own gaps for the pieces was measured 12-15 % slower: a scalar branch per gap costs more than the shared address path);
//   * K tiles of 64: a tile row is ONE 128-byte cache line, so a DMA piece (64 lanes x 16 bytes) fetches 8 whole lines.  With
//     K tiles of 32 (64-byte rows) every line was requested twice, a K tile apart, by 16-row pieces; the ablations
//     (profiles/r03_ab_gemm_sp_dma_lines.log) price the DMA traffic of the main loop at 19-33 % of the time and show that pieces
//     of whole lines give back 8-20 % of it (the vector address arithmetic of the first build, by contrast, cost nothing once
//     the pieces went through buffer descriptors: what mattered was the number of L2 requests);
//   * LDS ring of mixed depth, 152 / 160 KiB: THREE slots for the A tile (the streamed operand: an A piece is issued two K tiles
//     = ~3 800 cycles before it is needed, enough for an HBM miss), TWO for the W tile (shared by every CU of the column panel,
//     an L2 hit: issued 2-3 k-steps = >= 960 cycles ahead), counted vmcnt, ONE s_barrier per K tile (per 60-64 MFMAs of a wave);
//   * persistent: a workgroup walks over its output tiles (grouped tile order, see tile_origin) and the ring keeps running across
//     tile boundaries; the first k-step of an output tile accumulates onto the inline constant 0 (no accumulator clearing);
//   * epilogue straight from the accumulators (operand roles swapped, acc = mfma(W, A): a lane owns ONE output row and four
//     4-column pieces per 32-column sub-tile): v_permlane32_swap pairs the pieces of the two lane halves into 16-byte stores
//     (and un-pairs 16-byte residual loads), no LDS staging, no barrier.
// Synchronisation.  K tile t lives in A slot t % 3 and W slot t & 1; its four k-steps s = 0..3 use the fragment sets F0 / F1
// alternately; the fragments of step (t, s+1) -- of (t+1, 0) for s = 3 -- are read during step (t, s).  Once per K tile, at the
// START of step (t, 3):  lgkmcnt(0) (every read of tile t is retired), vmcnt(PA) (this wave's pieces of W(t+1) and A(t+1) have
// landed; only its PA pieces of A(t+2) may still fly), s_barrier.
//   RAW: the first reads of tile t+1 follow that barrier.
//   WAR: behind the barrier the slots of tile t are refilled: the 16 pieces [W(t+2) | A(t+3)] of a wave go out in this order
//        over the steps (t, 3), (t+1, 0), (t+1, 1), (t+1, 2) as 6 + 4 + 3 + 3, the W pieces first (they are needed one tile earlier).
//   vmcnt counts stores too, but loads return in order: "at most PA outstanding" implies that every load older than the PA
//   youngest loads has landed whatever the epilogue's stores do (at worst it waits for old stores as well).
// DMA pieces are issued unconditionally (past the last K tile they re-read the last tile's first columns into ring slots nobody
// reads again), so every count is a compile-time constant; the kernel drains them before it ends.
#pragma once
#ifndef SP_ABL
#define SP_ABL 0   // diagnostic builds only (tools/build_ab.sh): 1 no barrier, 2 no DMA in the loop, 4 no fragment reads in the loop, 8 no vmcnt wait,
                   // 16 every DMA piece re-reads the first K tile of the first rows (cache hits: issue cost without the memory system behind it)
#endif

#ifdef SP_TRACE
// diagnostic build only (tools/build_ab.sh trace -DSP_TRACE, tools/sp_trace.py): shader-clock stamps of workgroup 0, every wave, its
// first SP_TRACE_N K tiles; kept in the LDS left over by the ring and copied out when the kernel ends
#define SP_TRACE_N 48
__device__ unsigned long long g_sp_trace[4][SP_TRACE_N][5];
#define SP_STAMP(S)                                                                                           \
  if (!GEGLU && blockIdx.x == 0 && tr_kt < SP_TRACE_N && lane == 0)                                           \
    *reinterpret_cast<unsigned long long*>(smem + (3 * BM + 2 * BN) * 128 + ((wave * SP_TRACE_N + tr_kt) * 5 + (S)) * 8) = __builtin_readcyclecounter();
#else
#define SP_STAMP(S)
#endif

typedef unsigned uint4_t __attribute__((ext_vector_type(4)));      // what the raw buffer load returns

// zeros standing in for an absent bias / row-broadcast operand (N <= 16384 columns: checked by sp_eligible)
__device__ __attribute__((aligned(64))) half_t g_zero_cols[16384] = {};
// where the 16-byte pieces of rows beyond M go (one slot per lane; only the last row tile of a ragged M has such rows).  The epilogues
// store UNCONDITIONALLY, to a selected address: a store under `if (row < M)` is a memory operation the compiler's wait-count pass cannot
// count on, so in front of the next column's bias / residual / row-term registers it planted waits that let only the younger LOADS fly --
// i.e. every column waited until the previous column's stores had been ACKNOWLEDGED: 5 x ~1.9 k cycles, the "~10 k cycles" of a plain
// 192 x 320 epilogue (profiles/r06_ab_sp_epilogue_waits.log).
__device__ __attribute__((aligned(64))) half_t g_sp_dump[64 * 8];

// One accumulator element, AGPR -> VGPR, at the point of use.  The "a" constraint keeps the MFMA accumulators in the accumulator
// half of the register file for the whole kernel (left to itself the register allocator copies all 240-256 of them into
// VGPRs at the top of the epilogue, which evicts the main loop's fragments and pointers into scratch).
__device__ __forceinline__ float sp_acc(const floatx16& a, int r) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a[r]));
  return x;
}

// Plain epilogue of one wave's (32 MT) x (32 NT) accumulator block, straight from the registers.  acc = mfma(W, A): lane
// (lc = lane % 32, hi = lane / 32) holds, for output row mw + 32 i + lc, the columns 8 g + 4 hi + {0..3}, g = 0..3 of every
// 32-column sub-tile j.  v_permlane32_swap pairs the 8-byte pieces of the two lane halves into 16-byte stores (guide T21) and
// un-pairs 16-byte residual loads (same instruction: it is an involution).  act == ACT_NONE (launcher).  AGPR: read the accumulators
// with sp_acc (gemm_sp_kernel: 240+ accumulators pinned in the accumulator file); false: plain reads.
//
// Order: sub-tile COLUMN by column (j outer, the MT rows inner).  The bias of a column is converted once for its MT sub-tiles,
// and everything the column needs from memory (bias, residual, row-broadcast term) is requested ONE COLUMN AHEAD: with one wave
// per SIMD nobody else covers a load's latency, and the first form of this epilogue (loads at the point of use, one sub-tile at a
// time) exposed it 15 times per tile -- ~5 of the ~7.6 us a 192 x 320 tile spent outside its K loop.  RES / RA: residual /
// row-broadcast operand present (absent operands cost nothing); with both, the row-broadcast values are requested at the top of
// their own column instead (the registers for a second look-ahead set are not there).
template <int MT, int NT, bool RES, bool RA>
struct SpColumn {
  half4_t b[4];                          // bias, columns 8 g + 4 hi + {0..3}
  uint4 r[RES ? MT : 1][2];              // residual, 16-byte pieces (before the un-pairing swap)
  half4_t a[RA ? MT : 1][4];             // row-broadcast term
};

// Residual look-ahead (round 6): how many columns ahead of the one being written the residual is requested.  ONE, as since round 3 --
// measured, not assumed.  tools/sp_trace.py prices the epilogue of a 192 x 320 tile at ~10 k cycles without and 21-33 k cycles WITH a
// residual (N = K = 1280: 21 k against a K loop of 51 k; FF-out 27-31 k against 55 k; N = K = 640 31-35 k against 25 k:
// profiles/r06_sp_trace_gemm_tail.log).  Ablation builds (profiles/r06_sp_epilogue_ablation.log): the stores alone cost ~10 k, the residual
// loads alone 12-15 k, together their SUM -- a wave's vector-memory operations complete in order on one counter, so the next column's
// residual loads return only after the previous column's stores have been acknowledged, and under the lock-step burst of 256 CUs storing
// at once that takes thousands of cycles.  Hence neither two / three columns of look-ahead (-DSP_EPI_LA=2 / 3: 22.4 -> 21.0 k in the
// trace, 7.345 / 7.354 vs 7.352 / 7.350 frames/s: profiles/r06_ab_sp_residual_lookahead.log) nor pulling the residual into L2 with
// extra DMA pieces at the end of the K loop (-0.3 % end to end: profiles/r06_ab_sp_residual_prefetch.log) changes anything.  What would
// (all residual loads of a tile ahead of its first store; the residual injected through the matrix core as extra K tiles) is priced in
// profiles/HISTORY.md, not built.
#ifndef SP_EPI_LA
#define SP_EPI_LA 1            // A/B builds: -DSP_EPI_LA=n
#endif
template <int MT, int NT, bool CONV>
constexpr int sp_residual_lookahead() {
  constexpr int la = SP_EPI_LA < NT - 1 ? SP_EPI_LA : NT - 1;
  return CONV && MT * NT >= 15 && la > 1 ? 1 : la;      // 192 x 320 conv: the tap offsets leave no room for a third column set
}

template <int MT, int NT, bool RES, bool RA, bool AGPR = true, bool RB = false, bool CONV = false>
__device__ __forceinline__ void sp_plain_epilogue(const GemmParams& p, const floatx16 (&acc)[MT][NT], int mw, int nw, int lc, int hi) {
  // RB: the bias belongs to the output ROW (the launcher swapped the operands to produce a transposed output: V^T for attention)
  constexpr bool RA_AHEAD = RA && !RES;
  constexpr int LA = RES ? sp_residual_lookahead<MT, NT, CONV>() : 1;      // columns requested ahead of the one being written
  const half_t* bias = (p.bias && !RB ? p.bias : g_zero_cols) + (RB ? 0 : nw + 4 * hi);
  float rowb[MT];
  bool row_ok[MT];
  const half_t* rrow[MT];
  const half_t* arow[MT];
  half_t* crow[MT][2];                   // store pointers: rows (lc & 15) and 16 + (lc & 15) of sub-tile row i, this lane's 16-byte chunk
  bool crow_ok[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mw + i * 32 + lc;
    row_ok[i] = m < p.M;
    const int mc = row_ok[i] ? m : p.M - 1;
    rowb[i] = RB && p.bias ? (float)p.bias[mc] : 0.f;
    rrow[i] = RES ? p.residual + (size_t)mc * p.ldr + nw + 8 * hi : nullptr;
    arow[i] = RA ? p.rowadd + (size_t)(mc / p.rows_per_group) * p.ldra + nw + 4 * hi : nullptr;
    // Stores (round 6): 16 rows x 64 bytes per instruction instead of 32 rows x 32 bytes.  After the v_permlane32_swap pairing a lane holds
    // the two 16-byte pieces (pr = 0, 1) of ITS row; one v_permlane16_swap per dword then gathers both pieces of rows 0-15 in one register
    // set (lane l: row l & 15, piece (l >> 4) & 1, half hi) and those of rows 16-31 in the other: a row's four chunks sit in four lanes,
    // the memory pipeline sees 64-byte runs, one request per row instead of two.  Measured (profiles/r06_ab_sp_epilogue_waits.log): the epilogue
    // of a 192 x 320 tile is ~5.0 k cycles of arithmetic + ~4.5 k that appear with the stores; 64-byte runs take 0.1-1.3 k of the latter
    // (+0.2 % end to end), non-temporal stores nothing (-0.35 % end to end: the consumer misses the memory-side cache).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int mr = mw + i * 32 + 16 * h + (lc & 15);
      crow_ok[i][h] = mr < p.M;
      crow[i][h] = p.C + (size_t)(crow_ok[i][h] ? mr : p.M - 1) * p.ldc + nw + 16 * ((lc >> 4) & 1) + 8 * hi;
    }
  }
  half_t* const dump = g_sp_dump + ((lc + 32 * hi) << 3);
  SpColumn<MT, NT, RES, RA> col[LA + 1];
  auto request = [&](SpColumn<MT, NT, RES, RA>& d, int j, bool ahead) {
    if constexpr (!RB) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) d.b[gi] = *reinterpret_cast<const half4_t*>(bias + j * 32 + 8 * gi);
    }
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) d.r[i][pr] = *reinterpret_cast<const uint4*>(rrow[i] + j * 32 + 16 * pr);
    }
    if constexpr (RA) {
      if (ahead == RA_AHEAD) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) d.a[i][gi] = *reinterpret_cast<const half4_t*>(arow[i] + j * 32 + 8 * gi);
      }
    }
  };
#pragma unroll
  for (int j = 0; j < LA && j < NT; ++j) request(col[j % (LA + 1)], j, true);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    SpColumn<MT, NT, RES, RA>& c = col[j % (LA + 1)];
    if (j + LA < NT) request(col[(j + LA) % (LA + 1)], j + LA, true);
    if constexpr (RA && !RA_AHEAD) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) c.a[i][gi] = *reinterpret_cast<const half4_t*>(arow[i] + j * 32 + 8 * gi);
    }
    __builtin_amdgcn_sched_barrier(0);
    float bf[4][4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int e = 0; e < 4; ++e) bf[gi][e] = RB ? 0.f : (float)c.b[gi][e];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      unsigned rp[4][2];
      if constexpr (RES) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const uint4 r4 = c.r[i][pr];
          const auto s0 = __builtin_amdgcn_permlane32_swap(r4.x, r4.z, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(r4.y, r4.w, false, false);
          rp[2 * pr][0] = s0[0]; rp[2 * pr][1] = s1[0];
          rp[2 * pr + 1][0] = s0[1]; rp[2 * pr + 1][1] = s1[1];
        }
      }
      unsigned w[4][2];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (AGPR ? sp_acc(acc[i][j], 4 * gi + e) : acc[i][j][4 * gi + e]) + (RB ? rowb[i] : bf[gi][e]);
          if constexpr (RA) v[e] += (float)c.a[i][gi][e];
        }
        if constexpr (RES) {
          half4_t rv;
          __builtin_memcpy(&rv, rp[gi], 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        __builtin_memcpy(w[gi], &o, 8);
      }
      unsigned pc[2][4];                                             // piece pr of this lane's row: 8 columns
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(w[2 * pr][0], w[2 * pr + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(w[2 * pr][1], w[2 * pr + 1][1], false, false);
        pc[pr][0] = s0[0]; pc[pr][1] = s1[0]; pc[pr][2] = s0[1]; pc[pr][3] = s1[1];
      }
      unsigned lo[4], hi4[4];                                        // rows 0-15 / rows 16-31 of the sub-tile, both pieces
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const auto t = __builtin_amdgcn_permlane16_swap(pc[0][d], pc[1][d], false, false);
        lo[d] = t[0]; hi4[d] = t[1];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint4 v4 = h ? uint4{hi4[0], hi4[1], hi4[2], hi4[3]} : uint4{lo[0], lo[1], lo[2], lo[3]};
        if (SP_ABL & 64) asm volatile("" ::"v"(v4.x), "v"(v4.y), "v"(v4.z), "v"(v4.w));      // diagnostic: the arithmetic without the stores
        else *reinterpret_cast<uint4*>(crow_ok[i][h] ? crow[i][h] + j * 32 : dump) = v4;
      }
      __builtin_amdgcn_sched_barrier(0);                           // one sub-tile at a time: keeps the live ranges short
    }
  }
}

template <bool CONV, bool GEGLU, int MT, int NT, bool RESM = false>
__global__ __launch_bounds__(256, 1) void gemm_sp_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the stub: it cannot lower the buffer-descriptor type used below
  constexpr int BK = 64;
  constexpr int BM = 64 * MT, BN = 64 * NT;
  constexpr int ROWB = BK * 2, RPI = 1024 / ROWB;        // 128-byte rows, 8 rows per DMA piece
  constexpr int PA = BM / RPI / 4, PB = BN / RPI / 4;   // DMA pieces per wave per K tile: A rows (6 / 8), W rows (10 / 8)
  // issue schedule of a wave's PA + PB pieces per K tile over the four k-steps that follow the tile's barrier, W pieces first:
  // 6 + 4 + 3 + 3 (16 pieces), 6 + 4 + 2 + 2 (14), 4 + 4 + 2 + 2 (12: the 128 x 256 tile has 8 MFMAs per k-step to hide them behind)
  // 3 + 3 + 2 + 2 (10: the 192 x 128 tile has 6 MFMAs per k-step, a piece behind every other one at most)
  static_assert(PA + PB == 16 || PA + PB == 14 || PA + PB == 12 || PA + PB == 10, "issue schedule");
  constexpr int NP0 = PA + PB == 10 ? 3 : (PA + PB == 12 ? 4 : 6), NP1 = PA + PB == 10 ? 3 : 4, NP2 = (PA + PB - NP0 - NP1) / 2, NP3 = NP2;
  static_assert(NP0 + NP1 + NP2 + NP3 == PA + PB && PB >= NP0, "issue schedule");
  static_assert(!GEGLU || NT % 2 == 0, "GEGLU pairs 32-column sub-tiles (2q, 2q+1) of a wave");
  constexpr int ASZ = BM * ROWB, WSZ = BN * ROWB, WBASE = 3 * ASZ;        // ring: A slots 0..2, then W slots 0..1
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane >> 3, pslot = lane & 7;
  const int nk = p.K / BK;
  const int nwg = p.tiles_total;
  const int ntile = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // output tiles of this workgroup

  auto tile_origin = [&](int i, int& m0, int& n0) {
    // virtual workgroup id -> position t in the tile order: every XCD (workgroup b runs on XCD b % 8) walks a contiguous run of
    // the order, its 32 CUs 32 consecutive positions at a time (gridDim.x % 8 == 0 or gridDim.x == nwg)
    const int v = (int)blockIdx.x + i * (int)gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = v & 7, idx = v >> 3;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.reverse) t = nwg - 1 - t;
    // tile order: groups of group_m row panels, walked column by column inside a group, so that the tiles the CUs of one XCD work
    // on at the same time form a (group_m x 32 / group_m) block: they share group_m A panels AND 32 / group_m W panels through
    // that XCD's L2.  Row-major order (group_m = 1) makes the 32 CUs stream 32 different W panels: the wide-N GEMMs then run at
    // the fabric's ~4 TB/s, not at the matrix pipe's rate.
    const int gsz = p.group_m * p.tiles_n;
    const int grp = t / gsz, first = grp * p.group_m;
    const int rows = min(p.group_m, p.tiles_m - first);
    const int l = t - grp * gsz;
    const int tn = l / rows;
    m0 = (first + l - tn * rows) * BM;
    n0 = tn * BN;
  };

  // ------------------------------------------------------------------ issue side: two streams, A three K tiles ahead, W two
  // A piece is ONE `buffer_load_dwordx4 ... offen lds`: buffer descriptor (SGPRs) + per-lane byte offset (one VGPR, constant per
  // output tile -- per filter tap for a conv) + the K offset of the tile as the instruction's SCALAR offset: no vector instruction
  // per piece.  Lanes whose tap falls outside the image get an offset beyond the descriptor's range: the load returns zeros.
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A), 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.W), 0, 0x80000000u, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;            // >= num_records for every scalar offset < 2^31 (no 32-bit wrap)
  unsigned a_voff[PA], w_voff[PB];                 // per-lane byte offsets of this wave's pieces
  int a_oy[PA], a_ox[PA];                          // conv: input row / column of filter tap (0, 0) of the lane's output pixel
  unsigned a_img[PA];                              // conv: byte offset of the lane's image + its 16-byte slot
  const int kx0 = CONV && p.kw == 1 ? 1 : 0;       // 3 x 1 filter: the centre column is the only tap along x
  int ia_it = 0, ia_kt = 0, ia_k0 = 0, ia_c0 = 0, ia_ky = 0, ia_kx = kx0, ia_slot = 0;   // A stream: next tile to issue
  int iw_it = 0, iw_kt = 0, iw_k0 = 0, iw_slot = 0;                                   // W stream
  // per-lane offset of piece j for filter tap (ky, kx): once per tap (every Cin / 64 K tiles), not per K tile.  All pieces at once
  // at the tap change: the burst holds the MFMAs up for ~440 cycles (profiles/r03_sp_trace.log), but spreading it over the piece
  // gaps (each piece computing its next offset behind its last issue of the old tap) measured 3-7 % SLOWER on every conv
  // (profiles/r03_ab_transposed_sp.log): a scalar test per piece costs more than the burst per tap.
  auto conv_tap_offset = [&](int j, int ky, int kx) {
    const unsigned hup = p.Hin << p.upsample, wup = p.Win << p.upsample;
    const int iy = a_oy[j] + ky, ix = a_ox[j] + kx;
    const bool ok = (unsigned)iy < hup && (unsigned)ix < wup;
    const unsigned off = (((unsigned)(iy >> p.upsample) * (unsigned)p.Win + (unsigned)(ix >> p.upsample)) * (unsigned)p.ldx) * 2u + a_img[j];
    a_voff[j] = ok ? off : OOB;
  };
  auto conv_tap_offsets = [&]() {
#pragma unroll
    for (int j = 0; j < PA; ++j) conv_tap_offset(j, ia_ky, ia_kx);
  };
  auto set_sources_a = [&](int i) {
    int m0, n0;
    tile_origin(i, m0, n0);
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int row = (wave * PA + j) * RPI + lrow;
      const int lslot = pslot ^ ((row >> 1) & 7);             // source-side swizzle (the DMA writes LDS lane-linearly)
      const int m = m0 + row;
      const int mm = m < p.M ? m : p.M - 1;
      if (CONV) {
        const int hw = p.Hout * p.Wout;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        a_oy[j] = oy * p.stride - p.pad;
        a_ox[j] = ox * p.stride - p.pad;
        a_img[j] = (unsigned)b * (unsigned)(p.Hin * p.Win) * (unsigned)p.ldx * 2u + lslot * 16;   // < 2^31 (sp_eligible)
      } else {
        a_voff[j] = (unsigned)mm * (unsigned)p.lda * 2u + lslot * 16;                             // < 2^31 (sp_eligible)
      }
    }
    if (CONV) conv_tap_offsets();
  };
  auto set_sources_w = [&](int i) {
    int m0, n0;
    tile_origin(i, m0, n0);
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (wave * PB + j) * RPI + lrow;
      w_voff[j] = (unsigned)(n0 + row) * (unsigned)p.K * 2u + (pslot ^ ((row >> 1) & 7)) * 16;    // N % BN == 0 (launcher)
    }
  };
  set_sources_a(0);
  set_sources_w(0);
  auto issue_a = [&](int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(smem + ia_slot * ASZ + (wave * PA + j) * 1024), 16, (SP_ABL & 16) ? (unsigned)(lane * 16 + j * 1024) : a_voff[j], (SP_ABL & 16) ? 0 : (CONV ? ia_c0 : ia_k0) * 2, 0, 0);
  };
  auto issue_w = [&](int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(smem + WBASE + iw_slot * WSZ + (wave * PB + j) * 1024), 16, (SP_ABL & 16) ? (unsigned)(lane * 16 + j * 1024) : w_voff[j], (SP_ABL & 16) ? 0 : iw_k0 * 2, 0, 0);
  };
  auto advance_a = [&]() {
    ia_k0 += BK;
    if (++ia_slot == 3) ia_slot = 0;
    if (CONV) {
      ia_c0 += BK;
      if (ia_c0 == p.Cin) {
        ia_c0 = 0;
        if (p.kw == 1) ++ia_ky;
        else if (++ia_kx == 3) { ia_kx = 0; ++ia_ky; }
        if (ia_kt + 1 < nk) conv_tap_offsets();
      }
    }
    if (++ia_kt == nk) {
      ia_kt = 0;
      ia_k0 = ia_c0 = ia_ky = 0;
      ia_kx = kx0;
      if (++ia_it < ntile) set_sources_a(ia_it);    // past the last tile: keep its offsets (valid addresses, data never read)
      else if (CONV) conv_tap_offsets();
    }
  };
  auto advance_w = [&]() {
    iw_k0 += BK;
    iw_slot ^= 1;
    if (++iw_kt == nk) {
      iw_kt = 0;
      iw_k0 = 0;
      if (++iw_it < ntile) set_sources_w(iw_it);
    }
  };
  // piece q = 0..15 of the list [W pieces 0..PB-1 | A pieces 0..PA-1]; each stream moves on behind its last piece
  auto issue_q = [&](int q) {
    if (q < PB) {
      issue_w(q < PB ? q : 0);
      if (q == PB - 1) advance_w();
    } else {
      issue_a(q < PB ? 0 : q - PB);
      if (q == PB + PA - 1) advance_a();
    }
  };

  // ------------------------------------------------------------------ compute side
  const int frow = lane & 31, fhi = lane >> 5;
  int a_rd[4][MT], b_rd[4][NT];                    // per-lane LDS byte offsets of the fragment reads of k-step s = 0..3 (within a slot)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ra = wm * (32 * MT) + i * 32 + frow;
      a_rd[s][i] = ra * ROWB + (((s * 2 + fhi) ^ ((ra >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int rb = wn * (32 * NT) + j * 32 + frow;
      b_rd[s][j] = WBASE + rb * ROWB + (((s * 2 + fhi) ^ ((rb >> 1) & 7)) << 4);
    }
  }
  floatx16 acc[MT][NT];
  half8_t fa0[MT], fb0[NT], fa1[MT], fb1[NT];

  // ------------------------------------------------------------------ RESM: the residual enters through the matrix core (round 6)
  // acc = mfma(W, A) holds C^T: D[n][m] += sum_k First[n][k] Second[m][k].  With First = a 32 x 16 slice of the IDENTITY (First[n][k] =
  // [n == 16 c + k], c = 0, 1) and Second = the residual rows themselves (Second[m][k] = R[m][n0 + 16 c + k]: for lane (frow, fhi) the 16
  // contiguous bytes R[m = frow][n0 + 16 c + 8 fhi ..+7], ONE buffer_load_dwordx4 straight into the operand registers, no LDS), two MFMAs
  // add fp32(R) -- exactly: products by 1.0 and by 0.0 -- to a 32 x 32 accumulator sub-tile.  Being an accumulation like any other it can
  // go ANYWHERE in the K loop: sub-tile u = 0 .. MT NT - 1 of an output tile requests its two pieces in K tile u (behind the last W piece
  // of the list, at the top of k-step 1) and multiplies them in K tile u + 1 behind the barrier, 1.5 K tiles ~ 3 800 cycles later; two
  // register generations (u & 1), 16 + 8 + MT registers.  The residual read is thereby spread over the K loop as 2 extra loads per wave and
  // K tile next to its 10-16 DMA pieces, the epilogue is the bias-only one, and the lock-step burst of 256 CUs all reading their residual
  // tiles at once (12-25 k cycles per 192 x 320 tile, profiles/r06_sp_epilogue_ablation.log) is gone for MT NT x 2 extra MFMAs per tile
  // (+2.5 % at K = 1280, +0.4 % on a 3 x 3 conv).  vmcnt: a K tile that requests a residual piece pair lets PA + 2 operations fly over its
  // barrier (the A pieces and the pair, all younger than every W piece); the pair is older than everything the NEXT barrier lets fly.
  // In-place (residual == C) stays legal: an output tile reads its residual block before its own epilogue stores, and no other tile's.
  constexpr int NSUB = MT * NT;
  [[maybe_unused]] half8_t rid[2];
  [[maybe_unused]] uint4_t rfr[2][2];
  [[maybe_unused]] unsigned r_voff[MT];
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(RESM ? p.residual : p.A), 0, 0x80000000u, 0x00020000);
  if constexpr (RESM) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) rid[c][e] = frow == 16 * c + 8 * fhi + e ? (half_t)1 : (half_t)0;
  }
  [[maybe_unused]] auto set_sources_r = [&](int i) {
    int m0, n0;
    tile_origin(i, m0, n0);
#pragma unroll
    for (int ii = 0; ii < MT; ++ii) {
      const int m = m0 + wm * (32 * MT) + ii * 32 + frow;
      const int mm = m < p.M ? m : p.M - 1;
      r_voff[ii] = ((unsigned)mm * (unsigned)p.ldr + (unsigned)(n0 + wn * (32 * NT) + 8 * fhi)) * 2u;      // < 2^31 (launcher)
    }
  };
  // request the two pieces of sub-tile U (compile-time) / multiply the pieces of sub-tile U
#define SP_RES_ISSUE(U)                                                                                     \
  if constexpr ((U) < NSUB) {                                                                               \
    constexpr int i_ = (U) / NT < MT ? (U) / NT : 0, j_ = (U) % NT, g_ = (U) & 1;                          \
    rfr[g_][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, r_voff[i_] + j_ * 64, 0, 0);                \
    rfr[g_][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, r_voff[i_] + j_ * 64 + 32, 0, 0);           \
  }
#define SP_RES_MULT(U)                                                                                      \
  if constexpr ((U) < NSUB) {                                                                               \
    constexpr int i_ = (U) / NT < MT ? (U) / NT : 0, j_ = (U) % NT, g_ = (U) & 1;                          \
    half8_t r0_, r1_;                                                                                       \
    __builtin_memcpy(&r0_, &rfr[g_][0], 16);                                                                \
    __builtin_memcpy(&r1_, &rfr[g_][1], 16);                                                                \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rid[0], r0_, acc[i_][j_], 0, 0, 0);                \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rid[1], r1_, acc[i_][j_], 0, 0, 0);                \
  }
  static_assert(NSUB <= 16, "the peeled K tiles below cover 16 sub-tiles");

  // One k-step, issue order pinned by hand (sched_barrier(0) lets nothing cross): MFMA k of the current fragments (FAU, FBU),
  // then -- behind each of the first MT + NT MFMAs -- ONE ds_read_b128 of the next k-step's fragments (FAL, FBL: A slot offset
  // SA, W slot offset SW, k-step S), and NP DMA pieces Q0 .. Q0+NP-1 of the list, evenly spaced.
#define SP_STEP(FAU, FBU, FAL, FBL, SA, SW, S, ZERO, Q0, NP)                                                \
  {                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < MT * NT; ++k) {                                                   \
      const int i = k / NT, j = k % NT;                                                                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FBU[j], FAU[i], (ZERO) ? floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : acc[i][j], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      if (SP_ABL & 4) {                                                                                     \
      } else if (k < MT) {                                                                                  \
        FAL[k < MT ? k : 0] = *reinterpret_cast<const half8_t*>(smem + (SA) + a_rd[S][k < MT ? k : 0]);     \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      } else if (k < MT + NT) {                                                                             \
        FBL[k < MT ? 0 : k - MT] = *reinterpret_cast<const half8_t*>(smem + (SW) + b_rd[S][k < MT ? 0 : k - MT]); \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      }                                                                                                     \
      if (!(SP_ABL & 2)) {                                                                                  \
        constexpr int STRIDE = (NP) == 6 ? 2 : ((NP) == 4 ? (MT * NT >= 12 ? 3 : 2) : ((NP) == 3 ? (MT * NT >= 8 ? 4 : 2) : (MT * NT >= 12 ? 6 : (MT * NT >= 8 ? 4 : 3)))); \
        static_assert(STRIDE * ((NP) - 1) + 1 < MT * NT, "a DMA piece per MFMA gap at most");               \
        if (k % STRIDE == 1 && k / STRIDE < (NP)) {                                                         \
          issue_q((Q0) + k / STRIDE);                                                                       \
          __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                   \
      }                                                                                                     \
    }                                                                                                       \
  }

  // ------------------------------------------------------------------ prologue
  // W(0), A(0), A(1), then the first 6 pieces of the list [W(1) | A(2)] (the share of a step 3); the steps 0..2 of tile 0 issue
  // the other 10 like every later tile
#pragma unroll
  for (int j = 0; j < PB; ++j) issue_w(j);
  advance_w();
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int j = 0; j < PA; ++j) issue_a(j);
    advance_a();
  }
#pragma unroll
  for (int q = 0; q < NP0; ++q) issue_q(q);
  wait_vmcnt<PA + NP0>();                                              // this wave's pieces of W(0) and A(0)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < MT; ++i) fa0[i] = *reinterpret_cast<const half8_t*>(smem + a_rd[0][i]);
#pragma unroll
  for (int j = 0; j < NT; ++j) fb0[j] = *reinterpret_cast<const half8_t*>(smem + b_rd[0][j]);

  // One K tile of the main loop.  ZERO: first K tile of an output tile (its first k-step accumulates onto the inline constant 0).
  // The first K tile is PEELED out of the K loop instead of being selected inside it: a select would merge two definitions of
  // every accumulator at one program point, and the register allocator then shuffles the 240 accumulators through VGPRs and
  // scratch on every iteration.
#define SP_BODY(ZERO, KT)                                                                                   \
  {                                                                                                         \
    SP_STAMP(0)                                                                                             \
    SP_STEP(fa0, fb0, fa1, fb1, ca, cw, 1, ZERO, NP0, NP1)                                                  \
    SP_STAMP(1)                                                                                             \
    if constexpr (RESM && (KT) < NSUB) {                                                                    \
      SP_RES_ISSUE(KT)                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
    SP_STEP(fa1, fb1, fa0, fb0, ca, cw, 2, false, NP0 + NP1, NP2)                                           \
    SP_STAMP(2)                                                                                             \
    SP_STEP(fa0, fb0, fa1, fb1, ca, cw, 3, false, NP0 + NP1 + NP2, NP3)                                     \
    SP_STAMP(3)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    if (!(SP_ABL & 8)) {                 /* W(t+1), A(t+1) of this wave have landed; its A(t+2) pieces may fly */ \
      if (RESM && (KT) < NSUB) wait_vmcnt<PA + 2>();   /* ... and the residual pair requested in this K tile */ \
      else wait_vmcnt<PA>();                                                                                \
    }                                                                                                       \
    if (!(SP_ABL & 1)) __builtin_amdgcn_s_barrier();                                                        \
    SP_STAMP(4)                                                                                             \
    ca += ASZ;                                                                                              \
    if (ca == 3 * ASZ) ca = 0;                                                                              \
    cw ^= WSZ;                                                                                              \
    if constexpr (RESM && (KT) >= 1 && (KT) <= NSUB) {                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      SP_RES_MULT((KT) - 1)                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
    SP_STEP(fa1, fb1, fa0, fb0, ca, cw, 0, false, 0, NP0)                                                   \
    SP_TRACE_NEXT                                                                                           \
  }

  int ca = 0, cw = 0;                               // A / W slot offsets of the tile being multiplied
#ifdef SP_TRACE
  int tr_kt = 0;
#define SP_TRACE_NEXT ++tr_kt;
#else
#define SP_TRACE_NEXT
#endif
#pragma unroll 1
  for (int ct = 0; ct < ntile; ++ct) {
    if constexpr (RESM) {
      // the first NSUB + 1 K tiles of an output tile are peeled: K tile u requests sub-tile u's residual pieces, K tile u + 1 multiplies
      // them, all with compile-time accumulator indices (a runtime switch over the sub-tiles merges 16 definitions of every accumulator
      // and the register allocator answers with ~800 v_accvgpr_mov per K tile); nk >= NSUB + 1 (launcher)
      set_sources_r(ct);
      SP_BODY(true, 0)
#define SP_PEEL(U) if constexpr ((U) <= NSUB) SP_BODY(false, U)
      SP_PEEL(1) SP_PEEL(2) SP_PEEL(3) SP_PEEL(4) SP_PEEL(5) SP_PEEL(6) SP_PEEL(7) SP_PEEL(8)
      SP_PEEL(9) SP_PEEL(10) SP_PEEL(11) SP_PEEL(12) SP_PEEL(13) SP_PEEL(14) SP_PEEL(15) SP_PEEL(16)
#undef SP_PEEL
#pragma unroll 1
      for (int kt = NSUB + 1; kt < nk; ++kt) SP_BODY(false, 99)
    } else {
      SP_BODY(true, 0)
#pragma unroll 1
      for (int kt = 1; kt < nk; ++kt) SP_BODY(false, 99)
    }
    {
      // ---- epilogue of output tile ct, straight from the accumulators
      int m0, n0;
      tile_origin(ct, m0, n0);
      const int lc = lane & 31, hi = lane >> 5;
      // the last MFMAs are still writing their accumulators: 18 wait states before the first v_accvgpr_read (the hazard recognizer
      // does not look inside asm statements)
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      if constexpr (GEGLU) {
        // sub-tile 2q is h, 2q+1 is g of the same outputs (weight rows packed as [32 h | 32 g] blocks)
        // the bias of BOTH column pairs first: requested behind the stores of pair 0, pair 1's bias would be the youngest operation of the wave
        // when it is needed, and the wait for it (vmcnt(0)) would also wait for those stores to be acknowledged
        float2_t bh_[NT / 2][8], bg_[NT / 2][8];                    // pair c = 2 gi + e / 2: columns 8 gi + 4 hi + {e, e + 1}
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          const half_t* bias = (p.bias ? p.bias : g_zero_cols) + n0 + wn * (32 * NT) + q * 64 + 4 * hi;
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            const half4_t h4 = *reinterpret_cast<const half4_t*>(bias + 8 * gi);
            const half4_t g4 = *reinterpret_cast<const half4_t*>(bias + 32 + 8 * gi);
            bh_[q][2 * gi] = float2_t{(float)h4[0], (float)h4[1]};
            bh_[q][2 * gi + 1] = float2_t{(float)h4[2], (float)h4[3]};
            bg_[q][2 * gi] = float2_t{(float)g4[0], (float)g4[1]};
            bg_[q][2 * gi + 1] = float2_t{(float)g4[2], (float)g4[3]};
          }
        }
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          const int nc = n0 + wn * (32 * NT) + q * 64;
          const float2_t (&bh)[8] = bh_[q];
          const float2_t (&bg)[8] = bg_[q];
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            half_t* drow[2];                                        // rows (lc & 15) and 16 + (lc & 15): 64-byte runs per row (see sp_plain_epilogue)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int m = m0 + wm * (32 * MT) + i * 32 + 16 * h + (lc & 15);
              drow[h] = m < p.M ? p.C + (size_t)m * p.ldc + (nc >> 1) + 16 * ((lc >> 4) & 1) + 8 * hi : g_sp_dump + (lane << 3);
            }
            float2_t gl[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
              gl[c] = float2_t{sp_acc(acc[i][2 * q + 1], 2 * c), sp_acc(acc[i][2 * q + 1], 2 * c + 1)} + bg[c];
            __builtin_amdgcn_sched_barrier(0);
            gelu_fast2_x<8>(gl);                                   // the eight chains of the row side by side
            __builtin_amdgcn_sched_barrier(0);
            unsigned w[4][2];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const float2_t hv = (float2_t{sp_acc(acc[i][2 * q], 2 * c), sp_acc(acc[i][2 * q], 2 * c + 1)} + bh[c]) * gl[c];
              typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
              const h2_t o = {(half_t)hv.x, (half_t)hv.y};
              __builtin_memcpy(&w[c >> 1][c & 1], &o, 4);
            }
            unsigned pc[2][4];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const auto s0 = __builtin_amdgcn_permlane32_swap(w[2 * pr][0], w[2 * pr + 1][0], false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(w[2 * pr][1], w[2 * pr + 1][1], false, false);
              pc[pr][0] = s0[0]; pc[pr][1] = s1[0]; pc[pr][2] = s0[1]; pc[pr][3] = s1[1];
            }
            uint4 vv[2];
            {
              const auto t0 = __builtin_amdgcn_permlane16_swap(pc[0][0], pc[1][0], false, false);
              const auto t1 = __builtin_amdgcn_permlane16_swap(pc[0][1], pc[1][1], false, false);
              const auto t2 = __builtin_amdgcn_permlane16_swap(pc[0][2], pc[1][2], false, false);
              const auto t3 = __builtin_amdgcn_permlane16_swap(pc[0][3], pc[1][3], false, false);
              vv[0] = uint4{t0[0], t1[0], t2[0], t3[0]};
              vv[1] = uint4{t0[1], t1[1], t2[1], t3[1]};
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) *reinterpret_cast<uint4*>(drow[h]) = vv[h];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        // the bias is always added (absent: a page of zeros); residual and row-broadcast operand split the code (uniform branches)
        const int mw = m0 + wm * (32 * MT), nw = n0 + wn * (32 * NT);
        if constexpr (RESM) {                        // the residual is in the accumulators already
          if (p.rowadd) sp_plain_epilogue<MT, NT, false, true>(p, acc, mw, nw, lc, hi);
          else sp_plain_epilogue<MT, NT, false, false>(p, acc, mw, nw, lc, hi);
        } else if (p.bias_rows) sp_plain_epilogue<MT, NT, false, false, true, true>(p, acc, mw, nw, lc, hi);
        else if (p.residual && p.rowadd && !(SP_ABL & 32)) sp_plain_epilogue<MT, NT, true, true, true, false, CONV>(p, acc, mw, nw, lc, hi);
        else if (p.residual && !(SP_ABL & 32)) sp_plain_epilogue<MT, NT, true, false, true, false, CONV>(p, acc, mw, nw, lc, hi);
        else if (p.rowadd) sp_plain_epilogue<MT, NT, false, true>(p, acc, mw, nw, lc, hi);
        else sp_plain_epilogue<MT, NT, false, false>(p, acc, mw, nw, lc, hi);
      }
    }
  }
  wait_vmcnt<0>();                                                   // the DMA pieces issued past the last tile
#ifdef SP_TRACE
  if (!GEGLU && blockIdx.x == 0) {
    __syncthreads();
    for (int i = tid; i < 4 * SP_TRACE_N * 5; i += 256)
      (&g_sp_trace[0][0][0])[i] = *reinterpret_cast<const unsigned long long*>(smem + (3 * BM + 2 * BN) * 128 + i * 8);
  }
#endif
#undef SP_TRACE_NEXT
#undef SP_BODY
#undef SP_STEP
#undef SP_RES_MULT
#undef SP_RES_ISSUE
#endif
}

template <bool CONV, bool GEGLU, int NT = GEGLU ? 4 : 5>
static bool sp_eligible(const GemmParams& p) {
  constexpr int BN = 64 * NT;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // N <= 16384: the zero page standing in for absent column operands; the row-bias form (swapped operands) has no column operands
  if (p.transpose_out || p.N % BN != 0 || p.K % 64 != 0 || p.K < 128 || (p.N > 16384 && !p.bias_rows)) return false;
  if (p.bias_rows && (p.residual || p.rowadd || CONV || GEGLU)) return false;
  if (!GEGLU && p.act != ACT_NONE) return false;
  if (CONV && p.Cin % 64 != 0) return false;
  // the DMA pieces address A and W through buffer descriptors with 32-bit byte offsets; offsets from 2^31 up mean "outside"
  const unsigned long long a_bytes = CONV ? (unsigned long long)cdiv(p.M, p.Hout * p.Wout) * p.Hin * p.Win * p.ldx * 2
                                          : ((unsigned long long)(p.M - 1) * p.lda + p.K) * 2;
  if (a_bytes >= (1ull << 31) || (unsigned long long)p.N * p.K * 2 >= (1ull << 31)) return false;
  if (!al16(p.C) || p.ldc % 8 != 0) return false;
  if (p.bias && !p.bias_rows && !al16(p.bias)) return false;
  if (p.residual && (!al16(p.residual) || p.ldr % 8 != 0)) return false;
  if (p.rowadd && (!al16(p.rowadd) || p.ldra % 8 != 0)) return false;
  return true;
}

// Residual through the matrix core (RESM, see the kernel): every 32 x 32 sub-tile of the wave tile needs a K tile of its own plus one, and
// the residual is addressed through a buffer descriptor (32-bit byte offsets).  MD_SP_RESM = 0: the round-3 epilogue (A/B runs).
template <int MT, int NT>
static bool sp_resm(const GemmParams& p) {
  static const int resm = md_env_int("MD_SP_RESM", 1);
  return resm && p.residual && !p.bias_rows && !(SP_ABL & 32) && p.K / 64 >= MT * NT + 1 &&
         ((unsigned long long)(p.M - 1) * p.ldr + p.N) * 2 < (1ull << 31);
}

template <bool CONV, bool GEGLU, int NT = GEGLU ? 4 : 5, int MT = GEGLU ? 4 : 3>
static void launch_sp(GemmParams& p, hipStream_t stream) {
  constexpr int BM = 64 * MT, BN = 64 * NT;
#ifdef SP_TRACE
  constexpr size_t smem = (size_t)(3 * BM + 2 * BN) * 128 + (GEGLU ? 0 : 4 * SP_TRACE_N * 5 * 8);
  static_assert(smem <= 160 * 1024, "trace build: the stamps live behind the ring");
#else
  constexpr size_t smem = (size_t)(3 * BM + 2 * BN) * 128;          // A ring of three, W ring of two 64-deep K tiles
#endif
  md_ensure_dynamic_lds<gemm_sp_kernel<CONV, GEGLU, MT, NT>>((int)smem);
  constexpr int group_m = 8;       // row-major order (1) measured 3-38 % slower on the wide-N shapes (profiles/r03_ab_gemm_sp.log)
  p.tiles_n = p.N / BN;
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_total = p.tiles_m * p.tiles_n;
  p.group_m = group_m;
  // FeedForward's output projection (K = 4 N: the only plain GEMM whose A -- the GEGLU hidden tensor, 755 MB at the 96 x 96 level -- is known
  // to have just been written front to back by the previous kernel) walks A BACKWARDS when A exceeds the 256-MiB memory-side cache of
  // the MI355X: its tail is the part still cached (+0.1 % end to end, profiles/r05_ab_ffout_reverse_order.log: inside run-to-run noise,
  // hence restricted to the shape it was measured on; the permutation is a bijection of the tile order, results are unaffected).
  // MD_SP_REVERSE = 0 switches it off, 2 restores the round-5 rule (every plain GEMM with A > 256 MiB).
  static const int rev = md_env_int("MD_SP_REVERSE", 1);
  const bool big_a = (size_t)p.M * (size_t)p.lda * 2 > ((size_t)256 << 20);
  p.reverse = !CONV && !GEGLU && !p.bias_rows && big_a && (rev == 2 || (rev == 1 && p.K >= 4 * p.N));
  const int ncu = md_device_cus();
  const int grid = p.tiles_total < ncu ? p.tiles_total : ncu;
  if constexpr (!GEGLU) {
    if (sp_resm<MT, NT>(p)) {
      md_ensure_dynamic_lds<gemm_sp_kernel<CONV, false, MT, NT, true>>((int)smem);
      hipLaunchKernelGGL((gemm_sp_kernel<CONV, false, MT, NT, true>), dim3(grid), dim3(256), smem, stream, p);
      return;
    }
  }
  hipLaunchKernelGGL((gemm_sp_kernel<CONV, GEGLU, MT, NT>), dim3(grid), dim3(256), smem, stream, p);
}

#ifdef SP_TRACE
extern "C" int md_debug_sp_trace(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sp_trace), sizeof(unsigned long long) * 4 * SP_TRACE_N * 5);
}
#endif
