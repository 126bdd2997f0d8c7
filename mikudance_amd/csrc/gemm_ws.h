// W-stationary streaming GEMM for the HBM-bound short-K projections (included by gemm.hip).
//
//   C[M, N] = epi( A[M, K] . W[N, K]^T ),   K = 320 or 640, N a multiple of the column group (320 / 128), M large.
//
// The level-0 / level-1 Linear layers of both UNets (to_q / to_k / to_out, proj_in / proj_out, the motion module's qkv and
// out projections: reference src/models/attention.py:109-157,323-364, src/models/transformer_3d.py:66-98,
// src/models/motion_module.py:124-146,293-317) have K = N = C = 320 / 640 on M = 294 912 / 73 728 rows: 160 FLOP per byte of
// A + C, half the machine balance, i.e. they are HBM streams (377 MB per launch at C = 320) with a small GEMM attached.  The
// tiled kernel reloads the weight tile with every output tile and waits on a 5-step K loop per tile (2.4-3.6 TB/s).  Here the
// roles are turned around:
//   * the WEIGHTS are stationary in REGISTERS: one persistent 512-thread workgroup per CU, its four compute waves (one per
//     SIMD) each own 16*CB output columns x all of K as MFMA operand fragments (<= 200 VGPRs), loaded once;
//   * A streams HBM -> LDS through a deep ring of 16-row stages filled by direct-to-LDS DMA (global_load_lds_dwordx4), up to
//     110 KiB in flight per CU;
//   * waves are specialised so that every counted s_waitcnt sees ONE kind of memory operation:
//       waves 0-3  compute : LDS fragment reads + v_mfma_f32_16x16x32_f16 (C^T = W.A^T, so a lane ends up with 4 consecutive
//                            output columns of one row), fp32 results to an LDS staging tile (double buffered)
//       waves 4-5  loaders : issue the DMA of tile t+NS-1 and wait (vmcnt counts only their own DMAs: in-order) for tile t+1
//       waves 6-7  stores  : tile t-1: staging tile + bias (+ row-broadcast term) (+ residual, prefetched two tiles ahead)
//                            -> one rounding -> coalesced 16-byte global stores
//     ONE s_barrier per 16-row tile hands the stages round.
//   * a launch with G = N / group column groups runs G workgroups side by side on the same row stream inside one XCD, so
//     the G-1 re-reads of an A stage hit that XCD's L2.
// LDS bank conflicts: the DMA writes a stage lane-linearly, so the swizzle is applied to the per-lane SOURCE address and to
// the fragment read (same involution): 16-byte slot c of row r sits at c ^ s(r), s(r) = (r >> 1) & 7 for 40 slots per row
// (K = 320: consecutive rows already shift by 8 slots) and r & 15 for 80 (K = 640): every ds_read_b128 of a 16-row x 32-k
// fragment touches 16 distinct slots per lane group.
#pragma once

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

template <int KS, int CB>
struct WsCfg {
  static constexpr int K = 32 * KS;
  static constexpr int CPR = K / 8;               // 16-byte slots per A row
  static constexpr int GC = 64 * CB;              // output columns per workgroup
  static constexpr int TR = 16;                   // rows per tile
  static constexpr int STAGE = TR * K * 2;        // bytes per A stage
  static constexpr int CS_LD = GC + 4;            // fp32 staging pitch (floats): rows shift by 4 banks
  static constexpr int CSTAGE = TR * CS_LD * 4;
  static constexpr int NS = (160 * 1024 - 2 * CSTAGE) / STAGE > 12 ? 12 : (160 * 1024 - 2 * CSTAGE) / STAGE;
  static constexpr int DPT = STAGE / 1024;        // DMA wave-instructions per tile
  static constexpr int PER = DPT / 2;             // ... per loader wave
  static constexpr int SMEM = NS * STAGE + 2 * CSTAGE;
  static constexpr int CHUNKS = TR * GC / 8;      // 16-byte output pieces per tile
  static constexpr int SPL = CHUNKS / 128;        // ... per lane of the two store waves
  static_assert(DPT % 2 == 0 && CHUNKS % 128 == 0, "tile must split evenly over the loader / store waves");
  static_assert((NS - 2) * PER <= 63, "vmcnt immediate");
};

template <int CPR>
__device__ __forceinline__ int ws_swz(int row) {
  return CPR == 40 ? ((row >> 1) & 7) : (row & 15);
}

template <int KS, int CB>
__global__ __launch_bounds__(512, 1) void wsgemm_kernel(WsParams p) {
  using Cfg = WsCfg<KS, CB>;
  constexpr int K = Cfg::K, CPR = Cfg::CPR, GC = Cfg::GC, TR = Cfg::TR, STAGE = Cfg::STAGE, NS = Cfg::NS, CS_LD = Cfg::CS_LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + NS * STAGE);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // workgroup -> (XCD, column group, row stream): the G groups of one row stream share an XCD (blockIdx % 8)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = slot % p.groups, sl = slot / p.groups;
  if (sl >= p.spx) return;
  const int stream = xcd * p.spx + sl;
  const int ntiles = (p.M + TR - 1) / TR;
  const int my_tiles = stream < ntiles ? (ntiles - stream + p.streams - 1) / p.streams : 0;   // tiles stream, stream+S, ...
  if (my_tiles == 0) return;
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
    for (int it = 0; it < my_tiles; ++it) {
      __builtin_amdgcn_s_barrier();                 // b_it: tile `it` has landed; staging buffer it & 1 is free
      const char* st = ring + (it % NS) * STAGE;
      floatx4 acc[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8_t af = *reinterpret_cast<const half8_t*>(st + (rbase + ((4 * ks + kq) ^ sw)) * 16);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cb][ks], af, acc[cb], 0, 0, 0);
      }
      // acc[cb][r] = C[m = row][n = wave*16CB + cb*16 + 4*kq + r]
      float* cs = cst + (it & 1) * (TR * CS_LD) + row * CS_LD + wave * 16 * CB + 4 * kq;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<floatx4*>(cs + cb * 16) = acc[cb];
    }
    __builtin_amdgcn_s_barrier();                   // b_{my_tiles}: the last staging tile is complete
  } else if (wave < 6) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    auto issue = [&](int it) {
      const int m0 = (stream + it * p.streams) * TR;
      char* st = ring + (it % NS) * STAGE;
#pragma unroll
      for (int i = 0; i < Cfg::PER; ++i) {
        const int base = __builtin_amdgcn_readfirstlane((lw * Cfg::PER + i) * 64);   // first 16-byte slot of this instruction
        const int pidx = base + lane;
        const int r = pidx / CPR, c = pidx % CPR;
        const int gr = min(m0 + r, p.M - 1);
        const half_t* src = p.A + (size_t)gr * p.lda + ((c ^ ws_swz<CPR>(r)) << 3);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + base * 16), 16, 0, 0);
      }
    };
    const int pre = my_tiles < NS - 1 ? my_tiles : NS - 1;
    for (int it = 0; it < pre; ++it) issue(it);
    for (int it = 0; it < my_tiles; ++it) {
      // tile `it` must have landed before the barrier: at most the DMAs of the tiles it+1 .. it+NS-2 may stay in flight
      if (it + NS - 2 < my_tiles) wait_vmcnt<(NS - 2) * Cfg::PER>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                 // also: the compute waves are done with tile it-1 -> its stage is free
      if (it + NS - 1 < my_tiles) issue(it + NS - 1);   // into stage (it - 1) % NS
    }
    __builtin_amdgcn_s_barrier();                   // b_{my_tiles}
  } else {
    // ------------------------------------------------------------------------------------------------ store waves
    const int sid = (wave - 6) * 64 + lane;         // 0..127
    constexpr int CPRO = GC / 8;                    // 16-byte pieces per output row
    int prow[Cfg::SPL], pcol[Cfg::SPL];
    half8_t bias[Cfg::SPL];
#pragma unroll
    for (int i = 0; i < Cfg::SPL; ++i) {
      const int id = i * 128 + sid;
      prow[i] = id / CPRO;
      pcol[i] = (id % CPRO) * 8;
      bias[i] = p.bias ? *reinterpret_cast<const half8_t*>(p.bias + n0 + pcol[i]) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }
    half8_t res[2][Cfg::SPL];
    auto fetch_res = [&](int it, half8_t (&dst)[Cfg::SPL]) {
      const int m0 = (stream + it * p.streams) * TR;
#pragma unroll
      for (int i = 0; i < Cfg::SPL; ++i) {
        const int m = min(m0 + prow[i], p.M - 1);
        dst[i] = *reinterpret_cast<const half8_t*>(p.residual + (size_t)m * p.ldr + n0 + pcol[i]);
      }
    };
    auto store_tile = [&](int it, const half8_t (&rs)[Cfg::SPL]) {
      const int m0 = (stream + it * p.streams) * TR;
      const float* cs = cst + (it & 1) * (TR * CS_LD);
#pragma unroll
      for (int i = 0; i < Cfg::SPL; ++i) {
        const int m = m0 + prow[i];
        const floatx4 a = *reinterpret_cast<const floatx4*>(cs + prow[i] * CS_LD + pcol[i]);
        const floatx4 b = *reinterpret_cast<const floatx4*>(cs + prow[i] * CS_LD + pcol[i] + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)bias[i][j];
        if (p.rowadd && m < p.M) {
          const half8_t ra = *reinterpret_cast<const half8_t*>(p.rowadd + (size_t)(m / p.rows_per_group) * p.ldra + n0 + pcol[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)ra[j];
        }
        if (p.residual) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rs[i][j];
        }
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
        if (m < p.M) *reinterpret_cast<half8_t*>(p.C + (size_t)m * p.ldc + n0 + pcol[i]) = o;
      }
    };
    if (p.residual) {
      fetch_res(0, res[0]);
      if (my_tiles > 1) fetch_res(1, res[1]);
    }
    // barrier b_it (it = 0 .. my_tiles): afterwards the compute waves work on tile it and this wave stores tile it-1, then
    // prefetches the residual of tile it+1 (two tile periods ahead).  Unrolled by two so that the two residual buffers are
    // addressed statically: even tiles use res[0], odd tiles res[1].
    for (int it = 0; it <= my_tiles; it += 2) {
      __builtin_amdgcn_s_barrier();                                   // b_it
      if (it >= 1) {
        store_tile(it - 1, res[1]);
        if (p.residual && it + 1 < my_tiles) fetch_res(it + 1, res[1]);
      }
      if (it + 1 > my_tiles) break;
      __builtin_amdgcn_s_barrier();                                   // b_{it+1}
      store_tile(it, res[0]);
      if (p.residual && it + 2 < my_tiles) fetch_res(it + 2, res[0]);
    }
  }
}
