// MFMA GEMM / implicit-GEMM 3x3 NHWC convolution for gfx950.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )            fp16 in, fp32 accumulate (v_mfma_f32_32x32x16_f16)
//
// One kernel, two A-operand address generators:
//   PLAIN : A is a row-major [M][lda] fp16 matrix (Linear / 1x1 conv on NHWC tokens).
//   CONV3 : A is an NHWC fp16 image batch; row m = (b, oy, ox), column k = (ky*3+kx)*Cin + c; zero padding 1,
//           stride 1 or 2, optional nearest-2x upsample folded into the input addressing.  Cin % 64 == 0 so every
//           K chunk lies inside a single filter tap; out-of-image taps read a 64-byte zero page (branch-free select).
// Replaces the ATen conv2d / linear calls behind InflatedConv3d (reference src/models/resnet.py:9-17), diffusers
// ResnetBlock2D/Downsample2D/Upsample2D convs, Attention.to_q/k/v/to_out, FeedForward and the 1x1 proj_in/proj_out
// of Transformer2D/3DModel.
//
// Tile (64*WM) x 128 x BK per workgroup of 2*WM waves (wave grid WM x 2, 64x64 per wave = 2x2 MFMA 32x32 tiles):
//   WM = 2: 128x128, 256 threads (skinny / small problems, up to 3 workgroups per CU)
//   WM = 4: 256x128, 512 threads, 3-deep ring of 48-KiB stages (large problems: 25 % fewer operand bytes per MFMA)
// Operand tiles go HBM/L2 -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4, 16 B per lane, no VGPR round trip)
// into an NSTAGE-deep ring: NSTAGE-1 tiles are in flight while one is being multiplied, waits are COUNTED
// (s_waitcnt vmcnt(N), never 0 inside the loop while tiles remain) and there is one raw s_barrier per K step.
// The DMA writes LDS lane-linearly, so the bank-conflict swizzle is applied to the per-lane SOURCE address and to the
// fragment read address (same involution): the 16-B slot of row r is XORed with (r>>2)&3 (BK=32) / (r>>1)&7 (BK=64),
// which makes every ds_read_b128 fragment read conflict free.
// Epilogues: (a) bias / SiLU / ReLU / row-broadcast (time embedding) / residual / transposed V^T store: the fp32
// accumulators are staged through LDS (64-row passes) so that every global access is a full 16-byte piece of a
// coalesced row (8-byte row-strided accesses straight from the MFMA layout measured 40 % slower on the HBM-bound skinny
// GEMMs); residual rows are fetched before the staging barriers.  (b) GEGLU: operand roles swapped so that h and g of
// an output element meet in one lane; written straight from the accumulators (8 % faster than staging).
#include "common.h"
#include <stdlib.h>
#include <algorithm>


enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_GEGLU = 3, ACT_QUICKGELU = 4 };  // 4: x * sigmoid(1.702 x) (CLIP)

struct GemmParams {
  const half_t* A;
  const half_t* W;
  half_t* C;
  const half_t* bias;
  const half_t* residual;
  const half_t* rowadd;
  int lda, ldc, ldr, ldra;
  int M, N, K;
  int rows_per_group;
  int act;
  int transpose_out;
  int bias_rows;          // gemm_sp_kernel on swapped operands (transposed output): bias[m] per output ROW instead of bias[n] per column
  // conv
  int Hin, Win, Cin, Hout, Wout, stride, upsample, pad;   // pad: zero rows/cols before the image (1, or 0 for the VAE downsampler)
  int ldx;                // conv: input pixel pitch in elements (Cin, or wider when X is a channel slice of a wider NHWC tensor)
  int kw;                 // conv: filter taps along x, 3 (3 x 3) or 1 (3 x 1: the centre column only; K = 3 Cin)
  int tiles_n, tiles_total;
  int tiles_m, group_m;   // gemm_sp_kernel: tile order (group_m row panels of tiles are walked column by column)
  int reverse;            // gemm_sp_kernel: walk the tile order BACKWARDS (an A operand larger than the memory-side cache, written front to back by the previous kernel)
};

__device__ __attribute__((aligned(64))) half_t g_zero_page[32] = {};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// exact-erf GELU to fp16 accuracy, cheap enough for an epilogue (diffusers GEGLU uses F.gelu(approximate='none')).
//   gelu(x) = x Phi(x),  Phi(x) = 1 - r (x >= 0), r (x < 0),  r = erfc(|x| / sqrt 2) / 2 = 1 / (2 P(|x| / sqrt 2)^16)
// with P = 1 + a1 z + .. + a6 z^6 of Abramowitz-Stegun 7.1.28 (|erf error| <= 3e-7).  The 1/2 and the 1/sqrt 2 are folded into
// the coefficients (c_k = a_k 2^(1/16) 2^(-k/2)), so  gelu(x) = max(x, 0) - |x| / p(|x|)^16 :  6 FMAs, 4 squarings, ONE reciprocal,
// max, FMA -- no exponential, and no cancellation in the negative tail (there r is the result itself).  Measured in fp32 against
// the fp64 erf GELU: |error| <= 1.2e-6 |x|, relative error <= 2.4e-6 for x > 0 (tests/test_gelu_scheme_cpu.py).  Everything but the
// reciprocal and the max is written on float PAIRS so that it compiles to packed v_pk_fma_f32 / v_pk_mul_f32: ~11 issue slots per
// output instead of ~26 for the 7.1.26 form used until round 2 (one v_rcp + one v_exp + 9 scalar FMAs) -- the GEGLU epilogues are
// VALU bound.  The product h * gelu(g) is formed in fp32 and rounded once.
typedef float float2_t __attribute__((ext_vector_type(2)));
#ifdef MD_DEGRADE_GELU_TANH
// DIAGNOSTIC build only (tools/build_ab.sh, never shipped): the tanh approximation of GELU instead of the exact erf form -- a deliberately
// degraded epilogue that the parity budgets of tests/parity_budget.py must catch (profiles/r06_parity_budget_degraded.log)
__device__ __forceinline__ float md_gelu_tanh(float x) {
  const float u = 0.7978845608f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.f + (1.f - 2.f / (__expf(2.f * u) + 1.f)));
}
#endif
__device__ __forceinline__ float2_t gelu_fast2(float2_t x) {
#ifdef MD_DEGRADE_GELU_TANH
  return float2_t{md_gelu_tanh(x.x), md_gelu_tanh(x.y)};
#endif
  const float2_t ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
  float2_t p = float2_t{5.621299664e-06f, 5.621299664e-06f};
  p = p * ax + float2_t{5.105520901e-05f, 5.105520901e-05f};
  p = p * ax + float2_t{3.968613701e-05f, 3.968613701e-05f};
  p = p * ax + float2_t{3.422739239e-03f, 3.422739239e-03f};
  p = p * ax + float2_t{2.207699846e-02f, 2.207699846e-02f};
  p = p * ax + float2_t{5.207516304e-02f, 5.207516304e-02f};
  p = p * ax + float2_t{1.044273782e+00f, 1.044273782e+00f};
  p = p * p;
  p = p * p;
  p = p * p;
  p = p * p;                                       // |x| >~ 21: inf, r = 0, gelu = max(x, 0)
  const float2_t r = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
  const float2_t m = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
  // |x| = inf: p^16 = inf, r = 0 and inf * 0 would be NaN: the product takes |x| clamped to a finite value, so gelu(+inf) = +inf and
  // gelu(-inf) = 0 (the limits); NaN still travels through p and r.  One v_min per value: +2.6 % on the K = 320 GEGLU kernel, +-1 % on the
  // others (profiles/r05_ab_gelu_inf_guard.log).
  const float2_t axc = {fminf(ax.x, 3.0e38f), fminf(ax.y, 3.0e38f)};
  return m - axc * r;
}
// The same arithmetic on N independent pairs, stage by stage (pinned): a kernel with ONE wave per SIMD has nobody to cover the
// 6 + 4 + 1 + 1 dependent packed operations of a single chain (the compiler emits them back to back with s_nop between them).
template <int N>
__device__ __forceinline__ void gelu_fast2_x(float2_t (&x)[N]) {
#ifdef MD_DEGRADE_GELU_TANH
#pragma unroll
  for (int c = 0; c < N; ++c) x[c] = float2_t{md_gelu_tanh(x[c].x), md_gelu_tanh(x[c].y)};
  return;
#endif
  float2_t ax[N], p[N];
#pragma unroll
  for (int c = 0; c < N; ++c) {
    ax[c] = float2_t{__builtin_fabsf(x[c].x), __builtin_fabsf(x[c].y)};
    p[c] = float2_t{5.621299664e-06f, 5.621299664e-06f} * ax[c] + float2_t{5.105520901e-05f, 5.105520901e-05f};
  }
  __builtin_amdgcn_sched_barrier(0);
#define MD_GELU_STAGE(K)                                            \
  _Pragma("unroll") for (int c = 0; c < N; ++c) p[c] = p[c] * ax[c] + float2_t{K, K}; \
  __builtin_amdgcn_sched_barrier(0);
  MD_GELU_STAGE(3.968613701e-05f)
  MD_GELU_STAGE(3.422739239e-03f)
  MD_GELU_STAGE(2.207699846e-02f)
  MD_GELU_STAGE(5.207516304e-02f)
  MD_GELU_STAGE(1.044273782e+00f)
#undef MD_GELU_STAGE
#pragma unroll
  for (int sq = 0; sq < 4; ++sq) {
#pragma unroll
    for (int c = 0; c < N; ++c) p[c] = p[c] * p[c];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < N; ++c) p[c] = float2_t{__builtin_amdgcn_rcpf(p[c].x), __builtin_amdgcn_rcpf(p[c].y)};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < N; ++c) ax[c] = float2_t{fminf(ax[c].x, 3.0e38f), fminf(ax[c].y, 3.0e38f)};      // the |x| = inf guard (gelu_fast2)
#pragma unroll
  for (int c = 0; c < N; ++c) x[c] = float2_t{fmaxf(x[c].x, 0.f), fmaxf(x[c].y, 0.f)} - ax[c] * p[c];
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float2_t g = gelu_fast2(float2_t{x, x});
  return g.x;
}

// GEGLU epilogue of one 32x32 h|g accumulator pair, straight from the accumulators.  With acc = mfma(W frag, A frag) lane
// (m = lane % 32, hi = lane / 32) holds, for its output row m, the columns 8g + 4*hi + {0..3}, g = 0..3: four 8-byte pieces.
// v_permlane32_swap exchanges pieces between the two lane halves so that the low lane owns columns [16p, 16p+8) and the high
// lane [16p+8, 16p+16), p = 0, 1: TWO 16-byte stores per lane instead of four 8-byte ones (a row-per-lane epilogue is
// store-issue bound, not bandwidth bound: guide T21).  `drow` points at column 0 of the 32-column output group in row m.
__device__ __forceinline__ void geglu_store32(const floatx16& ah, const floatx16& ag, const half4_t (&bh)[4], const half4_t (&bg)[4],
                                              half_t* drow, int hi, bool row_ok) {
  unsigned w[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    half4_t o;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      const float2_t gl = gelu_fast2(float2_t{ag[4 * g + e] + (float)bg[g][e], ag[4 * g + e + 1] + (float)bg[g][e + 1]});
      o[e] = (half_t)((ah[4 * g + e] + (float)bh[g][e]) * gl.x);
      o[e + 1] = (half_t)((ah[4 * g + e + 1] + (float)bh[g][e + 1]) * gl.y);
    }
    __builtin_memcpy(w[g], &o, 8);
  }
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    // pieces g = 2pr (X0) and g = 2pr+1 (X1): afterwards d[.][0] = [X0.low | X1.low], d[.][1] = [X0.high | X1.high] per lane half
    unsigned lo[2], hi2[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const auto r = __builtin_amdgcn_permlane32_swap(w[2 * pr][d], w[2 * pr + 1][d], false, false);
      lo[d] = r[0];       // low lanes: own piece 2pr (cols +0..3)      high lanes: low lane's piece 2pr+1 (cols +8..11)
      hi2[d] = r[1];      // low lanes: high lane's piece 2pr (cols +4..7)  high lanes: own piece 2pr+1 (cols +12..15)
    }
    if (row_ok) {
      const uint4 v = {lo[0], lo[1], hi2[0], hi2[1]};
      *reinterpret_cast<uint4*>(drow + 16 * pr + 8 * hi) = v;
    }
  }
}

template <bool CONV, int BK, int NSTAGE, int WM, int WPS, bool GEGLU, int NJ, int MI>
__global__ __launch_bounds__(WM * 128, WPS) void gemm_kernel(GemmParams p) {
  constexpr int WROWS = 32 * MI;           // rows of the output tile owned by one wave (MI 32-row MFMA tiles)
  constexpr int BM = WM * WROWS;
  constexpr int BN = 64 * NJ;              // 2 waves along N, NJ 32-column MFMA tiles each
  constexpr int CS_LD = BN + 4;            // fp32 staging row pitch
  static_assert(!GEGLU || NJ == 2, "GEGLU pairs the two column sub-tiles of a wave");
  constexpr int NW = WM * 2;               // waves
  constexpr int T = NW * 64;               // threads
  (void)T;
  constexpr int ROWB = BK * 2;             // bytes per tile row
  constexpr int SLOTS = BK / 8;            // 16-B slots per row
  constexpr int RPI = 1024 / ROWB;         // rows covered by one wave-wide DMA instruction
  constexpr int IPA = (BM / RPI) / NW;     // DMA instructions per wave per tile, A operand
  constexpr int IPB = (BN / RPI) / NW;     // ... W operand
  constexpr int OPA = BM * ROWB;           // bytes of the A tile
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int G = IPA + IPB;             // DMA instructions per wave per tile
  constexpr int SW_SHIFT = BK == 32 ? 2 : 1;
  constexpr int SW_MASK = SLOTS - 1;
  static_assert(IPA >= 1 && IPB >= 1, "tile too small for the wave count");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Cs = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: workgroup b runs on XCD b%8; give each XCD a contiguous run of tiles (n fastest) so that
  // the A row-panel and the W panel it re-reads stay in that XCD's private L2.
  int bid = blockIdx.x;
  {
    const int nwg = p.tiles_total;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-lane DMA source coordinates
  const int lrow = lane / SLOTS, pslot = lane % SLOTS;
  const half_t* a_src[IPA];
  const half_t* w_src[IPB];
  int a_oy[IPA], a_ox[IPA];
#pragma unroll
  for (int j = 0; j < IPA; ++j) {
    const int row = (wave * IPA + j) * RPI + lrow;
    const int lslot = pslot ^ ((row >> SW_SHIFT) & SW_MASK);
    const int m = m0 + row;
    const int mm = m < p.M ? m : p.M - 1;
    if (CONV) {
      const int hw = p.Hout * p.Wout;
      const int b = mm / hw, rem = mm - b * hw;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      a_oy[j] = oy * p.stride - p.pad;      // input row of filter tap ky = 0 (in the possibly upsampled image)
      a_ox[j] = ox * p.stride - p.pad;
      a_src[j] = p.A + (size_t)b * p.Hin * p.Win * p.ldx + lslot * 8;
    } else {
      a_oy[j] = a_ox[j] = 0;
      a_src[j] = p.A + (size_t)mm * p.lda + lslot * 8;
    }
  }
#pragma unroll
  for (int j = 0; j < IPB; ++j) {
    const int row = (wave * IPB + j) * RPI + lrow;
    const int lslot = pslot ^ ((row >> SW_SHIFT) & SW_MASK);
    const int n = n0 + row;
    w_src[j] = p.W + (size_t)(n < p.N ? n : p.N - 1) * p.K + lslot * 8;
  }
  const half_t* zero_src = g_zero_page + 0;

  auto issue_tile = [&](int kt, int stage) {
    const int k0 = kt * BK;
    char* sa = smem + stage * STAGE + (wave * IPA) * 1024;
    char* sw = smem + stage * STAGE + OPA + (wave * IPB) * 1024;
    if (CONV) {
      const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
      const int ky = p.kw == 3 ? tap / 3 : tap, kx = p.kw == 3 ? tap - ky * 3 : 1;
      const unsigned hup = p.Hin << p.upsample, wup = p.Win << p.upsample;
#pragma unroll
      for (int j = 0; j < IPA; ++j) {
        // branch-free: the element offset is computed for every lane (24-bit multiplies), then the pointer is swapped
        // for the zero page where the tap falls outside the image
        const int iy = a_oy[j] + ky, ix = a_ox[j] + kx;
        const bool ok = (unsigned)iy < hup && (unsigned)ix < wup;
        unsigned off = __umul24(__umul24((unsigned)(iy >> p.upsample), (unsigned)p.Win) + (unsigned)(ix >> p.upsample), (unsigned)p.ldx) + c0;
        asm volatile("" : "+v"(off));
        const half_t* src = a_src[j] + off;
        src = ok ? src : zero_src;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + j * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < IPA; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + k0), (lptr_t)(sa + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < IPB; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + k0), (lptr_t)(sw + j * 1024), 16, 0, 0);
  };

  floatx16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue_tile(s, s);

  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[MI], b_off[NJ], a_sw[MI], b_sw[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ra = wm * WROWS + i * 32 + frow;
    a_off[i] = ra * ROWB;
    a_sw[i] = (ra >> SW_SHIFT) & SW_MASK;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int rb = wn * (32 * NJ) + j * 32 + frow;
    b_off[j] = OPA + rb * ROWB;
    b_sw[j] = (rb >> SW_SHIFT) & SW_MASK;
  }

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most min(NSTAGE-2, tiles issued after kt) x G of this wave's DMAs are outstanding
    const int ahead = nk - 1 - kt;
    if (NSTAGE == 1) {
      // single buffer (occupancy flavour): other workgroups on the CU cover this one's exposed load latency
      if (kt > 0) __builtin_amdgcn_s_barrier();
      issue_tile(kt, 0);
    }
    if (NSTAGE >= 4 && ahead >= 2) wait_vmcnt<2 * G>();
    else if (NSTAGE >= 3 && ahead >= 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // every wave has finished tile kt-1 -> its ring slot can be refilled with tile kt+NSTAGE-1
    if (NSTAGE > 1 && kt + NSTAGE - 1 < nk) {
      int st = stage + NSTAGE - 1;
      if (st >= NSTAGE) st -= NSTAGE;
      issue_tile(kt + NSTAGE - 1, st);
    }
    const char* sb = smem + stage * STAGE;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      half8_t af[MI], bf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const half8_t*>(sb + a_off[i] + (((s * 2 + fhi) ^ a_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const half8_t*>(sb + b_off[j] + (((s * 2 + fhi) ^ b_sw[j]) << 4));
      // the workgroups sharing this CU sit in different phases (DMA issue / LDS reads / epilogue): favour whoever has
      // its operands ready for the matrix pipe (guide T5)
      if (!CONV) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = GEGLU ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
      if (!CONV) __builtin_amdgcn_s_setprio(0);
    }
    if (++stage == NSTAGE) stage = 0;
  }

  if (GEGLU) {
    // GEGLU epilogue straight from the accumulators: with acc = mfma(W frag, A frag) a lane owns row m = ..+lane%32 and
    // columns n = ..+8g+4*(lane/32)+{0..3}, so h and g of an output element sit in the same lane and every store is one
    // 8-byte write (no LDS staging: measured 8 % faster than staging on the K=320 FF GEMMs).
    const int lc = lane & 31, hi = lane >> 5;
    // weight rows were packed as [32 h | 32 g] blocks: sub-tile j = 0 of this wave is h, j = 1 is g of the same columns
    const int Nout = p.N >> 1;
    const int nb = (n0 >> 1) + wn * 32;
    if (n0 + wn * 64 >= p.N) return;
    half4_t bh[4], bg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bh[g] = half4_t{0, 0, 0, 0};
      bg[g] = half4_t{0, 0, 0, 0};
      if (p.bias) {
        bh[g] = *reinterpret_cast<const half4_t*>(p.bias + n0 + wn * 64 + 8 * g + 4 * hi);
        bg[g] = *reinterpret_cast<const half4_t*>(p.bias + n0 + wn * 64 + 32 + 8 * g + 4 * hi);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WROWS + i * 32 + lc;
      const int mc = m < p.M ? m : p.M - 1;
      geglu_store32(acc[i][0], acc[i][NJ - 1], bh, bg, p.C + (size_t)mc * p.ldc + nb, hi, m < p.M);
    }
    (void)Nout;
    return;
  }


  // ---- epilogue: accumulators -> LDS (fp32, 64 rows per pass) -> coalesced rows
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  constexpr int TPR = BN / 8;     // threads per row of the plain epilogue (8 columns each)
  constexpr int RP = T / TPR;     // rows per sweep
  constexpr int NI = 64 / RP;     // sweeps per 64-row pass
#pragma unroll 1
  for (int pass = 0; pass < BM / 64; ++pass) {
    const int mp = m0 + pass * 64;
    // residual rows of this pass (plain epilogue): issue the loads BEFORE the staging barriers so that their HBM
    // latency overlaps the accumulator -> LDS traffic
    half8_t rv[NI];
    bool rvec[NI];
    {
      const int n = n0 + (tid % TPR) * 8;
      const bool nvec = (p.N - n) >= 8;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = mp + (tid / TPR) + RP * i;
        rvec[i] = false;
        if (p.residual && m < p.M && nvec) {
          const half_t* rs = p.residual + (size_t)m * p.ldr + n;
          if ((reinterpret_cast<uintptr_t>(rs) & 15) == 0) {
            rv[i] = *reinterpret_cast<const half8_t*>(rs);
            rvec[i] = true;
          }
        }
      }
    }
    __syncthreads();
    if (wm == (pass * 64) / WROWS) {
      // this wave owns the 64 rows of the pass: its m-tiles i0, i0+1
      const int i0 = ((pass * 64) % WROWS) / 32;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
            const int col = wn * (32 * NJ) + j * 32 + frow;
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
              if (i == i0 + ii) v = acc[i][j][r];     // static register index, uniform select
            Cs[row * CS_LD + col] = v;
          }
    }
    __syncthreads();

    if (p.transpose_out) {
      // out[n][m]: thread owns one column n and 8 consecutive rows -> one 16-B store along m
      const int col = tid % BN;
      const int n = n0 + col;
      if (n < p.N) {
        const float bv = p.bias ? (float)p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 8 / (T / BN); ++i) {
          const int rc = (tid / BN) + (T / BN) * i;
          const int m = mp + rc * 8;
          if (m >= p.M) continue;
          half8_t o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)(Cs[(rc * 8 + j) * CS_LD + col] + bv);
          half_t* dst = p.C + (size_t)n * p.ldc + m;
          if (m + 8 <= p.M && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            *reinterpret_cast<half8_t*>(dst) = o;
          } else {
            for (int j = 0; j < 8 && m + j < p.M; ++j) dst[j] = o[j];
          }
        }
      }
    } else {
      const int col8 = (tid % TPR) * 8;
      const int n = n0 + col8;
      const int nv = (p.N - n) < 8 ? (p.N - n) : 8;   // <= 0: this thread's columns are outside the matrix
      const bool nvec = nv == 8;
      float bsum[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
      if (p.bias && nv > 0) {
        if (nvec && ((reinterpret_cast<uintptr_t>(p.bias + n) & 15) == 0)) {
          const half8_t bv = *reinterpret_cast<const half8_t*>(p.bias + n);
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] = (float)bv[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < nv) bsum[j] = (float)p.bias[n + j];
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = (tid / TPR) + RP * i;
        const int m = mp + row;
        if (m >= p.M || nv <= 0) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Cs[row * CS_LD + col8 + j] + bsum[j];
        if (p.rowadd) {
          const half_t* ra_ = p.rowadd + (size_t)(m / p.rows_per_group) * p.ldra + n;
          if (nvec && ((reinterpret_cast<uintptr_t>(ra_) & 15) == 0)) {
            const half8_t av = *reinterpret_cast<const half8_t*>(ra_);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)av[j];
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < nv) v[j] += (float)ra_[j];
          }
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
        half_t* dst = p.C + (size_t)m * p.ldc + n;
        const bool vec = nvec && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
        if (p.residual) {
          if (rvec[i]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rv[i][j];
          } else {
            const half_t* rs = p.residual + (size_t)m * p.ldr + n;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < nv) v[j] += (float)rs[j];
          }
        }
        if (vec) {
          half8_t o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
          *reinterpret_cast<half8_t*>(dst) = o;
        } else {
          for (int j = 0; j < nv; ++j) dst[j] = (half_t)v[j];
        }
      }
    }
  }
}

// ---- the gemm_kernel flavours in use (the BK = 32 rings and the 8-wave 256-row tile of round 1 measured slower and are gone) ----
//   BK=64 x 2 stages = 64 KiB  -> 2 workgroups/CU, one 32-KiB tile in flight each      (3x3 convs)
//   BK=64 x 1 stage  = 34 KiB  -> 3 workgroups/CU (<= 168 registers), load latency covered by the other workgroups (Linear GEMMs)
//   256x128, BK=64 x 1 stage = 48 KiB, 4 waves x (128x64), 2 workgroups/CU               (large convs / GEGLU / deep-K GEMMs)
template <bool CONV, bool GEGLU, int NJ, int BK, int NSTAGE, int WM = 2, int WPS = 1, int MI = 2>
static void launch_variant(GemmParams& p, hipStream_t stream) {
  constexpr int BN = 64 * NJ;
  constexpr size_t ring = (size_t)NSTAGE * (WM * 32 * MI + BN) * BK * 2;
  constexpr size_t cs = (size_t)64 * (BN + 4) * 4;
  constexpr size_t smem = GEGLU ? ring : (ring > cs ? ring : cs);
  md_ensure_dynamic_lds<gemm_kernel<CONV, BK, NSTAGE, WM, WPS, GEGLU, NJ, MI>>((int)smem);
  p.tiles_n = cdiv(p.N, BN);
  p.tiles_total = cdiv(p.M, WM * 32 * MI) * p.tiles_n;
  hipLaunchKernelGGL((gemm_kernel<CONV, BK, NSTAGE, WM, WPS, GEGLU, NJ, MI>), dim3(p.tiles_total), dim3(WM * 128), smem, stream, p);
}

static int env_int(const char* name, int dflt) { return md_env_int(name, dflt); }


// Streams created with a CU mask (hipExtStreamCreateWithCUMask) own fewer CUs than the device has: the persistent launchers size their
// grids -- and the rounds model its tile choice -- for md_set_cu_limit's count instead (process-wide; 0 = the device's own count).
static std::atomic<int> g_cu_limit{0};

extern "C" int md_set_cu_limit(int ncu) {
  if (ncu < 0 || (ncu & 7)) return MD_ERR_ARG;                    // the XCD-aware tile order needs whole multiples of the 8 XCDs
  g_cu_limit.store(ncu, std::memory_order_relaxed);
  return MD_OK;
}

static int md_device_cus() {
  // CU count of the CURRENT device, cached per device (a process may drive several GPUs)
  const int lim = g_cu_limit.load(std::memory_order_relaxed);
  if (lim > 0) return lim;
  static std::atomic<int> cache[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = cache[dev & 63].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    if (n <= 0 || (n & 7)) n = 256;                               // MI355X: 256 CUs; the tile order needs a multiple of 8
    cache[dev & 63].store(n, std::memory_order_relaxed);
  }
  return n;
}

#include "gemm_ws.h"
#include "gemm_sp.h"

// Row streams of a streaming launch with G = p.groups column groups: spx per XCD on G spx of its 32 CUs, plus the extra streams that the
// 8 (32 - G spx) leftover CUs of the chip can form (gemm_ws.h).  MD_WS_EXTRA = 0 leaves them idle as until round 5 (A/B).
static void ws_streams(WsParams& p) {
  static const int extra = md_env_int("MD_WS_EXTRA", 1);
  p.spx = 32 / p.groups;
  p.xstreams = extra ? 8 * (32 - p.groups * p.spx) / p.groups : 0;
  p.streams = 8 * p.spx + p.xstreams;
}

template <int KS, int CB, int TPR, bool RES, bool RA>
static void launch_ws_variant(const WsParams& p, hipStream_t stream) {
  using Cfg = WsCfg<KS, CB, TPR>;
  md_ensure_dynamic_lds<wsgemm_kernel<KS, CB, TPR, RES, RA>>(Cfg::SMEM);
  hipLaunchKernelGGL((wsgemm_kernel<KS, CB, TPR, RES, RA>), dim3(256), dim3(512), Cfg::SMEM, stream, p);
}

template <int KS, int CB, int TPR>
static void launch_ws_tpr(const WsParams& p, hipStream_t stream) {
  if (p.residual && p.rowadd) launch_ws_variant<KS, CB, TPR, true, true>(p, stream);
  else if (p.residual) launch_ws_variant<KS, CB, TPR, true, false>(p, stream);
  else if (p.rowadd) launch_ws_variant<KS, CB, TPR, false, true>(p, stream);
  else launch_ws_variant<KS, CB, TPR, false, false>(p, stream);
}

template <int KS, int CB>
static void launch_ws(const GemmParams& g, hipStream_t stream) {
  constexpr int GC = 64 * CB;
  WsParams p = {};
  p.A = g.A; p.W = g.W; p.C = g.C; p.bias = g.bias; p.residual = g.residual; p.rowadd = g.rowadd;
  p.lda = g.lda; p.ldc = g.ldc; p.ldr = g.ldr; p.ldra = g.ldra; p.M = g.M; p.N = g.N; p.rows_per_group = g.rows_per_group;
  p.groups = g.N / GC;
  ws_streams(p);
  // one tile per barrier round (deepest DMA ring) when the launch streams A from HBM once and sits on the store path (one column
  // group); two tiles per round when several groups share A through L2 and the tile time is barrier / latency bound
  if (p.groups > 1) launch_ws_tpr<KS, CB, 2>(p, stream);
  else launch_ws_tpr<KS, CB, 1>(p, stream);
}

// GEGLU flavour of the streaming kernel: K = 320, packed N % 256 == 0 (128 output columns per workgroup), long M
static bool ws_geglu_eligible(const GemmParams& p) {
  if (p.act != ACT_GEGLU || p.K != 320 || p.N % 256 || p.N / 256 > 16 || p.M < 32768 || p.M % 16) return false;
  if (p.lda % 8 || p.ldc % 8) return false;
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return al(p.A) && al(p.W) && al(p.C) && al(p.bias);
}

static void launch_ws_geglu(const GemmParams& g, hipStream_t stream) {
  WsParams p = {};
  p.A = g.A; p.W = g.W; p.C = g.C; p.bias = g.bias; p.residual = nullptr; p.rowadd = nullptr;
  p.lda = g.lda; p.ldc = g.ldc; p.ldr = 0; p.ldra = 0; p.M = g.M; p.N = g.N; p.rows_per_group = 1;
  p.groups = g.N / 256;
  ws_streams(p);
  // the GELU arithmetic (shared by the four memory waves), not the barrier rounds, bounds a GEGLU tile: one tile per round
  constexpr int smem = WsCfg<10, 4, 1>::SMEM;
  md_ensure_dynamic_lds<wsgemm_kernel<10, 4, 1, false, false, true>>(smem);
  hipLaunchKernelGGL((wsgemm_kernel<10, 4, 1, false, false, true>), dim3(256), dim3(512), smem, stream, p);
}

// ---- fused-normalisation flavours of the streaming kernel (gemm_ws.h: PRO_LNF / PRO_AFF prologues) -----------------------------------
// K = 320 only (the 96 x 96 level), plain epilogue, N a multiple of 320, M >= 32768 rows in whole 16-row tiles, no residual.  Measured and
// NOT built in (profiles/r05_ab_fused_norms*.log): the K = 640 forms, GEGLU, and row statistics handed from the producer to the consumer.
static bool ws_fused_shape(int M, int N, int K, int lda, int ldc) {
  return K == 320 && N % 320 == 0 && N / 320 <= 8 && M >= 32768 && M % 16 == 0 && lda % 8 == 0 && ldc % 8 == 0;
}
static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int TPR, bool RA, int PRO>
static void launch_ws_fused(const WsParams& p, hipStream_t stream) {
  using Cfg = WsCfg<10, 5, TPR>;
  md_ensure_dynamic_lds<wsgemm_kernel<10, 5, TPR, false, RA, false, PRO>>(Cfg::SMEM);
  hipLaunchKernelGGL((wsgemm_kernel<10, 5, TPR, false, RA, false, PRO>), dim3(256), dim3(512), Cfg::SMEM, stream, p);
}

// epi: bit 0 residual, bit 1 row-broadcast operand (as md_gemm_plan).  1 when md_gemm_ln_f16 has a kernel for the problem (dense operands
// assumed), else 0: the caller then runs md_layernorm_f16 + md_gemm_f16 on the unfolded weights.
extern "C" int md_gemm_ln_plan(int M, int N, int K, int act, int epi) {
  return act == ACT_NONE && !(epi & 1) && ws_fused_shape(M, N, K, K, N);
}

extern "C" int md_gemm_ln_f16(const void* A, int lda, const void* Wf, const float* sc, void* C, int ldc, int M, int N, int K, float eps,
                              const void* rowadd, int ldra, int rows_per_group, int act, void* stream) {
  MD_CHECK_ARG(act == ACT_NONE && ws_fused_shape(M, N, K, lda, ldc),
               "md_gemm_ln: no fused LayerNorm kernel for M=%d N=%d K=%d act=%d (ask md_gemm_ln_plan first)", M, N, K, act);
  MD_CHECK_ARG(al16(A) && al16(Wf) && al16(C) && al16(sc) && al16(rowadd) && (!rowadd || (ldra % 8 == 0 && rows_per_group > 0)),
               "md_gemm_ln: operands must be 16-byte aligned (ldra %% 8 == 0, rows_per_group > 0 with rowadd)");
  WsParams p = {};
  p.A = (const half_t*)A; p.W = (const half_t*)Wf; p.C = (half_t*)C; p.rowadd = (const half_t*)rowadd;
  p.lda = lda; p.ldc = ldc; p.ldra = ldra; p.M = M; p.N = N; p.rows_per_group = rowadd ? rows_per_group : 1;
  p.lnf = sc; p.eps = eps;
  p.groups = N / 320; ws_streams(p);
  hipStream_t st = (hipStream_t)stream;
  if (p.groups > 1) {
    if (rowadd) launch_ws_fused<2, true, PRO_LNF>(p, st);
    else launch_ws_fused<2, false, PRO_LNF>(p, st);
  } else {
    if (rowadd) launch_ws_fused<1, true, PRO_LNF>(p, st);
    else launch_ws_fused<1, false, PRO_LNF>(p, st);
  }
  MD_CHECK_LAUNCH("md_gemm_ln");
  return MD_OK;
}

extern "C" int md_gemm_affine_plan(int M, int N, int K, int rows_per_image) {
  return rows_per_image > 0 && rows_per_image % 16 == 0 && M % rows_per_image == 0 && ws_fused_shape(M, N, K, K, N);
}

extern "C" int md_gemm_affine_f16(const void* A, int lda, const float* table, int rows_per_image, const void* W, void* C, int ldc, int M, int N,
                                  int K, const void* bias, void* stream) {
  MD_CHECK_ARG(rows_per_image > 0 && rows_per_image % 16 == 0 && M % rows_per_image == 0 && ws_fused_shape(M, N, K, lda, ldc),
               "md_gemm_affine: no fused kernel for M=%d N=%d K=%d rows_per_image=%d (ask md_gemm_affine_plan first)", M, N, K, rows_per_image);
  MD_CHECK_ARG(al16(A) && al16(W) && al16(C) && al16(table) && al16(bias), "md_gemm_affine: operands must be 16-byte aligned");
  MD_CHECK_ARG(A != C, "md_gemm_affine: in place is not supported");
  WsParams p = {};
  p.A = (const half_t*)A; p.W = (const half_t*)W; p.C = (half_t*)C; p.bias = (const half_t*)bias;
  p.lda = lda; p.ldc = ldc; p.M = M; p.N = N; p.rows_per_group = 1;
  p.aff = table; p.rows_per_image = rows_per_image;
  p.groups = N / 320; ws_streams(p);
  hipStream_t st = (hipStream_t)stream;
  if (p.groups > 1) launch_ws_fused<2, false, PRO_AFF>(p, st);
  else launch_ws_fused<1, false, PRO_AFF>(p, st);
  MD_CHECK_LAUNCH("md_gemm_affine");
  return MD_OK;
}

// W-stationary streaming kernel (gemm_ws.h): plain epilogues, K = 320 (N % 320 == 0) or K = 640 (N % 128 == 0), long M.
static bool ws_eligible(const GemmParams& p) {
  if (p.act != ACT_NONE || p.transpose_out || p.M < 32768 || p.M % 16) return false;
  const bool k320 = p.K == 320 && p.N % 320 == 0 && p.N / 320 <= 8;
  const bool k640 = p.K == 640 && p.N % 128 == 0 && p.N / 128 <= 16;
  if (!k320 && !k640) return false;
  if (p.lda % 8 || p.ldc % 8 || (p.residual && p.ldr % 8) || (p.rowadd && p.ldra % 8)) return false;
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return al(p.A) && al(p.W) && al(p.C) && al(p.bias) && al(p.residual) && al(p.rowadd);
}

// Plan codes (returned by dispatch_any; md_gemm_plan / md_conv3x3_plan expose them so that the table is testable without a GPU):
//   1MN  gemm_sp_kernel with wave tile (MT, NT) = (M, N): 135 = 192 x 320, 134 = 192 x 256, 124 = 128 x 256, 144 = 256 x 256 GEGLU;
//        +1000 when it runs on swapped operands (transposed output); +2000 when the residual enters through the matrix core (RESM: K tiles >
//        sub-tiles of the wave tile; gemm_sp.h)
//   210 / 220 / 230  wsgemm_kernel K = 320 / K = 640 / GEGLU
//   301 / 302 / 303  gemm_kernel 64-column tiles / 256 x 128 / 128 x 128
template <bool CONV, bool GEGLU, bool DRY>
static int dispatch_any(GemmParams& p, hipStream_t stream, const int sp, const int force_nt, const int ncu) {
  // Dispatch table, from same-box A/B runs on MI355X (profiles/r0*_ab_*.log).  One knob survives, used by the
  // parity tests: MD_GEMM_SP = 0 off | 1 every eligible problem | 2 automatic (default).  (The two-waves-per-SIMD ping-pong
  // kernels of rounds 1-2, gemm_pp.h, lost every shape they used to win to gemm_sp_kernel and were removed in round 3.)
  // gemm_sp_kernel's tile: 256 x 256 for GEGLU; 192 x 320, 192 x 256 or 128 x 256 otherwise, whichever needs least time by the model
  // rounds x (T0 + K tiles x t_k): rounds = ceil(tiles / CUs) of the persistent grid, T0 ~ 4 us per output tile outside its K loop,
  // t_k = 1.56 / 1.28 / 0.95 us per 64-deep K tile (15 / 12 / 8 MFMAs per k-step; profiles/r03_ab_gemm_sp_tiles.log).
  // N = 1280 on M = 18 432 tokens: 384 tiles of 192 x 320 are 1.5 rounds (2 paid), 480 tiles of 192 x 256 are 1.9;
  // on M = 4608 (the 12 x 12 level) 120 tiles of 192 x 256 leave half of the CUs idle, 180 tiles of 128 x 256 less than a third.
  // Transposed output (V^T of the attention kernels, bias only): the same kernel on SWAPPED operands -- C^T[N][M] = W[N][K] . A[M][K]^T
  // is a plain GEMM whose "A" is the weight, whose "W" is the token matrix (row pitch K required) and whose bias runs along the
  // output rows; M must be a multiple of 256 (it is the swapped problem's N).
  if constexpr (!CONV && !GEGLU) {
    if (p.transpose_out && sp > 0 && p.lda == p.K && p.act == ACT_NONE && !p.residual && !p.rowadd) {
      GemmParams q = p;
      q.A = p.W; q.W = p.A; q.lda = p.K; q.M = p.N; q.N = p.M; q.transpose_out = 0; q.bias_rows = 1;
      if (sp_eligible<false, false, 4>(q)) {
        auto cost = [&](int bm, double tk) {
          const long tiles = (long)cdiv(q.M, bm) * (q.N / 256);
          return (double)cdiv(tiles, ncu) * (4.0 + (q.K / 64) * tk);
        };
        const bool small = force_nt == 2 || (force_nt != 4 && cost(128, 0.95) < cost(192, 1.28));
        const long tiles = (long)cdiv(q.M, small ? 128 : 192) * (q.N / 256);
        if (sp == 1 || (tiles >= 112 && q.K >= 256)) {
          if constexpr (!DRY) {
            if (small) launch_sp<false, false, 4, 2>(q, stream);
            else launch_sp<false, false, 4>(q, stream);
          }
          return small ? 1124 : 1134;
        }
      }
    }
  }
  int nt = GEGLU ? 4 : 0;                                         // 5, 4: 192-row tiles; 2: 128 x 256
  if constexpr (!GEGLU) {
    auto cost = [&](int bm, int bn, double tk) {
      const long tiles = (long)cdiv(p.M, bm) * (p.N / bn);
      return (double)cdiv(tiles, ncu) * (4.0 + (p.K / 64) * tk);
    };
    const bool ok5 = (force_nt == 0 || force_nt == 5) && sp_eligible<CONV, false, 5>(p);
    const bool ok4 = sp_eligible<CONV, false, 4>(p);
    const double c5 = ok5 ? cost(192, 320, 1.56) : 1e30, c4 = ok4 && (force_nt == 0 || force_nt == 4) ? cost(192, 256, 1.28) : 1e30,
                 c2 = ok4 && (force_nt == 0 || force_nt == 2) ? cost(128, 256, 0.95) : 1e30;
    if (c5 < 1e30 || c4 < 1e30 || c2 < 1e30) nt = c5 <= c4 && c5 <= c2 ? 5 : (c4 <= c2 ? 4 : 2);
    const bool ok128 = sp_eligible<CONV, false, 2>(p);
    // N a multiple of 128 only (the 128-channel convs of the AutoencoderKL at full resolution): the 256 x 128 tile, wave tile 128 x 64
    const double c42 = ok128 && (force_nt == 0 || force_nt == 42) ? cost(256, 128, 0.95) : 1e30;
    if (nt == 0 && c42 < 1e30) nt = 42;
    // 192 x 128 (wave tile 96 x 64, 6 MFMAs per k-step): more and smaller tiles for the launches that cannot fill the chip once with
    // the larger ones -- the 12 x 12 level: M = 4608, N = 1280 gives 24 x 10 = 240 tiles on 256 CUs where 128 x 256 gives 180.
    static const double tk32 = env_int("MD_GEMM_SP_TK32", 800) * 1e-3;
    if (ok128 && (force_nt == 0 || force_nt == 32)) {
      const double best = std::min(std::min(c5, c42), std::min(c4, c2));
      if (force_nt == 32 || cost(192, 128, tk32) < best) nt = 32;
    }
  } else if (!sp_eligible<CONV, true>(p)) {
    nt = 0;
  }
  auto run_sp = [&]() {
    if constexpr (!DRY) {
      if constexpr (GEGLU) launch_sp<CONV, true>(p, stream);
      else if (nt == 4) launch_sp<CONV, false, 4>(p, stream);
      else if (nt == 2) launch_sp<CONV, false, 4, 2>(p, stream);
      else if (nt == 42) launch_sp<CONV, false, 2, 4>(p, stream);
      else if (nt == 32) launch_sp<CONV, false, 2, 3>(p, stream);
      else launch_sp<CONV, false, 5>(p, stream);
    }
    if constexpr (GEGLU) return 144;
    const bool resm = nt == 4 ? sp_resm<3, 4>(p) : (nt == 2 ? sp_resm<2, 4>(p) : (nt == 42 ? sp_resm<4, 2>(p) : (nt == 32 ? sp_resm<3, 2>(p) : sp_resm<3, 5>(p))));
    return (resm ? 2000 : 0) + (nt == 4 ? 134 : (nt == 2 ? 124 : (nt == 42 ? 142 : (nt == 32 ? 132 : 135))));
  };
  if (sp == 1 && nt) return run_sp();
  // 1. HBM-bound short-K projections on long token matrices: W-stationary streaming kernel (gemm_ws.h), plain and GEGLU (K = 320)
  if constexpr (!CONV && !GEGLU) {
    if (ws_eligible(p)) {
      if constexpr (!DRY) {
        if (p.K == 320) launch_ws<10, 5>(p, stream);
        else launch_ws<20, 2>(p, stream);
      }
      return p.K == 320 ? 210 : 220;
    }
  }
  if constexpr (!CONV && GEGLU) {
    if (ws_geglu_eligible(p)) {
      if constexpr (!DRY) launch_ws_geglu(p, stream);
      return 230;
    }
  }
  // 2. one-wave-per-SIMD flavour (gemm_sp.h), same-box table in profiles/r03_ab_gemm_sp_tiles.log: every 3x3 conv and every plain
  //    GEMM with K >= 640 that gives it at least 112 tiles (with the 192 x 256 tile the 12 x 12 level's 120 tiles run +15..25 % over
  //    the 128 x 128 kernel, M = 4608 GEMMs +1..18 %, M = 18 432 x N = 1280 +19..34 %); GEGLU GEMMs with K >= 640 (+24..29 %; at
  //    K = 320 the W-stationary kernel above is 9 % faster)
  if (sp > 0 && nt) {
    const long tiles = (long)cdiv(p.M, GEGLU || nt == 42 ? 256 : (nt == 2 ? 128 : 192)) * (p.N / (nt == 5 ? 320 : (nt == 42 || nt == 32 ? 128 : 256)));
    const bool pick = GEGLU ? p.K >= 640 : (tiles >= 112 && (CONV || p.K >= 640));
    if (pick) return run_sp();
  }
  // 3. the occupancy flavours of gemm_kernel
  if constexpr (!GEGLU) {
    if (p.N <= 64) {                                              // 64-column tiles: conv_out (N = 4), MAN's first conv
      if constexpr (!DRY) {
        if (CONV) launch_variant<CONV, false, 1, 64, 2>(p, stream);
        else launch_variant<CONV, false, 1, 64, 1, 2, 3>(p, stream);
      }
      return 301;
    }
  }
  // 256x128 tile, 4 waves x (128x64 per wave = 4x2 MFMA tiles, 8 independent accumulators), single 48-KiB stage, 2
  // workgroups/CU: 0.75 LDS fragment reads and 0.75x the DMA bytes per MFMA of the 128x128 tile.  Same-box A/B on MI355X:
  // +5..11 % on the 3x3 convs with >= 1024 such tiles and on every GEGLU GEMM, but slower on the HBM-bound skinny Linear GEMMs
  // and on the 24x24 / 12x12 convs (too few tiles to fill 256 CUs twice); plain Linear GEMMs only with a deep K loop
  const long tiles256 = (long)cdiv(p.M, 256) * cdiv(p.N, 128);
  if (tiles256 >= 1024 && (CONV || GEGLU || p.K >= 2048)) {
    if constexpr (!DRY) launch_variant<CONV, GEGLU, 2, 64, 1, 2, 2, 4>(p, stream);
    return 302;
  }
  // 128x128: Linear GEMMs single 32-KiB stage at 3 workgroups/CU (occupancy hides the DMA latency), 3x3 convs a 2-deep ring
  if constexpr (!DRY) {
    if (CONV) launch_variant<CONV, GEGLU, 2, 64, 2>(p, stream);
    else launch_variant<CONV, GEGLU, 2, 64, 1, 2, 3>(p, stream);
  }
  return 303;
}

// gemm_sp_kernel addresses A through a buffer descriptor with 32-bit byte offsets (< 2^31).  A token matrix beyond that (configs[4]:
// 983 040 tokens x 1280 channels = 2.5 GB) is cut into row blocks that each fit: same kernels, same results (rows are independent).
// Returns the number of row blocks (1 = no split) for a plain GEMM that the sp kernel would otherwise have to turn down.
static int sp_row_blocks(const GemmParams& p, int* rows_per_block) {
  const unsigned long long a_bytes = ((unsigned long long)(p.M - 1) * p.lda + p.K) * 2, lim = 1ull << 31;
  *rows_per_block = p.M;
  if (a_bytes < lim || p.transpose_out || p.rowadd || p.K < 640) return 1;
  int rows = (int)((lim - (unsigned long long)p.K * 2) / ((unsigned long long)p.lda * 2));
  rows -= rows % 3840;                             // whole tiles of every sp shape (192, 128, 256 rows) and of the 16-row streams
  if (rows <= 0) return 1;
  *rows_per_block = rows;
  return cdiv(p.M, rows);
}

// Row blocks are only worth their extra launches when the blocks really run on the sp kernel: the first (full-size) block is
// dispatched DRY first; anything but a 1xx plan (N not a multiple of 256 / 320 / 128, too few tiles) leaves the GEMM whole, on the
// multi-workgroup kernel that addresses A with 64-bit pointers.
template <bool GEGLU>
static int sp_row_block_plan(const GemmParams& p, int sp, int force_nt, int ncu, int* rows) {
  const int nb = sp > 0 ? sp_row_blocks(p, rows) : 1;
  if (nb <= 1) return 1;
  GemmParams c = p;
  c.M = *rows;
  const int plan = dispatch_any<false, GEGLU, true>(c, nullptr, sp, force_nt, ncu);
  return plan % 1000 / 100 == 1 ? nb : 1;      // 1xx, with or without the +2000 of the residual-through-the-matrix-core flavour
}

template <bool CONV, bool GEGLU>
static void launch_any(GemmParams& p, hipStream_t stream) {
  static const int sp = env_int("MD_GEMM_SP", 2);
  static const int force_nt = env_int("MD_GEMM_SP_NT", 0);        // A/B runs only: 5 / 4 / 2 / 32 / 42 pin 192x320 / 192x256 / 128x256 / 192x128 / 256x128
  if constexpr (!CONV) {
    int rows;
    const int nb = sp_row_block_plan<GEGLU>(p, sp, force_nt, md_device_cus(), &rows);
    if (nb > 1) {
      for (int b = 0; b < nb; ++b) {
        GemmParams c = p;
        const size_t off = (size_t)b * rows;
        c.A = p.A + off * p.lda;
        c.C = p.C + off * p.ldc;
        if (p.residual) c.residual = p.residual + off * p.ldr;
        c.M = b + 1 < nb ? rows : p.M - (int)off;
        dispatch_any<false, GEGLU, false>(c, stream, sp, force_nt, md_device_cus());
      }
      return;
    }
  }
  dispatch_any<CONV, GEGLU, false>(p, stream, sp, force_nt, md_device_cus());
}


static int launch_gemm(GemmParams& p, bool conv, hipStream_t stream) {
  MD_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "md_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  MD_CHECK_ARG(p.K % 64 == 0, "md_gemm: K=%d must be a multiple of 64 (pad channels when packing)", p.K);
  MD_CHECK_ARG((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0, "md_gemm: A/W must be 16-byte aligned");
  MD_CHECK_ARG(conv || p.lda % 8 == 0, "md_gemm: lda=%d must be a multiple of 8", p.lda);
  if (p.act == ACT_GEGLU) {
    MD_CHECK_ARG(p.N % 64 == 0 && !p.transpose_out && !p.residual && !p.rowadd && p.ldc % 8 == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0,
                 "md_gemm: GEGLU needs N %% 64 == 0 (N=%d), ldc %% 8 == 0, a 16-byte aligned output and no residual/rowadd/transpose", p.N);
  }
  if (p.transpose_out) MD_CHECK_ARG(!p.residual && !p.rowadd && p.act == ACT_NONE, "md_gemm: transposed store supports bias only");
  if (p.rowadd) MD_CHECK_ARG(p.rows_per_group > 0, "md_gemm: rows_per_group must be > 0 with rowadd");
  if (p.residual && p.residual != p.C) {
    // in-place residual is part of the contract (header: Aliasing); anything else that overlaps the output is not
    const uintptr_t c0 = reinterpret_cast<uintptr_t>(p.C), r0 = reinterpret_cast<uintptr_t>(p.residual);
    const size_t cbytes = ((size_t)(p.M - 1) * p.ldc + p.N) * 2, rbytes = ((size_t)(p.M - 1) * p.ldr + p.N) * 2;
    bool disjoint = r0 + rbytes <= c0 || c0 + cbytes <= r0;
    if (!disjoint && p.ldr == p.ldc && ((c0 > r0 ? c0 - r0 : r0 - c0) & 1) == 0) {
      // column-sliced siblings of ONE wider row-major buffer (residual = buf[:, :N], out = buf[:, N:]): the bounding ranges
      // interleave, but no element is shared when the two column intervals stay apart within the common row pitch
      const size_t ld = (size_t)p.ldc;
      const size_t d = ((c0 > r0 ? c0 - r0 : r0 - c0) / 2) % ld;
      disjoint = d >= (size_t)p.N && d + (size_t)p.N <= ld;
    }
    MD_CHECK_ARG(p.transpose_out || disjoint, "md_gemm: residual partially overlaps the output");
  } else if (p.residual) {
    MD_CHECK_ARG(p.ldr == p.ldc, "md_gemm: in-place residual needs ldr == ldc");
  }
  if (conv)
    launch_any<true, false>(p, stream);
  else if (p.act == ACT_GEGLU)
    launch_any<false, true>(p, stream);
  else
    launch_any<false, false>(p, stream);
  MD_CHECK_LAUNCH("md_gemm");
  return MD_OK;
}

extern "C" int md_gemm_f16(const void* A, int lda, const void* W, void* C, int ldc, int M, int N, int K, const void* bias, const void* residual,
                           int ldr, const void* rowadd, int ldra, int rows_per_group, int act, int transpose_out, void* stream) {
  GemmParams p = {};
  p.A = (const half_t*)A; p.W = (const half_t*)W; p.C = (half_t*)C;
  p.bias = (const half_t*)bias; p.residual = (const half_t*)residual; p.rowadd = (const half_t*)rowadd;
  p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldra = ldra;
  p.M = M; p.N = N; p.K = K; p.rows_per_group = rows_per_group; p.act = act; p.transpose_out = transpose_out; p.bias_rows = 0;
  return launch_gemm(p, false, (hipStream_t)stream);
}

static int conv_common(const void* X, int ldx, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout, int kw, int stride,
                       int upsample, int pad_lo, const void* bias, const void* residual, int ldr, const void* rowadd, int ldra,
                       int rows_per_group, int act, void* stream) {
  MD_CHECK_ARG(Cin % 64 == 0, "md_conv3x3: Cin=%d must be a multiple of 64 (zero-pad channels when packing)", Cin);
  MD_CHECK_ARG(ldx >= Cin && ldx % 8 == 0, "md_conv3x3: ldx=%d must be a multiple of 8 and >= Cin=%d", ldx, Cin);
  MD_CHECK_ARG(stride == 1 || stride == 2, "md_conv3x3: stride must be 1 or 2");
  MD_CHECK_ARG(upsample == 0 || (upsample == 1 && stride == 1), "md_conv3x3: upsample is 0 or 1 (nearest 2x) with stride 1");
  MD_CHECK_ARG(pad_lo == 1 || (pad_lo == 0 && stride == 2 && upsample == 0), "md_conv3x3: pad_lo is 1, or 0 with stride 2 (pad (0,1,0,1))");
  MD_CHECK_ARG(kw == 3 || (kw == 1 && stride == 1 && upsample == 0 && pad_lo == 1), "md_conv: kw is 3, or 1 (3 x 1 filter) with stride 1, no upsample, pad 1");
  // the A gather computes (iy * Win + ix) * ldx with 24-bit multiplies into a 32-bit element offset (per image)
  MD_CHECK_ARG((long)Hin * Win < (1L << 24) && ldx < (1 << 24) && (long)Hin * Win * ldx < (1L << 32),
               "md_conv3x3: image %dx%dx%d exceeds the tap arithmetic (Hin*Win < 2^24 pixels, Hin*Win*ldx < 2^32 elements)", Hin, Win, ldx);
  GemmParams p = {};
  p.A = (const half_t*)X; p.W = (const half_t*)W; p.C = (half_t*)Y;
  p.bias = (const half_t*)bias; p.residual = (const half_t*)residual; p.rowadd = (const half_t*)rowadd;
  p.ldc = ldy; p.ldr = ldr; p.ldra = ldra; p.lda = Cin;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.stride = stride; p.upsample = upsample; p.pad = pad_lo; p.ldx = ldx; p.kw = kw;
  const int hup = Hin << upsample, wup = Win << upsample;
  p.Hout = (hup + pad_lo + 1 - 3) / stride + 1;
  p.Wout = kw == 3 ? (wup + pad_lo + 1 - 3) / stride + 1 : Win;
  MD_CHECK_ARG(p.Hout > 0 && p.Wout > 0, "md_conv3x3: empty output %dx%d", p.Hout, p.Wout);
  p.M = B * p.Hout * p.Wout; p.N = Cout; p.K = 3 * kw * Cin;
  p.rows_per_group = rows_per_group; p.act = act; p.transpose_out = 0;
  return launch_gemm(p, true, (hipStream_t)stream);
}

extern "C" int md_conv3x3_nhwc_f16(const void* X, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout, int stride,
                                   int upsample, const void* bias, const void* residual, int ldr, const void* rowadd, int ldra,
                                   int rows_per_group, int act, void* stream) {
  return conv_common(X, Cin, W, Y, ldy, B, Hin, Win, Cin, Cout, 3, stride, upsample, 1, bias, residual, ldr, rowadd, ldra, rows_per_group, act, stream);
}

extern "C" int md_conv3x3_pad_nhwc_f16(const void* X, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout,
                                       int stride, int upsample, int pad_lo, const void* bias, const void* residual, int ldr,
                                       const void* rowadd, int ldra, int rows_per_group, int act, void* stream) {
  return conv_common(X, Cin, W, Y, ldy, B, Hin, Win, Cin, Cout, 3, stride, upsample, pad_lo, bias, residual, ldr, rowadd, ldra, rows_per_group, act, stream);
}

extern "C" int md_conv_nhwc_f16(const void* X, int ldx, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout, int kw, int stride,
                                int upsample, int pad_lo, const void* bias, const void* residual, int ldr, const void* rowadd, int ldra,
                                int rows_per_group, int act, void* stream) {
  return conv_common(X, ldx, W, Y, ldy, B, Hin, Win, Cin, Cout, kw, stride, upsample, pad_lo, bias, residual, ldr, rowadd, ldra, rows_per_group, act, stream);
}

// ---------------------------------------------------------------------------------------------------------------- dispatch queries
// Which kernel the AUTOMATIC dispatch (MD_GEMM_SP = 2, no pinned tile) picks for a problem on a chip with `ncu` compute units; nothing
// is launched and no device is touched, so the table is pinned by CPU tests (tests/test_host_cpu.py).  epi: bit 0 residual, bit 1
// row-broadcast operand, bit 2 bias.  Operands are taken as 16-byte aligned with dense rows (lda = K, ldc = N or M).
static const half_t* plan_ptr(int which) { return reinterpret_cast<const half_t*>((uintptr_t)0x100000 * (which + 1)); }

extern "C" int md_gemm_plan(int M, int N, int K, int act, int transpose_out, int epi, int ncu) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 64 || ncu <= 0) return MD_ERR_ARG;
  GemmParams p = {};
  p.A = plan_ptr(0); p.W = plan_ptr(1); p.C = const_cast<half_t*>(plan_ptr(2));
  p.residual = (epi & 1) ? plan_ptr(3) : nullptr; p.rowadd = (epi & 2) ? plan_ptr(4) : nullptr; p.bias = (epi & 4) ? plan_ptr(5) : nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = transpose_out ? M : (act == ACT_GEGLU ? N / 2 : N); p.ldr = N; p.ldra = N;
  p.rows_per_group = M; p.act = act; p.transpose_out = transpose_out;
  // a token matrix beyond 2^31 bytes runs in row blocks when they take the sp kernel: the plan returned is that of the FIRST (full-size)
  // block; the shorter last block is dispatched on its own (same rule, possibly a smaller tile) -- query it with its own M
  int rows;
  if (act == ACT_GEGLU) {
    if (sp_row_block_plan<true>(p, 2, 0, ncu, &rows) > 1) p.M = rows;
    return dispatch_any<false, true, true>(p, nullptr, 2, 0, ncu);
  }
  if (sp_row_block_plan<false>(p, 2, 0, ncu, &rows) > 1) p.M = rows;
  return dispatch_any<false, false, true>(p, nullptr, 2, 0, ncu);
}

extern "C" int md_conv3x3_plan(int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample, int epi, int ncu) {
  if (B <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cin % 64 || Cout <= 0 || ncu <= 0 || (stride != 1 && stride != 2) || (upsample && stride != 1)) return MD_ERR_ARG;
  GemmParams p = {};
  p.A = plan_ptr(0); p.W = plan_ptr(1); p.C = const_cast<half_t*>(plan_ptr(2));
  p.residual = (epi & 1) ? plan_ptr(3) : nullptr; p.rowadd = (epi & 2) ? plan_ptr(4) : nullptr; p.bias = (epi & 4) ? plan_ptr(5) : nullptr;
  p.ldc = Cout; p.ldr = Cout; p.ldra = Cout; p.lda = Cin;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.stride = stride; p.upsample = upsample; p.pad = 1; p.ldx = Cin; p.kw = 3;
  p.Hout = ((Hin << upsample) + 2 - 3) / stride + 1;
  p.Wout = ((Win << upsample) + 2 - 3) / stride + 1;
  p.M = B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin; p.rows_per_group = p.Hout * p.Wout; p.act = ACT_NONE;
  return dispatch_any<true, false, true>(p, nullptr, 2, 0, ncu);
}
