// MFMA GEMM / implicit-GEMM 3x3 NHWC convolution for gfx950.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )            fp16 in, fp32 accumulate (v_mfma_f32_32x32x16_f16)
//
// One kernel, two A-operand address generators:
//   PLAIN : A is a row-major [M][lda] fp16 matrix (Linear / 1x1 conv on NHWC tokens).
//   CONV3 : A is an NHWC fp16 image batch; row m = (b, oy, ox), column k = (ky*3+kx)*Cin + c; zero padding 1,
//           stride 1 or 2, optional nearest-2x upsample folded into the input addressing.  Cin % 64 == 0 so every
//           64-wide K chunk lies inside a single filter tap.
// Replaces the ATen conv2d / linear calls behind InflatedConv3d (reference src/models/resnet.py:9-17), diffusers
// ResnetBlock2D/Downsample2D/Upsample2D convs, Attention.to_q/k/v/to_out, FeedForward and the 1x1 proj_in/proj_out
// of Transformer2D/3DModel.
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA 32x32 tiles).
// A and W tiles are register-staged (global -> VGPR -> LDS, 16 B per lane, next tile's loads issued before the
// current tile's MFMAs) into a 2-deep LDS ring with one barrier per K step.  LDS rows are 128 B with the 16-B slot
// index XOR-swizzled by (row>>1)&7 so that ds_read_b128 fragment reads are bank-conflict free.
// The fp32 accumulators are staged through LDS in the epilogue so that bias / SiLU / ReLU / GEGLU / row-broadcast
// (time embedding) / residual are applied on full 16-byte coalesced rows, or stored transposed (V^T for attention).
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define CS_LD 132  // fp32 staging row pitch

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_GEGLU = 3 };

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
  // conv
  int Hin, Win, Cin, Hout, Wout, stride, upsample;
  int tiles_n, tiles_total;
};

__device__ __forceinline__ int swz_off(int row, int slot) { return row * (BK * 2) + ((slot ^ ((row >> 1) & 7)) << 4); }

template <bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                       // 2 stages x 128 rows x 128 B
  char* Bs = smem + 2 * BM * BK * 2;     // 2 stages x 128 rows x 128 B
  float* Cs = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
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

  // ---- per-thread load coordinates: 4 A chunks + 4 W chunks of 16 B
  const int slot = tid & 7;
  const int lrow = tid >> 3;  // 0..31
  const half_t* a_ptr[4];
  bool a_ok[4];
  int a_oy[4], a_ox[4];
  const half_t* w_ptr[4];
  bool w_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = lrow + 32 * i;
    const int m = m0 + row;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    if (CONV) {
      const int hw = p.Hout * p.Wout;
      const int b = mm / hw, rem = mm - b * hw;
      a_oy[i] = rem / p.Wout;
      a_ox[i] = rem - a_oy[i] * p.Wout;
      a_ptr[i] = p.A + (size_t)b * p.Hin * p.Win * p.Cin + slot * 8;
    } else {
      a_oy[i] = a_ox[i] = 0;
      a_ptr[i] = p.A + (size_t)mm * p.lda + slot * 8;
    }
    const int n = n0 + row;
    w_ok[i] = n < p.N;
    w_ptr[i] = p.W + (size_t)(w_ok[i] ? n : 0) * p.K + slot * 8;
  }

  half8_t ra[4], rb[4];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    if (CONV) {
      const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int hup = p.Hin << p.upsample, wup = p.Win << p.upsample;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = a_oy[i] * p.stride + ky - 1, ix = a_ox[i] * p.stride + kx - 1;
        const bool ok = a_ok[i] && iy >= 0 && iy < hup && ix >= 0 && ix < wup;
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) v = *reinterpret_cast<const half8_t*>(a_ptr[i] + ((size_t)(iy >> p.upsample) * p.Win + (ix >> p.upsample)) * p.Cin + c0);
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (a_ok[i]) v = *reinterpret_cast<const half8_t*>(a_ptr[i] + k0);
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (w_ok[i]) v = *reinterpret_cast<const half8_t*>(w_ptr[i] + k0);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int stage) {
    char* as = As + stage * (BM * BK * 2);
    char* bs = Bs + stage * (BN * BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<half8_t*>(as + swz_off(row, slot)) = ra[i];
      *reinterpret_cast<half8_t*>(bs + swz_off(row, slot)) = rb[i];
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frow = lane & 31, fhi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* as = As + stage * (BM * BK * 2);
    const char* bs = Bs + stage * (BN * BK * 2);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      half8_t af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const half8_t*>(as + swz_off(wm * 64 + i * 32 + frow, s * 2 + fhi));
        bf[i] = *reinterpret_cast<const half8_t*>(bs + swz_off(wn * 64 + i * 32 + frow, s * 2 + fhi));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(stage ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (fp32) -> coalesced rows
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
        const int col = wn * 64 + j * 32 + frow;
        Cs[row * CS_LD + col] = acc[i][j][r];
      }
  __syncthreads();

  if (p.transpose_out) {
    // out[n][m]: thread owns one column n and 8 consecutive rows -> one 16-B store along m
    const int col = tid & 127;
    const int n = n0 + col;
    if (n < p.N) {
      const float bv = p.bias ? (float)p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rc = (tid >> 7) + 2 * i;
        const int m = m0 + rc * 8;
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
    return;
  }

  if (p.act == ACT_GEGLU) {
    // weight rows were packed as [32 h | 32 g] blocks; the tile holds 64 output columns
    const int oc8 = (tid & 7) * 8;
    const int hcol = (oc8 >> 5) * 64 + (oc8 & 31), gcol = hcol + 32;
    const int nout0 = (n0 >> 1) + oc8;
    const int Nout = p.N >> 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int m = m0 + row;
      if (m >= p.M || nout0 >= Nout) continue;
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float h = Cs[row * CS_LD + hcol + j], g = Cs[row * CS_LD + gcol + j];
        if (p.bias) {
          h += (float)p.bias[n0 + hcol + j];
          g += (float)p.bias[n0 + gcol + j];
        }
        o[j] = (half_t)(h * gelu_erf_f(g));
      }
      *reinterpret_cast<half8_t*>(p.C + (size_t)m * p.ldc + nout0) = o;
    }
    return;
  }

  const int col8 = (tid & 15) * 8;
  const int n = n0 + col8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (tid >> 4) + 16 * i;
    const int m = m0 + row;
    if (m >= p.M || n >= p.N) continue;
    const int nv = (p.N - n) < 8 ? (p.N - n) : 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = Cs[row * CS_LD + col8 + j];
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nv) v[j] += (float)p.bias[n + j];
    }
    if (p.rowadd) {
      const half_t* ra_ = p.rowadd + (size_t)(m / p.rows_per_group) * p.ldra + n;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nv) v[j] += (float)ra_[j];
    }
    if (p.act == ACT_SILU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
    } else if (p.act == ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    half_t* dst = p.C + (size_t)m * p.ldc + n;
    const bool vec = nv == 8 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    if (p.residual) {
      const half_t* rs = p.residual + (size_t)m * p.ldr + n;
      if (vec && ((reinterpret_cast<uintptr_t>(rs) & 15) == 0)) {
        const half8_t rv = *reinterpret_cast<const half8_t*>(rs);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
      } else {
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

static const size_t kGemmSmem = (size_t)BM * CS_LD * 4 > (size_t)2 * (BM + BN) * BK * 2 ? (size_t)BM * CS_LD * 4 : (size_t)2 * (BM + BN) * BK * 2;

static int launch_gemm(GemmParams& p, bool conv, hipStream_t stream) {
  MD_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "md_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  MD_CHECK_ARG(p.K % BK == 0, "md_gemm: K=%d must be a multiple of %d (pad channels when packing)", p.K, BK);
  MD_CHECK_ARG((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0, "md_gemm: A/W must be 16-byte aligned");
  MD_CHECK_ARG(conv || p.lda % 8 == 0, "md_gemm: lda=%d must be a multiple of 8", p.lda);
  if (p.act == ACT_GEGLU) {
    MD_CHECK_ARG(p.N % 64 == 0 && !p.transpose_out && !p.residual && !p.rowadd && p.ldc % 8 == 0, "md_gemm: GEGLU needs N %% 64 == 0 (N=%d), ldc %% 8 == 0 and no residual/rowadd/transpose", p.N);
  }
  if (p.transpose_out) MD_CHECK_ARG(!p.residual && !p.rowadd && p.act == ACT_NONE, "md_gemm: transposed store supports bias only");
  if (p.rowadd) MD_CHECK_ARG(p.rows_per_group > 0, "md_gemm: rows_per_group must be > 0 with rowadd");
  p.tiles_n = cdiv(p.N, BN);
  p.tiles_total = cdiv(p.M, BM) * p.tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmSmem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmSmem);
    attr_set = true;
  }
  if (conv)
    hipLaunchKernelGGL(gemm_kernel<true>, dim3(p.tiles_total), dim3(256), kGemmSmem, stream, p);
  else
    hipLaunchKernelGGL(gemm_kernel<false>, dim3(p.tiles_total), dim3(256), kGemmSmem, stream, p);
  MD_CHECK_LAUNCH("md_gemm");
  return MD_OK;
}

extern "C" int md_gemm_f16(const void* A, int lda, const void* W, void* C, int ldc, int M, int N, int K, const void* bias, const void* residual,
                           int ldr, const void* rowadd, int ldra, int rows_per_group, int act, int transpose_out, void* stream) {
  GemmParams p = {};
  p.A = (const half_t*)A; p.W = (const half_t*)W; p.C = (half_t*)C;
  p.bias = (const half_t*)bias; p.residual = (const half_t*)residual; p.rowadd = (const half_t*)rowadd;
  p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldra = ldra;
  p.M = M; p.N = N; p.K = K; p.rows_per_group = rows_per_group; p.act = act; p.transpose_out = transpose_out;
  return launch_gemm(p, false, (hipStream_t)stream);
}

extern "C" int md_conv3x3_nhwc_f16(const void* X, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout, int stride,
                                   int upsample, const void* bias, const void* residual, int ldr, const void* rowadd, int ldra,
                                   int rows_per_group, int act, void* stream) {
  MD_CHECK_ARG(Cin % BK == 0, "md_conv3x3: Cin=%d must be a multiple of %d (zero-pad channels when packing)", Cin, BK);
  MD_CHECK_ARG(stride == 1 || stride == 2, "md_conv3x3: stride must be 1 or 2");
  MD_CHECK_ARG(upsample == 0 || (upsample == 1 && stride == 1), "md_conv3x3: upsample is 0 or 1 (nearest 2x) with stride 1");
  GemmParams p = {};
  p.A = (const half_t*)X; p.W = (const half_t*)W; p.C = (half_t*)Y;
  p.bias = (const half_t*)bias; p.residual = (const half_t*)residual; p.rowadd = (const half_t*)rowadd;
  p.ldc = ldy; p.ldr = ldr; p.ldra = ldra; p.lda = Cin;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.stride = stride; p.upsample = upsample;
  const int hup = Hin << upsample, wup = Win << upsample;
  p.Hout = (hup + 2 - 3) / stride + 1;
  p.Wout = (wup + 2 - 3) / stride + 1;
  p.M = B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin;
  p.rows_per_group = rows_per_group; p.act = act; p.transpose_out = 0;
  return launch_gemm(p, true, (hipStream_t)stream);
}
