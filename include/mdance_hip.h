/* libmdance_hip.so -- C ABI of the MI355X (gfx950) denoising-loop kernels.
 *
 * The reference (Kebii/MikuDance) has NO FFI on this path: every hot-path FLOP is a stock PyTorch/ATen call made
 * from Python.  The boundary is therefore op-level: each entry point below replaces the
 * ATen dispatches of the cited reference call sites (paths relative to the reference repository root).  The
 * reference-side binding is a ctypes stub; mikudance_amd/_lib.py is that stub in this repo.
 *
 * Conventions: all device pointers are raw addresses owned by the caller (PyTorch-ROCm allocates them); nothing is
 * allocated, freed or retained; kernels are enqueued on `stream` (a hipStream_t passed as void*) and the call
 * returns immediately.  Activations are fp16, token-major / NHWC: a (B,H,W,C) image batch IS the row-major matrix
 * [B*H*W][C].  Weights are fp16 [N][K] with K contiguous (nn.Linear layout; 3x3 conv weights as
 * [Cout][ky][kx][Cin]).  Return value: 0 on success, negative on error (md_last_error() has the message, per thread).
 *
 * Threading / devices: entry points may be called concurrently from several host threads on distinct streams, and one
 * process may drive several GPUs (hipSetDevice before the call): the library keeps no per-call state, its one-time kernel
 * attribute setup (> 64 KiB dynamic LDS) is done per device behind an atomic mask, and the MD_* tuning knobs are read from
 * the environment once per process (thread-safe initialisation).  Two calls on the SAME stream are ordered by the stream.
 *
 * Aliasing: outputs must not overlap inputs, with ONE exception that the callers rely on: `residual` may be exactly the
 * output (same pointer, same pitch) of md_gemm_f16 / md_conv3x3*_nhwc_f16 -- every kernel flavour reads a residual element
 * in the thread that later writes the same output element (or, in the streaming kernel, reads whole rows that no workgroup
 * has written yet); a partially overlapping residual is rejected.  md_groupnorm_nhwc_f16 and md_softmax_rows_f16 may run in
 * place (y == x).
 */
#ifndef MDANCE_HIP_H
#define MDANCE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MD_ACT_NONE 0
#define MD_ACT_SILU 1
#define MD_ACT_RELU 2
#define MD_ACT_GEGLU 3 /* W rows packed as alternating blocks of 32 'h' rows and 32 'g' rows; C gets N/2 columns */
#define MD_ACT_QUICKGELU 4 /* x * sigmoid(1.702 x): CLIP vision tower MLP (src/pipelines/pipeline_mikudance.py:406-416) */

int md_version(void);
const char* md_last_error(void);

/* C[M,N] = epi(A[M,K] . W[N,K]^T): bias[N], act, rowadd[(m / rows_per_group)][n] (time-embedding broadcast),
 * residual[M,N] (added last), or transpose_out (C is [N][ldc], used for V^T).  K % 64 == 0.
 * Replaces nn.Linear / 1x1 Conv2d: diffusers Attention.to_q/to_k/to_v/to_out.0 and FeedForward
 * (src/models/attention.py:109-157,323-364), proj_in/proj_out (src/models/transformer_3d.py:66-68,96-98;
 * src/models/transformer_2d.py:152-154,186-188; src/models/motion_module.py:124,146), conv_shortcut and
 * time_emb_proj (src/models/resnet.py:179-181,213-215), TimestepEmbedding (src/models/unet_3d_mix.py:99-102). */
int md_gemm_f16(const void* A, int lda, const void* W, void* C, int ldc, int M, int N, int K, const void* bias,
                const void* residual, int ldr, const void* rowadd, int ldra, int rows_per_group, int act,
                int transpose_out, void* stream);

/* Y = epi(conv3x3(X)) on NHWC, zero padding 1, stride 1|2, upsample=1 folds a nearest-2x upsample into the input
 * addressing.  Cin % 64 == 0 (zero-pad when packing).  Output (B, Hout, Wout, Cout) with row pitch ldy.
 * Replaces InflatedConv3d / Conv2d 3x3: src/models/resnet.py:9-17,71-88,106-120,165-167,194-196;
 * src/models/unet_3d_mix.py:94-96,267-269; src/models/unet_2d_mix.py:321-326; src/models/man_module.py:18-21. */
int md_conv3x3_nhwc_f16(const void* X, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout,
                        int stride, int upsample, const void* bias, const void* residual, int ldr,
                        const void* rowadd, int ldra, int rows_per_group, int act, void* stream);

/* Same, with pad_lo = 0: the zero padding is (0,1,0,1) (right / bottom only) and stride 2, i.e. F.pad(x, (0,1,0,1)) +
 * Conv2d(3x3, stride 2, padding 0): the downsampler of the third-party AutoencoderKL encoder (diffusers 0.24.0
 * Downsample2D(padding=0)) that src/pipelines/pipeline_mikudance.py:456-549 calls through self.vae.encode. */
int md_conv3x3_pad_nhwc_f16(const void* X, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout,
                            int stride, int upsample, int pad_lo, const void* bias, const void* residual, int ldr,
                            const void* rowadd, int ldra, int rows_per_group, int act, void* stream);

/* The general form of the two entries above: X may be a channel slice of a wider NHWC tensor (pixel pitch ldx >= Cin elements:
 * the skip connections of the UNets are produced straight into the concat buffer of the up-block resnet that consumes them, so
 * torch.cat([hidden, skip], 1) of src/models/unet_3d_blocks.py:736,877 / unet_2d_blocks.py never runs as a copy), and kw = 1 selects
 * a 3 x 1 filter (taps along H only, K = 3 Cin, stride 1, pad 1): nn.Conv3d(C, C, (3,1,1), padding (1,0,0)) of the third-party
 * AutoencoderKLTemporalDecoder (src/pipelines/pipeline_mikudance.py:132-150) on the (clips, frames, h*w, C) view, ONE launch and one
 * rounding instead of three accumulating GEMMs. */
int md_conv_nhwc_f16(const void* X, int ldx, const void* W, void* Y, int ldy, int B, int Hin, int Win, int Cin, int Cout, int kw,
                     int stride, int upsample, int pad_lo, const void* bias, const void* residual, int ldr, const void* rowadd,
                     int ldra, int rows_per_group, int act, void* stream);

/* In-place softmax(scale * x) over the rows of a row-major fp16 matrix [rows][ldx] (cols valid, cols % 8 == 0).
 * The score matrix of the single 512-channel head of the AutoencoderKL mid-block attention (QK^T and PV run on
 * md_gemm_f16): src/pipelines/pipeline_mikudance.py:115-130 (decode_latents), :456-549 (vae.encode). */
int md_softmax_rows_f16(void* x, int ldx, int rows, int cols, float scale, void* stream);

/* GroupNorm(G, eps) [+SiLU] over (B, HW, C) NHWC.  src/models/resnet.py:20-28,220-221,231,237;
 * src/models/transformer_3d.py:60-62,130; src/models/motion_module.py:121-123,164; unet_3d_mix.py:591-592.
 * Alignment (all md_groupnorm_* entries): x, y, gamma and beta must be 16-byte aligned and C % 8 == 0 (the apply sweep loads gamma /
 * beta eight channels at a time); MD_ERR_ARG otherwise.  md_groupnorm_table_f16 reads gamma / beta as scalars but keeps the same rule
 * so that a caller can switch between the fused and the literal pair without re-packing. */
size_t md_groupnorm_workspace_bytes(int B, int HW, int C, int G);
int md_groupnorm_nhwc_f16(const void* x, void* y, const void* gamma, const void* beta, int B, int HW, int C, int G,
                          float eps, int silu, void* workspace, size_t ws_bytes, void* stream);

/* Same with an input pixel pitch ldx >= C (x a channel slice of a wider NHWC tensor; y stays contiguous; in place only if ldx == C). */
int md_groupnorm_ld_nhwc_f16(const void* x, int ldx, void* y, const void* gamma, const void* beta, int B, int HW, int C, int G,
                             float eps, int silu, void* workspace, size_t ws_bytes, void* stream);

/* LayerNorm over rows of C.  add_mode 0: y only.  add_mode 1: y2[row] = y[row] + add[row - add_row_begin] for
 * rows >= add_row_begin (reference-attention bank ADD, src/models/mutual_mix_attention.py:169-170), y2 = y
 * below.  add_mode 2: y2[row] = y[row] + add[(row / rows_per_frame) % frames] (temporal positional encoding on the
 * query input only, src/models/motion_module.py:404-417).  src/models/attention.py:105-107,331-365. */
int md_layernorm_f16(const void* x, void* y, void* y2, const void* gamma, const void* beta, const void* add, int M,
                     int C, float eps, int add_mode, int add_row_begin, int rows_per_frame, int frames, void* stream);

/* LayerNorm FOLDED into the Linear that consumes it: C = LayerNorm(A; gamma, beta, eps) . W^T + bias [+ rowadd] computed from the RAW rows
 * of A, the normalised tensor never touching HBM.  The caller folds once per layer (mikudance_amd/packing.py ln_fold):
 *   Wf[n][k] = fp16(gamma[k] W[n][k]),   sc = fp32 [2][N]:  sc[0][n] = sum_k Wf[n][k],  sc[1][n] = sum_k beta[k] W[n][k] + bias[n]
 * and the kernel evaluates rstd_m * (A[m] . Wf[n] - mu_m * sc[0][n]) + sc[1][n] with the exact two-pass (mu, rstd) of each row taken
 * from the rows as they stream through LDS.  rowadd / rows_per_group as in md_gemm_f16 (the motion module's query-only positional
 * encoding as a per-frame row term); act must be MD_ACT_NONE.  Only shapes of the W-stationary streaming kernel exist (K = 320, N a
 * multiple of 320, >= 32768 rows: the 96 x 96 level): md_gemm_ln_plan(M, N, K, act, epi) returns 1 when this entry point has a kernel for
 * the problem (epi as md_gemm_plan), 0 when the caller has to run md_layernorm_f16 + md_gemm_f16 on the unfolded weights;
 * md_gemm_ln_f16 itself fails (MD_ERR_ARG) on anything else -- there is no silent fallback.  Replaces norm2 -> attn2.to_q
 * (src/models/attention.py:131-141,339-347; src/models/mutual_mix_attention.py:203-263) and norms[i] -> to_q / to_k / to_v of the
 * motion module (src/models/motion_module.py:245-268, 364-439). */
int md_gemm_ln_plan(int M, int N, int K, int act, int epi);
int md_gemm_ln_f16(const void* A, int lda, const void* Wf, const float* sc, void* C, int ldc, int M, int N, int K, float eps,
                   const void* rowadd, int ldra, int rows_per_group, int act, void* stream);

/* GroupNorm in front of a Linear / 1x1 conv without materialising the normalised tensor.  md_groupnorm_table_f16 runs the statistics
 * sweep only and writes table = fp32 [B][2][C]: scale[b][c] = rstd * gamma[c], shift[b][c] = beta[c] - mean * scale;
 * md_gemm_affine_f16 computes C = (A * scale[image] + shift[image], rounded to fp16) . W^T + bias, image = row / rows_per_image,
 * applying the affine to the rows as they stream through LDS: bit-identical to md_groupnorm_ld_nhwc_f16 (silu = 0, two-sweep form)
 * followed by md_gemm_f16.  A may be a channel slice (lda >= K).  md_gemm_affine_plan: 1 when the kernel exists for the shape (K = 320,
 * N a multiple of 320, >= 32768 rows, rows_per_image % 16 == 0).  Replaces norm -> proj_in of Transformer3DModel / Transformer2DModel
 * (src/models/transformer_3d.py:60-68,121-137; src/models/transformer_2d.py:296-321) and of the motion module's
 * TemporalTransformer3DModel (src/models/motion_module.py:121-124,159-170). */
int md_groupnorm_table_f16(const void* x, int ldx, const void* gamma, const void* beta, int B, int HW, int C, int G, float eps,
                           float* table, void* workspace, size_t ws_bytes, void* stream);
int md_gemm_affine_plan(int M, int N, int K, int rows_per_image);
int md_gemm_affine_f16(const void* A, int lda, const float* table, int rows_per_image, const void* W, void* C, int ldc, int M, int N,
                       int K, const void* bias, void* stream);

/* MAN: y = InstanceNorm(x) * (1 + gamma) + beta; gamma_beta is (B, HW, 2C) = [gamma | beta].
 * src/models/man_module.py:23-33. */
int md_instnorm_spade_f16(const void* x, const void* gamma_beta, void* y, int B, int HW, int C, float eps,
                          void* stream);

int md_instnorm_spade_ld_f16(const void* x, int ldx, const void* gamma_beta, void* y, int B, int HW, int C, float eps,
                             void* stream);                      /* x with a pixel pitch ldx >= C (a channel slice) */

/* O = softmax(Q K^T * scale) V per (batch, head); Vt is V transposed ([H*D][ldvt], md_gemm_f16 transpose_out);
 * kv_index (device int[B], may be NULL) maps a query batch to its K/V batch; kv_stride = tokens between K/V batches.
 * D in {8,16,32,40,64,80,160}.  Replaces F.scaled_dot_product_attention under diffusers AttnProcessor2_0 as
 * called at src/models/mutual_mix_attention.py:141-148,173-200,213-220,257-263. */
int md_attention_fwd_f16(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                         const int* kv_index, int B, int H, int D, int Lq, int Lk, int kv_stride, float scale,
                         void* stream);

/* Attention over FRAMES for every (clip-half, pixel, head); rows are (b*F + frame)*HW + pixel.  F <= 32.  O must not overlap Q, K or V
 * (column-sliced siblings of one wider row-major buffer, e.g. q | k | v of one GEMM, are fine): checked, MD_ERR_ARG otherwise.
 * src/models/motion_module.py:364-439 (VersatileAttention, Temporal mode). */
int md_temporal_attention_fwd_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                                  int ldo, int NB, int F, int HW, int H, int D, float scale, void* stream);

/* Layout packing at the API boundary (any strided fp16/fp32 source -> NHWC fp16 with zero channel padding and
 * nearest resize (Hin,Win)->(Ho,Wo); and back).  src/models/unet_2d_mix.py:1208-1210 (22-channel split),
 * src/models/man_module.py:27 (nearest resize), einops rearranges of src/models/resnet.py:12-16. */
int md_pack_nhwc_f16(const void* src, int src_is_f32, void* dst, int N, int F, long sB, long sF, long sC, long sY,
                     long sX, int c_begin, int c_count, int Cpad, int Ho, int Wo, int Hin, int Win, void* stream);
int md_unpack_nhwc_f16(const void* src, int ldc, void* dst, int dst_is_f32, int N, int F, long sB, long sF, long sC,
                       long sY, long sX, int C, int Ho, int Wo, void* stream);

/* torch.cat([hidden, skip], dim=channel) on token-major matrices.  src/models/unet_3d_blocks.py:736,877. */
int md_concat_channels_f16(const void* a, int Ca, const void* b, int Cb, void* out, long M, void* stream);

/* noise_pred[:, :, window] += pred; counter[window] += 1.  src/pipelines/pipeline_mikudance.py:662-664. */
int md_window_accumulate(const void* pred, void* noise_sum, void* counter, const int* window, int f, int Ftot, int HW,
                         int halves, void* stream);

/* (noise_pred / counter) -> classifier-free guidance -> DDIM v-prediction step (eta 0), latents updated in place.
 * src/pipelines/pipeline_mikudance.py:670-678 + diffusers DDIMScheduler.step. */
int md_cfg_ddim_step(void* latents, const void* noise_sum, const void* counter, int Ftot, int HW, int halves,
                     float guidance, float alpha_t, float alpha_prev, void* stream);

/* The same step for eta > 0 (DDIM's stochastic variant; `eta` of MikuDanceVideoPipeline.__call__,
 * src/pipelines/pipeline_mikudance.py:152-171,375 -> scheduler.step(..., eta=, generator=)): sigma_t = eta * sqrt((1 - a_prev) /
 * (1 - a_t) * (1 - a_t / a_prev)), prev = sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma_t^2) eps + sigma_t z with z = variance_noise,
 * fp16, laid out like the latents (Ftot, HW, 4) -- the caller draws it from ITS generator (diffusers randn_tensor). */
int md_cfg_ddim_step_eta(void* latents, const void* noise_sum, const void* counter, const void* variance_noise, int Ftot, int HW,
                         int halves, float guidance, float alpha_t, float alpha_prev, float eta, void* stream);

/* Persistent launchers (gemm_sp_kernel behind md_gemm_f16 / md_conv*_f16) start one workgroup per CU of the device.  A caller that launches
 * on a stream created with a CU mask (hipExtStreamCreateWithCUMask: a partition of the chip shared with another stream) tells the
 * library how many CUs that stream owns: grids and the tile-choice model then use `ncu` (a multiple of 8: the same number of CUs on each
 * of the 8 XCDs, which also keeps workgroup b on XCD b % 8) until md_set_cu_limit(0) restores the device's own count.  Process-wide, not
 * per stream; the streaming kernels (K = 320 / 640 projections) keep their 256-workgroup grids.  tools/cu_partition.py is the user
 * (profiles/r06_ab_cu_partition.log).  Returns MD_OK, or MD_ERR_ARG for a negative count or one that is not a multiple of 8. */
int md_set_cu_limit(int ncu);

/* Dispatch queries (no device access, nothing launched): which kernel the automatic dispatch of md_gemm_f16 / md_conv3x3_nhwc_f16
 * selects for a problem on a chip with `ncu` compute units, for dense 16-byte aligned operands.  epi: bit 0 residual, bit 1
 * row-broadcast operand, bit 2 bias.  Returns 1MN gemm_sp_kernel with wave tile (MT, NT) = (M, N) (135 = 192x320, 134 = 192x256,
 * 124 = 128x256, 132 = 192x128, 142 = 256x128, 144 = 256x256 GEGLU; +1000 on swapped operands for a transposed output, +2000 when the
 * residual enters the accumulators through the matrix core inside the K loop instead of in the epilogue: K tiles > 32 x 32 sub-tiles of the wave tile), 210 / 220 / 230 the W-stationary
 * streaming kernel (K = 320 / K = 640 / GEGLU), 301 / 302 / 303 the multi-workgroup kernel (64-column / 256x128 / 128x128 tiles),
 * or a negative MD_ERR code.  They pin the measured dispatch table in CPU tests. */
int md_gemm_plan(int M, int N, int K, int act, int transpose_out, int epi, int ncu);
int md_conv3x3_plan(int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample, int epi, int ncu);

#ifdef __cplusplus
}
#endif
#endif
