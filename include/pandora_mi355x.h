/*
 * pandora_mi355x.h — C-ABI of the MI355X (gfx950) kernels behind Open-Pandora's DDIM / 3-D U-Net
 * denoising hot path.
 *
 * The reference (OpenSparseLLMs/Open-Pandora) has no FFI for this path: every op below replaces a
 * PyTorch call site inside DynamiCrafter/lvdm (cited per entry point as file:line relative to the
 * reference checkout).  The library is called through ctypes by open-pandora_amd/capi.py; PyTorch
 * only owns the device memory and the stream.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()); tensors are dense in the documented
 *     layout unless a leading dimension (ld*, in ELEMENTS) is given;
 *   - activations are channels-last token matrices: [frames * H * W, C] row-major;
 *   - dtype: PM_F16 (IEEE half) or PM_BF16 for activations and weights; accumulation, statistics,
 *     biases and the DDIM latent are f32;
 *   - no allocation, no synchronisation, no exceptions: work is enqueued on `stream`
 *     (a hipStream_t passed as void*) and the call returns PM_OK (0) or a negative PM_E* code;
 *   - re-entrant: no mutable state between calls.  (Write-once process state only: per-device launch attributes
 *     and the CU count, cached on first use.  This library reads NO environment variable: the PANDORA_* kernel-tuning
 *     overrides exist only in the diagnostics build, include/pandora_mi355x_diag.h.)
 */
#ifndef PANDORA_MI355X_H
#define PANDORA_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_E_DTYPE (-1)   /* unsupported dtype code                                   */
#define PM_E_SHAPE (-2)   /* shape / alignment precondition violated                  */
#define PM_E_NULL (-3)    /* required pointer is NULL                                 */
#define PM_E_LAUNCH (-4)  /* hipLaunchKernel reported an error                        */
#define PM_E_WORKSPACE (-5) /* workspace too small                                    */

#define PM_F16 1
#define PM_BF16 2
#define PM_F32 3 /* only where an entry point says so (norm inputs, model outputs, the residual stream) */
#define PM_OUT_HILO 0x100 /* OR-ed into the out_dtype of pm_groupnorm_apply / pm_layernorm (the parity configuration): row m
                           * of y is [hi | lo] over 2C columns, hi = round16(v), lo = round16(v - hi) (ldy >= 2C).  A
                           * GEMM / conv over the 2C channels with the weights repeated then sees the normalised
                           * activation - the A operand whose 16-bit rounding is 57 % of the end-to-end error^2
                           * (tests/test_error_budget_gpu.py) - at about twice the mantissa, for twice the MFMA work */

/* GEMM-family flags.  The U-Net's residual stream (the tensor every block adds into) is kept in f32
 * so that 16-bit rounding happens once per branch operand instead of once per residual add. */
#define PM_FLAG_A_F32 1   /* the A operand (activations) is f32; rounded to `dtype` while staging   */
#define PM_FLAG_OUT_F32 2 /* C and `residual` are f32 (bias/act/residual add and the store in f32) */
#define PM_FLAG_RES_F32 4 /* `residual` is f32 while C stays 16-bit (last add of a block's stream)  */
#define PM_FLAG_A_LO 16 /* pm_gemm with PM_FLAG_A_F32: stage  a - round(a)  (the low part the plain pass drops).  A
                        * GEMM whose A operand IS the f32 residual stream (the 1x1 skip_connection of a ResBlock
                        * that changes width, openaimodel3d.py:185-190) puts the whole stream through one 16-bit
                        * rounding; two passes - plain, then PM_FLAG_A_LO accumulating onto the first result
                        * (residual = C, in place) - carry it at ~2x the mantissa instead */
#define PM_FLAG_BIAS_IS_SCALE 8 /* pm_gemm: `bias[n]` MULTIPLIES column n (in f32, before the one rounding of the
                                 * store) instead of adding: the softmax scale on the q third of a fused q|k|v
                                 * projection (attention.py:103 `* self.scale`), at no extra rounding */

#define PM_FLAG_W_WRAP 32 /* pm_gemm, 16-bit A: W has K/2 columns and is walked twice, C = A[:, :K/2] W^T + A[:, K/2:] W^T.
                           * With A = [hi | lo] from pm_split16 this is the split-operand product of PM_FLAG_A_LO in ONE
                           * pass through the DMA-staged 16-bit kernels (K/2 % 64 == 0) */

/* activation fused into a GEMM / conv epilogue */
#define PM_FLAG_STATS_I64 64 /* GEMM family, with `colstats` != NULL (r06): `colstats` is int64_t totals [NI][32][4][8] (NI = bits 8..15 of
                             * `flags`; M % NI == 0, rows per instance a multiple of pm_gemm_colstats_rows, N % 32 == 0) that the epilogue ADDS the
                             * GroupNorm sums of the stored values to - per (instance, group of N/32 columns): {sum, sumsq} as two
                             * fixed-point limbs each (2^-12 and 2^-44 units, element 0 of a 64-byte sector each; integer atomics: the totals do not depend on arrival order).
                             * The caller zeroes the buffer; pm_groupnorm_apply reads it with PM_TOTALS_I64.  No finalize launch.
                             * Shipped library: honoured on split-K plans (pm_gemm_colstats_rows == 16: the reduce pass adds the totals); an unsplit call
                             * returns PM_E_SHAPE - the MFMA epilogues carry the atomics in the diagnostics build only (a measured loss). */
#define PM_TOTALS_I64 0x200 /* OR-ed into the in_dtype of pm_groupnorm_stats (totals = int64 limbs as above, added to; `partials` unused)
                             * and into the out_dtype of pm_groupnorm_apply (totals = int64 limbs; bits 16..23 of out_dtype = nsum >= 1:
                             * instance i uses the integer sum of entries i*nsum .. i*nsum + nsum - 1, e.g. per-frame sums -> clip sums) */
#define PM_ACT_NONE 0
#define PM_ACT_SILU 1
#define PM_ACT_GEGLU 2 /* weight rows interleaved x/gate in blocks of 16 (see pm_gemm) */
#define PM_ACT_GELU 3  /* erf GELU (nn.GELU default): the image Resampler's feed-forward, resampler.py:27-34 */

const char* pm_strerror(int code);
int pm_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * pm_gemm:  C[M, N] = epilogue(A[M, K] · W[N, K]^T)            (MFMA 16x16x32, f32 accumulate)
 * replaces nn.Linear / 1x1 Conv call sites: attention.py:53-57,86-99,144 (to_q/k/v/out),
 * :269,290,302,306 (proj_in/out), :336,362,374,405, :418-442 (GEGLU feed-forward),
 * openaimodel3d.py:185-190 (1x1 skip_connection).
 *   epilogue: + bias[n] (f32, may be NULL) -> act -> + residual[m, n] (may be NULL) -> store.
 *   flags: PM_FLAG_A_F32 (lda % 4 == 0), PM_FLAG_OUT_F32 (not with GEGLU), PM_FLAG_BIAS_IS_SCALE (bias != NULL,
 *     not with GEGLU: epilogue = * bias[n] -> act -> + residual).
 *   PM_ACT_GEGLU: W holds 2*Nout rows interleaved [16 value rows | 16 gate rows] per 32-row group,
 *     bias likewise; C is [M, Nout] with Nout = N/2:  C = (v + bv) * gelu_erf(g + bg).
 *   requirements: K % 64 == 0, lda/ldw % 8 == 0, 16-byte aligned bases; any M, N >= 1.
 */
int pm_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
            const void* residual, int64_t ldr, void* C, int64_t ldc, int64_t M, int64_t N,
            int64_t K, int act, int flags, int dtype, void* workspace, size_t workspace_bytes,
            float* colstats, void* stream);

/* Fused GroupNorm statistics.  When `colstats` is not NULL the GEMM-family epilogue also writes, per
 * block of 64 output rows and output column, {sum, sum of squares} of exactly the values it stores:
 * colstats [ceil(M/rows)][Nout][2] f32, rows = pm_gemm_colstats_rows(...) (64, or 16 for a split-K call).  pm_groupnorm_finalize_colstats turns
 * them into GroupNorm totals [NI][groups][2] for NI instances of mtiles/NI consecutive row blocks each
 * (mtiles = ceil(M/rows); per-frame statistics: rows per frame % rows == 0; (T,H,W): NI = 1), replacing pm_groupnorm_stats' read
 * pass over the tensor for the GroupNorm that follows a conv (openaimodel3d.py:178-183,258-269). */
int pm_groupnorm_finalize_colstats(const float* colstats, float* totals, int64_t mtiles, int64_t C,
                                   int64_t NI, int groups, void* stream);

/* Split-K scratch.  GEMM-shaped calls whose output has too few 128x128 tiles to fill 256 CUs (the
 * deep U-Net levels: M = 640..2560 rows, K up to 23040) split the K loop over several workgroups that
 * write f32 partial slabs into `workspace`, followed by one reduce pass that applies the epilogue.
 * pm_gemm_workspace_bytes returns the bytes that call shape would use (0 = never splits); a NULL or
 * too small workspace is legal everywhere and simply disables splitting.  For the conv entry points
 * M = F*Ho*Wo, N = Cout, K = 9*Cin (3*Cin for the temporal conv). */
size_t pm_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int act);

/* Row-block size of the fused GroupNorm column sums (`colstats`) for a call shape: 64 when the call runs
 * unsplit (sums come from the main kernel's epilogue), 16 when it splits over K with the given workspace (sums
 * come from the reduce pass; needs N % 4 == 0, ldc % 4 == 0).  colstats is [ceil(M/rows)][Nout][2], and
 * pm_groupnorm_finalize_colstats takes mtiles = ceil(M/rows). */
int pm_gemm_colstats_rows(int64_t M, int64_t N, int64_t K, int act, size_t workspace_bytes);

/* Measurement aid (bench.py attributes launch times to kernels with it; no call site in the reference): which
 * kernel a pm_gemm call of this shape runs on, by the library's own plan - 0 = gemm_kernel (128x128 tile, two LDS
 * stages, two workgroups per CU), 1 = gemm_ring_kernel (wave-specialised 4-stage ring, one workgroup per CU),
 * 2 = gemm256_kernel (256x256 tile, 8 waves, ping-pong phases), 3 = gemm_ringw_kernel (the ring kernel on a 256x128 tile:
 * the large projections whose grids keep whole rounds at half the tile count), 4 = gemm_wide_kernel (256x256 tile, four waves of
 * 128x128, assembly main loop: long-K shapes in whole rounds), 5 = gemm_wide_stream_kernel (the same tile as one assembly statement,
 * K stream continuous across tiles, assembly epilogue: the 16-bit projection / GEGLU flavours on whole 256x256 tiles); negative = PM_E_SHAPE.  `flags` as for pm_gemm, `workspace_bytes` the split-K scratch the call would be given. */
int pm_gemm_kernel_choice(int64_t M, int64_t N, int64_t K, int act, int flags, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------
 * pm_conv2d_3x3: implicit-GEMM 3x3 convolution, padding 1, on channels-last frames.
 * replaces ResBlock in_layers[2] / out_layers[3] (openaimodel3d.py:157,182,221,232), the stem
 * (:390), Downsample.op stride 2 (:68-70), Upsample nearest x2 + conv (:96,100-108) and out[2] (:549).
 *   x: [F, H, W, Cin] (ldx = elements per pixel row, >= Cin);  Wp: packed [Cout, 9*Cin] with
 *   k = (ky*3 + kx)*Cin + c;  y: [F, Ho, Wo, Cout] with Ho = H*up/stride (ceil), idem Wo.
 *   upsample2x != 0: the conv reads a virtual (2H, 2W) nearest-neighbour image of x.
 *   pad_lo: zero rows/columns BEFORE the image (1 = symmetric padding 1; 0 = the (0,1,0,1) padding of
 *   the first-stage encoder's stride-2 Downsample, ae_modules.py:99-103); one row/column after, always.
 *   Ho = (H*up + pad_lo - 2) / stride + 1.
 *   epilogue as pm_gemm (bias f32 [Cout]; residual [F*Ho*Wo, Cout] with ldr).
 *   zero_page: >= 16 bytes of device zeros (source of padded taps).
 */
int pm_conv2d_3x3(const void* x, int64_t ldx, const void* Wp, const float* bias,
                  const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t F, int64_t H,
                  int64_t W, int64_t Cin, int64_t Cout, int stride, int upsample2x, int pad_lo,
                  const void* zero_page, int flags, int dtype, void* workspace,
                  size_t workspace_bytes, float* colstats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_conv_temporal_k3: Conv3d kernel (3,1,1), padding (1,0,0) = 3-tap conv along the frame axis.
 * replaces TemporalConvBlock.conv1..4 (openaimodel3d.py:258-269,275-282).
 *   x: [F, P, Cin] frames of P = H*W pixels (ldx per pixel row); Wp: packed [Cout, 3*Cin] with
 *   k = kt*Cin + c;  y: [F, P, Cout].  Frame f reads frames f-1, f, f+1.
 *   halo_lo / halo_hi: optional [P, Cin] frames standing for frame -1 / frame F (frame-sharded
 *   mode, same ldx); NULL means zero padding (clip boundary).
 */
int pm_conv_temporal_k3(const void* x, int64_t ldx, const void* halo_lo, const void* halo_hi,
                        const void* Wp, const float* bias, const void* residual, int64_t ldr,
                        void* y, int64_t ldy, int64_t F, int64_t P, int64_t Cin, int64_t Cout,
                        const void* zero_page, int flags, int dtype, void* workspace,
                  size_t workspace_bytes, float* colstats, void* stream);
/* The same over F / clip_frames independent clips batched along the frame axis (the cond / uncond pair of a CFG step as one
 * U-Net forward, ddim.py:233-234): frame f reads f-1 / f+1 only inside its own clip, zero padding at both ends of every
 * clip.  clip_frames divides F; halo frames only with clip_frames == F (pm_conv_temporal_k3 == this with clip_frames = F). */
int pm_conv_temporal_k3_clips(const void* x, int64_t ldx, const void* halo_lo, const void* halo_hi,
                              const void* Wp, const float* bias, const void* residual, int64_t ldr,
                              void* y, int64_t ldy, int64_t F, int64_t clip_frames, int64_t P, int64_t Cin,
                              int64_t Cout, const void* zero_page, int flags, int dtype, void* workspace,
                              size_t workspace_bytes, float* colstats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(32 groups) on channels-last data, optional fused SiLU.
 * replaces GroupNormSpecific / nn.GroupNorm call sites: per-frame statistics (openaimodel3d.py:154-158,
 * 178-183,546-550; attention.py:265,297) and (T,H,W) statistics (openaimodel3d.py:258-269;
 * attention.py:331,368).
 *   x: [NI, P, C] of in_dtype (PM_F16 / PM_BF16 / PM_F32): NI independent instances (NI = frames for per-frame statistics, NI = 1 with
 *   P = F*H*W for (T,H,W) statistics); C % 8 == 0, (C/8) <= 1024, C % groups == 0.
 *   pm_groupnorm_stats: per-chunk partial {sum, sum of squares} go to the scratch `partials`
 *   [NI, nchunks, groups, 2] f32 (nchunks = pm_groupnorm_nchunks(P, C)), then are summed in a fixed
 *   order (deterministic, no atomics) into `totals` [NI, groups, 2] f32.
 *   pm_groupnorm_apply: y = (x - mean) * rstd * gamma + beta, optionally SiLU, from `totals`;
 *   count = elements per group the totals were taken over (P * C/groups, or the all-rank total in
 *   frame-sharded mode, where the caller all-reduces `totals` between the two calls).
 */
int64_t pm_groupnorm_nchunks(int64_t P, int64_t C);
int pm_groupnorm_stats(const void* x, int64_t ldx, float* partials, float* totals, int64_t NI,
                       int64_t P, int64_t C, int groups, int in_dtype, void* stream);
int pm_groupnorm_apply(const void* x, int64_t ldx, const float* totals, const float* gamma,
                       const float* beta, void* y, int64_t ldy, int64_t NI, int64_t P, int64_t C,
                       int groups, double count, float eps, int silu, int in_dtype, int out_dtype,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_layernorm: LayerNorm over the last dimension; replaces BasicTransformerBlock.norm1/2/3
 * (attention.py:225-227,243-245).  x [M, C] of in_dtype (16-bit or PM_F32), y [M, C] of out_dtype
 * (16-bit); gamma, beta f32 [C]; C % 8 == 0, C <= 4096.
 */
int pm_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                 int64_t ldy, int64_t M, int64_t C, float eps, int in_dtype, int out_dtype,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_split16: the f32 residual stream as a 16-bit GEMM / conv operand.  y[:, :K] = round(x) and, with_lo != 0,
 * y[:, K:2K] = round(x - round(x)): what PM_FLAG_A_F32 (/ PM_FLAG_A_LO) do while staging, as one HBM-bound pass,
 * so that the consumer runs on the DMA-staged 16-bit kernels (the register-staged f32 loaders run at a third of
 * their rate).  Call sites: the 1x1 skip_connection of a width-changing ResBlock (openaimodel3d.py:185-190:
 * [hi | lo] + PM_FLAG_W_WRAP), Downsample.op / Upsample.conv on the stream (:68-70, :96-108).
 *   x f32 [M, K] (ldx % 4 == 0), y `dtype` [M, K or 2K] (ldy % 8 == 0), K % 8 == 0.
 */
int pm_split16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t K, int with_lo, int dtype,
               void* stream);

/* pm_split16_upsample2x: pm_split16 written through the nearest-neighbour x2 interpolation of Upsample
 * (openaimodel3d.py:96-108: F.interpolate(scale_factor=(1, 2, 2), mode="nearest") in front of its 3x3 conv): output row
 * (f, oy, ox) of a 2H x 2W frame = converted input row (f, oy >> 1, ox >> 1).  The conv behind it then runs in the fast
 * 3x3 mode on the DMA-staged 256x128 / 128x128 ring kernels instead of gathering the upsampled pixels per lane in the
 * general mode (2-stage kernel, 0.59-0.76 PFLOP/s on these 1.1-TFLOP launches).
 *   x f32 [F*H*W, K] (ldx % 4 == 0), y `dtype` [F*2H*2W, K or 2K] (ldy % 8 == 0), K % 8 == 0. */
int pm_split16_upsample2x(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t F, int64_t H, int64_t W, int64_t K,
                          int with_lo, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_ln_gemm:  C[M, N] = epilogue( LayerNorm(X)[M, K] · W[N, K]^T )  in ONE kernel: the LayerNorm output never
 * exists in HBM.  replaces the pairs norm1 -> attn1.to_q|k|v, norm2 -> attn2.to_q (or to_q|k|v), norm3 ->
 * ff.net[0] (GEGLU) of BasicTransformerBlock._forward (attention.py:242-246 with :86-99 and :418-442) at the
 * shallowest U-Net level, i.e. a pm_layernorm + pm_gemm call pair with identical results (same LayerNorm
 * arithmetic, the normalised rows rounded to `dtype` once, f32 accumulation).
 *   X f32 [M, K] (the residual stream; ldx % 4 == 0), gamma / beta f32 [K], W `dtype` [N, K] (ldw % 8 == 0),
 *   C `dtype` [M, N] (ldc % 8 == 0) or, with PM_ACT_GEGLU, [M, N/2] (ldc % 4 == 0; W / bias packed as for pm_gemm).
 *   epilogue: + bias[n] (may be NULL), or * bias[n] with PM_FLAG_BIAS_IS_SCALE (the only flag accepted); act is
 *   PM_ACT_NONE or PM_ACT_GEGLU.  K must be 320 (a panel of 128 normalised rows stays in LDS for the sweep
 *   over N; the level-1 width 640 was measured slower than the pair and is not served), N % 32 == 0.  pm_ln_gemm_supported says whether a shape is served AND worth it (enough
 *   rows to amortise a workgroup's pass over W); callers use the pm_layernorm + pm_gemm pair otherwise.
 */
int pm_ln_gemm(const float* X, int64_t ldx, const float* gamma, const float* beta, float eps, const void* W,
               int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
               int flags, int dtype, void* stream);
int pm_ln_gemm_supported(int64_t M, int64_t N, int64_t K, int act);

/* ------------------------------------------------------------------------------------------------
 * pm_attention: softmax(q k^T * scale) v with head dim 64, flash-style (no score matrix in HBM),
 * up to two key/value segments that share q and are softmax-normalised independently:
 *     out = attn(q, k1, v1) + w2 * attn(q, k2, v2)
 * replaces CrossAttention.forward for the spatial self-attention (attention.py:101-125), the text
 * + image cross-attention (:89-94,128-142) and CrossAttention.efficient_forward (:146-209).
 *   q:  element (b, i, h, d) at q  + b*q_bs  + i*q_rs  + h*64 + d,   i < Nq
 *   kX: element (b, j, h, d) at kX + b*kX_bs + j*kX_rs + h*64 + d,   j < NkX   (vX alike)
 *   o:  same addressing as q with o_bs / o_rs.     strides in elements, multiples of 8;
 *   a batch stride of 0 shares one key/value set between all b (text context).
 *   k2 == NULL disables the second segment.
 */
int pm_attention(const void* q, int64_t q_bs, int64_t q_rs, const void* k1, const void* v1,
                 int64_t k1_bs, int64_t k1_rs, int64_t Nk1, const void* k2, const void* v2,
                 int64_t k2_bs, int64_t k2_rs, int64_t Nk2, float w2, void* o, int64_t o_bs,
                 int64_t o_rs, int64_t B, int64_t heads, int64_t Nq, float scale, int dtype,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_attention_fp8: single-segment softmax(q k^T * scale) v, head dim 64, with fp8 (OCP e4m3) q / k / v / P operands
 * on the block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit scales: K = 64 per instruction at twice the bf16
 * rate), f32 accumulation and softmax statistics, 16-bit output.  BASELINE configs[4] ("fp8 MFMA attention") for the
 * spatial self-attention call site attention.py:101-125; same addressing as pm_attention (k and v share strides).
 * The call first packs q*scale*log2(e), k (fp8 rows) and v (fp8, transposed per head, keys in MFMA operand order) into
 * `workspace` (pm_attention_fp8_workspace_bytes: 64 bytes per (batch, head, padded row) for each of q, k, v).
 * Accuracy: e4m3 carries 3 mantissa bits - expect ~2-3e-2 norm-relative error per call (tests state the bound).
 */
size_t pm_attention_fp8_workspace_bytes(int64_t B, int64_t heads, int64_t Nq, int64_t Nk);
int pm_attention_fp8(const void* q, int64_t q_bs, int64_t q_rs, const void* k, const void* v, int64_t k_bs,
                     int64_t k_rs, int64_t Nk, void* o, int64_t o_bs, int64_t o_rs, int64_t B, int64_t heads,
                     int64_t Nq, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_attention_generic: softmax(q k^T * scale) v for ANY head dim D <= 128 (D % 8 == 0) and Nk <= 1024, f32
 * arithmetic, single key/value segment.  For the OpenCLIP ViT-H/14 image tower of the conditioning tail
 * (nn.MultiheadAttention with 16 heads of 80 channels over 257 tokens, condition.py:300-382, once per generate call);
 * element (b, i, h, d) at base + b*bs + i*rs + h*D + d as in pm_attention (k and v share strides).
 */
int pm_attention_generic(const void* q, int64_t q_bs, int64_t q_rs, const void* k, const void* v, int64_t k_bs,
                         int64_t k_rs, int64_t Nk, void* o, int64_t o_bs, int64_t o_rs, int64_t B, int64_t heads,
                         int64_t Nq, int64_t D, float scale, int dtype, void* stream);

/* (Diagnostics - kernel-variant overrides, ceiling probes, environment tuning switches - are NOT part of this library:
 * they exist only in the -DPM_DIAG build, libpandora_mi355x_diag.so, declared in include/pandora_mi355x_diag.h.  This
 * library reads no environment variable and has no mutable process-wide state.) */

/* ------------------------------------------------------------------------------------------------
 * pm_attention_temporal: self-attention over the frame axis at every pixel (head dim 64).
 * replaces CrossAttention.forward as used by TemporalTransformer (attention.py:365-412; both attn1
 * and attn2 are self-attention over T because only_self_att=True, :347-348,389-390).
 *   q: [Fq, P, heads*64] (ldq per pixel row); k, v: [Fk, P, heads*64] (ldk); o like q (ldo).
 *   Fq in {1,2,4,8,16} (local frames in frame-sharded mode), Fk <= 16.
 */
int pm_attention_temporal(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldk,
                          void* o, int64_t ldo, int64_t Fq, int64_t Fk, int64_t P, int64_t heads,
                          float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_gemv_f32: y[n] = act(sum_k W[n, k] * x[k] + bias[n]) with f32 x / y, 16-bit W.
 * replaces the M = 1 linears on the embedding path: time_embed / fps_embedding
 * (openaimodel3d.py:374-386,554-581) and every ResBlock.emb_layers (:171-177,222).
 *   silu_in != 0 applies SiLU to x first (emb_layers[0]); act as PM_ACT_NONE / PM_ACT_SILU.
 *   K % 8 == 0.
 */
int pm_gemv_f32(const void* W, int64_t ldw, const float* x, const float* bias, float* y,
                int64_t N, int64_t K, int silu_in, int act, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_ddim_update: one DDIM step of p_sample_ddim for the v-parameterisation, f32 latent.
 * replaces ddim.py:238 (CFG combine), :243-245,271 + ddpm3d.py:235-247 (v -> eps, v -> x0),
 * :273-277 (dynamic rescale), :282-288 (direction, noise, x_prev).
 *   v = e_u + cfg * (e_c - e_u)   (e_u may be NULL: v = e_c)
 *   eps = sqrt_ac * v + sqrt_1mac * x ;  x0 = (sqrt_ac * x - sqrt_1mac * v) * rescale
 *   x_prev = sqrt_a_prev * x0 + dir_coef * eps + sigma * noise   (noise may be NULL when sigma == 0)
 *   e_c / e_u are model outputs in `dtype` (PM_F16 / PM_BF16 / PM_F32); x, noise, x_prev, pred_x0 are
 *   f32; n = element count.
 */
int pm_ddim_update(const float* x, const void* e_c, const void* e_u, const float* noise,
                   float* x_prev, float* pred_x0, int64_t n, float cfg, float sqrt_ac,
                   float sqrt_1mac, float rescale, float sqrt_a_prev, float dir_coef, float sigma,
                   int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * pm_timestep_embedding: the sinusoidal timestep / fps embedding, y[i, j] = cos(t[i] * freqs[j]), y[i, half + j] =
 * sin(t[i] * freqs[j]) (f32 product, f32 cos / sin), j < half.  replaces timestep_embedding (utils_diffusion.py:8-28) in front
 * of time_embed / fps_embedding (openaimodel3d.py:554-581): the caller passes the reference's frequency table (its bf16-
 * quantised arange through exp, built once on the host).  t: n values on the device, int64 (t_is_i64 != 0) or f32.
 */
int pm_timestep_embedding(const void* t, int t_is_i64, const float* freqs, float* y, int64_t n, int64_t half, void* stream);

/* ------------------------------------------------------------------------------------------------
 * layout helpers on the path boundary (DiffusionWrapper 'hybrid' concat ddpm3d.py:1077-1081 and the
 * `b c t h w -> (b t) c h w` shuffles openaimodel3d.py:570,606):
 *   pm_pack_input:  x f32 [C1, F, P] and cond f32 [C2, F, P]  ->  y dtype [F, P, C1 + C2]
 *   pm_unpack_output: y [F, P, C] -> out [C, F, P], both of `dtype` (PM_F16 / PM_BF16 / PM_F32)
 */
int pm_pack_input(const float* x, const float* cond, void* y, int64_t C1, int64_t C2, int64_t F,
                  int64_t P, int dtype, void* stream);
int pm_unpack_output(const void* y, void* out, int64_t C, int64_t F, int64_t P, int dtype,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * First-stage decoder helpers (AutoencoderKL.decode after the loop,
 * lvdm/models/autoencoder.py:103-106, ddpm3d.py:630-655).  The decoder's convolutions, GroupNorm+swish
 * and 1x1 projections run on pm_conv2d_3x3 / pm_groupnorm_* / pm_gemm; two small kernels complete it:
 *   pm_latent_affine: y[f, p, :] = W . (x[:, f, p] * inv_scale) + b, channels-last and zero-padded to
 *     Cpad (1/scale_factor and post_quant_conv, ddpm3d.py:641-645 + autoencoder.py:104); x f32 [C, F, P].
 *   pm_softmax_rows: y = softmax(scale * x) per row, x f32 [M, N] scores -> y [M, N] in `dtype`
 *     (the single-head, 512-channel mid attention of ae_modules.py:52-75 is QK^T and PV on pm_gemm
 *     around this kernel); N % 4 == 0.
 */
int pm_latent_affine(const float* x, const float* W, const float* b, void* y, int64_t C, int64_t Cpad,
                     int64_t F, int64_t P, float inv_scale, int dtype, void* stream);
int pm_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t N,
                    float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Peer mailboxes: the latency-class exchanges of the frame-sharded forward (the 256-byte
 * (T,H,W)-GroupNorm partial sums of every rank + the boundary frames of the two neighbour ranks in front of each
 * temporal-conv stage, openaimodel3d.py:258-282) as ONE kernel launch per exchange: direct peer writes over xGMI
 * into hipIpc-mapped mailboxes, arrival counters at system scope, no RCCL call and no host round trip, so the
 * launch can sit inside a captured HIP graph.  No counterpart in the reference (its only collective,
 * lvdm/common.py:8-14, is never called); the bulk frames<->pixels all-to-all stays on RCCL.
 *   These five are the only entry points that allocate / map memory (pm_peer_create, pm_peer_open) or synchronise
 *   (pm_peer_create, pm_peer_status); pm_peer_exchange itself follows the conventions at the top of this file.
 *   pm_peer_mailbox_bytes: size of one rank's mailbox for `world` ranks (<= 16), up to nstat_max partial sums and
 *     halo frames of up to halo_bytes_max bytes (two slots, alternating by an epoch counter inside the mailbox).
 *   pm_peer_create: zeroed mailbox on the current device + its 64-byte hipIpc handle (*fine_grained: 1 if the
 *     allocation is fine-grained device memory, 0 if the runtime only shared plain device memory).
 *   pm_peer_open / pm_peer_close: map / unmap a peer's mailbox from its handle; pm_peer_destroy frees one's own.
 *   pm_peer_status: synchronous read of {exchanges completed, error}: error = 1 after a poll timed out (the
 *     kernel never hangs: it gives up after `timeout_s`, raises the word and lets the stream continue).
 *   pm_peer_exchange: enqueue one exchange on `stream`.  stats f32 [nstat] -> totals f32 [nstat] = the sum over
 *     ranks IN RANK ORDER (bitwise identical on every rank).  first / last (both or neither): this rank's first and
 *     last frame, halo_bytes each (multiple of 16) -> lo_out (the frame before this rank's first, from rank - 1;
 *     ignored on rank 0) and hi_out (after its last, from rank + 1; ignored on the last rank).  `peers` is a HOST
 *     array of `world` mapped mailbox pointers (entry `rank` unused).  Every rank of the group must enqueue the same
 *     sequence of exchanges with the same sizes.
 */
size_t pm_peer_mailbox_bytes(int world, int64_t nstat_max, int64_t halo_bytes_max);
int pm_peer_create(size_t bytes, void** base, void* handle, int* fine_grained);
int pm_peer_open(const void* handle, void** base);
int pm_peer_close(void* base);
int pm_peer_destroy(void* base);
int pm_peer_status(const void* base, int* epoch, int* error);
int pm_peer_exchange(void* mine, const void* const* peers, int rank, int world, const float* stats, int64_t nstat,
                     const void* first, const void* last, int64_t halo_bytes, float* totals, void* lo_out,
                     void* hi_out, int64_t nstat_max, int64_t halo_bytes_max, double timeout_s, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PANDORA_MI355X_H */
