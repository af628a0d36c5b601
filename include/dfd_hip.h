/* dfd_hip.h — C ABI of libdfd_hip.so: the MI355X (gfx950) kernels behind the
 * DeepfakeDetection image-classifier hot loop.
 *
 * The reference (thourihan/DeepfakeDetection) has no FFI of its own: its hot loop
 * calls third-party nn.Modules (efficientnet_pytorch 0.7.1 / timm 1.0.20) whose
 * arithmetic runs in ATen/cuDNN.  Every entry point below therefore cites the
 * reference CALL SITE whose arithmetic it carries:
 *
 *   forward          logits = model(inputs)            trainers/efficientnet.py:297, :254
 *                                                      orchestration/orchestrator.py:529, :590
 *   loss             criterion(logits, targets)        trainers/efficientnet.py:298, :412
 *   backward         scaler.scale(loss).backward()     trainers/efficientnet.py:302
 *   optimizer        scaler.step(opt)                  trainers/efficientnet.py:305-306, :487-491
 *   inference tail   softmax(dim=1) / argmax           orchestration/orchestrator.py:591-592
 *
 * Rules of the boundary ("B-inner"):
 *   - plain pointers and sizes only; no torch types.
 *   - every buffer (inputs, outputs, workspaces, partial-sum slabs) is allocated by
 *     the caller; the library never allocates, frees or keeps a pointer past return.
 *   - every call only enqueues work on `stream` and never synchronises; it is safe
 *     under hipGraph stream capture and re-entrant across host threads.
 *   - return value: DFD_OK or a negative DFD_E* code; no exceptions cross the ABI.
 *   - layouts: activations NHWC ([N*H*W][C] row-major) of dtype DFD_F32 or DFD_BF16,
 *     C % 8 == 0; depthwise weights [C][k][k] f32 (torch's [C,1,k,k]); pointwise
 *     weights [Cout][Cin]; BN statistics / coefficients / SE gates f32.
 *   - reductions over rows (BN statistics, weight gradients) are written as
 *     per-workgroup partial slabs and summed by a second kernel in a fixed order:
 *     results are bitwise reproducible run to run.
 */
#ifndef DFD_HIP_H
#define DFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dfd_stream;          /* a hipStream_t */

#define DFD_OK            0
#define DFD_EINVAL       (-1)      /* bad argument / shape not divisible as required */
#define DFD_EUNSUPPORTED (-2)      /* kernel size / stride / activation not implemented */
#define DFD_ELAUNCH      (-3)      /* hipGetLastError() != hipSuccess after the launch */
#define DFD_EWORKSPACE   (-4)      /* caller's workspace too small */

#define DFD_F32   0
#define DFD_BF16  1

#define DFD_ACT_NONE 0
#define DFD_ACT_SILU 1
#define DFD_ACT_RELU 2
#define DFD_ACT_GELU 3

/* upper bound on the number of partial-sum rows any kernel writes */
#define DFD_MAX_PARTIALS 1024

/* BN state: float[4][C] = scale, shift, mean, rstd  (z = scale*y + shift)
 * BN backward coefficients: float[3][C] = a, b, c   (dy = a*g + b*y + c)      */
#define DFD_BNSTATE_ROWS 4
#define DFD_BNCOEF_ROWS  3

/* ABI revision: 100 = first release; 101 = workspace arguments on the pooling entry points,
 * dfd_pool_ws, dfd_image_prep; 102 = dfd_prep_weights_multi, dfd_se_fc_fwd accepts a prepared w2t;
 * 110 = the EfficientFormerV2 / FasterViT set (section "token mixers" below), the *_ex BatchNorm entry
 * points (convolution bias and LayerScale folded into the BatchNorm coefficients), GELU in every
 * prologue, and the bookkeeping kernels (dfd_rand, dfd_step_tick, dfd_axpby, dfd_add);
 * 111 = dfd_se_fwd / dfd_se_bwd (squeeze-excite in two / three launches); dfd_rowtable_grad takes a workspace;
 * dfd_conv_fwd / dfd_conv_wgrad (implicit-GEMM dense convolution); dfd_sum_batch_begin / _end;
 * dfd_bn_eval_coeffs_multi;
 * 112 = the eval / inference form of the MBConv block: dfd_pwconv_fwd_eval, dfd_dwconv_fwd_eval(_tiles),
 * dfd_se_fwd_parts;
 * 120 = MX fp8 weights (dfd_mx_*), fused window attention on bf16 MFMA (dfd_wattn_*),
 * batched coordinate MLPs (dfd_coord_mlp_*_multi, dfd_relpos_bias_*_multi), dfd_dwconv_bwd_fused, dfd_resize_crop_u8;
 * 121 = dfd_bias_grad / dfd_bias_grad_ws; 122 = dfd_gemm_bias_act;
 * 130 = dfd_tune; the bf16 depthwise entry points run on the matrix cores where the shape allows (same signatures);
 * 131 = dfd_augment_u8, dfd_gemm_plan, dfd_dw_mm_plan, dfd_pwconv_bwd_fused;
 * 132 = dfd_pw_ntd_plan (mid-size 1x1 layers on the LDS-DMA ring kernel; same entry points), tune keys 4-7;
 * 133 = dfd_attn_scores / dfd_attn_apply; 134 = dfd_act_bn_bwd_se, dfd_sum_batch_end_deferred, dfd_sum_passengers_flush / _discard;
 * 135 = dfd_tune keys 8-13 (grids of the vector-unit depthwise kernels — their default changed, so partial-row counts did — and of the
 * tiled weight gradient); immediate partial-row sums of 33..256 rows in one launch (same order, same bits). */
int dfd_version(void);

/* Planner knobs (A/B switches and sizes the host-side kernel selection reads).  Process-wide plain ints: set them once at
 * start-up, before the first launch — they are not synchronised with concurrent callers.  Unknown key: DFD_EINVAL.
 *   0 DFD_TUNE_DW_MFMA    bit 0: the depthwise forward runs on the matrix cores (bf16, C % 16 == 0) for the shapes where that form
 *                         measured faster (5x5 stride 1 on maps of at most 64 pixels); bit 3: for every shape it can serve
 *                         (tests, A/B runs); 0 = the vector-unit kernels everywhere                          (default 1)
 *   1 DFD_TUNE_DW_LDS_KB  LDS budget of one matrix-core depthwise workgroup in KiB                           (default 156)
 *   2 DFD_TUNE_DW_GRID    workgroups a matrix-core depthwise launch aims for                                 (default 256)
 *   3 DFD_TUNE_DEBUG      timing-only ablations of the matrix-core depthwise kernels (results are WRONG when non-zero): bit 0
 *                         no activation arithmetic, 1 no tap loop, 2 no stores, 3 no loads; bits 8-12: the same for the LDS-DMA ring
 *                         kernel of dfd_pwconv_fwd (8 no prologue arithmetic, 9 no MFMAs, 10 no weight DMA, 11 no activation DMA,
 *                         12 no epilogue; scripts/pw_mid_ablate.py)                                         (default 0)
 *   4 DFD_TUNE_PW_NTD     bit 0: dfd_pwconv_fwd runs bf16 layers of 1 k .. 64 k rows on the LDS-DMA ring kernel (dfd_pw_ntd_plan
 *                         tells which); 0 = the register-staged tile kernel everywhere (A/B runs)             (default 1)
 *   5 DFD_TUNE_NTD_NS     stages of that kernel's LDS ring, 2..4; 0 = chosen from the LDS budget              (default 0)
 *   6 DFD_TUNE_NTD_MAXN   widest column tile of that kernel (A/B: 96 / 128); 0 = 192                          (default 0)
 *   7 DFD_TUNE_NTD_MINT   fewest 64-row tiles for which it is used: 16 measured best at batch 32 / 64 (4.40 -> 4.25 ms,
 *                         5.74 -> 5.64 ms per EfficientNet-B0 step), neutral at 256                          (default 16)
 *   8 / 9 / 10            workgroups a vector-unit depthwise launch aims for (all channel chunks together): forward / data gradient /
 *                         weight gradient.  1024 = what is co-resident at four per CU: every workgroup starts at once and walks
 *                         its work items, nobody queues behind a first round (measured against 768 / 1536 / 2048 / 3072 per
 *                         EfficientNet-B0 layer, scripts/dw_ab.py; 2048 was the value of rounds 2-4: B0 12.94 -> 12.75,
 *                         EfficientFormerV2-S1 17.00 -> 16.65 ms per step)                                    (default 1024)
 *   11                    fewest work slots per channel chunk of those launches                             (default 32)
 *   12 DFD_TUNE_TN_WGS    workgroups a tiled weight-gradient launch (k_pw_tn: output tiles x row splits) aims for.  512 = what is
 *                         co-resident at two per CU; measured 256 / 384 / 512 / 768 / 1024 on one box: EfficientNet-B0 12.88 / 12.80 /
 *                         12.62 / 13.03 / 12.96 ms, EfficientFormerV2-S1 16.71 / 16.57 / 16.44 / 16.94 / 17.11, FasterViT-0 22.04 /
 *                         21.46 / 19.59 / 20.27 / 20.20                                                       (default 512)
 *   13 DFD_TUNE_DWQ_WIDE  occupancy class of the vector-unit depthwise launches: -1 = by shape (5x5 stride-1 layers and the 3x3
 *                         stride-2 data gradient run THREE workgroups per CU with 48 KB of tile + tables and 3/4 of the grid
 *                         target, everything else four per CU with 39 KB), 0 = never wide, 1 = always wide (A/B)   (default -1) */
int dfd_tune(int key, int value);

/* Batched final summation of weight gradients.  The weight-gradient entry points whose result goes straight to the
 * optimizer (dfd_pwconv_wgrad, dfd_dwconv_bwd_weight, dfd_stem_conv_wgrad) end with a fixed-order sum of their workspace's
 * partial rows into dw.  (dfd_conv_wgrad, dfd_rowtable_grad and dfd_sum_rows always sum at once: their callers' next
 * launch reads the result.)  Between dfd_sum_batch_begin() and dfd_sum_batch_end() ON THE CALLING HOST THREAD these sums are
 * recorded instead of launched, and _end() (or the ninth recorded sum) adds them all up with one pair of launches —
 * same order, same bits.  Contract while a batch is open: each call gets its OWN workspace, which must stay untouched
 * until _end(); dw is valid only after _end(); all calls use one stream.  The state is thread-local: other threads are
 * unaffected (the ABI stays re-entrant).  begin inside an open batch / end without one: DFD_EINVAL.                   */
int dfd_sum_batch_begin(void);
int dfd_sum_batch_end(void);
/* dfd_sum_batch_end_deferred(): closes the batch like _end() but launches nothing — the two stages of the recorded sums ride along as
 * extra workgroups of the next two dfd_act_bn_bwd / dfd_act_bn_bwd_se launches on their stream (one per network block; the sums are read by
 * the optimizer only, so they need not sit on the backward pass's dependency chain), and dfd_sum_passengers_flush(stream) launches whatever
 * is still waiting.  Contract on top of _end()'s: the workspaces stay untouched, and dw is valid, only after the SECOND carrying launch
 * or the flush — callers keep three generations of workspaces and flush at the end of the backward pass (kernels.sum_batch).  A batch that
 * finds the previous one still waiting launches that one at once.  Process-wide state keyed by the stream.                          */
int dfd_sum_batch_end_deferred(void);
int dfd_sum_passengers_flush(dfd_stream stream);
/* drops every waiting batch without launching it (the first batch of a backward pass calls it: after a pass that ended in an exception
 * the waiting jobs point at memory the next pass may no longer own) */
int dfd_sum_passengers_discard(void);

/* ---------------------------------------------------------------- BatchNorm ---
 * F.batch_norm inside every conv-bn(-act) triple of the reference's modules
 * (efficientnet_pytorch MBConvBlock._bn0/_bn1/_bn2; timm BatchNormAct2d).      */

/* training mode: partial (sum, sumsq) slabs [nparts][2][C] -> bnstate, running stats */
int dfd_bn_finalize(const float* partials, int nparts, int C, double count,
                    const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* bnstate, dfd_stream stream);
/* eval mode: running stats -> bnstate */
int dfd_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* bnstate,
                       dfd_stream stream);
/* The same for every BatchNorm of a network at once (eval / inference forward: 49 launches of EfficientNet-B0 become 2).
 * jobs: host array, copied into kernel arguments; all pointers are device pointers, bnstate float[4][C] each.        */
typedef struct {
    const float *gamma, *beta, *conv_bias, *ls, *running_mean, *running_var;   /* gamma .. ls may be NULL */
    float* bnstate;
    float eps;
    int C;
} dfd_bn_eval_job;
int dfd_bn_eval_coeffs_multi(const dfd_bn_eval_job* jobs, int njobs, dfd_stream stream);
/* backward: partial (sum g, sum g*xhat) slabs -> dgamma, dbeta, coef.  train==0:
 * statistics were constants (eval-mode BN), so dy = gamma*rstd*g.               */
int dfd_bn_bwd_finalize(const float* partials, int nparts, int C, double count,
                        const float* gamma, const float* bnstate, int train,
                        float* dgamma, float* dbeta, int accumulate, float* coef,
                        dfd_stream stream);

/* The same three with the two per-channel vectors timm's ConvNorm / LayerScale2d add around a BatchNorm
 * (timm efficientformer_v2.py ConvNorm: conv(bias=True) -> BatchNorm2d; LayerScale2d: x * gamma):
 *   conv_bias [C] (optional): bias of the producing convolution.  It never enters the conv kernels: under
 *     batch statistics it cancels, so it only shifts the running mean (training) / the mean used (eval);
 *   ls_gamma  [C] (optional): LayerScale applied to the BatchNorm output, folded into (scale, shift).
 * Backward: partial sums are of dz = d loss / d (ls * BN(y)); dls / dbias are optional outputs.      */
int dfd_bn_finalize_ex(const float* partials, int nparts, int C, double count, const float* gamma,
                       const float* beta, const float* conv_bias, const float* ls_gamma,
                       float* running_mean, float* running_var, float momentum, float eps,
                       float* bnstate, dfd_stream stream);
int dfd_bn_eval_coeffs_ex(const float* gamma, const float* beta, const float* conv_bias,
                          const float* ls_gamma, const float* running_mean, const float* running_var,
                          float eps, int C, float* bnstate, dfd_stream stream);
int dfd_bn_bwd_finalize_ex(const float* partials, int nparts, int C, double count, const float* gamma,
                           const float* beta, const float* ls_gamma, const float* bnstate, int train,
                           float* dgamma, float* dbeta, float* dls, float* dbias, int accumulate,
                           float* coef, dfd_stream stream);

/* out = act(scale*y + shift) [* row_scale[n]] [+ residual]; all [N][HW][C].     */
int dfd_bn_act_apply(int dtype, const void* y, const float* bnstate, int act,
                     const void* residual, const float* row_scale, void* out,
                     int N, int HW, int C, dfd_stream stream);
/* partial sums of (g*rs, g*rs*xhat) per channel, xhat = (y-mean)*rstd            */
int dfd_bn_bwd_reduce(int dtype, const void* g, const void* y, const float* bnstate,
                      const float* row_scale, int N, int HW, int C,
                      float* partials, int pcap, int* nparts, dfd_stream stream);
/* Linear layer without statistics in ONE kernel (bf16, many rows, K % 64 == 0: csrc/dfd_gemm.hip; else DFD_EUNSUPPORTED and the
 * caller runs dfd_pwconv_fwd + dfd_bn_act_apply):  out = act(scale[n] * y + shift[n]) [* row_scale[row / HW]] [+ residual] with
 * y = bf16(a w^T) — the same arithmetic on the same rounded y as the two-kernel form, identical bits.  state: float[>=2][N]
 * (dfd_bn_eval_coeffs*: scale, shift = LayerScale and bias folded); act: DFD_ACT_NONE / DFD_ACT_GELU; yraw (optional, [M][N])
 * also receives y itself.  ABI 122.                                                                                          */
int dfd_gemm_bias_act(int dtype, const void* a, const void* w_nk, int M, int K, int N, const float* state, int act,
                      const void* residual, const float* row_scale, int HW, void* yraw, void* out, dfd_stream stream);
/* Rows per tile (256 or 128) with which the LDS-DMA product kernel (csrc/dfd_gemm.hip) serves a plain bf16 [M][K] x [N][K]^T
 * through dfd_pwconv_fwd / dfd_gemm_bias_act, 0 when the shape stays with the 128 x 128 kernel of dfd_pwconv.hip. */
int dfd_gemm_plan(int M, int K, int N);
/* Column-tile width (64 / 96 / 128 / 192) with which the LDS-DMA ring kernel serves a bf16 dfd_pwconv_fwd of this shape; 0: the
 * shape stays with the register-staged kernels (the prologue / epilogue combination can still decline at launch).            */
int dfd_pw_ntd_plan(int M, int K, int Nout);
/* dbias[c] (+)= sum over rows of g[row][c] (* row_scale[n]): the bias gradient of a Linear layer (no statistics).  ws:
 * dfd_bias_grad_ws bytes; its final fixed-order summation joins an open dfd_sum_batch like a weight gradient's.  ABI 121 */
size_t dfd_bias_grad_ws(int N, int HW, int C);
int dfd_bias_grad(int dtype, const void* g, const float* row_scale, int N, int HW, int C, float* dbias, int accumulate,
                  float* ws, size_t ws_bytes, dfd_stream stream);
/* dz = (D*gate[n,c] + dpool[n,c]/HW) * act'(scale*y+shift); writes dz and the partial
 * sums (dz, dz*xhat).  D, gate, dpool are each optional (NULL): D==NULL means the
 * incoming gradient is dpool/HW only (global-average-pool backward).             */
int dfd_act_bn_bwd(int dtype, const void* D, const void* y, const float* gate,
                   const float* dpool, const float* bnstate, int act, void* dz,
                   int N, int HW, int C, float* partials, int pcap, int* nparts,
                   dfd_stream stream);
/* dfd_act_bn_bwd with the squeeze-excite FC weight gradients of the same block riding along as extra workgroups of the launch:
 * dw1[R][C] = sum_n dh[n][r] pooled[n][c], db1, dw2[C][R] = sum_n g[n][c] h[n][r], db2, from dfd_se_bwd's workspace se_ws
 * (= [g: N x C | dh: N x R | h: N x R]) — what dfd_se_bwd's third launch computes when it is given dw1 / dw2 (pass NULL there).
 * Only AdamW reads these gradients, so they need not be a launch of their own on the block's dependency chain; same summation
 * order, same bits.  The workspace must stay untouched between dfd_se_bwd and this call (both on `stream`).                     */
int dfd_act_bn_bwd_se(int dtype, const void* D, const void* y, const float* gate, const float* dpool,
                      const float* bnstate, int act, void* dz, int N, int HW, int C, float* partials, int pcap,
                      int* nparts, const float* se_pooled, const float* se_ws, int R, float* dw1, float* db1, float* dw2,
                      float* db2, int accumulate, dfd_stream stream);
/* pooled[n,c] = mean_hw act(scale*y+shift): SE squeeze and the classifier's
 * global average pool.
 * ws (optional, may be NULL): dfd_pool_ws() bytes of scratch; lets the library split large
 * images over several workgroups (partial vectors, summed in a fixed order by a second
 * small kernel).  No initial contents required.                                          */
size_t dfd_pool_ws(int dtype, int N, int HW, int C);
int dfd_pool_act(int dtype, const void* y, const float* bnstate, int act, float* pooled,
                 int N, int HW, int C, void* ws, size_t ws_bytes, dfd_stream stream);
/* dgate[n,c] = sum_hw D * act(scale*y+shift)                                      */
int dfd_pool_bwd_reduce(int dtype, const void* D, const void* y, const float* bnstate,
                        int act, float* dgate, int N, int HW, int C, void* ws, size_t ws_bytes,
                        dfd_stream stream);
/* out = x * row_scale[n] (drop-connect on the gradient side)                      */
int dfd_scale_rows(int dtype, const void* x, const float* row_scale, void* out,
                   int N, int HW, int C, dfd_stream stream);

/* ------------------------------------------------------------ squeeze-excite ---
 * MBConvBlock._se_reduce/_se_expand (efficientnet_pytorch), SqueezeExcite (timm):
 * gate = sigmoid(W2 * act(W1*pooled + b1) + b2).                                 */
/* w2 is torch's [C][R]; w2t [R][C] is written by the forward (coalesced reads) and is
 * what the backward takes.  w2 == NULL: w2t was prepared by the caller (dfd_prep_weights_multi). */
int dfd_se_fc_fwd(const float* pooled, const float* w1, const float* b1, const float* w2,
                  const float* b2, int N, int C, int R, int act, float* hpre, float* gate,
                  float* w2t, dfd_stream stream);
/* ws: float[N*C + 2*N*R] scratch; R <= 128, C <= 4096 */
int dfd_se_fc_bwd(const float* dgate, const float* gate, const float* hpre,
                  const float* pooled, const float* w1, const float* w2t,
                  int N, int C, int R, int act, float* dpooled,
                  float* dw1, float* db1, float* dw2, float* db2, int accumulate,
                  float* ws, dfd_stream stream);
/* The whole squeeze-excite branch behind one call each way (reference call sites: the `x * self.se(x)` step of
 * timm's InvertedResidual.forward / efficientnet_pytorch's MBConvBlock.forward, reached from
 * trainers/efficientnet.py:297 `model(x)`): pooling kernel + ONE workgroup per image that adds the pooling
 * splits and runs both FC layers (forward: 2 launches instead of 4; backward: 3 instead of 5).
 *   forward : pooled = mean_hw act_in(scale*y+shift); hpre = W1 pooled + b1; gate = sigmoid(W2 act(hpre) + b2)
 *   backward: dgate = sum_hw D*act_in(scale*y+shift) -> dpooled, dw1, db1, dw2, db2 (as dfd_se_fc_bwd)
 * pool ws / ws_bytes as for dfd_pool_act; ws (backward) as for dfd_se_fc_bwd.  C % 4 == 0.               */
int dfd_se_fwd(int dtype, const void* y, const float* bnstate, int act_in, int N, int HW, int C,
               const float* w1, const float* b1, const float* w2, const float* b2, int R, int act,
               float* pooled, float* hpre, float* gate, float* w2t, void* ws, size_t ws_bytes,
               dfd_stream stream);
int dfd_se_bwd(int dtype, const void* D, const void* y, const float* bnstate, int act_in, int N, int HW,
               int C, const float* gate, const float* hpre, const float* pooled, const float* w1,
               const float* w2t, int R, int act, float* dgate, float* dpooled, float* dw1, float* db1,
               float* dw2, float* db2, int accumulate, void* pool_ws, size_t pool_ws_bytes, float* ws,
               dfd_stream stream);

/* ----------------------------------------------------------- depthwise conv ---
 * F.conv2d(groups=C) of MBConvBlock._depthwise_conv / timm conv_dw, k in {3,5},
 * stride in {1,2}, arbitrary (asymmetric "SAME") top/left padding.               */
typedef struct {
    int N, H, W, C;        /* input  [N][H][W][C]   */
    int Ho, Wo;            /* output [N][Ho][Wo][C] */
    int k, stride, pad_top, pad_left;
} dfd_dwconv_shape;

/* y = dwconv(act(scale*x+shift)) (in_bnstate==NULL: x used as is); optional stats */
int dfd_dwconv_fwd(int dtype, const void* x, const float* in_bnstate, int in_act,
                   const float* w, void* y, const dfd_dwconv_shape* s,
                   float* partials, int pcap, int* nparts, dfd_stream stream);
/* dy = a*dz + b*y + c (coef; coef==NULL: dy = dz).  da = dwconv^T(dy).
 * xin != NULL: dzin = da * act'(scale*xin+shift) and partial sums (dzin, dzin*xhat);
 * xin == NULL: dzin = da.                                                        */
int dfd_dwconv_bwd_data(int dtype, const void* dz, const void* y, const float* coef,
                        const float* w, const void* xin, const float* in_bnstate, int in_act,
                        void* dzin, const dfd_dwconv_shape* s,
                        float* partials, int pcap, int* nparts, dfd_stream stream);
/* dw[c][kh][kw] = sum dy * act(scale*xin+shift) (in_bnstate==NULL: xin as is)      */
int dfd_dwconv_bwd_weight(int dtype, const void* dz, const void* y, const float* coef,
                          const void* xin, const float* in_bnstate, int in_act,
                          float* dw, const dfd_dwconv_shape* s, int accumulate,
                          float* ws, size_t ws_bytes, dfd_stream stream);

/* Diagnostics: the tile plan the matrix-core depthwise forward would use for a shape (pro: the staging phase applies an
 * activation): out[12] = NI, TH, TW, runs, IH, IW, row pitch, plane pixels, work items, whole-image flag, LDS bytes in / out. */
int dfd_dw_mm_plan(const dfd_dwconv_shape* s, int pro, int* out);

/* ---- eval / inference form of the MBConv block -----------------------------------------------------------------
 * With running statistics every BatchNorm is an affine map known before the layer runs, so the PRODUCER can apply
 * its own BatchNorm + activation in the epilogue and store the activated tensor (training cannot: the batch
 * statistics only exist once the whole layer has run).  Replaces, for `model.eval()` forwards without autograd
 * (reference: orchestration/orchestrator.py:127-151 `class_probabilities`, evaluate.py:238-260, the validation loop
 * trainers/efficientnet.py:333-352), the raw-output + consumer-prologue chain of the training form:
 *   dfd_pwconv_fwd_eval : out = act(scale*(a @ w^T) + shift)               (the expand 1x1 convolution + bn1 + SiLU)
 *   dfd_dwconv_fwd_eval : y = act(scale*dwconv(x) + shift), and pool_parts[tile][N][C] = per-(tile, image) channel sums
 *                         of y as stored — the squeeze-excite pooling pass disappears
 *   dfd_se_fwd_parts    : dfd_se_fwd from those sums (mean = sum over tiles / HW)
 * out_bnstate: float[>=2][C] scale, shift (dfd_bn_eval_coeffs*).  out_act: DFD_ACT_SILU (the EfficientNet blocks);
 * anything else returns DFD_EUNSUPPORTED.  Values equal the training-form chain bit for bit except the pooled mean
 * (same addends, summed tile by tile instead of split by split).
 * dfd_dwconv_fwd_eval_tiles: rows of pool_parts the call will write (0 = invalid shape); ntiles returns the same. */
int dfd_pwconv_fwd_eval(int dtype, const void* a, const void* w, const float* out_bnstate, int out_act,
                        void* out, int M, int K, int Nout, dfd_stream stream);
int dfd_dwconv_fwd_eval_tiles(int dtype, const dfd_dwconv_shape* s);
int dfd_dwconv_fwd_eval(int dtype, const void* x, const float* w, const float* out_bnstate, int out_act, void* y,
                        const dfd_dwconv_shape* s, float* pool_parts, int* ntiles, dfd_stream stream);
int dfd_se_fwd_parts(const float* parts, int splits, int N, int HW, int C, const float* w1, const float* b1,
                     const float* w2, const float* b2, int R, int act, float* pooled, float* hpre, float* gate,
                     float* w2t, dfd_stream stream);
size_t dfd_dwconv_bwd_weight_ws(const dfd_dwconv_shape* s);
/* Data gradient AND weight gradient of a 3x3 stride-1 depthwise convolution in one kernel (dfd_dwconv_bwd_data with its
 * epilogue + dfd_dwconv_bwd_weight with its prologue, same arguments): dz / y / xin cross HBM once instead of twice.
 * ws as for dfd_dwconv_bwd_weight (dfd_dwconv_bwd_weight_ws bytes).  DFD_EUNSUPPORTED for other kernel sizes / strides:
 * callers then use the two separate entry points.                                                                      */
int dfd_dwconv_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const float* w, const void* xin,
                         const float* in_bnstate, int in_act, void* dzin, float* dw, const dfd_dwconv_shape* s,
                         float* partials, int pcap, int* nparts, int accumulate, float* ws, size_t ws_bytes,
                         dfd_stream stream);

/* ----------------------------------------------------------- pointwise conv ---
 * 1x1 convolutions (_expand_conv, _project_conv, _conv_head; timm conv_pw/conv_pwl/
 * conv_head) as MFMA GEMMs over the NHWC row matrix, with the producer's BN + act
 * (+ SE gate) applied to the A operand on the way in.                            */
#define DFD_PRO_NONE        0   /* a                                              */
#define DFD_PRO_BN_ACT      1   /* act(scale[k]*a + shift[k])                     */
#define DFD_PRO_BN_ACT_GATE 2   /* act(scale[k]*a + shift[k]) * gate[row/HW][k]   */
#define DFD_PRO_AFFINE2     3   /* ca[k]*a + cb[k]*a2 + cc[k]                     */
typedef struct {
    int mode;
    int act;
    int HW;                 /* rows per image (gate lookup) */
    int _pad;
    const void* a2;         /* AFFINE2: second operand, same shape/dtype as a */
    const float* coef;      /* BN_ACT*: bnstate [4][K]; AFFINE2: coef [3][K]  */
    const float* gate;      /* BN_ACT_GATE: [N][K] */
} dfd_prologue;

/* out[M][Nout] = P(a)[M][K] * w[Nout][K]^T (+ residual); w in `dtype`.
 * partials != NULL: (sum, sumsq) per output channel of the rounded output.       */
int dfd_pwconv_fwd(int dtype, const void* a, const dfd_prologue* pro, const void* w,
                   void* out, const void* residual, int M, int K, int Nout,
                   float* partials, int pcap, int* nparts, dfd_stream stream);
/* dw[Ni][Nj] = sum_m P(p)[m][i] * Q(q)[m][j]                                       */
int dfd_pwconv_wgrad(int dtype, const void* p, const dfd_prologue* pro_p, int Ni,
                     const void* q, const dfd_prologue* pro_q, int Nj, int M,
                     float* dw, int accumulate, float* ws, size_t ws_bytes,
                     dfd_stream stream);
/* The EXPAND 1x1 layer's backward in one pass over its widest operands (MBConvBlock._expand_conv / timm conv_pw):
 *   d = coef[0]*dz + coef[1]*y + coef[2]  (the BN-backward map, as the AFFINE2 prologue),  dx = d * w (+ residual),  dw = d^T * x.
 * dz, y [M][Cm]; x [M][Cin]; w_kn = the layer's weight in the activation dtype as [Cin][Cm] (the [K][N] copy dfd_pw_prep_weights /
 * dfd_prep_weights_multi write); residual [M][Cin] or NULL; dx [M][Cin]; dw f32 [Cm][Cin]; ws as for dfd_pwconv_wgrad(M, Cm, Cin).
 * Bit-identical to dfd_pwconv_fwd(dz, AFFINE2, w_kn, residual) + dfd_pwconv_wgrad(dz, AFFINE2, x) — (dz, y) cross HBM once
 * instead of twice.  bf16, Cin <= 32, Cm <= 144, M >= 196,608 (EfficientNet blocks 1-3 at the benchmark batch); anything else:
 * DFD_EUNSUPPORTED, the caller runs the two entry points. */
int dfd_pwconv_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const void* x, const void* w_kn,
                         const void* residual, int M, int Cm, int Cin, void* dx, float* dw, int accumulate, float* ws,
                         size_t ws_bytes, dfd_stream stream);
size_t dfd_pwconv_wgrad_ws(int M, int Ni, int Nj);
/* f32 master [N][K] -> w_nk [N][K] and w_kn [K][N] in `dtype` (either may be NULL) */
int dfd_pw_prep_weights(int dtype, const float* w, void* w_nk, void* w_kn, int N, int K,
                        dfd_stream stream);
/* Every derived weight of a network in one call (launched 32 jobs at a time): src f32 [N][K] -> nk ([N][K],
 * element type `dtype`) and / or kn ([K][N]); either destination may be NULL.  Also used for the [R][C] copy
 * of the squeeze-excite expand weight (dtype DFD_F32, kn only).  `jobs` is a HOST array.                   */
typedef struct dfd_prep_job {
    const float* src;
    void* nk;
    void* kn;
    int N, K;
    int dtype;
    int _pad;
} dfd_prep_job;
int dfd_prep_weights_multi(const dfd_prep_job* jobs, int njobs, dfd_stream stream);

/* ------------------------------------------------------------------- stem ---
 * _conv_stem / conv_stem: k x k stride-2 convolution on the 3-channel f32 image.  */
typedef struct {
    int N, H, W, Cout, Ho, Wo, k, stride, pad_top, pad_left;
} dfd_stem_shape;
int dfd_stem_conv_fwd(int dtype, const float* x, const float* w, void* y,
                      const dfd_stem_shape* s, float* partials, int pcap, int* nparts,
                      dfd_stream stream);
int dfd_stem_conv_wgrad(int dtype, const float* x, const void* dz, const void* y,
                        const float* coef, float* dw, const dfd_stem_shape* s,
                        int accumulate, float* ws, size_t ws_bytes, dfd_stream stream);
size_t dfd_stem_conv_wgrad_ws(const dfd_stem_shape* s);

/* ------------------------------------------------- classifier, loss, optimizer --- */
/* out = u >= p ? x/(1-p) : 0   (forward and, with x = grad, backward)              */
int dfd_dropout(const float* x, const float* u, float p, float* out, int n, dfd_stream stream);
int dfd_linear_fwd(const float* x, const float* w, const float* b, float* out,
                   int N, int K, int J, dfd_stream stream);
int dfd_linear_bwd(const float* dout, const float* x, const float* w, float* dx,
                   float* dw, float* db, int N, int K, int J, int accumulate,
                   dfd_stream stream);
/* nn.CrossEntropyLoss(label_smoothing) mean-reduced; dlogits (optional) is
 * d(loss*grad_scale)/dlogits.                                                     */
int dfd_ce_loss(const float* logits, const int64_t* targets, int N, int J,
                float label_smoothing, float grad_scale, float* row_loss, float* loss,
                float* dlogits, dfd_stream stream);
int dfd_softmax_argmax(const float* logits, int N, int J, float* probs, int64_t* preds,
                       dfd_stream stream);
/* Resize + CenterCrop / RandomResizedCrop of decoded uint8 RGB images on the device, bit-exact with Pillow's bilinear
 * Image.resize (two passes, 8-bit intermediate, anti-aliased when shrinking) — trainers/efficientnet.py:111-234,
 * orchestrator.py:316-347.  src: the batch's images, tightly packed HWC, one after the other in one buffer.  Per image:
 * out = window (cx, cy, OW x OH) of  crop(bx, by, bw, bh).resize((rw, rh)); pixels outside the resized image are 0.
 * jobs_dev: DEVICE array of N descriptors.  max_shrink: ceil of the largest bw/rw, bh/rh of the batch (<= 46).        */
typedef struct dfd_resize_job {
    long offset;            /* byte offset of the image in src                                  */
    int H, W;               /* source size                                                       */
    int bx, by, bw, bh;     /* box of the source that is resized                                 */
    int rw, rh;             /* size it is resized to                                             */
    int cx, cy;             /* origin of the output window in the resized image (may be < 0)    */
    int _pad[2];
} dfd_resize_job;
int dfd_resize_crop_u8(const unsigned char* src, const dfd_resize_job* jobs_dev, unsigned char* dst, int N, int OH, int OW,
                       int max_shrink, dfd_stream stream);
/* RandomRotation + ColorJitter on the device for a uint8 [N][H][W][3] batch (between dfd_resize_crop_u8 and dfd_image_prep): the
 * reference's DEFAULT 224-pixel training pipeline has both (trainers/efficientnet.py:173-181).  Byte-exact with the PIL pipeline of
 * deepfakedetection_amd/data.py (csrc/dfd_augment.hip restates Pillow's rotate / blend / HSV arithmetic; oracle/image_ref.py).
 * One job per picture, drawn by the host with the same distributions as the CPU transforms:
 *   mode   0 copy, 1 affine rotation with the 16.16 coefficients a[6] (data.rotate_plan), 2 / 3 / 4 rotation by 180 / 90 / 270 degrees
 *          (the last two for square pictures only);
 *   order  the permutation of (0 brightness, 1 contrast, 2 saturation, 3 hue) ColorJitter drew; enable bit k: operation k runs;
 *   fb, fc, fs the three blend factors; dh the hue shift in 8-bit hue units (0..255).
 * src != dst.  DFD_EUNSUPPORTED when H * W * 3 exceeds 156 KiB (the picture stays in one CU's LDS): callers keep the PIL pipeline. */
typedef struct {
    int mode;
    int a[6];
    int order[4];
    float fb, fc, fs;
    int dh;
    int enable;
} dfd_augment_job;   /* 64 bytes */
int dfd_augment_u8(const unsigned char* src, const dfd_augment_job* jobs_dev, unsigned char* dst, int N, int H, int W,
                   dfd_stream stream);

/* Input tail on the device (trainers/efficientnet.py:111-234): a uint8 NHWC
 * batch [N][H][W][3] -> RandomHorizontalFlip -> ToTensor (/255) -> Normalize((x-mean)/std) ->
 * RandomErasing(value 0) -> f32 NHWC, which is the stem kernel's input layout.  The random decisions
 * are made by the caller: flip[n] != 0 mirrors image n; erase[4n..] = {top, left, height, width},
 * height 0 = none; either pointer may be NULL.  mean3 / std3 are HOST arrays of 3 floats.      */
int dfd_image_prep(const unsigned char* src, float* dst, int N, int H, int W, const float* mean3,
                   const float* std3, const unsigned char* flip, const int* erase, dfd_stream stream);
/* torch.optim.AdamW step over a chunk table: int64 rows {param, grad, exp_avg,
 * exp_avg_sq, count}; hp = {lr, beta1, beta2, eps, weight_decay, 1-beta1^t,
 * 1-beta2^t, grad_scale} in device memory (so a captured graph sees new values).   */
#define DFD_ADAMW_TABLE_COLS 5
#define DFD_ADAMW_HP_LEN 8
int dfd_adamw_step(const int64_t* table, int nchunks, const float* hp, dfd_stream stream);

/* ------------------------------------------------------------ token mixers ---
 * What the EfficientFormerV2 (timm 1.0.20 efficientformer_v2.py: Attention2d, Attention2dDownsample,
 * LocalGlobalQuery, Downsample, Stem4) and FasterViT (fastervit 1.0.0 faster_vit.py: WindowAttention,
 * HAT, ConvBlock, PatchEmbed, LayerNorm) modules need beyond the MBConv kernels.  Reference call sites:
 * trainers/efficientformer_v2.py:244,215,369 and trainers/fastervit.py:271,235 (`model(inputs)`).       */

/* out = a[c]*dz + b[c]*y + c[c]  (BatchNorm input gradient, materialised)                                */
int dfd_affine2_apply(int dtype, const void* dz, const void* y, const float* coef, void* out, long rows,
                      int C, dfd_stream stream);
/* out = act(scale*y + shift + other); bnstate == NULL: y as is; other == NULL: no addend                 */
int dfd_bn_add_act(int dtype, const void* y, const float* bnstate, const void* other, int act, void* out,
                   long rows, int C, dfd_stream stream);
/* d = g * act'(scale*y + shift + other); partials (optional, needs bnstate): sums of (d, d*xhat)          */
int dfd_bn_add_act_bwd(int dtype, const void* g, const void* y, const float* bnstate, const void* other,
                       int act, void* d, long rows, int C, float* partials, int pcap, int* nparts,
                       dfd_stream stream);
/* per-channel (sum, sumsq) partial slabs [nparts][2][C] of a plain [rows][C] tensor                        */
int dfd_channel_stats(int dtype, const void* x, long rows, int C, float* partials, int pcap, int* nparts,
                      dfd_stream stream);
/* out[i] (+)= sum_p partials[p][i], fixed order; `partials` needs room for P + ceil(P/32) rows of L floats  */
int dfd_sum_rows(float* partials, int P, long L, float* out, int accumulate, dfd_stream stream);
/* dfd_sum_rows whose launch may be left to an open dfd_sum_batch_begin / _end (results nobody reads inside the batch:
 * parameter gradients summed straight into their destination)                                                    */
int dfd_sum_rows_deferred(float* partials, int P, long L, float* out, int accumulate, dfd_stream stream);
/* out [N][2h][2w][C] = act(bilinear_x2(s [N][h][w][C])), align_corners = False (nn.Upsample in Attention2d) */
int dfd_up2_act_fwd(int dtype, const void* s, int act, void* out, int N, int h, int w, int C, dfd_stream stream);
/* ws (optional): scratch of the size and type of g; with it the activation derivative is evaluated once per output
 * element (two launches) instead of once per interpolation tap                                                 */
int dfd_up2_act_bwd(int dtype, const void* g, const void* s, int act, void* ds, int N, int h, int w, int C,
                    void* ws, dfd_stream stream);
/* LocalGlobalQuery: out[n,i,j,:] = a[n,i,j,:] + bias[:] + x[n, i*stride, j*stride, :]  (AvgPool2d(1, stride))
 * and its gradient into x: dx[n, i*stride, j*stride, :] += g[n,i,j,:]                                        */
int dfd_subsample_add(int dtype, const void* a, const float* bias, const void* x, void* out, int N, int H,
                      int W, int stride, int C, dfd_stream stream);
int dfd_subsample_add_bwd(int dtype, const void* g, void* dx, int N, int H, int W, int stride, int C,
                          dfd_stream stream);

/* Small strided batched GEMM  C[b,h,m,n] = alpha * sum_k A[b,h,m,k] * B[b,h,k,n] (+ bias[h,m,n]).
 * Element (b, h, r, c) of an operand is at base + b*sb + h*sh + r*sr + c*sc (element units), so q / k / v
 * are read in place from the NHWC projection outputs and the result lands in NHWC.  dt_*: DFD_F32 / DFD_BF16.
 * round_a / round_b: round that operand to bf16 after loading (an f32 intermediate standing in for a tensor
 * a bf16 pipeline would have stored).  (M*(K+1) + K*N) * 4 bytes must fit in 150 KiB of LDS.               */
typedef struct { long sb, sh, sr, sc; } dfd_mat;
int dfd_bgemm(int dt_a, const void* A, const dfd_mat* sa, int dt_b, const void* B, const dfd_mat* sb,
              int dt_c, void* C, const dfd_mat* sc, const float* bias, float alpha, int nb, int nh, int M,
              int N, int K, int round_a, int round_b, dfd_stream stream);
/* Attention rows on S [B][H][Nq][Nk] f32:  [T1 = W1*S + b1 over heads ->] P = softmax_k(T1) [-> T2 = W2*P + b2]
 * th_w1 == NULL: no talking heads (T2 unused).  H <= 16, Nk <= 256.                                          */
int dfd_attn_softmax_fwd(const float* S, const float* th_w1, const float* th_b1, const float* th_w2,
                         const float* th_b2, float* P, float* T2, int B, int H, int Nq, int Nk,
                         dfd_stream stream);
/* dT2 -> dS (and dT1, the gradient at the softmax input, when talking heads are on)                           */
int dfd_attn_softmax_bwd(const float* dT2, const float* P, const float* th_w1, const float* th_w2,
                         float* dT1, float* dS, int B, int H, int Nq, int Nk, dfd_stream stream);
/* learned attention bias table [H][T] <-> full [H][L] through idx [L] (int32)                                 */
int dfd_bias_gather(const float* table, const int* idx, float* full, int H, int T, long L, dfd_stream stream);
int dfd_bias_scatter(const float* dfull, const int* idx, float* dtable, int H, int T, long L, int accumulate,
                     dfd_stream stream);

/* Dense k x k convolution, forward, as an implicit GEMM: the NT GEMM kernel gathers its A operand from the NHWC image
 * (row = output pixel, column = (tap, input channel)), applying the producer's BN + activation to the gathered values
 * (zero padding in the activated domain); nothing is materialised.  w_nk [Cout][k*k*C] in the activation dtype, column
 * (kh*k + kw)*C + c (dfd_conv_weight_perm + dfd_pw_prep_weights).  C % 8 == 0, Cout % 8 == 0.  Output, partials and
 * nparts as dfd_pwconv_fwd.  Reference call sites: the 3x3 convolutions of timm's EfficientFormerV2 stem and of
 * NVlabs FasterViT's PatchEmbed / ConvBlock / Downsample, reached from trainers/<model>.py `model(x)`.                  */
int dfd_conv_fwd(int dtype, const void* x, const dfd_dwconv_shape* s, const float* in_bnstate, int in_act,
                 const void* w_nk, int Cout, void* y, float* partials, int pcap, int* nparts, dfd_stream stream);

/* Weight gradient of the same convolution: dw[Cout][k*k*C] (f32, GEMM column order) = sum over output pixels of
 * P(p)[m][co] * im2col(act(bn(x)))[m][(tap, c)], the im2col operand gathered inside the TN kernel.  p / pro_p as for
 * dfd_pwconv_wgrad (DFD_PRO_NONE or the BN-backward map DFD_PRO_AFFINE2); ws: dfd_conv_wgrad_ws() bytes.            */
size_t dfd_conv_wgrad_ws(const dfd_dwconv_shape* s, int Cout);
int dfd_conv_wgrad(int dtype, const void* p, const dfd_prologue* pro_p, int Cout, const void* x,
                   const dfd_dwconv_shape* s, const float* in_bnstate, int in_act, float* dw, int accumulate,
                   float* ws, size_t ws_bytes, dfd_stream stream);

/* Dense k x k convolution = im2col + the 1x1 GEMM entry points (data gradient; forward: dfd_conv_fwd).  Shapes use dfd_dwconv_shape (C = input
 * channels).  col [N*Ho*Wo][k*k*C] with column index (kh*k + kw)*C + c; zero padding in the activated domain. */
int dfd_im2col(int dtype, const void* x, const float* in_bnstate, int in_act, void* col,
               const dfd_dwconv_shape* s, dfd_stream stream);
int dfd_col2im(int dtype, const void* dcol, void* dx, const dfd_dwconv_shape* s, dfd_stream stream);
/* torch's [O][I][k][k] f32 <-> the GEMM's [O][(kh,kw,i)]; to_gemm = 0 maps a weight gradient back; to_gemm = 2 writes
 * [I][(flipped tap, o)]: with it the data gradient of a stride-1 convolution is dfd_conv_fwd on the output gradient   */
int dfd_conv_weight_perm(const float* src, float* dst, int O, int I, int k, int to_gemm, int accumulate,
                         dfd_stream stream);

/* LayerNorm over C of [rows][C]; stats [rows][2] = (mean, rstd).  Backward partials [nparts][2][C]
 * (row 0: dgamma, row 1: dbeta), summed by dfd_sum_rows.                                                      */
int dfd_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y,
                      float* stats, long rows, int C, dfd_stream stream);
int dfd_layernorm_bwd(int dtype, const void* g, const void* x, const float* gamma, const float* stats,
                      const void* residual, void* dx, float* partials, int pcap, int* nparts, long rows, int C,
                      dfd_stream stream);

/* Token bookkeeping of windowed attention (fastervit faster_vit.py window_partition / window_reverse /
 * ct_dewindow / ct_window / cat / split): dst[didx[r]] = src[sidx[r]], r < n, rows of C elements; a NULL index
 * array is the identity.  For the one-to-one maps above the backward pass is the same call with the arrays swapped. */
int dfd_copy_rows(int dtype, const void* src, const int* sidx, void* dst, const int* didx, long n, int C,
                  dfd_stream stream);
/* PosEmbMLPSwinv1D: out[r] = x[r] + table[r % T] (table f32 [T][C]); dtable[t] = sum_w g[w*T + t]              */
int dfd_add_rowtable(int dtype, const void* x, const float* table, void* out, long rows, int T, int C,
                     dfd_stream stream);
size_t dfd_rowtable_grad_ws(int T, int C);       /* bytes of scratch for dfd_rowtable_grad (no initial contents required) */
int dfd_rowtable_grad(int dtype, const void* g, float* dtable, long rows, int T, int C, int accumulate,
                      float* ws, size_t ws_bytes, dfd_stream stream);
/* nn.AvgPool2d(k, stride) without padding on NHWC (TokenInitializer) and its gradient (dx [N][H][W][C])         */
int dfd_avgpool_fwd(int dtype, const void* x, void* out, int N, int H, int W, int k, int stride, int C,
                    dfd_stream stream);
int dfd_avgpool_bwd(int dtype, const void* g, void* dx, int N, int H, int W, int k, int stride, int C,
                    dfd_stream stream);
/* PosEmbMLPSwinv2D: full [H][S][S], S = n_local + n_global: 16*sigmoid(table[idx[i*n_local+j]][h]) in the local
 * block, zeros in the first n_global rows / columns; table f32 [T][H]; backward through the sigmoid into dtable  */
int dfd_relpos_bias_fwd(const float* table, const int* idx, float* full, int H, int n_local, int n_global,
                        dfd_stream stream);
int dfd_relpos_bias_bwd(const float* dfull, const float* table, const int* idx, float* dtable, int H, int T,
                        int n_local, int n_global, dfd_stream stream);

/* ---------------------------------------------------------------- bookkeeping ---
 * Small kernels that keep ATen off the model path (drop-connect / dropout randoms, counters, scaling).        */
/* out = (a_dev ? a*a_dev[0] : a) * x + b * y   (y may be NULL), f32                                           */
int dfd_axpby(const float* x, const float* y, float a, float b, const float* a_dev, float* out, long n,
              dfd_stream stream);
/* out = a + b, n elements of `dtype`, n % 8 == 0                                                               */
int dfd_add(int dtype, const void* a, const void* b, void* out, long n, dfd_stream stream);
/* Philox4x32-10 uniforms.  rng_state (device): {seed, offset}; counter = (offset, stream_id, index/4).
 * keep <= 0: out = U[0,1);  keep in (0,1]: out = floor(keep + u) / keep  (per-sample drop-path scale)          */
int dfd_rand(const uint64_t* rng_state, uint32_t stream_id, float keep, float* out, long n, dfd_stream stream);
/* once per forward pass: *counter_ptrs[i] += 1 (BatchNorm num_batches_tracked, int64) and rng offset += 1.
 * counter_ptrs is a DEVICE array of ncounters addresses.                                                       */
int dfd_step_tick(const int64_t* counter_ptrs, int ncounters, uint64_t* rng_state, dfd_stream stream);

/* ------------------------------------------------------- MX fp8 (FasterViT fp8 weights) ---
 * BASELINE config 5 ("FasterViT-0 bf16/fp8 weights ... on CDNA4 fp8 MFMA"); carries the Linear layers (qkv / proj /
 * fc1 / fc2) of the third-party module's forward at trainers/fastervit.py:271 (training), :235 (evaluate) and
 * orchestration/orchestrator.py:529,590 (inference) when `fp8_weights` is switched on.
 * Format: OCP microscaling FP8 — e4m3fn elements, one E8M0 scale byte (2^(b-127)) per 32 consecutive K elements:
 *   e = clamp(floor(log2(max|v|)) - 8, -127, 127);  q = e4m3_rne_sat(v * 2^-e);  scale byte = e + 127.
 * The GEMM is v_mfma_scale_f32_16x16x128_f8f6f4 (the one fp8 form above the bf16 MFMA rate); it takes BOTH operands in this
 * format, so activations are quantised the same way (dfd_mx_quant_rows) right before the product.  K % 128 == 0.          */
typedef struct dfd_mx_job {
    const float* src;      /* f32 master weight [N][K]                                                       */
    uint8_t* q;            /* e4m3fn [N][K]                                                                  */
    uint8_t* scale;        /* e8m0 [N][K/32]                                                                 */
    void* kn;              /* optional: DEQUANTISED weight, transposed [K][N], element type kn_dtype
                              (what the bf16 data-gradient GEMM multiplies by)                               */
    int N, K;
    int kn_dtype;
    int _pad;
} dfd_mx_job;
/* all fp8 weights of a network in one call (32 jobs per launch); `jobs` is a HOST array */
int dfd_mx_quant_weights_multi(const dfd_mx_job* jobs, int njobs, dfd_stream stream);
/* a [M][K] (dtype) [-> act(coef[0][k]*a + coef[1][k]) rounded to dtype: prologue modes DFD_PRO_NONE / DFD_PRO_BN_ACT]
 * -> q e4m3fn [M][K], scale e8m0 [M][K/32]                                                                              */
int dfd_mx_quant_rows(int dtype, const void* a, const dfd_prologue* pro, uint8_t* q, uint8_t* scale, long M, int K,
                      dfd_stream stream);
/* out[M][N] (dtype_out) = dequant(aq, ascale) . dequant(wq, wscale)^T, f32 accumulation; wq [N][K], N % 4 == 0       */
int dfd_mx_gemm(const uint8_t* aq, const uint8_t* ascale, const uint8_t* wq, const uint8_t* wscale, int dtype_out,
                void* out, long M, int K, int N, dfd_stream stream);

/* ------------------------------------------------ fused window attention (FasterViT) ---
 * WindowAttention of the third-party module (forward at trainers/fastervit.py:271, backward at :274), bf16, head_dim 32,
 * T <= 64 tokens per window:  o = softmax_k(scale * q k^T + bias[h]) v  with q / k / v read in place from the qkv projection
 * output [n][T][3*H*32] and o written as [n][T][H*32].  One wave per (window, head) on v_mfma_f32_16x16x32_bf16; S and P stay
 * in registers.  L [n][H][T] f32 = row max + log(row sum) (saved for the backward, which recomputes P).  bias f32 [H][T][T]
 * or NULL.  Backward writes dqkv [n][T][3*H*32] and, when dbias_parts != NULL, dfd_wattn_parts(n) rows of [H][T][T] f32
 * (sum over 4 windows each) for dfd_sum_rows.  DFD_EUNSUPPORTED for other head dimensions / longer windows (callers then
 * use dfd_bgemm + dfd_attn_softmax_*).                                                                                  */
int dfd_wattn_parts(int n);
int dfd_wattn_fwd(const void* qkv, const float* bias, void* out, float* L, int n, int T, int H, int hd, float scale,
                  dfd_stream stream);
int dfd_wattn_bwd(const void* qkv, const void* dout, const float* L, const float* bias, void* dqkv, float* dbias_parts,
                  int n, int T, int H, int hd, float scale, dfd_stream stream);

/* ------------------------------------------------ attention GEMMs of Attention2d (EfficientFormerV2) ---
 * timm's Attention2d (forward at trainers/efficientformer_v2.py:244) mixes the heads of a (query, key) pair before and after the
 * softmax ("talking heads"), so S / P / T2 stay f32 [B][H][Nq][Nk] tensors for dfd_attn_softmax_*; the six batched products
 * around them run here on v_mfma_f32_16x16x32_bf16, one wave per (image, head), instead of on dfd_bgemm:
 *   dfd_attn_scores  out[b][h][i][j] = alpha * sum_d x[b][i][h*D + d] * y[b][j][h*D + d] (+ bias[h][i][j])
 *                    x [n][Tx][H*D], y [n][Ty][H*D] bf16; out f32 [n][H][Tx][Ty]; bias f32 [H][Tx][Ty] or NULL         (S, dT2)
 *   dfd_attn_apply   out[b][i][h*D + d] = alpha * sum_t f[b][h][i][t] * x[b][t][h*D + d]      (f_trans = 0, f [n][H][To][Tc])
 *                                       = alpha * sum_t f[b][h][t][i] * x[b][t][h*D + d]      (f_trans = 1, f [n][H][Tc][To])
 *                    f f32 (rounded to bf16 for the product, f32 accumulation); x [n][Tc][H*D], out [n][To][H*D] bf16  (O, dQ | dV, dK)
 * Token counts <= 256 (walked in blocks of 64), D % 8 == 0, D <= 128; anything else: DFD_EUNSUPPORTED (callers keep dfd_bgemm).     */
int dfd_attn_scores(const void* x, const void* y, float* out, const float* bias, float alpha, int n, int H, int Tx, int Ty,
                    int D, dfd_stream stream);
int dfd_attn_apply(const float* f, int f_trans, const void* x, void* out, float alpha, int n, int H, int To, int Tc, int D,
                   dfd_stream stream);

/* -------------------------------------------- batched coordinate MLPs (FasterViT) ---
 * PosEmbMLPSwinv1D / PosEmbMLPSwinv2D of the third-party module:  table[t][:] = W2 . relu(W0 . coords[t] + b0), f32,
 * coords [T][2] constant, W0 [Hd][2], b0 [Hd], W2 [D][Hd] (Hd = 512).  They depend on parameters only, so a whole level's
 * tables are computed by one call before its first block and differentiated by one call after its last
 * (2 + 2 launches instead of ~280 per step).  `jobs` are HOST arrays (copied into kernel arguments, 24 per launch).
 * forward reads coords / w0 / b0 / w2 and writes table [T][D]; backward reads dtable [T][D] too and writes the non-NULL ones
 * of dw0 [Hd][2], db0 [Hd], dw2 [D][Hd] (overwriting).  T <= 176, Hd <= 1024, Hd % 4 == 0.                               */
typedef struct dfd_cmlp_job {
    const float *coords, *w0, *b0, *w2;
    float* table;
    const float* dtable;
    float *dw0, *db0, *dw2;
    int T, D, Hd, _pad;
} dfd_cmlp_job;
int dfd_coord_mlp_fwd_multi(const dfd_cmlp_job* jobs, int njobs, dfd_stream stream);
int dfd_coord_mlp_bwd_multi(const dfd_cmlp_job* jobs, int njobs, dfd_stream stream);
/* dfd_relpos_bias_fwd / _bwd for every attention layer of a level at once: table [T][H], idx int32 [n_local^2],
 * full / dfull [H][S][S] with S = n_local + n_global, dtable [T][H].                                                     */
typedef struct dfd_relpos_job {
    const float* table;
    const int* idx;
    float* full;
    const float* dfull;
    float* dtable;
    int H, T, n_local, n_global;
} dfd_relpos_job;
int dfd_relpos_bias_fwd_multi(const dfd_relpos_job* jobs, int njobs, dfd_stream stream);
int dfd_relpos_bias_bwd_multi(const dfd_relpos_job* jobs, int njobs, dfd_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* DFD_HIP_H */
