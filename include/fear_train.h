/*
 * fear_train.h — C ABI of the MI355X (gfx950) operators of the FEAR head TRAINING step (SURVEY.md §8f N3,
 * BASELINE.json configs[4] "training step: backbone + xcorr fwd/bwd"; first slice = BoxTower + FEARLoss).
 *
 * What the reference does with torch autograd + cuDNN is spelled out here as explicit forward / backward operators:
 *     SepConv (depthwise 3x3 -> pointwise 1x1)         model_training/model/blocks.py:45-72
 *     nn.BatchNorm2d in training mode (+ ReLU)         model/blocks.py:98-101, 115-119, 150-158
 *     MobileCorrelation  s = z^T x, cat[x, s]          model/blocks.py:121-126
 *     exp(adjust * x + bias), 0.1 * cls                model/blocks.py:186-192
 *     FEARLoss (BCE-with-logits halves + 1 - IoU)      model_training/train/loss.py:13-96
 * feartracker_amd/train_head.py composes them into `BoxTower.forward` + loss + backward (host code stays Python, like the
 * reference's Lightning step train/fear_lightning_model.py:60-66); gradients of all ranks are averaged by ONE RCCL
 * all-reduce of the flat gradient buffer (the reference: Lightning DDP, train/trainer.py:50-52).
 *
 * Conventions: plain C; every tensor pointer is a DEVICE pointer, fp32, "rows x channels" NHWC ([M = batch*H*W][ld], channels
 * contiguous, ld = row stride in floats) unless it says NCHW; `stream` is a hipStream_t passed as void*; calls are
 * asynchronous on it; 0 on success, negative FEAR_TRAIN_ERR_* otherwise.  `workspace` is caller-owned scratch of at least
 * fear_train_workspace_bytes(rows, max_channels) bytes; every reduction is two-stage in a fixed order (no atomics).
 */
#ifndef FEAR_TRAIN_H
#define FEAR_TRAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEAR_TRAIN_OK 0
#define FEAR_TRAIN_ERR_NULL (-1)
#define FEAR_TRAIN_ERR_SHAPE (-2)
#define FEAR_TRAIN_ERR_HIP (-4)
#define FEAR_TRAIN_ERR_WORKSPACE (-7)   /* workspace missing or too small */
#define FEAR_TRAIN_ERR_SYNC (-8)        /* the SyncBatchNorm all-reduce callback failed, or its buffer is too small (fear_train_sync_bind) */

size_t fear_train_workspace_bytes(long rows, int max_channels);

/* Layout rules for every operator below: activations are row-major [rows][ld] fp32 with 16-byte-aligned bases; channel
 * counts and leading dimensions are multiples of 4 floats (rows are read and written as float4s), ld >= the row's width;
 * violations return FEAR_TRAIN_ERR_SHAPE. */
/* nn.Conv2d(K, N, 1): y[m][n] = sum_k x[m][k] w[n][k] (+ bias[n]); K, N multiples of 4 (v_mfma_f32_16x16x4_f32) */
int fear_pw_forward(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, long M, int K, int N,
                    void* stream);
/* its input gradient dx[m][k] = (add ? add[m][k] : 0) + sum_n dy[m][n] w[n][k] */
int fear_pw_backward_data(const float* dy, int lddy, const float* w, const float* add, int ldadd, float* dx, int lddx, long M,
                          int K, int N, void* stream);
/* its weight gradient dw[n][k] = sum_m dy[m][n] x[m][k] (MFMA, reduction over the rows); N, K, lddy, ldx multiples of 4 */
int fear_pw_backward_weight(const float* dy, int lddy, const float* x, int ldx, float* dw, float* workspace, size_t ws_bytes,
                            long M, int K, int N, void* stream);
/* bias gradients: out[c] = sum_m dy[m][c] */
int fear_col_sum(const float* dy, int lddy, float* out, float* workspace, size_t ws_bytes, long M, int C, void* stream);

/* nn.Conv2d(C, C, k, groups=C, padding=k/2, stride=stride), k in {3,5}, stride in {1,2}: taps laid out [k*k][C] (tap-major) */
int fear_dw_forward(const float* x, int ldx, const float* w_taps, const float* bias, float* y, int ldy, int B, int H, int W,
                    int C, int k, int stride, void* stream);
/* input gradient (B, H, W = the INPUT map; dy is the (H/stride, W/stride) output-side gradient) */
int fear_dw_backward_data(const float* dy, int lddy, const float* w_taps, float* dx, int lddx, int B, int H, int W, int C, int k,
                          int stride, void* stream);
/* weight gradient dw_taps[t][c] = sum_{b,oy,ox} dy[b,oy,ox,c] * x[b, oy*stride+ky-k/2, ox*stride+kx-k/2, c] */
int fear_dw_backward_weight(const float* dy, int lddy, const float* x, int ldx, float* dw_taps, float* workspace,
                            size_t ws_bytes, int B, int H, int W, int C, int k, int stride, void* stream);

/* stem conv 3x3 stride 2 pad 1 (3 -> 16) as a GEMM: im2col of an NCHW image batch into rows of 28 floats
 * (k = (ci*3+ky)*3+kx, column 27 zero) -> fear_pw_forward / fear_pw_backward_weight with K = 28 */
int fear_stem_im2col(const float* x_nchw, float* rows28, long n, int H, int W, void* stream);

/* nn.BatchNorm2d(C) in training mode (+ optional ReLU): batch statistics over the M rows; mean / rstd are saved for the
 * backward; running_mean / running_var (may be NULL) follow torch: (1 - momentum) * running + momentum * stat, unbiased var */
int fear_bn_train_forward(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* mean,
                          float* rstd, float* running_mean, float* running_var, double momentum, double eps, long M, int C,
                          int relu, float* workspace, size_t ws_bytes, void* stream);
/* backward through (ReLU o BatchNorm): y_act = the forward output when a ReLU followed (its mask), else NULL */
int fear_bn_train_backward(const float* dy, int lddy, const float* y_act, int ldy, const float* x, int ldx, const float* mean,
                           const float* rstd, const float* gamma, float* dx, int lddx, float* dgamma, float* dbeta, long M,
                           int C, float* workspace, size_t ws_bytes, void* stream);

/* SyncBatchNorm (the reference's multi-GPU backends train with sync_bn: True, config/backend/{2,4}gpu.yaml -> trainer.py:52):
 * the same BatchNorm with its two reductions exposed, so that the ranks can add their float64 sums ([2][C] doubles on the
 * device: forward sum x | sum x^2, backward sum g | sum g * xhat with g = the ReLU-masked dy) with one all-reduce each:
 *   forward : fear_bn_reduce -> all-reduce(sums) -> fear_bn_forward_from_sums(count = rows of all ranks)
 *   backward: fear_bn_backward_reduce -> copy = local sums, all-reduce(sums) -> fear_bn_backward_from_sums
 * d gamma / d beta come from the LOCAL sums (they are averaged with every other gradient), dx from the global ones — the
 * split torch.nn.SyncBatchNorm makes.  With one rank the pair equals fear_bn_train_forward / _backward. */
int fear_bn_reduce(const float* x, int ldx, double* sums, long M, int C, float* workspace, size_t ws_bytes, void* stream);
int fear_bn_forward_from_sums(const float* x, int ldx, const double* sums, double count, const float* gamma, const float* beta,
                              float* y, int ldy, float* mean, float* rstd, float* running_mean, float* running_var,
                              double momentum, double eps, long M, int C, int relu, void* stream);
int fear_bn_backward_reduce(const float* dy, int lddy, const float* y_act, int ldy, const float* x, int ldx, const float* mean,
                            const float* rstd, double* sums, long M, int C, float* workspace, size_t ws_bytes, void* stream);
int fear_bn_backward_from_sums(const float* dy, int lddy, const float* y_act, int ldy, const float* x, int ldx, const float* mean,
                               const float* rstd, const float* gamma, const double* sums_all, double count,
                               const double* sums_local, float* dx, int lddx, float* dgamma, float* dbeta, float* workspace,
                               size_t ws_bytes, long M, int C, void* stream);

/* ---- fused conv + BatchNorm training operators (the trunk's step; model/blocks.py:27-35 over mobile_cv's conv-BN-ReLU units) ----
 * A BatchNorm'd activation is never written out.  A PRODUCER writes the convolution's raw output y and, from the same pass, the
 * float64 column sums of it (sums = [2][C]: sum y | sum y^2 — exactly what SyncBatchNorm all-reduces); fear_bn_finalize turns the
 * (possibly all-reduced) sums into mean / rstd / running statistics and the affine a = gamma * rstd, b = beta - mean * a; every
 * CONSUMER applies act(x) = fma(x, a, b) [then max(., 0)] to the raw tensor as it loads it: in_a / in_b / in_relu below (in_a = NULL:
 * the input is used as it is).  The ReLU mask of the backward is recomputed from the raw tensor with the same expression, so the
 * forward and the backward agree bit for bit on which elements are active.  Zero padding of a depthwise conv pads the
 * ACTIVATION (stays zero).  fear_bn_act materialises act(x) [+ residual] where a tensor is needed (block outputs).
 * Against the unfused operators above: 11 instead of 16 passes over every saved tensor, half the saved memory. */
int fear_pw_forward_stats(const float* x, int ldx, const float* in_a, const float* in_b, int in_relu, const float* w, float* y,
                          int ldy, long M, int K, int N, double* sums, float* workspace, size_t ws_bytes, void* stream);
int fear_dw_forward_stats(const float* x, int ldx, const float* in_a, const float* in_b, int in_relu, const float* w_taps, float* y,
                          int ldy, int B, int H, int W, int C, int k, int stride, double* sums, float* workspace, size_t ws_bytes,
                          void* stream);
/* bytes of workspace the two producers need for `rows` output rows of `channels` channels (partial sums per workgroup) */
size_t fear_train_stats_workspace_bytes(long rows, int channels);
int fear_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float* mean, float* rstd, float* a_out,
                     float* b_out, float* running_mean, float* running_var, double momentum, double eps, int C, void* stream);
int fear_bn_act(const float* x, int ldx, const float* a, const float* b, int relu, const float* residual, int ldr, float* y, int ldy,
                long M, int C, void* stream);
/* backward through (ReLU? o BatchNorm) of a raw tensor x: sums = [2][C] float64 (sum g | sum g * xhat, g = dy where act(x) > 0 when
 * relu); then dx from the (all ranks') sums and row count, d gamma / d beta from the local sums — the split of
 * fear_bn_backward_reduce / _from_sums above, with the mask taken from x instead of a stored activation */
int fear_bn_backward_reduce_x(const float* dy, int lddy, const float* x, int ldx, const float* act_a, const float* act_b, int relu,
                              const float* mean, const float* rstd, double* sums, long M, int C, float* workspace, size_t ws_bytes,
                              void* stream);
int fear_bn_backward_apply_x(const float* dy, int lddy, const float* x, int ldx, const float* act_a, const float* act_b, int relu,
                             const float* mean, const float* rstd, const float* gamma, const double* sums_all, double count,
                             const double* sums_local, float* dx, int lddx, float* dgamma, float* dbeta, float* workspace,
                             size_t ws_bytes, long M, int C, void* stream);
/* the layer-wise step's BatchNorm in the same affine form, activation written out (y = act(x) [+ residual]), and its backward with
 * the ReLU mask recomputed from x — the stored activation is not read on the way back; three launches each, one rank */
int fear_bn_train_forward_ab(const float* x, int ldx, const float* gamma, const float* beta, int relu, const float* residual, int ldr,
                             float* y, int ldy, float* mean, float* rstd, float* a_out, float* b_out, float* running_mean,
                             float* running_var, double momentum, double eps, long M, int C, float* workspace, size_t ws_bytes,
                             void* stream);
int fear_bn_train_backward_x(const float* dy, int lddy, const float* x, int ldx, const float* act_a, const float* act_b, int relu,
                             const float* mean, const float* rstd, const float* gamma, float* dx, int lddx, float* dgamma, float* dbeta,
                             long M, int C, float* workspace, size_t ws_bytes, void* stream);
/* weight gradients whose x operand is act(raw x) applied on load */
int fear_pw_backward_weight_act(const float* dy, int lddy, const float* x, int ldx, const float* in_a, const float* in_b, int in_relu,
                                float* dw, float* workspace, size_t ws_bytes, long M, int K, int N, void* stream);
int fear_dw_backward_weight_act(const float* dy, int lddy, const float* x, int ldx, const float* in_a, const float* in_b, int in_relu,
                                float* dw_taps, float* workspace, size_t ws_bytes, int B, int H, int W, int C, int k, int stride,
                                void* stream);

/* MobileCorrelation (blocks.py:121-123): s[b][p][j] = sum_c x[b][p][c] z[b][c][j]; z_nchw = (B, C, J) as the reference holds it */
int fear_xcorr_forward(const float* x, int ldx, const float* z_nchw, float* s_out, int lds, int B, int P, int C, int J,
                       void* stream);
/* dx[b][p][c] = (dx_add ? dx_add[b][p][c] : 0) + sum_j ds[b][p][j] z[b][c][j];   dz[b][c][j] = sum_p x[b][p][c] ds[b][p][j] */
int fear_xcorr_backward(const float* ds, int ldds, const float* x, int ldx, const float* z_nchw, const float* dx_add, int ldadd,
                        float* dx, int lddx, float* dz_nchw, int B, int P, int C, int J, void* stream);

/* bbox = exp(adjust * p + bias[c]) on rows of 4 (blocks.py:186-187), and its backward (dp, d adjust, d bias) */
int fear_exp_head_forward(const float* p, const float* adjust, const float* bias4, float* bbox, long M, void* stream);
int fear_exp_head_backward(const float* p, const float* adjust, const float* bbox, const float* dbbox, float* dp, float* dadjust,
                           float* dbias4, float* workspace, size_t ws_bytes, long M, void* stream);

/* FEARLoss forward + gradient (train/loss.py:45-96): bbox / gt_reg rows of 4 (ltrb), cls / gt_cls / gt_weight one value per row.
 * losses2 = {classification, regression} (each times its coefficient); dbbox / dcls = d(sum of both) / d(bbox, cls).
 * Selections of one or no cell: the reference indexes with `label.eq(1).nonzero().squeeze()` (loss.py:77-78), so exactly ONE
 * positive (or one negative) cell makes `_weighted_cls_loss` return a constant 0 for that half — mirrored: no loss, no gradient.
 * With NO positive / negative / weighted cell torch yields NaN (mean over nothing); this operator yields 0 for that term — the
 * one deliberate deviation (a NaN would poison the all-reduced gradient buffer of every rank).                                */
int fear_head_loss(const float* bbox, const float* cls, const float* gt_reg, const float* gt_cls, const float* gt_weight,
                   float coef_cls, float coef_reg, float* losses2, float* dbbox, float* dcls, float* workspace, size_t ws_bytes,
                   long M, void* stream);

/* boundary layout changes (the reference's tensors are NCHW): out[(b*HW + p)*ld_out + ch_off + c] = in[(b*C + c)*HW + p], and back */
int fear_nchw_to_nhwc(const float* in, float* out, long n, int C, int HW, int ld_out, int ch_off, void* stream);
int fear_nhwc_to_nchw(const float* in, float* out, long n, int C, int HW, int ld_in, int ch_off, void* stream);

/* out[m*ld_out + col_out] = scale * in[m*ld_in + col_in]  (cls = 0.1 * cls_pred(c), blocks.py:192, forward and backward) */
int fear_scale_column(const float* in, int ld_in, int col_in, float scale, float* out, int ld_out, int col_out, long M,
                      void* stream);
/* out = a + b over n floats: gradient accumulation where the two branches of the head meet */
int fear_add(const float* a, const float* b, float* out, long n, void* stream);

/* One torch.optim.Adam update (no amsgrad; weight_decay is the L2 form added to the gradient) of n parameters in place — the
 * reference's optimiser is Adam(lr = 1e-4) (train/base_lightning_model.py:63-64).  `step` counts from 1 (bias corrections
 * 1 - beta^step are taken in double on the host, like torch's Python scalars); exp_avg / exp_avg_sq start at zero. */
int fear_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int step, void* stream);

/* ---- block-fused operators of the trunk's training step (round 5; csrc/fear_train_block.h) --------------------------------------
 * One call per inverted-residual block and direction — model_training/model/blocks.py:22-35 over mobile_cv's conv-BN-ReLU units:
 * expand 1x1 + BN + ReLU (absent when `expand` = 0: cexp = cin), depthwise kxk stride s + BN + ReLU, project 1x1 + BN [+ input when
 * `residual`].  The call sequences its kernels itself; BatchNorm'd activations, BatchNorm input gradients and ReLU masks are formed
 * in registers by whichever kernel needs them (forward: statistics in the producing pass, normalisation on load in the consumer;
 * backward: the BatchNorm backward applied on load from the pair (gradient, raw tensor)); what is written is the raw conv outputs
 * e / d / p (saved for the backward) and, in the backward, two masked gradients in `scratch`.  Same arithmetic as the operators
 * above composed unit by unit (tests/test_train_head.py pins both against autograd).  SyncBatchNorm: fear_train_sync_bind below. */
/* The expansion's backward without its raw output (e = x W1^T is linear in the block input: BatchNorm1's input gradient folds into
 * the two consumers' own algebra, a cin x cin matrix each — csrc/fear_train.hip BnbIn): chosen by the call where it pays (up to 32
 * input channels: the large maps); these flags force it wherever it applies (cexp % 16 == 0, cin <= 128) or forbid it. */
#define FEAR_IRB_LINEAR_BN1 1
#define FEAR_IRB_NO_LINEAR_BN1 2
/* ... and the expansion never written at all (FearIrbSaved.e may be NULL): the depthwise kernels of both directions form the channels
 * they own from the pixel's inputs on the spot, BatchNorm1's batch statistics come from the input's Gram matrix.  For the shapes
 * fear_irb_virtual_ok() accepts (16 ... 32 input channels, stride 2, cexp a multiple of 16 from 64 up — FEAR-XS's 16 -> 96 at 128 x 128
 * (0.8 GB per 128 crops), 24 -> 144 at 64 x 64, 32 -> 192 at 32 x 32); the same flag in the forward and the backward call. */
#define FEAR_IRB_VIRTUAL_E 4
/* blocks of at most 32 channels throughout: the projection's weight gradient is summed inside the masked-gradient pass that reads the
 * same three tensors (0.9 GB less traffic per 128-pair step; off by default: it lengthens the chain of input gradients, see
 * csrc/fear_train_block.h irb_w3g) */
#define FEAR_IRB_FUSE_W3 8
typedef struct FearIrbBlock {
    int cin, cexp, cout, k, stride, expand, residual;
    int flags;                     /* 0 = let the call choose; FEAR_IRB_LINEAR_BN1 / FEAR_IRB_NO_LINEAR_BN1 (below) */
    const float* w_pw;             /* [cexp][cin]   (NULL without expansion) */
    const float* w_dw;             /* [k*k][cexp]   depthwise taps, tap-major */
    const float* w_pwl;            /* [cout][cexp] */
    const float* gamma[3];         /* BatchNorm of the expand | depthwise | project unit ([0] unused without expansion) */
    const float* beta[3];
    float* running_mean[3];        /* updated by the forward like fear_bn_train_forward (may be NULL) */
    float* running_var[3];
} FearIrbBlock;
typedef struct FearIrbSaved {      /* written by the forward, read by the backward; caller-allocated */
    float* e;                      /* [B*H*W][cexp]            raw expansion (NULL without expansion) */
    float* d;                      /* [B*(H/s)*(W/s)][cexp]    raw depthwise output */
    float* p;                      /* [B*(H/s)*(W/s)][cout]    raw projection */
    float* vec[3];                 /* per BatchNorm 4*C floats: mean | rstd | a = gamma*rstd | b = beta - mean*a */
} FearIrbSaved;
typedef struct FearIrbGrads {      /* parameter gradients, kernel layouts of FearIrbBlock */
    float* w_pw;
    float* w_dw;
    float* w_pwl;
    float* gamma[3];
    float* beta[3];
} FearIrbGrads;
size_t fear_irb_workspace_bytes(const FearIrbBlock* blk, int B, int H, int W);     /* 0: unsupported shape */
int fear_irb_virtual_ok(const FearIrbBlock* blk);                                   /* 1: FEAR_IRB_VIRTUAL_E may be set for this block */
size_t fear_irb_scratch_floats(const FearIrbBlock* blk, int B, int H, int W);      /* `scratch` of the backward */
/* x [B*H*W][cin] -> out [B*(H/s)*(W/s)][cout] */
int fear_irb_train_forward(const FearIrbBlock* blk, const FearIrbSaved* saved, const float* x, float* out, int B, int H, int W,
                           double momentum, double eps, float* workspace, size_t ws_bytes, void* stream);
/* dout = gradient w.r.t. `out`; dx (may be NULL when the block has an expansion and its input needs no gradient) = gradient w.r.t. x */
/* `wgrad_stream` (may be NULL = `stream`): the block's two pointwise weight gradients do not feed dx; given a second stream they are
 * issued there — ordered behind the kernels that produce their operands by events — and overlap the rest of the backward pass.  The
 * caller then keeps `scratch` private to this call, and makes whatever consumes the gradients (or reuses scratch / workspace / the
 * tensors handed in) wait for that stream. */
int fear_irb_train_backward(const FearIrbBlock* blk, const FearIrbSaved* saved, const FearIrbGrads* grads, const float* x, const float* dout,
                            float* dx, float* scratch, int B, int H, int W, float* workspace, size_t ws_bytes, void* stream,
                            void* wgrad_stream);
/* The running statistics of one BatchNorm from the `vec` its forward saved (mean | rstd | a | b), for forwards that ran with
 * running_mean = NULL: the shared trunk's two passes (template, search: model/fear_net.py:83-88) may then overlap on two streams,
 * and torch's update order — template pass first — is restored by applying the search pass's update afterwards. */
int fear_bn_running_update(const float* vec, double count, float* running_mean, float* running_var, double momentum, double eps, int C,
                           void* stream);
/* ... for a whole pass's BatchNorms in ONE launch (n items; the search pass's 47 deferred updates were 47 launches) */
typedef struct FearBnRunning { const float* vec; float* running_mean; float* running_var; int C; double count; } FearBnRunning;
int fear_bn_running_update_multi(const FearBnRunning* items, int n, double momentum, double eps, void* stream);
/* a lone pointwise conv + BatchNorm [+ ReLU] in the same style (the stem on its im2col rows, the AdjustLayer neck blocks.py:75-88):
 * raw = x w^T, vec as above, out = act(raw) materialised;  backward from dy = gradient w.r.t. out */
size_t fear_pwbn_workspace_bytes(long M, int K, int N);
int fear_pwbn_train_forward(const float* x, int ldx, const float* w, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float* raw, float* vec, int relu, float* out, long M, int K, int N, double momentum, double eps,
                            float* workspace, size_t ws_bytes, void* stream);
int fear_pwbn_train_backward(const float* dy, const float* raw, const float* vec, int relu, const float* x, int ldx, const float* w,
                             const float* gamma, float* dw, float* dgamma, float* dbeta, float* dx, long M, int K, int N, float* workspace,
                             size_t ws_bytes, void* stream, void* wgrad_stream);

/* the stem (3x3 stride-2 conv 3 -> 16 + BatchNorm + ReLU: the FBNet-C first stage behind model/fear_net.py:83-88) on the NCHW image —
 * fear_pwbn_train_* over fear_stem_im2col's rows without ever materialising them (w [16][28]: k = (ci*3 + ky)*3 + kx, column 27 = 0;
 * raw, out [n*(H/2)*(W/2)][16]; the image needs no gradient) */
size_t fear_stem_workspace_bytes(long n, int H, int W);
int fear_stem_train_forward(const float* x_nchw, const float* w, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float* raw, float* vec, float* out, long n, int H, int W, double momentum, double eps,
                            float* workspace, size_t ws_bytes, void* stream);
int fear_stem_train_backward(const float* dy, const float* raw, const float* vec, const float* x_nchw, const float* gamma, float* dw,
                             float* dgamma, float* dbeta, long n, int H, int W, float* workspace, size_t ws_bytes, void* stream,
                             void* wgrad_stream);

/* ---- SyncBatchNorm for the block-fused operators (round 6) -----------------------------------------------------------------------
 * The reference's multi-GPU backends train with `sync_bn: True` (model_training/config/backend/2gpu.yaml:5, 4gpu.yaml:5 ->
 * train/trainer.py:50-52).  In fear_irb_train_* / fear_pwbn_train_* / fear_stem_train_* / fear_sepbn_train_* a BatchNorm's two
 * reductions sit INSIDE one call (producer -> float64 column sums -> finalize -> consumer), so the ranks' all-reduce is a hook:
 * a stream is bound to a FearSync, and every BatchNorm finalize enqueued on that stream becomes
 *     local float64 sums [2][C] -> sync.buf        (forward: sum y | sum y^2;  backward: sum g | sum g * xhat, with d gamma / d beta
 *                                                    taken from the LOCAL sums — they are averaged with every other gradient)
 *     sync.all_reduce(user, buf, 2 C, 0, stream)   the caller's collective (RCCL through torch.distributed in feartracker_amd/train_net.py),
 *                                                    in place, ordered on `stream`; returns 0 on success
 *     finalize from the summed buffer with count = local rows * sync.world
 * (a virtual expansion's BatchNorm1 — statistics from the input's Gram matrix — all-reduces that fp32 matrix instead: is_f32 = 1).
 * Every rank must enqueue the same sequence of operators with the same row counts (DDP's equal batches), and a FearSync serves ONE
 * stream: two passes on two streams bind two of them (two buffers); their collectives then reach the communicator in the host's
 * issue order, which is the same on every rank.  Streams without a binding run the one-rank form; with world = 1 the bound form gives
 * the same numbers bit for bit.  Binding copies *sync; NULL unbinds.  At most 16 streams are bound at a time. */
typedef int (*fear_allreduce_fn)(void* user, void* buf, long n, int is_f32, void* stream);
typedef struct FearSync {
    fear_allreduce_fn all_reduce;
    void* user;
    double* buf;                   /* device scratch of this binding alone, >= FEAR_SYNC_BUF_BYTES */
    size_t buf_bytes;
    int world;                     /* ranks of the group (>= 1) */
} FearSync;
#define FEAR_SYNC_BUF_BYTES 16384  /* 2 x 1024 channels x 8 bytes */
int fear_train_sync_bind(void* stream, const FearSync* sync);

/* SepConv (depthwise 3x3 + pointwise, both with bias) + BatchNorm + ReLU: the layer of the head's encoders and towers
 * (SepConv + BatchNorm2d + ReLU: model_training/model/blocks.py:97-101 MatrixMobile, :115-119 MobileCorrelation, :151-161 BoxTower's towers), one call per direction.  Kernel layouts: depthwise taps
 * [9][cin], pointwise [cout][cin].  The pointwise bias sits in front of the BatchNorm: it cancels in the normalisation (its gradient
 * and the depthwise bias's are exactly zero and are not written) and only shifts the tracked running mean — `raw` is saved without it. */
typedef struct FearSepLayer {
    int cin, cout;
    const float* w_dw;  const float* b_dw;        /* b_dw, b_pw may be NULL */
    const float* w_pw;  const float* b_pw;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;      /* may be NULL: not tracked by this call */
} FearSepLayer;
typedef struct FearSepGrads { float* w_dw; float* w_pw; float* gamma; float* beta; } FearSepGrads;
size_t fear_sepbn_workspace_bytes(const FearSepLayer* layer, int B, int H, int W);      /* 0: unsupported shape */
/* x [B*H*W][ldx] -> saved d [M][cin], raw [M][cout], vec [4*cout] (mean | rstd | a | b) and out [M][ldo] = relu(a raw + b) */
int fear_sepbn_train_forward(const FearSepLayer* layer, const float* x, int ldx, float* d, float* raw, float* vec, float* out, int ldo,
                             int B, int H, int W, double momentum, double eps, float* workspace, size_t ws_bytes, void* stream);
/* dy [M][cout] = gradient w.r.t. out -> dx [M][cin] and the four parameter gradients.  `dd` [M][cin] (the depthwise output's gradient)
 * and `coef` [4*cout] are scratch PRIVATE to this call.  `wgrad_stream` (may be NULL = `stream`): the two weight gradients do not feed
 * dx; given a second stream they are issued there, ordered behind the kernels that produce their operands.  The caller then makes
 * whatever consumes the gradients — or reuses dd / coef / the tensors handed in — wait for that stream, and hands every call that
 * shares this workspace the same weight-gradient stream (its partial sums live there). */
int fear_sepbn_train_backward(const FearSepLayer* layer, const FearSepGrads* grads, const float* x, int ldx, const float* d,
                              const float* raw, const float* vec, const float* dy, float* dd, float* coef, float* dx, int B, int H,
                              int W, float* workspace, size_t ws_bytes, void* stream, void* wgrad_stream);

#ifdef __cplusplus
}
#endif
#endif /* FEAR_TRAIN_H */
