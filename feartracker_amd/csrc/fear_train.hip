// fear_train.hip — gfx950 operators of the FEAR head TRAINING step (BASELINE.json configs[4]):
// forward in train mode (BatchNorm on batch statistics) and backward of everything `BoxTower.forward` is made of
//     SepConv            model_training/model/blocks.py:45-72    depthwise 3x3 (+bias) -> pointwise 1x1 (+bias)
//     BatchNorm2d + ReLU model/blocks.py:98-101,115-119,150-158  (nn.BatchNorm2d, training=True)
//     MobileCorrelation  model/blocks.py:121-126                 s = z^T x, cat[x, s]
//     exp(adjust*x+bias) / 0.1*cls   model/blocks.py:186-192
//     FEARLoss           model_training/train/loss.py:13-96      BCE-with-logits (pos / neg halves) + (1 - IoU) on the weighted cells
// Host code (feartracker_amd/train_head.py) composes them; the reference relies on torch autograd + cuDNN for all of this.
//
// Layout: fp32 NHWC "rows x channels" ([M = batch*H*W][ld], channels contiguous) like the inference engine, so that a 1x1
// convolution's three GEMMs all run on v_mfma_f32_16x16x4_f32:
//     forward   Y[m][n]  = sum_k X[m][k]  W[n][k]          (pw_mfma_kernel, fear_kernels.h)
//     dgrad     dX[m][k] = sum_n dY[m][n] W[n][k]          (the same kernel, W read K-major: WKN = true)
//     wgrad     dW[n][k] = sum_m dY[m][n] X[m][k]          (pw_wgrad_kernel below: reduction over the pixels, split over row
//                                                           slices -> partial [slice][N][K] -> deterministic final sum)
// Every reduction over the batch (BN statistics, bias / BN-affine / depthwise-weight gradients, the loss) is two-stage and
// fixed-order: no atomics, bit-reproducible.
//
// This file is the second half of the library's single translation unit: fear_engine.hip includes it (the kernels of
// fear_kernels.h it reuses — pw_mfma_kernel, dw_conv_kernel — are then instantiated once).
#include "../../include/fear_train.h"

namespace {

using namespace fear;

// (sync_failed: a SyncBatchNorm all-reduce callback of this host thread reported an error since the last check — the finalize helpers
//  that call it return nothing, fear_train_block.h)
thread_local int sync_failed = 0;
#define LAUNCH_CHECK()                                        \
    do {                                                      \
        if (sync_failed) { sync_failed = 0; return FEAR_TRAIN_ERR_SYNC; } \
        if (hipGetLastError() != hipSuccess) return FEAR_TRAIN_ERR_HIP; \
    } while (0)

// Activation applied to a producer's raw output as it is loaded (fused training step, see "Fused conv + BatchNorm operators" below)
struct ActIn {
    const float* a;      // [C] or nullptr: x is used as it is
    const float* b;
    int relu;
};

__device__ __forceinline__ f32x4 act4(const f32x4& x, const f32x4& a, const f32x4& b, bool relu) {
    f32x4 y = (f32x4){__builtin_fmaf(x.x, a.x, b.x), __builtin_fmaf(x.y, a.y, b.y), __builtin_fmaf(x.z, a.z, b.z), __builtin_fmaf(x.w, a.w, b.w)};
    if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    return y;
}

// BatchNorm BACKWARD applied to a gradient operand as it is loaded (block-fused step, fear_train_block.h): the consumer of
// d(pre) = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) never sees that tensor in memory — it loads g (the gradient
// w.r.t. the BatchNorm's output, already masked by the ReLU behind it unless mask_a is given) and the BatchNorm's raw input E and
// forms d(pre) in registers from four per-channel vectors: coef = [A | s1 | mu | Q] with A = gamma * rstd, s1 = sum(g) / count,
// mu = mean, Q = rstd * sum(g * xhat) / count (col_finalize_kernel mode 4 writes them).
//
// Where the BatchNorm's input is itself LINEAR in something the consumer has (E = X W^T, an expansion's raw output), the E term
// need not be read at all:  d(pre) = A (g - s1 + mu Q)  -  (A Q) E,  and the second part folds into the consumer's own algebra
//     input gradient    d(pre) W   = [A (g - s1 + mu Q)] W  -  X (W^T diag(A Q) W)              (a cin x cin matrix)
//     weight gradient   d(pre)^T X = [A (g - s1 + mu Q)]^T X  -  diag(A Q) W (X^T X)            (the input's Gram matrix)
// — E = nullptr with coef given selects exactly the bracket (bnb4 with e = 0), fear_irb_train_backward supplies the rest: the
// expansion's two consumers read g and the cin-channel block input instead of two cexp-channel tensors.
struct BnbIn {
    const float* E;        // [rows][lde] the BatchNorm's input (raw conv output); nullptr with coef: see above
    const float* coef;     // [4][C]; nullptr: the operand is used as loaded
    const float* mask_a;   // optional: g is first masked where fma(E, mask_a, mask_b) <= 0 (a ReLU between the BatchNorm and g)
    const float* mask_b;
    int lde, C;
};

__device__ __forceinline__ f32x4 bnb4(const f32x4& g, const f32x4& e, const f32x4& A, const f32x4& s1, const f32x4& mu, const f32x4& Q) {
    return A * (g - s1 - (e - mu) * Q);
}

__device__ __forceinline__ f32x4 relu_mask4(const f32x4& g, const f32x4& e, const f32x4& ma, const f32x4& mb) {
    return (f32x4){__builtin_fmaf(e.x, ma.x, mb.x) > 0.f ? g.x : 0.f, __builtin_fmaf(e.y, ma.y, mb.y) > 0.f ? g.y : 0.f,
                   __builtin_fmaf(e.z, ma.z, mb.z) > 0.f ? g.z : 0.f, __builtin_fmaf(e.w, ma.w, mb.w) > 0.f ? g.w : 0.f};
}

// ------------------------------------------------------------------------------------------------
// Column reductions over rows: per-channel sums.  Block = 256 threads = (C/4 channel quads) x (R row lanes); a block reduces
// `rpb` rows (col_rows_per_block: 64, doubled until there are at most FEAR_COL_BLOCKS blocks) into partial[block][2][C];
// col_finalize_kernel adds the partials in double, 64 lanes per column in a fixed order.
//   MODE 0: s1 = sum x,        s2 = sum x^2   (float64)                      (BatchNorm forward statistics)
//   MODE 1: g = relu ? (y > 0 ? dy : 0) : dy;  s1 = sum g,  s2 = sum g * xhat,  xhat = (x - mean) * rstd   (BatchNorm backward)
//   MODE 2: s1 = sum dy                                                      (bias gradients)

struct ColArgs {
    const float* A;      // x (mode 0) / dy (modes 1, 2)
    const float* Yact;   // mode 1: activation output (ReLU mask) or nullptr
    const float* X;      // mode 1: the BatchNorm input
    const float* mean;   // mode 1
    const float* rstd;   // mode 1
    double* partial;     // [blocks][2][C] float64
    long M;
    int C, lda, ldy, ldx, rpb;
    const float* act_a;  // mode 1, fused step: the ReLU mask is recomputed from x, active where fma(x, act_a, act_b) > 0
    const float* act_b;  // (Yact is nullptr then)
};

typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 to_f64(const f32x4& v) { return (f64x4){(double)v.x, (double)v.y, (double)v.z, (double)v.w}; }

// Accumulation is in float64 end to end (per thread, across the row lanes, the per-block partials, the final sum): these sums
// feed gradients that are themselves small differences of large sums (a BatchNorm's d beta is sum(dy) of a tensor whose
// channel sums nearly cancel), and a coherent 1e-6 error per element — what fp32 accumulation over 64 rows leaves — shows up
// as a 1e-2 error two layers further down.  Measured against float64 autograd: 2.5e-2 -> 1e-5 (tests/test_train_head.py).
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(ColArgs a) {
    __shared__ f64x4 red[2][256];
    const int c4n = a.C >> 2;
    const int R = 256 / c4n;
    const int cq = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    const f64x4 zero = (f64x4){0.0, 0.0, 0.0, 0.0};
    f64x4 s1 = zero, s2 = zero;
    const long r0 = (long)blockIdx.x * a.rpb;
    const long r1 = r0 + a.rpb < a.M ? r0 + a.rpb : a.M;
    if (rl < R) {
        f32x4 mu = (f32x4){0.f, 0.f, 0.f, 0.f}, rs = mu;
        f32x4 ma = mu, mb = mu;
        if (MODE == 1) {
            mu = *reinterpret_cast<const f32x4*>(a.mean + cq * 4);
            rs = *reinterpret_cast<const f32x4*>(a.rstd + cq * 4);
            if (a.act_a) { ma = *reinterpret_cast<const f32x4*>(a.act_a + cq * 4); mb = *reinterpret_cast<const f32x4*>(a.act_b + cq * 4); }
        }
        // one row of this thread's column quad: the loads of U rows are issued together, the sums taken in row order (the same
        // order as a row-at-a-time loop: the result does not depend on U).  With one load in flight per thread the kernel ran at
        // 3.0-3.3 TB/s (16 KB in flight per CU, profiles/r04_train_traffic.txt) against 5.7-6.1 for the elementwise passes.
        auto accumulate = [&](f32x4 v, const f32x4& y, const f32x4& xin) {
            if (MODE == 0) {
                const f64x4 d = to_f64(v);
                s1 += d;
                s2 += d * d;
            } else if (MODE == 1) {
                if (a.Yact) { v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f; v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f; }
                if (a.act_a) {
                    v.x = __builtin_fmaf(xin.x, ma.x, mb.x) > 0.f ? v.x : 0.f; v.y = __builtin_fmaf(xin.y, ma.y, mb.y) > 0.f ? v.y : 0.f;
                    v.z = __builtin_fmaf(xin.z, ma.z, mb.z) > 0.f ? v.z : 0.f; v.w = __builtin_fmaf(xin.w, ma.w, mb.w) > 0.f ? v.w : 0.f;
                }
                const f32x4 xh = (xin - mu) * rs;
                s1 += to_f64(v);
                s2 += to_f64(v) * to_f64(xh);
            } else {
                s1 += to_f64(v);
            }
        };
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        long r = r0 + rl;
        auto rows = [&](auto utag) {
            constexpr int U = decltype(utag)::value;
            for (; r + (long)(U - 1) * R < r1; r += (long)U * R) {
                f32x4 v[U], y[U], xin[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long ru = r + (long)u * R;
                    v[u] = *reinterpret_cast<const f32x4*>(a.A + ru * a.lda + cq * 4);
                    y[u] = (MODE == 1 && a.Yact) ? *reinterpret_cast<const f32x4*>(a.Yact + ru * a.ldy + cq * 4) : z4;
                    xin[u] = MODE == 1 ? *reinterpret_cast<const f32x4*>(a.X + ru * a.ldx + cq * 4) : z4;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) accumulate(v[u], y[u], xin[u]);
            }
        };
        if (MODE != 1) rows(std::integral_constant<int, 8>{});     // one load per row: eight rows in flight
        rows(std::integral_constant<int, 4>{});
        rows(std::integral_constant<int, 1>{});
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int j = 1; j < R; ++j) {          // fixed order
            s1 += red[0][j * c4n + cq];
            s2 += red[1][j * c4n + cq];
        }
        double* p = a.partial + (long)blockIdx.x * 2 * a.C;
        *reinterpret_cast<f64x4*>(p + cq * 4) = s1;
        *reinterpret_cast<f64x4*>(p + a.C + cq * 4) = s2;
    }
}

// mode 0: mean / rstd (+ running statistics, torch semantics: biased variance normalises, unbiased one is tracked);
// mode 1: the two sums as they are (sum g -> out1, sum g*xhat -> out2);  mode 2: out1 only;  mode 3: float64 sums;
// mode 4: mode 1 + the BnbIn coefficient vectors (out1 / out2 may be NULL: coefficients only);  mode 6: modes 1 and 3 together
// SyncBatchNorm on the block-fused operators (fear_train_sync_bind): a finalize is then two launches around the caller's all-reduce —
// mode 3 (forward) or 6 (backward: d beta / d gamma from the LOCAL sums) leave the float64 sums in the sync buffer, the second launch
// is mode 0 / 4 over that buffer as its one partial row, with M = the rows of all ranks.
struct ColFinArgs {
    const double* partial;
    float* out1;
    float* out2;
    float* running_mean;   // mode 0, optional
    float* running_var;
    double* dsum;          // mode 3: the two sums as float64 [2][C] (what several ranks all-reduce for SyncBatchNorm)
    const float* gamma;    // mode 0, optional: also write the affine a = gamma * rstd, b = beta - mean * a (bn_finalize_kernel's)
    const float* beta;
    float* out_a;
    float* out_b;
    const float* mean_in;  // mode 4: the BatchNorm's saved mean / rstd (gamma above) ...
    const float* rstd_in;
    float* coef;           // ... -> [4][C] = gamma * rstd | s1 / M | mean | rstd * s2 / M  (BnbIn::coef), M = the row count
    const float* mean_shift;   // mode 0, optional: a per-channel constant the producer left out in front of the BatchNorm (a conv
                               // bias: it cancels in the normalisation) — only the tracked mean sees it
    int blocks, C, mode, rpb;
    double M, eps, momentum;
};

__global__ __launch_bounds__(1024) void col_finalize_kernel(ColFinArgs a) {
    // block = 16 columns x 64 lanes; lane j adds partials j, j+64, ... in order, the 64 lane sums are then added in lane order
    __shared__ double red[2][64][17];
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < a.C;
    double s1 = 0.0, s2 = 0.0;
    if (ok)
        for (int b = j; b < a.blocks; b += 64) {
            s1 += a.partial[(long)b * 2 * a.C + c];
            s2 += a.partial[(long)b * 2 * a.C + a.C + c];
        }
    red[0][j][cl] = s1;
    red[1][j][cl] = s2;
    __syncthreads();
    if (j != 0 || !ok) return;
    s1 = 0.0; s2 = 0.0;
    for (int l = 0; l < 64; ++l) { s1 += red[0][l][cl]; s2 += red[1][l][cl]; }
    if (a.mode == 0) {
        // single pass, float64 sums of x and x^2: the E[x^2] - mean^2 cancellation costs mean^2 / var * 2^-53 relative, far
        // below the fp32 inputs' own rounding for any activation a network produces
        const double mean = s1 / a.M;
        const double m2 = s2 - s1 * mean;
        double var = m2 / a.M;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + a.eps));
        a.out1[c] = mf;
        a.out2[c] = rf;
        if (a.out_a) {
            const float av = a.gamma[c] * rf;
            a.out_a[c] = av;
            a.out_b[c] = __builtin_fmaf(-mf, av, a.beta[c]);
        }
        if (a.running_mean) {
            const double tracked = a.mean_shift ? mean + (double)a.mean_shift[c] : mean;
            a.running_mean[c] = (float)((1.0 - a.momentum) * (double)a.running_mean[c] + a.momentum * tracked);
            const double unbiased = a.M > 1.0 ? var * a.M / (a.M - 1.0) : var;
            a.running_var[c] = (float)((1.0 - a.momentum) * (double)a.running_var[c] + a.momentum * unbiased);
        }
    } else if (a.mode == 3 || a.mode == 6) {
        a.dsum[c] = s1;
        a.dsum[a.C + c] = s2;
        if (a.mode == 6) {
            a.out1[c] = (float)s1;
            a.out2[c] = (float)s2;
        }
    } else if (a.mode == 4) {
        // BatchNorm backward: d beta = sum g, d gamma = sum g * xhat, and the coefficients its consumers apply on load
        if (a.out1) {
            a.out1[c] = (float)s1;
            a.out2[c] = (float)s2;
        }
        const float rs = a.rstd_in[c];
        a.coef[c] = a.gamma[c] * rs;
        a.coef[a.C + c] = (float)(s1 / a.M);
        a.coef[2 * a.C + c] = a.mean_in[c];
        a.coef[3 * a.C + c] = (float)((double)rs * (s2 / a.M));
    } else {
        a.out1[c] = (float)s1;
        if (a.out2) a.out2[c] = (float)s2;
    }
}

// SyncBatchNorm: mean / rstd (+ running statistics) of `count` rows from their float64 sums of x and x^2 (all ranks' sums added)
struct BnFromSumsArgs {
    const double* sums;    // [2][C]
    float* mean;
    float* rstd;
    float* running_mean;
    float* running_var;
    int C;
    double count, eps, momentum;
};

__global__ __launch_bounds__(256) void bn_from_sums_kernel(BnFromSumsArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    const double mean = a.sums[c] / a.count;
    double var = (a.sums[a.C + c] - a.sums[c] * mean) / a.count;
    if (var < 0.0) var = 0.0;
    a.mean[c] = (float)mean;
    a.rstd[c] = (float)(1.0 / sqrt(var + a.eps));
    if (a.running_mean) {
        a.running_mean[c] = (float)((1.0 - a.momentum) * (double)a.running_mean[c] + a.momentum * mean);
        const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
        a.running_var[c] = (float)((1.0 - a.momentum) * (double)a.running_var[c] + a.momentum * unbiased);
    }
}

// float64 [2][C] sums -> the float32 [C] vectors bn_bwd_apply_kernel reads (global sums) and d beta / d gamma (local sums)
__global__ __launch_bounds__(256) void sums_to_float_kernel(const double* sums, float* out1, float* out2, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    out1[c] = (float)sums[c];
    out2[c] = (float)sums[C + c];
}

// ------------------------------------------------------------------------------------------------
// BatchNorm apply (+ReLU) and its input gradient, elementwise over float4s.
struct BnApplyArgs {
    const float* X;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    float* Y;
    long M;
    int C, ldx, ldy, relu;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a) {
    const int c4n = a.C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.M * c4n) return;
    const long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(a.X + r * a.ldx + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + c), rs = *reinterpret_cast<const f32x4*>(a.rstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c), b = *reinterpret_cast<const f32x4*>(a.beta + c);
    f32x4 y = (x - mu) * rs * g + b;
    if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    *reinterpret_cast<f32x4*>(a.Y + r * a.ldy + c) = y;
}

struct BnBwdArgs {
    const float* dY;
    const float* Yact;   // ReLU mask source or nullptr
    const float* X;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* sum_g;      // [C] sum g
    const float* sum_gx;     // [C] sum g * xhat
    float* dX;
    long M;
    int C, lddy, ldy, ldx, lddx;
    double count;            // rows the sums were taken over (0 = M; SyncBatchNorm: all ranks' rows)
    const float* act_a;      // fused step: ReLU mask recomputed from X (active where fma(x, act_a, act_b) > 0), Yact = nullptr
    const float* act_b;
};

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
    const int c4n = a.C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.M * c4n) return;
    const long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    f32x4 g = *reinterpret_cast<const f32x4*>(a.dY + r * a.lddy + c);
    if (a.Yact) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(a.Yact + r * a.ldy + c);
        g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
    }
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + c), rs = *reinterpret_cast<const f32x4*>(a.rstd + c);
    const f32x4 xin = *reinterpret_cast<const f32x4*>(a.X + r * a.ldx + c);
    if (a.act_a) {
        const f32x4 ma = *reinterpret_cast<const f32x4*>(a.act_a + c), mb = *reinterpret_cast<const f32x4*>(a.act_b + c);
        g.x = __builtin_fmaf(xin.x, ma.x, mb.x) > 0.f ? g.x : 0.f; g.y = __builtin_fmaf(xin.y, ma.y, mb.y) > 0.f ? g.y : 0.f;
        g.z = __builtin_fmaf(xin.z, ma.z, mb.z) > 0.f ? g.z : 0.f; g.w = __builtin_fmaf(xin.w, ma.w, mb.w) > 0.f ? g.w : 0.f;
    }
    const f32x4 xh = (xin - mu) * rs;
    const float inv_m = (float)(1.0 / (a.count > 0.0 ? a.count : (double)a.M));
    const f32x4 sg = *reinterpret_cast<const f32x4*>(a.sum_g + c) * inv_m;
    const f32x4 sgx = *reinterpret_cast<const f32x4*>(a.sum_gx + c) * inv_m;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + c);
    *reinterpret_cast<f32x4*>(a.dX + r * a.lddx + c) = gm * rs * (g - sg - xh * sgx);
}

// ------------------------------------------------------------------------------------------------
// Pointwise-conv weight gradient on the matrix cores: dW[n][k] = sum_m dY[m][n] X[m][k]  (M in the millions, N and K small:
// the kernel is bound by reading dY and X, so the point is loads per MFMA and passes over the rows).
// A workgroup of four wavefronts owns a 64 (n) x 64 (k) tile of dW for one slice of rows (and one crop when batched); the
// waves take the slice's rows in interleaved groups of 16.  Per step of 4 rows lane (li, lk) loads TWO float4s: dY[m + lk][n0 +
// 4 li .. +3] and X[m + lk][k0 + 4 li .. +3]; MFMA (p, q) takes component p of the first as its A operand and component q of
// the second as its B operand, i.e. it computes dW[n0 + 4 i + p][k0 + 4 j + q] for its 16 x 16 (i, j) — sixteen MFMAs per pair
// of loads, the row/column permutation undone at the store.  (The first version — one wave per 16 x 64 strip with scalar dY
// loads — issued a load per two MFMAs and re-read X once per 16 output channels.)  The four waves' accumulators are added in
// a fixed order through LDS: (w0 + w2) + (w1 + w3).
// The stem's im2col row (fear_stem_im2col: k = (ci * 3 + ky) * 3 + kx of a 3 x 3 stride-2 pad-1 window, column 27 = 0) gathered from
// the NCHW image instead of read from a materialised [pixels][28] tensor (0.24 GB per 128 search crops, written once and read by
// the forward GEMM and by the weight gradient): the operand loaders of those two kernels take it element by element.
struct StemIn {
    const float* img;    // [n][3][H][W]; nullptr: the X operand is an ordinary row-major tensor
    int H, W;
};

__device__ __forceinline__ float stem_tap(const StemIn& st, long bimg, int oy, int ox, int k) {
    if (k >= 27) return 0.f;
    const int ci = k / 9, ky = (k % 9) / 3, kx = k % 3;
    const int y = oy * 2 - 1 + ky, x = ox * 2 - 1 + kx;
    return (y >= 0 && y < st.H && x >= 0 && x < st.W) ? st.img[((bimg * 3 + ci) * st.H + y) * st.W + x] : 0.f;
}

struct WgradArgs {
    const float* dY;     // [M][lddy]   (per crop: + crop * dy_crop_stride)
    const float* X;      // [M][ldx]
    float* P;            // partial [slices][crops][N][K]
    long rows_per_slice, M;      // M = rows per crop when batched
    long dy_crop_stride, x_crop_stride;
    int N, K, lddy, ldx, n_tiles, k_tiles, crops;      // N and K multiples of 4
    const float* act_a;  // fused step: X holds a producer's raw output, the operand is max(fma(x, act_a, act_b), 0) / fma(...)
    const float* act_b;
    int act_relu;
    BnbIn bn;            // block-fused step: the dY operand is the BatchNorm backward of (dY, bn.E), formed on load (crops == 1)
    StemIn stem;         // pw_wgrad_smallk_kernel<2, 1>: X = the stem's im2col rows, gathered (K = 28)
};

__global__ __launch_bounds__(256) void pw_wgrad_kernel(WgradArgs a) {
    __shared__ f32x4 red[2][16 * 64];                  // two accumulator sets of 16 float4 per lane
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int nt = blockIdx.x / a.k_tiles, kt = blockIdx.x % a.k_tiles;
    const int slice = blockIdx.y, crop = blockIdx.z;
    const int n4 = nt * 64 + li * 4, k4 = kt * 64 + li * 4;
    const bool nv = n4 < a.N, kv = k4 < a.K;
    const float* dy = a.dY + (long)crop * a.dy_crop_stride + (nv ? n4 : 0);
    const float* x = a.X + (long)crop * a.x_crop_stride + (kv ? k4 : 0);
    f32x4 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long m0 = (long)slice * a.rows_per_slice;
    const long m1 = m0 + a.rows_per_slice < a.M ? m0 + a.rows_per_slice : a.M;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool act = a.act_a != nullptr;
    f32x4 ia = zero, ib = zero;
    if (act && kv) { ia = *reinterpret_cast<const f32x4*>(a.act_a + k4); ib = *reinterpret_cast<const f32x4*>(a.act_b + k4); }
    const bool bnb = a.bn.coef != nullptr, bne = a.bn.E != nullptr, bmask = bne && a.bn.mask_a != nullptr;
    f32x4 cA = zero, cs1 = zero, cmu = zero, cQ = zero, cma = zero, cmb = zero;
    if (bnb && nv) {
        cA = *reinterpret_cast<const f32x4*>(a.bn.coef + n4); cs1 = *reinterpret_cast<const f32x4*>(a.bn.coef + a.bn.C + n4);
        cmu = *reinterpret_cast<const f32x4*>(a.bn.coef + 2 * a.bn.C + n4); cQ = *reinterpret_cast<const f32x4*>(a.bn.coef + 3 * a.bn.C + n4);
        if (bmask) { cma = *reinterpret_cast<const f32x4*>(a.bn.mask_a + n4); cmb = *reinterpret_cast<const f32x4*>(a.bn.mask_b + n4); }
    }
    const float* eb = bne ? a.bn.E + (nv ? n4 : 0) : nullptr;
    // row loop, software pipelined by hand: the 16 rows of step i + 1 are requested before the 64 MFMAs of step i are issued
    // (two register sets, ping-pong) — a wave otherwise waits out a memory round trip per step with nothing to issue
    auto load = [&](long m, f32x4 (&dv)[4], f32x4 (&xv)[4], f32x4 (&ev)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = m + u * 4 + lk;
            const bool rv = r < m1;
            dv[u] = rv && nv ? *reinterpret_cast<const f32x4*>(dy + r * a.lddy) : zero;
            xv[u] = rv && kv ? *reinterpret_cast<const f32x4*>(x + r * a.ldx) : zero;
            ev[u] = bne && rv && nv ? *reinterpret_cast<const f32x4*>(eb + r * a.bn.lde) : zero;
        }
    };
    auto compute = [&](long m, f32x4 (&dv)[4], f32x4 (&xv)[4], const f32x4 (&ev)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rv = m + u * 4 + lk < m1;
            if (act && rv && kv) xv[u] = act4(xv[u], ia, ib, a.act_relu != 0);
            if (bnb && rv && nv) {
                if (bmask) dv[u] = relu_mask4(dv[u], ev[u], cma, cmb);
                dv[u] = bnb4(dv[u], ev[u], cA, cs1, cmu, cQ);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][p], xv[u][q], acc[p][q], 0, 0, 0);
    };
    {
        f32x4 dA[4], xA[4], eA[4], dB[4], xB[4], eB[4];
        long m = m0 + wave * 16;
        if (m < m1) load(m, dA, xA, eA);
        for (; m < m1; m += 128) {
            const bool hasB = m + 64 < m1;
            if (hasB) load(m + 64, dB, xB, eB);
            compute(m, dA, xA, eA);
            if (hasB) {
                if (m + 128 < m1) load(m + 128, dA, xA, eA);
                compute(m + 64, dB, xB, eB);
            }
        }
    }
    // (w0 + w2) + (w1 + w3), fixed order
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 2][i * 64 + lane] = acc[i >> 2][i & 3];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] += red[wave][i * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[0][i * 64 + lane] = acc[i >> 2][i & 3];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] += red[0][i * 64 + lane];
    // acc[p][q] lane (li, lk), component r  =  dW[nt*64 + 16 lk + 4 r + p][kt*64 + 4 li + q]
    float* P = a.P + (((long)slice * a.crops + crop) * a.N) * a.K;
    if (!kv) return;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = nt * 64 + lk * 16 + r * 4 + p;
            if (nn < a.N) {
                const f32x4 v = (f32x4){acc[p][0][r], acc[p][1][r], acc[p][2][r], acc[p][3][r]};
                *reinterpret_cast<f32x4*>(P + (long)nn * a.K + k4) = v;
            }
        }
}

// pw_wgrad_kernel for K <= 32 input channels (the stem and every expansion / e1 projection of the 128x128 and 64x64 maps: the
// layers with the MOST rows).  The 64 x 64 tile pads K = 16 four times over on the matrix instructions — those launches ran at
// the MFMA rate of their padding, not at the HBM rate of their bytes (0.13 of it in the bench line's roofline).  Here the tile is
// 64 (n) x 16 KB (k): the B operand of MFMA (p, b) is ONE float per lane, X[m + lk][16 b + li], so a launch issues
// 4 KB instead of 16 MFMAs per four rows.  Rows are dealt to the waves and the partial tiles added exactly as in
// pw_wgrad_kernel — every dW element sees the same products in the same order: the results are bit-identical.
// MODE 0: as above | 1 (STEM): the X operand is the stem's im2col row, gathered from the image | 2 (SWAP): the two operands trade
// places — for a NARROW dY (N <= 32: the projections' weight gradients, 24-32 output channels against 96-192 expanded ones) the
// wide tensor takes the float4 side and the narrow one the per-lane side, so that a launch is N / 64 column tiles of streaming
// rows instead of a 64 x 64 (or 128 x 128) tile that is three quarters padding (dW[24][96] over 524 288 rows: 518 us as a 64-wide
// tile, 300 MB).  The prologues trade places with them (activation on the float4 side, BatchNorm backward — no mask — per lane)
// and the tile is stored transposed: the caller passes (dY, N) := (the wide tensor, its width), (X, K) := (the narrow one, its
// width) and still gets dW[K][N] row-major.
template <int KB, int MODE = 0>
__global__ __launch_bounds__(256) void pw_wgrad_smallk_kernel(WgradArgs a) {
    constexpr bool STEM = MODE == 1, SWAP = MODE == 2;
    __shared__ f32x4 red[2][4 * KB * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int nt = blockIdx.x, slice = blockIdx.y, crop = blockIdx.z;
    const int n4 = nt * 64 + li * 4;
    const bool nv = n4 < a.N;
    const float* dy = a.dY + (long)crop * a.dy_crop_stride + (nv ? n4 : 0);
    bool kv[KB];
    const float* x[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        kv[b] = b * 16 + li < a.K;
        x[b] = a.X + (long)crop * a.x_crop_stride + (kv[b] ? b * 16 + li : 0);
    }
    f32x4 acc[4][KB];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[p][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long m0 = (long)slice * a.rows_per_slice;
    const long m1 = m0 + a.rows_per_slice < a.M ? m0 + a.rows_per_slice : a.M;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool act = a.act_a != nullptr;
    float ia[KB], ib[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        ia[b] = !SWAP && act && kv[b] ? a.act_a[b * 16 + li] : 0.f;
        ib[b] = !SWAP && act && kv[b] ? a.act_b[b * 16 + li] : 0.f;
    }
    const bool bnb = a.bn.coef != nullptr, bne = a.bn.E != nullptr, bmask = !SWAP && bne && a.bn.mask_a != nullptr;
    f32x4 cA = zero, cs1 = zero, cmu = zero, cQ = zero, cma = zero, cmb = zero;
    // SWAP: the activation's a | b of this lane's four dY-side columns, and the BatchNorm-backward coefficients of its X-side columns
    f32x4 da = zero, db = zero;
    float xA[KB], xs1[KB], xmu[KB], xQ[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) { xA[b] = 0.f; xs1[b] = 0.f; xmu[b] = 0.f; xQ[b] = 0.f; }
    if (SWAP) {
        if (act && nv) { da = *reinterpret_cast<const f32x4*>(a.act_a + n4); db = *reinterpret_cast<const f32x4*>(a.act_b + n4); }
#pragma unroll
        for (int b = 0; b < KB; ++b)
            if (bnb && kv[b]) {
                const int k = b * 16 + li;
                xA[b] = a.bn.coef[k]; xs1[b] = a.bn.coef[a.bn.C + k]; xmu[b] = a.bn.coef[2 * a.bn.C + k]; xQ[b] = a.bn.coef[3 * a.bn.C + k];
            }
    }
    if (!SWAP && bnb && nv) {
        cA = *reinterpret_cast<const f32x4*>(a.bn.coef + n4); cs1 = *reinterpret_cast<const f32x4*>(a.bn.coef + a.bn.C + n4);
        cmu = *reinterpret_cast<const f32x4*>(a.bn.coef + 2 * a.bn.C + n4); cQ = *reinterpret_cast<const f32x4*>(a.bn.coef + 3 * a.bn.C + n4);
        if (bmask) { cma = *reinterpret_cast<const f32x4*>(a.bn.mask_a + n4); cmb = *reinterpret_cast<const f32x4*>(a.bn.mask_b + n4); }
    }
    const float* eb = !SWAP && bne ? a.bn.E + (nv ? n4 : 0) : nullptr;
    for (long m = m0 + wave * 16; m < m1; m += 64) {
        f32x4 dv[4];
        float xs[4][KB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = m + u * 4 + lk;
            const bool rv = r < m1;
            dv[u] = rv && nv ? *reinterpret_cast<const f32x4*>(dy + r * a.lddy) : zero;
            if (SWAP && act && rv && nv) dv[u] = act4(dv[u], da, db, a.act_relu != 0);
            if (!SWAP && bnb && rv && nv) {
                const f32x4 ev = bne ? *reinterpret_cast<const f32x4*>(eb + r * a.bn.lde) : zero;
                if (bmask) dv[u] = relu_mask4(dv[u], ev, cma, cmb);
                dv[u] = bnb4(dv[u], ev, cA, cs1, cmu, cQ);
            }
            long bimg = 0;
            int oy = 0, ox = 0;
            if (STEM) {
                const int Wo = a.stem.W / 2, Ho = a.stem.H / 2;
                ox = (int)(r % Wo); oy = (int)((r / Wo) % Ho); bimg = r / ((long)Wo * Ho);
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                float v = 0.f;
                if (STEM) { if (rv && kv[b]) v = stem_tap(a.stem, bimg, oy, ox, b * 16 + li); }
                else v = rv && kv[b] ? x[b][r * a.ldx] : 0.f;
                if (!SWAP && act && rv && kv[b]) {
                    v = __builtin_fmaf(v, ia[b], ib[b]);
                    if (a.act_relu) v = fmaxf(v, 0.f);
                }
                if (SWAP && bnb && rv && kv[b]) {
                    const float e = bne ? a.bn.E[r * a.bn.lde + b * 16 + li] : 0.f;
                    v = xA[b] * (v - xs1[b] - (e - xmu[b]) * xQ[b]);
                }
                xs[u][b] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int b = 0; b < KB; ++b) acc[p][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][p], xs[u][b], acc[p][b], 0, 0, 0);
    }
    // (w0 + w2) + (w1 + w3), fixed order
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 4 * KB; ++i) red[wave - 2][i * 64 + lane] = acc[i / KB][i % KB];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 4 * KB; ++i) acc[i / KB][i % KB] += red[wave][i * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int i = 0; i < 4 * KB; ++i) red[0][i * 64 + lane] = acc[i / KB][i % KB];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < 4 * KB; ++i) acc[i / KB][i % KB] += red[0][i * 64 + lane];
    // acc[p][b] lane (li, lk), component r  =  dW[nt*64 + 16 lk + 4 r + p][16 b + li]
    float* P = a.P + (((long)slice * a.crops + crop) * a.N) * a.K;
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        if (!kv[b]) continue;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = nt * 64 + lk * 16 + r * 4 + p;
                if (nn < a.N) P[SWAP ? (long)(b * 16 + li) * a.N + nn : (long)nn * a.K + b * 16 + li] = acc[p][b][r];
            }
    }
}

// out[i] = sum over slices of P[s][i], fixed order (i over crops*N*K): `lanes` threads per float4 (lane l adds slices l, l+lanes,
// ... in order, then the lane sums are added in lane order); lanes = 1 for a handful of slices, 16 for the long reductions
__global__ __launch_bounds__(256) void slice_sum_kernel(const float* P, float* out, long count, int slices, int lanes) {
    __shared__ f32x4 red[256];
    const int per = 256 / lanes;
    const int o = threadIdx.x % per, l = threadIdx.x / per;
    const long i = ((long)blockIdx.x * per + o) * 4;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < count)
        for (int j = l; j < slices; j += lanes) s += *reinterpret_cast<const f32x4*>(P + (long)j * count + i);
    if (lanes > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (l == 0)
            for (int k = 1; k < lanes; ++k) s += red[k * per + o];
    }
    if (l == 0 && i < count) *reinterpret_cast<f32x4*>(out + i) = s;
}

void launch_slice_sum(const float* P, float* out, long count, int slices, hipStream_t s) {
    const int lanes = slices >= 64 ? 16 : 1;
    const int per = 256 / lanes;
    hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((count / 4 + per - 1) / per)), dim3(256), 0, s, P, out, count, slices, lanes);
}

// ------------------------------------------------------------------------------------------------
// Depthwise-conv weight gradient (stride S, pad k/2): dW[t][c] = sum_{b,oy,ox} dY[b,oy,ox,c] * X[b, oy*S+ky-P, ox*S+kx-P, c].
// Block = (C/4 quads) x R pixel lanes over a slice of OUTPUT pixels; partial [block][KS*KS][C] -> slice_sum_kernel.
struct DwWgradArgs {
    const float* dY;
    const float* X;
    float* partial;
    long pixels;          // B*Ho*Wo
    int H, W, Ho, Wo, C, lddy, ldx, rpb;
    const float* act_a;   // fused step: X is a producer's raw output, the operand is its activation (see WgradArgs)
    const float* act_b;
    int act_relu;
};

template <int KS, int S>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwWgradArgs a) {
    constexpr int P = KS / 2, KK = KS * KS;
    __shared__ f32x4 red[256];
    const int c4n = a.C >> 2;
    const int R = 256 / c4n;
    const int cq = threadIdx.x % c4n, rl = threadIdx.x / c4n;
    f32x4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long p0 = (long)blockIdx.x * a.rpb;
    const long p1 = p0 + a.rpb < a.pixels ? p0 + a.rpb : a.pixels;
    constexpr int T = S == 1 ? 8 : 4, WIN = S * (T - 1) + KS;
    const bool act = a.act_a != nullptr;
    f32x4 ia = (f32x4){0.f, 0.f, 0.f, 0.f}, ib = ia;
    if (act && rl < R) { ia = *reinterpret_cast<const f32x4*>(a.act_a + cq * 4); ib = *reinterpret_cast<const f32x4*>(a.act_b + cq * 4); }
    if (a.Wo % T == 0) {
        // runs of T output pixels of one row: the KS input rows are walked once as a WIN-wide register window instead of KS*KS
        // loads per pixel (the loads, not the FMAs, bound this kernel: 25 of them per pixel and channel quad)
        if (rl < R)
            for (long u = p0 / T + rl; u < p1 / T; u += R) {
                const long p = u * T;
                const int ox0 = (int)(p % a.Wo);
                const int oy = (int)((p / a.Wo) % a.Ho);
                const long b = p / ((long)a.Wo * a.Ho);
                f32x4 g[T];
#pragma unroll
                for (int i = 0; i < T; ++i) g[i] = *reinterpret_cast<const f32x4*>(a.dY + (p + i) * a.lddy + cq * 4);
                const float* xb = a.X + b * a.H * a.W * a.ldx + cq * 4;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const int yy = oy * S + ky - P;
                    if (yy < 0 || yy >= a.H) continue;
                    f32x4 xr[WIN];
#pragma unroll
                    for (int i = 0; i < WIN; ++i) {
                        const int xx = ox0 * S + i - P;
                        xr[i] = (xx >= 0 && xx < a.W) ? *reinterpret_cast<const f32x4*>(xb + ((long)yy * a.W + xx) * a.ldx)
                                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (act && xx >= 0 && xx < a.W) xr[i] = act4(xr[i], ia, ib, a.act_relu != 0);      // padding stays zero
                    }
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                        for (int i = 0; i < T; ++i) acc[ky * KS + kx] += g[i] * xr[i * S + kx];
                }
            }
    } else if (rl < R) {
        for (long p = p0 + rl; p < p1; p += R) {
            const int ox = (int)(p % a.Wo);
            const int oy = (int)((p / a.Wo) % a.Ho);
            const long b = p / ((long)a.Wo * a.Ho);
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.dY + p * a.lddy + cq * 4);
            const float* xb = a.X + b * a.H * a.W * a.ldx + cq * 4;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int yy = oy * S + ky - P;
                if (yy < 0 || yy >= a.H) continue;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int xx = ox * S + kx - P;
                    if (xx < 0 || xx >= a.W) continue;
                    f32x4 xv = *reinterpret_cast<const f32x4*>(xb + ((long)yy * a.W + xx) * a.ldx);
                    if (act) xv = act4(xv, ia, ib, a.act_relu != 0);
                    acc[ky * KS + kx] += g * xv;
                }
            }
        }
    }
    float* out = a.partial + (long)blockIdx.x * KK * a.C;
    for (int t = 0; t < KK; ++t) {
        __syncthreads();
        red[threadIdx.x] = acc[t];
        __syncthreads();
        if (rl == 0) {
            f32x4 s = acc[t];
            for (int j = 1; j < R; ++j) s += red[j * c4n + cq];
            *reinterpret_cast<f32x4*>(out + (long)t * a.C + cq * 4) = s;
        }
    }
}

// Depthwise-conv input gradient for any stride (gather form, one thread per input pixel and channel quad):
// dX[b,y,x,c] = sum over taps (ky,kx) with (y + P - ky) = S*oy and (x + P - kx) = S*ox of dY[b,oy,ox,c] * W[ky*KS+kx][c]
struct DwDgradArgs {
    const float* dY;
    const float* Wt;      // [KS*KS][C]
    float* dX;
    long total;           // B*H*W*(C/4)
    int H, W, Ho, Wo, C, lddy, lddx;
};

template <int KS, int S>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(DwDgradArgs a) {
    constexpr int P = KS / 2;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.total) return;
    const int c4n = a.C >> 2;
    const int c = (int)(idx % c4n) * 4; idx /= c4n;
    const int x = (int)(idx % a.W); idx /= a.W;
    const int y = (int)(idx % a.H);
    const long b = idx / a.H;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long dybytes = (a.total / ((long)a.H * a.W * c4n)) * a.Ho * a.Wo * a.lddy * 4;      // B x Ho x Wo x lddy floats
    if (dybytes < (1L << 31)) {
        // Branch-free: only the taps whose parity matches the stride are visited (ky = ky0 + S j), and one that falls outside the
        // map is a buffer load with an out-of-range offset (returns zero, no memory access) — all loads of a thread are in flight
        // together.  With a branch around every tap the stride-2 kernels ran at 2.8-3.4 TB/s (profiles/r04_train_traffic_before.txt).
        // The products are added in the same (ky, kx) order as before.
        const __amdgpu_buffer_rsrc_t gyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dY), 0, (int)dybytes, 0x00020000);
        constexpr int NT = (KS + S - 1) / S;
        const int ky0 = (y + P) % S, kx0 = (x + P) % S;
        const int ty0 = (y + P - ky0) / S, tx0 = (x + P - kx0) / S;          // tap (ky0 + S j, kx0 + S i) reads dY[ty0 - j][tx0 - i]
        const int base = (int)(b * a.Ho * a.Wo) * a.lddy + c;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ky = ky0 + S * j, ty = ty0 - j;
            const bool yok = ky < KS && ty >= 0 && ty < a.Ho;
            f32x4 v[NT], w[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int kx = kx0 + S * i, tx = tx0 - i;
                const bool ok = yok && kx < KS && tx >= 0 && tx < a.Wo;
                v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gyr, ok ? (base + (ty * a.Wo + tx) * a.lddy) * 4 : (int)0x80000000, 0, 0));
                w[i] = *reinterpret_cast<const f32x4*>(a.Wt + (long)(ok ? ky * KS + kx : 0) * a.C + c);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) acc += v[i] * w[i];
        }
    } else {
        const float* gy = a.dY + b * a.Ho * a.Wo * a.lddy + c;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int ty = y + P - ky;
            if (ty < 0 || ty % S != 0 || ty / S >= a.Ho) continue;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int tx = x + P - kx;
                if (tx < 0 || tx % S != 0 || tx / S >= a.Wo) continue;
                acc += *reinterpret_cast<const f32x4*>(gy + ((long)(ty / S) * a.Wo + tx / S) * a.lddy) *
                       *reinterpret_cast<const f32x4*>(a.Wt + (long)(ky * KS + kx) * a.C + c);
            }
        }
    }
    *reinterpret_cast<f32x4*>(a.dX + ((b * a.H + y) * a.W + x) * a.lddx + c) = acc;
}

// Stem conv 3x3 stride 2 pad 1 (3 -> 16, fbnet_c stages[0]) as a GEMM for training: im2col of the caller's NCHW image into
// rows of 28 floats (k = (ci*3 + ky)*3 + kx, column 27 = 0), so that forward and weight gradient are fear_pw_forward /
// fear_pw_backward_weight with K = 28 (the image needs no gradient).
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* X, float* out, long n, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * Ho * Wo * 7) return;
    const int q = (int)(i % 7);                 // 7 float4 per row
    const long p = i / 7;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const long b = p / ((long)Wo * Ho);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = q * 4 + j;
        v[j] = 0.f;
        if (k < 27) {
            const int ci = k / 9, ky = (k % 9) / 3, kx = k % 3;
            const int y = oy * 2 - 1 + ky, x = ox * 2 - 1 + kx;
            if (y >= 0 && y < H && x >= 0 && x < W) v[j] = X[((b * 3 + ci) * H + y) * W + x];
        }
    }
    *reinterpret_cast<f32x4*>(out + p * 28 + q * 4) = (f32x4){v[0], v[1], v[2], v[3]};
}

// ------------------------------------------------------------------------------------------------
// Box head: bbox = exp(adjust * p + bias[c]) (blocks.py:186-187) forward, and backward
//   dp = dbbox * bbox * adjust;  d adjust = sum dbbox * bbox * p;  d bias[c] = sum dbbox * bbox
// p, bbox, dbbox, dp: NHWC rows of 4.  The two reductions reuse col_reduce (mode 0 style) through a staging tensor:
//   T[m][0..3] = dbbox*bbox (-> d bias),  U[m][0..3] = dbbox*bbox*p (-> d adjust = sum over m and c).
struct ExpHeadArgs {
    const float* P;      // [M][4]
    const float* adjust; // [1]
    const float* bias;   // [4]
    float* bbox;         // [M][4]
    const float* dbbox;  // bwd
    float* dP;           // bwd
    float* T;            // bwd [M][4]
    float* U;            // bwd [M][4]
    long M;
};

__global__ __launch_bounds__(256) void exp_head_fwd_kernel(ExpHeadArgs a) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.M) return;
    const f32x4 p = *reinterpret_cast<const f32x4*>(a.P + m * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias);
    const float s = a.adjust[0];
    *reinterpret_cast<f32x4*>(a.bbox + m * 4) = (f32x4){expf(s * p.x + b.x), expf(s * p.y + b.y), expf(s * p.z + b.z), expf(s * p.w + b.w)};
}

__global__ __launch_bounds__(256) void exp_head_bwd_kernel(ExpHeadArgs a) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.M) return;
    const f32x4 p = *reinterpret_cast<const f32x4*>(a.P + m * 4);
    const f32x4 y = *reinterpret_cast<const f32x4*>(a.bbox + m * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.dbbox + m * 4) * y;
    *reinterpret_cast<f32x4*>(a.T + m * 4) = g;
    *reinterpret_cast<f32x4*>(a.U + m * 4) = g * p;
    *reinterpret_cast<f32x4*>(a.dP + m * 4) = g * a.adjust[0];
}

// ------------------------------------------------------------------------------------------------
// FEARLoss (train/loss.py:45-96) forward + gradient, on the head's own NHWC rows:
//   classification: BCEWithLogits, mean over the cells with label 1 and mean over those with label 0, 0.5 each
//   regression:     mean over the cells with weight > 0 of 1 - (I + 1) / (U + 1)     (calc_iou, smooth = 1)
// Pass 1 (loss_partial_kernel): per block {n_pos, n_neg, n_reg, sum bce_pos, sum bce_neg, sum (1 - iou)};
// finalize (one thread): totals -> losses[2] and the three normalisers; pass 2 (loss_grad_kernel): gradients.
struct LossArgs {
    const float* bbox;     // [M][4]  predicted ltrb (after exp)
    const float* cls;      // [M]     logits (after the 0.1 factor)
    const float* gt_reg;   // [M][4]
    const float* gt_cls;   // [M]
    const float* gt_w;     // [M]
    float* partial;        // [blocks][8]
    float* totals;         // [8]: n_pos, n_neg, n_reg, loss_cls, loss_reg
    float* dbbox;          // [M][4]
    float* dcls;           // [M]
    long M;
    int blocks;
    float coef_cls, coef_reg;
};

__device__ __forceinline__ float bce_logits(float x, float y) {      // max(x,0) - x*y + log(1 + exp(-|x|)), like torch
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void loss_partial_kernel(LossArgs a) {
    __shared__ float red[6][256];
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < a.M) {
        const float x = a.cls[m], y = a.gt_cls[m];
        if (y == 1.f) { v[0] = 1.f; v[3] = bce_logits(x, y); }
        else if (y == 0.f) { v[1] = 1.f; v[4] = bce_logits(x, y); }
        if (a.gt_w[m] > 0.f) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(a.bbox + m * 4);
            const f32x4 t = *reinterpret_cast<const f32x4*>(a.gt_reg + m * 4);
            const float ta = (t.x + t.z) * (t.y + t.w), pa = (p.x + p.z) * (p.y + p.w);
            const float wi = fminf(p.x, t.x) + fminf(p.z, t.z), hi = fminf(p.w, t.w) + fminf(p.y, t.y);
            const float inter = wi * hi, uni = ta + pa - inter;
            v[2] = 1.f;
            v[5] = 1.f - (inter + 1.f) / (uni + 1.f);
        }
    }
    for (int j = 0; j < 6; ++j) red[j][threadIdx.x] = v[j];
    __syncthreads();
    if (threadIdx.x < 6) {
        float s = 0.f;
        for (int i = 0; i < 256; ++i) s += red[threadIdx.x][i];     // fixed order
        a.partial[(long)blockIdx.x * 8 + threadIdx.x] = s;
    }
}

__global__ void loss_finalize_kernel(LossArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < a.blocks; ++b)
        for (int j = 0; j < 6; ++j) s[j] += (double)a.partial[(long)b * 8 + j];
    // _weighted_cls_loss (loss.py:68-73): mean over the selected cells.  A selection of exactly ONE cell is a 0-dim index
    // after the reference's `.nonzero().squeeze()` and `_weighted_cls_loss` then returns a constant 0 — no loss and no
    // gradient from that half; mirrored here (loss_grad_kernel too).  An EMPTY selection is NaN in torch
    // (BCEWithLogitsLoss over nothing); here that half is 0 — the one stated deviation (fear_train.h).
    const double lp = s[0] > 1 ? s[3] / s[0] : 0.0, ln = s[1] > 1 ? s[4] / s[1] : 0.0;
    const double lr = s[2] > 0 ? s[5] / s[2] : 0.0;
    a.totals[0] = (float)s[0]; a.totals[1] = (float)s[1]; a.totals[2] = (float)s[2];
    a.totals[3] = (float)((0.5 * lp + 0.5 * ln) * a.coef_cls);
    a.totals[4] = (float)(lr * a.coef_reg);
}

__global__ __launch_bounds__(256) void loss_grad_kernel(LossArgs a) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.M) return;
    const float n_pos = a.totals[0], n_neg = a.totals[1], n_reg = a.totals[2];
    const float x = a.cls[m], y = a.gt_cls[m];
    const float sg = 1.f / (1.f + expf(-x));
    float dc = 0.f;
    if (y == 1.f) dc = n_pos > 1.f ? 0.5f * a.coef_cls * (sg - 1.f) / n_pos : 0.f;      // one selected cell: no gradient (see
    else if (y == 0.f) dc = n_neg > 1.f ? 0.5f * a.coef_cls * sg / n_neg : 0.f;           // loss_finalize_kernel)
    a.dcls[m] = dc;
    f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.gt_w[m] > 0.f) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.bbox + m * 4);
        const f32x4 t = *reinterpret_cast<const f32x4*>(a.gt_reg + m * 4);
        const float pw = p.x + p.z, ph = p.y + p.w;
        const float ta = (t.x + t.z) * (t.y + t.w), pa = pw * ph;
        const float wi = fminf(p.x, t.x) + fminf(p.z, t.z), hi = fminf(p.w, t.w) + fminf(p.y, t.y);
        const float inter = wi * hi, uni = ta + pa - inter;
        // d(1 - (I+1)/(U+1)) = -[dI (U+1) - (I+1) dU] / (U+1)^2,  dU = dPa - dI;  min(p, t) passes the gradient to p where
        // p < t (and half of it on an exact tie, like torch.min's backward)
        auto dmin = [](float pp, float tt) { return pp < tt ? 1.f : (pp == tt ? 0.5f : 0.f); };
        const float dI[4] = {dmin(p.x, t.x) * hi, dmin(p.y, t.y) * wi, dmin(p.z, t.z) * hi, dmin(p.w, t.w) * wi};
        const float dPa[4] = {ph, pw, ph, pw};
        const float den = (uni + 1.f) * (uni + 1.f);
        const float sc = a.coef_reg / n_reg;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = -sc * (dI[j] * (uni + 1.f) - (inter + 1.f) * (dPa[j] - dI[j])) / den;
        d = (f32x4){o[0], o[1], o[2], o[3]};
    }
    *reinterpret_cast<f32x4*>(a.dbbox + m * 4) = d;
}

// ------------------------------------------------------------------------------------------------
// NCHW <-> NHWC at the boundary (the reference's tensors are NCHW): one thread per element, coalesced on the NHWC side.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* in, float* out, long n, int C, int HW, int ldo, int off) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C * HW) return;
    const int c = (int)(i % C);
    const long r = i / C;
    const long b = r / HW, p = r % HW;
    out[r * ldo + off + c] = in[(b * C + c) * HW + p];
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* in, float* out, long n, int C, int HW, int ldi, int off) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C * HW) return;
    const int p = (int)(i % HW);
    const long r = i / HW;
    const long b = r / C;
    const int c = (int)(r % C);
    out[i] = in[(b * HW + p) * ldi + off + c];
}

__global__ void sum4_kernel(const float* in, float* out) { out[0] = (in[0] + in[1]) + (in[2] + in[3]); }

// out[m * ld_out + col_out] = scale * in[m * ld_in + col_in]   (cls = 0.1 * cls_pred(c), blocks.py:192, and its gradient)
__global__ __launch_bounds__(256) void scale_column_kernel(const float* in, int ld_in, int col_in, float scale, float* out,
                                                           int ld_out, int col_out, long M) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) out[m * ld_out + col_out] = scale * in[m * ld_in + col_in];
}

// gradient accumulation where two paths meet (the two branches of the head read the same features): out = a + b
__global__ __launch_bounds__(256) void add_kernel(const float* x, const float* y, float* out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(x + i) + *reinterpret_cast<const f32x4*>(y + i);
    else for (long j = i; j < n; ++j) out[j] = x[j] + y[j];
}

// torch.optim.Adam (no amsgrad), one element per thread: the reference's optimiser (train/base_lightning_model.py:63-64)
struct AdamArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
    float lr_over_bc1, beta1, beta2, eps, weight_decay, bc2_sqrt;
    float one_minus_beta1, one_minus_beta2;      // computed in double on the host like torch's scalar arguments (1.f - 0.999f is 1.3e-5 off 0.001f)
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float g = a.g[i];
    const float p = a.p[i];
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    const float m = a.m[i] + (g - a.m[i]) * a.one_minus_beta1;          // exp_avg.lerp_(grad, 1 - beta1)
    const float v = a.v[i] * a.beta2 + a.one_minus_beta2 * g * g;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    a.m[i] = m;
    a.v[i] = v;
    a.p[i] = p - a.lr_over_bc1 * (m / denom);
}

int train_pick_nt(int n_tiles) {
    for (int nt : {8, 7, 6, 4, 3, 2, 1})
        if (n_tiles % nt == 0) return nt;
    return 1;
}

// Grid of a pointwise GEMM over M rows and n_tiles 16-channel output tiles: one workgroup per 128 rows, and — when that leaves
// the GPU short of waves (the 16 x 16 maps of a 128-pair batch are 32 768 rows = 256 workgroups, ONE wave per SIMD, every operand
// fetched from L2 straight in front of the MFMAs that use it) — the output tiles dealt over gridDim.y as well (pw_mfma_kernel
// walks tiles blockIdx.y * NT, + NT * gridDim.y, ...), with fewer tiles per pass if that is what it takes.
#ifndef FEAR_PW_WGS
#define FEAR_PW_WGS 1024
#endif
dim3 train_pw_grid(long M, int n_tiles, int* nt_out) {
    const long wgs = (M + 127) / 128;
    int nt = train_pick_nt(n_tiles);
    int y = 1;
    if (wgs < FEAR_PW_WGS) {
        const int want = (int)((FEAR_PW_WGS + wgs - 1) / wgs);
        for (int cand : {8, 7, 6, 4, 3, 2, 1}) {            // the largest tile count per pass that still gives `want` passes
            if (cand > nt || n_tiles % cand) continue;
            nt = cand;
            if (n_tiles / cand >= want) break;
        }
        y = n_tiles / nt < want ? n_tiles / nt : want;
    }
    *nt_out = nt;
    return dim3((unsigned)wgs, (unsigned)y);
}

template <bool WKN>
void launch_pw(int nt, dim3 grid, hipStream_t s, const PwArgs& a) {
    switch (nt) {
        case 1: hipLaunchKernelGGL((pw_mfma_kernel<2, 1, WKN, FEAR_PW_KU(1)>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((pw_mfma_kernel<2, 2, WKN, FEAR_PW_KU(2)>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((pw_mfma_kernel<2, 3, WKN, FEAR_PW_KU(3)>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((pw_mfma_kernel<2, 4, WKN, FEAR_PW_KU(4)>), grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((pw_mfma_kernel<2, 6, WKN, FEAR_PW_KU(6)>), grid, dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL((pw_mfma_kernel<2, 7, WKN, FEAR_PW_KU(7)>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((pw_mfma_kernel<2, 8, WKN, FEAR_PW_KU(8)>), grid, dim3(256), 0, s, a); break;
    }
}

// rows are read and written as float4s: every leading dimension is a multiple of 4 floats and covers its row
bool ld_ok(int ld, int cols) { return ld >= cols && ld % 4 == 0; }

#ifndef FEAR_COL_BLOCKS
#define FEAR_COL_BLOCKS 1024   // most workgroups a column reduction is cut into (2048 measured no faster, and slows the depthwise weight gradient's slice sums)
#endif
#ifndef FEAR_COL_ROWS
#define FEAR_COL_ROWS 128     // fewest rows a column-reduction workgroup takes (64: 15.33, 128: 15.22, 256: 15.28 ms per step — the finalize behind it adds half the rows)
#endif
int col_rows_per_block(long M) {
    int r = FEAR_COL_ROWS;
    while ((M + r - 1) / r > FEAR_COL_BLOCKS) r *= 2;
    return r;
}
int col_blocks(long M) { const int r = col_rows_per_block(M); return (int)((M + r - 1) / r); }

// row slices of the pointwise weight gradient: 1024 rows each, but never more than 256 slices (the partials are
// [slices][N][K] floats; at the trunk's 128x128 maps a batch has millions of rows)
#ifndef FEAR_WGRAD_ROWS
#define FEAR_WGRAD_ROWS 1024
#endif
#ifndef FEAR_WGRAD_SMALLK
#define FEAR_WGRAD_SMALLK 1   // 0: every weight gradient on the 64 x 64 tile kernel (A/B)
#endif
long wgrad_rows_per_slice(long M) {
    long r = FEAR_WGRAD_ROWS;
    while ((M + r - 1) / r > 256) r *= 2;
    return r;
}
int wgrad_slices(long M) { const long r = wgrad_rows_per_slice(M); return (int)((M + r - 1) / r); }

// ================================================================================================
// Fused conv + BatchNorm operators of the trunk's training step.
//
// The unfused step wrote, for every conv + BN + ReLU, the raw conv output, read it for the statistics, read it again and wrote
// the activation, and read THAT in the consumer — five passes over tensors of up to 0.8 GB — and eleven more in the backward;
// rocprofv3 (profiles/r03_train_kernel_stats.csv): 38 % of the 31.6 ms step in BatchNorm passes at 2.6-6 TB/s, i.e.
// the kernels stream well and the traffic itself is the cost.  Here an activation is never written:
//   producer   raw conv output `pre` + its column sums (sum x, sum x^2) from the accumulators, same pass
//   finalize   sums -> mean, rstd, running statistics and the affine  a = gamma * rstd,  b = beta - mean * a
//   consumer   act(x) = max(fma(x, a, b), 0) (or without the max) applied to x = pre as it is loaded — forward conv, weight
//              gradient (its x operand) and the ReLU mask of the backward all use this ONE expression, so they agree bit for
//              bit on which elements are active
// Block outputs (the projection's BatchNorm, no ReLU, + residual) are the only activations materialised (fear_bn_act).
// Pointwise conv forward Y = act(X) W^T with the column sums of Y: pw_mfma_kernel's tiling (4 waves x 32 rows per workgroup,
// NT column tiles per pass), the input affine applied to the B-operand fragments, and per pass the accumulators' sum / sum of
// squares reduced over the wave's 32 rows in fp32 (two values per lane, then a 16-lane butterfly), over the four waves in
// float64 through LDS, and written as this workgroup's partial [2][N] (col_finalize_kernel adds the workgroups in float64).
struct PwStatArgs {
    const float* X;
    const float* W;      // [N][K]
    float* Y;
    ActIn in;
    double* partial;     // [gridDim.x][2][N]
    int ldx, ldy, M, K, N;
    int row_tiles;       // 128-row tiles per workgroup (0 = 1): the large maps cut their millions of rows into <= ~2 048 partials
    const float* stem_img;   // pw_stat_kernel<1, true>: X = the stem's im2col rows gathered from this NCHW image (K = 28), see StemIn
    int stem_H, stem_W;
};

__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes that share lane >> 4
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

template <int NT, bool STEM = false>
__global__ __launch_bounds__(256) void pw_stat_kernel(PwStatArgs a) {
    constexpr int MT = 2;
    __shared__ double red[4][2][NT * 16];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row_tiles = a.row_tiles > 0 ? a.row_tiles : 1;
    const bool affine = a.in.a != nullptr;
    const int n_tiles = (a.N + 15) >> 4;
    // gridDim.y > 1 (16 x 16 maps: few row blocks): the passes over the output tiles are dealt to different workgroups; each
    // writes its own columns of this row block's partial
    for (int nc = blockIdx.y * NT; nc < n_tiles; nc += NT * gridDim.y) {
        int nrow[NT];
        bool nvalid[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = (nc + nt) * 16 + li;
            nvalid[nt] = n < a.N;
            nrow[nt] = nvalid[nt] ? n : (a.N - 1);
        }
        // float64 column sums of this workgroup's rows: slot (wave, nt * 16 + 4 lk + c) belongs to lane (li = 0, lk) of that wave
        if (li == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int c = 0; c < 4; ++c) { red[wave][0][nt * 16 + lk * 4 + c] = 0.0; red[wave][1][nt * 16 + lk * 4 + c] = 0.0; }
        }
        for (int rt = 0; rt < row_tiles; ++rt) {
            const int m_wave = ((blockIdx.x * row_tiles + rt) * 4 + wave) * (MT * 16);
            if (m_wave >= a.M) break;      // (wave-uniform; no barrier inside this loop)
            const float* xrow[MT];
            bool mvalid[MT];
            int sox[MT], soy[MT], sb[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int m = m_wave + mt * 16 + li;
                mvalid[mt] = m < a.M;
                if (m >= a.M) m = a.M - 1;
                xrow[mt] = STEM ? nullptr : a.X + (long)m * a.ldx;
                if (STEM) {
                    const int Wo = a.stem_W / 2, Ho = a.stem_H / 2;
                    sox[mt] = m % Wo; soy[mt] = (m / Wo) % Ho; sb[mt] = m / (Wo * Ho);
                }
            }
            f32x4 acc[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int kg = 0; kg < a.K; kg += 16) {
                const int k = kg + lk * 4;
                const bool kvalid = k < a.K;
                f32x4 xf[MT], wf[NT];
                f32x4 ia = (f32x4){1.f, 1.f, 1.f, 1.f}, ib = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (affine && kvalid) {
                    ia = *reinterpret_cast<const f32x4*>(a.in.a + k);
                    ib = *reinterpret_cast<const f32x4*>(a.in.b + k);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    xf[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (kvalid) {
                        if (STEM) {
                            const StemIn st{a.stem_img, a.stem_H, a.stem_W};
#pragma unroll
                            for (int i = 0; i < 4; ++i) xf[mt][i] = stem_tap(st, sb[mt], soy[mt], sox[mt], k + i);
                        } else {
                            xf[mt] = *reinterpret_cast<const f32x4*>(xrow[mt] + k);
                        }
                        if (affine) xf[mt] = act4(xf[mt], ia, ib, a.in.relu != 0);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wf[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (kvalid && nvalid[nt]) wf[nt] = *reinterpret_cast<const f32x4*>(a.W + (long)nrow[nt] * a.K + k);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][i], xf[mt][i], acc[mt][nt], 0, 0, 0);
            }
            // store + statistics: lane holds channels n0 + 4*lk + {0..3} of pixels m_wave + mt*16 + li
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = (nc + nt) * 16 + lk * 4;
                f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (!mvalid[mt]) continue;
                    const f32x4 v = acc[mt][nt];
                    s1 += v;
                    s2 += v * v;
                    if (n < a.N) *reinterpret_cast<f32x4*>(a.Y + (long)(m_wave + mt * 16 + li) * a.ldy + n) = v;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float t1 = row16_sum(s1[c]), t2 = row16_sum(s2[c]);
                    if (li == 0) {
                        red[wave][0][nt * 16 + lk * 4 + c] += (double)t1;
                        red[wave][1][nt * 16 + lk * 4 + c] += (double)t2;
                    }
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * NT * 16; i += 256) {
            const int which = i / (NT * 16), col = i % (NT * 16);
            const int n = nc * 16 + col;
            if (n < a.N)
                a.partial[((long)blockIdx.x * 2 + which) * a.N + n] =
                    ((red[0][which][col] + red[1][which][col]) + red[2][which][col]) + red[3][which][col];      // fixed order
        }
        __syncthreads();
    }
}

// Depthwise conv forward with the input affine (zero padding stays zero: it pads the ACTIVATION) and the column sums of the
// output: dw_conv_kernel's mapping (a thread = 4 channels of a vertical strip of RO output pixels; consecutive threads walk
// channel quads, then x), the workgroup's per-channel sums through LDS in float64, fixed order.
struct DwStatArgs {
    const float* X;
    const float* Wt;     // [KS*KS][C]
    float* Y;
    ActIn in;
    double* partial;     // [gridDim.x][2][C]
    int ldx, ldy, B, H, W, C, Ho, Wo;
};

template <int KS, int S, int RO>
__global__ __launch_bounds__(256) void dw_stat_kernel(DwStatArgs a) {
    constexpr int P = KS / 2;
    constexpr int IR = (RO - 1) * S + KS;
    __shared__ f64x4 red[2][256];
    const int cgs = a.C >> 2;
    const int strips = (a.Ho + RO - 1) / RO;
    const long idx0 = (long)blockIdx.x * blockDim.x;
    long idx = idx0 + threadIdx.x;
    const long total = (long)a.B * strips * a.Wo * cgs;
    const bool live = idx < total;
    const f64x4 zero = (f64x4){0.0, 0.0, 0.0, 0.0};
    f64x4 t1 = zero, t2 = zero;
    if (live) {
        const int cg = idx % cgs; idx /= cgs;
        const int ox = idx % a.Wo; idx /= a.Wo;
        const int st = idx % strips;
        const int b = idx / strips;
        const int c = cg * 4;
        const int oy0 = st * RO;
        f32x4 w[KS * KS];
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) w[t] = *reinterpret_cast<const f32x4*>(a.Wt + (long)t * a.C + c);
        f32x4 acc[RO];
#pragma unroll
        for (int r = 0; r < RO; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool affine = a.in.a != nullptr;
        f32x4 ia = (f32x4){1.f, 1.f, 1.f, 1.f}, ib = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (affine) { ia = *reinterpret_cast<const f32x4*>(a.in.a + c); ib = *reinterpret_cast<const f32x4*>(a.in.b + c); }
        const float* xb = a.X + (long)b * a.H * a.W * a.ldx + c;
        const int iy0 = oy0 * S - P;
        const int ix0 = ox * S - P;
        const long xbytes = (long)a.B * a.H * a.W * a.ldx * 4;
        if (xbytes < (1L << 31)) {
            // branch-free taps (dw_conv_kernel's form, fear_kernels.h): a tap outside the map is a buffer load with an out-of-range
            // offset — no memory access, and all KS loads of a row are in flight together; its ACTIVATION is forced to zero
            // (act(0) = max(b, 0) is not the padding).  Same products in the same order as the branchy form below.
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)xbytes, 0x00020000);
            int pix0 = ((b * a.H + iy0) * a.W + ix0) * a.ldx + c;
            const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int iy = 0; iy < IR; ++iy) {
                const int y = iy0 + iy;
                const bool yin = y >= 0 && y < a.H;
                f32x4 v[KS];
                bool in[KS];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int x = ix0 + kx;
                    in[kx] = yin && x >= 0 && x < a.W;
                    v[kx] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, in[kx] ? (pix0 + (iy * a.W + kx) * a.ldx) * 4 : (int)0x80000000, 0, 0));
                }
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    if (affine) v[kx] = in[kx] ? act4(v[kx], ia, ib, a.in.relu != 0) : z4;
#pragma unroll
                    for (int r = 0; r < RO; ++r) {
                        const int ky = iy - r * S;
                        if (ky >= 0 && ky < KS) acc[r] += v[kx] * w[ky * KS + kx];
                    }
                }
                // two input rows of loads in flight at most: left to itself hipcc hoists all IR x KS buffer loads of a 5 x 5 strip
                // to the top (220 VGPRs next to the 100 of the taps) and spills hundreds of registers; neither sched_barrier nor a
                // memory clobber holds them back, a data dependence of the next rows' offsets on this row's sums does
                if (KS == 5 && (iy & 1) == 1) asm volatile("" : "+v"(pix0) : "v"(acc[0].x), "v"(acc[RO - 1].x), "v"(acc[1].x), "v"(acc[RO - 2].x));
            }
        } else {
#pragma unroll
        for (int iy = 0; iy < IR; ++iy) {
            const int y = iy0 + iy;
            if (y < 0 || y >= a.H) continue;
            const float* xr = xb + (long)y * a.W * a.ldx;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int x = ix0 + kx;
                if (x < 0 || x >= a.W) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(xr + (long)x * a.ldx);
                if (affine) v = act4(v, ia, ib, a.in.relu != 0);
#pragma unroll
                for (int r = 0; r < RO; ++r) {
                    const int ky = iy - r * S;
                    if (ky >= 0 && ky < KS) acc[r] += v * w[ky * KS + kx];
                }
            }
        }
        }
        f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
        for (int r = 0; r < RO; ++r) {
            const int oy = oy0 + r;
            if (oy >= a.Ho) break;
            s1 += acc[r];
            s2 += acc[r] * acc[r];
            *reinterpret_cast<f32x4*>(a.Y + (((long)b * a.Ho + oy) * a.Wo + ox) * a.ldy + c) = acc[r];
        }
        t1 = to_f64(s1);
        t2 = to_f64(s2);
    }
    red[0][threadIdx.x] = t1;
    red[1][threadIdx.x] = t2;
    __syncthreads();
    // thread t < cgs owns the channel quad (idx0 + t) % cgs and adds the entries t, t + cgs, ... of this workgroup (256 >= cgs:
    // every quad occurs at least once per workgroup)
    if ((int)threadIdx.x < cgs) {
        f64x4 s1 = zero, s2 = zero;
        for (int j = threadIdx.x; j < 256; j += cgs) { s1 += red[0][j]; s2 += red[1][j]; }
        const int cg = (int)((idx0 + threadIdx.x) % cgs);
        double* p = a.partial + (long)blockIdx.x * 2 * a.C;
        *reinterpret_cast<f64x4*>(p + cg * 4) = s1;
        *reinterpret_cast<f64x4*>(p + a.C + cg * 4) = s2;
    }
}

// sums [2][C] (float64, all ranks' when SyncBatchNorm) -> mean, rstd, running statistics, a = gamma * rstd, b = beta - mean * a
struct BnFinArgs {
    const double* sums;
    const float* gamma;
    const float* beta;
    float* mean;
    float* rstd;
    float* oa;
    float* ob;
    float* running_mean;
    float* running_var;
    int C;
    double count, eps, momentum;
};

__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFinArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    const double mean = a.sums[c] / a.count;
    double var = (a.sums[a.C + c] - a.sums[c] * mean) / a.count;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + a.eps));
    a.mean[c] = mf;
    a.rstd[c] = rf;
    const float av = a.gamma[c] * rf;
    a.oa[c] = av;
    a.ob[c] = __builtin_fmaf(-mf, av, a.beta[c]);
    if (a.running_mean) {
        a.running_mean[c] = (float)((1.0 - a.momentum) * (double)a.running_mean[c] + a.momentum * mean);
        const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
        a.running_var[c] = (float)((1.0 - a.momentum) * (double)a.running_var[c] + a.momentum * unbiased);
    }
}

// y = act(x) [+ residual]: the one activation the fused step materialises (block outputs)
struct BnActArgs {
    const float* X;
    const float* R;      // residual or nullptr
    float* Y;
    ActIn in;
    long M;
    int C, ldx, ldr, ldy;
};

__global__ __launch_bounds__(256) void bn_act_kernel(BnActArgs a) {
    const int c4n = a.C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.M * c4n) return;
    const long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    f32x4 y = act4(*reinterpret_cast<const f32x4*>(a.X + r * a.ldx + c), *reinterpret_cast<const f32x4*>(a.in.a + c),
                   *reinterpret_cast<const f32x4*>(a.in.b + c), a.in.relu != 0);
    if (a.R) y += *reinterpret_cast<const f32x4*>(a.R + r * a.ldr + c);
    *reinterpret_cast<f32x4*>(a.Y + r * a.ldy + c) = y;
}

#include "fear_train_gemm.h"

}  // namespace

extern "C" {

size_t fear_train_workspace_bytes(long rows, int max_channels) {
    // the largest users: pw wgrad partials [slices][N][K] (slices = ceil(rows / rows_per_slice), see wgrad_slices) with
    // N * K <= max_channels^2; column reductions [blocks <= FEAR_COL_BLOCKS][2][C] float64; depthwise wgrad [blocks][25][C]
    const size_t a = (size_t)wgrad_slices(rows) * (size_t)max_channels * max_channels;
    const size_t b = (size_t)col_blocks(rows) * 25 * (size_t)max_channels;      // <= FEAR_COL_BLOCKS blocks; col partials are 2 doubles = 4 floats
    return ((a > b ? a : b) + 1024) * sizeof(float);
}

int fear_pw_forward(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, long M, int K, int N,
                    void* stream) {
    if (M == 0) return FEAR_TRAIN_OK;
    if (!x || !w || !y) return FEAR_TRAIN_ERR_NULL;
    if (M < 0 || K < 4 || K % 4 || N < 4 || N % 4 || M > 0x7fffffffL || !ld_ok(ldx, K) || !ld_ok(ldy, N)) return FEAR_TRAIN_ERR_SHAPE;
    if (gemm_lds_applies(M, K, N)) {      // the head's 256 / 320-channel GEMMs at 16 x 16: LDS-staged, pipelined (fear_train_gemm.h)
        GemmArgs g{};
        g.X = x; g.ldx = ldx; g.W = w; g.bias = bias; g.Y = y; g.ldy = ldy; g.M = (int)M; g.K = K; g.N = N;
        launch_gemm_lds<0, 0, false>(g, static_cast<hipStream_t>(stream), nullptr);
        LAUNCH_CHECK();
        return FEAR_TRAIN_OK;
    }
    PwArgs a{};
    a.X = x; a.ldx = ldx; a.W = w; a.bias = bias; a.Y = y; a.ldy = ldy; a.M = (int)M; a.K = K; a.N = N;
    int nt = 1;
    const dim3 grid = train_pw_grid(M, (N + 15) / 16, &nt);
    launch_pw<false>(nt, grid, static_cast<hipStream_t>(stream), a);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_pw_backward_data(const float* dy, int lddy, const float* w, const float* add, int ldadd, float* dx, int lddx, long M,
                          int K, int N, void* stream) {
    if (M == 0) return FEAR_TRAIN_OK;
    if (!dy || !w || !dx) return FEAR_TRAIN_ERR_NULL;
    if (M < 0 || K < 4 || K % 4 || N < 4 || N % 4 || M > 0x7fffffffL || !ld_ok(lddy, N) || !ld_ok(lddx, K) || (add && !ld_ok(ldadd, K)))
        return FEAR_TRAIN_ERR_SHAPE;
    // dX[m][k] = sum_n dY[m][n] W[n][k]: a pointwise conv with "input channels" N, "output channels" K and the weight matrix
    // read K-major (W[n][k] row-major IS the K-major layout of that conv)
    if (gemm_lds_applies(M, N, K)) {
        GemmArgs g{};
        g.X = dy; g.ldx = lddy; g.W = w; g.R = add; g.ldr = ldadd; g.Y = dx; g.ldy = lddx; g.M = (int)M; g.K = N; g.N = K;
        launch_gemm_lds<0, 0, true>(g, static_cast<hipStream_t>(stream), nullptr);
        LAUNCH_CHECK();
        return FEAR_TRAIN_OK;
    }
    PwArgs a{};
    a.X = dy; a.ldx = lddy; a.W = w; a.Y = dx; a.ldy = lddx; a.M = (int)M; a.K = N; a.N = K;
    a.R = add; a.ldr = ldadd;
    a.rows_per_crop = (int)M; a.w_crop_stride = 0;
    int nt = 1;
    const dim3 grid = train_pw_grid(M, (K + 15) / 16, &nt);
    launch_pw<true>(nt, grid, static_cast<hipStream_t>(stream), a);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

static int wgrad_impl(const float* dy, int lddy, long dy_crop_stride, const float* x, int ldx, long x_crop_stride, float* dw,
                      float* workspace, size_t ws_bytes, long M, int K, int N, int crops, hipStream_t s, const float* act_a = nullptr,
                      const float* act_b = nullptr, int act_relu = 0, const BnbIn* bn = nullptr, const StemIn* stem = nullptr) {
    WgradArgs a{};
    if (stem) {
        if (crops != 1 || K != 28) return FEAR_TRAIN_ERR_SHAPE;
        a.stem = *stem;
    }
    a.dY = dy; a.X = x; a.lddy = lddy; a.ldx = ldx; a.N = N; a.K = K; a.M = M; a.crops = crops;
    a.act_a = act_a; a.act_b = act_b; a.act_relu = act_relu;
    if (bn) {
        if (crops != 1) return FEAR_TRAIN_ERR_SHAPE;
        a.bn = *bn;
    }
    a.dy_crop_stride = dy_crop_stride; a.x_crop_stride = x_crop_stride;
    // a narrow dY against a wide X (the projections of the large maps): the operands trade places in pw_wgrad_smallk_kernel<., 2>
    if (FEAR_WGRAD_SMALLK && crops == 1 && N <= 32 && K > 32 && !stem && !(bn && bn->mask_a) && (!bn || bn->E)) {
        a.dY = x; a.lddy = ldx; a.N = K; a.X = dy; a.ldx = lddy; a.K = N;
        a.n_tiles = (K + 63) / 64; a.k_tiles = 1;
        a.rows_per_slice = wgrad_rows_per_slice(M);
        const long want = (768 + a.n_tiles - 1) / a.n_tiles;
        long cap_ws = workspace ? (long)(ws_bytes / ((size_t)N * K * sizeof(float))) : 1;
        if (cap_ws < 1) cap_ws = 1;
        long sl = (M + a.rows_per_slice - 1) / a.rows_per_slice;
        if (want > sl) sl = want;
        if (sl > 1024) sl = 1024;
        if (sl > cap_ws) sl = cap_ws;
        long rps = ((M + sl - 1) / sl + 63) / 64 * 64;
        if (rps < 256) rps = 256;
        if (rps < a.rows_per_slice) a.rows_per_slice = rps;
        const int slices = (int)((M + a.rows_per_slice - 1) / a.rows_per_slice);
        if (slices == 1) {
            a.P = dw;
        } else {
            if (!workspace || ws_bytes < (size_t)slices * N * K * sizeof(float)) return FEAR_TRAIN_ERR_WORKSPACE;
            a.P = workspace;
        }
        if (N <= 16) hipLaunchKernelGGL((pw_wgrad_smallk_kernel<1, 2>), dim3(a.n_tiles, slices, 1), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((pw_wgrad_smallk_kernel<2, 2>), dim3(a.n_tiles, slices, 1), dim3(256), 0, s, a);
        if (slices > 1) launch_slice_sum(workspace, dw, (long)N * K, slices, s);
        LAUNCH_CHECK();
        return FEAR_TRAIN_OK;
    }
    a.n_tiles = (N + 63) / 64; a.k_tiles = (K + 63) / 64;
    // 128 x 128 tiles with the rows staged through LDS (wgrad_lds_kernel, fear_train_gemm.h) where both sides are wide enough for
    // the sharing to matter and the launch is one problem, not a batch of per-crop ones
#ifndef FEAR_WGRAD_LDS
#define FEAR_WGRAD_LDS 1
#endif
    // (launches of a few thousand rows are a handful of stages per workgroup: the staged kernel's prologue and its two
    //  barriers-per-stage floor — 29 us measured — lose against the 14 us of the register-only kernel there)
    const bool lds_tile = FEAR_WGRAD_LDS && crops == 1 && K > 32 && N >= 32 && M >= 16384;
    if (lds_tile) { a.n_tiles = (N + 127) / 128; a.k_tiles = (K + 127) / 128; }
    a.rows_per_slice = crops > 1 ? M : wgrad_rows_per_slice(M);
    if (crops == 1) {
        // few output tiles (the 16 x 16 maps' layers: 2-22 tiles x 8-32 slices of 1 024 rows) leave most of the 256 CUs idle while
        // every wave walks its 16 steps of 64 dependent MFMAs: such launches sat on a 31 us floor whatever their size.  Cut the rows
        // finer — down to 256 per slice — until there are ~768 workgroups, as far as the caller's workspace holds the partials.
        const long tiles = (FEAR_WGRAD_SMALLK && K <= 32) ? a.n_tiles : (long)a.n_tiles * a.k_tiles;
        const long want = (768 + tiles - 1) / tiles;
        long cap_ws = workspace ? (long)(ws_bytes / ((size_t)N * K * sizeof(float))) : 1;
        if (cap_ws < 1) cap_ws = 1;      // a workspace smaller than one partial: no finer slicing; the size check below reports it
        long sl = (M + a.rows_per_slice - 1) / a.rows_per_slice;
        if (want > sl) sl = want;
        // at most 256 slices — 1 024 where the partial is small (N * K <= 8 192: the stem and the 16-32-channel layers of the
        // 128 x 128 / 64 x 64 maps): those layers have millions of rows and ONE or two output tiles, 256 workgroups streamed them
        // at 1.1 TB/s (the stem's weight gradient: 425 us for 500 MB)
        const long max_sl = (long)N * K <= 8192 ? 1024 : 256;
        if (sl > max_sl) sl = max_sl;
        if (sl > cap_ws) sl = cap_ws;
        long rps = ((M + sl - 1) / sl + 63) / 64 * 64;
        if (rps < 256) rps = 256;
        if (rps < a.rows_per_slice) a.rows_per_slice = rps;
    }
    const int slices = (int)((M + a.rows_per_slice - 1) / a.rows_per_slice);
    const size_t need = (size_t)slices * crops * N * K * sizeof(float);
    if (slices == 1) {
        a.P = dw;
    } else {
        if (!workspace || ws_bytes < need) return FEAR_TRAIN_ERR_WORKSPACE;
        a.P = workspace;
    }
    if (FEAR_WGRAD_SMALLK && K <= 16) hipLaunchKernelGGL(pw_wgrad_smallk_kernel<1>, dim3(a.n_tiles, slices, crops), dim3(256), 0, s, a);
    else if (FEAR_WGRAD_SMALLK && K <= 32 && stem) hipLaunchKernelGGL((pw_wgrad_smallk_kernel<2, 1>), dim3(a.n_tiles, slices, crops), dim3(256), 0, s, a);
    else if (FEAR_WGRAD_SMALLK && K <= 32) hipLaunchKernelGGL(pw_wgrad_smallk_kernel<2>, dim3(a.n_tiles, slices, crops), dim3(256), 0, s, a);
    else if (lds_tile) hipLaunchKernelGGL(wgrad_lds_kernel<0>, dim3(a.n_tiles * a.k_tiles, slices), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pw_wgrad_kernel, dim3(a.n_tiles * a.k_tiles, slices, crops), dim3(256), 0, s, a);
    if (slices > 1) {
        const long count = (long)crops * N * K;
        launch_slice_sum(workspace, dw, count, slices, s);
    }
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_pw_backward_weight(const float* dy, int lddy, const float* x, int ldx, float* dw, float* workspace, size_t ws_bytes,
                            long M, int K, int N, void* stream) {
    if (!dy || !x || !dw) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || K < 4 || K % 4 || N < 4 || N % 4 || !ld_ok(lddy, N) || !ld_ok(ldx, K)) return FEAR_TRAIN_ERR_SHAPE;    // float4 loads of both operands
    return wgrad_impl(dy, lddy, 0, x, ldx, 0, dw, workspace, ws_bytes, M, K, N, 1, static_cast<hipStream_t>(stream));
}

int fear_col_sum(const float* dy, int lddy, float* out, float* workspace, size_t ws_bytes, long M, int C, void* stream) {
    if (!dy || !out || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(lddy, C)) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = dy; a.lda = lddy; a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<2>, dim3(blocks), dim3(256), 0, s, a);
    ColFinArgs f{};
    f.partial = reinterpret_cast<const double*>(workspace); f.out1 = out; f.out2 = nullptr; f.blocks = blocks; f.C = C; f.mode = 2; f.M = (double)M; f.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

static int dw_impl(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H, int W, int C,
                   int k, int stride, hipStream_t s) {
    DwArgs a{};
    a.X = x; a.ldx = ldx; a.Wt = w; a.bias = bias; a.Y = y; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = H / stride; a.Wo = W / stride; a.relu = 0;
    const long strips = (a.Ho + 3) / 4;
    const long total = (long)B * strips * a.Wo * (C / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_conv_kernel<3, 1, 4>), grid, dim3(256), 0, s, a);
    else if (k == 3) hipLaunchKernelGGL((dw_conv_kernel<3, 2, 4>), grid, dim3(256), 0, s, a);
    else if (stride == 1) hipLaunchKernelGGL((dw_conv_kernel<5, 1, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dw_conv_kernel<5, 2, 4>), grid, dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

static bool dw_shape_ok(int B, int H, int W, int C, int k, int stride) {
    return B >= 0 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0 && C <= 1024 && (k == 3 || k == 5) && (stride == 1 || stride == 2) &&
           H % stride == 0 && W % stride == 0;
}

int fear_dw_forward(const float* x, int ldx, const float* w_taps, const float* bias, float* y, int ldy, int B, int H, int W,
                    int C, int k, int stride, void* stream) {
    if (B == 0) return FEAR_TRAIN_OK;
    if (!x || !w_taps || !y) return FEAR_TRAIN_ERR_NULL;
    if (!dw_shape_ok(B, H, W, C, k, stride)) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(ldx, C) || !ld_ok(ldy, C)) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    return dw_impl(x, ldx, w_taps, bias, y, ldy, B, H, W, C, k, stride, static_cast<hipStream_t>(stream));
}

static int dw_dgrad_impl(const float* dy, int lddy, const float* w_taps, float* dx, int lddx, int B, int H, int W, int C, int k, int stride,
                         hipStream_t s) {
    DwDgradArgs a{};
    a.dY = dy; a.Wt = w_taps; a.dX = dx; a.H = H; a.W = W; a.Ho = H / stride; a.Wo = W / stride; a.C = C; a.lddy = lddy; a.lddx = lddx;
    a.total = (long)B * H * W * (C / 4);
    dim3 grid((unsigned)((a.total + 255) / 256));
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_dgrad_kernel<3, 1>), grid, dim3(256), 0, s, a);
    else if (k == 3) hipLaunchKernelGGL((dw_dgrad_kernel<3, 2>), grid, dim3(256), 0, s, a);
    else if (stride == 1) hipLaunchKernelGGL((dw_dgrad_kernel<5, 1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dw_dgrad_kernel<5, 2>), grid, dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_dw_backward_data(const float* dy, int lddy, const float* w_taps, float* dx, int lddx, int B, int H, int W, int C, int k,
                          int stride, void* stream) {
    if (B == 0) return FEAR_TRAIN_OK;
    if (!dy || !w_taps || !dx) return FEAR_TRAIN_ERR_NULL;
    if (!dw_shape_ok(B, H, W, C, k, stride)) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(lddy, C) || !ld_ok(lddx, C)) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    return dw_dgrad_impl(dy, lddy, w_taps, dx, lddx, B, H, W, C, k, stride, static_cast<hipStream_t>(stream));
}

static int dw_wgrad_impl(const float* dy, int lddy, const float* x, int ldx, float* dw_taps, float* workspace, size_t ws_bytes, int B,
                         int H, int W, int C, int k, int stride, hipStream_t s, const float* act_a, const float* act_b, int act_relu) {
    const int Ho = H / stride, Wo = W / stride;
    const long pixels = (long)B * Ho * Wo;
    const int blocks = col_blocks(pixels);
    const long count = (long)k * k * C;
    if (ws_bytes < (size_t)blocks * count * sizeof(float)) return FEAR_TRAIN_ERR_WORKSPACE;
    DwWgradArgs a{};
    a.dY = dy; a.X = x; a.partial = workspace; a.pixels = pixels; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.C = C; a.lddy = lddy; a.ldx = ldx; a.rpb = col_rows_per_block(pixels);
    a.act_a = act_a; a.act_b = act_b; a.act_relu = act_relu;
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_wgrad_kernel<3, 1>), dim3(blocks), dim3(256), 0, s, a);
    else if (k == 3) hipLaunchKernelGGL((dw_wgrad_kernel<3, 2>), dim3(blocks), dim3(256), 0, s, a);
    else if (stride == 1) hipLaunchKernelGGL((dw_wgrad_kernel<5, 1>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dw_wgrad_kernel<5, 2>), dim3(blocks), dim3(256), 0, s, a);
    launch_slice_sum(workspace, dw_taps, count, blocks, s);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_dw_backward_weight(const float* dy, int lddy, const float* x, int ldx, float* dw_taps, float* workspace,
                            size_t ws_bytes, int B, int H, int W, int C, int k, int stride, void* stream) {
    if (!dy || !x || !dw_taps || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (B < 1 || !dw_shape_ok(B, H, W, C, k, stride)) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(lddy, C) || !ld_ok(ldx, C)) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    return dw_wgrad_impl(dy, lddy, x, ldx, dw_taps, workspace, ws_bytes, B, H, W, C, k, stride, static_cast<hipStream_t>(stream), nullptr,
                         nullptr, 0);
}

int fear_stem_im2col(const float* x_nchw, float* rows28, long n, int H, int W, void* stream) {
    if (n == 0) return FEAR_TRAIN_OK;
    if (!x_nchw || !rows28) return FEAR_TRAIN_ERR_NULL;
    if (n < 0 || H < 2 || W < 2 || H % 2 || W % 2) return FEAR_TRAIN_ERR_SHAPE;
    const long total = n * (H / 2) * (W / 2) * 7;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x_nchw,
                       rows28, n, H, W);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_train_forward(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* mean,
                          float* rstd, float* running_mean, float* running_var, double momentum, double eps, long M, int C,
                          int relu, float* workspace, size_t ws_bytes, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(ldx, C) || !ld_ok(ldy, C)) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = x; a.lda = ldx; a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(blocks), dim3(256), 0, s, a);
    ColFinArgs f{};
    f.partial = reinterpret_cast<const double*>(workspace); f.out1 = mean; f.out2 = rstd; f.running_mean = running_mean; f.running_var = running_var;
    f.blocks = blocks; f.C = C; f.mode = 0; f.M = (double)M; f.rpb = col_rows_per_block(M); f.eps = eps; f.momentum = momentum;
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    BnApplyArgs b{};
    b.X = x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.beta = beta; b.Y = y; b.M = M; b.C = C; b.ldx = ldx; b.ldy = ldy;
    b.relu = relu;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_train_backward(const float* dy, int lddy, const float* y_act, int ldy, const float* x, int ldx, const float* mean,
                           const float* rstd, const float* gamma, float* dx, int lddx, float* dgamma, float* dbeta, long M,
                           int C, float* workspace, size_t ws_bytes, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(lddy, C) || !ld_ok(ldx, C) || !ld_ok(lddx, C) || (y_act && !ld_ok(ldy, C))) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = dy; a.lda = lddy; a.Yact = y_act; a.ldy = ldy; a.X = x; a.ldx = ldx; a.mean = mean; a.rstd = rstd;
    a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
    ColFinArgs f{};
    f.partial = reinterpret_cast<const double*>(workspace); f.out1 = dbeta; f.out2 = dgamma; f.blocks = blocks; f.C = C; f.mode = 1; f.M = (double)M; f.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    BnBwdArgs b{};
    b.dY = dy; b.Yact = y_act; b.X = x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.sum_g = dbeta; b.sum_gx = dgamma;
    b.dX = dx; b.M = M; b.C = C; b.lddy = lddy; b.ldy = ldy; b.ldx = ldx; b.lddx = lddx;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

// ---- fused conv + BatchNorm operators (see "Fused conv + BatchNorm operators" above) ----------------------------------------
static int finalize_sums(const double* partial, int blocks, int C, double* sums, hipStream_t s) {
    ColFinArgs f{};
    f.partial = partial; f.dsum = sums; f.blocks = blocks; f.C = C; f.mode = 3;
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    return 0;
}

int fear_pw_forward_stats(const float* x, int ldx, const float* in_a, const float* in_b, int in_relu, const float* w, float* y,
                          int ldy, long M, int K, int N, double* sums, float* workspace, size_t ws_bytes, void* stream) {
    if (!x || !w || !y || !sums || !workspace || (in_a && !in_b)) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || K < 4 || K % 4 || N < 4 || N % 4 || N > 1024 || M > 0x7fffffffL || !ld_ok(ldx, K) || !ld_ok(ldy, N)) return FEAR_TRAIN_ERR_SHAPE;
    const int blocks = (int)((M + 127) / 128);
    if (ws_bytes < (size_t)blocks * 2 * N * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PwStatArgs a{};
    a.X = x; a.ldx = ldx; a.W = w; a.Y = y; a.ldy = ldy; a.M = (int)M; a.K = K; a.N = N;
    a.in.a = in_a; a.in.b = in_b; a.in.relu = in_relu;
    a.partial = reinterpret_cast<double*>(workspace);
    dim3 grid((unsigned)blocks);
    switch (train_pick_nt((N + 15) / 16)) {
        case 1: hipLaunchKernelGGL((pw_stat_kernel<1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((pw_stat_kernel<2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((pw_stat_kernel<3>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((pw_stat_kernel<4>), grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((pw_stat_kernel<6>), grid, dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL((pw_stat_kernel<7>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((pw_stat_kernel<8>), grid, dim3(256), 0, s, a); break;
    }
    finalize_sums(a.partial, blocks, N, sums, s);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_dw_forward_stats(const float* x, int ldx, const float* in_a, const float* in_b, int in_relu, const float* w_taps, float* y,
                          int ldy, int B, int H, int W, int C, int k, int stride, double* sums, float* workspace, size_t ws_bytes,
                          void* stream) {
    if (!x || !w_taps || !y || !sums || !workspace || (in_a && !in_b)) return FEAR_TRAIN_ERR_NULL;
    if (B < 1 || !dw_shape_ok(B, H, W, C, k, stride) || !ld_ok(ldx, C) || !ld_ok(ldy, C)) return FEAR_TRAIN_ERR_SHAPE;
    DwStatArgs a{};
    a.X = x; a.ldx = ldx; a.Wt = w_taps; a.Y = y; a.ldy = ldy; a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = H / stride; a.Wo = W / stride;
    a.in.a = in_a; a.in.b = in_b; a.in.relu = in_relu;
    const long strips = (a.Ho + 3) / 4;
    const long total = (long)B * strips * a.Wo * (C / 4);
    const long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffL) return FEAR_TRAIN_ERR_SHAPE;
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    a.partial = reinterpret_cast<double*>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)blocks);
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_stat_kernel<3, 1, 4>), grid, dim3(256), 0, s, a);
    else if (k == 3) hipLaunchKernelGGL((dw_stat_kernel<3, 2, 4>), grid, dim3(256), 0, s, a);
    else if (stride == 1) hipLaunchKernelGGL((dw_stat_kernel<5, 1, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dw_stat_kernel<5, 2, 4>), grid, dim3(256), 0, s, a);
    finalize_sums(a.partial, (int)blocks, C, sums, s);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

size_t fear_train_stats_workspace_bytes(long rows, int channels) {
    // partial [workgroups][2][C] float64.  Pointwise producer: one workgroup per 128 rows.  Depthwise producer: one per 256 threads,
    // a thread = 4 channels x 4 output rows of one column -> rows * C / 4096 workgroups (+ ragged strips / the last one).
    const size_t pw = (size_t)((rows + 127) / 128 + 1);
    const size_t dw = (size_t)(rows * (long)((channels + 3) / 4) / 512 + 64);      // strips round up: allow 2x rows * C / 4096
    return (pw > dw ? pw : dw) * 2 * (size_t)channels * sizeof(double);
}

int fear_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float* mean, float* rstd, float* a_out,
                     float* b_out, float* running_mean, float* running_var, double momentum, double eps, int C, void* stream) {
    if (!sums || !gamma || !beta || !mean || !rstd || !a_out || !b_out) return FEAR_TRAIN_ERR_NULL;
    if (C < 1 || !(count >= 1.0)) return FEAR_TRAIN_ERR_SHAPE;
    BnFinArgs f{};
    f.sums = sums; f.gamma = gamma; f.beta = beta; f.mean = mean; f.rstd = rstd; f.oa = a_out; f.ob = b_out;
    f.running_mean = running_mean; f.running_var = running_var; f.C = C; f.count = count; f.eps = eps; f.momentum = momentum;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), f);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_act(const float* x, int ldx, const float* a, const float* b, int relu, const float* residual, int ldr, float* y, int ldy,
                long M, int C, void* stream) {
    if (M == 0) return FEAR_TRAIN_OK;
    if (!x || !a || !b || !y) return FEAR_TRAIN_ERR_NULL;
    if (M < 0 || C < 4 || C % 4 || !ld_ok(ldx, C) || !ld_ok(ldy, C) || (residual && !ld_ok(ldr, C))) return FEAR_TRAIN_ERR_SHAPE;
    BnActArgs k{};
    k.X = x; k.R = residual; k.Y = y; k.in.a = a; k.in.b = b; k.in.relu = relu; k.M = M; k.C = C; k.ldx = ldx; k.ldr = ldr; k.ldy = ldy;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), k);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_backward_reduce_x(const float* dy, int lddy, const float* x, int ldx, const float* act_a, const float* act_b, int relu,
                              const float* mean, const float* rstd, double* sums, long M, int C, float* workspace, size_t ws_bytes,
                              void* stream) {
    if (!dy || !x || !mean || !rstd || !sums || !workspace || (relu && (!act_a || !act_b))) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024 || !ld_ok(lddy, C) || !ld_ok(ldx, C)) return FEAR_TRAIN_ERR_SHAPE;
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = dy; a.lda = lddy; a.X = x; a.ldx = ldx; a.mean = mean; a.rstd = rstd;
    a.act_a = relu ? act_a : nullptr; a.act_b = relu ? act_b : nullptr;
    a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
    finalize_sums(a.partial, blocks, C, sums, s);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_backward_apply_x(const float* dy, int lddy, const float* x, int ldx, const float* act_a, const float* act_b, int relu,
                             const float* mean, const float* rstd, const float* gamma, const double* sums_all, double count,
                             const double* sums_local, float* dx, int lddx, float* dgamma, float* dbeta, float* workspace,
                             size_t ws_bytes, long M, int C, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !sums_all || !sums_local || !dx || !dgamma || !dbeta || !workspace ||
        (relu && (!act_a || !act_b)))
        return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024 || !(count >= (double)M) || !ld_ok(lddy, C) || !ld_ok(ldx, C) || !ld_ok(lddx, C))
        return FEAR_TRAIN_ERR_SHAPE;
    if (ws_bytes < (size_t)2 * C * sizeof(float)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sums_to_float_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums_local, dbeta, dgamma, C);
    float* g1 = workspace;
    float* g2 = workspace + C;
    hipLaunchKernelGGL(sums_to_float_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums_all, g1, g2, C);
    BnBwdArgs b{};
    b.dY = dy; b.X = x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.sum_g = g1; b.sum_gx = g2;
    b.dX = dx; b.M = M; b.C = C; b.lddy = lddy; b.ldx = ldx; b.lddx = lddx; b.count = count;
    b.act_a = relu ? act_a : nullptr; b.act_b = relu ? act_b : nullptr;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_pw_backward_weight_act(const float* dy, int lddy, const float* x, int ldx, const float* in_a, const float* in_b, int in_relu,
                                float* dw, float* workspace, size_t ws_bytes, long M, int K, int N, void* stream) {
    if (!dy || !x || !dw || (in_a && !in_b)) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || K < 4 || K % 4 || N < 4 || N % 4 || !ld_ok(lddy, N) || !ld_ok(ldx, K)) return FEAR_TRAIN_ERR_SHAPE;
    return wgrad_impl(dy, lddy, 0, x, ldx, 0, dw, workspace, ws_bytes, M, K, N, 1, static_cast<hipStream_t>(stream), in_a, in_b, in_relu);
}

int fear_dw_backward_weight_act(const float* dy, int lddy, const float* x, int ldx, const float* in_a, const float* in_b, int in_relu,
                                float* dw_taps, float* workspace, size_t ws_bytes, int B, int H, int W, int C, int k, int stride,
                                void* stream) {
    if (!dy || !x || !dw_taps || !workspace || (in_a && !in_b)) return FEAR_TRAIN_ERR_NULL;
    if (B < 1 || !dw_shape_ok(B, H, W, C, k, stride) || !ld_ok(lddy, C) || !ld_ok(ldx, C)) return FEAR_TRAIN_ERR_SHAPE;
    return dw_wgrad_impl(dy, lddy, x, ldx, dw_taps, workspace, ws_bytes, B, H, W, C, k, stride, static_cast<hipStream_t>(stream), in_a,
                         in_b, in_relu);
}

// BatchNorm (train mode) of a raw conv output in the affine form of the fused operators, WITH the activation written out:
// column sums -> mean / rstd / running statistics and a = gamma * rstd, b = beta - mean * a -> y = fma(x, a, b) [max 0] [+ residual].
// Same three launches as fear_bn_train_forward; its backward (fear_bn_train_backward_x) recomputes the ReLU mask from x with the
// same fma, so the stored activation is never read again on the way back (2 of the 7 passes over a ReLU layer's tensors).
int fear_bn_train_forward_ab(const float* x, int ldx, const float* gamma, const float* beta, int relu, const float* residual, int ldr,
                             float* y, int ldy, float* mean, float* rstd, float* a_out, float* b_out, float* running_mean,
                             float* running_var, double momentum, double eps, long M, int C, float* workspace, size_t ws_bytes,
                             void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || !a_out || !b_out || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024 || !ld_ok(ldx, C) || !ld_ok(ldy, C) || (residual && !ld_ok(ldr, C))) return FEAR_TRAIN_ERR_SHAPE;
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = x; a.lda = ldx; a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(blocks), dim3(256), 0, s, a);
    ColFinArgs f{};
    f.partial = reinterpret_cast<const double*>(workspace); f.out1 = mean; f.out2 = rstd; f.running_mean = running_mean; f.running_var = running_var;
    f.gamma = gamma; f.beta = beta; f.out_a = a_out; f.out_b = b_out;
    f.blocks = blocks; f.C = C; f.mode = 0; f.M = (double)M; f.rpb = a.rpb; f.eps = eps; f.momentum = momentum;
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    BnActArgs k{};
    k.X = x; k.R = residual; k.Y = y; k.in.a = a_out; k.in.b = b_out; k.in.relu = relu; k.M = M; k.C = C; k.ldx = ldx; k.ldr = ldr; k.ldy = ldy;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, k);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_train_backward_x(const float* dy, int lddy, const float* x, int ldx, const float* act_a, const float* act_b, int relu,
                             const float* mean, const float* rstd, const float* gamma, float* dx, int lddx, float* dgamma, float* dbeta,
                             long M, int C, float* workspace, size_t ws_bytes, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !workspace || (relu && (!act_a || !act_b)))
        return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024 || !ld_ok(lddy, C) || !ld_ok(ldx, C) || !ld_ok(lddx, C)) return FEAR_TRAIN_ERR_SHAPE;
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = dy; a.lda = lddy; a.X = x; a.ldx = ldx; a.mean = mean; a.rstd = rstd;
    a.act_a = relu ? act_a : nullptr; a.act_b = relu ? act_b : nullptr;
    a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
    ColFinArgs f{};
    f.partial = reinterpret_cast<const double*>(workspace); f.out1 = dbeta; f.out2 = dgamma; f.blocks = blocks; f.C = C; f.mode = 1; f.M = (double)M; f.rpb = a.rpb;
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    BnBwdArgs b{};
    b.dY = dy; b.X = x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.sum_g = dbeta; b.sum_gx = dgamma;
    b.dX = dx; b.M = M; b.C = C; b.lddy = lddy; b.ldx = ldx; b.lddx = lddx;
    b.act_a = a.act_a; b.act_b = a.act_b;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

// ---- SyncBatchNorm (the reference's multi-GPU backends set sync_bn: True, config/backend/*.yaml): the two reductions of a
// BatchNorm as separate operators, so that the caller can all-reduce the float64 sums between them.
int fear_bn_reduce(const float* x, int ldx, double* sums, long M, int C, float* workspace, size_t ws_bytes, void* stream) {
    if (!x || !sums || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(ldx, C)) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = x; a.lda = ldx; a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(blocks), dim3(256), 0, s, a);
    ColFinArgs f{};
    f.partial = reinterpret_cast<const double*>(workspace); f.dsum = sums; f.blocks = blocks; f.C = C; f.mode = 3; f.M = (double)M;
    f.rpb = a.rpb;
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_forward_from_sums(const float* x, int ldx, const double* sums, double count, const float* gamma, const float* beta,
                              float* y, int ldy, float* mean, float* rstd, float* running_mean, float* running_var,
                              double momentum, double eps, long M, int C, int relu, void* stream) {
    if (!x || !sums || !gamma || !beta || !y || !mean || !rstd) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024 || !(count >= (double)M)) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(ldx, C) || !ld_ok(ldy, C)) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    hipStream_t s = static_cast<hipStream_t>(stream);
    BnFromSumsArgs f{};
    f.sums = sums; f.mean = mean; f.rstd = rstd; f.running_mean = running_mean; f.running_var = running_var; f.C = C;
    f.count = count; f.eps = eps; f.momentum = momentum;
    hipLaunchKernelGGL(bn_from_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, s, f);
    BnApplyArgs b{};
    b.X = x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.beta = beta; b.Y = y; b.M = M; b.C = C; b.ldx = ldx; b.ldy = ldy;
    b.relu = relu;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_backward_reduce(const float* dy, int lddy, const float* y_act, int ldy, const float* x, int ldx, const float* mean,
                            const float* rstd, double* sums, long M, int C, float* workspace, size_t ws_bytes, void* stream) {
    if (!dy || !x || !mean || !rstd || !sums || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(lddy, C) || !ld_ok(ldx, C) || (y_act && !ld_ok(ldy, C))) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    const int blocks = col_blocks(M);
    if (ws_bytes < (size_t)blocks * 2 * C * sizeof(double)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColArgs a{};
    a.A = dy; a.lda = lddy; a.Yact = y_act; a.ldy = ldy; a.X = x; a.ldx = ldx; a.mean = mean; a.rstd = rstd;
    a.partial = reinterpret_cast<double*>(workspace); a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
    ColFinArgs f{};
    f.partial = reinterpret_cast<const double*>(workspace); f.dsum = sums; f.blocks = blocks; f.C = C; f.mode = 3; f.M = (double)M;
    f.rpb = a.rpb;
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_backward_from_sums(const float* dy, int lddy, const float* y_act, int ldy, const float* x, int ldx, const float* mean,
                               const float* rstd, const float* gamma, const double* sums_all, double count,
                               const double* sums_local, float* dx, int lddx, float* dgamma, float* dbeta, float* workspace,
                               size_t ws_bytes, long M, int C, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !sums_all || !sums_local || !dx || !dgamma || !dbeta || !workspace)
        return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || C < 4 || C % 4 || C > 1024 || !(count >= (double)M)) return FEAR_TRAIN_ERR_SHAPE;
    if (!ld_ok(lddy, C) || !ld_ok(ldx, C) || !ld_ok(lddx, C) || (y_act && !ld_ok(ldy, C))) return FEAR_TRAIN_ERR_SHAPE;      // float4 rows
    if (ws_bytes < (size_t)2 * C * sizeof(float)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the parameter gradients are this rank's sums (data-parallel averaging happens with all the other gradients); the input
    // gradient needs the sums and the row count of ALL ranks
    hipLaunchKernelGGL(sums_to_float_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums_local, dbeta, dgamma, C);
    float* g1 = workspace;
    float* g2 = workspace + C;
    hipLaunchKernelGGL(sums_to_float_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums_all, g1, g2, C);
    BnBwdArgs b{};
    b.dY = dy; b.Yact = y_act; b.X = x; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.sum_g = g1; b.sum_gx = g2;
    b.dX = dx; b.M = M; b.C = C; b.lddy = lddy; b.ldy = ldy; b.ldx = ldx; b.lddx = lddx; b.count = count;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_xcorr_forward(const float* x, int ldx, const float* z_nchw, float* s_out, int lds, int B, int P, int C, int J,
                       void* stream) {
    // s[b][p][j] = sum_c x[b][p][c] z[b][c][j]   (MobileCorrelation, blocks.py:121-123); z is the caller's NCHW (C, J) block
    if (B == 0) return FEAR_TRAIN_OK;
    if (!x || !z_nchw || !s_out) return FEAR_TRAIN_ERR_NULL;
    // a wave of pw_mfma_kernel owns 32 rows and picks its crop's weight matrix once: rows of two crops must not share a wave
    if (B < 0 || P < 1 || P % 32 || C < 4 || J < 4 || C % 4 || J % 4 || !ld_ok(ldx, C) || !ld_ok(lds, J)) return FEAR_TRAIN_ERR_SHAPE;
    PwArgs a{};
    a.X = x; a.ldx = ldx; a.W = z_nchw; a.Y = s_out; a.ldy = lds; a.M = B * P; a.K = C; a.N = J;
    a.rows_per_crop = P; a.w_crop_stride = (long)C * J;
    dim3 grid((unsigned)((a.M + 127) / 128));
    launch_pw<true>(train_pick_nt((J + 15) / 16), grid, static_cast<hipStream_t>(stream), a);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_xcorr_backward(const float* ds, int ldds, const float* x, int ldx, const float* z_nchw, const float* dx_add, int ldadd,
                        float* dx, int lddx, float* dz_nchw, int B, int P, int C, int J, void* stream) {
    // dx[b][p][c] = dx_add[b][p][c] + sum_j ds[b][p][j] z[b][c][j];   dz[b][c][j] = sum_p x[b][p][c] ds[b][p][j]
    if (B == 0) return FEAR_TRAIN_OK;
    if (!ds || !x || !z_nchw || !dx || !dz_nchw) return FEAR_TRAIN_ERR_NULL;
    if (B < 0 || P < 1 || P % 128 || C % 4 || J % 4) return FEAR_TRAIN_ERR_SHAPE;      // a 128-row tile must not straddle crops
    hipStream_t s = static_cast<hipStream_t>(stream);
    PwArgs a{};
    a.X = ds; a.ldx = ldds; a.W = z_nchw; a.Y = dx; a.ldy = lddx; a.M = B * P; a.K = J; a.N = C;
    a.R = dx_add; a.ldr = ldadd;
    a.rows_per_crop = P; a.w_crop_stride = (long)C * J;      // per-crop [N = C][K = J] row-major weights = z as it is
    dim3 grid((unsigned)((a.M + 127) / 128));
    launch_pw<false>(train_pick_nt((C + 15) / 16), grid, s, a);
    LAUNCH_CHECK();
    return wgrad_impl(x, ldx, (long)P * ldx, ds, ldds, (long)P * ldds, dz_nchw, nullptr, 0, P, J, C, B, s);
}

int fear_exp_head_forward(const float* p, const float* adjust, const float* bias4, float* bbox, long M, void* stream) {
    if (M == 0) return FEAR_TRAIN_OK;
    if (!p || !adjust || !bias4 || !bbox) return FEAR_TRAIN_ERR_NULL;
    ExpHeadArgs a{};
    a.P = p; a.adjust = adjust; a.bias = bias4; a.bbox = bbox; a.M = M;
    hipLaunchKernelGGL(exp_head_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_exp_head_backward(const float* p, const float* adjust, const float* bbox, const float* dbbox, float* dp, float* dadjust,
                           float* dbias4, float* workspace, size_t ws_bytes, long M, void* stream) {
    if (!p || !adjust || !bbox || !dbbox || !dp || !dadjust || !dbias4 || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0) return FEAR_TRAIN_ERR_SHAPE;
    const int blocks = col_blocks(M);
    const size_t stage = (size_t)M * 4;
    if (ws_bytes < (2 * stage + (size_t)blocks * 16 + 8) * sizeof(float)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* T = workspace;
    float* U = workspace + stage;
    double* part = reinterpret_cast<double*>(U + stage);      // [blocks][2][4] float64 partials (stage is a multiple of 4 floats)
    float* u4 = U + stage + (size_t)blocks * 16;
    ExpHeadArgs a{};
    a.P = p; a.adjust = adjust; a.bbox = const_cast<float*>(bbox); a.dbbox = dbbox; a.dP = dp; a.T = T; a.U = U; a.M = M;
    hipLaunchKernelGGL(exp_head_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, a);
    for (int pass = 0; pass < 2; ++pass) {
        ColArgs c{};
        c.A = pass == 0 ? T : U; c.lda = 4; c.partial = part; c.M = M; c.C = 4; c.rpb = col_rows_per_block(M);
        hipLaunchKernelGGL(col_reduce_kernel<2>, dim3(blocks), dim3(256), 0, s, c);
        ColFinArgs f{};
        f.partial = part; f.out1 = pass == 0 ? dbias4 : u4; f.blocks = blocks; f.C = 4; f.mode = 2; f.M = (double)M; f.rpb = col_rows_per_block(M);
        hipLaunchKernelGGL(col_finalize_kernel, dim3(1), dim3(1024), 0, s, f);
    }
    // d adjust = the four per-channel sums of dbbox * bbox * p added up (adjust is one scalar shared by the four channels)
    hipLaunchKernelGGL(sum4_kernel, dim3(1), dim3(1), 0, s, u4, dadjust);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_head_loss(const float* bbox, const float* cls, const float* gt_reg, const float* gt_cls, const float* gt_weight,
                   float coef_cls, float coef_reg, float* losses2, float* dbbox, float* dcls, float* workspace, size_t ws_bytes,
                   long M, void* stream) {
    if (!bbox || !cls || !gt_reg || !gt_cls || !gt_weight || !losses2 || !dbbox || !dcls || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0) return FEAR_TRAIN_ERR_SHAPE;
    const int blocks = (int)((M + 255) / 256);
    if (ws_bytes < ((size_t)blocks * 8 + 8) * sizeof(float)) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    LossArgs a{};
    a.bbox = bbox; a.cls = cls; a.gt_reg = gt_reg; a.gt_cls = gt_cls; a.gt_w = gt_weight;
    a.partial = workspace; a.totals = workspace + (size_t)blocks * 8; a.dbbox = dbbox; a.dcls = dcls; a.M = M; a.blocks = blocks;
    a.coef_cls = coef_cls; a.coef_reg = coef_reg;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks), dim3(256), 0, s, a);
    if (hipMemcpyAsync(losses2, a.totals + 3, 2 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return FEAR_TRAIN_ERR_HIP;
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_nchw_to_nhwc(const float* in, float* out, long n, int C, int HW, int ld_out, int ch_off, void* stream) {
    if (n == 0) return FEAR_TRAIN_OK;
    if (!in || !out) return FEAR_TRAIN_ERR_NULL;
    const long total = n * C * HW;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
                       n, C, HW, ld_out, ch_off);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_nhwc_to_nchw(const float* in, float* out, long n, int C, int HW, int ld_in, int ch_off, void* stream) {
    if (n == 0) return FEAR_TRAIN_OK;
    if (!in || !out) return FEAR_TRAIN_ERR_NULL;
    const long total = n * C * HW;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
                       n, C, HW, ld_in, ch_off);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_scale_column(const float* in, int ld_in, int col_in, float scale, float* out, int ld_out, int col_out, long M,
                      void* stream) {
    if (M == 0) return FEAR_TRAIN_OK;
    if (!in || !out) return FEAR_TRAIN_ERR_NULL;
    hipLaunchKernelGGL(scale_column_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in, ld_in,
                       col_in, scale, out, ld_out, col_out, M);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_add(const float* a, const float* b, float* out, long n, void* stream) {
    if (n == 0) return FEAR_TRAIN_OK;
    if (!a || !b || !out) return FEAR_TRAIN_ERR_NULL;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, out, n);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int step, void* stream) {
    if (n == 0) return FEAR_TRAIN_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return FEAR_TRAIN_ERR_NULL;
    if (n < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return FEAR_TRAIN_ERR_SHAPE;
    AdamArgs a{};
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay;
    a.one_minus_beta1 = (float)(1.0 - beta1); a.one_minus_beta2 = (float)(1.0 - beta2);
    a.lr_over_bc1 = (float)(lr / (1.0 - pow(beta1, step)));           // step_size = lr / bias_correction1
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

}  // extern "C"

#include "fear_train_block.h"
