// fear_train_block.h — block-fused operators of the trunk's TRAINING step (BASELINE.json configs[4]; round 5).
//
// One C-ABI call per inverted-residual block and direction (model_training/model/blocks.py:22-35 over mobile_cv's
// conv-BN-ReLU units: expand 1x1 + BN + ReLU, depthwise kxk + BN + ReLU, project 1x1 + BN [+ input]); the call sequences its
// kernels on the device side of the boundary — the host issues ~70 calls per step instead of ~1 100 launches through ctypes —
// and no BatchNorm'd activation, no BatchNorm input gradient and no ReLU mask is ever written to memory:
//
//   forward    e = W1 x            + column sums of e          (pw_stat_kernel / gemm_lds_kernel)   saved: e
//              d = DW act1(e)      + column sums of d          (dw_fwd_kernel)                      saved: d
//              p = W3 act2(d)      + column sums of p          (pw_stat_kernel / gemm_lds_kernel)   saved: p
//              out = a3 p + b3 [+ x]                           (bn_act_kernel)
//   backward   sums of (dout, p)                               (col_reduce_kernel<1>)      -> d gamma3, d beta3, coef3
//              g2 = (dp W3) * [act2(d) > 0], dp = BN3'(dout, p) formed on load, + sums of (g2, dhat)   (pw_bwd_kernel<MS> / gemm_lds_kernel)
//              dW3 = dp^T act2(d), both operands formed on load                            (weight-gradient stream)
//              dd = BN2'(g2, d) formed on load into an LDS tile; g1 = (DW^T dd) * [act1(e) > 0]; d taps = sum dd (x) act1(e);
//              sums of (g1, ehat) — one pass over g2, d, e                                 (dw_bwd_kernel)
//              dx = de W1 [+ dout], de = BN1'(g1, e) formed on load                        (pw_bwd_kernel / gemm_lds_kernel)
//              dW1 = de^T x                                                                (weight-gradient stream)
// where BNk'(g, x) = gamma rstd (g - mean(g) - xhat mean(g xhat)) (struct BnbIn).  Passes over the expanded tensors of a block,
// forward + backward: e 8 (was 14 layer by layer), d 7 (was 14); launches 19 (was ~36).
//
// On top of that recipe, where the block's input is narrow (second half of round 5):
//   * BN1' without e (cin <= 32): e = x W1^T is linear in the block input, so de folds into the two consumers' own algebra — dx from
//     [A (g1 - s1 + mu Q) | x] [W1 ; -T], dW1 from [A (g1 - ...)]^T x - diag(A Q) W1 (x^T x) — see BnbIn (fear_train.hip), irb_lin_*;
//   * the stride-2 expansions are never written (FEAR_IRB_VIRTUAL_E): the two depthwise kernels form their tile of e on the matrix
//     pipe, BatchNorm1's statistics come from the input's Gram matrix (gram_kernel, irb_virtual_stats_kernel); the 3 x 3 ones also sum
//     the expansion's weight gradient while g1 is on chip (dw_bwd_kernel<.., W1G>);
//   * blocks of 16 / 24 channels throughout sum the projection's weight gradient inside the masked-gradient pass (pw_bwd_kernel<.., W3G>).
// Also here: the head's SepConv + BatchNorm + ReLU layer (fear_sepbn_train_*), the lone conv + BatchNorm units (fear_pwbn_train_*:
// the neck) and the stem on the NCHW image (fear_stem_train_*).
//
// Included at the end of fear_train.hip (same translation unit: it reuses that file's kernels and helpers).

namespace {

// ------------------------------------------------------------------------------------------------
// Y[m][n] = sum_k X'[m][k] Wt[k][n]  with  X' = BnbIn(X) formed on load and Wt K-major ([Kred][Nout] row-major: the [N][K] matrix
// of the convolution whose INPUT gradient this is).  pw_stat_kernel's tiling (4 waves x 32 rows per workgroup, NT column tiles
// per pass, passes dealt over gridDim.y).  MS = false: + R, store.  MS = true: Y is masked where fma(D, a, b) <= 0 (the ReLU of
// the layer whose raw output D is), stored, and its column sums sum(y) / sum(y * dhat), dhat = (D - mean) * rstd, leave in the
// pass (the next BatchNorm-backward's two reductions): fp32 over a wave's 32 rows, float64 across waves and workgroups.
struct PwBwdArgs {
    const float* G;
    BnbIn bn;
    const float* X2;     // optional: the reduction's rows K1 ... Kred - 1 come from this second tensor [M][ldx2], as loaded (see BnbIn)
    const float* W;      // [Kred][Nout]
    const float* R;      // optional [M][ldr] added to Y (MS = false)
    float* Y;
    const float* D;      // MS: [M][ldd] raw tensor behind the ReLU
    const float* dvec;   // MS: [4][Nout] mean | rstd | a | b of D's BatchNorm
    double* partial;     // MS: [gridDim.x][2][Nout]
    int ldg, ldx2, ldr, ldy, ldd;
    int M, Kred, Nout;
    int K1;              // with X2: a multiple of 16
    int row_tiles;       // 128-row tiles per workgroup (0 = 1)
    float* p3;           // W3G: per-workgroup partials [gridDim.x][Kred][Nout] of the projection's weight gradient, see pw_bwd_kernel
};

// W3G (with MS, NT = all column tiles, Kred and Nout of the same tile count — the blocks without an expansion, 16 / 24 channels on the
// 128 x 128 / 64 x 64 maps): the projection's weight gradient dW3[k][n] = sum_m bnb(G)[m][k] act(D)[m][n] is summed here as well — both
// operands pass through this kernel anyway, a separate weight-gradient launch read three 134 MB tensors again for a 16 x 16 result.
// Lane (k or n index li, row lk) re-reads its elements (L1 hits); one MFMA per four rows and tile; one partial per workgroup.
template <int NT, bool MS, bool W3G = false>
__global__ __launch_bounds__(256) void pw_bwd_kernel(PwBwdArgs a) {
    constexpr int MT = 2;
    static_assert(!W3G || (MS && NT <= 2), "the in-kernel weight gradient: the narrow blocks' masked-gradient pass");
    __shared__ double red[MS ? 4 : 1][2][NT * 16];
    __shared__ f32x4 red3[W3G ? 3 * 64 : 1];
    f32x4 acc3[W3G ? NT : 1][W3G ? NT : 1];
#pragma unroll
    for (int kt = 0; kt < (W3G ? NT : 1); ++kt)
#pragma unroll
        for (int nt = 0; nt < (W3G ? NT : 1); ++nt) acc3[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row_tiles = a.row_tiles > 0 ? a.row_tiles : 1;
    const bool bnb = a.bn.coef != nullptr, bne = a.bn.E != nullptr, bmask = bne && a.bn.mask_a != nullptr;
    const bool two = a.X2 != nullptr;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n_tiles = (a.Nout + 15) >> 4;
    for (int nc = blockIdx.y * NT; nc < n_tiles; nc += NT * gridDim.y) {
        int ncol[NT];
        bool nvalid[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = (nc + nt) * 16 + li;
            nvalid[nt] = n < a.Nout;
            ncol[nt] = nvalid[nt] ? n : (a.Nout - 1);
        }
        if (MS && li == 0) {      // float64 column sums of this workgroup's rows (pw_stat_kernel's scheme)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int c = 0; c < 4; ++c) { red[wave][0][nt * 16 + lk * 4 + c] = 0.0; red[wave][1][nt * 16 + lk * 4 + c] = 0.0; }
        }
        for (int rt = 0; rt < row_tiles; ++rt) {
            const int m_wave = ((blockIdx.x * row_tiles + rt) * 4 + wave) * (MT * 16);
            if (m_wave >= a.M) break;      // (wave-uniform; no barrier inside this loop)
            const float* grow[MT];
            const float* erow[MT];
            const float* x2row[MT];
            bool mvalid[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int m = m_wave + mt * 16 + li;
                mvalid[mt] = m < a.M;
                if (m >= a.M) m = a.M - 1;
                grow[mt] = a.G + (long)m * a.ldg;
                erow[mt] = bne ? a.bn.E + (long)m * a.bn.lde : nullptr;
                x2row[mt] = two ? a.X2 + (long)m * a.ldx2 : nullptr;
            }
            f32x4 acc[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero;
            for (int kg = 0; kg < a.Kred; kg += 16) {
                const int k = kg + lk * 4;
                const bool kvalid = k < a.Kred;      // Kred is a multiple of 4
                f32x4 xf[MT], wf[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[mt] = zero;
                if (kvalid && two && kg >= a.K1) {          // (K1 is a multiple of 16: uniform over the wave)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const f32x4*>(x2row[mt] + (k - a.K1));
                } else if (kvalid) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const f32x4*>(grow[mt] + k);
                    if (bnb) {
                        f32x4 ev[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) ev[mt] = bne ? *reinterpret_cast<const f32x4*>(erow[mt] + k) : zero;
                        const f32x4 cA = *reinterpret_cast<const f32x4*>(a.bn.coef + k), cs1 = *reinterpret_cast<const f32x4*>(a.bn.coef + a.bn.C + k);
                        const f32x4 cmu = *reinterpret_cast<const f32x4*>(a.bn.coef + 2 * a.bn.C + k), cQ = *reinterpret_cast<const f32x4*>(a.bn.coef + 3 * a.bn.C + k);
                        if (bmask) {
                            const f32x4 cma = *reinterpret_cast<const f32x4*>(a.bn.mask_a + k), cmb = *reinterpret_cast<const f32x4*>(a.bn.mask_b + k);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) xf[mt] = relu_mask4(xf[mt], ev[mt], cma, cmb);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) xf[mt] = bnb4(xf[mt], ev[mt], cA, cs1, cmu, cQ);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wf[nt] = zero;
                    if (kvalid && nvalid[nt]) {
                        const float* p = a.W + (long)k * a.Nout + ncol[nt];
                        wf[nt] = (f32x4){p[0], p[a.Nout], p[2 * a.Nout], p[3 * a.Nout]};
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][i], xf[mt][i], acc[mt][nt], 0, 0, 0);
            }
            // epilogue: lane holds columns n0 + 4 lk + {0..3} of rows m_wave + mt * 16 + li
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = (nc + nt) * 16 + lk * 4;
                const bool nok = n < a.Nout;      // Nout is a multiple of 4
                if (MS) {
                    f32x4 dmu = zero, drs = zero, da = zero, db = zero;
                    if (nok) {
                        dmu = *reinterpret_cast<const f32x4*>(a.dvec + n); drs = *reinterpret_cast<const f32x4*>(a.dvec + a.Nout + n);
                        da = *reinterpret_cast<const f32x4*>(a.dvec + 2 * a.Nout + n); db = *reinterpret_cast<const f32x4*>(a.dvec + 3 * a.Nout + n);
                    }
                    f32x4 s1 = zero, s2 = zero;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (!mvalid[mt] || !nok) continue;
                        const long m = m_wave + mt * 16 + li;
                        const f32x4 dv = *reinterpret_cast<const f32x4*>(a.D + m * a.ldd + n);
                        const f32x4 v = relu_mask4(acc[mt][nt], dv, da, db);
                        *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
                        s1 += v;
                        s2 += v * ((dv - dmu) * drs);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float t1 = row16_sum(s1[c]), t2 = row16_sum(s2[c]);
                        if (li == 0) {
                            red[wave][0][nt * 16 + lk * 4 + c] += (double)t1;
                            red[wave][1][nt * 16 + lk * 4 + c] += (double)t2;
                        }
                    }
                } else {
                    if (!nok) continue;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (!mvalid[mt]) continue;
                        const long m = m_wave + mt * 16 + li;
                        f32x4 v = acc[mt][nt];
                        if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
                        *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
                    }
                }
            }
            if constexpr (W3G) {
                // (nc == 0: one pass over all column tiles)  this wave's 32 rows in groups of four
                float cA[NT], cs[NT], cm[NT], cq[NT], da1[NT], db1[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int k = t * 16 + li;
                    const bool kv = k < a.Kred, nv = k < a.Nout;
                    cA[t] = kv ? a.bn.coef[k] : 0.f; cs[t] = kv ? a.bn.coef[a.bn.C + k] : 0.f;
                    cm[t] = kv ? a.bn.coef[2 * a.bn.C + k] : 0.f; cq[t] = kv ? a.bn.coef[3 * a.bn.C + k] : 0.f;
                    da1[t] = nv ? a.dvec[2 * a.Nout + k] : 0.f; db1[t] = nv ? a.dvec[3 * a.Nout + k] : 0.f;
                }
#pragma unroll 2
                for (int g = 0; g < MT * 4; ++g) {
                    const long m = (long)m_wave + g * 4 + lk;
                    const bool mv = m < a.M;
                    float av[NT], bv[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int k = t * 16 + li;
                        av[t] = 0.f; bv[t] = 0.f;
                        if (mv && k < a.Kred) av[t] = cA[t] * (a.G[m * a.ldg + k] - cs[t] - (a.bn.E[m * a.bn.lde + k] - cm[t]) * cq[t]);
                        if (mv && k < a.Nout) bv[t] = fmaxf(__builtin_fmaf(a.D[m * a.ldd + k], da1[t], db1[t]), 0.f);
                    }
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc3[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kt], bv[nt], acc3[kt][nt], 0, 0, 0);
                }
            }
        }
        if (MS) {
            __syncthreads();
            for (int i = threadIdx.x; i < 2 * NT * 16; i += 256) {
                const int which = i / (NT * 16), col = i % (NT * 16);
                const int n = nc * 16 + col;
                if (n < a.Nout)
                    a.partial[((long)blockIdx.x * 2 + which) * a.Nout + n] =
                        ((red[0][which][col] + red[1][which][col]) + red[2][which][col]) + red[3][which][col];      // fixed order
            }
            __syncthreads();
        }
    }
    if constexpr (W3G) {      // the four waves' shares in wave order, one partial [Kred][Nout] per workgroup
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                __syncthreads();
                if (wave > 0) red3[(wave - 1) * 64 + lane] = acc3[kt][nt];
                __syncthreads();
                if (wave == 0) {
                    f32x4 v = acc3[kt][nt];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += red3[w * 64 + lane];
                    const int n = nt * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = kt * 16 + 4 * lk + r;      // lane (n, q), component r = row 4 q + r of the tile
                        if (k < a.Kred && n < a.Nout) a.p3[((long)blockIdx.x * a.Kred + k) * a.Nout + n] = v[r];
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The VIRTUAL expansion of a 16 ... 32-channel block input (FEAR_IRB_VIRTUAL_E; the 16 -> 96 expansion of the 128 x 128 map is 0.8 GB per
// 128 crops, written once and read three times): e = x W1^T is never stored.  Its two remaining consumers — the depthwise kernels;
// the expansion's own backward reads g1 and x, see BnbIn — form their tile of it on the matrix pipe as the tile is staged: per 16
// pixels and 16 input channels one 16-byte load per lane (lane (pixel j, k quarter kk) holds x[j][4 kk ..]) and four MFMAs per 16
// channels against the W1 fragments a lane keeps (lane (channel i, kk): W1[i][4 kk ..]); the result lane (pixel j, q) is the float4 of channels 4 q ..
// 4 q + 3 of pixel j — the (pixel, channel quad) unit both kernels work in.  The SAME products in the same order in both
// directions, so the forward's activation and the backward's mask see the same numbers.
// BatchNorm1's batch statistics follow from linearity as well: sum_m e = W1 (sum_m x), sum_m e^2 = diag(W1 G W1^T), G = x^T x.
struct VirtE {
    const float* X;       // [pixels][cin] the block input
    const float* W1;      // [C][cin] the expansion's weights
    int cin;              // 16 ... 32 (a multiple of 4): NC = ceil(cin / 16) chunks of 16 reduction columns, the last one zero-padded
};

// e[16 channels][16 pixels] from the fragments above (NC chunks of 16 input channels): lane (pixel j, q) gets channels 4 q .. 4 q + 3
template <int NC>
__device__ __forceinline__ f32x4 virt_e_tile(const f32x4 (&wa)[NC], const f32x4 (&xb)[NC]) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < NC; ++ci)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ci][t], xb[ci][t], acc, 0, 0, 0);
    return acc;
}

// G = x^T x [KP][KP] and s = sum_m x [KP] (KP = 16 NC >= cin, zero-padded) of a narrow tensor in one pass: lane (channel i, row kk)
// loads x[row kk][16 ci + i] — with 16 channels a wave's load is 64 consecutive floats — and that one register is the MFMA fragment of
// four rows for BOTH operands (D[i][j] += sum_kk x[kk][i] x[kk][j]); the column sums are the same product against ones.  Per
// workgroup one partial [KP * KP + KP] = G | s, summed by slice_sum_kernel in a fixed order.
template <int NC>
__global__ __launch_bounds__(256) void gram_kernel(const float* X, long M, int cin, long rows_per_wg, float* P) {
    __shared__ f32x4 red[3][NC * NC + NC][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.x * rows_per_wg;
    const long m1 = m0 + rows_per_wg < M ? m0 + rows_per_wg : M;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 g[NC][NC], sm[NC];
#pragma unroll
    for (int a_ = 0; a_ < NC; ++a_) {
        sm[a_] = zero;
#pragma unroll
        for (int b_ = 0; b_ < NC; ++b_) g[a_][b_] = zero;
    }
    constexpr int U = NC == 1 ? 8 : 4;
    const long ngroups = m1 > m0 ? (m1 - m0 + 3) / 4 : 0;      // groups of four rows, dealt to the waves U at a time
    bool cok[NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) cok[ci] = ci * 16 + (lane & 15) < cin;
    for (long g0 = (long)wave * U; g0 < ngroups; g0 += 4 * U) {
        float v[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = m0 + (g0 + u) * 4 + (lane >> 4);
            const bool rok = g0 + u < ngroups && r < m1;
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) v[u][ci] = rok && cok[ci] ? X[r * cin + ci * 16 + (lane & 15)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a_ = 0; a_ < NC; ++a_) {
#pragma unroll
                for (int b_ = 0; b_ < NC; ++b_) g[a_][b_] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][a_], v[u][b_], g[a_][b_], 0, 0, 0);
                sm[a_] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][a_], 1.0f, sm[a_], 0, 0, 0);
            }
    }
    if (wave > 0) {
#pragma unroll
        for (int a_ = 0; a_ < NC; ++a_) {
            red[wave - 1][NC * NC + a_][lane] = sm[a_];
#pragma unroll
            for (int b_ = 0; b_ < NC; ++b_) red[wave - 1][a_ * NC + b_][lane] = g[a_][b_];
        }
    }
    __syncthreads();
    if (wave != 0) return;
    constexpr int KP = 16 * NC;
    float* out = P + (long)blockIdx.x * (KP * KP + KP);
    const int j = lane & 15, q = lane >> 4;
#pragma unroll
    for (int a_ = 0; a_ < NC; ++a_) {
#pragma unroll
        for (int w = 0; w < 3; ++w) sm[a_] += red[w][NC * NC + a_][lane];      // fixed order
#pragma unroll
        for (int b_ = 0; b_ < NC; ++b_) {
#pragma unroll
            for (int w = 0; w < 3; ++w) g[a_][b_] += red[w][a_ * NC + b_][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(a_ * 16 + 4 * q + r) * KP + b_ * 16 + j] = g[a_][b_][r];      // lane (j, q), component r
        }
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[KP * KP + a_ * 16 + 4 * q + r] = sm[a_][r];
        }
    }
}

// BatchNorm1 of a virtual expansion from (G | s) [KP * KP + KP]: mean, rstd, the affine a | b and the running statistics
// (col_finalize mode 0's arithmetic on float64 sums)
template <int KP>
__global__ __launch_bounds__(256) void irb_virtual_stats_kernel(const float* GS, const float* W1, const float* gamma, const float* beta, float* vec,
                                                              float* running_mean, float* running_var, int C, int cin, double M, double eps,
                                                              double momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double w[KP];                                   // (registers: every index below is a compile-time constant; G's are uniform loads)
#pragma unroll
    for (int k = 0; k < KP; ++k) w[k] = k < cin ? (double)W1[(long)c * cin + k] : 0.0;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        s1 += w[k] * (double)GS[KP * KP + k];
        double t = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < KP; ++k2) t += (double)GS[k * KP + k2] * w[k2];
        s2 += w[k] * t;
    }
    const double mean = s1 / M;
    double var = s2 / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + eps));
    vec[c] = mf;
    vec[C + c] = rf;
    const float av = gamma[c] * rf;
    vec[2 * C + c] = av;
    vec[3 * C + c] = __builtin_fmaf(-mf, av, beta[c]);
    if (running_mean) {
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward through  ReLU o BN2 o depthwise o ReLU o BN1  in one pass (the middle of an inverted-residual block).
//   inputs   G2 [B*Ho*Wo][C]  gradient w.r.t. the depthwise unit's activation, already masked by its ReLU (pw_bwd_kernel<MS>)
//            D  [B*Ho*Wo][C]  raw depthwise output (BN2's input);  coef2 = BN2's BnbIn coefficients
//            E  [B*H*W][C]    raw expansion (BN1's input), act1 = [mean | rstd | a | b] of BN1 — or, BN1 = false (blocks without an
//                             expansion), the block input itself, which is the depthwise conv's operand as it stands
//   outputs  Y = g1 = (DW^T dd) * [act1(e) > 0]      (BN1)      |   Y = DW^T dd [+ R]    (no BN1: the block's input gradient)
//            d taps[t][c] = sum dd[o] * act1(e)[o * S + t - P]   (per-workgroup partials, fixed-order final sum)
//            sum g1, sum g1 * ehat                               (BN1; float64 partials)
//   with dd = A2 (g2 - s1 - (d - mu2) Q2), never written to memory.
// A workgroup owns a slab of 4 SQ channels (SQ channel quads) and walks 16 x 16 tiles of the INPUT map: the tile's dd region
// ((16 + 2P)^2 output pixels at stride 1, (8 + ...)^2 at stride 2; zero outside the map = the convolution's padding) is formed
// once into LDS, then every thread — a fixed channel quad and a fixed pixel lane — gathers its taps from LDS: the input
// gradient, and the products for the tap gradients, which stay in registers across all tiles of the workgroup.  At stride 2 a
// pixel only meets the taps of its parity class (ky = (y + P) mod 2 + 2 j), so a thread keeps ONE class (its pixels are dealt
// that way) and its register slot (j, i) means tap (ky0 + 2 j, kx0 + 2 i): every register index is a compile-time constant.
struct DwBwdArgs {
    const float* G2;
    const float* D;
    const float* coef2;   // [4][C]
    const float* Wt;      // [KS*KS][C]
    const float* E;
    const float* act1;    // BN1: [4][C]
    const float* R;       // no BN1: optional [B*H*W][ldr] added to Y
    float* Y;
    float* ptaps;         // [wgs_per_slab][KS*KS][C]
    double* psums;        // BN1: [wgs_per_slab][2][C]
    int ldo, lde, ldr, ldy;
    int B, H, W, Ho, Wo, C;
    int tiles_x, tiles_y, wgs_per_slab, nslab;
    VirtE ve;             // VE: E is not read, e = ve.X ve.W1^T on the spot
    float* pw1;           // VE, optional: per-workgroup partials [wgs_per_slab][C][cin] of sum_m g1[m][c] x[m][k] — the expansion's weight
                          // gradient before its BatchNorm1 algebra (irb_lin_wgrad_fix2_kernel), formed from the tile while g1 is on chip
};

// TS: the tile side, 16 — or 8 for maps of at most 8 x 8 pixels (the template branch's stride-16 stage: a 16 x 16 tile would be three
// quarters outside the map, dd region and sweeps alike; built for the 5 x 5 stride-1 kernels that stage consists of)
// W1G (with VE): the expansion's weight gradient is accumulated here as well (DwBwdArgs::pw1; built for the 3 x 3 kernels: the 5 x 5
// ones have no registers left for its accumulators)
template <int KS, int S, int SQ, bool BN1, int VE = 0, int TS = 16, bool W1G = false>      // VE = NC chunks of 16 input channels of a virtual expansion (0: E is read)
__global__ __launch_bounds__(256, 2) void dw_bwd_kernel(DwBwdArgs a) {
    static_assert(!VE || (BN1 && S == 2), "the virtual expansion is built for the stride-2 kernels");
    static_assert(TS == 16 || (TS == 8 && S == 1 && KS == 5), "8 x 8 tiles: the 5 x 5 stride-1 kernels only");
    static_assert(!W1G || VE != 0, "the in-kernel weight gradient belongs to the virtual expansion");
    constexpr int P = KS / 2, KK = KS * KS;
    constexpr int LO = P / S;                          // output rows / columns in front of the tile's first own one
    constexpr int OR = (TS - 1 + P) / S + LO + 1;      // side of the dd region a tile reads
    constexpr int PITCH = (OR + 1) * SQ;               // float4s per region row: one pixel of padding turns consecutive rows by half
                                                       // the LDS banks, so the lanes of a ds_read_b128 group (consecutive rows) differ
    constexpr int NT = (KS + S - 1) / S;               // taps per dimension a pixel meets
    constexpr int PL = 256 / SQ;                       // pixel lanes
    constexpr bool WREG = NT * NT <= 9;                // tap weights in registers (else in LDS)
    constexpr int T = KS == 5 ? 2 : 4;                 // stride 1: a thread takes runs of T pixels along x (register window over the taps)
    // 5 x 5 stride 1: TWO lanes share a run — lane half h takes the tap rows ky = KH h ... KH h + KH - 1 (3 + 2) — so that a lane
    // carries 15 tap-gradient accumulators instead of 25 (60 registers instead of 100: 170 instead of 256 per lane, three
    // workgroups per CU instead of two at one wave per SIMD each); the halves of the input gradient meet through one shuffle
    constexpr bool HS = S == 1 && KS == 5;
    constexpr int KH = (KS + 1) / 2;
    constexpr int AJ = HS ? KH : NT;                   // accumulator rows per lane
    constexpr int NCLS = HS ? 2 : S * S;               // lane classes whose accumulator slots mean different taps (halves / parities)
    constexpr int SMEM = OR * PITCH > 512 ? OR * PITCH : 512;
    __shared__ f32x4 tile[SMEM];                        // the dd region; after the last tile, the reduction buffer
    __shared__ f32x4 wl[WREG ? 1 : KK * SQ];
    f32x4* red = tile;
    f64x4* red64 = reinterpret_cast<f64x4*>(tile);
    const int tid = threadIdx.x;
    const int cq_l = tid % SQ, pl = tid / SQ;
    const int slab = blockIdx.x % a.nslab, wslot = blockIdx.x / a.nslab;
    const int c = (slab * SQ + cq_l) * 4;
    const bool cv = c < a.C;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 cA = zero, cs1 = zero, cmu = zero, cQ = zero, mu1 = zero, rs1 = zero, a1 = zero, b1 = zero;
    if (cv) {
        cA = *reinterpret_cast<const f32x4*>(a.coef2 + c); cs1 = *reinterpret_cast<const f32x4*>(a.coef2 + a.C + c);
        cmu = *reinterpret_cast<const f32x4*>(a.coef2 + 2 * a.C + c); cQ = *reinterpret_cast<const f32x4*>(a.coef2 + 3 * a.C + c);
        if (BN1) {
            mu1 = *reinterpret_cast<const f32x4*>(a.act1 + c); rs1 = *reinterpret_cast<const f32x4*>(a.act1 + a.C + c);
            a1 = *reinterpret_cast<const f32x4*>(a.act1 + 2 * a.C + c); b1 = *reinterpret_cast<const f32x4*>(a.act1 + 3 * a.C + c);
        }
    }
    // VE: the tile's raw expansion [256 pixels][SQ quads], formed on the matrix pipe while the dd region is staged
    __shared__ f32x4 es[VE ? TS * TS * SQ : 1];
    f32x4 wa[VE ? SQ / 4 : 1][VE ? VE : 1];
    if (VE) {
#pragma unroll
        for (int ct = 0; ct < SQ / 4; ++ct)
#pragma unroll
            for (int ci = 0; ci < VE; ++ci) {
                const int ch = slab * SQ * 4 + ct * 16 + (tid & 15), k = ci * 16 + 4 * ((tid & 63) >> 4);
                wa[VE ? ct : 0][VE ? ci : 0] = ch < a.C && k < a.ve.cin ? *reinterpret_cast<const f32x4*>(a.ve.W1 + (long)ch * a.ve.cin + k) : zero;
            }
    }
    f32x4 acc1[W1G ? SQ / 4 : 1][W1G ? VE : 1];
#pragma unroll
    for (int ct = 0; ct < (W1G ? SQ / 4 : 1); ++ct)
#pragma unroll
        for (int ci = 0; ci < (W1G ? VE : 1); ++ci) acc1[ct][ci] = zero;
    // this thread's parity class and the first tap it meets in each dimension
    const int cls = S == 1 ? 0 : (pl & 3);
    const int py = S == 1 ? 0 : (cls >> 1), px = S == 1 ? 0 : (cls & 1);
    const int ky0 = (py + P) % S, kx0 = (px + P) % S;
    f32x4 wreg[WREG ? NT : 1][WREG ? NT : 1];
    if (WREG) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int ky = ky0 + S * j, kx = kx0 + S * i;
                wreg[WREG ? j : 0][WREG ? i : 0] = (cv && ky < KS && kx < KS) ? *reinterpret_cast<const f32x4*>(a.Wt + (long)(ky * KS + kx) * a.C + c) : zero;
            }
    } else {
        for (int t = pl; t < KK; t += PL) wl[t * SQ + cq_l] = cv ? *reinterpret_cast<const f32x4*>(a.Wt + (long)t * a.C + c) : zero;
    }
    f32x4 acc[AJ][NT];
#pragma unroll
    for (int j = 0; j < AJ; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = zero;
    f64x4 S1 = (f64x4){0.0, 0.0, 0.0, 0.0}, S2 = S1;

    const long obytes = (long)a.B * a.Ho * a.Wo * a.ldo * 4;      // < 2^31: checked on the host
    const __amdgpu_buffer_rsrc_t g2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.G2), 0, (int)obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.D), 0, (int)obytes, 0x00020000);
    const int n_items = a.B * a.tiles_y * a.tiles_x;
    for (int item = wslot; item < n_items; item += a.wgs_per_slab) {
        const int tx = item % a.tiles_x, ty = (item / a.tiles_x) % a.tiles_y, b = item / (a.tiles_x * a.tiles_y);
        const int iy0 = ty * TS, ix0 = tx * TS;
        const int lo_y = iy0 / S - LO, lo_x = ix0 / S - LO;
        f32x4 s1f = zero, s2f = zero;      // this tile's share of the two sums (at most 16 values per lane), then float64
        int wq = cq_l;                      // index of this thread's weights in LDS, opaque to the compiler: it would otherwise hoist
        asm volatile("" : "+v"(wq));        // all 25 LDS weight reads out of the tile loop into 100 registers
        // ---- phase 1: dd of the tile's output region -> LDS (zero outside the map / beyond the channels)
        // (U loads of each tensor in flight per round: the region is 12.5 float4 per thread and tensor for a 5 x 5 stride-1 tile
        //  with 32-channel slabs — at U = 4 that was four memory round trips per tile on a kernel whose tiles are short)
        constexpr int NIDX = OR * OR * SQ, U = (!HS && (NIDX + 255) / 256 > 8) ? 7 : 4;
        for (int i0 = 0; i0 < NIDX; i0 += 256 * U) {
            f32x4 gv[U], dv[U];
            bool in[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + u * 256 + tid;
                const int pix = idx / SQ;                 // idx % SQ == cq_l (256 is a multiple of SQ)
                const int r = pix / OR, cc = pix - r * OR;
                const int oy = lo_y + r, ox = lo_x + cc;
                in[u] = idx < NIDX && cv && oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
                const int off = in[u] ? (((b * a.Ho + oy) * a.Wo + ox) * a.ldo + c) * 4 : (int)0x80000000;
                gv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g2r, off, 0, 0));
                dv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, off, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + u * 256 + tid;
                const int pix = idx / SQ;
                const int r = pix / OR, cc = pix - r * OR;
                if (idx < NIDX) tile[r * PITCH + cc * SQ + cq_l] = in[u] ? bnb4(gv[u], dv[u], cA, cs1, cmu, cQ) : zero;
            }
        }
        if constexpr (VE) {
            const int wave = tid >> 6, j = tid & 15, kk = (tid & 63) >> 4;
#pragma unroll 1
            for (int rt = wave; rt < TS * TS / 16; rt += 4) {
                const int p = rt * 16 + j;                       // tile pixel of this lane's column
                const int iy = iy0 + p / TS, ix = ix0 + p % TS;
                const bool inb = iy < a.H && ix < a.W;
                f32x4 xb[VE ? VE : 1];
#pragma unroll
                for (int ci = 0; ci < VE; ++ci)
                    xb[ci] = inb && ci * 16 + 4 * kk < a.ve.cin
                                 ? *reinterpret_cast<const f32x4*>(a.ve.X + (((long)b * a.H + iy) * a.W + ix) * a.ve.cin + ci * 16 + 4 * kk) : zero;
#pragma unroll
                for (int ct = 0; ct < SQ / 4; ++ct) es[p * SQ + ct * 4 + kk] = virt_e_tile<VE ? VE : 1>(wa[ct], xb);
            }
        }
        __syncthreads();
        // ---- phase 2
        if (HS) {
            const int half = pl & 1, prl = pl >> 1;
            constexpr int RPS = (PL / 2) / (TS / T);       // rows per sweep
#pragma unroll 1
            for (int sw = 0; sw < TS / RPS; ++sw) {
                const int iy_l = sw * RPS + prl % RPS, ix_l = (prl / RPS) * T;
                const int iy = iy0 + iy_l;
                const long prow = ((long)b * a.H + iy) * a.W + ix0 + ix_l;
                bool pin[T];
                f32x4 e4[T], av[T], de[T];
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    pin[i] = cv && iy < a.H && ix0 + ix_l + i < a.W;
                    e4[i] = pin[i] ? *reinterpret_cast<const f32x4*>(a.E + (prow + i) * a.lde + c) : zero;
                }
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    av[i] = e4[i];
                    if (BN1) {
                        const f32x4 pre = act4(e4[i], a1, b1, false);
                        av[i] = (f32x4){fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f), fmaxf(pre.z, 0.f), fmaxf(pre.w, 0.f)};
                    }
                    if (!pin[i]) av[i] = zero;
                    de[i] = zero;
                }
                const int ky_first = half * KH;
                int lb = (iy_l + 2 * P - ky_first) * PITCH + ix_l * SQ + cq_l;
#pragma unroll
                for (int jj = 0; jj < KH; ++jj) {
                    const bool kyv = ky_first + jj < KS;           // (the second half's last row does not exist)
                    const int base = kyv ? lb - jj * PITCH : lb;
                    f32x4 win[T + KS - 1];
#pragma unroll
                    for (int j = 0; j < T + KS - 1; ++j) {
                        win[j] = tile[base + j * SQ];
                        if (!kyv) win[j] = zero;
                    }
                    const int wrow = (kyv ? ky_first + jj : 0) * KS;
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const f32x4 w = wl[(wrow + kx) * SQ + wq];
#pragma unroll
                        for (int i = 0; i < T; ++i) {
                            const f32x4 v = win[i + KS - 1 - kx];
                            de[i] += v * w;
                            acc[jj][kx] += v * av[i];
                        }
                    }
                    asm volatile("" : "+v"(lb), "+v"(wq) : "v"(de[0]), "v"(de[T - 1]), "v"(acc[jj][0]), "v"(acc[jj][1]), "v"(acc[jj][2]), "v"(acc[jj][KS - 2]), "v"(acc[jj][KS - 1]));
                }
#pragma unroll
                for (int i = 0; i < T; ++i) {                      // the other half's tap rows (its lane is SQ lanes away)
                    de[i].x += __shfl_xor(de[i].x, SQ, 64); de[i].y += __shfl_xor(de[i].y, SQ, 64);
                    de[i].z += __shfl_xor(de[i].z, SQ, 64); de[i].w += __shfl_xor(de[i].w, SQ, 64);
                }
                if (half == 0) {
#pragma unroll
                    for (int i = 0; i < T; ++i) {
                        if (!pin[i]) continue;
                        if (BN1) {
                            const f32x4 pre = act4(e4[i], a1, b1, false);
                            const f32x4 g1 = (f32x4){pre.x > 0.f ? de[i].x : 0.f, pre.y > 0.f ? de[i].y : 0.f, pre.z > 0.f ? de[i].z : 0.f, pre.w > 0.f ? de[i].w : 0.f};
                            *reinterpret_cast<f32x4*>(a.Y + (prow + i) * a.ldy + c) = g1;
                            s1f += g1;
                            s2f += g1 * ((e4[i] - mu1) * rs1);
                        } else {
                            f32x4 r4 = zero;
                            if (a.R) r4 = *reinterpret_cast<const f32x4*>(a.R + (prow + i) * a.ldr + c);
                            *reinterpret_cast<f32x4*>(a.Y + (prow + i) * a.ldy + c) = de[i] + r4;
                        }
                    }
                }
            }
        } else if (S == 1) {
            // runs of T pixels along x: per tap row a window of T + KS - 1 region columns is read once and serves all T x KS
            // (pixel, tap) pairs; consecutive pixel lanes are consecutive rows
            constexpr int RPS = PL / (TS / T);             // rows per sweep
#pragma unroll 1
            for (int sw = 0; sw < TS / RPS; ++sw) {
                const int iy_l = sw * RPS + pl % RPS, ix_l = (pl / RPS) * T;
                const int iy = iy0 + iy_l;
                const long prow = ((long)b * a.H + iy) * a.W + ix0 + ix_l;
                bool pin[T];
                f32x4 e4[T], av[T], de[T];
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    pin[i] = cv && iy < a.H && ix0 + ix_l + i < a.W;
                    e4[i] = pin[i] ? *reinterpret_cast<const f32x4*>(a.E + (prow + i) * a.lde + c) : zero;
                }
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    av[i] = e4[i];
                    if (BN1) {
                        const f32x4 pre = act4(e4[i], a1, b1, false);
                        av[i] = (f32x4){fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f), fmaxf(pre.z, 0.f), fmaxf(pre.w, 0.f)};
                    }
                    if (!pin[i]) av[i] = zero;
                    de[i] = zero;
                }
                int lb = (iy_l + 2 * P) * PITCH + ix_l * SQ + cq_l;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const int base = lb - ky * PITCH;
                    f32x4 win[T + KS - 1];
#pragma unroll
                    for (int j = 0; j < T + KS - 1; ++j) win[j] = tile[base + j * SQ];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const f32x4 w = WREG ? wreg[WREG ? ky : 0][WREG ? kx : 0] : wl[(ky * KS + kx) * SQ + wq];
#pragma unroll
                        for (int i = 0; i < T; ++i) {
                            const f32x4 v = win[i + KS - 1 - kx];
                            de[i] += v * w;
                            acc[HS ? 0 : ky][kx] += v * av[i];
                        }
                    }
                    // one tap row's window in flight: hipcc hoists all KS of them (100+ registers, spills at two workgroups per
                    // CU) and neither sched_barrier nor the loop structure stops it; a data dependence of the next row's address does
                    if (KS == 5) asm volatile("" : "+v"(lb), "+v"(wq) : "v"(de[0]), "v"(de[T - 1]), "v"(acc[HS ? 0 : ky][0]), "v"(acc[HS ? 0 : ky][1]), "v"(acc[HS ? 0 : ky][2]), "v"(acc[HS ? 0 : ky][KS - 2]), "v"(acc[HS ? 0 : ky][KS - 1]));
                    else asm volatile("" : "+v"(lb), "+v"(wq) : "v"(de[0]), "v"(de[1]), "v"(de[T - 2]), "v"(de[T - 1]), "v"(acc[ky][0]), "v"(acc[ky][1]), "v"(acc[ky][KS - 1]));
                }
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    if (!pin[i]) continue;
                    if (BN1) {
                        const f32x4 pre = act4(e4[i], a1, b1, false);
                        const f32x4 g1 = (f32x4){pre.x > 0.f ? de[i].x : 0.f, pre.y > 0.f ? de[i].y : 0.f, pre.z > 0.f ? de[i].z : 0.f, pre.w > 0.f ? de[i].w : 0.f};
                        *reinterpret_cast<f32x4*>(a.Y + (prow + i) * a.ldy + c) = g1;
                        s1f += g1;
                        s2f += g1 * ((e4[i] - mu1) * rs1);
                    } else {
                        f32x4 r4 = zero;
                        if (a.R) r4 = *reinterpret_cast<const f32x4*>(a.R + (prow + i) * a.ldr + c);
                        *reinterpret_cast<f32x4*>(a.Y + (prow + i) * a.ldy + c) = de[i] + r4;
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int sw = 0; sw < SQ; ++sw) {
                const int id = sw * (PL / 4) + (pl >> 2);      // 2 x 2 pixel group of the tile
                const int iy_l = 2 * (id / (TS / 2)) + py, ix_l = 2 * (id % (TS / 2)) + px;
                const int iy = iy0 + iy_l, ix = ix0 + ix_l;
                const bool pin = cv && iy < a.H && ix < a.W;
                const long prow = ((long)b * a.H + iy) * a.W + ix;
                f32x4 e4 = zero, r4 = zero;
                if (pin) {
                    if constexpr (VE) {
                        e4 = es[(iy_l * TS + ix_l) * SQ + cq_l];
                    } else {
                        e4 = *reinterpret_cast<const f32x4*>(a.E + prow * a.lde + c);
                    }
                    if (!BN1 && a.R) r4 = *reinterpret_cast<const f32x4*>(a.R + prow * a.ldr + c);
                }
                f32x4 av = e4;                                 // the depthwise conv's operand at this pixel
                f32x4 pre = zero;
                if (BN1) {
                    pre = act4(e4, a1, b1, false);
                    av = (f32x4){fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f), fmaxf(pre.z, 0.f), fmaxf(pre.w, 0.f)};
                }
                if (!pin) av = zero;
                f32x4 de = zero;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int ky = ky0 + S * j;
                    const bool jv = ky < KS;
                    const int rr = jv ? (iy_l + P - ky) / S + LO : 0;       // (iy_l + P - ky) is a multiple of S for this thread's class
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const int kx = kx0 + S * i;
                        const bool tv = jv && kx < KS;
                        const int cc = tv ? (ix_l + P - kx) / S + LO : 0;
                        f32x4 v = tile[rr * PITCH + cc * SQ + cq_l];
                        if (!tv) v = zero;
                        const f32x4 w = WREG ? wreg[WREG ? j : 0][WREG ? i : 0] : wl[(tv ? ky * KS + kx : 0) * SQ + wq];
                        de += v * w;
                        acc[j][i] += v * av;
                    }
                }
                f32x4 g1 = zero;
                if (pin) {
                    if (BN1) {
                        g1 = (f32x4){pre.x > 0.f ? de.x : 0.f, pre.y > 0.f ? de.y : 0.f, pre.z > 0.f ? de.z : 0.f, pre.w > 0.f ? de.w : 0.f};
                        *reinterpret_cast<f32x4*>(a.Y + prow * a.ldy + c) = g1;
                        s1f += g1;
                        s2f += g1 * ((e4 - mu1) * rs1);
                    } else {
                        *reinterpret_cast<f32x4*>(a.Y + prow * a.ldy + c) = de + r4;
                    }
                }
                if constexpr (W1G) es[(iy_l * TS + ix_l) * SQ + cq_l] = g1;      // (this thread's own slot: e was read from it above)
            }
            if constexpr (W1G) {
                // the expansion's weight gradient while g1 is on chip: sum over the tile's pixels of g1[p][c] x[p][k] — lane (channel i,
                // pixel kk) reads g1 from LDS, lane (input channel j, pixel kk) reads x; four pixels per MFMA, the groups dealt to the waves
                {
                    __syncthreads();
                    const int wave = tid >> 6, lj = tid & 15, kk = (tid & 63) >> 4;
#pragma unroll 1
                    for (int pg = wave; pg < TS * TS / 4; pg += 4) {
                        const int p = pg * 4 + kk;
                        const int iy = iy0 + p / TS, ix = ix0 + p % TS;
                        const bool inb = iy < a.H && ix < a.W;
                        float bx[VE];
#pragma unroll
                        for (int ci = 0; ci < VE; ++ci) {
                            const int k = ci * 16 + lj;
                            bx[ci] = inb && k < a.ve.cin ? a.ve.X[(((long)b * a.H + iy) * a.W + ix) * a.ve.cin + k] : 0.f;
                        }
#pragma unroll
                        for (int ct = 0; ct < SQ / 4; ++ct) {
                            const float gv = reinterpret_cast<const float*>(&es[p * SQ + ct * 4 + (lj >> 2)])[lj & 3];
#pragma unroll
                            for (int ci = 0; ci < VE; ++ci) acc1[ct][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv, bx[ci], acc1[ct][ci], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (BN1) { S1 += to_f64(s1f); S2 += to_f64(s2f); }
        __syncthreads();      // the next item overwrites the tile
    }
    // ---- the workgroup's partial tap gradients: slot (j, i) of the threads of one class and channel quad, added in lane order
#pragma unroll
    for (int j = 0; j < AJ; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            __syncthreads();
            red[tid] = acc[j][i];
            __syncthreads();
            if (tid < NCLS * SQ) {
                const int rc = tid / SQ, q = tid % SQ;
                f32x4 sum = zero;
                for (int g = 0; g < PL / NCLS; ++g) sum += red[(g * NCLS + rc) * SQ + q];
                const int rpy = S == 1 ? 0 : (rc >> 1), rpx = S == 1 ? 0 : (rc & 1);
                const int ky = HS ? rc * KH + j : (rpy + P) % S + S * j, kx = (rpx + P) % S + S * i;
                const int cc = (slab * SQ + q) * 4;
                if (ky < KS && kx < KS && cc < a.C)
                    *reinterpret_cast<f32x4*>(a.ptaps + ((long)wslot * KK + ky * KS + kx) * a.C + cc) = sum;
            }
        }
    if (BN1) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
            red64[tid] = which == 0 ? S1 : S2;
            __syncthreads();
            if (tid < SQ) {
                f64x4 sum = (f64x4){0.0, 0.0, 0.0, 0.0};
                for (int g = 0; g < PL; ++g) sum += red64[g * SQ + tid];
                const int cc = (slab * SQ + tid) * 4;
                if (cc < a.C) *reinterpret_cast<f64x4*>(a.psums + ((long)wslot * 2 + which) * a.C + cc) = sum;
            }
        }
    }
    if constexpr (W1G) {
        {      // the four waves' shares added in wave order, one partial per workgroup
            const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
            for (int ct = 0; ct < SQ / 4; ++ct)
#pragma unroll
                for (int ci = 0; ci < VE; ++ci) {
                    __syncthreads();
                    if (wave > 0) red[(wave - 1) * 64 + lane] = acc1[ct][ci];
                    __syncthreads();
                    if (wave == 0) {
                        f32x4 v = acc1[ct][ci];
#pragma unroll
                        for (int w = 0; w < 3; ++w) v += red[w * 64 + lane];
                        const int k = ci * 16 + (lane & 15);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ch = slab * SQ * 4 + ct * 16 + 4 * (lane >> 4) + r;      // lane (k, q), component r = channel 4 q + r
                            if (ch < a.C && k < a.ve.cin) a.pw1[((long)wslot * a.C + ch) * a.ve.cin + k] = v[r];
                        }
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise conv FORWARD of the block-fused step: Y = DW act(X) with the input activation applied once per input pixel as the
// tile's input region goes to LDS (zero outside the map: the padding pads the ACTIVATION), and the column sums sum(y), sum(y^2)
// of the raw output — float64 per thread over all tiles of a persistent workgroup, ONE partial row per workgroup (the
// thread-per-strip dw_stat_kernel wrote one per 256 threads: 58 MB of partials for a 672-channel 16 x 16 map, and ran at one
// wave per SIMD).  Same slab / tile / lane scheme as dw_bwd_kernel; output tiles of 16 x 16 (stride 1) or 8 x 8 (stride 2).
struct DwFwdArgs {
    const float* X;
    ActIn in;
    const float* Wt;      // [KS*KS][C]
    float* Y;
    double* psums;        // [wgs_per_slab][2][C]
    int ldx, ldy;
    int B, H, W, Ho, Wo, C;
    int tiles_x, tiles_y, wgs_per_slab, nslab;
    VirtE ve;             // VE: X is not read, the operand is act(ve.X ve.W1^T)
};

template <int KS, int S, int SQ, int VE = 0, int TS1 = 16>      // VE = NC chunks of 16 input channels of a virtual expansion (0: X is read)
__global__ __launch_bounds__(256, 2) void dw_fwd_kernel(DwFwdArgs a) {
    constexpr int P = KS / 2, KK = KS * KS;
    constexpr int TO = S == 1 ? TS1 : 8;               // output tile side (TS1 = 8: stride-1 maps of at most 8 x 8 pixels, see dw_bwd_kernel)
    constexpr int IR = (TO - 1) * S + KS;              // input region side
    constexpr int PITCH = (IR + 1) * SQ;
    constexpr int PL = 256 / SQ;
    constexpr int T = S == 1 ? (TO == 8 ? 2 : 4) : 1;  // output pixels per thread and sweep (a run along x)
    constexpr int SMEM = IR * PITCH > 512 ? IR * PITCH : 512;
    __shared__ f32x4 tile[SMEM];
    f64x4* red64 = reinterpret_cast<f64x4*>(tile);
    const int tid = threadIdx.x;
    const int cq_l = tid % SQ, pl = tid / SQ;
    const int slab = blockIdx.x % a.nslab, wslot = blockIdx.x / a.nslab;
    const int c = (slab * SQ + cq_l) * 4;
    const bool cv = c < a.C;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool affine = a.in.a != nullptr;
    f32x4 ia = zero, ib = zero;
    if (affine && cv) { ia = *reinterpret_cast<const f32x4*>(a.in.a + c); ib = *reinterpret_cast<const f32x4*>(a.in.b + c); }
    constexpr bool WREG = KK <= 9;                      // 3 x 3 taps in registers, 5 x 5 in LDS (100 registers otherwise)
    __shared__ f32x4 wl[WREG ? 1 : KK * SQ];
    f32x4 wr[WREG ? KK : 1];
    if (WREG) {
#pragma unroll
        for (int t = 0; t < KK; ++t) wr[WREG ? t : 0] = cv ? *reinterpret_cast<const f32x4*>(a.Wt + (long)t * a.C + c) : zero;
    } else {
        for (int t = pl; t < KK; t += PL) wl[t * SQ + cq_l] = cv ? *reinterpret_cast<const f32x4*>(a.Wt + (long)t * a.C + c) : zero;
    }
    f64x4 S1 = (f64x4){0.0, 0.0, 0.0, 0.0}, S2 = S1;
    // VE: W1 fragments (lane (channel, k quarter)) and the activation of the two channel quads this lane's results belong to
    f32x4 wa[VE ? SQ / 4 : 1][VE ? VE : 1], va[VE ? SQ / 4 : 1], vb[VE ? SQ / 4 : 1];
    if (VE) {
#pragma unroll
        for (int ct = 0; ct < SQ / 4; ++ct) {
#pragma unroll
            for (int ci = 0; ci < VE; ++ci) {
                const int ch = slab * SQ * 4 + ct * 16 + (tid & 15), k = ci * 16 + 4 * ((tid & 63) >> 4);
                wa[VE ? ct : 0][VE ? ci : 0] = ch < a.C && k < a.ve.cin ? *reinterpret_cast<const f32x4*>(a.ve.W1 + (long)ch * a.ve.cin + k) : zero;
            }
            const int c4 = (slab * SQ + ct * 4 + ((tid & 63) >> 4)) * 4;
            va[VE ? ct : 0] = c4 < a.C ? *reinterpret_cast<const f32x4*>(a.in.a + c4) : zero;
            vb[VE ? ct : 0] = c4 < a.C ? *reinterpret_cast<const f32x4*>(a.in.b + c4) : zero;
        }
    }
    const long xbytes = (long)a.B * a.H * a.W * (VE ? a.ve.cin : a.ldx) * 4;      // < 2^31: checked on the host
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(VE ? a.ve.X : a.X), 0, (int)xbytes, 0x00020000);
    const int n_items = a.B * a.tiles_y * a.tiles_x;
    for (int item = wslot; item < n_items; item += a.wgs_per_slab) {
        const int tx = item % a.tiles_x, ty = (item / a.tiles_x) % a.tiles_y, b = item / (a.tiles_x * a.tiles_y);
        const int oy0 = ty * TO, ox0 = tx * TO;
        const int iy0 = oy0 * S - P, ix0 = ox0 * S - P;
        f32x4 s1f = zero, s2f = zero;      // this tile's share of the sums (at most 16 values per lane), then float64
        int wq = cq_l;
        asm volatile("" : "+v"(wq));        // (LDS weight index, opaque: see dw_bwd_kernel)
        constexpr int NIDX = IR * IR * SQ, U = 4;
        if constexpr (VE) {
            const int wave = tid >> 6, j = tid & 15, kk = (tid & 63) >> 4;
#pragma unroll 1
            for (int rt = wave; rt < (IR * IR + 15) / 16; rt += 4) {
                const int pix = rt * 16 + j;                     // region pixel of this lane's column
                const int r = pix / IR, cc = pix - r * IR;
                const int y = iy0 + r, x = ix0 + cc;
                const bool inb = pix < IR * IR && y >= 0 && y < a.H && x >= 0 && x < a.W;
                f32x4 xb[VE ? VE : 1];
#pragma unroll
                for (int ci = 0; ci < VE; ++ci) {
                    const int off = inb && ci * 16 + 4 * kk < a.ve.cin ? (((b * a.H + y) * a.W + x) * a.ve.cin + ci * 16 + 4 * kk) * 4 : (int)0x80000000;
                    xb[ci] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
                }
#pragma unroll
                for (int ct = 0; ct < SQ / 4; ++ct) {
                    const f32x4 v = act4(virt_e_tile<VE ? VE : 1>(wa[ct], xb), va[ct], vb[ct], a.in.relu != 0);
                    const bool cok = (slab * SQ + ct * 4 + kk) * 4 < a.C;
                    if (pix < IR * IR) tile[r * PITCH + cc * SQ + ct * 4 + kk] = inb && cok ? v : zero;
                }
            }
        } else
        for (int i0 = 0; i0 < NIDX; i0 += 256 * U) {
            f32x4 xv[U];
            bool in[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + u * 256 + tid;
                const int pix = idx / SQ;
                const int r = pix / IR, cc = pix - r * IR;
                const int y = iy0 + r, x = ix0 + cc;
                in[u] = idx < NIDX && cv && y >= 0 && y < a.H && x >= 0 && x < a.W;
                const int off = in[u] ? (((b * a.H + y) * a.W + x) * a.ldx + c) * 4 : (int)0x80000000;
                xv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + u * 256 + tid;
                const int pix = idx / SQ;
                const int r = pix / IR, cc = pix - r * IR;
                f32x4 v = xv[u];
                if (affine) v = act4(v, ia, ib, a.in.relu != 0);
                if (idx < NIDX) tile[r * PITCH + cc * SQ + cq_l] = in[u] ? v : zero;
            }
        }
        __syncthreads();
        if (S == 1) {
            constexpr int RPS = PL / (TO / T);             // output rows per sweep; consecutive pixel lanes are consecutive rows
#pragma unroll 1
            for (int sw = 0; sw < TO / RPS; ++sw) {
                const int oy_l = sw * RPS + pl % RPS, ox_l = (pl / RPS) * T;
                f32x4 out[T];
#pragma unroll
                for (int i = 0; i < T; ++i) out[i] = zero;
                int lb = oy_l * PITCH + ox_l * SQ + cq_l;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const int base = lb + ky * PITCH;
                    f32x4 win[T + KS - 1];
#pragma unroll
                    for (int j = 0; j < T + KS - 1; ++j) win[j] = tile[base + j * SQ];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const f32x4 w = WREG ? wr[WREG ? ky * KS + kx : 0] : wl[(ky * KS + kx) * SQ + wq];
#pragma unroll
                        for (int i = 0; i < T; ++i) out[i] += win[i + kx] * w;
                    }
                    asm volatile("" : "+v"(lb), "+v"(wq) : "v"(out[0]), "v"(out[1]), "v"(out[T - 2]), "v"(out[T - 1]));      // (see dw_bwd_kernel)
                }
                const int oy = oy0 + oy_l;
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    const int ox = ox0 + ox_l + i;
                    if (cv && oy < a.Ho && ox < a.Wo) {
                        *reinterpret_cast<f32x4*>(a.Y + (((long)b * a.Ho + oy) * a.Wo + ox) * a.ldy + c) = out[i];
                        s1f += out[i];
                        s2f += out[i] * out[i];
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int sw = 0; sw < TO * TO / PL; ++sw) {
                const int id = sw * PL + pl;
                const int oy_l = id / TO, ox_l = id % TO;
                f32x4 out = zero;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
                        out += tile[(oy_l * S + ky) * PITCH + (ox_l * S + kx) * SQ + cq_l] * (WREG ? wr[WREG ? ky * KS + kx : 0] : wl[(ky * KS + kx) * SQ + wq]);
                const int oy = oy0 + oy_l, ox = ox0 + ox_l;
                if (cv && oy < a.Ho && ox < a.Wo) {
                    *reinterpret_cast<f32x4*>(a.Y + (((long)b * a.Ho + oy) * a.Wo + ox) * a.ldy + c) = out;
                    s1f += out;
                    s2f += out * out;
                }
            }
        }
        S1 += to_f64(s1f);
        S2 += to_f64(s2f);
        __syncthreads();
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
        red64[tid] = which == 0 ? S1 : S2;
        __syncthreads();
        if (tid < SQ) {
            f64x4 sum = (f64x4){0.0, 0.0, 0.0, 0.0};
            for (int g = 0; g < PL; ++g) sum += red64[g * SQ + tid];
            const int cc = (slab * SQ + tid) * 4;
            if (cc < a.C) *reinterpret_cast<f64x4*>(a.psums + ((long)wslot * 2 + which) * a.C + cc) = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

template <bool MS>
void launch_pw_bwd(const PwBwdArgs& a, dim3 grid, int nt, hipStream_t s) {
    switch (nt) {
        case 1: hipLaunchKernelGGL((pw_bwd_kernel<1, MS>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((pw_bwd_kernel<2, MS>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((pw_bwd_kernel<3, MS>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((pw_bwd_kernel<4, MS>), grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((pw_bwd_kernel<6, MS>), grid, dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL((pw_bwd_kernel<7, MS>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((pw_bwd_kernel<8, MS>), grid, dim3(256), 0, s, a); break;
    }
}

void launch_pw_stat(const PwStatArgs& a, dim3 grid, int nt, hipStream_t s) {
    switch (nt) {
        case 1: hipLaunchKernelGGL((pw_stat_kernel<1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((pw_stat_kernel<2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((pw_stat_kernel<3>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((pw_stat_kernel<4>), grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((pw_stat_kernel<6>), grid, dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL((pw_stat_kernel<7>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((pw_stat_kernel<8>), grid, dim3(256), 0, s, a); break;
    }
}

// the workspace of one block call, cut into the regions its kernels use side by side
struct BlockWs {
    double* col;      // column-sum partials of whichever producer runs (stream order: its finalize has read them before the next writes)
    float* wg;        // pointwise weight-gradient row slices
    float* taps;      // depthwise tap-gradient partials
    float* coef;      // 3 x [4][Cmax] BnbIn coefficients
    size_t col_bytes, wg_bytes, taps_bytes, coef_bytes, total;
};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// persistent workgroups of the depthwise kernels, all slabs together: the 512 a device holds at two per CU — one round, no tail, and a
// quarter of the partial rows of the 2 048 this started with (15.3 -> 15.0 ms per step; 384 / 768 / 1 024: 15.1)
#ifndef FEAR_DW_WGS
#define FEAR_DW_WGS 512
#endif
int dw_bwd_wgs_per_slab(int n_items, int nslab) {
    int target = FEAR_DW_WGS / nslab;
    if (target < 1) target = 1;
    if (target >= n_items) return n_items;
    const int per = (n_items + target - 1) / target;      // items per workgroup, then as few workgroups as that needs
    return (n_items + per - 1) / per;
}
// 32-channel slabs (128-byte rows per pixel and tensor) from 64 channels up, a ragged last slab included (144 = 4.5 slabs: 11 % of
// the lanes idle, against 64-byte segments for all of them with 16-channel slabs: the 24 -> 144 block's pass ran at 3 TB/s)
int dw_bwd_sq(int C) { return C >= 64 ? 8 : 4; }

BlockWs block_ws(long rows_in, long rows_out, int cin, int cexp, int cout, int k, float* base) {
    BlockWs w{};
    const long rows = rows_in > rows_out ? rows_in : rows_out;
    const int cmax = cexp > cout ? (cexp > cin ? cexp : cin) : (cout > cin ? cout : cin);
    size_t col = fear_train_stats_workspace_bytes(rows, cmax);
    const size_t colr = (size_t)col_blocks(rows) * 2 * cmax * sizeof(double);
    if (colr > col) col = colr;
    const size_t lds = rows <= FEAR_GEMM_LDS_MAX_ROWS ? (size_t)((rows + 63) / 64) * 2 * cmax * sizeof(double) : 0;      // gemm_lds_kernel: 64-row blocks
    if (lds > col) col = lds;
    const size_t dwp = (size_t)2048 * 2 * cmax * sizeof(double);           // dw_bwd_kernel's sums: <= 2048 workgroups per slab
    if (dwp > col) col = dwp;
    w.col_bytes = align256(col);
    const size_t nk = (size_t)cexp * (cin > cout ? cin : cout);
    size_t wg = (size_t)wgrad_slices(rows) * nk * sizeof(float);
    size_t more = (size_t)1024 * nk * sizeof(float);      // room for wgrad_impl's finer row slicing (up to 1 024 slices of small partials)
    if (more > ((size_t)32 << 20)) more = (size_t)32 << 20;
    if (more > wg) wg = more;
    w.wg_bytes = align256(wg);
    const int sq = dw_bwd_sq(cexp);
    const int nslab = (cexp / 4 + sq - 1) / sq;
    const int wps = FEAR_DW_WGS / nslab > 1 ? FEAR_DW_WGS / nslab : 1;    // most workgroups per slab dw_bwd_wgs_per_slab hands out
    w.taps_bytes = align256((size_t)wps * k * k * cexp * sizeof(float));
    // (also the Gram matrix | column sums of a virtual expansion's input in the forward: up to 32 * 32 + 32 floats)
    w.coef_bytes = align256((size_t)(3 * 4 * cmax > 1056 ? 3 * 4 * cmax : 1056) * sizeof(float));
    w.total = w.col_bytes + w.wg_bytes + w.taps_bytes + w.coef_bytes;
    // (a size query passes base = nullptr: no arithmetic on it — an offset applied to a null pointer is undefined behaviour and traps
    //  in the UBSan build, libfear_hip_debug.so)
    w.col = nullptr; w.wg = nullptr; w.taps = nullptr; w.coef = nullptr;
    if (base) {
        char* p = reinterpret_cast<char*>(base);
        w.col = reinterpret_cast<double*>(p); p += w.col_bytes;
        w.wg = reinterpret_cast<float*>(p); p += w.wg_bytes;
        w.taps = reinterpret_cast<float*>(p); p += w.taps_bytes;
        w.coef = reinterpret_cast<float*>(p);
    }
    return w;
}

// ---- SyncBatchNorm hook (include/fear_train.h, fear_train_sync_bind): streams bound to a FearSync.  A handful of entries, looked up
// once per finalize under a mutex (the host side of a step is one thread per rank; the lock is for whoever drives two devices).
struct SyncSlot { hipStream_t s; FearSync sy; bool used; };
SyncSlot g_sync_slots[16];
std::mutex g_sync_mutex;
bool sync_of(hipStream_t s, FearSync* out) {
    std::lock_guard<std::mutex> lock(g_sync_mutex);
    for (const SyncSlot& e : g_sync_slots)
        if (e.used && e.s == s) { *out = e.sy; return true; }
    return false;
}
// the caller's all-reduce of n elements of the sync buffer; a failure is reported by the entry point's LAUNCH_CHECK
void sync_all_reduce(const FearSync& sy, long n, int is_f32, hipStream_t s) {
    if ((size_t)n * (is_f32 ? 4 : 8) > sy.buf_bytes || sy.all_reduce(sy.user, sy.buf, n, is_f32, s) != 0) sync_failed = 1;
}

// column-sum partials [blocks][2][C] -> mean | rstd | a | b (vec) + running statistics
void finalize_forward(const double* partial, int blocks, int C, double count, const float* gamma, const float* beta, float* vec,
                      float* running_mean, float* running_var, double momentum, double eps, hipStream_t s, const float* mean_shift = nullptr) {
    ColFinArgs f{};
    f.mean_shift = mean_shift;
    f.partial = partial; f.out1 = vec; f.out2 = vec + C; f.out_a = vec + 2 * C; f.out_b = vec + 3 * C; f.gamma = gamma; f.beta = beta;
    f.running_mean = running_mean; f.running_var = running_var; f.blocks = blocks; f.C = C; f.mode = 0; f.M = count; f.eps = eps; f.momentum = momentum;
    FearSync sy;
    if (sync_of(s, &sy)) {
        // SyncBatchNorm: this rank's float64 sums -> the ranks' all-reduce -> statistics of all ranks' rows
        ColFinArgs r{};
        r.partial = partial; r.blocks = blocks; r.C = C; r.mode = 3; r.dsum = sy.buf;
        hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, r);
        sync_all_reduce(sy, 2L * C, 0, s);
        f.partial = sy.buf; f.blocks = 1; f.M = count * sy.world;
    }
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
}

// column-sum partials of (g, g * xhat) -> d beta, d gamma, BnbIn coefficients
void finalize_backward(const double* partial, int blocks, int C, double count, const float* gamma, const float* vec, float* dgamma, float* dbeta,
                       float* coef, hipStream_t s) {
    ColFinArgs f{};
    f.partial = partial; f.out1 = dbeta; f.out2 = dgamma; f.gamma = gamma; f.mean_in = vec; f.rstd_in = vec + C; f.coef = coef;
    f.blocks = blocks; f.C = C; f.mode = 4; f.M = count;
    FearSync sy;
    if (sync_of(s, &sy)) {
        // SyncBatchNorm: d beta / d gamma from this rank's sums (they are averaged with every other gradient), the input gradient's
        // coefficients from all ranks' — the split torch.nn.SyncBatchNorm makes
        ColFinArgs r = f;
        r.mode = 6; r.dsum = sy.buf;
        hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, r);
        sync_all_reduce(sy, 2L * C, 0, s);
        f.partial = sy.buf; f.blocks = 1; f.out1 = nullptr; f.out2 = nullptr; f.M = count * sy.world;
    }
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, f);
}

// Grid of the row-tiled GEMMs with a statistics epilogue: train_pw_grid's, with the row blocks of the large maps fattened to
// `row_tiles` 128-row tiles each so that a launch leaves at most ~2 048 partial rows for its finalize (a 128 x 128 map of 128
// crops has 16 384 tiles: 25 MB of float64 partials at 96 channels, a 130 us finalize)
dim3 dgrad_grid(long M, int Kred, int Nout, int* nt);
dim3 stat_grid(long M, int K, int N, int* nt, int* row_tiles) {
    dim3 g = dgrad_grid(M, K, N, nt);      // (a projection 672 -> 112 re-reads its normalised-on-load operand once per pass as well)
    // (three tiles per pass instead of six for the narrow reductions of the large maps — twice the occupancy, the small operand
    //  read once more — measured no better: 395 vs 343 us for 16 -> 96 at 128 x 128)
    int rt = (int)((g.x + 2047) / 2048);
    if (rt < 1) rt = 1;
    *row_tiles = rt;
    g.x = (g.x + rt - 1) / rt;
    return g;
}

// input-gradient GEMM (reduction over Kred, Nout output columns): when the reduction side is the wide one (an expansion's
// gradient coming back to cin channels) the operand — two tensors with the BatchNorm backward formed on load — is what costs,
// so all output tiles go in ONE pass (NT = n_tiles) instead of train_pw_grid's passes of one tile dealt over gridDim.y, which
// re-read it once per pass (672 -> 112 at 16 x 16: 228 us for a 31 us GEMM)
dim3 dgrad_grid(long M, int Kred, int Nout, int* nt) {
    const int n_tiles = (Nout + 15) / 16;
    if (Kred >= 2 * Nout && n_tiles <= 8 && train_pick_nt(n_tiles) == n_tiles) {
        *nt = n_tiles;
        return dim3((unsigned)((M + 127) / 128), 1);
    }
    return train_pw_grid(M, n_tiles, nt);
}

template <int KS, int S>
void launch_dw_fwd_ks(const DwFwdArgs& a, int sq, dim3 grid, hipStream_t s) {
    if (sq == 8) hipLaunchKernelGGL((dw_fwd_kernel<KS, S, 8>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dw_fwd_kernel<KS, S, 4>), grid, dim3(256), 0, s, a);
}

// Y = act(X) W^T + sums -> vec:  the forward producer of a pointwise unit
void pw_forward_unit(const float* x, int ldx, const float* in_vec, int in_relu, const float* w, float* y, long M, int K, int N, const float* gamma,
                     const float* beta, float* vec, float* rm, float* rv, double momentum, double eps, double* col, hipStream_t s,
                     const float* mean_shift = nullptr) {
    if (gemm_lds_applies(M, K, N)) {      // few row blocks: the LDS-staged, pipelined GEMM (fear_train_gemm.h), same epilogue
        GemmArgs g{};
        g.X = x; g.ldx = ldx; g.W = w; g.Y = y; g.ldy = N; g.M = (int)M; g.K = K; g.N = N; g.partial = col;
        if (in_vec) { g.in.a = in_vec + 2 * K; g.in.b = in_vec + 3 * K; g.in.relu = in_relu; }
        int blocks = 0;
        launch_gemm_lds<1, 1, false>(g, s, &blocks);
        finalize_forward(col, blocks, N, (double)M, gamma, beta, vec, rm, rv, momentum, eps, s, mean_shift);
        return;
    }
    PwStatArgs a{};
    a.X = x; a.ldx = ldx; a.W = w; a.Y = y; a.ldy = N; a.M = (int)M; a.K = K; a.N = N;
    if (in_vec) { a.in.a = in_vec + 2 * K; a.in.b = in_vec + 3 * K; a.in.relu = in_relu; }
    a.partial = col;
    int nt = 1;
    const dim3 grid = stat_grid(M, K, N, &nt, &a.row_tiles);
    launch_pw_stat(a, grid, nt, s);
    finalize_forward(col, (int)grid.x, N, (double)M, gamma, beta, vec, rm, rv, momentum, eps, s, mean_shift);
}

// sums of (g, g * xhat) over rows of (dy, x_raw) [mask: a ReLU behind the BatchNorm] -> d beta, d gamma, coef
void bn_backward_sums(const float* dy, int lddy, const float* raw, int ldx, const float* vec, int relu, const float* gamma, float* dgamma,
                      float* dbeta, float* coef, long M, int C, double* col, hipStream_t s) {
    ColArgs a{};
    a.A = dy; a.lda = lddy; a.X = raw; a.ldx = ldx; a.mean = vec; a.rstd = vec + C;
    a.act_a = relu ? vec + 2 * C : nullptr; a.act_b = relu ? vec + 3 * C : nullptr;
    a.partial = col; a.M = M; a.C = C; a.rpb = col_rows_per_block(M);
    const int blocks = col_blocks(M);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
    finalize_backward(col, blocks, C, (double)M, gamma, vec, dgamma, dbeta, coef, s);
}

// The E-free form of an expansion's backward (BnbIn, fear_train.hip): from BN1's coefficients [A | s1 | mu | Q] and the expansion
// weights W1 [cexp][cin], the extended K-major weight matrix of the input-gradient GEMM
//     Wext [cexp + cin][cin] = [ W1 ; -T ],   T = W1^T diag(A Q) W1
// so that  dx = [A (g1 - s1 + mu Q) | x] Wext  (the GEMM's operand is g1 with BnbIn::E = nullptr, then the block input as loaded).
// (one workgroup per row of T, the reduction over cexp dealt to 1024 / kp lanes per element, kp = cin rounded up to a power of two —
// the kernel sits between BN1's coefficients and the input-gradient GEMM on the chain of input gradients; further workgroups copy W1)
__global__ __launch_bounds__(1024) void irb_lin_weights_kernel(const float* coef, const float* W1, float* Wext, int cexp, int cin, int kp_log2) {
    __shared__ double part[1024];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= cin) {
        const int idx = ((int)blockIdx.x - cin) * 1024 + tid;
        if (idx < cexp * cin) Wext[idx] = W1[idx];
        return;
    }
    const int kp = 1 << kp_log2, J = 1024 >> kp_log2;
    const int k1 = blockIdx.x, k = tid & (kp - 1), j = tid >> kp_log2;
    double t = 0.0;
    if (k < cin)
        for (int c = j; c < cexp; c += J)
            t += (double)W1[(long)c * cin + k1] * ((double)coef[c] * (double)coef[3 * cexp + c]) * (double)W1[(long)c * cin + k];
    part[tid] = t;
    __syncthreads();
    if (j != 0 || k >= cin) return;
    for (int l = 1; l < J; ++l) t += part[l * kp + k];      // fixed order
    Wext[(long)(cexp + k1) * cin + k] = (float)-t;
}

// ... and of its weight gradient: dW1 [cexp][cin] (holding [A (g1 - s1 + mu Q)]^T x) -= diag(A Q) W1 G, G = x^T x [cin][cin]
__global__ __launch_bounds__(256) void irb_lin_wgrad_fix_kernel(const float* coef, const float* W1, const float* G, int ldg, float* dW1, int cexp, int cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cexp * cin) return;
    const int c = idx / cin, k = idx - c * cin;
    double t = 0.0;
    for (int k1 = 0; k1 < cin; ++k1) t += (double)W1[(long)c * cin + k1] * (double)G[(long)k1 * ldg + k];
    dW1[idx] = (float)((double)dW1[idx] - (double)coef[c] * (double)coef[3 * cexp + c] * t);
}

// ... and where dw_bwd_kernel<.., W1G> has already summed R[c][k] = sum_m g1[m][c] x[m][k] (dW1 holds it): the whole of BatchNorm1's algebra
//     dW1 = A (R - (s1 - mu Q) Sx) - (A Q) W1 G,   Sx = column sums of x (gram_kernel's tail: G | Sx with row pitch ldg)
__global__ __launch_bounds__(256) void irb_lin_wgrad_fix2_kernel(const float* coef, const float* W1, const float* G, int ldg, float* dW1, int cexp, int cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cexp * cin) return;
    const int c = idx / cin, k = idx - c * cin;
    double t = 0.0;
    for (int k1 = 0; k1 < cin; ++k1) t += (double)W1[(long)c * cin + k1] * (double)G[(long)k1 * ldg + k];
    const double A = coef[c], s1 = coef[cexp + c], mu = coef[2 * cexp + c], Q = coef[3 * cexp + c];
    dW1[idx] = (float)(A * ((double)dW1[idx] - (s1 - mu * Q) * (double)G[(long)ldg * ldg + k]) - A * Q * t);
}

// running statistics of a BatchNorm from its saved vec = [mean | rstd | a | b] (the forward ran with running_mean = NULL so that
// two passes of the shared trunk can overlap on two streams; torch's order — template pass first — is restored by applying the
// search pass's update afterwards): biased variance = 1 / rstd^2 - eps, tracked unbiased
__global__ __launch_bounds__(256) void bn_running_update_kernel(const float* vec, float* rm, float* rv, int C, double count, double momentum, double eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mean = (double)vec[c], rs = (double)vec[C + c];
    double var = 1.0 / (rs * rs) - eps;
    if (var < 0.0) var = 0.0;
    rm[c] = (float)((1.0 - momentum) * (double)rm[c] + momentum * mean);
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rv[c] = (float)((1.0 - momentum) * (double)rv[c] + momentum * unbiased);
}

// the same update for up to 64 BatchNorms in one launch (block = one BatchNorm): the deferred updates of a trunk pass were 47 launches
struct BnRunMulti { const float* vec[64]; float* rm[64]; float* rv[64]; int C[64]; double count[64]; };
__global__ __launch_bounds__(256) void bn_running_update_multi_kernel(BnRunMulti t, double momentum, double eps) {
    const int e = blockIdx.x, C = t.C[e];
    const float* vec = t.vec[e];
    const double count = t.count[e];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double mean = (double)vec[c], rs = (double)vec[C + c];
        double var = 1.0 / (rs * rs) - eps;
        if (var < 0.0) var = 0.0;
        t.rm[e][c] = (float)((1.0 - momentum) * (double)t.rm[e][c] + momentum * mean);
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        t.rv[e][c] = (float)((1.0 - momentum) * (double)t.rv[e][c] + momentum * unbiased);
    }
}

template <int KS, int S>
void launch_dw_bwd_ks(const DwBwdArgs& a, int sq, bool bn1, dim3 grid, hipStream_t s) {
    if (sq == 8) {
        if (bn1) hipLaunchKernelGGL((dw_bwd_kernel<KS, S, 8, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dw_bwd_kernel<KS, S, 8, false>), grid, dim3(256), 0, s, a);
    } else {
        if (bn1) hipLaunchKernelGGL((dw_bwd_kernel<KS, S, 4, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dw_bwd_kernel<KS, S, 4, false>), grid, dim3(256), 0, s, a);
    }
}

// Events that order the weight-gradient stream behind the kernels that produce its operands: a ring, created on first use and
// never destroyed (a wait refers to the record that preceded it, so re-recording an event later does not disturb waits already
// enqueued; 512 is far more than a step has in flight)
hipEvent_t ring_event() {
    // one ring per device (an event belongs to the device that was current when it was created), positions handed out atomically:
    // two host threads may drive two networks at once
    static hipEvent_t ring[16][512];
    static std::atomic<unsigned> pos{0};
    static std::mutex create;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    hipEvent_t& e = ring[dev][pos.fetch_add(1) % 512];
    if (!e) {
        std::lock_guard<std::mutex> lock(create);
        if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return e;
}
// `to` waits for everything issued on `from` so far
bool stream_follow(hipStream_t to, hipStream_t from) {
    hipEvent_t e = ring_event();
    return e && hipEventRecord(e, from) == hipSuccess && hipStreamWaitEvent(to, e, 0) == hipSuccess;
}

int irb_cmax(const FearIrbBlock* b) {
    return b->cexp > b->cout ? (b->cexp > b->cin ? b->cexp : b->cin) : (b->cout > b->cin ? b->cout : b->cin);
}

bool irb_shape_ok(const FearIrbBlock* b, int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return false;
    if (b->cin < 4 || b->cin % 4 || b->cexp < 4 || b->cexp % 4 || b->cout < 4 || b->cout % 4 || b->cexp > 1024 || b->cout > 1024) return false;
    if (!(b->k == 3 || b->k == 5) || !(b->stride == 1 || b->stride == 2) || H % b->stride || W % b->stride) return false;
    if (!b->expand && b->cexp != b->cin) return false;
    if (b->residual && (b->stride != 1 || b->cin != b->cout)) return false;
    if ((long)B * H * W * b->cexp * 4 >= (1L << 31)) return false;      // 32-bit buffer offsets in the depthwise kernels
    return true;
}

// floats of dw_bwd_kernel's tap-gradient partials [workgroups per slab][k * k][cexp] (block_ws's bound on the workgroups)
size_t irb_taps_floats(const FearIrbBlock* b) {
    const int sq = dw_bwd_sq(b->cexp), nslab = (b->cexp / 4 + sq - 1) / sq;
    const int wps = FEAR_DW_WGS / nslab > 1 ? FEAR_DW_WGS / nslab : 1;
    return (size_t)wps * b->k * b->k * b->cexp;
}
// (FEAR_IRB_FUSE_W3: on request only — with the virtual expansions' in-kernel weight gradient already on the chain of input gradients,
//  this one as well makes that chain the longest of the three streams: 14.65 ms with either, 14.84 with both, 14.75 with neither)
bool irb_w3g(const FearIrbBlock* b) { return (b->flags & FEAR_IRB_FUSE_W3) && b->cexp <= 32 && b->cout <= 32 && (b->cexp + 15) / 16 == (b->cout + 15) / 16; }
size_t irb_w3g_floats(const FearIrbBlock* b) { return irb_w3g(b) ? (size_t)2048 * b->cout * b->cexp : 0; }
bool irb_w1g(const FearIrbBlock* b) { return (b->flags & FEAR_IRB_VIRTUAL_E) && b->k == 3; }
size_t irb_w1g_floats(const FearIrbBlock* b) { return irb_w1g(b) ? irb_taps_floats(b) / (b->k * b->k) * b->cin : 0; }      // [workgroups per slab][cexp][cin]
size_t irb_lin_floats(const FearIrbBlock* b) {
    return b->expand ? (size_t)(b->cexp + b->cin) * b->cin + (b->cin * b->cin > 1056 ? b->cin * b->cin : 1056) : 0;
}

// the shapes the virtual expansion is built for: the stride-2 blocks with 16 ... 32 input channels and 32-channel slabs (FEAR-XS:
// 16 -> 96 at 128 x 128, 24 -> 144 at 64 x 64, 32 -> 192 at 32 x 32); the stride-1 depthwise backward has no LDS left for the tile of e
bool irb_virtual_shape(const FearIrbBlock* b) {
    return b->expand && b->cin >= 16 && b->cin <= 32 && b->stride == 2 && b->cexp >= 64 && b->cexp % 16 == 0 && !(b->flags & FEAR_IRB_NO_LINEAR_BN1);
}
bool irb_virtual(const FearIrbBlock* b) { return (b->flags & FEAR_IRB_VIRTUAL_E) != 0; }

}  // namespace

extern "C" {

int fear_train_sync_bind(void* stream, const FearSync* sync) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sync && (!sync->all_reduce || !sync->buf)) return FEAR_TRAIN_ERR_NULL;
    if (sync && (sync->world < 1 || sync->buf_bytes < FEAR_SYNC_BUF_BYTES)) return FEAR_TRAIN_ERR_SHAPE;
    std::lock_guard<std::mutex> lock(g_sync_mutex);
    SyncSlot* slot = nullptr;
    for (SyncSlot& e : g_sync_slots)
        if (e.used && e.s == s) slot = &e;
    if (!sync) {
        if (slot) slot->used = false;
        return FEAR_TRAIN_OK;
    }
    if (!slot)
        for (SyncSlot& e : g_sync_slots)
            if (!e.used) { slot = &e; break; }
    if (!slot) return FEAR_TRAIN_ERR_SHAPE;      // more than 16 streams bound
    slot->s = s; slot->sy = *sync; slot->used = true;
    return FEAR_TRAIN_OK;
}

size_t fear_irb_workspace_bytes(const FearIrbBlock* b, int B, int H, int W) {
    if (!b || !irb_shape_ok(b, B, H, W)) return 0;
    const long rows_in = (long)B * H * W, rows_out = rows_in / (b->stride * b->stride);
    return block_ws(rows_in, rows_out, b->cin, b->cexp, b->cout, b->k, nullptr).total;
}

int fear_irb_virtual_ok(const FearIrbBlock* b) { return b && irb_virtual_shape(b) ? 1 : 0; }

size_t fear_irb_scratch_floats(const FearIrbBlock* b, int B, int H, int W) {
    if (!b || !irb_shape_ok(b, B, H, W)) return 0;
    const long rows_in = (long)B * H * W, rows_out = rows_in / (b->stride * b->stride);
    // g2 | g1 | the three BatchNorms' backward coefficient vectors (3 x [4][cmax]; they must outlive the call when the weight
    // gradients run on their own stream, so they do not live in the shared workspace)
    // | the extended weight matrix and the input's Gram matrix of the expansion's E-free backward (BnbIn)
    // | the depthwise tap gradients' per-workgroup partials (their final sum runs on the weight-gradient stream as well)
    // | (virtual 3 x 3 expansions) the per-workgroup partials of the expansion's weight gradient formed inside the depthwise backward
    // | (blocks of at most 32 channels throughout) those of the projection's weight gradient formed inside the masked-gradient pass
    return (size_t)rows_out * b->cexp + (b->expand ? (size_t)rows_in * b->cexp : 0) + (size_t)12 * irb_cmax(b) + irb_lin_floats(b) +
           irb_taps_floats(b) + irb_w1g_floats(b) + irb_w3g_floats(b);
}

int fear_irb_train_forward(const FearIrbBlock* b, const FearIrbSaved* sv, const float* x, float* out, int B, int H, int W, double momentum,
                           double eps, float* workspace, size_t ws_bytes, void* stream) {
    if (!b || !sv || !x || !out || !workspace || !sv->d || !sv->p || !sv->vec[1] || !sv->vec[2] || !b->w_dw || !b->w_pwl) return FEAR_TRAIN_ERR_NULL;
    if (b->expand && ((!sv->e && !irb_virtual(b)) || !sv->vec[0] || !b->w_pw)) return FEAR_TRAIN_ERR_NULL;
    for (int i = b->expand ? 0 : 1; i < 3; ++i)
        if (!b->gamma[i] || !b->beta[i]) return FEAR_TRAIN_ERR_NULL;
    if (!irb_shape_ok(b, B, H, W) || (irb_virtual(b) && !irb_virtual_shape(b))) return FEAR_TRAIN_ERR_SHAPE;
    const long rows_in = (long)B * H * W, rows_out = rows_in / (b->stride * b->stride);
    const BlockWs ws = block_ws(rows_in, rows_out, b->cin, b->cexp, b->cout, b->k, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ho = H / b->stride, Wo = W / b->stride;
    const bool virt = irb_virtual(b);
    if (virt) {
        // BatchNorm1's statistics of the expansion that is never written: G = x^T x and the column sums of x in one pass, then W1 on them
        long rpw = (rows_in + 511) / 512;      // (512 workgroups: one round)
        rpw = (rpw + 127) / 128 * 128;
        const int wgs = (int)((rows_in + rpw - 1) / rpw);
        const int nc = b->cin > 16 ? 2 : 1, kp = 16 * nc, per = kp * kp + kp;
        if ((size_t)wgs * per * sizeof(float) > ws.wg_bytes || ws.coef_bytes < per * sizeof(float)) return FEAR_TRAIN_ERR_WORKSPACE;
        if (nc == 1) hipLaunchKernelGGL(gram_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, s, x, rows_in, b->cin, rpw, ws.wg);
        else hipLaunchKernelGGL(gram_kernel<2>, dim3((unsigned)wgs), dim3(256), 0, s, x, rows_in, b->cin, rpw, ws.wg);
        launch_slice_sum(ws.wg, ws.coef, per, wgs, s);
        double count1 = (double)rows_in;
        FearSync sy;
        if (sync_of(s, &sy)) {
            // SyncBatchNorm: the expansion's statistics are linear in (G, column sums) — those are what the ranks add up
            if (hipMemcpyAsync(sy.buf, ws.coef, per * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return FEAR_TRAIN_ERR_HIP;
            sync_all_reduce(sy, per, 1, s);
            if (hipMemcpyAsync(ws.coef, sy.buf, per * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return FEAR_TRAIN_ERR_HIP;
            count1 *= sy.world;
        }
        if (nc == 1)
            hipLaunchKernelGGL(irb_virtual_stats_kernel<16>, dim3((unsigned)((b->cexp + 63) / 64)), dim3(64), 0, s, ws.coef, b->w_pw, b->gamma[0],
                               b->beta[0], sv->vec[0], b->running_mean[0], b->running_var[0], b->cexp, b->cin, count1, eps, momentum);
        else
            hipLaunchKernelGGL(irb_virtual_stats_kernel<32>, dim3((unsigned)((b->cexp + 63) / 64)), dim3(64), 0, s, ws.coef, b->w_pw, b->gamma[0],
                               b->beta[0], sv->vec[0], b->running_mean[0], b->running_var[0], b->cexp, b->cin, count1, eps, momentum);
    }
    // expand 1x1 (+ statistics)
    if (b->expand && !virt)
        pw_forward_unit(x, b->cin, nullptr, 0, b->w_pw, sv->e, rows_in, b->cin, b->cexp, b->gamma[0], b->beta[0], sv->vec[0], b->running_mean[0],
                        b->running_var[0], momentum, eps, ws.col, s);
    // depthwise over act1(e) (or over the block input) (+ statistics)
    {
        DwFwdArgs a{};
        a.X = b->expand ? sv->e : x; a.ldx = b->cexp; a.Wt = b->w_dw; a.Y = sv->d; a.ldy = b->cexp;
        a.B = B; a.H = H; a.W = W; a.C = b->cexp; a.Ho = Ho; a.Wo = Wo;
        if (b->expand) { a.in.a = sv->vec[0] + 2 * b->cexp; a.in.b = sv->vec[0] + 3 * b->cexp; a.in.relu = 1; }
        const int sq = dw_bwd_sq(b->cexp);
        const bool small_map = b->k == 5 && b->stride == 1 && sq == 8 && Ho <= 8 && Wo <= 8;      // 8 x 8 tiles (the template branch's last stage)
        const int to = b->stride == 1 ? (small_map ? 8 : 16) : 8;
        a.tiles_x = (Wo + to - 1) / to; a.tiles_y = (Ho + to - 1) / to;
        a.nslab = (b->cexp / 4 + sq - 1) / sq;
        a.wgs_per_slab = dw_bwd_wgs_per_slab(B * a.tiles_x * a.tiles_y, a.nslab);
        a.psums = ws.col;
        const dim3 grid((unsigned)(a.wgs_per_slab * a.nslab));
        if (virt) {
            a.X = nullptr; a.ve.X = x; a.ve.W1 = b->w_pw; a.ve.cin = b->cin;
            if (b->k == 3 && b->cin <= 16) hipLaunchKernelGGL((dw_fwd_kernel<3, 2, 8, 1>), grid, dim3(256), 0, s, a);
            else if (b->k == 3) hipLaunchKernelGGL((dw_fwd_kernel<3, 2, 8, 2>), grid, dim3(256), 0, s, a);
            else if (b->cin <= 16) hipLaunchKernelGGL((dw_fwd_kernel<5, 2, 8, 1>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((dw_fwd_kernel<5, 2, 8, 2>), grid, dim3(256), 0, s, a);
        } else if (small_map) hipLaunchKernelGGL((dw_fwd_kernel<5, 1, 8, 0, 8>), grid, dim3(256), 0, s, a);
        else if (b->k == 3 && b->stride == 1) launch_dw_fwd_ks<3, 1>(a, sq, grid, s);
        else if (b->k == 3) launch_dw_fwd_ks<3, 2>(a, sq, grid, s);
        else if (b->stride == 1) launch_dw_fwd_ks<5, 1>(a, sq, grid, s);
        else launch_dw_fwd_ks<5, 2>(a, sq, grid, s);
        finalize_forward(ws.col, a.wgs_per_slab, b->cexp, (double)rows_out, b->gamma[1], b->beta[1], sv->vec[1], b->running_mean[1],
                         b->running_var[1], momentum, eps, s);
    }
    // project 1x1 over act2(d) (+ statistics)
    pw_forward_unit(sv->d, b->cexp, sv->vec[1], 1, b->w_pwl, sv->p, rows_out, b->cexp, b->cout, b->gamma[2], b->beta[2], sv->vec[2],
                    b->running_mean[2], b->running_var[2], momentum, eps, ws.col, s);
    // block output = BN3(p) [+ x]
    {
        BnActArgs k{};
        k.X = sv->p; k.R = b->residual ? x : nullptr; k.Y = out; k.in.a = sv->vec[2] + 2 * b->cout; k.in.b = sv->vec[2] + 3 * b->cout; k.in.relu = 0;
        k.M = rows_out; k.C = b->cout; k.ldx = b->cout; k.ldr = b->cin; k.ldy = b->cout;
        const long n4 = rows_out * (b->cout / 4);
        hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, k);
    }
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_irb_train_backward(const FearIrbBlock* b, const FearIrbSaved* sv, const FearIrbGrads* gr, const float* x, const float* dout, float* dx,
                            float* scratch, int B, int H, int W, float* workspace, size_t ws_bytes, void* stream, void* wgrad_stream) {
    if (!b || !sv || !gr || !x || !dout || !scratch || !workspace || !sv->d || !sv->p || !sv->vec[1] || !sv->vec[2] || !gr->w_dw || !gr->w_pwl)
        return FEAR_TRAIN_ERR_NULL;
    if (b->expand && ((!sv->e && !irb_virtual(b)) || !sv->vec[0] || !gr->w_pw)) return FEAR_TRAIN_ERR_NULL;
    if (!b->expand && !dx) return FEAR_TRAIN_ERR_NULL;
    for (int i = b->expand ? 0 : 1; i < 3; ++i)
        if (!gr->gamma[i] || !gr->beta[i] || !b->gamma[i]) return FEAR_TRAIN_ERR_NULL;
    if (!irb_shape_ok(b, B, H, W) || (irb_virtual(b) && !irb_virtual_shape(b))) return FEAR_TRAIN_ERR_SHAPE;
    const long rows_in = (long)B * H * W, rows_out = rows_in / (b->stride * b->stride);
    const BlockWs ws = block_ws(rows_in, rows_out, b->cin, b->cexp, b->cout, b->k, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ho = H / b->stride, Wo = W / b->stride, cexp = b->cexp, cout = b->cout, cin = b->cin;
    const int cmax = irb_cmax(b);
    float* g2 = scratch;
    float* g1 = scratch + (size_t)rows_out * cexp;
    float* coef1 = g1 + (b->expand ? (size_t)rows_in * cexp : 0);
    float* coef2 = coef1 + 4 * cmax;
    float* coef3 = coef1 + 8 * cmax;
    // the two pointwise weight gradients are off the chain that leads to dx: with a `wgrad_stream` they are issued there, behind
    // an event that follows the kernels producing their operands, and overlap the rest of this block's (and the next blocks')
    // backward — every kernel of a 16 x 16 map is a few hundred workgroups and leaves most of the device idle on its own
    hipStream_t sw = wgrad_stream ? static_cast<hipStream_t>(wgrad_stream) : s;
    // BN3 (no ReLU): sums over (dout, p)
    bn_backward_sums(dout, cout, sv->p, cout, sv->vec[2], 0, b->gamma[2], gr->gamma[2], gr->beta[2], coef3, rows_out, cout, ws.col, s);
    BnbIn bn3{};
    bn3.E = sv->p; bn3.coef = coef3; bn3.lde = cout; bn3.C = cout;
    bool w3g = false;
    int w3g_slices = 0;
    float* w3g_part = coef1 + 12 * cmax + irb_lin_floats(b) + irb_taps_floats(b) + irb_w1g_floats(b);      // [<= 2048][cout][cexp]
    // g2 = (dp W3) masked by act2(d) > 0, + sums of (g2, dhat)
    if (gemm_lds_applies(rows_out, cout, cexp)) {
        GemmArgs g{};
        g.X = dout; g.ldx = cout; g.bn = bn3; g.W = b->w_pwl; g.Y = g2; g.ldy = cexp; g.D = sv->d; g.ldd = cexp; g.dvec = sv->vec[1];
        g.partial = ws.col; g.M = (int)rows_out; g.K = cout; g.N = cexp;
        int blocks = 0;
        // (checked before the launch that writes them: at most one partial row per 64 output rows)
        if ((size_t)((rows_out + 63) / 64) * 2 * cexp * sizeof(double) > ws.col_bytes) return FEAR_TRAIN_ERR_WORKSPACE;
        launch_gemm_lds<2, 2, true>(g, s, &blocks);
        finalize_backward(ws.col, blocks, cexp, (double)rows_out, b->gamma[1], sv->vec[1], gr->gamma[1], gr->beta[1], coef2, s);
    } else {
        PwBwdArgs a{};
        a.G = dout; a.ldg = cout; a.bn = bn3; a.W = b->w_pwl; a.Y = g2; a.ldy = cexp; a.D = sv->d; a.ldd = cexp; a.dvec = sv->vec[1];
        a.partial = ws.col; a.M = (int)rows_out; a.Kred = cout; a.Nout = cexp;
        int nt = 1;
        const dim3 grid = stat_grid(rows_out, cout, cexp, &nt, &a.row_tiles);
        if ((size_t)grid.x * 2 * cexp * sizeof(double) > ws.col_bytes) return FEAR_TRAIN_ERR_WORKSPACE;
        // the narrow blocks (16 / 24 channels throughout, the 128 x 128 / 64 x 64 maps): the projection's weight gradient in the same pass
        w3g = irb_w3g(b) && grid.y == 1 && nt == (cexp + 15) / 16 && grid.x <= 2048;
        if (w3g) {
            a.p3 = w3g_part; w3g_slices = (int)grid.x;
            if (nt == 1) hipLaunchKernelGGL((pw_bwd_kernel<1, true, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((pw_bwd_kernel<2, true, true>), grid, dim3(256), 0, s, a);
        } else {
            launch_pw_bwd<true>(a, grid, nt, s);
        }
        finalize_backward(ws.col, (int)grid.x, cexp, (double)rows_out, b->gamma[1], sv->vec[1], gr->gamma[1], gr->beta[1], coef2, s);
    }
    // dW3 = dp^T act2(d)
    {
        if (sw != s && !stream_follow(sw, s)) return FEAR_TRAIN_ERR_HIP;      // coef3 exists
        if (w3g) launch_slice_sum(w3g_part, gr->w_pwl, (long)cout * cexp, w3g_slices, sw);
        const int rc = w3g ? FEAR_TRAIN_OK : wgrad_impl(dout, cout, 0, sv->d, cexp, 0, gr->w_pwl, ws.wg, ws.wg_bytes, rows_out, cexp, cout, 1, sw, sv->vec[1] + 2 * cexp,
                                  sv->vec[1] + 3 * cexp, 1, &bn3);
        if (rc != FEAR_TRAIN_OK) return rc;
    }
    // depthwise + both BatchNorms around it
    {
        DwBwdArgs a{};
        a.G2 = g2; a.D = sv->d; a.ldo = cexp; a.coef2 = coef2; a.Wt = b->w_dw;
        a.E = b->expand ? sv->e : x; a.lde = cexp; a.act1 = b->expand ? sv->vec[0] : nullptr;
        a.R = (!b->expand && b->residual) ? dout : nullptr; a.ldr = cout;
        a.Y = b->expand ? g1 : dx; a.ldy = cexp;
        float* taps = coef1 + 12 * cmax + irb_lin_floats(b);
        a.ptaps = taps; a.psums = ws.col;
        a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.C = cexp;
        const int sq = dw_bwd_sq(cexp);
        const bool small_map = b->k == 5 && b->stride == 1 && sq == 8 && b->expand && H <= 8 && W <= 8;      // 8 x 8 tiles
        const int ts = small_map ? 8 : 16;
        a.tiles_x = (W + ts - 1) / ts; a.tiles_y = (H + ts - 1) / ts;
        a.nslab = (cexp / 4 + sq - 1) / sq;
        a.wgs_per_slab = dw_bwd_wgs_per_slab(B * a.tiles_x * a.tiles_y, a.nslab);
        const dim3 grid((unsigned)(a.wgs_per_slab * a.nslab));
        if (irb_virtual(b)) {
            a.E = nullptr; a.ve.X = x; a.ve.W1 = b->w_pw; a.ve.cin = cin;
            if (irb_w1g(b)) a.pw1 = taps + irb_taps_floats(b);
            if (b->k == 3 && cin <= 16 && a.pw1) hipLaunchKernelGGL((dw_bwd_kernel<3, 2, 8, true, 1, 16, true>), grid, dim3(256), 0, s, a);
            else if (b->k == 3 && a.pw1) hipLaunchKernelGGL((dw_bwd_kernel<3, 2, 8, true, 2, 16, true>), grid, dim3(256), 0, s, a);
            else if (b->k == 3 && cin <= 16) hipLaunchKernelGGL((dw_bwd_kernel<3, 2, 8, true, 1>), grid, dim3(256), 0, s, a);
            else if (b->k == 3) hipLaunchKernelGGL((dw_bwd_kernel<3, 2, 8, true, 2>), grid, dim3(256), 0, s, a);
            else if (cin <= 16) hipLaunchKernelGGL((dw_bwd_kernel<5, 2, 8, true, 1>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((dw_bwd_kernel<5, 2, 8, true, 2>), grid, dim3(256), 0, s, a);
        } else if (small_map) hipLaunchKernelGGL((dw_bwd_kernel<5, 1, 8, true, 0, 8>), grid, dim3(256), 0, s, a);
        else if (b->k == 3 && b->stride == 1) launch_dw_bwd_ks<3, 1>(a, sq, b->expand != 0, grid, s);
        else if (b->k == 3) launch_dw_bwd_ks<3, 2>(a, sq, b->expand != 0, grid, s);
        else if (b->stride == 1) launch_dw_bwd_ks<5, 1>(a, sq, b->expand != 0, grid, s);
        else launch_dw_bwd_ks<5, 2>(a, sq, b->expand != 0, grid, s);
        if (b->expand)
            finalize_backward(ws.col, a.wgs_per_slab, cexp, (double)rows_in, b->gamma[0], sv->vec[0], gr->gamma[0], gr->beta[0], coef1, s);
        // the tap gradients' final sum is a weight gradient too: off the chain (the partials live in the call's private scratch)
        if (sw != s && !stream_follow(sw, s)) return FEAR_TRAIN_ERR_HIP;          // the partials, g1 and coef1 exist
        launch_slice_sum(taps, gr->w_dw, (long)b->k * b->k * cexp, a.wgs_per_slab, sw);
        if (a.pw1) launch_slice_sum(a.pw1, gr->w_pw, (long)cexp * cin, a.wgs_per_slab, sw);      // R[c][k], completed below
    }
    if (b->expand) {
        // BN1's backward without its input: e = x W1^T is linear in the block input, so both consumers read g1 (cexp channels) and x
        // (cin channels) instead of g1 and e — see BnbIn.  G = x^T x and the two small correction kernels are the price.
        // where it pays — 16-32 input channels, the expansions of the 128 x 128 ... 32 x 32 maps: bandwidth-bound launches that read
        // 0.1-0.8 GB less each (16.71 -> 16.49 ms per step).  With 64 / 112 input channels (the 16 x 16 maps) the T kernel on the
        // chain, the Gram matrix and the correction cost more than the lighter loads save: 16.64 / 16.80 ms with those included.
        // (cexp % 16: a GEMM stage of 16 reduction columns must not straddle g1 and x.)
        const bool lin = cexp % 16 == 0 && cin <= 128 && !(b->flags & FEAR_IRB_NO_LINEAR_BN1) &&
                         (cin <= 32 || (b->flags & FEAR_IRB_LINEAR_BN1));      // (always true for a virtual expansion: irb_virtual_shape)
        BnbIn bn1{};
        bn1.coef = coef1; bn1.C = cexp;
        if (!lin) { bn1.E = sv->e; bn1.lde = cexp; }
        float* wext = coef1 + 12 * cmax;                       // [cexp + cin][cin]
        float* gram = wext + (size_t)(cexp + cin) * cin;       // [cin][cin]
        int rc = FEAR_TRAIN_OK, ldg = cin;
        if (lin && cin <= 32) {
            // (the one-load-per-four-rows Gram kernel of the virtual expansion's forward; its result is [KP][KP] | column sums)
            long rpw = (rows_in + 511) / 512;      // (512 workgroups: one round)
            rpw = (rpw + 127) / 128 * 128;
            const int wgs = (int)((rows_in + rpw - 1) / rpw);
            const int nc = cin > 16 ? 2 : 1, per = 16 * nc * 16 * nc + 16 * nc;
            ldg = 16 * nc;
            if ((size_t)wgs * per * sizeof(float) > ws.wg_bytes) return FEAR_TRAIN_ERR_WORKSPACE;
            if (nc == 1) hipLaunchKernelGGL(gram_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, sw, x, rows_in, cin, rpw, ws.wg);
            else hipLaunchKernelGGL(gram_kernel<2>, dim3((unsigned)wgs), dim3(256), 0, sw, x, rows_in, cin, rpw, ws.wg);
            launch_slice_sum(ws.wg, gram, per, wgs, sw);
        } else if (lin) {
            rc = wgrad_impl(x, cin, 0, x, cin, 0, gram, ws.wg, ws.wg_bytes, rows_in, cin, cin, 1, sw);
        }
        if (rc != FEAR_TRAIN_OK) return rc;
        const bool w1g = irb_w1g(b);      // (then lin and cin <= 32: the Gram kernel's [KP][KP] | column sums above)
        if (!w1g) rc = wgrad_impl(g1, cexp, 0, x, cin, 0, gr->w_pw, ws.wg, ws.wg_bytes, rows_in, cin, cexp, 1, sw, nullptr, nullptr, 0, &bn1);
        if (rc != FEAR_TRAIN_OK) return rc;
        if (w1g)
            hipLaunchKernelGGL(irb_lin_wgrad_fix2_kernel, dim3((unsigned)((cexp * cin + 255) / 256)), dim3(256), 0, sw, coef1, b->w_pw, gram, ldg,
                               gr->w_pw, cexp, cin);
        else if (lin)
            hipLaunchKernelGGL(irb_lin_wgrad_fix_kernel, dim3((unsigned)((cexp * cin + 255) / 256)), dim3(256), 0, sw, coef1, b->w_pw, gram, ldg,
                               gr->w_pw, cexp, cin);
        if (dx) {
            if (lin)
            {
                int kp_log2 = 4;
                while ((1 << kp_log2) < cin) ++kp_log2;
                hipLaunchKernelGGL(irb_lin_weights_kernel, dim3((unsigned)(cin + (cexp * cin + 1023) / 1024)), dim3(1024), 0, s, coef1, b->w_pw,
                                   wext, cexp, cin, kp_log2);
            }
            const int kred = lin ? cexp + cin : cexp;
            if (gemm_lds_applies(rows_in, kred, cin)) {
                GemmArgs g{};
                g.X = g1; g.ldx = cexp; g.bn = bn1; g.W = lin ? wext : b->w_pw; g.R = b->residual ? dout : nullptr; g.ldr = cout;
                g.Y = dx; g.ldy = cin; g.M = (int)rows_in; g.K = kred; g.N = cin;
                if (lin) {
                    g.X2 = x; g.ldx2 = cin; g.K1 = cexp;
                    launch_gemm_lds<3, 0, true>(g, s, nullptr);
                } else {
                    launch_gemm_lds<2, 0, true>(g, s, nullptr);
                }
            } else {
                PwBwdArgs a{};
                a.G = g1; a.ldg = cexp; a.bn = bn1; a.W = lin ? wext : b->w_pw; a.R = b->residual ? dout : nullptr; a.ldr = cout;
                if (lin) { a.X2 = x; a.ldx2 = cin; a.K1 = cexp; }
                a.Y = dx; a.ldy = cin; a.M = (int)rows_in; a.Kred = kred; a.Nout = cin;
                int nt = 1;
                const dim3 grid = dgrad_grid(rows_in, kred, cin, &nt);
                launch_pw_bwd<false>(a, grid, nt, s);
            }
        }
    }
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_running_update(const float* vec, double count, float* running_mean, float* running_var, double momentum, double eps, int C,
                           void* stream) {
    if (!vec || !running_mean || !running_var) return FEAR_TRAIN_ERR_NULL;
    if (C < 1 || !(count >= 1.0)) return FEAR_TRAIN_ERR_SHAPE;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), vec, running_mean,
                       running_var, C, count, momentum, eps);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_bn_running_update_multi(const FearBnRunning* items, int n, double momentum, double eps, void* stream) {
    if (!items || n < 0) return FEAR_TRAIN_ERR_NULL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i0 = 0; i0 < n; i0 += 64) {
        BnRunMulti t{};
        const int m = n - i0 < 64 ? n - i0 : 64;
        for (int i = 0; i < m; ++i) {
            const FearBnRunning& it = items[i0 + i];
            if (!it.vec || !it.running_mean || !it.running_var) return FEAR_TRAIN_ERR_NULL;
            if (it.C < 1 || !(it.count >= 1.0)) return FEAR_TRAIN_ERR_SHAPE;
            t.vec[i] = it.vec; t.rm[i] = it.running_mean; t.rv[i] = it.running_var; t.C[i] = it.C; t.count[i] = it.count;
        }
        hipLaunchKernelGGL(bn_running_update_multi_kernel, dim3((unsigned)m), dim3(256), 0, s, t, momentum, eps);
    }
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

size_t fear_pwbn_workspace_bytes(long M, int K, int N) {
    if (M < 1 || K < 4 || N < 4) return 0;
    return block_ws(M, M, K, N, N, 3, nullptr).total;
}

int fear_pwbn_train_forward(const float* x, int ldx, const float* w, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float* raw, float* vec, int relu, float* out, long M, int K, int N, double momentum, double eps,
                            float* workspace, size_t ws_bytes, void* stream) {
    if (!x || !w || !gamma || !beta || !raw || !vec || !out || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || K < 4 || K % 4 || N < 4 || N % 4 || N > 1024 || M > 0x7fffffffL || !ld_ok(ldx, K)) return FEAR_TRAIN_ERR_SHAPE;
    const BlockWs ws = block_ws(M, M, K, N, N, 3, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    pw_forward_unit(x, ldx, nullptr, 0, w, raw, M, K, N, gamma, beta, vec, running_mean, running_var, momentum, eps, ws.col, s);
    BnActArgs k{};
    k.X = raw; k.Y = out; k.in.a = vec + 2 * N; k.in.b = vec + 3 * N; k.in.relu = relu; k.M = M; k.C = N; k.ldx = N; k.ldy = N;
    const long n4 = M * (N / 4);
    hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, k);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_pwbn_train_backward(const float* dy, const float* raw, const float* vec, int relu, const float* x, int ldx, const float* w,
                             const float* gamma, float* dw, float* dgamma, float* dbeta, float* dx, long M, int K, int N, float* workspace,
                             size_t ws_bytes, void* stream, void* wgrad_stream) {
    if (!dy || !raw || !vec || !x || !w || !gamma || !dw || !dgamma || !dbeta || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (M <= 0 || K < 4 || K % 4 || N < 4 || N % 4 || N > 1024 || M > 0x7fffffffL || !ld_ok(ldx, K)) return FEAR_TRAIN_ERR_SHAPE;
    const BlockWs ws = block_ws(M, M, K, N, N, 3, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipStream_t sw = wgrad_stream ? static_cast<hipStream_t>(wgrad_stream) : s;
    // (this unit's coefficient vectors live in the shared workspace: whatever an earlier call left running on the weight-gradient
    //  stream may still read them — it is waited for first)
    if (sw != s && !stream_follow(s, sw)) return FEAR_TRAIN_ERR_HIP;
    bn_backward_sums(dy, N, raw, N, vec, relu, gamma, dgamma, dbeta, ws.coef, M, N, ws.col, s);
    BnbIn bn{};
    bn.E = raw; bn.coef = ws.coef; bn.lde = N; bn.C = N;
    if (relu) { bn.mask_a = vec + 2 * N; bn.mask_b = vec + 3 * N; }
    if (dx && gemm_lds_applies(M, N, K)) {
        GemmArgs g{};
        g.X = dy; g.ldx = N; g.bn = bn; g.W = w; g.Y = dx; g.ldy = K; g.M = (int)M; g.K = N; g.N = K;
        launch_gemm_lds<2, 0, true>(g, s, nullptr);
    } else if (dx) {
        PwBwdArgs a{};
        a.G = dy; a.ldg = N; a.bn = bn; a.W = w; a.Y = dx; a.ldy = K; a.M = (int)M; a.Kred = N; a.Nout = K;
        int nt = 1;
        const dim3 grid = dgrad_grid(M, N, K, &nt);
        launch_pw_bwd<false>(a, grid, nt, s);
    }
    if (sw != s && !stream_follow(sw, s)) return FEAR_TRAIN_ERR_HIP;
    const int rc = wgrad_impl(dy, N, 0, x, ldx, 0, dw, ws.wg, ws.wg_bytes, M, K, N, 1, sw, nullptr, nullptr, 0, &bn);
    if (rc != FEAR_TRAIN_OK) return rc;
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

// ------------------------------------------------------------------------------------------------
// SepConv (depthwise 3 x 3 + pointwise, both with bias) + BatchNorm + ReLU — the layer the head's towers are made of
// (model_training/model/blocks.py:97-101, 115-119, 151-161) — one call per direction, in the style of the trunk's blocks:
//   forward   d = dw(x) + b_dw;  raw = d W^T (statistics in the GEMM's epilogue; the pointwise bias is left out: it cancels in the
//             normalisation and only shifts the tracked mean);  out = relu(a raw + b)
//   backward  sums over (dy masked, raw) -> coefficients;  dd = bnb(dy) W (the BatchNorm backward formed on load);  dx = DW^T dd;
//             d W = bnb(dy)^T d and the tap gradients on `wgrad_stream`, off the chain of input gradients
// workspace regions: col (stream) | wg, taps (wgrad stream only)
struct SepWs {
    double* col;
    float* wg;
    float* taps;
    size_t col_bytes, wg_bytes, taps_bytes, total;
};

static SepWs sep_ws(long M, int cin, int cout, float* base) {
    SepWs w{};
    const int cmax = cin > cout ? cin : cout;
    size_t col = fear_train_stats_workspace_bytes(M, cmax);
    const size_t colr = (size_t)col_blocks(M) * 2 * cmax * sizeof(double);
    if (colr > col) col = colr;
    const size_t lds = M <= FEAR_GEMM_LDS_MAX_ROWS ? (size_t)((M + 63) / 64) * 2 * cmax * sizeof(double) : 0;
    if (lds > col) col = lds;
    w.col_bytes = align256(col);
    const size_t nk = (size_t)cin * cout;
    size_t wg = (size_t)wgrad_slices(M) * nk * sizeof(float);
    size_t more = (size_t)1024 * nk * sizeof(float);
    if (more > ((size_t)32 << 20)) more = (size_t)32 << 20;
    if (more > wg) wg = more;
    w.wg_bytes = align256(wg);
    w.taps_bytes = align256((size_t)col_blocks(M) * 9 * cin * sizeof(float));
    w.total = w.col_bytes + w.wg_bytes + w.taps_bytes;
    w.col = nullptr; w.wg = nullptr; w.taps = nullptr;
    if (base) {                    // (nullptr: a size query — see block_ws)
        char* p = reinterpret_cast<char*>(base);
        w.col = reinterpret_cast<double*>(p); p += w.col_bytes;
        w.wg = reinterpret_cast<float*>(p); p += w.wg_bytes;
        w.taps = reinterpret_cast<float*>(p);
    }
    return w;
}

static bool sep_shape_ok(const FearSepLayer* L, int B, int H, int W) {
    return L && B >= 1 && H >= 1 && W >= 1 && L->cin >= 4 && L->cin % 4 == 0 && L->cin <= 1024 && L->cout >= 4 && L->cout % 4 == 0 &&
           L->cout <= 1024 && (long)B * H * W * (L->cin > L->cout ? L->cin : L->cout) * 4 < 0x7fffffffL;
}

size_t fear_sepbn_workspace_bytes(const FearSepLayer* L, int B, int H, int W) {
    if (!sep_shape_ok(L, B, H, W)) return 0;
    return sep_ws((long)B * H * W, L->cin, L->cout, nullptr).total;
}

int fear_sepbn_train_forward(const FearSepLayer* L, const float* x, int ldx, float* d, float* raw, float* vec, float* out, int ldo, int B,
                             int H, int W, double momentum, double eps, float* workspace, size_t ws_bytes, void* stream) {
    if (!L || !x || !d || !raw || !vec || !out || !workspace || !L->w_dw || !L->w_pw || !L->gamma || !L->beta) return FEAR_TRAIN_ERR_NULL;
    if (!sep_shape_ok(L, B, H, W) || !ld_ok(ldx, L->cin) || !ld_ok(ldo, L->cout)) return FEAR_TRAIN_ERR_SHAPE;
    const long M = (long)B * H * W;
    const int K = L->cin, N = L->cout;
    const SepWs ws = sep_ws(M, K, N, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = dw_impl(x, ldx, L->w_dw, L->b_dw, d, K, B, H, W, K, 3, 1, s);
    if (rc != FEAR_TRAIN_OK) return rc;
    pw_forward_unit(d, K, nullptr, 0, L->w_pw, raw, M, K, N, L->gamma, L->beta, vec, L->running_mean, L->running_var, momentum, eps, ws.col, s,
                    L->b_pw);
    BnActArgs k{};
    k.X = raw; k.Y = out; k.in.a = vec + 2 * N; k.in.b = vec + 3 * N; k.in.relu = 1; k.M = M; k.C = N; k.ldx = N; k.ldy = ldo;
    const long n4 = M * (N / 4);
    hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, k);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_sepbn_train_backward(const FearSepLayer* L, const FearSepGrads* gr, const float* x, int ldx, const float* d, const float* raw,
                              const float* vec, const float* dy, float* dd, float* coef, float* dx, int B, int H, int W, float* workspace,
                              size_t ws_bytes, void* stream, void* wgrad_stream) {
    if (!L || !gr || !x || !d || !raw || !vec || !dy || !dd || !coef || !dx || !workspace || !L->w_dw || !L->w_pw || !L->gamma)
        return FEAR_TRAIN_ERR_NULL;
    if (!gr->w_dw || !gr->w_pw || !gr->gamma || !gr->beta) return FEAR_TRAIN_ERR_NULL;
    if (!sep_shape_ok(L, B, H, W) || !ld_ok(ldx, L->cin)) return FEAR_TRAIN_ERR_SHAPE;
    const long M = (long)B * H * W;
    const int K = L->cin, N = L->cout;
    const SepWs ws = sep_ws(M, K, N, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipStream_t sw = wgrad_stream ? static_cast<hipStream_t>(wgrad_stream) : s;
    bn_backward_sums(dy, N, raw, N, vec, 1, L->gamma, gr->gamma, gr->beta, coef, M, N, ws.col, s);
    BnbIn bn{};
    bn.E = raw; bn.coef = coef; bn.lde = N; bn.C = N; bn.mask_a = vec + 2 * N; bn.mask_b = vec + 3 * N;
    if (gemm_lds_applies(M, N, K)) {
        GemmArgs g{};
        g.X = dy; g.ldx = N; g.bn = bn; g.W = L->w_pw; g.Y = dd; g.ldy = K; g.M = (int)M; g.K = N; g.N = K;
        launch_gemm_lds<2, 0, true>(g, s, nullptr);
    } else {
        PwBwdArgs a{};
        a.G = dy; a.ldg = N; a.bn = bn; a.W = L->w_pw; a.Y = dd; a.ldy = K; a.M = (int)M; a.Kred = N; a.Nout = K;
        int nt = 1;
        const dim3 grid = dgrad_grid(M, N, K, &nt);
        launch_pw_bwd<false>(a, grid, nt, s);
    }
    if (sw != s && !stream_follow(sw, s)) return FEAR_TRAIN_ERR_HIP;
    int rc = wgrad_impl(dy, N, 0, d, K, 0, gr->w_pw, ws.wg, ws.wg_bytes, M, K, N, 1, sw, nullptr, nullptr, 0, &bn);
    if (rc != FEAR_TRAIN_OK) return rc;
    rc = dw_wgrad_impl(dd, K, x, ldx, gr->w_dw, ws.taps, ws.taps_bytes, B, H, W, K, 3, 1, sw, nullptr, nullptr, 0);
    if (rc != FEAR_TRAIN_OK) return rc;
    rc = dw_dgrad_impl(dd, K, L->w_dw, dx, K, B, H, W, K, 3, 1, s);
    if (rc != FEAR_TRAIN_OK) return rc;
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

// ------------------------------------------------------------------------------------------------
// The stem (3 x 3 stride-2 conv 3 -> 16 + BatchNorm + ReLU, fbnet_c stages[0]) on the NCHW image: fear_pwbn_train_* over
// fear_stem_im2col's rows with the rows never materialised — the forward GEMM and the weight gradient gather them (StemIn).
static bool stem_shape_ok(long n, int H, int W) { return n >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && n * (H / 2) * (W / 2) < 0x7fffffffL / 28; }

size_t fear_stem_workspace_bytes(long n, int H, int W) {
    if (!stem_shape_ok(n, H, W)) return 0;
    const long M = n * (H / 2) * (W / 2);
    return block_ws(M, M, 28, 16, 16, 3, nullptr).total;
}

int fear_stem_train_forward(const float* x_nchw, const float* w, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float* raw, float* vec, float* out, long n, int H, int W, double momentum, double eps, float* workspace,
                            size_t ws_bytes, void* stream) {
    if (!x_nchw || !w || !gamma || !beta || !raw || !vec || !out || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (!stem_shape_ok(n, H, W)) return FEAR_TRAIN_ERR_SHAPE;
    const long M = n * (H / 2) * (W / 2);
    const BlockWs ws = block_ws(M, M, 28, 16, 16, 3, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PwStatArgs a{};
    a.stem_img = x_nchw; a.stem_H = H; a.stem_W = W; a.W = w; a.Y = raw; a.ldy = 16; a.M = (int)M; a.K = 28; a.N = 16; a.partial = ws.col;
    int nt = 1;
    const dim3 grid = stat_grid(M, 28, 16, &nt, &a.row_tiles);      // (16 output channels: one column tile, nt = 1)
    if (nt != 1 || (size_t)grid.x * 2 * 16 * sizeof(double) > ws.col_bytes) return FEAR_TRAIN_ERR_WORKSPACE;
    hipLaunchKernelGGL((pw_stat_kernel<1, true>), grid, dim3(256), 0, s, a);
    finalize_forward(ws.col, (int)grid.x, 16, (double)M, gamma, beta, vec, running_mean, running_var, momentum, eps, s);
    BnActArgs k{};
    k.X = raw; k.Y = out; k.in.a = vec + 2 * 16; k.in.b = vec + 3 * 16; k.in.relu = 1; k.M = M; k.C = 16; k.ldx = 16; k.ldy = 16;
    hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)((M * 4 + 255) / 256)), dim3(256), 0, s, k);
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

int fear_stem_train_backward(const float* dy, const float* raw, const float* vec, const float* x_nchw, const float* gamma, float* dw,
                             float* dgamma, float* dbeta, long n, int H, int W, float* workspace, size_t ws_bytes, void* stream,
                             void* wgrad_stream) {
    if (!dy || !raw || !vec || !x_nchw || !gamma || !dw || !dgamma || !dbeta || !workspace) return FEAR_TRAIN_ERR_NULL;
    if (!stem_shape_ok(n, H, W)) return FEAR_TRAIN_ERR_SHAPE;
    const long M = n * (H / 2) * (W / 2);
    const BlockWs ws = block_ws(M, M, 28, 16, 16, 3, workspace);
    if (ws_bytes < ws.total) return FEAR_TRAIN_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipStream_t sw = wgrad_stream ? static_cast<hipStream_t>(wgrad_stream) : s;
    if (sw != s && !stream_follow(s, sw)) return FEAR_TRAIN_ERR_HIP;      // (the coefficient vectors live in the shared workspace, as in fear_pwbn_train_backward)
    bn_backward_sums(dy, 16, raw, 16, vec, 1, gamma, dgamma, dbeta, ws.coef, M, 16, ws.col, s);
    BnbIn bn{};
    bn.E = raw; bn.coef = ws.coef; bn.lde = 16; bn.C = 16; bn.mask_a = vec + 2 * 16; bn.mask_b = vec + 3 * 16;
    if (sw != s && !stream_follow(sw, s)) return FEAR_TRAIN_ERR_HIP;
    const StemIn st{x_nchw, H, W};
    const int rc = wgrad_impl(dy, 16, 0, nullptr, 28, 0, dw, ws.wg, ws.wg_bytes, M, 28, 16, 1, sw, nullptr, nullptr, 0, &bn, &st);
    if (rc != FEAR_TRAIN_OK) return rc;
    LAUNCH_CHECK();
    return FEAR_TRAIN_OK;
}

}  // extern "C"
