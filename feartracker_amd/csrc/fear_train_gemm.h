// fear_train_gemm.h — LDS-staged, double-buffered fp32 MFMA GEMM of the training step's small-map layers (round 5).
//
//     Y[m][n] = sum_k X'[m][k] W'[k][n]        M = pixels of a 32 x 32 ... 8 x 8 map of a batch (8 192 ... 131 072 rows), K, N = 16 ... 672
//
// The first-generation training GEMMs (pw_mfma_kernel, pw_stat_kernel, pw_bwd_kernel) fetch every MFMA fragment straight from
// L1 / L2 right in front of the MFMAs that consume it: fine where a launch is thousands of workgroups deep and bandwidth bound (the
// 128 x 128 ... 32 x 32 maps), but the stride-16 stage and the head are 256 row blocks — one wave per SIMD — and every trip of the
// k loop there is a memory round trip followed by its MFMAs: 35-55 TFLOP/s of the 157 (profiles/r05_train_kernel_stats.csv before
// this kernel: the 672 -> 112 projection 178 us for 31 us of MFMAs).  Here a workgroup of four waves owns a (64 MT) x (16 NTW)
// tile; per k-group of 16 the X tile and the W tile are staged in LDS in fragment order ([k quad][row][4]: a wave's
// ds_read_b128 of one fragment is 16 consecutive 16-byte words per lane group, conflict free), two buffers, and the global loads of
// k-group g + 1 are issued BEFORE the MFMAs of group g and committed to the other buffer after them: one barrier per k-group,
// no load on the critical path.  Weight fragments are read once per workgroup from global memory instead of once per wave, and a
// K-major weight matrix (the input-gradient GEMMs) is transposed on its way into LDS instead of gathered with four scalar
// loads per fragment.
//
// The fused prologues / epilogues are the first-generation kernels', applied where an element passes through registers anyway:
//   XT  = 0 plain | 1 activation on load, max(fma(x, a, b), 0) (ActIn) | 2 BatchNorm backward on load (BnbIn) | 3 its E-free
//         bracket on load (bn.coef only), followed by plain columns of a second tensor X2 (an expansion's input gradient, see BnbIn)
//   EPI = 0 (+ bias) (+ R) -> Y | 1 Y + column sums sum(y), sum(y^2) | 2 Y masked by act(D) > 0 + column sums sum(y), sum(y * dhat)
// Same products in the same k order as those kernels (k ascending, four per MFMA).
//
// Included by fear_train.hip inside its anonymous namespace.

struct GemmArgs {
    const float* X;      // [M][ldx]
    const float* X2;     // XT 3: the reduction's columns K1 ... K - 1 come from this second tensor [M][ldx2], as loaded
    const float* W;      // WKN ? [K][N] row-major : [N][K] row-major
    const float* bias;   // EPI 0: [N] or nullptr
    const float* R;      // EPI 0: optional [M][ldr] added
    float* Y;            // [M][ldy]
    ActIn in;            // XT 1
    BnbIn bn;            // XT 2
    const float* D;      // EPI 2: [M][ldd] raw tensor behind the ReLU that masks Y
    const float* dvec;   // EPI 2: [4][N] mean | rstd | a | b of D's BatchNorm
    double* partial;     // EPI 1, 2: [gridDim.x][2][N]
    int ldx, ldx2, ldr, ldy, ldd;
    int M, K, N;
    int K1;              // XT 3: a multiple of 16 (a stage never straddles the two sources)
    int relu;            // EPI 0: max(., 0) on the way out
};

template <int MT, int NTW, int XT, int EPI, bool WKN>
__global__ __launch_bounds__(256) void gemm_lds_kernel(GemmArgs a) {
    constexpr int BM = 64 * MT, BN = 16 * NTW;
    constexpr int XQ = BM * 4;                 // float4s of an X stage
    constexpr int WQ = BN * 4;                 // float4s of a W stage
    constexpr int XL = MT;                     // X float4s per thread and stage
    constexpr int WL = (WQ + 255) / 256;       // W float4s per thread and stage
    // W stage: [buffer][k quad][column] float4s (a fragment = one ds_read_b128) — or, for a K-major weight matrix (WKN), the rows as
    // they are loaded: [buffer][k][BN + 4] floats (round 6).  The K-major form used to be transposed on its way into LDS with four
    // ds_write_b32 per float4, 64 B between consecutive lanes: a 16-way bank conflict on every store, ~900 LDS cycles per k-group
    // against the ~900 cycles of MFMAs a SIMD has per k-group — those launches were LDS-bound (conflict share 0.76, MFMA pipe 0.30:
    // profiles/r06_train_sq_counters.txt).  Now its float4s go in with one conflict-free ds_write_b128 and a fragment is four
    // ds_read_b32 (lane group lk reads rows 4 lk .. 4 lk + 3; the row pitch BN + 4 puts the two lane groups of a 32-lane read on
    // disjoint banks).
    constexpr int WTP = BN + 4;                // floats per k row of the K-major stage
    __shared__ f32x4 Xs[2][4][BM];             // [buffer][k quad][row]
    __shared__ f32x4 Wraw[2 * 4 * WTP];        // >= 2 x 4 x BN float4s either way
    f32x4 (*Ws)[4][BN] = reinterpret_cast<f32x4 (*)[4][BN]>(Wraw);
    float (*Wt)[16][WTP] = reinterpret_cast<float (*)[16][WTP]>(Wraw);
    __shared__ double red[EPI ? 4 : 1][2][EPI ? BN : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool xact = XT == 1 && a.in.a != nullptr;
    const bool bmask = XT == 2 && a.bn.mask_a != nullptr;

    // ---- staging registers: thread t stages k quad t & 3 of rows (t >> 2) + 64 j; the per-channel vectors of the prologue travel
    //      with the stage (requested with its loads, used when it is committed).
    //      (The four quads of a row sit in planes a multiple of 128 B apart, so the eight lanes of a ds_write_b128 service group —
    //      two rows x four quads — meet 4-way on the banks: the 0.35 conflict share of the forward GEMMs in
    //      profiles/r06_train_sq_counters.txt.  Dealing eight consecutive lanes eight consecutive rows of ONE quad makes the stores
    //      conflict free and was measured SLOWER, 84.7 -> 90.4 us for 32 768 x 672 -> 112 with statistics: a quad of lanes then
    //      loads 16 bytes from each of four rows instead of 64 contiguous bytes, and the global side costs more than the LDS side
    //      gains — tools/gemm_ab.py, profiles/r06_gemm_ab.txt.)
    f32x4 xr[XL], er[XT == 2 ? XL : 1], wr[WL], pv[XT == 2 ? 6 : (XT == 1 || XT == 3 ? 2 : 1)];
    const int kq = tid & 3, row_t = tid >> 2;
    bool x_ok[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) x_ok[j] = m0 + row_t + 64 * j < a.M;
    auto stage_load = [&](int k0) {
        const int k = k0 + 4 * kq;
        const bool kok = k < a.K;                        // K is a multiple of 4
        const bool second = XT == 3 && k0 >= a.K1;      // (uniform over the workgroup)
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const bool ok = x_ok[j] && kok;
            const long off = (long)(m0 + row_t + 64 * j);
            if (XT == 3 && second) xr[j] = ok ? *reinterpret_cast<const f32x4*>(a.X2 + off * a.ldx2 + (k - a.K1)) : zero;
            else xr[j] = ok ? *reinterpret_cast<const f32x4*>(a.X + off * a.ldx + k) : zero;
            if (XT == 2) er[XT == 2 ? j : 0] = ok ? *reinterpret_cast<const f32x4*>(a.bn.E + off * a.bn.lde + k) : zero;
        }
        if (XT == 1 && xact && kok) {
            pv[0] = *reinterpret_cast<const f32x4*>(a.in.a + k);
            pv[XT == 1 ? 1 : 0] = *reinterpret_cast<const f32x4*>(a.in.b + k);
        }
        if (XT == 3 && kok && !second) {
            // A | A (mu Q - s1): the E-free bracket of BnbIn as one fma per element
            const f32x4 cA = *reinterpret_cast<const f32x4*>(a.bn.coef + k), cs1 = *reinterpret_cast<const f32x4*>(a.bn.coef + a.bn.C + k);
            const f32x4 cmu = *reinterpret_cast<const f32x4*>(a.bn.coef + 2 * a.bn.C + k), cQ = *reinterpret_cast<const f32x4*>(a.bn.coef + 3 * a.bn.C + k);
            pv[0] = cA;
            pv[XT == 3 ? 1 : 0] = cA * (cmu * cQ - cs1);
        }
        if (XT == 2 && kok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pv[XT == 2 ? q : 0] = *reinterpret_cast<const f32x4*>(a.bn.coef + q * a.bn.C + k);
            if (bmask) {
                pv[XT == 2 ? 4 : 0] = *reinterpret_cast<const f32x4*>(a.bn.mask_a + k);
                pv[XT == 2 ? 5 : 0] = *reinterpret_cast<const f32x4*>(a.bn.mask_b + k);
            }
        }
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const int idx = tid + 256 * j;
            wr[j] = zero;
            if (idx < WQ) {
                if (WKN) {
                    // W[K][N]: float4 along n of row k0 + kk
                    const int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
                    const int n = n0 + 4 * n4;
                    if (k0 + kk < a.K && n < a.N) wr[j] = *reinterpret_cast<const f32x4*>(a.W + (long)(k0 + kk) * a.N + n);      // N is a multiple of 4
                } else {
                    const int col = idx >> 2;
                    if (n0 + col < a.N && kok) wr[j] = *reinterpret_cast<const f32x4*>(a.W + (long)(n0 + col) * a.K + k);      // (idx & 3 == kq)
                }
            }
        }
    };
    auto stage_store = [&](int k0, int buf) {
        const bool kok = k0 + 4 * kq < a.K;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            f32x4 v = xr[j];
            if (x_ok[j] && kok) {
                if (XT == 1 && xact) {
                    v = act4(v, pv[0], pv[XT == 1 ? 1 : 0], a.in.relu != 0);
                } else if (XT == 3) {
                    if (k0 < a.K1) v = act4(v, pv[0], pv[XT == 3 ? 1 : 0], false);
                } else if (XT == 2) {
                    const f32x4 e = er[XT == 2 ? j : 0];
                    if (bmask) v = relu_mask4(v, e, pv[XT == 2 ? 4 : 0], pv[XT == 2 ? 5 : 0]);
                    v = bnb4(v, e, pv[0], pv[XT == 2 ? 1 : 0], pv[XT == 2 ? 2 : 0], pv[XT == 2 ? 3 : 0]);
                }
            }
            Xs[buf][kq][row_t + 64 * j] = v;
        }
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const int idx = tid + 256 * j;
            if (idx < WQ) {
                if (WKN) {
                    const int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
                    *reinterpret_cast<f32x4*>(&Wt[buf][kk][4 * n4]) = wr[j];
                } else {
                    Ws[buf][kq][idx >> 2] = wr[j];
                }
            }
        }
    };

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = zero;

    const int nk = (a.K + 15) >> 4;
    stage_load(0);
    stage_store(0, 0);
    __syncthreads();
    auto kstep = [&](int kg, auto more_c) {
        constexpr bool MORE = decltype(more_c)::value;
        const int buf = kg & 1;
        if (MORE) stage_load((kg + 1) * 16);
        f32x4 xf[MT], wf[NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = Xs[buf][lk][wave * 16 * MT + mt * 16 + li];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            if (WKN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[nt][i] = Wt[buf][4 * lk + i][nt * 16 + li];
            } else {
                wf[nt] = Ws[buf][lk][nt * 16 + li];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][i], xf[mt][i], acc[mt][nt], 0, 0, 0);
        if (MORE) stage_store((kg + 1) * 16, buf ^ 1);
        __syncthreads();
    };
    for (int kg = 0; kg < nk - 1; ++kg) kstep(kg, std::true_type{});      // (the last k-group peeled: no prefetch branches in the loop)
    kstep(nk - 1, std::false_type{});

    // ---- epilogue: lane holds columns n0 + nt * 16 + 4 lk + {0..3} of rows m0 + wave * 16 MT + mt * 16 + li
    const int mw = m0 + wave * 16 * MT;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = n0 + nt * 16 + lk * 4;
        const bool nok = n < a.N;
        if (EPI == 0) {
            if (!nok) continue;
            f32x4 b4 = zero;
            if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const long m = mw + mt * 16 + li;
                if (m >= a.M) continue;
                f32x4 v = acc[mt][nt] + b4;
                if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
            }
        } else {
            f32x4 dmu = zero, drs = zero, da = zero, db = zero;
            if (EPI == 2 && nok) {
                dmu = *reinterpret_cast<const f32x4*>(a.dvec + n); drs = *reinterpret_cast<const f32x4*>(a.dvec + a.N + n);
                da = *reinterpret_cast<const f32x4*>(a.dvec + 2 * a.N + n); db = *reinterpret_cast<const f32x4*>(a.dvec + 3 * a.N + n);
            }
            f32x4 s1 = zero, s2 = zero;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const long m = mw + mt * 16 + li;
                if (m >= a.M || !nok) continue;
                f32x4 v = acc[mt][nt];
                if (EPI == 2) {
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(a.D + m * a.ldd + n);
                    v = relu_mask4(v, dv, da, db);
                    s2 += v * ((dv - dmu) * drs);
                } else {
                    s2 += v * v;
                }
                s1 += v;
                *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t1 = row16_sum(s1[c]), t2 = row16_sum(s2[c]);
                if (li == 0) {
                    red[EPI ? wave : 0][0][EPI ? nt * 16 + lk * 4 + c : 0] = (double)t1;
                    red[EPI ? wave : 0][1][EPI ? nt * 16 + lk * 4 + c : 0] = (double)t2;
                }
            }
        }
    }
    if (EPI) {
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, col = i - which * BN;
            const int n = n0 + col;
            if (n < a.N)
                a.partial[((long)blockIdx.x * 2 + which) * a.N + n] =
                    ((red[0][which][EPI ? col : 0] + red[EPI ? 1 : 0][which][EPI ? col : 0]) + red[EPI ? 2 : 0][which][EPI ? col : 0]) +
                    red[EPI ? 3 : 0][which][EPI ? col : 0];      // fixed order
        }
    }
}

// which launches run on the LDS-staged kernel: few row blocks (the first-generation kernels are bandwidth bound and fine above
// that) and a reduction long enough for the pipeline to matter
// (the 32 x 32 maps of a 128-pair batch — 131 072 rows — included: their first-generation kernels with a 32-channel reduction and six
//  or twelve output tiles per pass sit at one wave per SIMD and 1.3 TB/s; measured 15.63 -> 15.45 ms per step.  From 524 288 rows up
//  the first generation is as fast or faster: 15.53 / 15.72 ms with the 64 x 64 / 128 x 128 maps included.)
#ifndef FEAR_GEMM_LDS_MAX_ROWS
#define FEAR_GEMM_LDS_MAX_ROWS 131072
#endif
bool gemm_lds_applies(long M, int K, int N) { return M <= FEAR_GEMM_LDS_MAX_ROWS && K >= 32 && N >= 16; }

// column tiles per workgroup: the widest of {8, 7, 6, 4} that divides the tile count, else 8 with a ragged last column block
int gemm_lds_ntw(int n_tiles) {
    for (int c : {8, 7, 6, 4})
        if (n_tiles % c == 0) return c;
    return n_tiles < 4 ? 4 : 8;
}

template <int XT, int EPI, bool WKN>
int launch_gemm_lds(const GemmArgs& a, hipStream_t s, int* row_blocks) {
    const int n_tiles = (a.N + 15) / 16;
    const int ntw = gemm_lds_ntw(n_tiles);
    const int col_blocks_ = (n_tiles + ntw - 1) / ntw;
    // 128-row tiles when that still gives every CU two workgroups, else 64-row tiles
    const bool mt2 = (long)((a.M + 127) / 128) * col_blocks_ >= 512;
    const int bm = mt2 ? 128 : 64;
    const dim3 grid((unsigned)((a.M + bm - 1) / bm), (unsigned)col_blocks_);
    if (row_blocks) *row_blocks = (int)grid.x;
#define FEAR_GEMM_CASE(NTW_)                                                                                          \
    case NTW_:                                                                                                        \
        if (mt2) hipLaunchKernelGGL((gemm_lds_kernel<2, NTW_, XT, EPI, WKN>), grid, dim3(256), 0, s, a);              \
        else hipLaunchKernelGGL((gemm_lds_kernel<1, NTW_, XT, EPI, WKN>), grid, dim3(256), 0, s, a);                  \
        break;
    switch (ntw) {
        FEAR_GEMM_CASE(4)
        FEAR_GEMM_CASE(6)
        FEAR_GEMM_CASE(7)
        default:
        FEAR_GEMM_CASE(8)
    }
#undef FEAR_GEMM_CASE
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient dW[n][k] = sum_m dY'[m][n] X'[m][k] on a 128 (n) x 128 (k) workgroup tile, rows staged through LDS.
// pw_wgrad_kernel's workgroup owns a 64 x 64 tile and its four waves split the ROWS: every dY element is fetched once per 64
// columns of k and every X element once per 64 of n — for dW[672][112] over 32 768 rows that is 513 MB through L2 per call (with
// the BatchNorm-backward operand), 5 TB/s at the 100 us the call takes: the kernel is bound by L2, not by its 31 us of MFMAs.
// Here the four waves split the TILE (2 x 2 sub-tiles of 64 x 64) and share the rows: a stage of 16 rows of dY (128 channels) and
// X (128 channels) is loaded once per workgroup into LDS — half the L2 traffic per output element twice over — double buffered
// like gemm_lds_kernel (next stage's loads in flight under this stage's 64 MFMAs per wave, one barrier per stage), and no
// cross-wave reduction at the end: every wave owns its sub-tile.  Fragments by pw_wgrad_kernel's trick: lane (li, lk) reads the
// float4s dY[row 4 s + lk][64 wn + 4 li ..] and X[row 4 s + lk][64 wk + 4 li ..]; MFMA (p, q) takes component p of the first as A and
// q of the second as B.  Operand prologues as in WgradArgs (activation on X, BatchNorm backward on dY).  One rank, crops == 1.
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(WgradArgs a) {
    constexpr int R = 16, TC = 128;
    __shared__ f32x4 Ds[2][R][TC / 4];
    __shared__ f32x4 Xs[2][R][TC / 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    const int nt = blockIdx.x / a.k_tiles, kt = blockIdx.x % a.k_tiles;        // 128-wide tiles here
    const int slice = blockIdx.y;
    const int n0 = nt * TC, k0 = kt * TC;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging: thread t stages channel quad t & 31 of rows (t >> 5) and (t >> 5) + 8
    const int c4 = tid & 31, r_t = tid >> 5;
    const int nch = n0 + 4 * c4, kch = k0 + 4 * c4;
    const bool nv = nch < a.N, kv = kch < a.K;
    const bool act = a.act_a != nullptr, bnb = a.bn.coef != nullptr, bne = a.bn.E != nullptr, bmask = bne && a.bn.mask_a != nullptr;
    f32x4 ia = zero, ib = zero, cA = zero, cs1 = zero, cmu = zero, cQ = zero, cma = zero, cmb = zero;
    if (act && kv) { ia = *reinterpret_cast<const f32x4*>(a.act_a + kch); ib = *reinterpret_cast<const f32x4*>(a.act_b + kch); }
    if (bnb && nv) {
        cA = *reinterpret_cast<const f32x4*>(a.bn.coef + nch); cs1 = *reinterpret_cast<const f32x4*>(a.bn.coef + a.bn.C + nch);
        cmu = *reinterpret_cast<const f32x4*>(a.bn.coef + 2 * a.bn.C + nch); cQ = *reinterpret_cast<const f32x4*>(a.bn.coef + 3 * a.bn.C + nch);
        if (bmask) { cma = *reinterpret_cast<const f32x4*>(a.bn.mask_a + nch); cmb = *reinterpret_cast<const f32x4*>(a.bn.mask_b + nch); }
    }
    const long m0 = (long)slice * a.rows_per_slice;
    const long m1 = m0 + a.rows_per_slice < a.M ? m0 + a.rows_per_slice : a.M;
    // two register sets: the loads of stage s + 2 are issued before the MFMAs of stage s and committed to LDS after those of
    // stage s + 1 — a stage is 64 MFMAs per wave (0.85 us), less than a memory round trip, so a prefetch distance of one stage
    // left every stage waiting for its successor's data (33 us floor on launches of 16 stages)
    f32x4 dr[2][2], er[2][2], xr[2][2];
    auto stage_load = [&](long m, int set) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long r = m + r_t + 8 * j;
            const bool rv = r < m1;
            dr[set][j] = rv && nv ? *reinterpret_cast<const f32x4*>(a.dY + r * a.lddy + nch) : zero;
            er[set][j] = bne && rv && nv ? *reinterpret_cast<const f32x4*>(a.bn.E + r * a.bn.lde + nch) : zero;
            xr[set][j] = rv && kv ? *reinterpret_cast<const f32x4*>(a.X + r * a.ldx + kch) : zero;
        }
    };
    auto stage_store = [&](long m, int buf, int set) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool rv = m + r_t + 8 * j < m1;
            f32x4 d = dr[set][j], x = xr[set][j];
            if (bnb && rv && nv) {
                if (bmask) d = relu_mask4(d, er[set][j], cma, cmb);
                d = bnb4(d, er[set][j], cA, cs1, cmu, cQ);
            }
            if (act && rv && kv) x = act4(x, ia, ib, a.act_relu != 0);
            Ds[buf][r_t + 8 * j][c4] = d;
            Xs[buf][r_t + 8 * j][c4] = x;
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = zero;
    // a wave whose 64 x 64 sub-tile lies outside dW altogether (the ragged last tiles) only takes part in the staging
    const bool live = n0 + 64 * wn < a.N && k0 + 64 * wk < a.K;
    const long nst = (m1 - m0 + R - 1) / R;
    if (nst > 0) {
        stage_load(m0, 0);
        if (nst > 1) stage_load(m0 + R, 1);
        stage_store(m0, 0, 0);
    }
    __syncthreads();
    auto step = [&](long st, int set_cur) {      // set_cur = st & 1, a compile-time constant at both call sites
        const int buf = set_cur;
        if (st + 2 < nst) stage_load(m0 + (st + 2) * R, set_cur);            // (stage st left this register set one step ago)
        if (live) {
#pragma unroll
            for (int s4 = 0; s4 < R / 4; ++s4) {
                const f32x4 dv = Ds[buf][4 * s4 + lk][16 * wn + li];
                const f32x4 xv = Xs[buf][4 * s4 + lk][16 * wk + li];
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[p], xv[q], acc[p][q], 0, 0, 0);
            }
        }
        if (st + 1 < nst) stage_store(m0 + (st + 1) * R, buf ^ 1, set_cur ^ 1);
        __syncthreads();
    };
    // (pairs without a branch between the two instantiations of the step, the odd tail peeled: chain16_block's lesson)
    long st = 0;
    for (; st + 1 < nst; st += 2) {
        step(st, 0);
        step(st + 1, 1);
    }
    if (st < nst) step(st, 0);
    if (!live) return;
    // acc[p][q] lane (li, lk), component r  =  dW[n0 + 64 wn + 16 lk + 4 r + p][k0 + 64 wk + 4 li + q]
    float* P = a.P + (long)slice * a.N * a.K;
    const int kcol = k0 + 64 * wk + 4 * li;
    if (kcol >= a.K) return;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nn = n0 + 64 * wn + lk * 16 + r * 4 + p;
            if (nn < a.N) *reinterpret_cast<f32x4*>(P + (long)nn * a.K + kcol) = (f32x4){acc[p][0][r], acc[p][1][r], acc[p][2][r], acc[p][3][r]};
        }
}
