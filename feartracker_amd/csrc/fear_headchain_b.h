// fear_headchain_b.h — the one-launch BoxTower (fear_headchain.h) for FEAR_OPT_MATH = 2, the bf16 matrix-pipe mode of BASELINE
// configs[3]: same structure — a workgroup per (crop, branch), four SepConvs back to back, eight passes of 32 output channels per
// layer, activations handed from layer to layer through the two-chunk LDS tile — with the pointwise GEMMs on
// v_mfma_f32_16x16x32_bf16 (activations and weights rounded to bf16, fp32 accumulate; depthwise, bias, ReLU fp32: exactly the
// rounding points of the sep16 `*_h` kernels it replaces, so the maps agree with theirs to fp32 summation order).
// What changes with a GEMM that costs a sixteenth: a pass is 32-40 MFMAs, so nothing is worth hiding behind it — the hand-over runs
// right after the tile barrier — and the B fragments are bf16: BOTH the layer in flight and the next layer's (8-10 fragment pairs
// x 2 rows x 4 VGPRs each) stay in registers; there is no scratch.  A K = 32 MFMA step takes the two 16-channel chunks 2P, 2P + 1:
// lane (li, lk) holds channels 4lk..4lk+3 of BOTH chunks of its pixel (the fp32 depthwise's own layout, converted in place), and the
// weight fragments are packed on the host in the same k order.
#pragma once
#include <vector>

namespace fear {

struct HeadChainBBranch {
    const float* W[4];       // per layer: 8 x [NP x 2 fragments (64 lanes x 8 bf16) | bias 32 fp32]   (headchain_b_pack)
    const float* Wd[4];      // per layer: its depthwise taps + bias, NC x [Wd[k*k][16] | bd[16]] fp32
    const float* Z;          // template features [crop][256][64] fp32 (the caller's NCHW (256, 8, 8) tensor)
    long z_stride;
    const float* P_W;        // prediction SepConv: 16 x [Wd | bd] fp32, then 8 fragments (bf16) of its 1x1 (rows >= pred_cout zero)
    const float* P_bp;
    float* P_Y;
    long pred_stride;
    int pred_cout, pred_act;
};

struct HeadChainBArgs {
    const float* X;          // neck output [crop * 256][ldx] fp32
    int ldx;
    int n_crops;             // launch with 16 * ceil(n_crops / 8) workgroups (id mapping as in headchain_kernel)
    int relu_dw, relu_out;
    HeadChainBBranch br[2];
};

template <int KS>
struct HeadChainBGeom {
    static constexpr int C = 256, TZ = 64, CC = C + TZ, S = 16, P = KS / 2, PW = S + 2 * P;
    static constexpr int EQ = (PW * PW * 4 + 63) / 64 * 64, EBUF = 4 * EQ;
    static constexpr int NPASS = 8, NTP = 2, WDF = KS * KS * 16 + 16;
    static constexpr int wpass(int cin) { return (cin / 32) * NTP * 256 + NTP * 16; }
    static constexpr int WMAX = wpass(CC);
    static constexpr int TAPS = (CC / 16) * WDF;                                // one layer's depthwise taps
    static constexpr int ZS = NTP * 16 * TZ;
    static constexpr int PRED = 16 * WDF + 8 * 256;
    static constexpr int ZREG = PRED > 2 * ZS ? PRED : 2 * ZS;
    static constexpr int LDS_FLOATS = 2 * WMAX + NTP * EBUF + 2 * TAPS + ZREG;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// round to nearest even, like v_cvt_pk_bf16_f32
inline unsigned short headchain_b_bf16(float f) {
    unsigned x;
    __builtin_memcpy(&x, &f, 4);
    if ((x & 0x7f800000u) == 0x7f800000u) return (unsigned short)((x >> 16) | ((x & 0xffffu) ? 0x40u : 0));
    x += 0x7fffu + ((x >> 16) & 1u);
    return (unsigned short)(x >> 16);
}
// Host side.  w: the pointwise conv's weights [cout][cin] fp32 (row major).  One bf16 fragment per (pair P of 16-channel input
// chunks, output tile nt): lane l = (n = nt*16 + (l & 15), lk = l >> 4) holds, for j = 0..7, W[n][32P + 16*(j >> 2) + 4*lk + (j & 3)].
inline void headchain_b_push_frag(std::vector<float>& out, const float* w, int cin, int rows, int row0, int P) {
    for (int l = 0; l < 64; ++l) {
        unsigned short hv[8];
        for (int j = 0; j < 8; ++j) {
            const int n = row0 + (l & 15), k = 32 * P + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
            hv[j] = n < rows ? headchain_b_bf16(w[(size_t)n * cin + k]) : 0;
        }
        float f4[4];
        __builtin_memcpy(f4, hv, 16);
        out.insert(out.end(), f4, f4 + 4);
    }
}
inline std::vector<float> headchain_b_pack(const float* w, int cin, int cout, const float* bias) {
    std::vector<float> out;
    for (int p = 0; p < cout / 32; ++p) {
        for (int P = 0; P < cin / 32; ++P)
            for (int nt = 2 * p; nt < 2 * p + 2; ++nt) headchain_b_push_frag(out, w, cin, cout, nt * 16, P);
        out.insert(out.end(), bias + 32 * p, bias + 32 * p + 32);
    }
    return out;
}
// depthwise taps of a layer in the kernels' order: per 16-channel chunk [k*k][16] | bias[16]; dw: [C][k*k] (OIHW with I = 1)
inline std::vector<float> headchain_b_taps(const float* dw, const float* dbias, int c, int ks) {
    std::vector<float> out;
    const int kk = ks * ks;
    for (int c0 = 0; c0 < c; c0 += 16) {
        for (int t = 0; t < kk; ++t)
            for (int ch = 0; ch < 16; ++ch) out.push_back(dw[(size_t)(c0 + ch) * kk + t]);
        for (int ch = 0; ch < 16; ++ch) out.push_back(dbias ? dbias[c0 + ch] : 0.f);
    }
    return out;
}

typedef __bf16 hcb_bf8 __attribute__((ext_vector_type(8)));
typedef float hcb_f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ hcb_bf8 hcb_cvt(const f32x4& a, const f32x4& b) {
    return __builtin_convertvector(__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7), hcb_bf8);
}

template <int KS>
__global__ __launch_bounds__(512) void headchain_b_kernel(HeadChainBArgs a) {
    using G = HeadChainBGeom<KS>;
    using std::integral_constant;
    using V8 = hcb_bf8;
    constexpr int C = G::C, TZ = G::TZ, CC = G::CC, S = G::S, P = G::P, PW = G::PW, EP = 4, EQ = G::EQ, EBUF = G::EBUF;
    constexpr int NPASS = G::NPASS, NTP = G::NTP, WDF = G::WDF, WMAX = G::WMAX, TAPS = G::TAPS, ZS = G::ZS;
    constexpr int NS = KS * (KS + 1), RA = 3, NTZ = TZ / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Wb = lds;                      // [2][WMAX]
    float* const Et = lds + 2 * WMAX;           // [NTP][EBUF]
    float* const Tp = Et + NTP * EBUF;          // [2][TAPS]: depthwise taps of layer L in Tp[L & 1]
    float* const Zr = Tp + 2 * TAPS;            // template slices | prediction weights

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const unsigned wg = blockIdx.x;
    const int branch = (wg >> 3) & 1;
    const long crop = (long)(wg >> 4) * 8 + (wg & 7);
    const HeadChainBBranch& b = a.br[branch];
    if (crop >= a.n_crops) return;
    const int y0 = wave * 2;

    for (int i = tid * 4; i < NTP * EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(Et + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    lds_copy_async<G::wpass(C)>(b.W[0], Wb, wave, lane);
    lds_copy_async<C / 16 * WDF>(b.Wd[0], Tp, wave, lane);
    lds_copy_async<CC / 16 * WDF>(b.Wd[1], Tp + TAPS, wave, lane);
    lds_copy_async<ZS>(b.Z + crop * b.z_stride, Zr, wave, lane);

    auto tile_put = [&](int s, const f32x4& v0, const f32x4& v1) {
        float* E = Et + s * EBUF;
        *reinterpret_cast<f32x4*>(E + ((y0 + P) * PW + li + P) * EP + lk * EQ) = v0;
        *reinterpret_cast<f32x4*>(E + ((y0 + 1 + P) * PW + li + P) * EP + lk * EQ) = v1;
    };
    auto tile_dw = [&](int s, const float* wdc, f32x4& o0, f32x4& o1, bool relu) {
        const float* wd = wdc + lk * 4;
        const float* e0 = Et + s * EBUF + (y0 * PW + li) * EP + lk * EQ;
        f32x4 n0 = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16), n1 = n0, wprev = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 ev[RA], wv[RA];
#pragma unroll
        for (int t = 0; t < RA; ++t) {
            const int kx = t / (KS + 1), iy = t % (KS + 1);
            ev[t] = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * EP);
            if (iy < KS) wv[t] = *reinterpret_cast<const f32x4*>(wd + (iy * KS + kx) * 16);
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int iy = t % (KS + 1);
            const f32x4 e = ev[t % RA], w = wv[t % RA];
            if (t + RA < NS) {
                const int kx2 = (t + RA) / (KS + 1), iy2 = (t + RA) % (KS + 1);
                ev[t % RA] = *reinterpret_cast<const f32x4*>(e0 + (iy2 * PW + kx2) * EP);
                if (iy2 < KS) wv[t % RA] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 16);
            }
            if (iy < KS) pk_fma4(n0, e, w);
            if (iy >= 1) pk_fma4(n1, e, wprev);
            wprev = w;
        }
        pk_fma_settle(n0, n1);
        if (relu) {
            n0.x = fmaxf(n0.x, 0.f); n0.y = fmaxf(n0.y, 0.f); n0.z = fmaxf(n0.z, 0.f); n0.w = fmaxf(n0.w, 0.f);
            n1.x = fmaxf(n1.x, 0.f); n1.y = fmaxf(n1.y, 0.f); n1.z = fmaxf(n1.z, 0.f); n1.w = fmaxf(n1.w, 0.f);
        }
        o0 = n0;
        o1 = n1;
    };
    // two tile slots -> one bf16 B fragment pair (rows y0, y0 + 1) of the depthwise with the taps at wd (2 x WDF)
    auto tile_dw_pair = [&](const float* wd, V8& o0, V8& o1, bool relu) {
        f32x4 a0, a1, b0, b1;
        tile_dw(0, wd, a0, a1, relu);
        tile_dw(1, wd + WDF, b0, b1, relu);
        o0 = hcb_cvt(a0, b0);
        o1 = hcb_cvt(a1, b1);
    };

    // B fragments [pair of input chunks][row]: layers 0 and 2 read dA and fill dB, layers 1 and 3 the other way round
    V8 dA[C / 32][2], dB[CC / 32][2];

    // ---------------- prologue: the neck output -> depthwise of layer 0 -> dA
    {
        const float* X0 = a.X + crop * 256 * a.ldx;
        f32x4 x[C / 16][2];
#pragma unroll
        for (int c = 0; c < C / 16; ++c)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                x[c][mt] = *reinterpret_cast<const f32x4*>(X0 + (long)((y0 + mt) * S + li) * a.ldx + c * 16 + lk * 4);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < C / 32; ++r) {
            tile_put(0, x[2 * r][0], x[2 * r][1]);
            tile_put(1, x[2 * r + 1][0], x[2 * r + 1][1]);
            __syncthreads();
            tile_dw_pair(Tp + (2 * r) * WDF, dA[r][0], dA[r][1], a.relu_dw);
            __syncthreads();
        }
    }

    f32x4 cacc[2][NTZ];
    f32x4 pacc[2];

    // One SepConv layer: 8 passes.  MODE 0 plain, 1 + correlation (layer 0), 2 prediction head (layer 3).  L = layer index.
    auto layer = [&](auto l_tag, auto cin_tag, auto mode_tag, auto& din, auto& dout) {
        constexpr int L = decltype(l_tag)::value, CIN = decltype(cin_tag)::value, MODE = decltype(mode_tag)::value;
        constexpr int NP = CIN / 32, WP = G::wpass(CIN);
        const float* Wl = b.W[L];
        const float* taps = Tp + ((L + 1) & 1) * TAPS;             // the NEXT layer's depthwise taps
        if (MODE == 1) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < NTZ; ++q) cacc[mt][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (MODE == 2) pacc[0] = pacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (the passes are written out: their index selects REGISTERS — dout[p] — so it must be a constant)
        static_for<0, NPASS>([&](auto p_tag) {
            constexpr int p = decltype(p_tag)::value;
            // asynchronous copies for the next pass / layer (their buffers were last read before the previous pass's barrier B)
            if (p + 1 < NPASS) lds_copy_async<WP>(Wl + (long)(p + 1) * WP, Wb + ((p + 1) & 1) * WMAX, wave, lane);
            else if (L < 3) lds_copy_async<G::wpass(L == 0 ? CC : C)>(b.W[L < 3 ? L + 1 : 3], Wb + ((p + 1) & 1) * WMAX, wave, lane);
            if (MODE == 1 && p + 1 < NPASS) lds_copy_async<ZS>(b.Z + crop * b.z_stride + (long)(p + 1) * ZS, Zr + ((p + 1) & 1) * ZS, wave, lane);
            if (p == 1 && L < 2) {
                // the depthwise taps of layer L + 2 into the buffer layer L's own taps left (read for the last time in layer L - 1)
                if (L == 0) lds_copy_async<C / 16 * WDF>(b.Wd[2], Tp + (L & 1) * TAPS, wave, lane);
                else lds_copy_async<C / 16 * WDF>(b.Wd[3], Tp + (L & 1) * TAPS, wave, lane);
            }
            if (p == 1 && L == 2) lds_copy_async<G::PRED>(b.P_W, Zr, wave, lane);      // (Zr: last read in layer 0's correlation)
            const float* wb = Wb + (p & 1) * WMAX;
            f32x4 acc[2][NTP];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int P2 = 0; P2 < NP; ++P2)
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    const V8 wf = *reinterpret_cast<const V8*>(wb + (P2 * NTP + nt) * 256 + lane * 4);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, din[P2][0], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, din[P2][1], acc[1][nt], 0, 0, 0);
                }
            f32x4 v[2][NTP];
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(wb + NP * NTP * 256 + nt * 16 + lk * 4);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 t = acc[mt][nt] + bv;
                    if (a.relu_out) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
                    v[mt][nt] = t;
                }
            }
            if (MODE == 1) {
                // correlation z^T y of this pass's 32 channels: one K = 32 step per 16 template positions.  A fragment: lane
                // (t = q*16 + li, lk) holds z[32p + 16*(j >> 2) + 4*lk + (j & 3)][t], j = 0..7 (rounded to bf16 like the activations)
                const float* zs = Zr + (p & 1) * ZS + li;
                const V8 y0v = hcb_cvt(v[0][0], v[0][1]), y1v = hcb_cvt(v[1][0], v[1][1]);
#pragma unroll
                for (int q = 0; q < NTZ; ++q) {
                    f32x4 za, zb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        za[i] = zs[(lk * 4 + i) * TZ + q * 16];
                        zb[i] = zs[(16 + lk * 4 + i) * TZ + q * 16];
                    }
                    const V8 zf = hcb_cvt(za, zb);
                    cacc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, y0v, cacc[0][q], 0, 0, 0);
                    cacc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, y1v, cacc[1][q], 0, 0, 0);
                }
            }
            __syncthreads();                       // B: next block landed; every wave is past the previous hand-over's tile reads
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) tile_put(nt, v[0][nt], v[1][nt]);
            __syncthreads();                       // A: tile complete
            if (MODE != 2) {
                tile_dw_pair(taps + (2 * p) * WDF, dout[p][0], dout[p][1], a.relu_dw);
            } else {
                V8 n0, n1;
                tile_dw_pair(Zr + (2 * p) * WDF, n0, n1, false);
                const V8 wq = *reinterpret_cast<const V8*>(Zr + 16 * WDF + p * 256 + lane * 4);
                pacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq, n0, pacc[0], 0, 0, 0);
                pacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq, n1, pacc[1], 0, 0, 0);
            }
        });
        if (MODE == 1) {
            // the 64 correlation channels = input chunks 16..19 (pairs 8, 9) of layer 1
#pragma unroll
            for (int r = 0; r < NTZ / 2; ++r) {
                __syncthreads();
                tile_put(0, cacc[0][2 * r], cacc[1][2 * r]);
                tile_put(1, cacc[0][2 * r + 1], cacc[1][2 * r + 1]);
                __syncthreads();
                tile_dw_pair(taps + (C / 16 + 2 * r) * WDF, dout[C / 32 + r][0], dout[C / 32 + r][1], a.relu_dw);
            }
        }
        if (MODE == 2 && lk == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int px = (y0 + mt) * S + li;
                const float vals[4] = {pacc[mt].x, pacc[mt].y, pacc[mt].z, pacc[mt].w};
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n < b.pred_cout) {
                        float o = vals[n] + b.P_bp[n];
                        if (b.pred_act == 2) o = expf(o);
                        b.P_Y[crop * b.pred_stride + n * 256 + px] = o;
                    }
            }
        }
    };
    layer(integral_constant<int, 0>{}, integral_constant<int, C>{}, integral_constant<int, 1>{}, dA, dB);
    layer(integral_constant<int, 1>{}, integral_constant<int, CC>{}, integral_constant<int, 0>{}, dB, dA);
    layer(integral_constant<int, 2>{}, integral_constant<int, C>{}, integral_constant<int, 0>{}, dA, dB);
    layer(integral_constant<int, 3>{}, integral_constant<int, C>{}, integral_constant<int, 2>{}, dB, dA);
}

}  // namespace fear
