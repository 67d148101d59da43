// fear_kernels.h — gfx950 (CDNA4, wave64) device kernels of the FEAR-XS inference path.
//
// Activations are fp32 NHWC ("pixel-major") inside the engine: a 1x1 convolution is then a
// row-major GEMM  Y[m][n] = sum_k X[m][k] * W[n][k]  with m = pixel, k/n = channels, and both
// operands K-contiguous, which is exactly the shape v_mfma_f32_16x16x4_f32 fragments want.
//
// Kernel inventory (reference operator each one replaces):
//   stem_conv_kernel   3x3 s2 conv 3->16 + ReLU, NCHW in -> NHWC out   (fbnet_c stages[0], via model/blocks.py:29)
//   pw_mfma_kernel     1x1 conv as MFMA GEMM + bias/ReLU/residual       (IR expand/project, AdjustLayer blocks.py:78-81,
//                                                                        SepConv.pointwise blocks.py:67)
//                      also the pixel-wise correlation z^T x            (MobileCorrelation, blocks.py:121-123)
//   dw_conv_kernel     depthwise kxk (k=3/5, s=1/2) + bias/ReLU         (IR depthwise, SepConv.depthwise blocks.py:57-66)
//   pw_small_kernel    1x1 conv with Cout<=4 (+exp), NCHW out           (bbox_pred / cls_pred, blocks.py:167-168,186-192)
//   decode_kernel      sigmoid + arg-max + ltrb->xywh                   (FEARBoxCoder.decode, dataset/box_coder.py:75-107)
//   decode_smooth_kernel  the smooth=True post-processing              (Tracker._postprocess, base_tracker.py:149-205)
//   normalize_kernel   uint8 HWC -> normalised fp32 NCHW                (Tracker._preprocess_image, base_tracker.py:97-103)
#pragma once
#include <type_traits>
#ifndef IR16_GS
#define IR16_GS 2      // tap steps per scheduling group of ir16_interval
#endif
#ifndef IR16_D
#define IR16_D 4       // LDS read-ahead of ir16_interval, in tap steps
#endif
#ifndef IR16H_ASYNC
#define IR16H_ASYNC 1  // ir16h_fused_kernel (blocks with expansion): packed weights global -> LDS by asynchronous copies (0: through registers, rounds 2-5)
#endif
#ifndef IR16H_D
#define IR16H_D 4      // LDS read-ahead of ir16h_fused_kernel's depthwise, in tap steps (two reads per step and channel half)
#endif
#ifndef CHAIN16_PEEL
#define CHAIN16_PEEL 1     // chain16_block: the last chunk peeled out of the chunk loop (0: a run-time branch inside the loop, rounds 1-5)
#endif
#ifndef FEAR_V4_GS
#define FEAR_V4_GS 2      // tap steps per scheduling group of ir_tile_v4_kernel
#endif
#ifndef FEAR_V4_GUARD
#define FEAR_V4_GUARD 0   // 1: skip the expansion MFMAs of m-tiles beyond the clipped region (a scalar branch per MFMA)
#endif
#ifndef FEAR_ABL
#define FEAR_ABL 0      // timing ablations for tools/kbench only (bit mask); the product always builds with 0
                        // (tile kernels, round 4: 8192 keep the barrier after the last chunk, 16384 projection bias loaded in the
                        //  epilogue, 65536 e1 residual re-read from global memory; 4096 = per-phase wall-clock stamps; the other
                        //  bits are named where they are tested)
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace fear {

// Storage of an activation tensor the tile kernels read / write: fp32 (every mode), or bf16 for the HBM-bound front of the trunk
// in the bf16 arithmetic mode (FEAR_OPT_MATH = 2 + FEAR_OPT_BF16_STORE: stem output ... input of the 64 -> 32 block).  The tile
// kernels take the choice as the template parameter IO (bits below); offsets are in ELEMENTS either way.
constexpr int IO_X_BF16 = 1, IO_Y_BF16 = 2, IO_R_BF16 = 4;
typedef __bf16 act_bf4 __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ f32x4 ld_act4(const float* base, long off) {
    if (BF) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
        return (f32x4){__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
    }
    return *reinterpret_cast<const f32x4*>(base + off);
}
template <bool BF>
__device__ __forceinline__ void st_act4(float* base, long off, const f32x4& v) {
    if (BF) *reinterpret_cast<act_bf4*>(reinterpret_cast<unsigned short*>(base) + off) = __builtin_convertvector(v, act_bf4);   // RNE
    else *reinterpret_cast<f32x4*>(base + off) = v;
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution / correlation on the matrix cores.
//
// One wavefront owns MT pixel tiles (16 pixels each) and, per pass, NT channel tiles (16 output
// channels each).  Fragments are loaded straight from global/L2 in MFMA order — no LDS staging:
//   v_mfma_f32_16x16x4_f32: A-operand lane l = A[i = l&15][k = l>>4], B-operand lane l = B[k = l>>4][j = l&15]
// We feed A := weights (i = output channel), B := activations (j = pixel).  A lane loads one float4
//   W[n0 + (l&15)][kg*16 + 4*(l>>4) + 0..3]   and   X[m0 + (l&15)][kg*16 + 4*(l>>4) + 0..3]
// and issues 4 MFMAs (one per float4 component); the k permutation is the same on both operands so
// the sum is unchanged.  The accumulator lane then holds Y[m0 + (l&15)][n0 + 4*(l>>4) + 0..3]: four
// consecutive output channels of one pixel -> one 16-byte store, 64 contiguous bytes per pixel row.
//
// WKN = true: weights are given K-major, W[k][n] (the template features z[c][j] of the correlation,
// NCHW (256, 8x8) as handed over by the caller) with a per-crop stride.
struct PwArgs {
    const float* X;      // [M][ldx]
    const float* W;      // [N][K]  (or [K][N] per crop when WKN)
    const float* bias;   // [N] or nullptr
    const float* R;      // residual [M][ldr] or nullptr
    float* Y;            // [M][ldy]  (or NCHW when nchw_hw > 0)
    int ldx, ldr, ldy;
    int M, K, N;
    int relu;
    int nchw_hw;         // >0: store Y as [M / hw][N][hw] (NCHW), hw = pixels per crop
    int rows_per_crop;   // > 0: every block of this many rows (one crop's pixels) has its own weight matrix ...
    long w_crop_stride;  // ... this many floats after the previous one (the correlation and its gradients); 0: shared weights
};

#ifndef FEAR_PW_KU
#define FEAR_PW_KU(NT) ((NT) <= FEAR_PW_KU_NT ? 2 : 1)   // KU of the throughput / training launches of pw_mfma_kernel<MT, NT, ..>
#endif
#ifndef FEAR_PW_KU_NT
#define FEAR_PW_KU_NT 4
#endif
// KU = k-groups (16 input channels each) whose operands are loaded together before their MFMAs are issued.  1 for the
// throughput launches (the other workgroups of a CU hide the loads); 8 for the small-batch plans, where a launch is a
// handful of workgroups alone on their CUs and every trip of the k loop otherwise costs one L2 / HBM round trip (the 256-channel
// correlation at one crop: 16 trips, 14.7 us).  Same MFMAs in the same order: bit-identical results.
template <int MT, int NT, bool WKN, int KU = 1>
__global__ __launch_bounds__(256) void pw_mfma_kernel(PwArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int m_wave = (blockIdx.x * 4 + wave) * (MT * 16);
    if (m_wave >= a.M) return;

    const float* Wp = a.W;
    if (a.rows_per_crop > 0) Wp += (long)(m_wave / a.rows_per_crop) * a.w_crop_stride;   // per-crop weight matrices

    const float* xrow[MT];
    bool mvalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = m_wave + mt * 16 + li;
        mvalid[mt] = m < a.M;
        if (m >= a.M) m = a.M - 1;
        xrow[mt] = a.X + (long)m * a.ldx;
    }

    const int n_tiles = (a.N + 15) >> 4;
    // gridDim.y > 1 (small-batch plan): the passes over the output channel tiles are dealt to different workgroups
    for (int nc = blockIdx.y * NT; nc < n_tiles; nc += NT * gridDim.y) {
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        int nrow[NT];
        bool nvalid[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int n = (nc + nt) * 16 + li;
            nvalid[nt] = n < a.N;
            nrow[nt] = nvalid[nt] ? n : (a.N - 1);
        }

        for (int kg0 = 0; kg0 < a.K; kg0 += 16 * KU) {
            f32x4 xf[KU][MT], wf[KU][NT];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k = kg0 + u * 16 + lk * 4;
                const bool kvalid = k < a.K;   // K is a multiple of 4 (asserted on the host)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    xf[u][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (kvalid) xf[u][mt] = *reinterpret_cast<const f32x4*>(xrow[mt] + k);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wf[u][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (kvalid && nvalid[nt]) {
                        if (WKN) {
                            const float* p = Wp + (long)k * a.N + nrow[nt];
                            wf[u][nt] = (f32x4){p[0], p[a.N], p[2 * a.N], p[3 * a.N]};
                        } else {
                            wf[u][nt] = *reinterpret_cast<const f32x4*>(Wp + (long)nrow[nt] * a.K + k);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                if (KU > 1 && kg0 + u * 16 >= a.K) break;      // (zero operands: nothing to add)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][nt][i], xf[u][mt][i], acc[mt][nt], 0, 0, 0);
            }
        }

        // epilogue: lane holds channels n0 + 4*lk + {0..3} of pixel m0 + li
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = (nc + nt) * 16 + lk * 4;
            if (n >= a.N) continue;   // N is a multiple of 4
            f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (a.bias) b = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (!mvalid[mt]) continue;
                const long m = m_wave + mt * 16 + li;
                f32x4 v = acc[mt][nt] + b;
                if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
                if (a.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if (a.nchw_hw > 0) {
                    const long crop = m / a.nchw_hw, px = m % a.nchw_hw;
                    float* y = a.Y + (crop * a.N + n) * a.nchw_hw + px;
                    y[0] = v.x; y[a.nchw_hw] = v.y; y[2 * (long)a.nchw_hw] = v.z; y[3 * (long)a.nchw_hw] = v.w;
                } else {
                    *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise KSxKS convolution, stride S, pad KS/2, NHWC.  A thread owns 4 channels (one float4) of a
// vertical strip of RO output pixels, so consecutive lanes walk (channel-group, x): every global
// load/store instruction of a wavefront covers one contiguous 1 KiB run, and each input row is loaded
// once per strip and reused by all output rows it contributes to.
struct DwArgs {
    const float* X;   // [B][H][W][ldx]
    const float* Wt;  // [KS*KS][C]  (tap-major, channel-contiguous)
    const float* bias;  // [C] or nullptr
    float* Y;         // [B][Ho][Wo][ldy]
    int ldx, ldy;
    int B, H, W, C, Ho, Wo;
    int relu;
};

template <int KS, int S, int RO>
__global__ __launch_bounds__(256) void dw_conv_kernel(DwArgs a) {
    constexpr int P = KS / 2;
    constexpr int IR = (RO - 1) * S + KS;   // input rows touched by one strip
    const int cgs = a.C >> 2;
    const int strips = (a.Ho + RO - 1) / RO;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * strips * a.Wo * cgs;
    if (idx >= total) return;
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
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
#pragma unroll
    for (int r = 0; r < RO; ++r) acc[r] = b4;

    const int iy0 = oy0 * S - P;
    const int ix0 = ox * S - P;
    const long xbytes = (long)a.B * a.H * a.W * a.ldx * 4;
    if (xbytes < (1L << 31)) {
        // Branch-free taps: a tap outside the map is a buffer load with an out-of-range offset, which returns the zero padding
        // without a memory access.  With a branch around every load (the form below) the compiler emitted load, wait, FMA one
        // tap after the other — IR x KS dependent memory round trips per thread; the training step's depthwise forward ran at
        // 3.3 TB/s (profiles/r04_train_traffic_before.txt).  Same products added in the same order (a zero tap adds +0).
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)xbytes, 0x00020000);
        const int pix0 = ((b * a.H + iy0) * a.W + ix0) * a.ldx + c;          // may be "negative": only used when the tap is inside
#pragma unroll
        for (int iy = 0; iy < IR; ++iy) {
            const int y = iy0 + iy;
            const bool yin = y >= 0 && y < a.H;
            f32x4 v[KS];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int x = ix0 + kx;
                const bool in = yin && x >= 0 && x < a.W;
                v[kx] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, in ? (pix0 + (iy * a.W + kx) * a.ldx) * 4 : (int)0x80000000, 0, 0));
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                for (int r = 0; r < RO; ++r) {
                    const int ky = iy - r * S;          // compile-time after unrolling
                    if (ky >= 0 && ky < KS) acc[r] += v[kx] * w[ky * KS + kx];
                }
        }
    } else {
        const float* xb = a.X + (long)b * a.H * a.W * a.ldx + c;
#pragma unroll
        for (int iy = 0; iy < IR; ++iy) {
            const int y = iy0 + iy;
            if (y < 0 || y >= a.H) continue;
            const float* xr = xb + (long)y * a.W * a.ldx;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int x = ix0 + kx;
                if (x < 0 || x >= a.W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(xr + (long)x * a.ldx);
#pragma unroll
                for (int r = 0; r < RO; ++r) {
                    const int ky = iy - r * S;          // compile-time after unrolling
                    if (ky >= 0 && ky < KS) acc[r] += v * w[ky * KS + kx];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RO; ++r) {
        const int oy = oy0 + r;
        if (oy >= a.Ho) break;
        f32x4 v = acc[r];
        if (a.relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<f32x4*>(a.Y + (((long)b * a.Ho + oy) * a.Wo + ox) * a.ldy + c) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Stem: 3x3 stride-2 pad-1 convolution 3 -> 16 channels + ReLU.  Reads the caller's NCHW image,
// writes NHWC.  One thread per output pixel; the 432 weights are wave-uniform (scalar loads).
struct StemArgs {
    const float* X;   // [B][3][H][W]
    const float* Wt;  // [27][16]  ((ci*3+ky)*3+kx major, cout contiguous)
    const float* bias;  // [16]
    float* Y;         // [B][H/2][W/2][16]
    int B, H, W, Ho, Wo;
};

__global__ __launch_bounds__(256) void stem_conv_kernel(StemArgs a) {
    // weights + bias staged once per block; every lane reads the same LDS address (broadcast)
    __shared__ __attribute__((aligned(16))) float sw[27 * 16 + 16];
    for (int i = threadIdx.x; i < 27 * 16 + 16; i += 256) sw[i] = i < 27 * 16 ? a.Wt[i] : a.bias[i - 27 * 16];
    __syncthreads();
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * a.Ho * a.Wo;
    if (idx >= total) return;
    const int ox = idx % a.Wo;
    const int oy = (idx / a.Wo) % a.Ho;
    const int b = idx / ((long)a.Wo * a.Ho);
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = *reinterpret_cast<const f32x4*>(sw + 27 * 16 + 4 * q);
    const float* xb = a.X + (long)b * 3 * a.H * a.W;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = oy * 2 - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int x = ox * 2 - 1 + kx;
                float v = 0.f;
                if (y >= 0 && y < a.H && x >= 0 && x < a.W) v = xb[((long)ci * a.H + y) * a.W + x];
                const f32x4* wt = reinterpret_cast<const f32x4*>(sw + ((ci * 3 + ky) * 3 + kx) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += wt[q] * v;
            }
        }
    f32x4* y = reinterpret_cast<f32x4*>(a.Y + idx * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        y[q] = (f32x4){fmaxf(acc[q].x, 0.f), fmaxf(acc[q].y, 0.f), fmaxf(acc[q].z, 0.f), fmaxf(acc[q].w, 0.f)};
}

// ------------------------------------------------------------------------------------------------
// Stem on the matrix cores: the 3x3 stride-2 conv is an implicit GEMM with K = 27 (padded to 32):
// per 16-pixel row segment 8 MFMA 16x16x4 (A = weights [16][32], B = im2col gather straight from the
// caller's NCHW image), ReLU, one 16-byte NHWC store per lane.  One wavefront walks one output row.
struct StemMfmaArgs {
    const float* X;     // [B][3][H][W]
    const float* Wp;    // [16][32]  k = (ci*3+ky)*3+kx, zero padded 27..31
    const float* bias;  // [16]
    float* Y;           // [B][H/2][W/2][16]
    int B, H, W, Ho, Wo;
};

__global__ __launch_bounds__(256) void stem_mfma_kernel(StemMfmaArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long row = (long)blockIdx.x * 4 + wave;          // (crop, oy)
    if (row >= (long)a.B * a.Ho) return;
    const int oy = row % a.Ho;
    const long b = row / a.Ho;
    f32x4 wf[2];
    wf[0] = *reinterpret_cast<const f32x4*>(a.Wp + li * 32 + lk * 4);
    wf[1] = *reinterpret_cast<const f32x4*>(a.Wp + li * 32 + 16 + lk * 4);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + lk * 4);
    // the 8 taps this lane gathers: k = kg*16 + 4*lk + i
    int toff[8], tky[8], tkx[8];
    bool tval[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = (t >> 2) * 16 + lk * 4 + (t & 3);
        tval[t] = k < 27;
        const int kk = tval[t] ? k : 0;
        const int ci = kk / 9, ky = (kk % 9) / 3, kx = kk % 3;
        tky[t] = ky; tkx[t] = kx;
        toff[t] = (ci * a.H + ky) * a.W + kx;
    }
    const float* xb = a.X + b * 3 * a.H * a.W + (long)(2 * oy - 1) * a.W - 1;
    float* yb = a.Y + row * a.Wo * 16;
#pragma unroll 2
    for (int ox0 = 0; ox0 < a.Wo; ox0 += 16) {
        const int ox = ox0 + li;
        const float* xp = xb + 2 * ox;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bool ok = tval[t] && (2 * oy - 1 + tky[t] >= 0) && (2 * ox - 1 + tkx[t] >= 0);
            v[t] = ok ? xp[toff[t]] : 0.f;
        }
        f32x4 acc = bias;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t >> 2][t & 3], v[t], acc, 0, 0, 0);
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        *reinterpret_cast<f32x4*>(yb + (long)ox * 16 + lk * 4) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution with a handful of output channels (bbox_pred: 4 + exp, cls_pred: 1): 16 lanes
// cooperate on one pixel, each summing a K/16 slice with float4 loads, then a 4-step xor-shuffle
// reduction inside the 16-lane group.  Writes NCHW (the layout of the reference's output maps).
struct PwSmallArgs {
    const float* X;   // [M][ldx]
    const float* W;   // [N][K]
    const float* bias;  // [N]
    float* Y;         // [M/hw][N][hw]
    int ldx, M, K, N, hw;
    int act;          // 0 none, 2 exp
    long crop_stride; // floats between consecutive crops' maps in Y (N * hw for a dense tensor)
};

template <int N>
__global__ __launch_bounds__(256) void pw_small_kernel(PwSmallArgs a) {
    const int lane16 = threadIdx.x & 15;
    const long m = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool valid = m < a.M;
    const float* x = a.X + (valid ? m : 0) * a.ldx;
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    for (int k = lane16 * 4; k < a.K; k += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(a.W + (long)n * a.K + k);
            acc[n] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) acc[n] += __shfl_xor(acc[n], off, 16);
    }
    if (valid && lane16 == 0) {
        const long crop = m / a.hw, px = m % a.hw;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float v = acc[n] + a.bias[n];
            if (a.act == 2) v = expf(v);
            a.Y[crop * a.crop_stride + (long)n * a.hw + px] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FEARBoxCoder.decode with use_sigmoid=True: one wavefront per crop.  Scores are sigmoid(cls) in
// fp32 (ties after rounding resolve to the first cell, like torch.argmax); the box arithmetic runs
// in float64 against the float64 grid, as in the reference.
struct DecodeArgs {
    const float* cls;   // [n][1][S][S]
    const float* bbox;  // [n][4][S][S]
    int32_t* rc;        // [n][2]
    double* xywh;       // [n][4]
    float* score;       // [n]
    int n, S, stride, instance;
};

__global__ __launch_bounds__(64) void decode_kernel(DecodeArgs a) {
    const int crop = blockIdx.x;
    const int lane = threadIdx.x;
    const int cells = a.S * a.S;
    const float* c = a.cls + (long)crop * cells;
    float best = -1.f;
    int best_i = 0x7fffffff;
    for (int i = lane; i < cells; i += 64) {
        const float s = 1.f / (1.f + expf(-c[i]));
        if (s > best) { best = s; best_i = i; }   // strictly greater keeps the first index per lane
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(best_i, off, 64);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    if (lane == 0) {
        // a map of NaNs never satisfies `s > best`: stay inside the map (cell 0 = torch.argmax's answer for an all-NaN map)
        if ((unsigned)best_i >= (unsigned)cells) best_i = 0;
        const int r = best_i / a.S, col = best_i % a.S;
        const double gx = (double)(col - a.S / 2) * a.stride + a.instance / 2;
        const double gy = (double)(r - a.S / 2) * a.stride + a.instance / 2;
        const float* bb = a.bbox + (long)crop * 4 * cells + best_i;
        const double x0 = gx - (double)bb[0], y0 = gy - (double)bb[cells];
        const double x1 = gx + (double)bb[2 * cells], y1 = gy + (double)bb[3 * cells];
        a.rc[crop * 2] = r;
        a.rc[crop * 2 + 1] = col;
        a.xywh[crop * 4 + 0] = x0;
        a.xywh[crop * 4 + 1] = y0;
        a.xywh[crop * 4 + 2] = x1 - x0;
        a.xywh[crop * 4 + 3] = y1 - y0;
        a.score[crop] = best;
    }
}

// ------------------------------------------------------------------------------------------------
// Sum of the split-K partial projections + bias (+ residual) (+ ReLU):  Y[m][n] = sum_w P[w][m][n] + b[n] + R[m][n]
struct SplitKReduceArgs {
    const float* P;      // [W][M][ldp]
    const float* bias;   // [N]
    const float* R;      // [M][ldr] or nullptr
    float* Y;            // [M][ldy]
    long part_stride;    // floats between consecutive partials
    int W, M, N, ldp, ldr, ldy, relu;
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitKReduceArgs a) {
    const int q = a.N / 4;                                   // float4 per row
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.M * q) return;
    const long m = idx / q;
    const int n = (int)(idx - m * q) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.bias + n);
    // the partials are loaded a dozen at a time and added in order: a launch is a few thousand threads on an otherwise idle GPU,
    // and one load per trip made every one of the 12-24 partials a full memory round trip (7-9 us per launch at one crop)
    constexpr int UB = 12;
    const float* p0 = a.P + m * a.ldp + n;
    for (int w0 = 0; w0 < a.W; w0 += UB) {
        f32x4 pv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (w0 + u < a.W) pv[u] = *reinterpret_cast<const f32x4*>(p0 + (long)(w0 + u) * a.part_stride);
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (w0 + u < a.W) v += pv[u];
    }
    if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
}

// ------------------------------------------------------------------------------------------------
// Tracker._postprocess with tracking_config["smooth"] (base_tracker.py:149-205): scale / aspect-ratio change penalty
// against the previous size, cosine-window blend, first-maximum arg-max of the blended score, box decode at that cell and
// the penalty*score-weighted size smoothing.  float64 where the reference is float64 (grids, sizes, penalty, blended score
// — the box coder's grids are float64 tensors, so everything they touch is promoted), fp32 where it is fp32 (sigmoid of the
// logits; the three-factor learning rate is rounded to fp32 after each product like the reference's 0-dim fp32 tensor).
// One wave per crop.
struct DecodeSmoothArgs {
    const float* cls;        // [n][S*S] logits
    const float* bbox;       // [n][4][S*S] l, t, r, b
    const double* prev_size; // [n][2] previous (w, h) in search-crop pixels
    const double* window;    // [S*S] cosine window
    int32_t* rc;             // [n][2]
    double* xywh;            // [n][4]  x, y, smoothed w, smoothed h
    float* score;            // [n]     sigmoid(cls) at the arg-max cell
    int n, S, stride, instance;
    double penalty_k, window_influence, lr;
};

__device__ __forceinline__ double smooth_penalty(double w, double h, double pw, double ph, double penalty_k) {
    auto sq = [](double a, double b) { const double pad = (a + b) * 0.5; return sqrt((a + pad) * (b + pad)); };
    auto lim = [](double r) { return fmax(r, 1.0 / r); };
    const double s_c = lim(sq(w, h) / sq(pw, ph));
    const double r_c = lim((pw / ph) / (w / h));
    return exp(-(r_c * s_c - 1.0) * penalty_k);
}

__global__ __launch_bounds__(64) void decode_smooth_kernel(DecodeSmoothArgs a) {
    const int crop = blockIdx.x;
    const int lane = threadIdx.x;
    const int cells = a.S * a.S;
    const float* c = a.cls + (long)crop * cells;
    const float* bb = a.bbox + (long)crop * 4 * cells;
    const double pw = a.prev_size[crop * 2], ph = a.prev_size[crop * 2 + 1];
    double best = -1.0;
    int best_i = 0x7fffffff;
    for (int i = lane; i < cells; i += 64) {
        const int r = i / a.S, col = i % a.S;
        const double gx = (double)(col - a.S / 2) * a.stride + a.instance / 2;
        const double gy = (double)(r - a.S / 2) * a.stride + a.instance / 2;
        const double x0 = gx - (double)bb[i], y0 = gy - (double)bb[cells + i];
        const double x1 = gx + (double)bb[2 * cells + i], y1 = gy + (double)bb[3 * cells + i];
        const double pen = smooth_penalty(x1 - x0, y1 - y0, pw, ph, a.penalty_k);
        const float sg = 1.f / (1.f + expf(-c[i]));
        const double ps = pen * (double)sg * (1.0 - a.window_influence) + a.window[i] * a.window_influence;
        if (ps > best) { best = ps; best_i = i; }       // strictly greater keeps the first index per lane
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(best_i, off, 64);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    if (lane == 0) {
        // a map of NaNs never satisfies `s > best`: stay inside the map (cell 0 = torch.argmax's answer for an all-NaN map)
        if ((unsigned)best_i >= (unsigned)cells) best_i = 0;
        const int r = best_i / a.S, col = best_i % a.S;
        const double gx = (double)(col - a.S / 2) * a.stride + a.instance / 2;
        const double gy = (double)(r - a.S / 2) * a.stride + a.instance / 2;
        const double x0 = gx - (double)bb[best_i], y0 = gy - (double)bb[cells + best_i];
        const double x1 = gx + (double)bb[2 * cells + best_i], y1 = gy + (double)bb[3 * cells + best_i];
        const double w = x1 - x0, h = y1 - y0;
        const double pen = smooth_penalty(w, h, pw, ph, a.penalty_k);
        const float sg = 1.f / (1.f + expf(-c[best_i]));
        // lr = fp32(fp32(penalty) * fp32(score)) * fp32(lr), each product rounded to fp32 (base_tracker.py:159)
        const float lr32 = ((float)pen * sg) * (float)a.lr;
        const double lr = (double)lr32;
        // _smooth_size (base_tracker.py:126-139), evaluated as written there
        const double sw = w * lr, sh = h * lr, qw = pw * (1.0 - lr), qh = ph * (1.0 - lr);
        a.rc[crop * 2] = r;
        a.rc[crop * 2 + 1] = col;
        a.xywh[crop * 4 + 0] = x0;
        a.xywh[crop * 4 + 1] = y0;
        a.xywh[crop * 4 + 2] = qw + lr * (sw + qw);
        a.xywh[crop * 4 + 3] = qh + lr * (sh + qh);
        a.score[crop] = sg;
    }
}

// ------------------------------------------------------------------------------------------------
// uint8 HWC RGB -> normalised fp32 NCHW: (px - 255*mean) * (1/(255*std)), fp32 like albumentations.
struct NormArgs {
    const uint8_t* X;  // [n][hw][hw][3]
    float* Y;          // [n][3][hw][hw]
    long pixels;       // n*hw*hw
    int plane;         // hw*hw
    float mean[3], inv_std[3];
};

__global__ __launch_bounds__(256) void normalize_kernel(NormArgs a) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.pixels) return;
    const long crop = p / a.plane, px = p % a.plane;
    const uint8_t* s = a.X + p * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float v = (float)s[ch];
        v -= a.mean[ch];
        v *= a.inv_std[ch];
        a.Y[(crop * 3 + ch) * a.plane + px] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// get_extended_crop + normalise on device: context box -> constant border with the (saturate-
// cast) mean colour -> cv2.INTER_LINEAR uint8 resize (11-bit fixed point, half-pixel centres) -> ImageNet
// normalisation -> fp32 NCHW.  Reference: model_training/utils/utils.py:215-253 (cv2.copyMakeBorder + A.Resize)
// and tracker/base_tracker.py:70-103.  Bit-exact against the host restatement feartracker_amd/geometry.py and against the
// independent table-driven restatement of resize.cpp the tests carry, cv_ref.c (crop parity against real cv2 is unpinned in this image, see
// DESIGN.md).  One thread per output pixel.
struct CropArgs {
    const uint8_t* frame;   // [H][W][3] RGB
    const int* ctx;         // [n][4] context box x, y, w, h (frame coordinates, may leave the frame)
    const uint8_t* pad;     // [n][3] border colour (already saturate-cast)
    float* out;             // [n][3][S][S]
    int H, W, S, n;
    float mean[3], inv_std[3];
};

// One axis of OpenCV's 8u INTER_LINEAR coefficient table (resize.cpp), in its order of operations: scale = 1 / (dst / src) in
// double, the position rounded to float FIRST, floored, the fraction = the float difference; weights = cvRound(w * 2048).
// Columns (CLAMP) pin positions outside [0, src - 1] to the edge pixel with weights 2048 | 0; rows keep the table's weights and
// clip the two row INDICES (an edge row blended with itself — not the same number after the >> 16 truncations).  The
// double products are rounded separately (__dmul_rn / __dadd_rn: hipcc would otherwise contract them into one fma).
template <bool CLAMP>
__device__ __forceinline__ void linear_tap(int d, int dst, int src, int& i0, int& i1, int& w0, int& w1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);
    int idx = (int)floorf(f);
    f = __fsub_rn(f, (float)idx);
    if (CLAMP) {
        if (idx < 0) { f = 0.f; idx = 0; }
        if (idx >= src - 1) { f = 0.f; idx = src - 1; }
    }
    w0 = (int)rintf(__fmul_rn(__fsub_rn(1.0f, f), 2048.0f));
    w1 = (int)rintf(__fmul_rn(f, 2048.0f));
    i0 = min(max(idx, 0), src - 1);
    i1 = min(max(idx + 1, 0), src - 1);
}

__global__ __launch_bounds__(256) void crop_resize_normalize_kernel(CropArgs a) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int plane = a.S * a.S;
    if (p >= (long)a.n * plane) return;
    const int crop = p / plane, px = p % plane;
    const int dy = px / a.S, dx = px % a.S;
    const int cx = a.ctx[crop * 4], cy = a.ctx[crop * 4 + 1], cw = a.ctx[crop * 4 + 2], ch = a.ctx[crop * 4 + 3];
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    linear_tap<true>(dx, a.S, cw, x0, x1, ax0, ax1);
    linear_tap<false>(dy, a.S, ch, y0, y1, ay0, ay1);
    const bool same = (cw == a.S) && (ch == a.S);      // the reference's resize is the identity then
    const bool half = (cw == 2 * a.S) && (ch == 2 * a.S);   // cv::resize runs an exact 2x2 decimation as the 2x2 box mean
    auto sample = [&](int sx, int sy, int c) -> int {
        const int fx = cx + sx, fy = cy + sy;
        if (fx < 0 || fx >= a.W || fy < 0 || fy >= a.H) return a.pad[crop * 3 + c];
        return a.frame[((long)fy * a.W + fx) * 3 + c];
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v;
        if (same) {
            v = sample(dx, dy, c);
        } else if (half) {
            v = (sample(2 * dx, 2 * dy, c) + sample(2 * dx + 1, 2 * dy, c) + sample(2 * dx, 2 * dy + 1, c) +
                 sample(2 * dx + 1, 2 * dy + 1, c) + 2) >> 2;
        } else {
            const int r0 = sample(x0, y0, c) * ax0 + sample(x1, y0, c) * ax1;
            const int r1 = sample(x0, y1, c) * ax0 + sample(x1, y1, c) * ax1;
            v = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
        }
        float f = (float)v;
        f -= a.mean[c];
        f *= a.inv_std[c];
        a.out[((long)crop * 3 + c) * plane + px] = f;
    }
}

// ================================================================================================
// Fused block kernels.  (The first generation — ir16_fused_kernel / ir_tile_fused_kernel, weights loaded from
// global per chunk — was replaced by the v2 kernels below after the tools/kbench.hip ablation; see DESIGN.md.)
//
//   IR block (EXPAND):   y = [x +] P( relu( D( relu( E(x) ) ) ) )      E: 1x1 CIN->CEXP, D: depthwise KSxKS,
//   SepConv (!EXPAND):   y = act( P( D(x) ) )                           P: 1x1 CEXP->COUT
//
// The expanded tensor never leaves the CU: it is produced 16 channels at a time by MFMA (phase A) into a
// zero-ringed LDS tile, consumed by the depthwise conv on the VALU (phase B) whose float4 results are —
// by construction of the lane mapping — already the B-operand fragments of the projection MFMAs
// (phase C), which accumulate over the chunks in registers.
//
// Lane mapping (wave w, lane l): li = l&15 = pixel column x, lk = l>>4; MFMA 16x16x4 f32 fragments as in
// pw_mfma_kernel (weights = A operand, pixels = B operand).

// ================================================================================================
// ir16v2: one workgroup (8 waves) owns one crop's whole 16x16 map, so the depthwise halo is plain zero padding
// and nothing is recomputed.  Engineered around what the ablation (tools/kbench.hip) of the first generation
// showed — the per-chunk global weight loads (expand / depthwise / projection weights, identical for
// all 8 waves) were exposed latency, ~40 % of the kernel.  Now every weight a chunk needs is
//   * pre-packed on the host per chunk, MFMA fragments in lane order (one contiguous 1 KiB ds_read_b128
//     per fragment, conflict free),
//   * prefetched global -> registers at the START of a barrier interval and written to a double-buffered
//     LDS stage at its END (1-3 float4 per thread), so the HBM/L2 latency hides behind a whole interval of
//     MFMA + VALU work,
//   * read from LDS by every wave.
// CE = 16 channels per chunk (one k-group), E double buffered, A-part of the weights prefetched two
// chunks ahead, B/C-part one chunk ahead.  Packed layout per chunk (floats):
//   A-part  (EXPAND): KG fragments x 256 (lane l: We[c0 + l&15][kg*16 + 4*(l>>4) + 0..3]) | be[16]
//   BC-part          : NTP fragments x 256 (lane l: Wp[nt*16 + l&15][c0 + 4*(l>>4) + 0..3]) | Wd[KS*KS][16] | bd[16]
struct Ir2Args {
    const float* X;    // [B*256][ldx]
    const float* Wpk;  // packed per-chunk weights (layout above), chunk stride = AP + BP floats
    const float* bp;   // [COUT]
    const float* R;    // residual or nullptr
    float* Y;
    int ldx, ldr, ldy;
    int relu_dw, relu_out;
    // prediction heads (bbox_pred / cls_pred): COUT is padded to 16 in the kernel, only the first pred_cout channels
    // are real; they are written NCHW ([crop][pred_cout][256]) with optional exp (pred_act == 2)
    int pred_cout, pred_act;
    long pred_stride;     // floats between consecutive crops' maps (pred_cout * 256 for a dense NCHW tensor; 5 * 256 when the
                          // caller's bbox and cls maps are the two slices of one packed (n,5,16,16) tensor)
    // sep16_kernel<..., CORR = true>: per-crop template features z [crop][COUT][64] (the caller's NCHW (C, 8, 8) tensor);
    // the pixel-wise correlation of the block's output with z is written to channels [COUT, COUT + 64) of Y
    const float* Z;
    long z_stride;
    // sep16_kernel<..., PRED = true>: the prediction SepConv (dw KSxKS + 1x1 to pred_cout <= 4 channels [+ exp]) that consumes
    // this layer's output runs in the epilogue; P_Wpk = its packed weights (COUT/16 chunks x [1 fragment | Wd | bd]),
    // P_bp its bias, P_Y the caller's NCHW map.  The layer's own output is then not written at all.
    const float* P_Wpk;
    const float* P_bp;
    float* P_Y;
    // split-K variants (small batches): every crop is handled by gridDim.y workgroups, workgroup y taking the kc_count
    // 16-channel chunks from chunk y * kc_count on and writing its RAW partial projection (no bias / residual / activation)
    // to Y + y * kc_part_stride; splitk_reduce_kernel adds the partials up
    int kc_count;
    long kc_part_stride;
    // N-split (sep16_kernel<CIN, 16, KS> at very small batches): the layer's COUT_total = 16 * gridDim.y output channels are
    // cut into 16-channel slices, workgroup y runs the whole depthwise and projects onto slice y — finished outputs (bias,
    // ReLU), no partials and no reduce launch; the depthwise is recomputed per slice, which costs nothing when the GPU is
    // otherwise idle.  nsplit_wstride = floats between the packed weight sets of two slices (0 = not an N-split launch).
    long nsplit_wstride;
};

template <int CIN, int CEXP, int COUT, int KS, bool EXPAND>
struct Ir2Geom {
    // ES = 24 floats (6 x 16 B) per pixel: with lane stride = one pixel and lk stride = one 16-B unit every
    // 16-lane service group of ds_read_b128 hits 16 distinct 16-B slots (conflict free; ES = 20 was 2-way)
    static constexpr int CE = 16, S = 16, P = KS / 2, PW = S + 2 * P, ES = 24;
    static constexpr int NCHUNK = CEXP / CE, NTP = COUT / 16, KG = EXPAND ? CIN / 16 : 0;
    static constexpr int AP = EXPAND ? KG * 256 + 16 : 0;
    static constexpr int BP = NTP * 256 + KS * KS * 16 + 16;
    static constexpr int EBUF = PW * PW * ES;
    static constexpr int LDS_FLOATS = 2 * EBUF + 2 * AP + 2 * BP;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// Asynchronous global -> LDS copy of NF contiguous floats (a packed weight block) by the 8 waves of a workgroup, 1 KiB per
// wave instruction (global_load_lds_dwordx4: lane l's 16 bytes land at dst + 16*l; no VGPRs, no ds_write).  The copy is
// complete for the issuing wave after s_waitcnt vmcnt(0) — __syncthreads() includes it — and visible to the others after
// the barrier.  dst must not be read by anyone during the interval the copy is in flight.
template <int NF>
__device__ __forceinline__ void lds_copy_async(const float* __restrict__ src, float* dst, int wave, int lane) {
    static_assert(NF % 4 == 0, "16-byte granules");
    constexpr int NK = NF / 256, REM = NF % 256;
#pragma unroll
    for (int k = 0; k < (NK + 7) / 8; ++k) {
        const int piece = wave + 8 * k;
        if (piece < NK)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + piece * 256), 16, 0, 0);
    }
    if (REM > 0 && wave == NK % 8 && lane * 4 < REM)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + NK * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dst + NK * 256), 16, 0, 0);
}

// d += a * b as two v_pk_fma_f32.  Written as inline asm because hipcc's post-RA peephole "unpacks" packed fp32 FMAs that
// follow an MFMA into two v_fma_f32 (it assumes they hide in the MFMA's shadow; fp32 MFMAs occupy the same vector ALU on
// gfx950 — tools/coexec.hip — so the unpacked pair simply costs twice the issue cycles: 4.3 vs 2.5 cycles per FMA pair).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pk_fma4(f32x4& d, const f32x4& a, const f32x4& b) {
    f32x2 dl = __builtin_shufflevector(d, d, 0, 1), dh = __builtin_shufflevector(d, d, 2, 3);
    const f32x2 al = __builtin_shufflevector(a, a, 0, 1), ah = __builtin_shufflevector(a, a, 2, 3);
    const f32x2 bl = __builtin_shufflevector(b, b, 0, 1), bh = __builtin_shufflevector(b, b, 2, 3);
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dl) : "v"(al), "v"(bl));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dh) : "v"(ah), "v"(bh));
    d = __builtin_shufflevector(dl, dh, 0, 1, 2, 3);
}

// hipcc's hazard recognizer does not look inside inline asm: an MFMA whose A/B operand is a VGPR written by the v_pk_fma_f32
// right in front of it gets NO wait states and reads a stale value (found with tools/sepcheck.hip: the last tap of the
// prediction head's last channel chunk was dropped from one output row — 5e-3 relative — as soon as a re-layout of the LDS
// tile let the scheduler place the chain's last FMA directly before the MFMAs).  Call this after the last pk_fma4 of a chain
// whose results feed MFMAs: it pins eight wait states between the two (once per chunk: noise).
__device__ __forceinline__ void pk_fma_settle(f32x4& a, f32x4& b) { asm volatile("s_nop 7" : "+v"(a), "+v"(b)); }

// One barrier interval of the fused 16x16 block kernels (ir16v2_fused_kernel, chain16_block): depthwise + projection of
// chunk c from E / wb, and (HAS_A) the expansion of chunk c + 1 from wa into En.
// The depthwise is a chain of KS*(KS+1) tap steps (kx outer, iy inner, so the weight of (iy, kx) feeds row 0 now and
// row 1 in the next step); its LDS reads run D steps ahead of the FMAs that consume them and the expansion MFMAs are
// dealt out between the steps, so the wave never sits on an LDS round trip (hipcc's own order was read -> wait -> 4 FMAs,
// ~120 idle cycles per step, in lockstep on every wave).  sched_barrier(0) after each step keeps the scheduler from
// re-serialising the pipeline.
template <int KS, int PW, int ES, int KG, int NTP, bool HAS_A, int XM, int XK>
__device__ __forceinline__ void ir16_interval(const float* __restrict__ E, float* __restrict__ En, const float* __restrict__ wa,
                                              const float* __restrict__ wb, const f32x4 (&xf)[XM][XK], f32x4 (&accp)[2][NTP],
                                              int y0, int li, int lk, int lane, bool relu_dw) {
    // GS tap steps form one scheduling group: [LDS reads for the group D steps ahead][the group's MFMAs][the group's packed
    // FMAs] — switching the vector ALU between MFMA and VALU instructions costs ~8 cycles (tools/coexec.hip modes 7/9), so
    // the groups are made as coarse as the read-ahead window allows
    constexpr int NS = KS * (KS + 1), GS = IR16_GS, D = IR16_D, P = KS / 2;
    constexpr int NU = HAS_A ? KG * 4 : 0;                     // expansion MFMA units (each = mt 0 and mt 1)
    static_assert(!HAS_A || (XM == 2 && XK >= KG), "fragment array");
    static_assert(NS % GS == 0 && D >= GS && D % GS == 0, "step grouping");
    const float* wd = wb + NTP * 256 + lk * 4;
    const float* e0 = E + (y0 * PW + li) * ES + lk * 4;
    f32x4 acc[2];
    f32x4 wfq[2];
    if (HAS_A) {
        acc[0] = acc[1] = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);   // bias
        wfq[0] = *reinterpret_cast<const f32x4*>(wa + lane * 4);
    }
    f32x4 d0 = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
    f32x4 d1 = d0;
    f32x4 ev[D], wv[D], wprev = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const int kx = t / (KS + 1), iy = t % (KS + 1);
        ev[t] = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * ES);
        if (iy < KS) wv[t] = *reinterpret_cast<const f32x4*>(wd + (iy * KS + kx) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NS; g += GS) {
        f32x4 e[GS], w[GS];
#pragma unroll
        for (int s0 = 0; s0 < GS; ++s0) {
            const int t = g + s0;
            e[s0] = ev[t % D];
            w[s0] = wv[t % D];
            if (t + D < NS && !(FEAR_ABL & 4)) {
                const int kx2 = (t + D) / (KS + 1), iy2 = (t + D) % (KS + 1);
                ev[t % D] = *reinterpret_cast<const f32x4*>(e0 + (iy2 * PW + kx2) * ES);
                if (iy2 < KS) wv[t % D] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 16);
            }
        }
        if (HAS_A) {
#pragma unroll
            for (int u = g * NU / NS; u < (g + GS) * NU / NS; ++u) {
                const int kg = u / 4, i = u % 4;
                if (i == 0 && kg + 1 < KG) wfq[(kg + 1) & 1] = *reinterpret_cast<const f32x4*>(wa + (kg + 1) * 256 + lane * 4);
                if (FEAR_ABL & 8) continue;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], xf[0][HAS_A ? kg : 0][i], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], xf[HAS_A ? 1 : 0][HAS_A ? kg : 0][i], acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s0 = 0; s0 < GS; ++s0) {
            const int iy = (g + s0) % (KS + 1);
            if (FEAR_ABL & 32) {
                d0.x += e[s0].x + w[s0].x;
            } else {
                if (iy < KS) pk_fma4(d0, e[s0], w[s0]);
                if (iy >= 1) pk_fma4(d1, e[s0], wprev);
            }
            wprev = w[s0];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    pk_fma_settle(d0, d1);
    f32x4 wpq[2];
    wpq[0] = *reinterpret_cast<const f32x4*>(wb + lane * 4);
    if (HAS_A) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 v = acc[mt];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4*>(En + ((y0 + mt + P) * PW + li + P) * ES + lk * 4) = v;
        }
    }
    if (relu_dw) {
        d0.x = fmaxf(d0.x, 0.f); d0.y = fmaxf(d0.y, 0.f); d0.z = fmaxf(d0.z, 0.f); d0.w = fmaxf(d0.w, 0.f);
        d1.x = fmaxf(d1.x, 0.f); d1.y = fmaxf(d1.y, 0.f); d1.z = fmaxf(d1.z, 0.f); d1.w = fmaxf(d1.w, 0.f);
    }
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        if (nt + 1 < NTP) wpq[(nt + 1) & 1] = *reinterpret_cast<const f32x4*>(wb + (nt + 1) * 256 + lane * 4);
        if (FEAR_ABL & 16) { accp[0][nt] += d0 * wpq[nt & 1]; accp[1][nt] += d1 * wpq[nt & 1]; continue; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            accp[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d0[i], accp[0][nt], 0, 0, 0);
            accp[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d1[i], accp[1][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}


template <int CIN, int CEXP, int COUT, int KS, bool EXPAND, bool SPLITK = false>
__global__ __launch_bounds__(512) void ir16v2_fused_kernel(Ir2Args a) {
    using G = Ir2Geom<CIN, CEXP, COUT, KS, EXPAND>;
    constexpr int S = G::S, P = G::P, PW = G::PW, ES = G::ES, NTP = G::NTP, KG = G::KG;
    // split-K: this workgroup's chunk range (run-time); otherwise the whole expansion
    const int NCHUNK = SPLITK ? a.kc_count : G::NCHUNK;
    const float* const Wpk = SPLITK ? a.Wpk + (long)blockIdx.y * a.kc_count * (G::AP + G::BP) : a.Wpk;
    constexpr int AP = G::AP, BP = G::BP, EBUF = G::EBUF, CST = AP + BP;
    constexpr int AP4 = AP / 4, BP4 = BP / 4;                 // float4 counts
    constexpr int NRA = (AP4 + 511) / 512, NRB = (BP4 + 511) / 512;
    static_assert(CEXP % 16 == 0 && COUT % 16 == 0 && (!EXPAND || CIN % 16 == 0), "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Ebuf = lds;                    // [2][EBUF]
    float* const WA = lds + 2 * EBUF;           // [2][AP]
    float* const WB = WA + 2 * AP;              // [2][BP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long crop = blockIdx.x;
    const float* Xc = a.X + crop * 256 * a.ldx;
    const int y0 = wave * 2;

    for (int i = tid * 4; i < 2 * EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 xf[EXPAND ? 2 : 1][EXPAND ? KG : 1];
    if (EXPAND) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)
                xf[mt][kg] = *reinterpret_cast<const f32x4*>(Xc + (long)((y0 + mt) * S + li) * a.ldx + kg * 16 + lk * 4);
    }

    // ---- staging helpers: global -> regs (issue early), regs -> LDS (commit late)
    f32x4 ra[EXPAND ? NRA : 2], rb[NRB];
    auto load_a = [&](int c) {      // EXPAND: A-part of chunk c; !EXPAND: the X channels of chunk c for this lane's 2 pixels
        if (EXPAND) {
#pragma unroll
            for (int r = 0; r < NRA; ++r) {
                const int idx = tid + r * 512;
                if (idx < AP4) ra[r] = *reinterpret_cast<const f32x4*>(Wpk + (long)c * CST + idx * 4);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                ra[mt] = *reinterpret_cast<const f32x4*>(Xc + (long)((y0 + mt) * S + li) * a.ldx + c * 16 + lk * 4);
        }
    };
    auto store_a = [&](int c) {
        if (EXPAND) {
            float* dst = WA + (c & 1) * AP;
#pragma unroll
            for (int r = 0; r < NRA; ++r) {
                const int idx = tid + r * 512;
                if (idx < AP4) *reinterpret_cast<f32x4*>(dst + idx * 4) = ra[r];
            }
        } else {
            float* E = Ebuf + (c & 1) * EBUF;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                *reinterpret_cast<f32x4*>(E + ((y0 + mt + P) * PW + li + P) * ES + lk * 4) = ra[mt];
        }
    };
    auto load_b = [&](int c) {
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            const int idx = tid + r * 512;
            if (idx < BP4) rb[r] = *reinterpret_cast<const f32x4*>(Wpk + (long)c * CST + AP + idx * 4);
        }
    };
    auto store_b = [&](int c) {
        float* dst = WB + (c & 1) * BP;
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            const int idx = tid + r * 512;
            if (idx < BP4) *reinterpret_cast<f32x4*>(dst + idx * 4) = rb[r];
        }
    };

    // phase A (EXPAND): E[c] <- relu(We_chunk . x + be) on the matrix cores, weights from LDS stage c&1
    auto phase_a = [&](int c) {
        const float* wa = WA + (c & 1) * AP;
        float* E = Ebuf + (c & 1) * EBUF;
        f32x4 acc[2];
        acc[0] = acc[1] = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);   // bias
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const f32x4 wf = *reinterpret_cast<const f32x4*>(wa + kg * 256 + lane * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], xf[mt][kg][i], acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 v = acc[mt];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4*>(E + ((y0 + mt + P) * PW + li + P) * ES + lk * 4) = v;
        }
    };

    f32x4 accp[2][NTP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) accp[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stage A(0), A(1), BC(0); produce E[0].  A(1) is fetched together with A(0) into registers of its own: one
    //      memory round trip before the first MFMA instead of two (a split-K workgroup at one crop runs 1-2 chunks in all)
    f32x4 ra1[EXPAND ? NRA : 1];
    load_a(0);
    load_b(0);
    if (EXPAND && NCHUNK > 1) {
#pragma unroll
        for (int r = 0; r < NRA; ++r) {
            const int idx = tid + r * 512;
            if (idx < AP4) ra1[r] = *reinterpret_cast<const f32x4*>(Wpk + (long)CST + idx * 4);
        }
    }
    __syncthreads();                       // zero fill done before the first E / stage writes
    store_a(0);
    store_b(0);
    if (EXPAND) {
        if (NCHUNK > 1) {
#pragma unroll
            for (int r = 0; r < NRA; ++r) {
                const int idx = tid + r * 512;
                if (idx < AP4) *reinterpret_cast<f32x4*>(WA + AP + idx * 4) = ra1[r];
            }
        }
        __syncthreads();
        phase_a(0);
    }
    __syncthreads();

    // (the last chunk — nothing left to expand — is peeled, as in chain16_block: no run-time branch around two instantiations of
    //  the interval inside the loop)
    auto chunk = [&](int c, auto more_c) {
        constexpr bool MORE = decltype(more_c)::value;
        // prefetch (registers only): EXPAND: A-part two chunks ahead; !EXPAND: next chunk's activations
        const int ca = EXPAND ? c + 2 : c + 1;
        if (!(FEAR_ABL & 2)) {
            if (ca < NCHUNK) load_a(ca);
            if (MORE) load_b(c + 1);
        }
        // (fp32 MFMA executes on the vector ALUs on gfx950 — tools/coexec.hip: an MFMA wave and a VALU wave on one
        //  SIMD take the SUM of their times — so staggering phases between waves buys nothing; what matters is that
        //  neither wave of a SIMD waits on LDS latency)
        const float* Ec = Ebuf + (c & 1) * EBUF;
        float* En = Ebuf + ((c + 1) & 1) * EBUF;
        const float* wa = WA + ((c + 1) & 1) * AP;
        const float* wb = WB + (c & 1) * BP;
        ir16_interval<KS, PW, ES, KG, NTP, EXPAND && MORE>(Ec, En, wa, wb, xf, accp, y0, li, lk, lane, a.relu_dw);
        if (!(FEAR_ABL & 2)) {
            if (ca < NCHUNK) store_a(ca);
            if (MORE) store_b(c + 1);
        }
        if (!(FEAR_ABL & 1)) __syncthreads();
    };
    for (int c = 0; c < NCHUNK - 1; ++c) chunk(c, std::true_type{});
    chunk(NCHUNK - 1, std::false_type{});

    if (SPLITK) {                   // raw partial sums; bias, residual and activation belong to splitk_reduce_kernel
        float* Yp = a.Y + (long)blockIdx.y * a.kc_part_stride;
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const long m = crop * 256 + (y0 + mt) * S + li;
                *reinterpret_cast<f32x4*>(Yp + m * a.ldy + nt * 16 + lk * 4) = accp[mt][nt];
            }
        return;
    }
    if (a.pred_cout > 0) {          // prediction head: lanes lk == 0 hold channels 0..3 of their pixel
        if (lk == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int px = (y0 + mt) * S + li;
                const f32x4 v = accp[mt][0];
                const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (n < a.pred_cout) {
                        float o = vals[n] + a.bp[n];
                        if (a.pred_act == 2) o = expf(o);
                        a.Y[crop * a.pred_stride + n * 256 + px] = o;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        const int n = nt * 16 + lk * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bp + n);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long m = crop * 256 + (y0 + mt) * S + li;
            f32x4 v = accp[mt][nt] + b;
            if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
            if (a.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
        }
    }
}

// ================================================================================================
// Tiled fused block kernel, v2: ir_tile_fused_kernel with the ir16v2 weight path — per-chunk host-packed
// fragments (zero padded where CIN / COUT / CEXP are not multiples of 16), prefetched global -> registers
// during chunk c and committed to the alternate LDS stage before the chunk's last barrier.  CE = 16,
// single E tile (so several workgroups fit a CU and overlap each other's phases and HBM traffic).
// ================================================================================================
// sep16: depthwise-separable conv (dw KSxKS + 1x1, the BoxTower's SepConv) on a 16x16 map, one crop per block — the
// no-expansion sibling of ir16v2 with the depthwise of chunk c + 1 software-pipelined UNDER the projection MFMAs of chunk c:
// the wave's instruction stream per tap step is [LDS reads 4 steps ahead][~NTP*8/NS projection MFMAs][4 packed FMAs], so
// neither the LDS round trips nor the FMAs leave the vector ALUs idle between MFMAs.
//   interval c reads  E[(c+1)&1], WD[(c+1)&1]  (depthwise of chunk c+1)  and  WP[c&1] (projection of chunk c)
//   interval c writes E[c&1] <- X chunk c+2, WD[c&1] <- wd(c+2), WP[(c+1)&1] <- wp(c+1)   (all last read before barrier c-1)
// Packed weights: the BC-part layout of Ir2Args (AP = 0): per chunk NTP fragments x 256 | Wd[KS*KS][16] | bd[16].
template <int CIN, int COUT, int KS, bool CORR = false>
struct Sep16Geom {
    static constexpr int S = 16, P = KS / 2, PW = S + 2 * P, NCHUNK = CIN / 16, NTP = COUT / 16;
    static constexpr int WPF = NTP * 256, WDF = KS * KS * 16 + 16, CST = WPF + WDF;
    // LDS tile of one 16-channel chunk: four planes (one per channel quad) of [pixel][4], a multiple of 64 floats apart —
    // conflict free for the stride-1 ds_read_b128 pattern and for the stores, nothing padded (see IrT2Geom).  16 instead of
    // 24 floats per pixel brings the kernel from 96 to 77 KB of LDS: two workgroups per CU when the two head branches run
    // side by side.
    static constexpr int EP = 4, EQ = (PW * PW * 4 + 63) / 64 * 64;
    static constexpr int EBUF = 4 * EQ;
    static constexpr int ZF = CORR ? COUT * 64 : 0;            // the crop's template features, resident for the epilogue
    static constexpr int LDS_BYTES = (2 * EBUF + 2 * WPF + 2 * WDF + ZF) * 4;
};

// CORR = true appends the pixel-wise correlation with the crop's template features (MobileCorrelation, blocks.py:121-123):
// corr[px][t] = sum_c y[px][c] * z[c][t] — the block's output fragments are the B operand as they stand, z (64 KiB,
// fetched into LDS by an asynchronous copy at kernel start) supplies the A fragments; the 64 correlation channels go to
// Y[.., COUT .. COUT+64), i.e. next to the features in the concat buffer the following SepConv reads.
// PRED = true: the prediction head that follows a tower (SepConv to <= 4 channels) is computed in the epilogue, chunk by
// chunk: the finished output fragments of 16 channels go to an LDS tile (never to HBM), the head's depthwise runs from it
// and its one-tile projection accumulates over the COUT/16 chunks.
// SPLITK = k > 0 (small batches): gridDim.y workgroups per crop, workgroup y handling the k input chunks from chunk y*k on
// and writing its raw partial projection (see Ir2Args::kc_count); bias, ReLU and the fused epilogues are not available.
template <int CIN, int COUT, int KS, bool CORR = false, bool PRED = false, int SPLITK = 0>
__global__ __launch_bounds__(512) void sep16_kernel(Ir2Args a) {
    using G = Sep16Geom<CIN, COUT, KS, CORR>;
    static_assert(!(CORR && PRED) && !(SPLITK && (CORR || PRED)), "one epilogue at a time");
    static_assert(SPLITK == 0 || (SPLITK >= 2 && G::NCHUNK % SPLITK == 0), "SPLITK = chunks per workgroup (compile time)");
    constexpr int S = G::S, P = G::P, PW = G::PW, EP = G::EP, EQ = G::EQ, NTP = G::NTP;
    constexpr int NCHUNK = SPLITK ? SPLITK : G::NCHUNK;         // split-K: this workgroup's share of the input chunks
    const int c_base = SPLITK ? (int)blockIdx.y * SPLITK : 0;
    constexpr int WPF = G::WPF, WDF = G::WDF, CST = G::CST, EBUF = G::EBUF;
    constexpr int WP4 = WPF / 4, WD4 = WDF / 4, NRP = (WP4 + 511) / 512;
    constexpr int NS = KS * (KS + 1), D = 4, NU = NTP * 4;
    static_assert(CIN % 16 == 0 && COUT % 16 == 0 && G::NCHUNK >= 2 && WD4 <= 512 && NS >= D, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Ebuf = lds;                    // [2][EBUF]
    float* const WP = lds + 2 * EBUF;           // [2][WPF]
    float* const WD = WP + 2 * WPF;             // [2][WDF]
    float* const ZL = WD + 2 * WDF;             // [COUT][64] (CORR)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long crop = blockIdx.x;
    const float* Xc = a.X + crop * 256 * a.ldx;
    const int y0 = wave * 2;
    if (!SPLITK && !CORR && !PRED && a.nsplit_wstride) {      // N-split: this workgroup's 16-channel output slice
        a.Wpk += (long)blockIdx.y * a.nsplit_wstride;
        a.bp += blockIdx.y * COUT;
        a.Y += blockIdx.y * COUT;
    }

    for (int i = tid * 4; i < 2 * EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (CORR) lds_copy_async<G::ZF>(a.Z + crop * a.z_stride, ZL, wave, lane);

    f32x4 rx[2];
    auto load_x = [&](int c) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            rx[mt] = *reinterpret_cast<const f32x4*>(Xc + (long)((y0 + mt) * S + li) * a.ldx + (c_base + c) * 16 + lk * 4);
    };
    auto store_x = [&](int c) {
        float* E = Ebuf + (c & 1) * EBUF;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4*>(E + ((y0 + mt + P) * PW + li + P) * EP + lk * EQ) = rx[mt];
    };
    // weights go global -> LDS directly (asynchronous, no registers): the projection fragments of chunk c into WP[c & 1],
    // the depthwise taps + bias into WD[c & 1]
    auto stage_p = [&](int c) { lds_copy_async<WPF>(a.Wpk + (long)(c_base + c) * CST, WP + (c & 1) * WPF, wave, lane); };
    auto stage_d = [&](int c) { lds_copy_async<WDF>(a.Wpk + (long)(c_base + c) * CST + WPF, WD + (c & 1) * WDF, wave, lane); };

    f32x4 accp[2][NTP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) accp[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // depthwise of chunk cd (reads run D tap steps ahead) with the projection MFMAs of the previous chunk dealt between
    // the steps; PROJ = false for the prologue (nothing to project yet), DW = false for the last interval
    f32x4 d0, d1;                       // depthwise result of the chunk being projected
    auto interval = [&](int c, auto proj_tag, auto dw_tag) {
        constexpr bool PROJ = decltype(proj_tag)::value && !(FEAR_ABL & 16), DW = decltype(dw_tag)::value && !(FEAR_ABL & 64);
        if (FEAR_ABL & 64) { d0 = d1 = rx[0]; }
        const int cd = PROJ ? c + 1 : c;
        const float* wd = WD + (cd & 1) * WDF + lk * 4;
        const float* e0 = Ebuf + (cd & 1) * EBUF + (y0 * PW + li) * EP + lk * EQ;
        const float* wp = WP + (c & 1) * WPF;
        f32x4 n0, n1, ev[D], wv[D], wprev = (f32x4){0.f, 0.f, 0.f, 0.f}, wpq[2];
        if (PROJ) wpq[0] = *reinterpret_cast<const f32x4*>(wp + lane * 4);
        if (DW) {
            n0 = n1 = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
#pragma unroll
            for (int t = 0; t < D; ++t) {
                const int kx = t / (KS + 1), iy = t % (KS + 1);
                ev[t] = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * EP);
                if (iy < KS) wv[t] = *reinterpret_cast<const f32x4*>(wd + (iy * KS + kx) * 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int iy = t % (KS + 1);
            f32x4 e, w;
            if (DW) {
                e = ev[t % D]; w = wv[t % D];
                if (t + D < NS) {
                    const int kx2 = (t + D) / (KS + 1), iy2 = (t + D) % (KS + 1);
                    ev[t % D] = *reinterpret_cast<const f32x4*>(e0 + (iy2 * PW + kx2) * EP);
                    if (iy2 < KS) wv[t % D] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 16);
                }
            }
            if (PROJ) {
#pragma unroll
                for (int u = t * NU / NS; u < (t + 1) * NU / NS; ++u) {
                    const int nt = u / 4, i = u % 4;
                    if (i == 0 && nt + 1 < NTP) wpq[(nt + 1) & 1] = *reinterpret_cast<const f32x4*>(wp + (nt + 1) * 256 + lane * 4);
                    accp[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d0[i], accp[0][nt], 0, 0, 0);
                    accp[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d1[i], accp[1][nt], 0, 0, 0);
                }
            }
            if (DW) {
                if (iy < KS) pk_fma4(n0, e, w);
                if (iy >= 1) pk_fma4(n1, e, wprev);
                wprev = w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DW) {
            pk_fma_settle(n0, n1);
            if (a.relu_dw) {
                n0.x = fmaxf(n0.x, 0.f); n0.y = fmaxf(n0.y, 0.f); n0.z = fmaxf(n0.z, 0.f); n0.w = fmaxf(n0.w, 0.f);
                n1.x = fmaxf(n1.x, 0.f); n1.y = fmaxf(n1.y, 0.f); n1.z = fmaxf(n1.z, 0.f); n1.w = fmaxf(n1.w, 0.f);
            }
            d0 = n0;
            d1 = n1;
        }
    };

    // ---- prologue: E[0], E[1], WD[0], WD[1], WP[0]; depthwise of chunk 0
    load_x(0);
    __syncthreads();                       // zero fill done
    stage_d(0);
    stage_p(0);
    stage_d(1);
    store_x(0);
    load_x(1);
    store_x(1);
    __syncthreads();
    interval(0, std::false_type{}, std::true_type{});
    __syncthreads();                       // E[0] / WD[0] are overwritten at the end of interval 0

    for (int c = 0; c < ((FEAR_ABL & 256) ? 0 : NCHUNK); ++c) {
        if (!(FEAR_ABL & 2)) {
            // in flight during this interval: WD[c&1] (last read by the depthwise of chunk c, before the previous barrier),
            // WP[(c+1)&1] (last read by the projection of chunk c-1)
            if (c + 2 < NCHUNK) { load_x(c + 2); stage_d(c + 2); }
            if (c + 1 < NCHUNK) stage_p(c + 1);
        }
        auto commit = [&] {
            if (FEAR_ABL & 2) return;
            if (c + 2 < NCHUNK) store_x(c + 2);
        };
        if (c + 1 < NCHUNK) interval(c, std::true_type{}, std::true_type{});
        else interval(c, std::true_type{}, std::false_type{});
        commit();      // (committing mid-interval stalls on the global loads: they need most of an interval to land)
        if (!(FEAR_ABL & 1)) __syncthreads();
    }

    if (SPLITK) {                   // raw partial sums; bias and ReLU belong to splitk_reduce_kernel
        float* Yp = a.Y + (long)blockIdx.y * a.kc_part_stride;
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const long m = crop * 256 + (y0 + mt) * S + li;
                *reinterpret_cast<f32x4*>(Yp + m * a.ldy + nt * 16 + lk * 4) = accp[mt][nt];
            }
        return;
    }
    if (PRED) {
        constexpr int PCH = 256 + KS * KS * 16 + 16;          // packed floats per chunk of the head: 1 fragment | Wd | bd
        static_assert(NTP * PCH <= 2 * WPF, "the head's weights are staged in the projection-weight area");
        lds_copy_async<NTP * PCH>(a.P_Wpk, WP, wave, lane);   // WP / WD are free: the main loop ended with a barrier
        f32x4 pacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < NTP; ++c) {         // fully unrolled: accp[..][c] must be a compile-time register reference
            // this layer's output channels [16c, 16c+16) -> E[c & 1] (last read two iterations ago, one barrier in between)
            float* E = Ebuf + (c & 1) * EBUF;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bp + c * 16 + lk * 4);
            f32x4 v0 = accp[0][c] + bv, v1 = accp[1][c] + bv;
            if (a.relu_out) {
                v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
            }
            *reinterpret_cast<f32x4*>(E + ((y0 + P) * PW + li + P) * EP + lk * EQ) = v0;
            *reinterpret_cast<f32x4*>(E + ((y0 + 1 + P) * PW + li + P) * EP + lk * EQ) = v1;
            __syncthreads();                                   // (first pass: also completes the weight copy)
            const float* wpk = WP + c * PCH;
            const float* wd = wpk + 256 + lk * 4;
            const float* e0 = E + (y0 * PW + li) * EP + lk * EQ;
            f32x4 n0 = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16), n1 = n0, wprev = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 ev[D], wv[D];
#pragma unroll
            for (int t = 0; t < D; ++t) {
                const int kx = t / (KS + 1), iy = t % (KS + 1);
                ev[t] = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * EP);
                if (iy < KS) wv[t] = *reinterpret_cast<const f32x4*>(wd + (iy * KS + kx) * 16);
            }
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int iy = t % (KS + 1);
                const f32x4 e = ev[t % D], w = wv[t % D];
                if (t + D < NS) {
                    const int kx2 = (t + D) / (KS + 1), iy2 = (t + D) % (KS + 1);
                    ev[t % D] = *reinterpret_cast<const f32x4*>(e0 + (iy2 * PW + kx2) * EP);
                    if (iy2 < KS) wv[t % D] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 16);
                }
                if (iy < KS) pk_fma4(n0, e, w);
                if (iy >= 1) pk_fma4(n1, e, wprev);
                wprev = w;
            }
            // (no activation between the head's depthwise and its 1x1: SepConv = dw -> pw, as in the towers)
            pk_fma_settle(n0, n1);
            const f32x4 wf = *reinterpret_cast<const f32x4*>(wpk + lane * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], n0[i], pacc[0], 0, 0, 0);
                pacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], n1[i], pacc[1], 0, 0, 0);
            }
        }
        if (lk == 0) {              // lanes lk == 0 hold channels 0..3 of their pixel
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int px = (y0 + mt) * S + li;
                const float vals[4] = {pacc[mt].x, pacc[mt].y, pacc[mt].z, pacc[mt].w};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (n < a.pred_cout) {
                        float o = vals[n] + a.P_bp[n];
                        if (a.pred_act == 2) o = expf(o);
                        a.P_Y[crop * a.pred_stride + n * 256 + px] = o;
                    }
                }
            }
        }
        return;
    }
    if (a.pred_cout > 0) {          // prediction head: lanes lk == 0 hold channels 0..3 of their pixel
        if (lk == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int px = (y0 + mt) * S + li;
                const f32x4 v = accp[mt][0];
                const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (n < a.pred_cout) {
                        float o = vals[n] + a.bp[n];
                        if (a.pred_act == 2) o = expf(o);
                        a.Y[crop * a.pred_stride + n * 256 + px] = o;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        const int n = nt * 16 + lk * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bp + n);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long m = crop * 256 + (y0 + mt) * S + li;
            f32x4 v = accp[mt][nt] + b;
            if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
            if (a.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!(FEAR_ABL & 2048) || v.x == 1234.5f) *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;      // (ablation: no output stores)
            if (CORR) accp[mt][nt] = v;          // the finished feature fragment = B operand of the correlation
        }
    }
    if (CORR) {
        constexpr int TZ = 64, NTZ = TZ / 16;
        f32x4 cacc[2][NTZ];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < NTZ; ++q) cacc[mt][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kg = 0; kg < NTP; ++kg) {
            // A fragment of (kg, q): lane l holds z[kg*16 + 4*(l>>4) + i][q*16 + (l&15)], i = 0..3
            const float* zr = ZL + (kg * 16 + lk * 4) * TZ + li;
            f32x4 zf[NTZ];
#pragma unroll
            for (int q = 0; q < NTZ; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) zf[q][i] = zr[i * TZ + q * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < NTZ; ++q) {
                    cacc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[q][i], accp[0][kg][i], cacc[0][q], 0, 0, 0);
                    cacc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[q][i], accp[1][kg][i], cacc[1][q], 0, 0, 0);
                }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long m = crop * 256 + (y0 + mt) * S + li;
#pragma unroll
            for (int q = 0; q < NTZ; ++q) *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + COUT + q * 16 + lk * 4) = cacc[mt][q];
        }
    }
}

template <int CIN, int CEXPP, int COUT, int KS, int ST, int TW, int TH, bool EXPAND>
struct IrT2Geom {
    static constexpr int CE = 16, P = KS / 2, IWR = (TW - 1) * ST + KS, IHR = (TH - 1) * ST + KS, ES = CE + 4;
    static constexpr int SEG = TW / 16, NMT_OUT = TH * SEG, MTC = NMT_OUT / 8;
    static constexpr int NMT_IN_MAX = (IHR * IWR + 15) / 16, MTA = (NMT_IN_MAX + 7) / 8;
    static constexpr int NCHUNK = CEXPP / CE, NTP = (COUT + 15) / 16, KG = EXPAND ? (CIN + 15) / 16 : 0;
    static constexpr int AP = EXPAND ? KG * 256 + 16 : 0;
    static constexpr int BP = NTP * 256 + KS * KS * 16 + 16;
    // LDS tile of one 16-channel chunk: FOUR planes, one per channel quad, of [pixel][4 floats].
    // ds_read_b128 is served in four groups of 16 lanes that mix two neighbouring channel quads (lanes {0-3,12-15,20-27} =
    // quad 0 of pixels 0-3,12-15 + quad 1 of pixels 4-11, ...; MI355X_MICROARCH.md §LDS) over 64 banks = 16 units of 16 B.
    // A wave reads 16 pixels ST apart: unit = ST * li + (plane offset of its quad).  Stride 1: the two quads of a group cover
    // pixels {0-3,12-15} and {4-11} -> all 16 units when the planes are a multiple of 16 units apart.  Stride 2: each quad
    // covers the 8 even units -> the planes must be an odd number of units apart (size = 1 unit mod 16).  ds_write_b128 (8
    // consecutive lanes = 8 consecutive pixels of one quad, banks mod 32) is conflict free as well, and nothing is padded:
    // 16 floats per pixel.  (A [pixel][20] layout measured 28-34 % conflict cycles in the stride-1 kernels, a two-plane
    // [pixel][8] one 8-25 %: profiles/r01_sq_counters.txt, r02_sq_counters.txt.)
    static constexpr bool PLANES = true;
    static constexpr int EPX = 4;                                     // floats between neighbouring pixels of a plane
    static constexpr int PLANE = (IHR * IWR * 4 + 63) / 64 * 64 + (ST == 2 ? 4 : 0);
    static constexpr int EBUF = 4 * PLANE;
    static constexpr bool KHALF = EXPAND && CIN % 16 == 8;   // the last k-group holds 8 channels: 2 MFMA steps instead of 4
    static constexpr int eo(int pix, int quad) { return quad * PLANE + pix * 4; }
    static constexpr int DUMMY = 256;      // 64 lanes x 16 B: where lanes outside the clipped region park their phase-A store
    static constexpr int NSTAGE = NCHUNK > 1 ? 2 : 1;      // a one-chunk block (the stem tile) needs no second weight stage
    static constexpr int LDS_BYTES = (EBUF + NSTAGE * (AP + BP) + DUMMY) * 4;
};

// Workgroups of a 1-D grid are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b % 8), each with its own L2.  The tile
// kernels want neighbouring tiles of a crop — which share their halo rows and columns — behind the SAME L2: logical tile
// index = (b % 8) * (n / 8) + b / 8 hands every XCD one contiguous eighth of the tile list (whole crops), walked in order.
// FEAR_XCD_SWIZZLE=0 keeps the linear order (tools/kbench A/B).
#ifndef FEAR_XCD_SWIZZLE
#define FEAR_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ unsigned xcd_tile_index(unsigned b, unsigned n) {
    return (FEAR_XCD_SWIZZLE && n % 8 == 0) ? (b & 7) * (n >> 3) + (b >> 3) : b;
}

struct IrT2Args {
    Ir2Args b;
    int H, W, tiles_x, tiles_y;
};

// STEM = true fuses the network stem in front of an e1 block: phase A is then the stem's 3x3 stride-2 conv as an
// implicit GEMM (K = 27 -> 32) gathered straight from the caller's NCHW image (a.X), producing the stem output
// (= the block input) only in LDS; the block's residual is read back from that LDS tile.  CIN must be 27.
// KSPLIT = k > 0 (a handful of crops): gridDim.y workgroups per tile, workgroup y running the k expansion chunks from chunk
// y*k on and writing its RAW partial projection to Y + y * kc_part_stride (splitk_reduce_kernel adds bias / residual / ReLU).
template <int CIN, int CEXPP, int COUT, int KS, int ST, int TW, int TH, bool EXPAND, int MINW, bool STEM = false, int KSPLIT = 0, int IO = 0>
__global__ __launch_bounds__(512, MINW) void ir_tile_v2_kernel(IrT2Args t) {
    static_assert(!STEM || (EXPAND && CIN == 27 && ST == 1 && CEXPP == 16), "stem mode");
    static_assert(IO == 0 || (KSPLIT == 0 && (STEM ? IO == IO_Y_BF16 : !EXPAND)), "bf16 storage: the stem tile's output, the e1 blocks' input / output / residual");
    static_assert(KSPLIT == 0 || (!STEM && EXPAND && KSPLIT >= 2 && (CEXPP / 16) % KSPLIT == 0), "KSPLIT = chunks per workgroup");
    using G = IrT2Geom<CIN, CEXPP, COUT, KS, ST, TW, TH, EXPAND>;
    const Ir2Args& a = t.b;
    constexpr int P = G::P, IWR = G::IWR, IHR = G::IHR, ES = G::ES, SEG = G::SEG, MTC = G::MTC, MTA = G::MTA;
    constexpr int NCHUNK = KSPLIT ? KSPLIT : G::NCHUNK, NTP = G::NTP, KG = G::KG, AP = G::AP, BP = G::BP, EBUF = G::EBUF, EPX = G::EPX;
    constexpr bool KHALF = G::KHALF;
    constexpr int CST = AP + BP, W4 = CST / 4, NRW = (W4 + 511) / 512;
    const int c_base = KSPLIT ? (int)blockIdx.y * KSPLIT : 0;         // first chunk of this workgroup
    static_assert(G::NMT_OUT % 8 == 0 && (SEG == 1 || SEG == 2), "tile shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const E = lds;               // [EBUF]
    float* const WS = lds + EBUF;       // [2][AP + BP]

    const long long tk_start = (FEAR_ABL & 4096) ? wall_clock64() : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index in an SGPR: the per-m-tile trip counts and every wave-dependent address stay on the scalar unit (a VGPR
    // copy made each `(wave + 8 * i) * 16 >= NPIX` test an exec-mask update with two VALU instructions, 5.8 cycles each on the
    // ALU the MFMAs share: profiles/r03_issue_probe.txt)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int tiles = t.tiles_x * t.tiles_y;
    const unsigned tix = xcd_tile_index(blockIdx.x, gridDim.x);
    const long crop = tix / tiles;
    const int tile = tix % tiles;
    const int ox0 = (tile % t.tiles_x) * TW, oy0 = (tile / t.tiles_x) * TH;
    const int ix0 = ox0 * ST - P, iy0 = oy0 * ST - P;
    const int cx_lo = max(ix0, 0), cy_lo = max(iy0, 0);
    const int CW = min(ix0 + IWR, t.W) - cx_lo, CH = min(iy0 + IHR, t.H) - cy_lo;
    const int NPIX = CW * CH;
    const float inv_cw = __builtin_amdgcn_rcpf((float)CW);      // 1 ulp is plenty: (q + 0.5) / CW stays 0.5 / CW away from every integer
    const int Wo = t.W / ST, Ho = t.H / ST;
    const float* Xc = STEM ? a.X + crop * 3 * (2 * t.H) * (2 * t.W) : a.X + ((IO & IO_X_BF16) ? 0 : crop * t.H * t.W * a.ldx);
    const long xbase = (IO & IO_X_BF16) ? crop * t.H * t.W * a.ldx : 0;      // (bf16 input: offsets in elements from a.X)

    // stem mode stages the image patch in the E area first (see below) and zeroes the out-of-image positions afterwards
    if (!STEM && !(FEAR_ABL & 1024))
        for (int i = tid * 4; i < EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(E + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weights of chunk 0 -> registers (committed after the zero-fill barrier)
    f32x4 rw[NRW];
    auto load_w = [&](int c) {
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
            const int idx = tid + r * 512;
            if (idx < W4) rw[r] = *reinterpret_cast<const f32x4*>(a.Wpk + (long)(c_base + c) * CST + idx * 4);
        }
    };
    auto store_w = [&](int c) {
        float* dst = WS + (c & 1) * CST;
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
            const int idx = tid + r * 512;
            if (idx < W4) *reinterpret_cast<f32x4*>(dst + idx * 4) = rw[r];
        }
    };
    load_w(0);

    int eoff[MTA];
    unsigned xoff[MTA];
    f32x4 xf[EXPAND ? MTA : 1][EXPAND ? KG : 1];
    // stem mode: the image patch under the tile's stem-output region (3 planes x (2*IHR+1) rows x PWID floats, zero outside
    // the image) is staged in LDS with aligned 16-byte loads — it aliases the E tile, which is only written after every wave
    // has gathered its im2col fragments (the loop-top barrier) — and the 8 taps per lane (k = kg*16 + 4*lk + j) are
    // ds_read_b32 gathers from it.  (Gathering straight from global memory cost 40 scattered dword loads per lane: 40 % of
    // the kernel, tools/kbench FEAR_ABL=512.)
    constexpr int PR = 2 * IHR + 1, PWID = ((2 * IWR + 1 + 3 + 3) / 4) * 4, PF4 = PWID / 4;
    static_assert(!STEM || 3 * PR * PWID <= EBUF, "the image patch must fit in the E tile it aliases");
    int s_off[STEM ? 8 : 1];
    if (STEM) {
        // patch row r <-> image row 2*iy0 - 1 + r; patch col c <-> image col 2*ix0 - 2 + c (2*ix0 - 2 is a multiple of 4:
        // tiles start at multiples of 32 and the halo is one stem pixel)
        // Integer VALU instructions cost 5.8 cycles of the ALU the MFMAs run on (profiles/r03_issue_probe.txt) and this kernel was
        // 63 % non-MFMA vector instructions, so the index arithmetic is kept off the vector ALU: a thread owns ONE patch column
        // (f) and a row slot (one division by a constant per thread, not two per float4), walks the three planes and its rows with
        // adds, and the im2col tap offsets come from a compile-time table instead of eight k / 9, k % 9 / 3, k % 3 chains.
        const int img_h = 2 * t.H, img_w = 2 * t.W;
        const int row0 = 2 * iy0 - 1, col0 = 2 * ix0 - 2;
        constexpr int RS = 512 / PF4, NR = (PR + RS - 1) / RS;  // row slots: RS * PF4 <= 512 threads stage, the rest idle here
        const int rs = tid / PF4, f = tid - rs * PF4;
        // All 3 * NR loads of a thread are issued back to back and committed to LDS afterwards — with a branch around each load
        // the compiler had serialised them (load, s_waitcnt vmcnt(0), ds_write, six times: six memory round trips in a row at the
        // head of every tile).  Buffer loads make them branch free: a lane outside the image (or beyond the patch rows) gets an
        // out-of-range offset, for which the hardware returns zeros — the conv's zero padding — without touching memory.  The
        // per-plane part of the address is the instruction's scalar offset: one VGPR offset per row slot.
        f32x4 pv[3][NR];
        if (!(FEAR_ABL & 512)) {
            const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xc), 0, 3 * img_h * img_w * 4, 0x00020000);
            const int ix = col0 + 4 * f;
            const bool xin = rs < RS && ix >= 0 && ix < img_w;
            const int voff = ((row0 + rs) * img_w + ix) * 4;
            int vo[NR];
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int pr = n * RS + rs, iy = row0 + pr;
                vo[n] = (xin && pr < PR && iy >= 0 && iy < img_h) ? voff + n * RS * img_w * 4 : (int)0x80000000;   // (the range check is on this offset alone: it must not be negative for a valid lane)
            }
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int n = 0; n < NR; ++n)
                    pv[ci][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img, vo[n], ci * img_h * img_w * 4, 0));
        }
        // tap j (= MFMA step j) of lane group lk is k = 4 * j + lk = (ci * 3 + ky) * 3 + kx (pack_fused16_host's stem order: the
        // K padding is lane group 3 of step 6 — zero weights, any finite value will do — and step 7, which is never issued);
        // stem pixel (gy, gx) reads image col 2*gx - 1 + kx = col0 + 2*(gx - ix0) + 1 + kx
        struct Tab {
            int v[4][8];
            constexpr Tab() : v{} {
                for (int g = 0; g < 4; ++g)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 4 * j + g;
                        const int kk = k < 27 ? k : 0;
                        v[g][j] = ((kk / 9) * PR + (kk % 9) / 3) * PWID + kk % 3 + 1;
                    }
            }
        };
        static constexpr Tab tab{};
#pragma unroll
        for (int j = 0; j < 8; ++j) s_off[j] = tab.v[lk][j];
        if (rs < RS && !(FEAR_ABL & 512)) {
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int n = 0; n < NR; ++n)
                    if (n * RS + rs < PR) *reinterpret_cast<f32x4*>(E + (ci * PR + n * RS + rs) * PWID + 4 * f) = pv[ci][n];
        }
        __syncthreads();                               // patch complete
    }
    if constexpr (STEM) {
        // m-tile i of this wave is pixels q = (wave + 8 i) * 16 + li of the clipped region (row-major, CW columns).  One division for
        // i = 0; every further m-tile is 128 pixels on = qa rows + qb columns with at most one wrap, and both offsets a lane needs
        // (its pixel in the E tile, its pixel in the patch) are linear in (row, column): a compare and three selects per m-tile
        // instead of a division and three multiplications — the prologue runs once per tile and the stem has ONE chunk, so here it
        // is a tenth of the kernel's instruction issue.
        const int q0 = wave * 16 + li;
        const int cy0 = (int)(((float)q0 + 0.5f) * inv_cw), cx0 = q0 - cy0 * CW;
        const int qa = 128 / CW, qb = 128 - qa * CW;
        const int dpe0 = qa * IWR + qb, dpe1 = dpe0 + IWR - CW, dpq0 = qa * PWID + qb, dpq1 = dpq0 + PWID - CW;
        int cx = cx0;
        int pe = (cy0 + cy_lo - iy0) * IWR + cx0 + cx_lo - ix0;
        int pq = (cy0 + cy_lo - iy0) * PWID + cx0 + cx_lo - ix0;
#pragma unroll
        for (int i = 0; i < MTA; ++i) {
            // lanes beyond the clipped region store to a dummy slot (phase A stays branch free); what they gather is any LDS word
            eoff[i] = q0 < NPIX - 128 * i ? G::eo(pe, lk) : EBUF + G::NSTAGE * CST + lane * 4;
            xoff[i] = 0;
            const float* pp = E + 2 * pq;
            float v[8];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = pp[s_off[j]];
            xf[i][0] = (f32x4){v[0], v[1], v[2], v[3]};
            xf[i][1] = (f32x4){v[4], v[5], v[6], 0.f};
            if (i + 1 < MTA) {
                cx += qb;
                const bool wrap = cx >= CW;
                cx -= wrap ? CW : 0;
                pe += wrap ? dpe1 : dpe0;
                pq += wrap ? dpq1 : dpq0;
            }
        }
    }
    if constexpr (!STEM) {
        // the same stepping for the other tiles: pixel in the E tile and pixel in the input map (a 32-bit offset from the crop's
        // base: the loads become [scalar base + lane offset]) are both linear in (row, column).  The old form — a division, a
        // 64-bit multiply-add chain and the tile-local multiplication per m-tile — was 200 vector instructions of prologue in the
        // stage-2 tile, a tenth of its issue.  Lanes beyond the clipped region read pixel 0 and store to a dummy slot.
        const int q0 = wave * 16 + li;
        const int cy0 = (int)(((float)q0 + 0.5f) * inv_cw), cx0 = q0 - cy0 * CW;   // exact for q0 < 2^16, CW <= 64 (garbage past NPIX: masked)
        const int qa = 128 / CW, qb = 128 - qa * CW;
        const int dpe0 = qa * IWR + qb, dpe1 = dpe0 + IWR - CW;
        const int dpx0 = (qa * t.W + qb) * a.ldx, dpx1 = dpx0 + (t.W - CW) * a.ldx;
        int cx = cx0;
        int pe = (cy0 + cy_lo - iy0) * IWR + cx0 + cx_lo - ix0;
        int px = ((cy0 + cy_lo) * t.W + cx0 + cx_lo) * a.ldx;
#pragma unroll
        for (int i = 0; i < MTA; ++i) {
            const bool valid = q0 < NPIX - 128 * i;
            eoff[i] = valid ? G::eo(pe, lk) : EBUF + G::NSTAGE * CST + lane * 4;
            xoff[i] = valid ? (unsigned)px : 0u;
            if (EXPAND) {
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    xf[i][kg] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (KHALF && kg == KG - 1) {       // 8 channels: lane group lk holds channels 2*lk, 2*lk+1 (k-step q = .x / .y)
                        const float2 h2 = *reinterpret_cast<const float2*>(Xc + (xoff[i] + (unsigned)(kg * 16 + lk * 2)));
                        xf[i][kg].x = h2.x; xf[i][kg].y = h2.y;
                    } else if (kg * 16 + lk * 4 < CIN) xf[i][kg] = *reinterpret_cast<const f32x4*>(Xc + (xoff[i] + (unsigned)(kg * 16 + lk * 4)));
                }
            }
            if (i + 1 < MTA) {
                cx += qb;
                const bool wrap = cx >= CW;
                cx -= wrap ? CW : 0;
                pe += wrap ? dpe1 : dpe0;
                px += wrap ? dpx1 : dpx0;
            }
        }
    }
    // !EXPAND: activations of the next chunk are prefetched into registers as well
    f32x4 rx[EXPAND ? 1 : MTA];
    auto load_x = [&](int c) {
        if (!EXPAND) {
#pragma unroll
            for (int i = 0; i < MTA; ++i) {
                rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (c * 16 + lk * 4 < CIN) rx[i] = ld_act4<(IO & IO_X_BF16) != 0>(Xc, xbase + xoff[i] + c * 16 + lk * 4);
            }
        }
    };
    load_x(0);
    __syncthreads();
    store_w(0);

    const int seg = SEG == 1 ? 0 : (wave & 1);
    const int r0 = (SEG == 1 ? wave : (wave >> 1)) * MTC;
    f32x4 accp[MTC][NTP];
#pragma unroll
    for (int r = 0; r < MTC; ++r)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) accp[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the projection bias: read in the epilogue it is one more exposed memory round trip at the tail of every tile.  The one- and
    // two-chunk kernels (stem tile, e1 blocks), where that tail is a visible share, fetch it here; the others would pay for the
    // 4 * NTP registers with occupancy (stage 2: 76 -> 84 VGPRs = two workgroups per CU instead of three, +3 %) and keep it late.
    constexpr bool BIAS_EARLY = !KSPLIT && NCHUNK <= 2 && !(FEAR_ABL & 16384);
    f32x4 bpv[NTP];
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        bpv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (BIAS_EARLY && nt * 16 + lk * 4 < COUT) bpv[nt] = *reinterpret_cast<const f32x4*>(a.bp + nt * 16 + lk * 4);
    }

    const bool res_from_tile = !EXPAND && !KSPLIT && !(FEAR_ABL & 65536) && a.R == a.X && a.ldr == a.ldx && ST == 1 && CIN == COUT;
    const long long tk_begin = (FEAR_ABL & 4096) ? wall_clock64() : 0;   // kbench -DFEAR_ABL=4096: 10 ns ticks per region
    long long tm[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < ((FEAR_ABL & 256) ? 0 : NCHUNK); ++c) {
        const float* wa = WS + (c & 1) * CST;
        const float* wb = wa + AP;
        const long long tk0 = (FEAR_ABL & 4096) ? wall_clock64() : 0;
        if (EXPAND && !(FEAR_ABL & 1)) __syncthreads();          // stage c&1 committed (first chunk: by store_w(0) above)
        const long long tk1 = (FEAR_ABL & 4096) ? wall_clock64() : 0;
        // ---- phase A: E <- relu(expand) (or the raw activations)
        if (EXPAND) {
            f32x4 wf[KG > 0 ? KG : 1];
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) wf[kg] = *reinterpret_cast<const f32x4*>(wa + kg * 256 + lane * 4);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);
            auto expand_tile = [&](int i) -> f32x4 {
                f32x4 acc = bias;
#pragma unroll
                for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                    for (int q = 0; q < ((KHALF && kg == KG - 1) ? 2 : (STEM && kg == 1) ? 3 : 4); ++q) {
                        if (FEAR_ABL & 8) { acc += wf[kg] * xf[i][kg][q]; continue; }
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kg][q], xf[i][kg][q], acc, 0, 0, 0);
                    }
                acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
                return acc;
            };
#pragma unroll
            for (int i = 0; i < MTA; ++i) {
                if ((wave + 8 * i) * 16 >= NPIX) break;            // wave-uniform: whole m-tile beyond the clipped region
                *reinterpret_cast<f32x4*>(E + eoff[i]) = expand_tile(i);
            }
            if (STEM && (CW < IWR || CH < IHR) && !(FEAR_ABL & 1024)) {
                // border tile: the positions outside the stem-output map (the depthwise's zero padding) still hold patch bytes
                for (int p = tid; p < IHR * IWR; p += 512) {
                    const int ry = p / IWR, rxx = p - ry * IWR;
                    const int gy = iy0 + ry, gx = ix0 + rxx;
                    if (gy < 0 || gy >= t.H || gx < 0 || gx >= t.W) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(E + G::eo(p, q)) = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MTA; ++i) {
                if ((wave + 8 * i) * 16 >= NPIX) break;
                *reinterpret_cast<f32x4*>(E + eoff[i]) = rx[i];
            }
        }
        // prefetch the next chunk's weights (and activations) while this chunk computes
        if (c + 1 < NCHUNK && !(FEAR_ABL & 2)) { load_w(c + 1); load_x(c + 1); }
        const long long tk2 = (FEAR_ABL & 4096) ? wall_clock64() : 0;
        if (!(FEAR_ABL & 1)) __syncthreads();
        const long long tk3 = (FEAR_ABL & 4096) ? wall_clock64() : 0;
        // ---- phase B: depthwise from E, weights from the LDS stage
        f32x4 d[MTC];
        {
            const float* wd = wb + NTP * 256 + lk * 4;
            const f32x4 bd = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
#pragma unroll
            for (int r = 0; r < MTC; ++r) d[r] = bd;
            const float* Ebase = E + G::eo((r0 * ST) * IWR + (seg * 16 + li) * ST, lk);
#pragma unroll
            for (int iy = 0; iy < (MTC - 1) * ST + KS; ++iy) {
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    if ((FEAR_ABL & 4) && (iy | kx)) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Ebase + (iy * IWR + kx) * EPX);
#pragma unroll
                    for (int r = 0; r < MTC; ++r) {
                        const int ky = iy - r * ST;
                        if (FEAR_ABL & 32) { d[r].x += v.x; continue; }
                        if (ky >= 0 && ky < KS) d[r] += v * *reinterpret_cast<const f32x4*>(wd + (ky * KS + kx) * 16);
                    }
                }
            }
            if (a.relu_dw) {
#pragma unroll
                for (int r = 0; r < MTC; ++r) {
                    d[r].x = fmaxf(d[r].x, 0.f); d[r].y = fmaxf(d[r].y, 0.f); d[r].z = fmaxf(d[r].z, 0.f); d[r].w = fmaxf(d[r].w, 0.f);
                }
            }
        }
        // an e1 block's residual IS its input, and the raw input channels 16c .. 16c + 15 are this chunk's E tile: the residual of
        // output tile c is added here from LDS — the epilogue's global re-read was an exposed L2 round trip at the tail of every tile
        if (!EXPAND && res_from_tile) {
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt)
                if (c == nt) {
#pragma unroll
                    for (int r = 0; r < MTC; ++r)
                        accp[r][nt] += *reinterpret_cast<const f32x4*>(E + G::eo(((r0 + r) * ST + P) * IWR + (seg * 16 + li) * ST + P, lk));
                }
        }
        const long long tk4 = (FEAR_ABL & 4096) ? wall_clock64() : 0;
        // ---- phase C: projection
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(wb + nt * 256 + lane * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < MTC; ++r) {
                    if (FEAR_ABL & 16) { accp[r][nt] += wp * d[r][q]; continue; }
                    accp[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[q], d[r][q], accp[r][nt], 0, 0, 0);
                }
        }
        const long long tk5 = (FEAR_ABL & 4096) ? wall_clock64() : 0;
        if (c + 1 < NCHUNK && !(FEAR_ABL & 2)) store_w(c + 1);
        // !EXPAND: E is rewritten by the next chunk (EXPAND syncs at loop top); after the last chunk nothing writes LDS any more
        if (((!EXPAND && c + 1 < NCHUNK) || ((FEAR_ABL & 8192) && c + 1 == NCHUNK)) && !(FEAR_ABL & 1)) __syncthreads();
        if (FEAR_ABL & 4096) {
            const long long tk6 = wall_clock64();
            tm[0] += tk1 - tk0; tm[1] += tk2 - tk1; tm[2] += tk3 - tk2; tm[3] += tk4 - tk3; tm[4] += tk5 - tk4; tm[5] += tk6 - tk5;
        }
    }
    const long long tk_loop_end = (FEAR_ABL & 4096) ? wall_clock64() : 0;

    if (KSPLIT) {
        float* Yp = a.Y + (long)blockIdx.y * a.kc_part_stride;
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) {
            const int n = nt * 16 + lk * 4;
            if (n >= COUT) continue;
#pragma unroll
            for (int r = 0; r < MTC; ++r) {
                const long m = (crop * Ho + oy0 + r0 + r) * Wo + ox0 + seg * 16 + li;
                *reinterpret_cast<f32x4*>(Yp + m * a.ldy + n) = accp[r][nt];
            }
        }
        return;
    }
    // Addresses: the row of a wave is uniform (crop, tile and wave index live in SGPRs), so a store is [scalar base + one 32-bit
    // lane offset] — the 64-bit pixel index per row cost two v_mad_u64_u32, two v_mul_lo_u32 and a 64-bit add each.  The ReLU is
    // a max against 0 or -inf (a uniform select) instead of a branch with a register copy on its other side.
    const long m0 = (crop * Ho + oy0 + r0) * Wo + ox0 + seg * 16;
    const unsigned ylane = (unsigned)(li * a.ldy + lk * 4), rlane = (unsigned)(li * a.ldr + lk * 4);
    const float relu_lo = a.relu_out ? 0.f : -3.0e38f;
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        if (nt * 16 >= COUT) continue;
        const bool n_ok = COUT % 16 == 0 || nt * 16 + lk * 4 < COUT;
#pragma unroll
        for (int r = 0; r < MTC; ++r) {
            const long mrow = m0 + (long)r * Wo;
            if (!BIAS_EARLY && r == 0 && n_ok) bpv[nt] = *reinterpret_cast<const f32x4*>(a.bp + nt * 16 + lk * 4);
            f32x4 v = accp[r][nt] + bpv[nt];
            if (STEM) v += *reinterpret_cast<const f32x4*>(E + G::eo((r0 + r + P) * IWR + seg * 16 + li + P, lk));   // NTP == 1: n = 4 * lk
            else if (a.R && n_ok && !res_from_tile) v += ld_act4<(IO & IO_R_BF16) != 0>(a.R, mrow * a.ldr + nt * 16 + (long)rlane);
            v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);
            if (n_ok && (!(FEAR_ABL & 2048) || v.x == 1234.5f)) st_act4<(IO & IO_Y_BF16) != 0>(a.Y, mrow * a.ldy + nt * 16 + (long)ylane, v);
        }
    }
    if ((FEAR_ABL & 4096) && a.P_Y && blockIdx.x == 1000 && lane == 0) {
        float* dbg = a.P_Y + wave * 10;
        for (int q = 0; q < 6; ++q) dbg[q] = (float)tm[q];
        dbg[6] = (float)(tk_begin - tk_start);
        dbg[7] = (float)(wall_clock64() - tk_loop_end);
    }
}

// ================================================================================================
// ir_tile_v4: ir_tile_v2's tile, weights and LDS footprint with the three phases of a chunk OVERLAPPED inside each wave.
//
// v2 runs expansion (MFMA), depthwise (LDS reads + packed FMAs) and projection (MFMA) of a 16-channel chunk one after the
// other with a barrier in between; every wave of the workgroup is in the same phase, so the LDS pipe idles during the MFMA
// phases and the ALU during the depthwise's LDS round trips.  With two or three workgroups on a CU the other workgroups fill
// part of that; the tiles whose E tile takes half the LDS (stages 6, 9, 10: one workgroup per CU, 2 waves per SIMD) have
// nobody to fill it: 59-75 % ALU-busy (profiles/r02_sq_counters.txt) against 85 % for the small tiles.
// Here the depthwise of chunk c is a chain of tap steps whose ds_read_b128s run D steps ahead of the FMAs that use them, and
// the expansion MFMAs of chunk c + 1 are dealt out between the steps — ir16_interval's pipeline — but the expanded chunk
// c + 1 stays in REGISTERS (one float4 per m-tile) until the barrier that ends the reads of chunk c, so the E tile is not
// double buffered and the LDS footprint is v2's (the double-buffered variant measured in round 1 lost its occupancy):
//     interval 1   taps(c) from E  ||  expansion MFMAs (c + 1) -> registers          barrier
//     interval 2   registers -> E (chunk c + 1)  ||  projection MFMAs (c)             barrier
// ds_read issue is free beside MFMAs, a ds_write_b128 costs ~20 cycles of the SIMD's ALU time, an MFMA <-> VALU switch ~8
// (profiles/r03_issue_probe.txt): the E writes sit next to the projection MFMAs, the FMAs of a step are issued together.
// Weights: A-part (expansion) and BC-part (depthwise + projection) of the packed per-chunk blocks are double buffered
// separately and arrive by asynchronous global -> LDS copies issued one interval ahead.
// compile-time loop: f(std::integral_constant<int, B>{}), ..., f(std::integral_constant<int, E - 1>{}) — every index a constant
// expression whatever the unroller's size heuristics decide (a partially unrolled tap loop turns the register arrays it indexes
// into scratch memory)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ================================================================================================
// sep16 for a handful of crops (the batch-1 tracker): one 16-channel output slice of ROWS map rows per workgroup — the N-split
// launch of sep16_kernel<CIN, 16, KS> cut once more, over rows — with one row per wave and nothing fetched inside the chunk loop.
// At one crop a sep16_kernel<CIN, 16, KS> workgroup is ALONE on its CU and still does the whole depthwise of the crop
// (CIN x 256 pixels, recomputed by each of the COUT/16 slices): 2 waves per SIMD x 2 rows x CIN/16 chunks of
// [18 packed FMAs + 8 MFMAs + LDS round trip] = 10 us of instruction issue behind a 5 us launch, six launches in a row per
// tower (profiles/r02_batch1_timeline.txt).  (Prefetching everything into registers / LDS first does not help — measured
// 22.6 vs 18.6 us: the loop is issue-bound, not latency-bound.)  Cutting the map into 16 / ROWS row groups puts one row on a
// SIMD: a fraction of the issue time per workgroup, 16 / ROWS times the workgroups (128 per layer and crop at ROWS = 2);
// the ROWS + 2P input rows of a group are loaded once, before the loop (a 0.2 us interval cannot hide a global load), the
// slice's packed weight set (CIN/16 x [1 fragment | Wd | bd], 26-33 KB) likewise.
// Same arithmetic as sep16_kernel (bias, then taps kx outer / ky inner; projection chunk by chunk): with KSPLIT = 1 the outputs
// are bit-identical to it; with KSPLIT wave groups the projection is summed per group and the groups added in order
// (deterministic, 1e-6 from the single-group sum).
template <int CIN, int KS, int ROWS, int KSPLIT = 1>
struct Sep16TinyGeom {
    static constexpr int S = 16, P = KS / 2, PW = S + 2 * P, PH = ROWS + 2 * P, NCHUNK = CIN / 16;
    static constexpr int WPF = 256, WDF = KS * KS * 16 + 16, CST = WPF + WDF;
    static constexpr int EP = 4, EQ = (PH * PW * 4 + 63) / 64 * 64;      // four planes (one per channel quad) of [pixel][4]
    static constexpr int EBUF = 4 * EQ;
    static constexpr int LDS_BYTES = (KSPLIT * 2 * EBUF + NCHUNK * CST) * 4;
};

// KSPLIT > 1: KSPLIT wave groups per workgroup, group kh running the input chunks [kh, kh + 1) * NCHUNK / KSPLIT on its own pair
// of LDS tiles; the groups' projections are added in group order at the end (through LDS).
template <int CIN, int KS, int ROWS, int KSPLIT = 1>
__global__ __launch_bounds__(64 * ROWS * KSPLIT) void sep16_tiny_kernel(Ir2Args a) {
    using G = Sep16TinyGeom<CIN, KS, ROWS, KSPLIT>;
    constexpr int S = G::S, P = G::P, PW = G::PW, EP = G::EP, EQ = G::EQ, NCHUNK = G::NCHUNK / KSPLIT, NCHUNK_ALL = G::NCHUNK;
    constexpr int WPF = G::WPF, CST = G::CST, EBUF = G::EBUF, NT = 64 * ROWS * KSPLIT;
    static_assert(G::NCHUNK % KSPLIT == 0, "chunks per wave group");
    constexpr int NS = KS * KS, D = 4, NU = 4;
    constexpr int W4 = NCHUNK_ALL * CST / 4, NRW = (W4 + NT - 1) / NT;
    static_assert(CIN % 16 == 0 && NCHUNK >= 2 && NS >= D && 16 % ROWS == 0 && 2 * P <= ROWS, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const WALL = lds + KSPLIT * 2 * EBUF;   // [NCHUNK_ALL][CST]

    const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % ROWS, kh = wave_all / ROWS;
    float* const Ebuf = lds + kh * 2 * EBUF;       // [2][EBUF] of this wave group
    const int cg0 = kh * NCHUNK;                   // first input chunk of this wave group
    const int li = lane & 15, lk = lane >> 4;
    const long crop = blockIdx.x;
    const float* Xc = a.X + crop * 256 * a.ldx;
    const int r0 = blockIdx.z * ROWS;           // first output row of this workgroup; wave w computes row r0 + w
    if (a.nsplit_wstride) {                     // N-split: this workgroup's 16-channel output slice
        a.Wpk += (long)blockIdx.y * a.nsplit_wstride;
        a.bp += blockIdx.y * 16;
        a.Y += blockIdx.y * 16;
        if (a.R) a.R += blockIdx.y * 16;
    }
    // the slice's weights: global -> registers -> LDS, all chunks at once
    f32x4 rw[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) {
        const int idx = tid + r * NT;
        if (idx < W4) rw[r] = *reinterpret_cast<const f32x4*>(a.Wpk + (long)idx * 4);
    }
    // the tile rows this wave stages: slot = wave (input row r0 - P + wave) and, for the first 2P waves, slot = ROWS + wave
    const int ya = r0 - P + wave, yb = r0 - P + ROWS + wave;
    const bool has_a = ya >= 0 && ya < S, has_b = wave < 2 * P && yb < S;
    f32x4 rx[NCHUNK][2];
    static_for<0, NCHUNK>([&](auto C) {
        constexpr int c = decltype(C)::value;
        if (has_a) rx[c][0] = *reinterpret_cast<const f32x4*>(Xc + (long)(ya * S + li) * a.ldx + (cg0 + c) * 16 + lk * 4);
        if (has_b) rx[c][1] = *reinterpret_cast<const f32x4*>(Xc + (long)(yb * S + li) * a.ldx + (cg0 + c) * 16 + lk * 4);
    });
    for (int i = tid * 4; i < KSPLIT * 2 * EBUF; i += NT * 4) *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NRW; ++r) {
        const int idx = tid + r * NT;
        if (idx < W4) *reinterpret_cast<f32x4*>(WALL + idx * 4) = rw[r];
    }
    __syncthreads();                            // zero fill and weights in place (rows outside the map are never written: zero)

    auto store_x = [&](auto C) {
        constexpr int c = decltype(C)::value;
        float* E = Ebuf + (c & 1) * EBUF;
        if (has_a) *reinterpret_cast<f32x4*>(E + (wave * PW + li + P) * EP + lk * EQ) = rx[c][0];
        if (has_b) *reinterpret_cast<f32x4*>(E + ((ROWS + wave) * PW + li + P) * EP + lk * EQ) = rx[c][1];
    };

    f32x4 accp = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 d0;                           // depthwise result of the chunk being projected
    // projection of chunk c (PROJ) dealt between the tap steps of the depthwise of chunk c + 1 (DW)
    auto interval = [&](int c, auto proj_tag, auto dw_tag) {
        constexpr bool PROJ = decltype(proj_tag)::value, DW = decltype(dw_tag)::value;
        const int cd = PROJ ? c + 1 : c;
        const float* wd = WALL + (cg0 + cd) * CST + WPF + lk * 4;
        const float* e0 = Ebuf + (cd & 1) * EBUF + (wave * PW + li) * EP + lk * EQ;
        f32x4 n0, ev[D], wv[D], wpq = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (PROJ) wpq = *reinterpret_cast<const f32x4*>(WALL + (cg0 + c) * CST + lane * 4);
        if (DW) {
            n0 = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
#pragma unroll
            for (int t = 0; t < D; ++t) {
                const int kx = t / KS, ky = t % KS;
                ev[t] = *reinterpret_cast<const f32x4*>(e0 + (ky * PW + kx) * EP);
                wv[t] = *reinterpret_cast<const f32x4*>(wd + (ky * KS + kx) * 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            f32x4 e, w;
            if (DW) {
                e = ev[t % D]; w = wv[t % D];
                if (t + D < NS) {
                    const int kx2 = (t + D) / KS, ky2 = (t + D) % KS;
                    ev[t % D] = *reinterpret_cast<const f32x4*>(e0 + (ky2 * PW + kx2) * EP);
                    wv[t % D] = *reinterpret_cast<const f32x4*>(wd + (ky2 * KS + kx2) * 16);
                }
            }
            if (PROJ) {
#pragma unroll
                for (int u = t * NU / NS; u < (t + 1) * NU / NS; ++u)
                    accp = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[u], d0[u], accp, 0, 0, 0);
            }
            if (DW) pk_fma4(n0, e, w);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DW) {
            f32x4 dummy = n0;
            pk_fma_settle(n0, dummy);
            if (a.relu_dw) { n0.x = fmaxf(n0.x, 0.f); n0.y = fmaxf(n0.y, 0.f); n0.z = fmaxf(n0.z, 0.f); n0.w = fmaxf(n0.w, 0.f); }
            d0 = n0;
        }
    };

    store_x(std::integral_constant<int, 0>{});
    store_x(std::integral_constant<int, 1>{});
    __syncthreads();
    interval(0, std::false_type{}, std::true_type{});
    __syncthreads();                            // E[0] is overwritten at the end of interval 0
    static_for<0, NCHUNK>([&](auto C) {
        constexpr int c = decltype(C)::value;
        if constexpr (c + 1 < NCHUNK) interval(c, std::true_type{}, std::true_type{});
        else interval(c, std::true_type{}, std::false_type{});
        if constexpr (c + 2 < NCHUNK) store_x(std::integral_constant<int, c + 2>{});      // E[c & 1]: last read before the previous barrier
        if constexpr (c + 1 < NCHUNK) __syncthreads();
    });

    if (KSPLIT > 1) {               // add the wave groups' projections up, in group order
        __syncthreads();            // every tile has been read for the last time
        f32x4* red = reinterpret_cast<f32x4*>(lds);
        if (kh > 0) red[((kh - 1) * ROWS + wave) * 64 + lane] = accp;
        __syncthreads();
        if (kh > 0) return;
#pragma unroll
        for (int k = 1; k < KSPLIT; ++k) accp += red[((k - 1) * ROWS + wave) * 64 + lane];
    }
    const int px = (r0 + wave) * S + li;
    if (a.pred_cout > 0) {          // prediction head: lanes lk == 0 hold channels 0..3 of their pixel
        if (lk == 0) {
            const float vals[4] = {accp.x, accp.y, accp.z, accp.w};
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (n < a.pred_cout) {
                    float o = vals[n] + a.bp[n];
                    if (a.pred_act == 2) o = expf(o);
                    a.Y[crop * a.pred_stride + n * 256 + px] = o;
                }
            }
        }
        return;
    }
    const int n = lk * 4;
    const long m = crop * 256 + px;
    f32x4 v = accp + *reinterpret_cast<const f32x4*>(a.bp + n);
    if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
    if (a.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
}

template <int CIN, int CEXPP, int COUT, int KS, int ST, int TW, int TH>
struct IrT4Geom : IrT2Geom<CIN, CEXPP, COUT, KS, ST, TW, TH, true> {
    using B = IrT2Geom<CIN, CEXPP, COUT, KS, ST, TW, TH, true>;
    static constexpr int IR = (B::MTC - 1) * ST + KS;          // input rows under a wave's MTC output rows
    static constexpr int NS = KS * IR;                         // tap steps per chunk: kx outer, input row inner
    static constexpr int D = IR < 4 ? IR : 4;                  // LDS read-ahead in tap steps (never more than one column ahead)
    static constexpr int MPT = B::KHALF ? (B::KG - 1) * 4 + 2 : B::KG * 4;      // expansion MFMAs per m-tile
    static constexpr int NU = B::MTA * MPT;
    static constexpr int LDS_BYTES = (B::EBUF + 2 * B::AP + 2 * B::BP + B::DUMMY) * 4;
};

template <int CIN, int CEXPP, int COUT, int KS, int ST, int TW, int TH, int MINW>
__global__ __launch_bounds__(512, MINW) void ir_tile_v4_kernel(IrT2Args t) {
    using G = IrT4Geom<CIN, CEXPP, COUT, KS, ST, TW, TH>;
    const Ir2Args& a = t.b;
    constexpr int P = G::P, IWR = G::IWR, IHR = G::IHR, SEG = G::SEG, MTC = G::MTC, MTA = G::MTA;
    constexpr int NCHUNK = G::NCHUNK, NTP = G::NTP, KG = G::KG, AP = G::AP, BP = G::BP, EBUF = G::EBUF, EPX = G::EPX;
    constexpr bool KHALF = G::KHALF;
    constexpr int CST = AP + BP, IR = G::IR, NS = G::NS, D = G::D, MPT = G::MPT, NU = G::NU;
    static_assert(G::NMT_OUT % 8 == 0 && (SEG == 1 || SEG == 2) && NCHUNK >= 2 && KG >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const E = lds;                       // [EBUF]
    float* const WA = lds + EBUF;               // [2][AP]
    float* const WB = WA + 2 * AP;              // [2][BP]
    float* const DUMMYP = WB + 2 * BP;          // where lanes beyond the clipped region park their E stores

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int tiles = t.tiles_x * t.tiles_y;
    const unsigned tix = xcd_tile_index(blockIdx.x, gridDim.x);
    const long crop = tix / tiles;
    const int tile = tix % tiles;
    const int ox0 = (tile % t.tiles_x) * TW, oy0 = (tile / t.tiles_x) * TH;
    const int ix0 = ox0 * ST - P, iy0 = oy0 * ST - P;
    const int cx_lo = max(ix0, 0), cy_lo = max(iy0, 0);
    const int CW = min(ix0 + IWR, t.W) - cx_lo, CH = min(iy0 + IHR, t.H) - cy_lo;
    const int NPIX = CW * CH;
    const float inv_cw = __builtin_amdgcn_rcpf((float)CW);
    const int Wo = t.W / ST, Ho = t.H / ST;
    const float* Xc = a.X + crop * t.H * t.W * a.ldx;
    // m-tiles of the clipped input region this wave expands: wave, wave + 8, ... < ceil(NPIX / 16)
    const int n_mt = (NPIX + 15) >> 4;
    const int cnt = wave < n_mt ? (n_mt - wave + 7) >> 3 : 0;

    for (int i = tid * 4; i < EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(E + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage_a = [&](int c) { lds_copy_async<AP>(a.Wpk + (long)c * CST, WA + (c & 1) * AP, wave, lane); };
    auto stage_b = [&](int c) { lds_copy_async<BP>(a.Wpk + (long)c * CST + AP, WB + (c & 1) * BP, wave, lane); };
    stage_a(0);
    stage_b(0);
    stage_a(1);

    int eoff[MTA];
    f32x4 xf[MTA][KG];
    {   // m-tile i is 128 pixels on from m-tile i - 1: stepped (see ir_tile_v2_kernel), loads as [scalar crop base + lane offset]
        const int q0 = wave * 16 + li;
        const int cy0 = (int)(((float)q0 + 0.5f) * inv_cw), cx0 = q0 - cy0 * CW;
        const int qa = 128 / CW, qb = 128 - qa * CW;
        const int dpe0 = qa * IWR + qb, dpe1 = dpe0 + IWR - CW;
        const int dpx0 = (qa * t.W + qb) * a.ldx, dpx1 = dpx0 + (t.W - CW) * a.ldx;
        int cx = cx0;
        int pe = (cy0 + cy_lo - iy0) * IWR + cx0 + cx_lo - ix0;
        int px = ((cy0 + cy_lo) * t.W + cx0 + cx_lo) * a.ldx;
        static_for<0, MTA>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const bool valid = q0 < NPIX - 128 * i;
            eoff[i] = valid ? G::eo(pe, lk) : (int)(DUMMYP - E) + lane * 4;
            const unsigned xoff = valid ? (unsigned)px : 0u;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                xf[i][kg] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (KHALF && kg == KG - 1) {       // 8 channels: lane group lk holds channels 2*lk, 2*lk+1 (k-step q = .x / .y)
                    const float2 h2 = *reinterpret_cast<const float2*>(Xc + (xoff + (unsigned)(kg * 16 + lk * 2)));
                    xf[i][kg].x = h2.x; xf[i][kg].y = h2.y;
                } else if (kg * 16 + lk * 4 < CIN) xf[i][kg] = *reinterpret_cast<const f32x4*>(Xc + (xoff + (unsigned)(kg * 16 + lk * 4)));
            }
            if (i + 1 < MTA) {
                cx += qb;
                const bool wrap = cx >= CW;
                cx -= wrap ? CW : 0;
                pe += wrap ? dpe1 : dpe0;
                px += wrap ? dpx1 : dpx0;
            }
        });
    }

    auto relu4 = [](f32x4& v) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); };

    __syncthreads();                           // zero fill done, A(0) / BC(0) / A(1) landed
    {   // chunk 0's expansion, not overlapped with anything
        const float* wa = WA;
        f32x4 wf[KG];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) wf[kg] = *reinterpret_cast<const f32x4*>(wa + kg * 256 + lane * 4);
        const f32x4 bias = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);
        static_for<0, MTA>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (i < cnt) {                     // wave-uniform
                f32x4 acc = bias;
#pragma unroll
                for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                    for (int q = 0; q < ((KHALF && kg == KG - 1) ? 2 : 4); ++q)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kg][q], xf[i][kg][q], acc, 0, 0, 0);
                relu4(acc);
                *reinterpret_cast<f32x4*>(E + eoff[i]) = acc;
            }
        });
    }
    __syncthreads();

    const int seg = SEG == 1 ? 0 : (wave & 1);
    const int r0 = (SEG == 1 ? wave : (wave >> 1)) * MTC;
    f32x4 accp[MTC][NTP];
#pragma unroll
    for (int r = 0; r < MTC; ++r)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) accp[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* const Ebase = E + G::eo((r0 * ST) * IWR + (seg * 16 + li) * ST, lk);
    // projection bias fetched here, not at the tail: this kernel has its CU to itself, nothing hides a round trip there
    f32x4 bpv[NTP];
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        bpv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (nt * 16 + lk * 4 < COUT) bpv[nt] = *reinterpret_cast<const f32x4*>(a.bp + nt * 16 + lk * 4);
    }

    // (peeling the last chunk, which pays in chain16 / chain32, measured neutral here: 116.4 k -> 116.1 k crops/s, profiles/r06_peel_ab.txt)
    for (int c = 0; c < NCHUNK; ++c) {
        const bool more = c + 1 < NCHUNK;
        if (c + 2 < NCHUNK) stage_a(c + 2);
        if (more) stage_b(c + 1);
        const float* wa = WA + ((c + 1) & 1) * AP;         // expansion weights of chunk c + 1
        const float* wb = WB + (c & 1) * BP;               // depthwise + projection weights of chunk c
        const float* wd = wb + NTP * 256 + lk * 4;
        // ---- interval 1: taps of chunk c, expansion MFMAs of chunk c + 1 dealt between them
        f32x4 wf[KG];
        f32x4 eacc[MTA];
        if (more) {
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) wf[kg] = *reinterpret_cast<const f32x4*>(wa + kg * 256 + lane * 4);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);
#pragma unroll
            for (int i = 0; i < MTA; ++i) eacc[i] = bias;
        }
        f32x4 d[MTC];
        {
            const f32x4 bd = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
#pragma unroll
            for (int r = 0; r < MTC; ++r) d[r] = bd;
        }
        f32x4 ev[D], wv[2][KS];
#pragma unroll
        for (int s = 0; s < D; ++s) {                      // prime the read-ahead window (D <= IR: all in column 0)
            ev[s] = *reinterpret_cast<const f32x4*>(Ebase + (s * IWR) * EPX);
            if (s < KS) wv[0][s] = *reinterpret_cast<const f32x4*>(wd + (s * KS) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        // GS tap steps form one scheduling group: [the group's LDS reads, D steps ahead][its share of the expansion MFMAs][its
        // packed FMAs] — an MFMA <-> VALU switch costs ~8 cycles, so FMAs and MFMAs are issued in runs
        constexpr int GS = FEAR_V4_GS;
        static_for<0, (NS + GS - 1) / GS>([&](auto Gi) {
            constexpr int g0 = decltype(Gi)::value * GS, g1 = (g0 + GS < NS) ? g0 + GS : NS;
            f32x4 eg[GS];
            static_for<g0, g1>([&](auto S) {
                constexpr int s = decltype(S)::value;
                eg[s - g0] = ev[s % D];
                if constexpr (s + D < NS) {
                    constexpr int kx2 = (s + D) / IR, iy2 = (s + D) % IR;
                    ev[s % D] = *reinterpret_cast<const f32x4*>(Ebase + (iy2 * IWR + kx2) * EPX);
                    if constexpr (iy2 < KS) wv[kx2 & 1][iy2] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 16);
                }
            });
            if (more) {
                static_for<g0 * NU / NS, g1 * NU / NS>([&](auto U) {
                    // units in m-tile pairs: (i0, k0), (i1, k0), (i0, k1), (i1, k1), ... so consecutive MFMAs never share an accumulator
                    constexpr int u = decltype(U)::value;
                    constexpr int pair = u / (2 * MPT), w = u % (2 * MPT);
                    constexpr bool single = 2 * pair + 1 >= MTA;               // the last m-tile of an odd MTA stands alone
                    constexpr int i = single ? 2 * pair : 2 * pair + (w & 1);
                    constexpr int k = single ? w : w >> 1;                      // k-step index within the m-tile, 0 .. MPT - 1
                    if constexpr (!(single && w >= MPT)) {
                        constexpr int kg = k / 4, q = k % 4;
                        if (!FEAR_V4_GUARD || i < cnt) eacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kg][q], xf[i][kg][q], eacc[i], 0, 0, 0);
                    }
                });
            }
            static_for<g0, g1>([&](auto S) {
                constexpr int s = decltype(S)::value;
                constexpr int kx = s / IR, iy = s % IR;
                static_for<0, MTC>([&](auto R) {
                    constexpr int r = decltype(R)::value;
                    constexpr int ky = iy - r * ST;
                    if constexpr (ky >= 0 && ky < KS) pk_fma4(d[r], eg[s - g0], wv[kx & 1][ky]);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if (a.relu_dw) {
#pragma unroll
            for (int r = 0; r < MTC; ++r) relu4(d[r]);
        }
        asm volatile("s_nop 7");                // the inline-asm FMA results feed MFMAs below (hipcc's hazard recognizer does not see them)
        __syncthreads();                       // every wave has read chunk c out of E
        // ---- interval 2: chunk c + 1 into E, projection of chunk c
        if (more) {
            static_for<0, MTA>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (i < cnt) {
                    relu4(eacc[i]);
                    *reinterpret_cast<f32x4*>(E + eoff[i]) = eacc[i];
                }
            });
        }
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(wb + nt * 256 + lane * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < MTC; ++r) accp[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[q], d[r][q], accp[r][nt], 0, 0, 0);
        }
        if (more) __syncthreads();             // E holds chunk c + 1; the staged weights have landed
    }

    // stores as [scalar row base + 32-bit lane offset], ReLU as a max against 0 / -big (see ir_tile_v2_kernel's epilogue)
    const long m0 = (crop * Ho + oy0 + r0) * Wo + ox0 + seg * 16;
    const unsigned ylane = (unsigned)(li * a.ldy + lk * 4), rlane = (unsigned)(li * a.ldr + lk * 4);
    const float relu_lo = a.relu_out ? 0.f : -3.0e38f;
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        if (nt * 16 >= COUT) continue;
        const bool n_ok = COUT % 16 == 0 || nt * 16 + lk * 4 < COUT;
#pragma unroll
        for (int r = 0; r < MTC; ++r) {
            const long mrow = m0 + (long)r * Wo;
            f32x4 v = accp[r][nt] + bpv[nt];
            if (a.R && n_ok) v += *reinterpret_cast<const f32x4*>(a.R + (mrow * a.ldr + nt * 16 + (long)rlane));
            v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);
            if (n_ok) *reinterpret_cast<f32x4*>(a.Y + (mrow * a.ldy + nt * 16 + (long)ylane)) = v;
        }
    }
}

// ================================================================================================
// ir16h: the fused 16x16 block on the REAL matrix pipe.
//
// tools/coexec.hip shows that v_mfma_f32_16x16x4_f32 executes on the vector ALUs (an fp32-MFMA wave and a VALU
// wave on one SIMD take the SUM of their times), so in ir16v2 the depthwise VALU work can never hide under the
// pointwise MFMAs.  The f16 MFMA (v_mfma_f32_16x16x32_f16) runs on the dedicated matrix pipe at 16x the rate and
// co-executes with the VALU.  FEAR's weights ARE fp16 numbers (the model ships as fp16), so a pointwise conv can
// be computed as            W . x  =  W . hi(x) + W . lo(x),   hi = fp16(x),  lo = fp16(x - hi)
// with exact fp16 x fp16 products accumulated in fp32 by two matrix-pipe MFMAs: the activation is represented to
// 2^-22 relative (fp32 itself rounds every product to 2^-24), the accumulation stays fp32, and everything that is
// not a GEMM (bias, ReLU, the depthwise conv, residual adds) stays plain fp32 on the VALU.  Measured deviation
// from the exact fp32-MFMA path is ~1e-6 relative (same size as the difference between two fp32 summation
// orders); the tolerance of the path is 1e-3.  Range assumption: |activation| < 65504 (fp16 max).
//
// Geometry differs from ir16v2 only by the K granularity of the MFMA (32 instead of 4): chunks of CE = 32
// expanded channels, a lane owns 8 consecutive channels (two float4) of its pixel.
//   A-part per chunk: 2 n-tiles x KG32 fragments (64 lanes x 8 halfs: We[c0 + nt*16 + l&15][kg*32 + 8*(l>>4) + j])
//                     | be[32] fp32
//   BC-part         : NTP fragments (Wp[nt*16 + l&15][c0 + 8*(l>>4) + j]) | Wd[KS*KS][32] fp32 | bd[32] fp32
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int CIN, int CEXP, int COUT, int KS, bool EXPAND>
struct IrHGeom {
    // LDS tile of the expanded activations: pixel (y, x) of the 16x16 map lives at index (y + P) * RW + x + P with
    // RW = 16 + P: ONE zero gutter of P columns per row serves as the right padding of row y and the left padding of
    // row y + 1 (a tap that runs off the right edge lands in the next row's gutter).  Per pixel 40 floats =
    // [h = channel half][lk = owning lane group][4] (+8 pad): a lane's two 16-B reads per tap are a channel-half
    // plane apart, which makes every ds_read_b128 service group conflict free (pixel stride 10 units, lk stride 1).
    static constexpr int CE = 32, S = 16, P = KS / 2, RW = S + P, ES = 40;
    static constexpr int NCHUNK = (CEXP + CE - 1) / CE, NTP = COUT / 16, KG = EXPAND ? (CIN + 31) / 32 : 0;
    // stage sizes in floats (a fragment = 64 lanes x 16 B = 256 floats)
    static constexpr int AP = EXPAND ? 2 * KG * 256 + 32 : 0;
    static constexpr int BP = NTP * 256 + KS * KS * 32 + 32;
    static constexpr int EBUF = ((S + 2 * P) * RW + P) * ES;
    static constexpr int LDS_FLOATS = 2 * EBUF + 2 * AP + 2 * BP;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_half8(const f32x4& a, const f32x4& b, h8& hi, h8& lo) {
    const f32x8 x = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    hi = __builtin_convertvector(x, h8);
    lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x8), h8);
}

// The two matrix-pipe arithmetic modes of the *_h kernels (FEAR_OPT_MATH):
//   MM = 1  fp32 activation = fp16 hi + fp16 lo, exact-fp16 weights, two v_mfma_f32_16x16x32_f16 per product: fp32-grade
//   MM = 2  activation and weights rounded to bf16 (RNE, v_cvt_pk_bf16_f32), ONE v_mfma_f32_16x16x32_bf16, fp32
//           accumulate: genuinely reduced precision (8 mantissa bits per operand) — the "bf16 MFMA pointwise-conv path" of
//           BASELINE.json configs[3]; bias, ReLU, depthwise, residuals stay fp32 as in mode 1
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int MM> struct MatOps;
template <> struct MatOps<1> {
    using V = h8;
    static __device__ __forceinline__ void split(const f32x4& a, const f32x4& b, V& hi, V& lo) { split_half8(a, b, hi, lo); }
    static __device__ __forceinline__ f32x4 mma(const V& w, const V& hi, const V& lo, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, hi, acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, lo, acc, 0, 0, 0);
    }
    // both operands are activations (the correlation): (whi + wlo)(xhi + xlo) without the lo x lo term (2^-22 relative)
    static __device__ __forceinline__ f32x4 mma2(const V& whi, const V& wlo, const V& hi, const V& lo, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, lo, acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, hi, acc, 0, 0, 0);
    }
};
template <> struct MatOps<2> {
    using V = bf8;
    static __device__ __forceinline__ void split(const f32x4& a, const f32x4& b, V& hi, V& lo) {
        hi = __builtin_convertvector(__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7), bf8);
        lo = hi;      // unused
    }
    static __device__ __forceinline__ f32x4 mma(const V& w, const V& hi, const V&, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hi, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma2(const V& whi, const V&, const V& hi, const V&, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, hi, acc, 0, 0, 0);
    }
};

// ------------------------------------------------------------------------------------------------
// pw_mfma_kernel's GEMM (the neck, the pixel-wise correlation, any pointwise conv without a fused block) on the matrix pipe
// for FEAR_OPT_MATH = 1 / 2: same workgroup tiling and epilogue, k in groups of 32 (lane (li, lk) holds 8 consecutive input
// channels of its pixel / output channel), MatOps arithmetic — MM = 1: activations as fp16 hi + lo against the exact-fp16
// weights (two MFMAs), and for the correlation (WKN: the "weights" are the crop's template features, activations too) both
// operands split, three MFMAs; MM = 2: both operands rounded to bf16, one MFMA.  fp32 accumulate, fp32 epilogue.
// In those modes the fp32 kernel was 9 % of the fp16-split step and 6 % of the bf16 FEAR-M step for 1.2 % of the FLOPs.
template <int MT, int NT, bool WKN, int MM>
__global__ __launch_bounds__(256) void pw_h_kernel(PwArgs a) {
    using MX = MatOps<MM>;
    using V8 = typename MX::V;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int m_wave = (blockIdx.x * 4 + wave) * (MT * 16);
    if (m_wave >= a.M) return;
    const float* Wp = a.W;
    if (a.rows_per_crop > 0) Wp += (long)(m_wave / a.rows_per_crop) * a.w_crop_stride;   // per-crop weight matrices
    const float* xrow[MT];
    bool mvalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = m_wave + mt * 16 + li;
        mvalid[mt] = m < a.M;
        if (m >= a.M) m = a.M - 1;
        xrow[mt] = a.X + (long)m * a.ldx;
    }
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n_tiles = (a.N + 15) >> 4;
    for (int nc = blockIdx.y * NT; nc < n_tiles; nc += NT * gridDim.y) {
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero;
        int nrow[NT];
        bool nvalid[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = (nc + nt) * 16 + li;
            nvalid[nt] = n < a.N;
            nrow[nt] = nvalid[nt] ? n : (a.N - 1);
        }
#pragma unroll 2
        for (int kg = 0; kg < a.K; kg += 32) {
            const int k = kg + lk * 8;
            const bool k0v = k < a.K, k1v = k + 4 < a.K;           // K is a multiple of 4
            V8 xhi[MT], xlo[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 v0 = k0v ? *reinterpret_cast<const f32x4*>(xrow[mt] + k) : zero;
                const f32x4 v1 = k1v ? *reinterpret_cast<const f32x4*>(xrow[mt] + k + 4) : zero;
                MX::split(v0, v1, xhi[mt], xlo[mt]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 w0 = zero, w1 = zero;
                if (nvalid[nt]) {
                    if (WKN) {
                        const float* p = Wp + (long)k * a.N + nrow[nt];
                        if (k0v) w0 = (f32x4){p[0], p[a.N], p[2 * a.N], p[3 * a.N]};
                        if (k1v) w1 = (f32x4){p[4 * (long)a.N], p[5 * (long)a.N], p[6 * (long)a.N], p[7 * (long)a.N]};
                    } else {
                        const float* p = Wp + (long)nrow[nt] * a.K + k;
                        if (k0v) w0 = *reinterpret_cast<const f32x4*>(p);
                        if (k1v) w1 = *reinterpret_cast<const f32x4*>(p + 4);
                    }
                }
                V8 whi, wlo;
                MX::split(w0, w1, whi, wlo);      // conv weights are exact fp16 numbers: their lo half is zero (MM = 1) / unused (MM = 2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = WKN ? MX::mma2(whi, wlo, xhi[mt], xlo[mt], acc[mt][nt]) : MX::mma(whi, xhi[mt], xlo[mt], acc[mt][nt]);
            }
        }
        // epilogue: lane holds channels n0 + 4*lk + {0..3} of pixel m0 + li
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = (nc + nt) * 16 + lk * 4;
            if (n >= a.N) continue;   // N is a multiple of 4
            f32x4 b = zero;
            if (a.bias) b = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (!mvalid[mt]) continue;
                const long m = m_wave + mt * 16 + li;
                f32x4 v = acc[mt][nt] + b;
                if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (a.nchw_hw > 0) {
                    const long crop = m / a.nchw_hw, px = m % a.nchw_hw;
                    float* y = a.Y + (crop * a.N + n) * a.nchw_hw + px;
                    y[0] = v.x; y[a.nchw_hw] = v.y; y[2 * (long)a.nchw_hw] = v.z; y[3 * (long)a.nchw_hw] = v.w;
                } else {
                    *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
                }
            }
        }
    }
}

template <int CIN, int CEXP, int COUT, int KS, bool EXPAND, int MM = 1>
__global__ __launch_bounds__(512) void ir16h_fused_kernel(Ir2Args a) {
    using MX = MatOps<MM>;
    using V8 = typename MX::V;
    using G = IrHGeom<CIN, CEXP, COUT, KS, EXPAND>;
    constexpr int S = G::S, P = G::P, PW = G::RW, ES = G::ES, NCHUNK = G::NCHUNK, NTP = G::NTP, KG = G::KG;
    constexpr int AP = G::AP, BP = G::BP, EBUF = G::EBUF, CST = AP + BP;
    constexpr int AP4 = AP / 4, BP4 = BP / 4;
    constexpr int NRA = (AP4 + 511) / 512, NRB = (BP4 + 511) / 512;
    static_assert(COUT % 16 == 0 && (EXPAND || CIN == CEXP), "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Ebuf = lds;                    // [2][EBUF]
    float* const WA = lds + 2 * EBUF;           // [2][AP]
    float* const WB = WA + 2 * AP;              // [2][BP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long crop = blockIdx.x;
    const float* Xc = a.X + crop * 256 * a.ldx;
    const int y0 = wave * 2;

    // the tile's zero ring (= the convolution's padding): the P rows above and below the map, the P gutter columns of every row and
    // the tail — 106 of the 362 pixel slots at k = 5.  The 256 interior pixels are written by every chunk before they are read
    // (all 32 channels of a chunk: padded channels carry zero weights), so they are not cleared: the full clear was 14 LDS stores
    // per thread at the head of every launch, 5 us of an 83 us workgroup in the bf16 mode.
    constexpr int NPE = (S + 2 * P) * PW + P, ES4 = ES / 4;
    for (int i = tid; i < 2 * NPE * ES4; i += 512) {
        const int b = i / (NPE * ES4), r = i - b * (NPE * ES4);
        const int pe = r / ES4, row = pe / PW, col = pe - row * PW;
        if (row >= P && row < P + S && col >= P) continue;          // interior
        *reinterpret_cast<f32x4*>(lds + b * EBUF + r * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // activation fragments (hi / lo halves) of this wave's two pixel rows, resident for every chunk
    V8 xhi[EXPAND ? 2 : 1][EXPAND ? KG : 1], xlo[EXPAND ? 2 : 1][EXPAND ? KG : 1];
    if (EXPAND) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int k = kg * 32 + lk * 8;
                const float* px = Xc + (long)((y0 + mt) * S + li) * a.ldx + k;
                f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (k < CIN) v0 = *reinterpret_cast<const f32x4*>(px);
                if (k + 4 < CIN) v1 = *reinterpret_cast<const f32x4*>(px + 4);
                MX::split(v0, v1, xhi[mt][kg], xlo[mt][kg]);
            }
    }

    f32x4 ra[EXPAND ? (NRA > 0 ? NRA : 1) : 4], rb[NRB];
    auto load_a = [&](int c) {      // EXPAND: A-part of chunk c; !EXPAND: this lane's 8 channels x 2 pixels of chunk c
        if (EXPAND) {
#pragma unroll
            for (int r = 0; r < NRA; ++r) {
                const int idx = tid + r * 512;
                if (idx < AP4) ra[r] = *reinterpret_cast<const f32x4*>(a.Wpk + (long)c * CST + idx * 4);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int k = c * 32 + lk * 8;
                const float* px = Xc + (long)((y0 + mt) * S + li) * a.ldx + k;
                ra[2 * mt] = ra[2 * mt + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (k < CIN) ra[2 * mt] = *reinterpret_cast<const f32x4*>(px);
                if (k + 4 < CIN) ra[2 * mt + 1] = *reinterpret_cast<const f32x4*>(px + 4);
            }
        }
    };
    auto store_a = [&](int c) {
        if (EXPAND) {
            float* dst = WA + (c & 1) * AP;
#pragma unroll
            for (int r = 0; r < NRA; ++r) {
                const int idx = tid + r * 512;
                if (idx < AP4) *reinterpret_cast<f32x4*>(dst + idx * 4) = ra[r];
            }
        } else {
            float* E = Ebuf + (c & 1) * EBUF;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float* e = E + ((y0 + mt + P) * PW + li + P) * ES + lk * 4;      // [h][lk][4]: h = 0 | h = 1 at +16
                *reinterpret_cast<f32x4*>(e) = ra[2 * mt];
                *reinterpret_cast<f32x4*>(e + 16) = ra[2 * mt + 1];
            }
        }
    };
    auto load_b = [&](int c) {
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            const int idx = tid + r * 512;
            if (idx < BP4) rb[r] = *reinterpret_cast<const f32x4*>(a.Wpk + (long)c * CST + AP + idx * 4);
        }
    };
    auto store_b = [&](int c) {
        float* dst = WB + (c & 1) * BP;
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            const int idx = tid + r * 512;
            if (idx < BP4) *reinterpret_cast<f32x4*>(dst + idx * 4) = rb[r];
        }
    };

    // phase A: E[c] <- relu(We_chunk . (x_hi + x_lo) + be): 2 n-tiles x 2 pixel rows, matrix pipe
    auto phase_a = [&](int c) {
        const float* wa = WA + (c & 1) * AP;
        float* E = Ebuf + (c & 1) * EBUF;
        f32x4 acc[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(wa + 2 * KG * 256 + nt * 16 + lk * 4);
            acc[0][nt] = bias;
            acc[1][nt] = bias;
        }
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const V8 wf = *reinterpret_cast<const V8*>(wa + (nt * KG + kg) * 256 + lane * 4);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (FEAR_ABL & 8) { acc[mt][nt].x += (float)wf[0]; continue; }
                    acc[mt][nt] = MX::mma(wf, xhi[mt][kg], xlo[mt][kg], acc[mt][nt]);
                }
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 v = acc[mt][nt];
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                *reinterpret_cast<f32x4*>(E + ((y0 + mt + P) * PW + li + P) * ES + (lk & 1) * 16 + (2 * nt + (lk >> 1)) * 4) = v;
            }
    };

    f32x4 accp[2][NTP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) accp[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // phase B (fp32 VALU depthwise, 8 channels x 2 rows per lane) + phase C (projection on the matrix pipe)
    auto phase_bc = [&](int c) {
        const float* E = Ebuf + (c & 1) * EBUF;
        const float* wb = WB + (c & 1) * BP;
        const float* wd = wb + NTP * 256 + lk * 8;
        f32x4 d0[2], d1[2];
        if constexpr (MM == 2 && EXPAND) {      // (the head's SepConvs carry 16 output tiles of accumulators: no room for the rings)
            // The depthwise as ir16_interval runs it (round 6): a chain of KS (KS + 1) tap steps (kx outer, input row inner: the weight of
            // (iy, kx) feeds row 0 now and row 1 in the next step), both channel halves of the lane side by side, the two LDS reads of a
            // step and half — activation and tap — issued IR16H_D steps ahead of the packed FMAs that consume them, sched_barrier(0)
            // after every step so that hipcc keeps the order.  Round 2's form (an 8-wide accumulator per half, hipcc's own schedule:
            // read -> wait -> FMAs, tap weights re-read per column) ran the 112 -> 672 -> 112 block at 177 us per 512 crops against
            // 28 us of packed FMAs and ~60 us of LDS reads: latency-bound at two waves per SIMD.  Same products in the same
            // order: the block's output is the same bits.
            constexpr int NS = KS * (KS + 1), D = IR16H_D;
            static_assert(NS >= D, "read-ahead");
            f32x4 wprev[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                d0[h] = *reinterpret_cast<const f32x4*>(wd + KS * KS * 32 + h * 4);
                d1[h] = d0[h];
                wprev[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const float* e0 = E + (y0 * PW + li) * ES + lk * 4;
            f32x4 ev[D][2], wv[D][2];
#pragma unroll
            for (int t = 0; t < D; ++t) {
                const int kx = t / (KS + 1), iy = t % (KS + 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ev[t][h] = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * ES + h * 16);
                    if (iy < KS) wv[t][h] = *reinterpret_cast<const f32x4*>(wd + (iy * KS + kx) * 32 + h * 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int iy = t % (KS + 1);
                f32x4 e[2], w[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    e[h] = ev[t % D][h];
                    w[h] = wv[t % D][h];
                    if (t + D < NS && !(FEAR_ABL & 4)) {
                        const int kx2 = (t + D) / (KS + 1), iy2 = (t + D) % (KS + 1);
                        ev[t % D][h] = *reinterpret_cast<const f32x4*>(e0 + (iy2 * PW + kx2) * ES + h * 16);
                        if (iy2 < KS) wv[t % D][h] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 32 + h * 4);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (FEAR_ABL & 32) { d0[h].x += e[h].x + w[h].x; continue; }
                    if (iy < KS) pk_fma4(d0[h], e[h], w[h]);
                    if (iy >= 1) pk_fma4(d1[h], e[h], wprev[h]);
                    wprev[h] = w[h];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 7" : "+v"(d0[0]), "+v"(d0[1]), "+v"(d1[0]), "+v"(d1[1]));      // (inline-asm FMA results feed conversions / MFMAs: pk_fma_settle)
        } else {
            // (the fp16-split mode keeps round 2's form: with the lo halves of every operand resident it sits at 254 registers, and the
            //  read-ahead rings of the pipelined form cost it 3 % — 147.0 k -> 143.0 k crops/s on FEAR-XS, same box)
            // Both output rows of a lane are carried in ONE 8-wide accumulator per channel half ({row0 x4, row1 x4}):
            // an input row iy feeds row 0 with tap row iy and row 1 with tap row iy-1, i.e. one 8-wide FMA with the
            // activation duplicated.  (Written as two float4 chains, hipcc computes the chains in two passes and spills
            // every LDS value in between: ~500 VGPRs of scratch traffic.)
            f32x8 d8[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 bd = *reinterpret_cast<const f32x4*>(wd + KS * KS * 32 + h * 4);
                d8[h] = __builtin_shufflevector(bd, bd, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const float* e0 = E + (y0 * PW + li) * ES + lk * 4;
            const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            // column-outer; the two channel halves are two independent 8-wide chains interleaved for ILP
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                f32x4 w[2][KS];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky) w[h][ky] = *reinterpret_cast<const f32x4*>(wd + (ky * KS + kx) * 32 + h * 4);
#pragma unroll
                for (int iy = 0; iy < KS + 1; ++iy) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * ES + h * 16);
                        const f32x8 v8 = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7);
                        const f32x8 w8 = __builtin_shufflevector(iy < KS ? w[h][iy < KS ? iy : 0] : zero4,
                                                                 iy >= 1 ? w[h][iy >= 1 ? iy - 1 : 0] : zero4, 0, 1, 2, 3, 4, 5, 6, 7);
                        d8[h] += v8 * w8;
                    }
                }
            }
            d0[0] = __builtin_shufflevector(d8[0], d8[0], 0, 1, 2, 3); d1[0] = __builtin_shufflevector(d8[0], d8[0], 4, 5, 6, 7);
            d0[1] = __builtin_shufflevector(d8[1], d8[1], 0, 1, 2, 3); d1[1] = __builtin_shufflevector(d8[1], d8[1], 4, 5, 6, 7);
        }
        V8 dhi[2], dlo[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 q0 = r == 0 ? d0[0] : d1[0];
            f32x4 q1 = r == 0 ? d0[1] : d1[1];
            if (a.relu_dw) {
                q0.x = fmaxf(q0.x, 0.f); q0.y = fmaxf(q0.y, 0.f); q0.z = fmaxf(q0.z, 0.f); q0.w = fmaxf(q0.w, 0.f);
                q1.x = fmaxf(q1.x, 0.f); q1.y = fmaxf(q1.y, 0.f); q1.z = fmaxf(q1.z, 0.f); q1.w = fmaxf(q1.w, 0.f);
            }
            MX::split(q0, q1, dhi[r], dlo[r]);
        }
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) {
            const V8 wp = *reinterpret_cast<const V8*>(wb + nt * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (FEAR_ABL & 16) { accp[r][nt].x += (float)wp[0] + (float)dhi[r][0]; continue; }
                accp[r][nt] = MX::mma(wp, dhi[r], dlo[r], accp[r][nt]);
            }
        }
    };

    if constexpr (EXPAND && IR16H_ASYNC) {
        // The packed weights go global -> LDS by asynchronous copies (global_load_lds_dwordx4: no staging registers, no ds_write),
        // as in chain16_block: issued at the top of an interval into the stage the previous interval read last, complete at the
        // interval's barrier.  (Round 2's form — two or three loads and ds_write_b128 per thread and chunk through `ra` / `rb` —
        // stays for the blocks without expansion, whose A part is the activation itself.)
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        auto stage_a = [&](int c) { lds_copy_async<AP>(a.Wpk + (long)c * CST, WA + (c & 1) * AP, wave_s, lane); };
        auto stage_b = [&](int c) { lds_copy_async<BP>(a.Wpk + (long)c * CST + AP, WB + (c & 1) * BP, wave_s, lane); };
        stage_a(0);
        stage_b(0);
        if (NCHUNK > 1) stage_a(1);
        __syncthreads();                   // (also: the zero ring is in place)
        phase_a(0);
        __syncthreads();
        for (int c = 0; c < NCHUNK - 1; ++c) {
            if (c + 2 < NCHUNK) stage_a(c + 2);
            stage_b(c + 1);
            phase_a(c + 1);
            phase_bc(c);
            __syncthreads();
        }
        phase_bc(NCHUNK - 1);
        __syncthreads();
    } else {
    // ---- prologue: stage A(0), A(1), BC(0); produce E[0]
    load_a(0);
    load_b(0);
    __syncthreads();
    store_a(0);
    store_b(0);
    if (EXPAND) {
        if (NCHUNK > 1) { load_a(1); store_a(1); }
        __syncthreads();
        phase_a(0);
    }
    __syncthreads();

    // (the last chunk is peeled: no run-time branch around the expansion phase inside the loop — chain16_block's change)
    for (int c = 0; c < NCHUNK - 1; ++c) {
        const int ca = EXPAND ? c + 2 : c + 1;
        if (ca < NCHUNK) load_a(ca);
        load_b(c + 1);
        if (EXPAND) phase_a(c + 1);
        phase_bc(c);
        if (ca < NCHUNK) store_a(ca);
        store_b(c + 1);
        __syncthreads();
    }
    phase_bc(NCHUNK - 1);
    __syncthreads();
    }

    if (a.pred_cout > 0) {          // prediction head: lanes lk == 0 hold channels 0..3 of their pixel
        if (lk == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int px = (y0 + mt) * S + li;
                const f32x4 v = accp[mt][0];
                const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (n < a.pred_cout) {
                        float o = vals[n] + a.bp[n];
                        if (a.pred_act == 2) o = expf(o);
                        a.Y[crop * a.pred_stride + n * 256 + px] = o;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        const int n = nt * 16 + lk * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bp + n);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long m = crop * 256 + (y0 + mt) * S + li;
            f32x4 v = accp[mt][nt] + b;
            if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + n);
            if (a.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!(FEAR_ABL & 2048) || v.x == 1234.5f) *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + n) = v;
        }
    }
}

// ================================================================================================
// ir_tile_h: the spatially tiled fused block (ir_tile_v2_kernel) on the f16 matrix pipe with hi+lo split
// activations (see ir16h_fused_kernel for the arithmetic).  CE = 32 channels per chunk, a lane owns 8 channels
// of its pixel; NW waves per workgroup (4 or 8) so that small tiles keep the LDS tile small enough for several
// workgroups per CU.  Stride-2 blocks use one output row per wave (MTC = 1); stride-1 blocks process output
// rows in pairs with the 8-wide accumulator formulation.
template <int CIN, int CEXPP, int COUT, int KS, int ST, int TW, int TH, bool EXPAND, int NW>
struct IrTHGeom {
    static constexpr int CE = 32, P = KS / 2, IWR = (TW - 1) * ST + KS, IHR = (TH - 1) * ST + KS, ES = CE + 4;
    static constexpr int SEG = TW / 16, NMT_OUT = TH * SEG, MTC = NMT_OUT / NW;
    static constexpr int NMT_IN_MAX = (IHR * IWR + 15) / 16, MTA = (NMT_IN_MAX + NW - 1) / NW;
    static constexpr int NCHUNK = CEXPP / CE, NTP = (COUT + 15) / 16, KG = EXPAND ? (CIN + 31) / 32 : 0;
    static constexpr int AP = EXPAND ? 2 * KG * 256 + 32 : 0;
    static constexpr int BP = NTP * 256 + KS * KS * 32 + 32;
    static constexpr int EBUF = IHR * IWR * ES;
    static constexpr int LDS_BYTES = (EBUF + 2 * (AP + BP)) * 4;
};

template <int CIN, int CEXPP, int COUT, int KS, int ST, int TW, int TH, bool EXPAND, int NW, int MINW, int MM = 1, int IO = 0>
__global__ __launch_bounds__(64 * NW, MINW) void ir_tile_h_kernel(IrT2Args t) {
    static_assert(IO == 0 || (MM == 2 && EXPAND && (IO & IO_R_BF16) == 0 && CIN % 8 == 0), "bf16 storage: input / output of the bf16-mode expand blocks");
    using MX = MatOps<MM>;
    using V8 = typename MX::V;
    using G = IrTHGeom<CIN, CEXPP, COUT, KS, ST, TW, TH, EXPAND, NW>;
    const Ir2Args& a = t.b;
    constexpr int P = G::P, IWR = G::IWR, IHR = G::IHR, ES = G::ES, SEG = G::SEG, MTC = G::MTC, MTA = G::MTA;
    constexpr int NCHUNK = G::NCHUNK, NTP = G::NTP, KG = G::KG, AP = G::AP, BP = G::BP, EBUF = G::EBUF;
    constexpr int NT = 64 * NW, CST = AP + BP, W4 = CST / 4, NRW = (W4 + NT - 1) / NT;
    static_assert(CEXPP % 32 == 0 && G::NMT_OUT % NW == 0 && (SEG == 1 || SEG == 2) && NW % SEG == 0, "tile shape");
    static_assert(ST == 1 ? (MTC % 2 == 0) : (MTC == 1), "stride-1: row pairs; stride-2: one row per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const E = lds;               // [EBUF]
    float* const WS = lds + EBUF;       // [2][AP + BP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int tiles = t.tiles_x * t.tiles_y;
    const unsigned tix = xcd_tile_index(blockIdx.x, gridDim.x);
    const long crop = tix / tiles;
    const int tile = tix % tiles;
    const int ox0 = (tile % t.tiles_x) * TW, oy0 = (tile / t.tiles_x) * TH;
    const int ix0 = ox0 * ST - P, iy0 = oy0 * ST - P;
    const int cx_lo = max(ix0, 0), cy_lo = max(iy0, 0);
    const int CW = min(ix0 + IWR, t.W) - cx_lo, CH = min(iy0 + IHR, t.H) - cy_lo;
    const int NPIX = CW * CH;
    const float inv_cw = __builtin_amdgcn_rcpf((float)CW);      // 1 ulp is plenty: (q + 0.5) / CW stays 0.5 / CW away from every integer
    const int Wo = t.W / ST, Ho = t.H / ST;
    const float* Xc = a.X + crop * t.H * t.W * a.ldx;

    for (int i = tid * 4; i < EBUF; i += NT * 4) *reinterpret_cast<f32x4*>(E + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 rw[NRW];
    auto load_w = [&](int c) {
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
            const int idx = tid + r * NT;
            if (idx < W4) rw[r] = *reinterpret_cast<const f32x4*>(a.Wpk + (long)c * CST + idx * 4);
        }
    };
    auto store_w = [&](int c) {
        float* dst = WS + (c & 1) * CST;
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
            const int idx = tid + r * NT;
            if (idx < W4) *reinterpret_cast<f32x4*>(dst + idx * 4) = rw[r];
        }
    };
    load_w(0);

    int eoff[MTA];
    unsigned xoff[MTA];
    V8 xhi[EXPAND ? MTA : 1][EXPAND ? KG : 1], xlo[EXPAND ? MTA : 1][EXPAND ? KG : 1];
    {   // m-tile i is 16 * NW pixels on from m-tile i - 1: stepped (see ir_tile_v2_kernel); loads as [scalar crop base + lane offset]
        constexpr int QS = 16 * NW;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const int q0 = wave_u * 16 + li;
        const int cy0 = (int)(((float)q0 + 0.5f) * inv_cw), cx0 = q0 - cy0 * CW;   // exact for q0 < 2^16, CW <= 64 (garbage past NPIX: masked)
        const int qa = QS / CW, qb = QS - qa * CW;
        const int dpe0 = qa * IWR + qb, dpe1 = dpe0 + IWR - CW;
        const int dpx0 = (qa * t.W + qb) * a.ldx, dpx1 = dpx0 + (t.W - CW) * a.ldx;
        int cx = cx0;
        int pe = (cy0 + cy_lo - iy0) * IWR + cx0 + cx_lo - ix0;
        int px = ((cy0 + cy_lo) * t.W + cx0 + cx_lo) * a.ldx;
        const unsigned short* Xb = reinterpret_cast<const unsigned short*>(a.X) + crop * t.H * t.W * a.ldx;
#pragma unroll
        for (int i = 0; i < MTA; ++i) {
            const bool valid = q0 < NPIX - QS * i;
            eoff[i] = valid ? pe * ES : -1;
            xoff[i] = valid ? (unsigned)px : 0u;
            if (EXPAND) {
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    const int k = kg * 32 + lk * 8;
                    if (IO & IO_X_BF16) {
                        // the stored activations ARE the bf16 operands: eight channels = one 16-byte load, no conversion
                        uint4 u = (uint4){0u, 0u, 0u, 0u};
                        if (k < CIN) u = *reinterpret_cast<const uint4*>(Xb + (xoff[i] + (unsigned)k));
                        xhi[i][kg] = __builtin_bit_cast(V8, u);
                        xlo[i][kg] = xhi[i][kg];
                        continue;
                    }
                    f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
                    if (k < CIN) v0 = *reinterpret_cast<const f32x4*>(Xc + (xoff[i] + (unsigned)k));
                    if (k + 4 < CIN) v1 = *reinterpret_cast<const f32x4*>(Xc + (xoff[i] + (unsigned)(k + 4)));
                    MX::split(v0, v1, xhi[i][kg], xlo[i][kg]);
                }
            }
            if (i + 1 < MTA) {
                cx += qb;
                const bool wrap = cx >= CW;
                cx -= wrap ? CW : 0;
                pe += wrap ? dpe1 : dpe0;
                px += wrap ? dpx1 : dpx0;
            }
        }
    }
    f32x4 rx[EXPAND ? 1 : 2 * MTA];
    auto load_x = [&](int c) {
        if (!EXPAND) {
#pragma unroll
            for (int i = 0; i < MTA; ++i) {
                const int k = c * 32 + lk * 8;
                rx[2 * i] = rx[2 * i + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (k < CIN) rx[2 * i] = *reinterpret_cast<const f32x4*>(Xc + xoff[i] + k);
                if (k + 4 < CIN) rx[2 * i + 1] = *reinterpret_cast<const f32x4*>(Xc + xoff[i] + k + 4);
            }
        }
    };
    load_x(0);
    __syncthreads();
    store_w(0);

    const int seg = SEG == 1 ? 0 : (wave & 1);
    const int r0 = (SEG == 1 ? wave : (wave >> 1)) * MTC;
    f32x4 accp[MTC][NTP];
#pragma unroll
    for (int r = 0; r < MTC; ++r)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) accp[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < NCHUNK; ++c) {
        const float* wa = WS + (c & 1) * CST;
        const float* wb = wa + AP;
        if (EXPAND) __syncthreads();
        // ---- phase A
        if (EXPAND) {
            V8 wf[2][KG > 0 ? KG : 1];
            f32x4 bias[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bias[nt] = *reinterpret_cast<const f32x4*>(wa + 2 * KG * 256 + nt * 16 + lk * 4);
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) wf[nt][kg] = *reinterpret_cast<const V8*>(wa + (nt * KG + kg) * 256 + lane * 4);
            }
#pragma unroll
            for (int i = 0; i < MTA; ++i) {
                if ((wave + NW * i) * 16 >= NPIX) break;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 acc = bias[nt];
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) {
                        acc = MX::mma(wf[nt][kg], xhi[i][kg], xlo[i][kg], acc);
                    }
                    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
                    if (eoff[i] >= 0) *reinterpret_cast<f32x4*>(E + eoff[i] + nt * 16 + lk * 4) = acc;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MTA; ++i) {
                if ((wave + NW * i) * 16 >= NPIX) break;
                if (eoff[i] >= 0) {
                    *reinterpret_cast<f32x4*>(E + eoff[i] + lk * 8) = rx[2 * i];
                    *reinterpret_cast<f32x4*>(E + eoff[i] + lk * 8 + 4) = rx[2 * i + 1];
                }
            }
        }
        if (c + 1 < NCHUNK) { load_w(c + 1); load_x(c + 1); }
        __syncthreads();
        // ---- phase B: fp32 depthwise, 8 channels per lane
        const float* wd = wb + NTP * 256 + lk * 8;
        const float* Ebase = E + ((r0 * ST) * IWR + (seg * 16 + li) * ST) * ES + lk * 8;
        V8 dhi[MTC], dlo[MTC];
        if (ST == 1) {
#pragma unroll
            for (int pr = 0; pr < MTC / 2; ++pr) {        // output rows 2pr, 2pr+1 in one 8-wide accumulator per half
                f32x8 d8[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 bd = *reinterpret_cast<const f32x4*>(wd + KS * KS * 32 + h * 4);
                    d8[h] = __builtin_shufflevector(bd, bd, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        f32x4 w[KS];
#pragma unroll
                        for (int ky = 0; ky < KS; ++ky) w[ky] = *reinterpret_cast<const f32x4*>(wd + (ky * KS + kx) * 32 + h * 4);
#pragma unroll
                        for (int iy = 0; iy < KS + 1; ++iy) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(Ebase + ((2 * pr + iy) * IWR + kx) * ES + h * 4);
                            const f32x8 v8 = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7);
                            const f32x8 w8 = __builtin_shufflevector(iy < KS ? w[iy < KS ? iy : 0] : zero4,
                                                                     iy >= 1 ? w[iy >= 1 ? iy - 1 : 0] : zero4, 0, 1, 2, 3, 4, 5, 6, 7);
                            d8[h] += v8 * w8;
                        }
                    }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    f32x4 q0 = rr == 0 ? __builtin_shufflevector(d8[0], d8[0], 0, 1, 2, 3) : __builtin_shufflevector(d8[0], d8[0], 4, 5, 6, 7);
                    f32x4 q1 = rr == 0 ? __builtin_shufflevector(d8[1], d8[1], 0, 1, 2, 3) : __builtin_shufflevector(d8[1], d8[1], 4, 5, 6, 7);
                    if (a.relu_dw) {
                        q0.x = fmaxf(q0.x, 0.f); q0.y = fmaxf(q0.y, 0.f); q0.z = fmaxf(q0.z, 0.f); q0.w = fmaxf(q0.w, 0.f);
                        q1.x = fmaxf(q1.x, 0.f); q1.y = fmaxf(q1.y, 0.f); q1.z = fmaxf(q1.z, 0.f); q1.w = fmaxf(q1.w, 0.f);
                    }
                    MX::split(q0, q1, dhi[2 * pr + rr], dlo[2 * pr + rr]);
                }
            }
        } else {
            f32x4 d[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) d[h] = *reinterpret_cast<const f32x4*>(wd + KS * KS * 32 + h * 4);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        d[h] += *reinterpret_cast<const f32x4*>(Ebase + (ky * IWR + kx) * ES + h * 4) *
                                *reinterpret_cast<const f32x4*>(wd + (ky * KS + kx) * 32 + h * 4);
            if (a.relu_dw) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    d[h].x = fmaxf(d[h].x, 0.f); d[h].y = fmaxf(d[h].y, 0.f); d[h].z = fmaxf(d[h].z, 0.f); d[h].w = fmaxf(d[h].w, 0.f);
                }
            }
            MX::split(d[0], d[1], dhi[0], dlo[0]);
        }
        // ---- phase C
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) {
            const V8 wp = *reinterpret_cast<const V8*>(wb + nt * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < MTC; ++r) {
                accp[r][nt] = MX::mma(wp, dhi[r], dlo[r], accp[r][nt]);
            }
        }
        if (c + 1 < NCHUNK) store_w(c + 1);
        if (!EXPAND && c + 1 < NCHUNK) __syncthreads();     // (after the last chunk nothing writes LDS any more)
    }

    // stores as [scalar row base + 32-bit lane offset], ReLU as a max against 0 / -big (see ir_tile_v2_kernel's epilogue)
    const int seg_u = __builtin_amdgcn_readfirstlane(seg), r0_u = __builtin_amdgcn_readfirstlane(r0);
    const long m0 = (crop * Ho + oy0 + r0_u) * Wo + ox0 + seg_u * 16;
    const unsigned ylane = (unsigned)(li * a.ldy + lk * 4), rlane = (unsigned)(li * a.ldr + lk * 4);
    const float relu_lo = a.relu_out ? 0.f : -3.0e38f;
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        if (nt * 16 >= COUT) continue;
        const bool n_ok = COUT % 16 == 0 || nt * 16 + lk * 4 < COUT;
        f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (n_ok) b = *reinterpret_cast<const f32x4*>(a.bp + nt * 16 + lk * 4);
#pragma unroll
        for (int r = 0; r < MTC; ++r) {
            const long mrow = m0 + (long)r * Wo;
            f32x4 v = accp[r][nt] + b;
            if (a.R && n_ok) v += *reinterpret_cast<const f32x4*>(a.R + (mrow * a.ldr + nt * 16 + (long)rlane));
            v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);
            if (n_ok) st_act4<(IO & IO_Y_BF16) != 0>(a.Y, mrow * a.ldy + nt * 16 + (long)ylane, v);
        }
    }
}

// ================================================================================================
// chain16: the whole stride-16 trunk stage (a run of inverted-residual blocks on the 16x16 map + the 1x1 neck) as
// ONE kernel.  It rests on the fragment identity of the 16x16x4 MFMA used throughout: the accumulator fragment
// of a block's projection (lane = pixel l&15, channels 4*(l>>4)..+3 of each 16-channel tile) IS the B-operand
// fragment of the next block's expansion, so the activations of a crop stay in registers from block to block
// (the residual add is a register add) and only the first block reads, and only the neck writes, global memory.
// Each block runs the ir16v2 chunk pipeline (E tile double buffered in LDS, packed weights staged through LDS);
// what disappears are 7 kernel prologues/epilogues, 7 activation round trips and the separate neck launch.
template <int CIN_, int CEXP_, int COUT_, int KS_, bool RES_>
struct ChainBlk {
    static constexpr int CIN = CIN_, CEXP = CEXP_, COUT = COUT_, KS = KS_;
    static constexpr bool RES = RES_;
};

struct Chain16Args {
    const float* X;        // [B*256][ldx]  input of the first block
    float* Y;              // [B*256][ldy]  neck output
    int ldx, ldy;
    const float* Wpk[8];   // per block: ir16v2 packed weights (Ir2Geom layout)
    const float* bp[8];    // per block: projection bias [COUT]
    const float* neck_pk;  // neck weights as fragments [NTN][KGN][256]
    const float* neck_b;   // [COUT_NECK]
};

template <int KS, int AP_MAX, int BP_MAX>
struct Chain16Lds {
    static constexpr int P = KS / 2, PW = 16 + 2 * P, ES = 24;
    static constexpr int EBUF = PW * PW * ES;
    static constexpr int FLOATS = 2 * EBUF + 2 * AP_MAX + 2 * BP_MAX;
};

// one block of the chain: xin (registers) -> yout (registers)
// NB / WpkNext: the block that follows (void: none) — its first weight stages A(0), B(0), A(1) are copied during THIS block's last
// interval (an even chunk count leaves exactly those three stages unread there), and it is then instantiated with PRE = true: no
// staging and no barrier of its own in front of its first expansion.  (At a block boundary the copies' latency was exposed: issue,
// barrier, nothing in between — ~1 us per block, 10 boundaries in the chained launch.)
template <class B, int KS_LDS, int AP_MAX, int BP_MAX, class NB = void, bool PRE = false>
__device__ __forceinline__ void chain16_block(const f32x4 (&xin)[2][B::CIN / 16], f32x4 (&yout)[2][B::COUT / 16],
                                              const float* __restrict__ Wpk, const float* __restrict__ bp, float* lds,
                                              const float* __restrict__ WpkNext = nullptr) {
    using G = Ir2Geom<B::CIN, B::CEXP, B::COUT, B::KS, true>;
    using L = Chain16Lds<KS_LDS, AP_MAX, BP_MAX>;
    static_assert(B::KS == KS_LDS, "all blocks of a chain share the depthwise kernel size (LDS tile geometry)");
    constexpr int S = 16, P = G::P, PW = G::PW, ES = G::ES, NCHUNK = G::NCHUNK, NTP = G::NTP, KG = G::KG;
    constexpr int AP = G::AP, BP = G::BP, EBUF = G::EBUF, CST = AP + BP;
    constexpr int AP4 = AP / 4, BP4 = BP / 4, NRA = (AP4 + 511) / 512, NRB = (BP4 + 511) / 512;
    static_assert(AP <= AP_MAX && BP <= BP_MAX && EBUF == L::EBUF, "LDS carve");
    float* const Ebuf = lds;
    float* const WA = lds + 2 * L::EBUF;
    float* const WB = WA + 2 * AP_MAX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int y0 = wave * 2;

    // the packed weights go global -> LDS by asynchronous copies (no staging registers, no ds_write: three ds_write_b128 per thread
    // and interval were ~120 cycles of the SIMD's ALU time, profiles/r03_issue_probe.txt), issued at the start of an interval into
    // the stage the previous interval read last and complete at the interval's barrier
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto stage_a = [&](int c) { lds_copy_async<AP>(Wpk + (long)c * CST, WA + (c & 1) * AP_MAX, wave_s, lane); };
    auto stage_b = [&](int c) { lds_copy_async<BP>(Wpk + (long)c * CST + AP, WB + (c & 1) * BP_MAX, wave_s, lane); };
    auto phase_a = [&](int c) {
        const float* wa = WA + (c & 1) * AP_MAX;
        float* E = Ebuf + (c & 1) * EBUF;
        f32x4 acc[2];
        acc[0] = acc[1] = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const f32x4 wf = *reinterpret_cast<const f32x4*>(wa + kg * 256 + lane * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], xin[mt][kg][i], acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 v = acc[mt];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4*>(E + ((y0 + mt + P) * PW + li + P) * ES + lk * 4) = v;
        }
    };
    // the output fragments double as the projection accumulators, initialised with bias (+ residual)
    f32x4 (&accp)[2][NTP] = yout;
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + nt * 16 + lk * 4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            accp[mt][nt] = b;
            if (B::RES) accp[mt][nt] += xin[mt][nt];      // CIN == COUT: the block input fragment is the residual
        }
    }
    // prologue (the previous block / the kernel prologue ended with a barrier: stages and E are free)
    __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from moving code across block boundaries
    if constexpr (!PRE) {
        stage_a(0);
        stage_b(0);
        if (NCHUNK > 1) stage_a(1);
        __syncthreads();
    }
    phase_a(0);
    __syncthreads();
#if CHAIN16_PEEL
    // (the last chunk — nothing left to expand — is peeled instead of branching around two instantiations of the interval inside
    //  the loop: chain32's A/B of the same change, profiles/r06_chain32_kbench.txt)
    for (int c = 0; c < NCHUNK - 1; ++c) {
        if (c + 2 < NCHUNK) stage_a(c + 2);
        stage_b(c + 1);
        const float* Ec = Ebuf + (c & 1) * EBUF;
        float* En = Ebuf + ((c + 1) & 1) * EBUF;
        const float* wa = WA + ((c + 1) & 1) * AP_MAX;
        const float* wb = WB + (c & 1) * BP_MAX;
        ir16_interval<B::KS, PW, ES, KG, NTP, true>(Ec, En, wa, wb, xin, accp, y0, li, lk, lane, true);
        __syncthreads();
    }
    {
        constexpr int c = NCHUNK - 1;
        if constexpr (!std::is_void<NB>::value) {
            using GN = Ir2Geom<NB::CIN, NB::CEXP, NB::COUT, NB::KS, true>;
            static_assert(NCHUNK % 2 == 0 && GN::AP <= AP_MAX && GN::BP <= BP_MAX, "the last interval reads stage 1 of B only");
            lds_copy_async<GN::AP>(WpkNext, WA, wave_s, lane);
            lds_copy_async<GN::BP>(WpkNext + GN::AP, WB, wave_s, lane);
            if (GN::NCHUNK > 1) lds_copy_async<GN::AP>(WpkNext + (GN::AP + GN::BP), WA + AP_MAX, wave_s, lane);
        }
        ir16_interval<B::KS, PW, ES, KG, NTP, false>(Ebuf + (c & 1) * EBUF, Ebuf + ((c + 1) & 1) * EBUF, WA + ((c + 1) & 1) * AP_MAX,
                                                     WB + (c & 1) * BP_MAX, xin, accp, y0, li, lk, lane, true);
        __syncthreads();
    }
#else
    for (int c = 0; c < NCHUNK; ++c) {
        if (c + 2 < NCHUNK) stage_a(c + 2);
        if (c + 1 < NCHUNK) stage_b(c + 1);
        const float* Ec = Ebuf + (c & 1) * EBUF;
        float* En = Ebuf + ((c + 1) & 1) * EBUF;
        const float* wa = WA + ((c + 1) & 1) * AP_MAX;
        const float* wb = WB + (c & 1) * BP_MAX;
        if (c + 1 < NCHUNK) ir16_interval<B::KS, PW, ES, KG, NTP, true>(Ec, En, wa, wb, xin, accp, y0, li, lk, lane, true);
        else ir16_interval<B::KS, PW, ES, KG, NTP, false>(Ec, En, wa, wb, xin, accp, y0, li, lk, lane, true);
        __syncthreads();
    }
#endif
}

// FEAR-XS stride-16 stage: 7 blocks + neck.  (The engine matches the model's block table against this chain.)
// chain16_body: from the first block's input fragments (x0: wave w holds map rows 2w, 2w + 1) to the neck's output in global
// memory; the caller has zero-filled the two E tiles (Chain16Lds: the first 2 * EBUF floats of `lds`) — no barrier needed in
// between, the first block's prologue has one.  Shared by chain16_kernel and the fused chain32_16_kernel (fear_chain32.h).
template <class B0, class B1, class B2, class B3, class B4, class B5, class B6, int CNECK>
__device__ __forceinline__ void chain16_body(const f32x4 (&x0)[2][B0::CIN / 16], const Chain16Args& a, float* lds, long crop) {
    constexpr int KS = B0::KS;
    constexpr int APM = Ir2Geom<B4::CIN, B4::CEXP, B4::COUT, KS, true>::AP;      // widest expand fragments (CIN = 112)
    constexpr int BPM = Ir2Geom<B4::CIN, B4::CEXP, B4::COUT, KS, true>::BP;
    using L = Chain16Lds<KS, APM, BPM>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int y0 = wave * 2;

    f32x4 x1[2][B0::COUT / 16];
    chain16_block<B0, KS, APM, BPM, B1, false>(x0, x1, a.Wpk[0], a.bp[0], lds, a.Wpk[1]);
    f32x4 x2[2][B1::COUT / 16];
    chain16_block<B1, KS, APM, BPM, B2, true>(x1, x2, a.Wpk[1], a.bp[1], lds, a.Wpk[2]);
    f32x4 x3[2][B2::COUT / 16];
    chain16_block<B2, KS, APM, BPM, B3, true>(x2, x3, a.Wpk[2], a.bp[2], lds, a.Wpk[3]);
    f32x4 x4[2][B3::COUT / 16];
    chain16_block<B3, KS, APM, BPM, B4, true>(x3, x4, a.Wpk[3], a.bp[3], lds, a.Wpk[4]);
    f32x4 x5[2][B4::COUT / 16];
    chain16_block<B4, KS, APM, BPM, B5, true>(x4, x5, a.Wpk[4], a.bp[4], lds, a.Wpk[5]);
    f32x4 x6[2][B5::COUT / 16];
    chain16_block<B5, KS, APM, BPM, B6, true>(x5, x6, a.Wpk[5], a.bp[5], lds, a.Wpk[6]);
    f32x4 x7[2][B6::COUT / 16];
    chain16_block<B6, KS, APM, BPM, void, true>(x6, x7, a.Wpk[6], a.bp[6], lds);

    // ---- neck: Y = Wn . x7 + bn, 16 output tiles in groups of 4; fragments [nt][kg][256] staged through the (now
    //      free) E area in two 4-tile slots
    constexpr int KGN = B6::COUT / 16, NTN = CNECK / 16, GRP = 4, GFL = GRP * KGN * 256;   // floats per group
    constexpr int G4 = GFL / 4, NRG = (G4 + 511) / 512;
    static_assert(2 * GFL <= 2 * L::EBUF && NTN % GRP == 0, "neck staging");
    f32x4 rg[NRG];
    auto load_g = [&](int g) {
#pragma unroll
        for (int r = 0; r < NRG; ++r) {
            const int idx = tid + r * 512;
            if (idx < G4) rg[r] = *reinterpret_cast<const f32x4*>(a.neck_pk + (long)g * GFL + idx * 4);
        }
    };
    auto store_g = [&](int g) {
        float* dst = lds + (g & 1) * GFL;
#pragma unroll
        for (int r = 0; r < NRG; ++r) {
            const int idx = tid + r * 512;
            if (idx < G4) *reinterpret_cast<f32x4*>(dst + idx * 4) = rg[r];
        }
    };
    load_g(0);
    store_g(0);
    __syncthreads();
    for (int g = 0; g < NTN / GRP; ++g) {
        if (g + 1 < NTN / GRP) load_g(g + 1);
        const float* wg = lds + (g & 1) * GFL;
        f32x4 acc[2][GRP];
#pragma unroll
        for (int q = 0; q < GRP; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.neck_b + (g * GRP + q) * 16 + lk * 4);
            acc[0][q] = b;
            acc[1][q] = b;
        }
#pragma unroll
        for (int q = 0; q < GRP; ++q)
#pragma unroll
            for (int kg = 0; kg < KGN; ++kg) {
                const f32x4 wf = *reinterpret_cast<const f32x4*>(wg + (q * KGN + kg) * 256 + lane * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], x7[0][kg][i], acc[0][q], 0, 0, 0);
                    acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], x7[1][kg][i], acc[1][q], 0, 0, 0);
                }
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const long m = crop * 256 + (y0 + mt) * 16 + li;
                *reinterpret_cast<f32x4*>(a.Y + m * a.ldy + (g * GRP + q) * 16 + lk * 4) = acc[mt][q];
            }
        if (g + 1 < NTN / GRP) store_g(g + 1);
        __syncthreads();
    }
}


template <class B0, class B1, class B2, class B3, class B4, class B5, class B6, int CNECK>
__global__ __launch_bounds__(512) void chain16_kernel(Chain16Args a) {
    constexpr int KS = B0::KS;
    constexpr int APM = Ir2Geom<B4::CIN, B4::CEXP, B4::COUT, KS, true>::AP;
    constexpr int BPM = Ir2Geom<B4::CIN, B4::CEXP, B4::COUT, KS, true>::BP;
    using L = Chain16Lds<KS, APM, BPM>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long crop = blockIdx.x;
    const int y0 = wave * 2;
    for (int i = tid * 4; i < 2 * L::EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 x0[2][B0::CIN / 16];
    const float* Xc = a.X + crop * 256 * a.ldx;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kg = 0; kg < B0::CIN / 16; ++kg)
            x0[mt][kg] = *reinterpret_cast<const f32x4*>(Xc + (long)((y0 + mt) * 16 + li) * a.ldx + kg * 16 + lk * 4);
    __syncthreads();
    chain16_body<B0, B1, B2, B3, B4, B5, B6, CNECK>(x0, a, lds, crop);
}

}  // namespace fear
