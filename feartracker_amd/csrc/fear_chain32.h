// fear_chain32.h — the 32 x 32 trunk stage of FEAR-XS (four inverted-residual blocks, 32 channels between them, the last one
// stride 2 down to the 16 x 16 map) as ONE register-resident kernel, in chain16_kernel's style.  Reference: model/blocks.py:8-42
// builds the trunk from mobile_cv's fbnet_c stage table (stage 3: t = 3 / 6 / 6 blocks with c = 32, then the first block of stage 4
// with stride 2); the layer-wise plan runs each as pw + dw + pw (+ residual), the tile plan as four ir_tile_v2 launches.
//
// Why: the four tile kernels sit at 0.51-0.61 of the fp32 peak (profiles/r06_per_op_math0.csv) against chain16's 0.75, and what
// separates them is structure — halo recomputation (14-20 % of a 5 x 5 block on 16 x 16 / 16 x 32 tiles of a 32 x 32 map), two
// barriers per 16-channel chunk at 1-3 workgroups per CU, a prologue / epilogue per tile and block.  A 32 x 32 x 32 map is 128 KB:
// 64 registers per lane of an 8-wave workgroup.  So one workgroup per crop keeps the map in registers from block to block
// (fragment identity of the 16x16x4 MFMA, see chain16), no halo is ever recomputed, and only the first block reads and the
// last block writes global memory.
//
// Work split: wave w owns map rows 4w .. 4w+3, i.e. 8 m-tiles of 16 pixels (column half ch, row r: m = 4 ch + r); lane (li, lk)
// holds channels 4 lk .. 4 lk + 3 of each 16-channel tile of pixel li, as everywhere.
//
// LDS: ONE tile of the expanded 16-channel chunk, 36 x 36 pixel slots (2-pixel zero ring: the 5 x 5 blocks' padding; the 3 x 3
// block reads it one pixel in), kept as four channel-quad planes [lk][slot][4 floats] of 20 736 B (a multiple of 256 B: the 16
// lanes of a ds_read_b128 service group — 8 of an even lk, 8 of the next odd lk with the complementary pixel set,
// MI355X_MICROARCH.md §LDS — hit 16 distinct 16-byte slots): 82 944 B.  Two such tiles do not fit beside the weight stages, so
// the chunk pipeline is ir_tile_v4's: the expansion of chunk c + 1 runs on the matrix pipe BETWEEN the depthwise steps of
// chunk c (ir16_interval's interleaving) into 32 parked registers, and is stored after the barrier that ends the reads of chunk
// c: two barriers per chunk around eight ds_write_b128 per lane (measured: the barriers cost nothing, profiles/r06_chain32_kbench.txt).
// The depthwise of a column half is ONE chain over the wave's four rows (c32_half4; C32_ROWS4 = 0 keeps the two row-pair chains of
// the first form, c32_sub, for A/Bs); the last chunk of a block is peeled out of the chunk loop; the next block's first weight
// stages are copied during the current block's last interval.  What each of these is worth: the same file.
//
// chain32_kernel is the stage as a launch of its own, chain32_16_kernel the stage + chain16's seven blocks + the neck in one launch
// (the engine's default): the stride-2 block's output fragments are chain16's input fragments.
//
// The stride-2 block reads every other column: its tile is stored column-de-interleaved (even padded columns in slots 0..17,
// odd ones in 18..35 of a row), so that the 16 output pixels of an m-tile read 16 consecutive slots for every tap.
#pragma once
#include <type_traits>

#ifndef C32_D
#define C32_D 2        // LDS read-ahead of the stride-1 depthwise chain, in tap steps (one activation + one tap read per step)
#endif
#ifndef C32_GS
#define C32_GS 2       // tap steps per scheduling group of the stride-1 chain ([reads][MFMAs][packed FMAs] per group)
#endif
#ifndef C32_D2
#define C32_D2 3       // read-ahead of the stride-2 chain (35 steps: 5 columns x 7 input rows)
#endif
#ifndef C32_GS2
#define C32_GS2 1
#endif
#ifndef C32_ABL
#define C32_ABL 0      // tools/kbench ablations (bit mask): 1 no depthwise FMAs, 2 no expansion MFMAs, 4 no projection MFMAs, 8 no LDS read-ahead reloads, 16 no chunk barriers, 32 no tile stores, 64 no weight staging
#endif

namespace fear {

template <int CIN_, int CEXP_, int COUT_, int KS_, int STRIDE_, bool RES_>
struct C32Blk {
    static constexpr int CIN = CIN_, CEXP = CEXP_, COUT = COUT_, KS = KS_, STRIDE = STRIDE_;
    static constexpr bool RES = RES_;
    static constexpr int MO = STRIDE_ == 1 ? 8 : 2;      // output m-tiles per wave
};

struct Chain32Args {
    const float* X;        // [B*1024][ldx]  input of the first block (32 x 32 map, NHWC)
    float* Y;              // [B*256][ldy]   output of the last block (16 x 16 map)
    int ldx, ldy;
    const float* Wpk[4];   // per block: packed weights in Ir2Geom's chunk layout (pack_fused16)
    const float* bp[4];    // per block: projection bias [COUT]
};

struct C32Geom {
    static constexpr int S = 32, PT = 2, PW = S + 2 * PT, HALF = PW / 2, NPIX = PW * PW;
    static constexpr int PLANE = NPIX * 4;             // floats per channel-quad plane
    static constexpr int EBUF = 4 * PLANE;
    static constexpr int AP_MAX = 2 * 256 + 16;        // CIN = 32
    static constexpr int BP_MAX = 4 * 256 + 25 * 16 + 16;      // COUT = 64, 5 x 5
    static constexpr int FLOATS = EBUF + 2 * AP_MAX + 2 * BP_MAX;
    static constexpr int LDS_BYTES = FLOATS * 4;
    static_assert(PLANE * 4 % 256 == 0, "planes a multiple of 256 B apart (conflict-free ds_read_b128 groups)");
};

__device__ __forceinline__ f32x4 c32_relu(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}

// The first D tap steps' reads of a row-pair chain (activation + tap weight per step) and the depthwise bias.
template <int KS, int D>
__device__ __forceinline__ void c32_prime(const float* __restrict__ e0, const float* __restrict__ wd, f32x4 (&ev)[D], f32x4 (&wv)[D], f32x4& dbias) {
    constexpr int PW = C32Geom::PW;
    dbias = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const int kx = t / (KS + 1), iy = t % (KS + 1);
        ev[t] = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * 4);
        if (iy < KS) wv[t] = *reinterpret_cast<const f32x4*>(wd + (iy * KS + kx) * 16);
    }
}

// One row pair x column half of a stride-1 chunk interval: the depthwise of output rows y0, y0 + 1 (16 columns) as ir16_interval's
// chain of KS (KS + 1) tap steps (column kx outer, input row iy inner: the tap weight of (iy, kx) feeds row 0 now and row 1 in the
// next step), its projection into p0 / p1, and — between the steps — the expansion of two m-tiles of the NEXT chunk into a0 / a1
// (bias-initialised here; ReLU at the store).  e0 = this lane's pixel of the tile at tap (0, 0) of row y0.  The first D steps'
// reads arrive in ev / wv / dbias (c32_prime: issued by the caller, or by the previous quarter under its projection MFMAs); enext
// = the next quarter's e0 (nullptr: none).
template <int KS, int KG, int NTP, bool HAS_A>
__device__ __forceinline__ void c32_sub(const float* __restrict__ e0, const float* __restrict__ wa, const float* __restrict__ wb,
                                        const f32x4 (&x0)[KG], const f32x4 (&x1)[KG], f32x4& a0, f32x4& a1, f32x4 (&p0)[NTP],
                                        f32x4 (&p1)[NTP], int lk, int lane, f32x4 (&ev)[C32_D], f32x4 (&wv)[C32_D], f32x4& dbias, const float* enext) {
    constexpr int PW = C32Geom::PW, NS = KS * (KS + 1), GS = C32_GS, D = C32_D;
    constexpr int NU = HAS_A ? KG * 4 : 0;
    const float* wd = wb + NTP * 256 + lk * 4;
    f32x4 wfq[2];
    if (HAS_A) {
        a0 = a1 = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);      // expansion bias
        wfq[0] = *reinterpret_cast<const f32x4*>(wa + lane * 4);
    }
    f32x4 d0 = dbias;
    f32x4 d1 = d0;
    f32x4 wprev = (f32x4){0.f, 0.f, 0.f, 0.f}, wpq[2];
#pragma unroll
    for (int g = 0; g < NS; g += GS) {
        f32x4 e[GS], w[GS];
        if (g == NS - GS) wpq[0] = *reinterpret_cast<const f32x4*>(wb + lane * 4);      // the projection's first A fragment, a group ahead
#pragma unroll
        for (int s0 = 0; s0 < GS; ++s0) {
            const int t = g + s0;
            e[s0] = ev[t % D];
            w[s0] = wv[t % D];
            if (t + D < NS && !(C32_ABL & 8)) {
                const int kx2 = (t + D) / (KS + 1), iy2 = (t + D) % (KS + 1);
                ev[t % D] = *reinterpret_cast<const f32x4*>(e0 + (iy2 * PW + kx2) * 4);
                if (iy2 < KS) wv[t % D] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 16);
            }
        }
        if (HAS_A) {
#pragma unroll
            for (int u = g * NU / NS; u < (g + GS) * NU / NS; ++u) {
                const int kg = u / 4, i = u % 4;
                if (i == 0 && kg + 1 < KG) wfq[(kg + 1) & 1] = *reinterpret_cast<const f32x4*>(wa + (kg + 1) * 256 + lane * 4);
                if (C32_ABL & 2) continue;
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], x0[HAS_A ? kg : 0][i], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], x1[HAS_A ? kg : 0][i], a1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int s0 = 0; s0 < GS; ++s0) {
            const int iy = (g + s0) % (KS + 1);
            if (C32_ABL & 1) {
                asm volatile("" :: "v"(e[s0]), "v"(w[s0]));
            } else {
                if (iy < KS) pk_fma4(d0, e[s0], w[s0]);
                if (iy >= 1) pk_fma4(d1, e[s0], wprev);
            }
            wprev = w[s0];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    pk_fma_settle(d0, d1);
    d0 = c32_relu(d0);
    d1 = c32_relu(d1);
    if (enext) c32_prime<KS, D>(enext, wd, ev, wv, dbias);
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        if (nt + 1 < NTP) wpq[(nt + 1) & 1] = *reinterpret_cast<const f32x4*>(wb + (nt + 1) * 256 + lane * 4);
        if (C32_ABL & 4) { p0[nt] += d0 * wpq[nt & 1]; p1[nt] += d1 * wpq[nt & 1]; continue; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d0[i], p0[nt], 0, 0, 0);
            p1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d1[i], p1[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifndef C32_ROWS4
#define C32_ROWS4 1    // 1: the stride-1 depthwise as ONE chain over the wave's four rows per column half (c32_half4) instead of two row-pair chains
#endif
#ifndef C32_D4
#define C32_D4 4       // read-ahead of that chain, in tap steps
#endif
// The first D steps' reads of a four-row chain (KS + 3 input rows per column).
template <int KS, int D>
__device__ __forceinline__ void c32_prime4(const float* __restrict__ e0, const float* __restrict__ wd, f32x4 (&ev)[D], f32x4 (&wv)[D], f32x4& dbias) {
    constexpr int PW = C32Geom::PW, NR = KS + 3;
    dbias = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const int kx = t / NR, iy = t % NR;
        ev[t] = *reinterpret_cast<const f32x4*>(e0 + (iy * PW + kx) * 4);
        if (iy < KS) wv[t] = *reinterpret_cast<const f32x4*>(wd + (iy * KS + kx) * 16);
    }
}

// One column half of a stride-1 chunk interval as ONE depthwise chain over the wave's four rows: KS (KS + 3) tap steps (column kx
// outer, input row iy inner); the activation read of a step feeds up to four output rows (row r with tap row iy - r), the tap weight
// read at step iy stays in a four-deep window until row 3 has used it: 8 + 5 LDS reads per column instead of 2 x (6 + 5) —
// each ds_read_b128 costs ~5 issue cycles of a kernel that is issue-bound (profiles/r06_chain32_kbench.txt).
template <int KS, int KG, int NTP, bool HAS_A>
__device__ __forceinline__ void c32_half4(const float* __restrict__ e0, const float* __restrict__ wa, const float* __restrict__ wb,
                                          const f32x4 (&x0)[KG], const f32x4 (&x1)[KG], const f32x4 (&x2)[KG], const f32x4 (&x3)[KG],
                                          f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4 (&p0)[NTP], f32x4 (&p1)[NTP], f32x4 (&p2)[NTP],
                                          f32x4 (&p3)[NTP], int lk, int lane, f32x4 (&ev)[C32_D4], f32x4 (&wv)[C32_D4], f32x4& dbias,
                                          const float* enext) {
    constexpr int PW = C32Geom::PW, NR = KS + 3, NS = KS * NR, D = C32_D4;
    constexpr int NU = HAS_A ? KG * 4 : 0;
    static_assert(NS >= D + 1, "read-ahead");
    const float* wd = wb + NTP * 256 + lk * 4;
    f32x4 wfq[2];
    if (HAS_A) {
        a0 = a1 = a2 = a3 = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);      // expansion bias
        wfq[0] = *reinterpret_cast<const f32x4*>(wa + lane * 4);
    }
    f32x4 d0 = dbias, d1 = dbias, d2 = dbias, d3 = dbias;
    f32x4 wwin[4], wpq[2];
    // (static_for, not `#pragma unroll`: past ~30 steps of this size hipcc leaves the loop rolled — ring and window become
    //  scratch arrays, 575-2000 us)
    static_for<0, NS>([&](auto T) {
        constexpr int t = decltype(T)::value, iy = t % NR;
        const f32x4 e = ev[t % D];
        if constexpr (iy < KS) wwin[iy & 3] = wv[t % D];
        if constexpr (t == NS - 1) wpq[0] = *reinterpret_cast<const f32x4*>(wb + lane * 4);      // the projection's first A fragment, a step ahead
        if constexpr (t + D < NS && !(C32_ABL & 8)) {
            constexpr int kx2 = (t + D) / NR, iy2 = (t + D) % NR;
            ev[t % D] = *reinterpret_cast<const f32x4*>(e0 + (iy2 * PW + kx2) * 4);
            if constexpr (iy2 < KS) wv[t % D] = *reinterpret_cast<const f32x4*>(wd + (iy2 * KS + kx2) * 16);
        }
        if constexpr (HAS_A) {
            static_for<t * NU / NS, (t + 1) * NU / NS>([&](auto U) {
                constexpr int u = decltype(U)::value, kg = u / 4, i = u % 4;
                if constexpr (i == 0 && kg + 1 < KG) wfq[(kg + 1) & 1] = *reinterpret_cast<const f32x4*>(wa + (kg + 1) * 256 + lane * 4);
                if constexpr (!(C32_ABL & 2)) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], x0[kg][i], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], x1[kg][i], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], x2[kg][i], a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], x3[kg][i], a3, 0, 0, 0);
                }
            });
        }
        if constexpr (C32_ABL & 1) {
            asm volatile("" :: "v"(e));
        } else {
            if constexpr (iy < KS) pk_fma4(d0, e, wwin[iy & 3]);
            if constexpr (iy >= 1 && iy - 1 < KS) pk_fma4(d1, e, wwin[(iy - 1) & 3]);
            if constexpr (iy >= 2 && iy - 2 < KS) pk_fma4(d2, e, wwin[(iy - 2) & 3]);
            if constexpr (iy >= 3) pk_fma4(d3, e, wwin[(iy - 3) & 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));      // (pk_fma_settle: inline-asm FMA results feed MFMAs)
    d0 = c32_relu(d0);
    d1 = c32_relu(d1);
    d2 = c32_relu(d2);
    d3 = c32_relu(d3);
    if (enext) c32_prime4<KS, D>(enext, wd, ev, wv, dbias);
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        if (nt + 1 < NTP) wpq[(nt + 1) & 1] = *reinterpret_cast<const f32x4*>(wb + (nt + 1) * 256 + lane * 4);
        if (C32_ABL & 4) { p0[nt] += d0 * wpq[nt & 1]; p1[nt] += d1 * wpq[nt & 1]; p2[nt] += d2 * wpq[nt & 1]; p3[nt] += d3 * wpq[nt & 1]; continue; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d0[i], p0[nt], 0, 0, 0);
            p1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d1[i], p1[nt], 0, 0, 0);
            p2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d2[i], p2[nt], 0, 0, 0);
            p3[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d3[i], p3[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The chunk interval of the stride-2 block (5 x 5): output rows 2w, 2w + 1 of the 16 x 16 map are two m-tiles per wave.  Output
// row 0 reads input rows rr = 0..4 of the wave's seven (4w - 2 .. 4w + 4) with tap row rr, output row 1 rows rr = 2..6 with tap
// row rr - 2: 35 steps (column outer), the tap weight of a step feeds row 0 now and row 1 two steps later.  The tile is column-
// de-interleaved (see the header): tap column kx of output pixel li sits in slot (kx & 1) * 18 + li + (kx >> 1).  The expansion
// of all eight m-tiles of the next chunk is dealt out between the steps.
template <int KG, int NTP, bool HAS_A>
__device__ __forceinline__ void c32_interval_s2(const float* __restrict__ e0, const float* __restrict__ wa, const float* __restrict__ wb,
                                                const f32x4 (&xin)[8][KG], f32x4 (&park)[8], f32x4 (&accp)[2][NTP], int lk, int lane) {
    constexpr int KS = 5, NR = 7, PW = C32Geom::PW, HALF = C32Geom::HALF, NS = KS * NR, GS = C32_GS2, D = C32_D2;
    constexpr int NU = HAS_A ? 4 * KG * 4 : 0;       // units of two MFMAs (an m-tile pair)
    static_assert(NS % GS == 0 && D >= GS && D % GS == 0, "step grouping");
    const float* wd = wb + NTP * 256 + lk * 4;
    f32x4 wfq[2];
    if (HAS_A) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);
#pragma unroll
        for (int m = 0; m < 8; ++m) park[m] = b;
    }
    f32x4 d0 = *reinterpret_cast<const f32x4*>(wd + KS * KS * 16);
    f32x4 d1 = d0;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ev[D], wv[D], wp1 = zero4, wp2 = zero4;
    auto eoff = [&](int t) { const int kx = t / NR, rr = t % NR; return (rr * PW + (kx & 1) * HALF + (kx >> 1)) * 4; };
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const int kx = t / NR, rr = t % NR;
        ev[t] = *reinterpret_cast<const f32x4*>(e0 + eoff(t));
        if (rr < KS) wv[t] = *reinterpret_cast<const f32x4*>(wd + (rr * KS + kx) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NS; g += GS) {
        f32x4 e[GS], w[GS];
#pragma unroll
        for (int s0 = 0; s0 < GS; ++s0) {
            const int t = g + s0;
            e[s0] = ev[t % D];
            w[s0] = (t % NR) < KS ? wv[t % D] : zero4;
            if (t + D < NS && !(C32_ABL & 8)) {
                const int kx2 = (t + D) / NR, rr2 = (t + D) % NR;
                ev[t % D] = *reinterpret_cast<const f32x4*>(e0 + eoff(t + D));
                if (rr2 < KS) wv[t % D] = *reinterpret_cast<const f32x4*>(wd + (rr2 * KS + kx2) * 16);
            }
        }
        if (HAS_A) {
#pragma unroll
            for (int u = g * NU / NS; u < (g + GS) * NU / NS; ++u) {
                const int q = u / (KG * 4), kg = (u / 4) % KG, i = u % 4;
                if (i == 0) wfq[kg & 1] = *reinterpret_cast<const f32x4*>(wa + kg * 256 + lane * 4);
                if (C32_ABL & 2) continue;
                park[2 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], xin[2 * q][HAS_A ? kg : 0][i], park[2 * q], 0, 0, 0);
                park[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfq[kg & 1][i], xin[2 * q + 1][HAS_A ? kg : 0][i], park[2 * q + 1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s0 = 0; s0 < GS; ++s0) {
            const int rr = (g + s0) % NR;
            if (C32_ABL & 1) {
                asm volatile("" :: "v"(e[s0]), "v"(w[s0]));
            } else {
                if (rr < KS) pk_fma4(d0, e[s0], w[s0]);
                if (rr >= 2) pk_fma4(d1, e[s0], wp2);
            }
            wp2 = wp1;
            wp1 = w[s0];
            if (rr == NR - 1) { wp1 = zero4; wp2 = zero4; }     // (a new column: the taps restart; the values are not read before they are set)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    pk_fma_settle(d0, d1);
    f32x4 wpq[2];
    wpq[0] = *reinterpret_cast<const f32x4*>(wb + lane * 4);
    d0 = c32_relu(d0);
    d1 = c32_relu(d1);
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        if (nt + 1 < NTP) wpq[(nt + 1) & 1] = *reinterpret_cast<const f32x4*>(wb + (nt + 1) * 256 + lane * 4);
        if (C32_ABL & 4) { accp[0][nt] += d0 * wpq[nt & 1]; accp[1][nt] += d1 * wpq[nt & 1]; continue; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            accp[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d0[i], accp[0][nt], 0, 0, 0);
            accp[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpq[nt & 1][i], d1[i], accp[1][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one block of the chain: xin (registers, 8 m-tiles) -> yout (registers: 8 m-tiles, or the wave's 2 of the 16 x 16 map)
// (NB / WpkNext / PRE: the next block's first weight stages copied during this block's last interval — see chain16_block)
template <class B, class NB = void, bool PRE = false>
__device__ __forceinline__ void chain32_block(const f32x4 (&xin)[8][B::CIN / 16], f32x4 (&yout)[B::MO][B::COUT / 16],
                                              const float* __restrict__ Wpk, const float* __restrict__ bp, float* lds,
                                              const float* __restrict__ WpkNext = nullptr) {
    using G = Ir2Geom<B::CIN, B::CEXP, B::COUT, B::KS, true>;      // the packed chunk layout (AP | BP), nothing else
    using L = C32Geom;
    constexpr int KS = B::KS, P = KS / 2, PT = L::PT, PW = L::PW, NCHUNK = G::NCHUNK, NTP = G::NTP, KG = G::KG;
    constexpr int AP = G::AP, BP = G::BP, CST = AP + BP;
    constexpr bool S2 = B::STRIDE == 2;
    static_assert(AP <= L::AP_MAX && BP <= L::BP_MAX && (!S2 || KS == 5) && (!B::RES || (B::CIN == B::COUT && !S2)), "block shape");
    float* const E = lds;
    float* const WA = lds + L::EBUF;
    float* const WB = WA + 2 * L::AP_MAX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto stage_a = [&](int c) { lds_copy_async<AP>(Wpk + (long)c * CST, WA + (c & 1) * L::AP_MAX, wave_s, lane); };
    auto stage_b = [&](int c) { lds_copy_async<BP>(Wpk + (long)c * CST + AP, WB + (c & 1) * L::BP_MAX, wave_s, lane); };

    f32x4 park[8];
    // the parked expansion of the next chunk -> the tile (ReLU here)
    float* const est = E + lk * L::PLANE;
    auto store_park = [&]() {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int y = wave * 4 + (m & 3) + PT, xs = (m >> 2) * 16 + li + PT;
            const int slot = S2 ? y * PW + (xs & 1) * L::HALF + (xs >> 1) : y * PW + xs;
            *reinterpret_cast<f32x4*>(est + slot * 4) = c32_relu(park[m]);
        }
    };
    f32x4 (&accp)[B::MO][NTP] = yout;
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp + nt * 16 + lk * 4);
#pragma unroll
        for (int m = 0; m < B::MO; ++m) {
            accp[m][nt] = b;
            if constexpr (B::RES) accp[m][nt] += xin[m][nt];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!PRE) {
        stage_a(0);
        stage_b(0);
        if (NCHUNK > 1) stage_a(1);
        __syncthreads();
    }
    {   // chunk 0's expansion: nothing to overlap it with
        const float* wa = WA;
        const f32x4 b = *reinterpret_cast<const f32x4*>(wa + KG * 256 + lk * 4);
#pragma unroll
        for (int m = 0; m < 8; ++m) park[m] = b;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const f32x4 wf = *reinterpret_cast<const f32x4*>(wa + kg * 256 + lane * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 8; ++m) park[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], xin[m][kg][i], park[m], 0, 0, 0);
        }
        store_park();
    }
    __syncthreads();
    // (the last chunk — nothing left to expand — is peeled: with `more` as a run-time branch around two instantiations of the
    //  interval hipcc's register allocator spills hundreds of registers at the join)
    auto interval = [&](int c, auto more_c) {
        constexpr bool MORE = decltype(more_c)::value;
        if (c + 2 < NCHUNK && !(C32_ABL & 64)) stage_a(c + 2);
        if (MORE && !(C32_ABL & 64)) stage_b(c + 1);
        const float* wa = WA + ((c + 1) & 1) * L::AP_MAX;
        const float* wb = WB + (c & 1) * L::BP_MAX;
        if constexpr (S2) {
            const float* e0 = E + lk * L::PLANE + (wave * 4 * PW + li) * 4;
            c32_interval_s2<KG, NTP, MORE>(e0, wa, wb, xin, park, accp, lk, lane);
        } else {
#if C32_ROWS4
            f32x4 ev[C32_D4], wv[C32_D4], dbias;
            const float* eb = E + lk * L::PLANE + ((wave * 4 + PT - P) * PW + li + PT - P) * 4;
            c32_prime4<KS, C32_D4>(eb, wb + NTP * 256 + lk * 4, ev, wv, dbias);
            __builtin_amdgcn_sched_barrier(0);
            c32_half4<KS, KG, NTP, MORE>(eb, wa, wb, xin[0], xin[1], xin[2], xin[3], park[0], park[1], park[2], park[3], accp[0], accp[1], accp[2],
                                         accp[3], lk, lane, ev, wv, dbias, eb + 64);
            c32_half4<KS, KG, NTP, MORE>(eb + 64, wa, wb, xin[4], xin[5], xin[6], xin[7], park[4], park[5], park[6], park[7], accp[4], accp[5],
                                         accp[6], accp[7], lk, lane, ev, wv, dbias, nullptr);
        }
#else
            f32x4 ev[C32_D], wv[C32_D], dbias;
            const float* eb = E + lk * L::PLANE + ((wave * 4 + PT - P) * PW + li + PT - P) * 4;
            c32_prime<KS, C32_D>(eb, wb + NTP * 256 + lk * 4, ev, wv, dbias);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {      // column half q >> 1, row pair q & 1
                const float* e0 = eb + ((q & 1) * 2 * PW + (q >> 1) * 16) * 4;
                const float* en = eb + (((q + 1) & 1) * 2 * PW + ((q + 1) >> 1) * 16) * 4;
                c32_sub<KS, KG, NTP, MORE>(e0, wa, wb, xin[2 * q], xin[2 * q + 1], park[2 * q], park[2 * q + 1], accp[2 * q], accp[2 * q + 1], lk, lane, ev, wv, dbias, q < 3 ? en : nullptr);
            }
        }
#endif
        if (!(C32_ABL & 16)) __syncthreads();
        if constexpr (MORE) {
            if (!(C32_ABL & 32)) store_park();
            if (!(C32_ABL & 16)) __syncthreads();
        }
    };
    for (int c = 0; c < NCHUNK - 1; ++c) interval(c, std::true_type{});
    if constexpr (!std::is_void<NB>::value) {
        using GN = Ir2Geom<NB::CIN, NB::CEXP, NB::COUT, NB::KS, true>;
        static_assert(NCHUNK % 2 == 0 && GN::AP <= L::AP_MAX && GN::BP <= L::BP_MAX, "the last interval reads stage 1 of B only");
        lds_copy_async<GN::AP>(WpkNext, WA, wave_s, lane);
        lds_copy_async<GN::BP>(WpkNext + GN::AP, WB, wave_s, lane);
        if (GN::NCHUNK > 1) lds_copy_async<GN::AP>(WpkNext + (GN::AP + GN::BP), WA + L::AP_MAX, wave_s, lane);
    }
    interval(NCHUNK - 1, std::false_type{});
}

// chain32_body: from the crop's 32 x 32 x 32 map in global memory to the output fragments of the stride-2 block (wave w: rows
// 2w, 2w + 1 of the 16 x 16 map — chain16's input fragments).
template <class B0, class B1, class B2, class B3>
__device__ __forceinline__ void chain32_body(const Chain32Args& a, float* lds, long crop, f32x4 (&y4)[2][B3::COUT / 16]) {
    using L = C32Geom;
    static_assert(B0::STRIDE == 1 && B1::STRIDE == 1 && B2::STRIDE == 1 && B3::STRIDE == 2, "three blocks on the 32 x 32 map, then the stride-2 block");
    static_assert(B0::COUT == B1::CIN && B1::COUT == B2::CIN && B2::COUT == B3::CIN, "chain");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    // the whole tile once: its zero ring is the padding of every chunk of every block (the interior is rewritten per chunk)
    for (int i = tid * 4; i < L::EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 x0[8][B0::CIN / 16];
    const float* Xc = a.X + crop * 1024 * a.ldx;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int y = wave * 4 + (m & 3), x = (m >> 2) * 16 + li;
#pragma unroll
        for (int kg = 0; kg < B0::CIN / 16; ++kg)
            x0[m][kg] = *reinterpret_cast<const f32x4*>(Xc + (long)(y * 32 + x) * a.ldx + kg * 16 + lk * 4);
    }
    __syncthreads();

    f32x4 x1[8][B0::COUT / 16];
    chain32_block<B0, B1, false>(x0, x1, a.Wpk[0], a.bp[0], lds, a.Wpk[1]);
    f32x4 x2[8][B1::COUT / 16];
    chain32_block<B1, B2, true>(x1, x2, a.Wpk[1], a.bp[1], lds, a.Wpk[2]);
    f32x4 x3[8][B2::COUT / 16];
    chain32_block<B2, B3, true>(x2, x3, a.Wpk[2], a.bp[2], lds, a.Wpk[3]);
    // the stride-2 block's tile is column-de-interleaved: slots 17 and 18 of a row (interior so far) become its padding
    // (the last barrier of B2 is behind us; B3 stores after its first barrier)
    if (tid < 4 * L::PW * 2) {
        const int pl = tid / (L::PW * 2), r = tid % (L::PW * 2);
        *reinterpret_cast<f32x4*>(lds + pl * L::PLANE + ((r >> 1) * L::PW + L::HALF - 1 + (r & 1)) * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    chain32_block<B3, void, true>(x3, y4, a.Wpk[3], a.bp[3], lds);
}

template <class B0, class B1, class B2, class B3>
__global__ __launch_bounds__(512) void chain32_kernel(Chain32Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long crop = blockIdx.x;
    f32x4 y4[2][B3::COUT / 16];
    chain32_body<B0, B1, B2, B3>(a, lds, crop, y4);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < B3::COUT / 16; ++nt)
            *reinterpret_cast<f32x4*>(a.Y + (crop * 256 + (wave * 2 + mt) * 16 + li) * (long)a.ldy + nt * 16 + lk * 4) = y4[mt][nt];
}

// chain32 + chain16 + neck as ONE launch: the output fragments of the stride-2 block ARE chain16's input fragments (wave w: rows
// 2w, 2w + 1 of the 16 x 16 map), so the 64-channel map between the two stages never leaves the registers either; what
// disappears is a launch boundary at which 256 CUs wait for the slowest workgroup of the first kernel, chain16's prologue (its
// first loads) and a 16.8 MB round trip.  LDS: the larger of the two carves; chain16's two E tiles are zero-filled behind the
// last barrier of the stride-2 block (their readers come after the barrier in chain16's first block prologue).
template <class A0, class A1, class A2, class A3, class B0, class B1, class B2, class B3, class B4, class B5, class B6, int CNECK>
__global__ __launch_bounds__(512) void chain32_16_kernel(Chain32Args a32, Chain16Args a16) {
    static_assert(A3::COUT == B0::CIN, "the stride-2 block feeds the stride-16 stage");
    constexpr int KS = B0::KS;
    using L16 = Chain16Lds<KS, Ir2Geom<B4::CIN, B4::CEXP, B4::COUT, KS, true>::AP, Ir2Geom<B4::CIN, B4::CEXP, B4::COUT, KS, true>::BP>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const long crop = blockIdx.x;
    f32x4 x0[2][B0::CIN / 16];
    chain32_body<A0, A1, A2, A3>(a32, lds, crop, x0);
    for (int i = tid * 4; i < 2 * L16::EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    chain16_body<B0, B1, B2, B3, B4, B5, B6, CNECK>(x0, a16, lds, crop);
}

}  // namespace fear
