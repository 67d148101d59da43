// fear_headchain.h — the BoxTower of FEAR-XS (model/blocks.py:129-194) as ONE launch: a workgroup owns one (crop, branch)
// and runs the branch's four SepConvs back to back — encode (+ pixel-wise correlation, blocks.py:121-123), the correlation
// SepConv over the 320-channel concat, and the two tower SepConvs (+ the prediction SepConv, blocks.py:167-168,186-192).
//
// What the measurements said (tools/headchain_check.hip): in sep16_kernel and in a first chained version
// the MFMA stretches run at the instruction rate, but every barrier interval (one per 16-channel input chunk: 16-24 per layer)
// loses 1 600-2 000 cycles at its start and end, global memory answers in ~3 us under load, and a layer's activation makes a
// round trip through HBM / MALL.  So this kernel is organised around ONE idea: a wave keeps the depthwise results of its
// two map rows for ALL input channels in registers (16 chunks x 2 rows x 4 = 128 VGPRs) — they are the B fragments of the
// pointwise GEMM — and the GEMM runs as EIGHT passes over 32 output channels each (accumulators: 16 VGPRs), with the pass's
// weight fragments resident in LDS (33-41 KB, the next pass's block copied in asynchronously meanwhile):
//   pass p    256 (320) MFMAs per wave, no barrier inside, A fragments from LDS, B fragments from registers
//   epilogue  bias / ReLU; the finished 32 channels (two 16-channel chunks of the NEXT layer's input) go to a two-chunk LDS
//             tile with halo -> barrier -> the next layer's depthwise of those two chunks runs from the tile (the waves
//             exchange their boundary rows through it) -> barrier.  A layer's activation never leaves the CU; the depthwise
//             results wait in a wave-private scratch (coalesced 1 KB stores / loads, no sharing) until the layer is over and
//             are pulled back into the registers chunk by chunk inside the last pass, behind the MFMAs that free them.
//   layer 0   + correlation: z^T y accumulates over the passes from 8 KB template slices (asynchronous copies)
//   layer 3   the tile feeds the prediction head's depthwise instead; its one-tile projection accumulates over the passes
// Two barriers per pass = 16 per layer, each after ~7 us of uninterrupted MFMA work.
// Same products in the same order as the eight sep16 launches it replaces: bit-identical maps (tests/test_gpu_parity.py).
#pragma once
#include <vector>
#ifndef HC_TS
#define HC_TS 2        // tap steps of the deferred hand-over per MFMA group (1, 2, 3, 4, 6)
#endif
#ifndef HC_ABL
#define HC_ABL 0       // timing ablations for tools/headchain_check only (bit mask); the product always builds with 0
                       // 1: no scratch stores of the next layer's depthwise results, 2: no scratch reloads (1 | 2: the parking is
                       // free), 4: no prologue fetch of the neck output, 8 / 128: time stamps
#endif

namespace fear {

struct HeadChainBranch {
    const float* W[4];       // per layer: pass-major packed weights (headchain_pack): 8 x [NC x 2 fragments | bias 32 | next layer's dw 4 x 160]
    const float* Wd0;        // layer 0's own depthwise weights, 16 chunks x [Wd[k*k][16] | bd[16]]
    const float* WdC;        // layer 1's depthwise weights of the four correlation chunks (16..19)
    const float* Z;          // template features [crop][256][64] (the caller's NCHW (256, 8, 8) tensor)
    long z_stride;
    const float* P_Wpk;      // prediction SepConv: 16 chunks x [1 fragment | Wd[k*k][16] | bd[16]]
    const float* P_bp;
    float* P_Y;              // the caller's NCHW map ([crop][pred_cout][256], pred_stride floats between crops)
    long pred_stride;
    int pred_cout, pred_act;
    float* D;                // [crop][8 waves][20 chunks][2 rows][64 lanes][4]: depthwise results waiting for their layer (wave private)
};

struct HeadChainArgs {
    const float* X;          // neck output [crop * 256][ldx]
    int ldx;
    int n_crops;             // launch with 2 * 8 * ceil(n_crops / 8) workgroups (see the id mapping in the kernel)
    int relu_dw, relu_out;
    HeadChainBranch br[2];   // blockIdx.y: 0 = classification, 1 = regression
    long long* dbg;          // HC_ABL & 8: wall-clock stamps of workgroup (100, 0) (tools/headchain_check)
};

// Workgroup barrier that does NOT wait for the wave's vector-memory operations (only for its LDS traffic): __syncthreads() waits
// for all of them, also for stores nobody else will ever read, whose write acknowledgements take microseconds under load.
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }

template <int KS>
struct HeadChainGeom {
    static constexpr int C = 256, TZ = 64, CC = C + TZ, S = 16, P = KS / 2, PW = S + 2 * P;
    static constexpr int EQ = (PW * PW * 4 + 63) / 64 * 64, EBUF = 4 * EQ;      // the four-plane tile of Sep16Geom, one chunk
    static constexpr int NPASS = 8, NTP = 2;                                    // 8 passes x 2 output tiles
    static constexpr int WDF = KS * KS * 16 + 16;                               // depthwise taps + bias of one chunk
    static constexpr int wpass(int cin) { return (cin / 16) * NTP * 256 + NTP * 16 + 2 * NTP * WDF; }
    static constexpr int WMAX = wpass(CC);
    static constexpr int PCH = 256 + WDF;                                       // prediction head, per chunk
    static constexpr int ZS = NTP * 16 * TZ;                                    // template slice of one pass
    static constexpr int ZREG = 16 * PCH > 2 * ZS ? 16 * PCH : 2 * ZS;          // two template slices | prediction weights
    static constexpr int LDS_FLOATS = 2 * WMAX + NTP * EBUF + ZREG;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int D_FLOATS = 8 * (CC / 16) * 2 * 256;      // depthwise results of one (crop, branch)
    static_assert(2 * ZS + C / 16 * WDF <= ZREG && 4 * WDF <= ZS, "prologue / correlation depthwise weights are staged behind / in the template slices");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// Host side: pass-major weights of one layer from the Sep16Geom-packed blocks (per 16-channel input chunk: 16 fragments |
// Wd[k*k][16] | bd[16]) of this layer and — for the depthwise that runs in this layer's hand-over — of the next one.
inline std::vector<float> headchain_pack(const float* sep_this, int cin, const float* bias256, const float* sep_next, int ks) {
    const int nc = cin / 16, wdf = ks * ks * 16 + 16, cst = 16 * 256 + wdf;
    std::vector<float> out;
    for (int p = 0; p < 8; ++p) {
        for (int c = 0; c < nc; ++c)
            for (int nt = 2 * p; nt < 2 * p + 2; ++nt) out.insert(out.end(), sep_this + (size_t)c * cst + nt * 256, sep_this + (size_t)c * cst + nt * 256 + 256);
        out.insert(out.end(), bias256 + 32 * p, bias256 + 32 * p + 32);
        for (int c = 2 * p - 2; c < 2 * p + 2; ++c) {      // this pass deals with the hand-over of the previous one (and, in the last pass, its own)
            if (sep_next && c >= 0) out.insert(out.end(), sep_next + (size_t)c * cst + 16 * 256, sep_next + (size_t)c * cst + 16 * 256 + wdf);
            else out.insert(out.end(), wdf, 0.f);
        }
    }
    return out;
}
// the depthwise parts (Wd | bd) of chunks [c0, c0 + n) of a Sep16Geom-packed layer
inline std::vector<float> headchain_pack_dw(const float* sep, int c0, int n, int ks) {
    const int wdf = ks * ks * 16 + 16, cst = 16 * 256 + wdf;
    std::vector<float> out;
    for (int c = c0; c < c0 + n; ++c) out.insert(out.end(), sep + (size_t)c * cst + 16 * 256, sep + (size_t)c * cst + 16 * 256 + wdf);
    return out;
}

template <int KS>
__global__ __launch_bounds__(512) void headchain_kernel(HeadChainArgs a) {
    using G = HeadChainGeom<KS>;
    using std::integral_constant;
    constexpr int C = G::C, TZ = G::TZ, CC = G::CC, S = G::S, P = G::P, PW = G::PW, EP = 4, EQ = G::EQ, EBUF = G::EBUF;
    constexpr int NPASS = G::NPASS, NTP = G::NTP, WDF = G::WDF, WMAX = G::WMAX, PCH = G::PCH, ZS = G::ZS;
    constexpr int NS = KS * (KS + 1), RA = 4;
    static_assert(NS >= RA, "read-ahead");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Wb = lds;                      // [2][WMAX]: the weight block of the pass in flight | of the next pass
    float* const Et = lds + 2 * WMAX;           // [NTP][EBUF]: hand-over tile, two chunks with halo
    float* const Zr = Et + NTP * EBUF;          // [2][ZS] template slices (layer 0) | correlation dw weights | prediction weights (layer 3)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    // workgroup id -> (crop, branch): ids 16k..16k+7 are the classification branch of crops 8k..8k+7, ids 16k+8..16k+15 their
    // regression branch — the dispatcher deals consecutive workgroups round-robin over the 8 XCDs, so the two workgroups that read
    // the same crop's neck output at the same time share an L2
    const unsigned wg = blockIdx.x;
    const int branch = (wg >> 3) & 1;
    const long crop = (long)(wg >> 4) * 8 + (wg & 7);
    const HeadChainBranch& b = a.br[branch];
    if (crop >= a.n_crops) return;
    const int y0 = wave * 2;
    // wave-private scratch: [wave][chunk][row][64 lanes][4] — 2 KB between chunks, so that a handful of base registers and the
    // instructions' 13-bit immediate offsets reach all of a wave's 20 chunks (with 16 KB between chunks hipcc kept an address
    // register pair per chunk alive across the GEMM and spilled them)
    // Addressing: uniform base + a 32-bit lane offset that is materialised AT the access (the empty asm pins it): with plain pointer
    // arithmetic hipcc precomputes one 64-bit address per chunk at the top of the layer and spills them.
    char* const Dws = reinterpret_cast<char*>(b.D + crop * (long)G::D_FLOATS + wave * (CC / 16 * 512));
    const unsigned lane16 = lane * 16;
    auto dptr = [&](int chunk, int row) {
        unsigned off = lane16 + (chunk * 2048 + row * 1024);
        asm volatile("" : "+v"(off));
        return reinterpret_cast<f32x4*>(Dws + off);
    };

    int stamp_i = 0;
    auto stamp = [&] {
        if ((HC_ABL & 8) && crop == 100 && branch == 0 && (wave == 0 || wave == 4) && lane == 0 && stamp_i < 40)
            a.dbg[(wave >> 2) * 40 + stamp_i++] = wall_clock64();
    };
    stamp();

    // the tile's halo is zero for the whole kernel (= the convolutions' padding): only the interior is ever rewritten
    for (int i = tid * 4; i < NTP * EBUF; i += 512 * 4) *reinterpret_cast<f32x4*>(Et + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the weight blocks alternate between Wb[0] and Wb[1] (the next pass's block is copied in during the current GEMM), the template
    // slices of layer 0 between Zr[0] and Zr[1]; layer 0's own depthwise taps wait behind them
    float* const Wd0s = Zr + 2 * ZS;
    lds_copy_async<G::wpass(C)>(b.W[0], Wb, wave, lane);
    lds_copy_async<C / 16 * WDF>(b.Wd0, Wd0s, wave, lane);
    lds_copy_async<ZS>(b.Z + crop * b.z_stride, Zr, wave, lane);

    f32x4 d[C / 16][2];                 // B fragments of the layer in flight: depthwise result [input chunk][row] of this wave's two rows

    // a finished 16-channel fragment pair (this wave's two rows) -> interior of tile slot s
    auto tile_put = [&](int s, const f32x4& v0, const f32x4& v1) {
        float* E = Et + s * EBUF;
        *reinterpret_cast<f32x4*>(E + ((y0 + P) * PW + li + P) * EP + lk * EQ) = v0;
        *reinterpret_cast<f32x4*>(E + ((y0 + 1 + P) * PW + li + P) * EP + lk * EQ) = v1;
    };
    // depthwise KS x KS of tile slot s (taps + bias at wd: [k*k][16] | [16]) for this wave's two rows; LDS reads RA tap steps ahead
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

    // ---------------- kernel prologue: the neck output -> depthwise of layer 0 -> d[][] (through the tile, two chunks a round).
    // The 32 loads of the wave's input rows are issued at once, BEHIND the asynchronous weight copies, and the rounds start as soon
    // as their own two chunks have arrived (vector-memory operations complete in issue order).  The prologue is fetch-bound — 256
    // workgroups pull 256 KB each at the same moment, ~15 us — not LDS- or barrier-bound: a variant without tile and barriers
    // (rows above / below from global memory, horizontal neighbours by DPP row shifts; bit-identical) took 16.6-21.5 us, its
    // 50 % extra halo fetch costing more than the eight rounds it removed.
    {
        const float* X0 = a.X + crop * 256 * a.ldx;
        __builtin_amdgcn_sched_barrier(0);     // (the copies above stay in front of the loads below: the wait further down counts on it)
#pragma unroll
        for (int c = 0; c < C / 16; ++c) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                d[c][mt] = (HC_ABL & 4) ? (f32x4){0.5f, 0.25f, 1.f, 2.f} : *reinterpret_cast<const f32x4*>(X0 + (long)((y0 + mt) * S + li) * a.ldx + c * 16 + lk * 4);
            __builtin_amdgcn_sched_barrier(0);     // (in chunk order: hipcc would issue the first chunks last)
        }
        // zero fill done and weight blocks landed: everything but the youngest 32 operations (the input rows).
        // (the builtin, not inline asm: hipcc's own wait insertion must SEE that the copies are complete — while it believes an
        // LDS-writing load is in flight it turns every later wait into vmcnt(0).  gfx9 encoding: vmcnt[3:0] | expcnt << 4 |
        // lgkmcnt << 8 | vmcnt[5:4] << 14)
        static_assert(2 * (C / 16) == 32, "waitcnt immediate below");
        __builtin_amdgcn_s_waitcnt((32 & 15) | (7 << 4) | (0 << 8) | ((32 >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        // (inline-asm LDS stores: in front of a plain one hipcc waits for EVERY vector-memory operation in flight — it cannot tell the
        // store from the asynchronous global -> LDS copies — i.e. for all 32 loads; the asm form waits for the registers it reads)
        auto tile_put_asm = [&](int s, const f32x4& v0, const f32x4& v1) {
            const unsigned a0 = (unsigned)(uintptr_t)(Et + s * EBUF + ((y0 + P) * PW + li + P) * EP + lk * EQ);
            asm volatile("ds_write_b128 %0, %1" : : "v"(a0), "v"(v0) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(a0), "v"(v1), "n"(PW * EP * 4) : "memory");
        };
#pragma unroll
        for (int r = 0; r < C / 32; ++r) {
            tile_put_asm(0, d[2 * r][0], d[2 * r][1]);
            tile_put_asm(1, d[2 * r + 1][0], d[2 * r + 1][1]);
            barrier_lds_only();
            tile_dw(0, Wd0s + (2 * r) * WDF, d[2 * r][0], d[2 * r][1], a.relu_dw);
            tile_dw(1, Wd0s + (2 * r + 1) * WDF, d[2 * r + 1][0], d[2 * r + 1][1], a.relu_dw);
            barrier_lds_only();
        }
    }
    stamp();

    f32x4 cacc[2][TZ / 16];             // layer 0: correlation accumulators
    f32x4 pacc[2];                      // layer 3: prediction accumulators
    f32x4 ds[TZ / 16][2];               // layer 1: the B fragments of the correlation chunks (16..19), written by layer 0's last hand-over
    f32x4 wf0, wf1;                     // the first two weight fragments of the next pass (read between its two barriers)
    wf0 = *reinterpret_cast<const f32x4*>(Wb + lane * 4);
    wf1 = *reinterpret_cast<const f32x4*>(Wb + 256 + lane * 4);

    // One SepConv layer = 8 passes.  MODE 0: plain, 1: + correlation (layer 0), 2: prediction head instead of a hand-over (layer 3).
    // On entry d[][] holds the layer's depthwise results (layer 1: chunks 16..19 in ds[][]), Wb[0] its pass-0 block.
    auto layer = [&](auto cin_tag, auto mode_tag, const float* Wl, const float* Wnext) {
        constexpr int CIN = decltype(cin_tag)::value, MODE = decltype(mode_tag)::value;
        constexpr int NC = CIN / 16, WP = G::wpass(CIN);
        constexpr int NTZ = TZ / 16;
        if (MODE == 1) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < NTZ; ++q) cacc[mt][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (MODE == 2) pacc[0] = pacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int fs_i = 0;
        auto fstamp = [&](int p) {          // HC_ABL & 128: s_memtime stamps inside passes 2..4 of the third layer
            if ((HC_ABL & 128) && MODE == 0 && CIN == C && p >= 2 && p < 5 && crop == 100 && branch == 0 && lane == 0 && fs_i < 21)
                a.dbg[80 + wave * 21 + fs_i++] = __builtin_amdgcn_s_memtime();
        };
        // The hand-over of pass q (its 32 finished channels = the NEXT layer's input chunks 2q, 2q + 1 are in the tile): the next layer's
        // depthwise of those chunks -> the scratch (DST < 0) or straight into d[DST], d[DST + 1]; layer 3: the prediction head instead.
        auto handoff_out = [&](int q, int nt, const f32x4& n0, const f32x4& n1, auto dst_tag) {
            constexpr int DST = decltype(dst_tag)::value;
            if (MODE != 2) {
                if (DST < 0) { if (!(HC_ABL & 1)) { *dptr(2 * q + nt, 0) = n0; *dptr(2 * q + nt, 1) = n1; } }
                else { d[DST < 0 ? 0 : DST][0] = n0; d[DST < 0 ? 0 : DST][1] = n1; }
            } else {
                // prediction SepConv's 1x1 to <= 4 channels on the depthwise of the finished chunk
                const f32x4 wq = *reinterpret_cast<const f32x4*>(Zr + (long)(2 * q + nt) * PCH + lane * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i], n0[i], pacc[0], 0, 0, 0);
                    pacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i], n1[i], pacc[1], 0, 0, 0);
                }
            }
        };
        // taps + bias of the depthwise that runs on pass q's chunk nt (layer 3: the prediction head's, resident in Zr)
        auto handoff_wd = [&](int q, int nt, const float* wdw) { return MODE == 2 ? Zr + (long)(2 * q + nt) * PCH + 256 : wdw + nt * WDF; };
        // One pass.  HAND: the hand-over of the PREVIOUS pass is dealt, one tap step per MFMA group, into this pass's GEMM — on its own
        // it is LDS-bound (44 tile / tap reads per wave and chunk against 36 packed FMAs) and nothing else can run beside it: a wave
        // that streams MFMAs keeps the SIMD's ALU to itself, whatever the other wave's priority (measured: s_setprio does not help),
        // so latencies hide only behind the issuing wave's OWN MFMAs.  Barrier B (every wave has read the tile) therefore sits at
        // the END of the pass, right in front of the tile's next writes, barrier A (tile complete, next weight block landed) behind them.
        auto pass = [&](int p, auto last_tag, auto hand_tag) {
            constexpr bool LAST = decltype(last_tag)::value, HAND = decltype(hand_tag)::value;
            constexpr int DWOFF = NC * NTP * 256 + NTP * 16;     // block: fragments | bias | dw taps of the next layer's chunks 2p-2 .. 2p+1
            constexpr int NG = NC * NTP;                         // MFMA groups (8 MFMAs each)
            // the deferred hand-over: TS tap steps per MFMA group it is dealt to (an MFMA <-> VALU switch costs ~8 cycles each way:
            // profiles/r03_issue_probe.txt), operands read one group ahead into rotating registers
            constexpr int TS = HC_TS, GPC = NS / TS + 2, NE = 2 * TS, NW = 2 * TS + 1;
            static_assert(NS % TS == 0 && NG >= 2 * GPC, "hand-over schedule");
            fstamp(p);
            const float* wb = Wb + (p & 1) * WMAX;
            f32x4 acc[2][NTP];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // ---- the pass's GEMM: no barrier inside; fragment reads run two fragments ahead
            f32x4 wf[3];
            wf[0] = wf0;
            wf[1] = wf1;
            // state of the deferred depthwise: tap step t of chunk h is dealt to group h * (NS + 2) + 1 + t; its LDS reads are issued
            // one group (8 MFMAs, ~260 cycles) ahead
            // (operands in rotating registers — he[t % NE], hw[t % NW] — so that a step neither copies nor overwrites what the
            // previous one still multiplies with)
            f32x4 hn0, hn1, he[NE], hw[NW];
            const float* hwd = nullptr;
            const float* he0 = nullptr;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    const int u = c * NTP + nt;
                    if (u + 2 < NG) wf[(u + 2) % 3] = *reinterpret_cast<const f32x4*>(wb + (u + 2) * 256 + lane * 4);
                    const f32x4& b0 = c < C / 16 ? d[c < C / 16 ? c : 0][0] : ds[c < C / 16 ? 0 : c - C / 16][0];
                    const f32x4& b1 = c < C / 16 ? d[c < C / 16 ? c : 0][1] : ds[c < C / 16 ? 0 : c - C / 16][1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u % 3][i], b0[i], acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u % 3][i], b1[i], acc[1][nt], 0, 0, 0);
                    }
                    if (HAND && u < 2 * GPC) {
                        const int h = u / GPC, g = u % GPC;                  // chunk of the pair, position in its schedule
                        if (g == 0) {                                        // begin: bias, the first TS tap steps' operands
                            hwd = handoff_wd(p - 1, h, wb + DWOFF) + lk * 4;
                            he0 = Et + h * EBUF + (y0 * PW + li) * EP + lk * EQ;
                            hn0 = *reinterpret_cast<const f32x4*>(hwd + KS * KS * 16);
#pragma unroll
                            for (int t = 0; t < TS; ++t) {
                                const int kx = t / (KS + 1), iy = t % (KS + 1);
                                he[t % NE] = *reinterpret_cast<const f32x4*>(he0 + (iy * PW + kx) * EP);
                                if (iy < KS) hw[t % NW] = *reinterpret_cast<const f32x4*>(hwd + (iy * KS + kx) * 16);
                            }
                        } else if (g <= NS / TS) {
#pragma unroll
                            for (int j = 0; j < TS; ++j) {                   // operands of the NEXT group's steps
                                const int t2 = g * TS + j;
                                if (t2 < NS) {
                                    const int kx2 = t2 / (KS + 1), iy2 = t2 % (KS + 1);
                                    he[t2 % NE] = *reinterpret_cast<const f32x4*>(he0 + (iy2 * PW + kx2) * EP);
                                    if (iy2 < KS) hw[t2 % NW] = *reinterpret_cast<const f32x4*>(hwd + (iy2 * KS + kx2) * 16);
                                }
                            }
#pragma unroll
                            for (int j = 0; j < TS; ++j) {
                                const int t = (g - 1) * TS + j, iy = t % (KS + 1);
                                if (t == 0) hn1 = hn0;
                                if (iy < KS) pk_fma4(hn0, he[t % NE], hw[t % NW]);
                                if (iy >= 1) pk_fma4(hn1, he[t % NE], hw[(t + NW - 1) % NW]);      // the previous step's tap
                            }
                        } else {                                             // end: the chunk's depthwise is complete
                            pk_fma_settle(hn0, hn1);
                            if (MODE != 2 && a.relu_dw) {
                                hn0.x = fmaxf(hn0.x, 0.f); hn0.y = fmaxf(hn0.y, 0.f); hn0.z = fmaxf(hn0.z, 0.f); hn0.w = fmaxf(hn0.w, 0.f);
                                hn1.x = fmaxf(hn1.x, 0.f); hn1.y = fmaxf(hn1.y, 0.f); hn1.z = fmaxf(hn1.z, 0.f); hn1.w = fmaxf(hn1.w, 0.f);
                            }
                            handoff_out(p - 1, h, hn0, hn1, integral_constant<int, -1>{});
                        }
                    }
                }
                if (c == 1) {
                    // asynchronous copies for the NEXT pass, issued from inside the GEMM (issue slots are free here; between the
                    // barriers they were ~500 cycles of the critical path): their buffers died at the previous pass's barrier B
                    if (!LAST) lds_copy_async<WP>(Wl + (long)(p + 1) * WP, Wb + ((p + 1) & 1) * WMAX, wave, lane);
                    else if (Wnext) lds_copy_async<G::wpass(MODE == 1 ? CC : C)>(Wnext, Wb + ((p + 1) & 1) * WMAX, wave, lane);
                    if (MODE == 1) {
                        if (!LAST) lds_copy_async<ZS>(b.Z + crop * b.z_stride + (long)(p + 1) * ZS, Zr + ((p + 1) & 1) * ZS, wave, lane);
                        else lds_copy_async<4 * WDF>(b.WdC, Zr + ((p + 1) & 1) * ZS, wave, lane);     // the correlation chunks' depthwise weights
                    }
                }
                if (LAST && MODE != 2 && c < C / 16 - NTP) {
                    // the layer is over for chunk c: pull the NEXT layer's depthwise result of chunk c into the freed registers
                    // (written to the scratch by this lane in the hand-over of pass c / 2 — the one of pass 6 a few groups ago; the
                    // last two chunks come straight from the hand-over below)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) if (!(HC_ABL & 2)) d[c][mt] = *dptr(c, mt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            fstamp(p);
            // ---- epilogue: bias / ReLU; the finished fragments are the next layer's input chunks 2p, 2p + 1
            f32x4 v[2][NTP];
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(wb + NC * NTP * 256 + nt * 16 + lk * 4);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 t = acc[mt][nt] + bv;
                    if (a.relu_out) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
                    v[mt][nt] = t;
                }
            }
            if (MODE == 1) {
                // pixel-wise correlation with the template (MobileCorrelation, blocks.py:121-123): the finished fragments are the B
                // operand as they stand; A fragment of (kg, q): lane l holds z[kg*16 + 4*(l>>4) + i][q*16 + (l&15)]
                const float* zs = Zr + (p & 1) * ZS;
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    const float* zr = zs + (nt * 16 + lk * 4) * TZ + li;
                    f32x4 zf[NTZ];
#pragma unroll
                    for (int q = 0; q < NTZ; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) zf[q][i] = zr[i * TZ + q * 16];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < NTZ; ++q) {
                            cacc[0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[q][i], v[0][nt][i], cacc[0][q], 0, 0, 0);
                            cacc[1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[q][i], v[1][nt][i], cacc[1][q], 0, 0, 0);
                        }
                }
            }
            fstamp(p);
            // B: every wave has read the tile (hand-over of pass p - 1) and this pass's weight block for the last time; the NEXT pass's
            // block (issued early in this pass's GEMM) has landed.  (The hand-over's scratch stores are wave private; they are old by now.)
            __syncthreads();
            fstamp(p);
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) tile_put(nt, v[0][nt], v[1][nt]);
            // between the barriers: the next pass's first fragments into registers — its GEMM starts with its MFMAs
            if (!LAST || Wnext) {
                const float* wn = Wb + ((p + 1) & 1) * WMAX;
                wf0 = *reinterpret_cast<const f32x4*>(wn + lane * 4);
                wf1 = *reinterpret_cast<const f32x4*>(wn + 256 + lane * 4);
            }
            barrier_lds_only();                    // A: tile complete
            fstamp(p);
            if (LAST) {
                // the layer's last hand-over is not deferred: its results are B fragments of the next layer's first GEMM
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    f32x4 n0, n1;
                    tile_dw(nt, handoff_wd(p, nt, wb + DWOFF + NTP * WDF), n0, n1, MODE != 2 && a.relu_dw);
                    if (nt == 0) handoff_out(p, nt, n0, n1, integral_constant<int, MODE == 2 ? -1 : C / 16 - NTP>{});
                    else handoff_out(p, nt, n0, n1, integral_constant<int, MODE == 2 ? -1 : C / 16 - NTP + 1>{});
                }
                barrier_lds_only();
            }
        };
        pass(0, std::false_type{}, std::false_type{});
        for (int p = 1; p < NPASS - 1; ++p) pass(p, std::false_type{}, std::true_type{});
        {
            // (the pass index stays a run-time value in the peeled last pass too: with a constant the LDS addresses of its weight
            // block become literals beyond the 64 KB reach of ds_read's immediate offset, one address register per fragment)
            int p_last = NPASS - 1;
            asm volatile("" : "+s"(p_last));
            pass(p_last, std::true_type{}, std::true_type{});
        }
        if (MODE == 1) {
            // ---- the 64 correlation channels = input chunks 16..19 of layer 1: hand-over, two chunks a round, into ds[][]
            const float* wdc = Zr + (NPASS & 1) * ZS;           // (copied in during the last pass)
#pragma unroll
            for (int r = 0; r < NTZ / 2; ++r) {
                tile_put(0, cacc[0][2 * r], cacc[1][2 * r]);
                tile_put(1, cacc[0][2 * r + 1], cacc[1][2 * r + 1]);
                __syncthreads();
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    // (straight into the registers layer 1 reads them from: 160 of its B-fragment registers instead of 128 + 32 reloaded every pass)
                    tile_dw(nt, wdc + (2 * r + nt) * WDF, ds[2 * r + nt][0], ds[2 * r + nt][1], a.relu_dw);
                }
                __syncthreads();
            }
        }
        if (MODE == 2 && lk == 0) {             // lanes lk == 0 hold channels 0..3 of their pixel
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
        stamp();
    };

    layer(integral_constant<int, C>{}, integral_constant<int, 1>{}, b.W[0], b.W[1]);
    layer(integral_constant<int, CC>{}, integral_constant<int, 0>{}, b.W[1], b.W[2]);
    lds_copy_async<16 * PCH>(b.P_Wpk, Zr, wave, lane);      // (Zr: last read in layer 0's correlation hand-over; complete at the next barrier)
    layer(integral_constant<int, C>{}, integral_constant<int, 0>{}, b.W[2], b.W[3]);
    layer(integral_constant<int, C>{}, integral_constant<int, 2>{}, b.W[3], nullptr);
}

}  // namespace fear
