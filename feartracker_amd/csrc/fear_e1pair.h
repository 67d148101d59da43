// fear_e1pair.h — two consecutive e1 inverted-residual blocks (no expansion: depthwise 3x3 + ReLU, pointwise C -> C, + input) of
// the 64x64 / 24-channel stage as ONE kernel.  Reference: model/blocks.py:8-42 builds them from mobile_cv's fbnet_c stage table
// (stage 2, blocks 2 and 3: t = 1, c = 24, stride 1); in the engine's layer-wise plan each is dw_conv + pw (+ residual).
//
// Each of the two blocks alone is a pure memory stream (1.6 GFLOP against 100 MB in + 100 MB out per 256 crops: 48-53 us, the
// tile kernel's workgroups spend 3 of their 7 us waiting for the input tile).  Fused, the intermediate map never leaves the CU:
//   input tile 20x20 (the 16x16 output tile + a 2-pixel halo), all 24 channels        global -> LDS   (38 KB, zero outside the map)
//   block A on the 18x18 region the second depthwise reads                            LDS -> LDS      (31 KB, zero outside the map)
//   block B on the 16x16 tile                                                         LDS -> global
// i.e. half the HBM traffic for 27 % more arithmetic in block A (324 instead of 256 pixels per tile) — arithmetic these blocks
// have to spare.  73 KB of LDS per workgroup: two workgroups per CU, one loading while the other computes.
//
// Work split (512 threads = 8 waves): block A's 324 pixels are 21 m-tiles of 16 pixels (18 row segments of 16 columns + 3 m-tiles for
// the two remaining columns), dealt round-robin (3, 3, 3, 3, 3, 2, 2, 2); block B's 16 rows are 2 m-tiles per wave.  Per m-tile a lane (pixel li, group lk)
// computes the depthwise of channel quad lk (chunk 0: channels 0-15) as one float4 and of channels 16 + 2 lk, + 1 (chunk 1: the
// remaining 8 channels) as one float2 — those ARE the B fragments of the projection, 4 + 2 MFMA k-steps per 16 output channels
// (a 16-channel second chunk would spend half its depthwise lanes and MFMA steps on padding).
// LDS tiles are kept as six planes [channel quad][pixel][4 floats]: a wave's 16 pixels are 16 consecutive 16-byte units of a
// plane (conflict free, MI355X_MICROARCH.md §LDS), the float2 reads of chunk 1 are the two halves of planes 4 and 5.
#pragma once
#include <type_traits>
#include <vector>

#ifndef E1P_ABL
#define E1P_ABL 0      // tools/kbench ablations: 1 no block A, 2 no block B, 4 no input loads, 8 stamps
#endif
#ifndef E1P_NA
#define E1P_NA 3       // of a thread's five prefetch loads, [0, E1P_NA) are in flight during block A and the rest during block B
#endif
#ifndef E1P_PERM
#define E1P_PERM 1     // 1: a wave's 64 fill slots are dealt to its lanes with stride 9 (lane 8g + j takes slot 8 ((g + j) mod 8) + j) and the plane
                       // pairs sit 48 B beyond a multiple of 256 B apart: every 8-lane store group then hits 8 different 16-byte bank slots
                       // (conflict free; a wave's loads are still one contiguous 1 KB).  0: slots in lane order, pairs 32 B apart: 2-way
#endif
#ifndef E1P_SKEW
#define E1P_SKEW (E1P_PERM ? 12 : 8)     // floats between the plane PAIRS of the input tile beyond a multiple of 256 B (E1PairGeom::xb); 0 = round 4's layout (tools/kbench A/B)
#endif

namespace fear {

struct E1PairArgs {
    const float* X;      // [crops][H][W][24]
    float* Y;            // [crops][H][W][24]
    const float* Wpk;    // two blocks of E1PairGeom::WBLK floats (headchain-style host packing: e1pair_pack_block)
    int H, W, tiles_x, tiles_y;
    int tpw;             // consecutive tiles per workgroup (0 = 1); launch with crops * tiles_x * tiles_y / tpw workgroups
};

struct E1PairGeom {
    static constexpr int C = 24, T = 16, XW = T + 4, MW = T + 2, NQ = C / 4;
    static constexpr int XPL = XW * XW * 4;                        // floats per plane of the input tile (1600 = 25 x 64)
    // Plane q of the input tile starts xb(q) floats into it.  The tile fill stores the loads in their global order (channel quad
    // fastest: a wave's 64 loads are 1 KB of consecutive addresses), so the 8 lanes of a ds_write_b128 service group hold the six
    // quads of ONE pixel + two of the next; with every plane a multiple of 128 B apart those six land on the same four banks — a
    // 6-way conflict on every group: 1 600 LDS cycles per fill against 300, the whole of the kernel's SQ_LDS_BANK_CONFLICT (0.27 of
    // its LDS cycles, profiles/r05_sq_counters.txt).  The ds_read_b128 groups mix quads (0, 1) and (2, 3) — those pairs must stay a
    // multiple of 256 B apart — and the float2 reads pair the halves of quad 4 and of quad 5; between the PAIRS the distance is free:
    // 32 B more per pair leaves 2-way conflicts (600 cycles).
    static constexpr int XSKEW = E1P_SKEW;
    static constexpr int xb(int q) { return q * XPL + (q >> 1) * XSKEW; }
    static constexpr int XT_FLOATS = NQ * XPL + 2 * XSKEW;
    static constexpr int NSLOT = NQ * XW * XW, NIT = (NSLOT + 511) / 512;     // float4 slots of the input tile, per thread
    static constexpr int MPL = (MW * MW * 4 + 63) / 64 * 64;       // 1296 -> 1344
    static constexpr int NPIX_A = MW * MW, NMT_A = (NPIX_A + 15) / 16, MTA = (NMT_A + 7) / 8;
    // one block, packed: 4 projection fragments [chunk 0, out tile 0 | chunk 0, tile 1 | chunk 1, tile 0 | chunk 1, tile 1] of 256
    // floats (lane l: n = 16 nt + (l & 15); chunk 0: k = 4 (l >> 4) + i; chunk 1: k = 16 + 2 (l >> 4) + i for i < 2, zero else),
    // depthwise taps of chunk 0 [9][16] and of chunk 1 [9][8], depthwise bias [24], projection bias [24 -> 32]
    static constexpr int FRAG = 4 * 256, TAP0 = FRAG, TAP1 = TAP0 + 9 * 16, BD = TAP1 + 9 * 8, BP = BD + 24, WBLK = BP + 32;
    static constexpr int WS = WBLK - FRAG;                         // the part that lives in LDS (taps and biases: 272 floats)
    static_assert(WS % 4 == 0 && WBLK % 4 == 0, "16-byte granules");
    static constexpr int LDS_FLOATS = XT_FLOATS + NQ * MPL + 2 * WS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;               // 72 896 B: two workgroups per CU
};

// host side: dw = [24][9] depthwise weights, bd = [24] or nullptr, pw = [24][24] (out, in), bp = [24]
inline void e1pair_pack_block(std::vector<float>& out, const float* dw, const float* bd, const float* pw, const float* bp) {
    using G = E1PairGeom;
    for (int c = 0; c < 2; ++c)
        for (int nt = 0; nt < 2; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) {
                    const int n = nt * 16 + (l & 15);
                    const int k = c == 0 ? (l >> 4) * 4 + i : (i < 2 ? 16 + 2 * (l >> 4) + i : G::C);
                    out.push_back(n < G::C && k < G::C ? pw[n * G::C + k] : 0.f);
                }
    for (int t = 0; t < 9; ++t)
        for (int ch = 0; ch < 16; ++ch) out.push_back(dw[ch * 9 + t]);
    for (int t = 0; t < 9; ++t)
        for (int ch = 16; ch < 24; ++ch) out.push_back(dw[ch * 9 + t]);
    for (int ch = 0; ch < 24; ++ch) out.push_back(bd ? bd[ch] : 0.f);
    for (int ch = 0; ch < 32; ++ch) out.push_back(ch < 24 ? bp[ch] : 0.f);
}

__device__ __forceinline__ void pk_fma2(f32x2& d, const f32x2& a, const f32x2& b) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// ReLU of the depthwise result + the 24 -> 24 projection of the 16-pixel m-tile the wave's lanes form:
// acc[nt] = channels 16 nt + 4 lk .. + 3 of pixel li; the accumulators start from the projection bias
__device__ __forceinline__ void e1_project(f32x4 d4, f32x2 d2, const f32x4 (&wp)[4], const f32x4 (&bias)[2], f32x4 (&acc)[2]) {
    d4.x = fmaxf(d4.x, 0.f); d4.y = fmaxf(d4.y, 0.f); d4.z = fmaxf(d4.z, 0.f); d4.w = fmaxf(d4.w, 0.f);
    d2.x = fmaxf(d2.x, 0.f); d2.y = fmaxf(d2.y, 0.f);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        f32x4 a = bias[nt];
#pragma unroll
        for (int q = 0; q < 4; ++q) a = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[nt][q], d4[q], a, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) a = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[2 + nt][q], d2[q], a, 0, 0, 0);
        acc[nt] = a;
    }
}

// which of its wave's 64 consecutive fill slots a thread takes (E1P_PERM)
__device__ __forceinline__ int e1_fill_slot(int tid) {
    if (!E1P_PERM) return tid;
    const int l = tid & 63, g = l >> 3, j = l & 7;
    return (tid & ~63) | (8 * ((g + j) & 7) + j);
}

// The input tile: six planes of 400 pixels, 2400 float4 slots over 512 threads (slots IT0 .. IT1 - 1 of each thread); a slot outside
// the map reads zeros (the buffer load's out-of-range value: the depthwise's zero padding), all loads in flight together.
// (Free functions on the register array, not lambdas: captured by reference in a closure that is called from two places, hipcc kept
// the array in scratch memory.  The slot arithmetic is recomputed at every call — hoisted out of the tile loop it was ~70 registers
// of offsets and predicates alive across both blocks; the empty asm makes the thread index opaque.)
template <int IT0, int IT1>
__device__ __forceinline__ void e1_issue_loads(const E1PairArgs& a, unsigned tix, int tid, f32x4 (&xv)[E1PairGeom::NIT]) {
    using G = E1PairGeom;
    constexpr int T = G::T, XW = G::XW, NQ = G::NQ, C = G::C;
    const int tiles = a.tiles_x * a.tiles_y;
    const long crop = tix / tiles;
    const int tile = tix % tiles;
    const int ox0 = (tile % a.tiles_x) * T, oy0 = (tile / a.tiles_x) * T;
    const float* Xc = a.X + crop * a.H * a.W * C;
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xc), 0, a.H * a.W * C * 4, 0x00020000);
    int t = e1_fill_slot(tid);
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const int s = it * 512 + t;                   // quad fastest: a wave's 64 loads are 1 KB of consecutive addresses
        const int pix = s / NQ, q = s - pix * NQ;
        const int py = pix / XW, px = pix - py * XW;
        const int gy = oy0 - 2 + py, gx = ox0 - 2 + px;
        const bool ok = s < G::NSLOT && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        if (E1P_ABL & 4) { xv[it] = (f32x4){1.f, 2.f, 3.f, 4.f}; continue; }
        xv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img, ok ? ((gy * a.W + gx) * C + q * 4) * 4 : (int)0x80000000, 0, 0));
    }
}
template <int IT0, int IT1>
__device__ __forceinline__ void e1_commit_tile(float* Xt, int tid, const f32x4 (&xv)[E1PairGeom::NIT]) {
    using G = E1PairGeom;
    int t = e1_fill_slot(tid);
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const int s = it * 512 + t;
        const int pix = s / G::NQ, q = s - pix * G::NQ;
        if (s < G::NSLOT) *reinterpret_cast<f32x4*>(Xt + q * G::XPL + (q >> 1) * G::XSKEW + pix * 4) = xv[it];
    }
}

__global__ __launch_bounds__(512, 4) void e1pair_kernel(E1PairArgs a) {
    using G = E1PairGeom;
    constexpr int T = G::T, XW = G::XW, MW = G::MW, XPL = G::XPL, MPL = G::MPL, NQ = G::NQ, C = G::C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Xt = lds;                   // [NQ] planes at G::xb(q)
    float* const Mt = lds + G::XT_FLOATS;    // [NQ][MPL]
    float* const WSl = Mt + NQ * MPL;        // [2][WS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int tiles = a.tiles_x * a.tiles_y;
    // A workgroup can walk a.tpw consecutive tiles with the loads of tile t + 1 issued right after the barrier that publishes tile t
    // (they land in registers while block A computes and go to LDS behind the A -> B barrier, in front of block B).  Built in round 6
    // on the hypothesis that the workgroup's wait for its input tile (3 of its 10 us) was what kept the kernel at 0.25 of the fp32
    // peak; measured (tools/kbench, profiles/r06_e1pair_kbench.txt): 94.8 us at one tile per workgroup, 95 / 96 / 97.5 / 121 us at
    // 2 / 4 / 8 / 16 — the two resident workgroups already cover each other's loads, the kernel is bound by instruction issue.
    // The engine launches one tile per workgroup (FEAR_E1PAIR_TPW_MAX).
    const int tpw = a.tpw > 0 ? a.tpw : 1;
    const unsigned t_first = xcd_tile_index(blockIdx.x, gridDim.x) * (unsigned)tpw;

    // ---- the input tile: six planes of 400 pixels, 2400 float4 slots over 512 threads; a slot outside the map reads zeros (the
    //      buffer load's out-of-range value: the depthwise's zero padding), all loads in flight together
    constexpr int NIT = G::NIT;
    constexpr int NA = E1P_NA;                    // loads [0, NA) of the next tile are in flight during block A, [NA, NIT) during block B
    f32x4 xv0[NIT];                          // (the first tile's registers are not the prefetch's: one live range over prologue and loop
    e1_issue_loads<0, NIT>(a, t_first, tid, xv0);   //  was spilled as a whole)
    // taps and biases go to LDS; a block's four projection fragments are fetched (L2-resident: every workgroup reads the same 4 KB)
    // at the head of its phase — holding both blocks' across the tile loop did not fit beside the prefetched tile
    auto load_wp = [&](int b, f32x4 (&wp)[4]) {
#pragma unroll
        for (int f = 0; f < 4; ++f) wp[f] = *reinterpret_cast<const f32x4*>(a.Wpk + b * G::WBLK + f * 256 + lane * 4);
    };
    if (tid < 2 * G::WS / 4) {
        const int b = tid / (G::WS / 4), j = tid - b * (G::WS / 4);
        *reinterpret_cast<f32x4*>(WSl + b * G::WS + j * 4) = *reinterpret_cast<const f32x4*>(a.Wpk + b * G::WBLK + G::FRAG + j * 4);
    }
    e1_commit_tile<0, NIT>(Xt, tid, xv0);

    for (int ti = 0; ti < tpw; ++ti) {
        const unsigned tix = t_first + ti;
        const long crop = tix / tiles;
        const int tile = tix % tiles;
        const int ox0 = (tile % a.tiles_x) * T, oy0 = (tile / a.tiles_x) * T;
        float* Yc = a.Y + crop * a.H * a.W * C;
        __syncthreads();                         // tile ti complete in Xt; every wave is through block B of tile ti - 1 (Mt is free)
        f32x4 xv[NIT];
        // (lane coordinates opaque per tile: hoisted out of the tile loop, the m-tiles' pixel offsets and masks were 26 spilled registers)
        int li_o = li, lk_o = lk;
        asm volatile("" : "+v"(li_o), "+v"(lk_o));
        if (ti + 1 < tpw) e1_issue_loads<0, NA>(a, tix + 1, tid, xv);

        // ---- block A on the 18x18 region around the tile -> Mt (zero outside the map: block B's depthwise pads with zeros)
        // The wave's two or three m-tiles go through the depthwise TOGETHER, tap by tap: a tap's weights are read from LDS where
        // they are used and dropped — the same number of LDS reads as loading all 54 tap registers once per phase, without
        // holding them beside the prefetched tile (round 4's form, one m-tile at a time from register-resident taps, spilled the
        // prefetch to scratch and waited for it on the spot).
        if (!(E1P_ABL & 1)) {
            const float* ws = WSl;
            f32x4 wp[4];
            load_wp(0, wp);
            auto block_a = [&](auto nm_tag) {
                constexpr int NM = decltype(nm_tag)::value;
                // m-tiles 0 .. 17 are the first 16 columns of a row (16 consecutive 16-byte units of a plane: conflict-free LDS reads and
                // stores), m-tiles 18 .. 20 the two remaining columns of the 18 rows, two pixels per row (row-major m-tiles over the
                // 18-wide region straddled a row end in almost every m-tile: 35 % of the kernel's LDS cycles were bank conflicts)
                int pc[NM], cpix[NM];
                bool valid[NM];
                float inside[NM];
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    const int mt = wave + 8 * i;
                    const int j = (mt - MW) * 16 + li_o;                       // (mt >= 18)
                    valid[i] = mt < MW || j < 2 * MW;
                    const int my = mt < MW ? mt : (valid[i] ? j >> 1 : 0), mx = mt < MW ? li_o : 16 + (j & 1);
                    pc[i] = my * MW + mx;
                    cpix[i] = (my + 1) * XW + mx + 1;
                    const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
                    inside[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? 1.f : 0.f;
                }
                f32x4 d4[NM];
                f32x2 d2[NM];
                {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(ws + (G::BD - G::FRAG) + lk_o * 4);
                    const f32x2 b1 = *reinterpret_cast<const f32x2*>(ws + (G::BD - G::FRAG) + 16 + lk_o * 2);
#pragma unroll
                    for (int i = 0; i < NM; ++i) { d4[i] = b0; d2[i] = b1; }
                }
                const float* s0 = Xt + lk_o * XPL + (lk_o >> 1) * G::XSKEW;
                const float* s1 = Xt + (4 + (lk_o >> 1)) * XPL + 2 * G::XSKEW + (lk_o & 1) * 2;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + (G::TAP0 - G::FRAG) + k * 16 + lk_o * 4);
                    const f32x2 w2 = *reinterpret_cast<const f32x2*>(ws + (G::TAP1 - G::FRAG) + k * 8 + lk_o * 2);
                    const int off = ((k / 3 - 1) * XW + (k % 3 - 1)) * 4;
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(s0 + cpix[i] * 4 + off);
                        const f32x2 v2 = *reinterpret_cast<const f32x2*>(s1 + cpix[i] * 4 + off);
                        pk_fma4(d4[i], v4, w4);
                        pk_fma2(d2[i], v2, w2);
                    }
                }
                f32x4 bpv[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bpv[nt] = *reinterpret_cast<const f32x4*>(ws + (G::BP - G::FRAG) + (nt * 4 + lk_o) * 4);
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    f32x4 acc[2];
                    e1_project(d4[i], d2[i], wp, bpv, acc);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (nt == 1 && lk_o >= 2) continue;                    // channels 24 .. 31 do not exist
                        const int qd = nt * 4 + lk_o;
                        const f32x4 v = (acc[nt] + *reinterpret_cast<const f32x4*>(Xt + qd * XPL + (qd >> 1) * G::XSKEW + cpix[i] * 4)) * inside[i];   // (finite values: x * 0 = 0)
                        if (valid[i]) *reinterpret_cast<f32x4*>(Mt + qd * MPL + pc[i] * 4) = v;
                    }
                }
            };
            if (wave + 16 < G::NMT_A) block_a(std::integral_constant<int, 3>{});      // wave-uniform: 21 m-tiles over 8 waves
            else block_a(std::integral_constant<int, 2>{});
        }
        __syncthreads();                         // Mt complete; nobody reads Xt any more
        if (ti + 1 < tpw) { e1_commit_tile<0, NA>(Xt, tid, xv); e1_issue_loads<NA, NIT>(a, tix + 1, tid, xv); }

        // ---- block B on the tile's 16 rows (two per wave, sharing their depthwise reads) -> global
        if (!(E1P_ABL & 2)) {
            const float* ws = WSl + G::WS;
            f32x4 wp[4];
            load_wp(1, wp);
            f32x4 bpv[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bpv[nt] = *reinterpret_cast<const f32x4*>(ws + (G::BP - G::FRAG) + (nt * 4 + lk_o) * 4);
            const int row0 = wave * 2;
            const int cpix = (row0 + 1) * MW + li_o + 1;
            // depthwise of the wave's two rows, input row by input row: a value feeds row 0 through tap row iy and row 1 through tap
            // row iy - 1, so two tap rows (read from LDS as they come up) are live at a time instead of all nine taps
            f32x4 d4[2];
            f32x2 d2[2];
            d4[0] = d4[1] = *reinterpret_cast<const f32x4*>(ws + (G::BD - G::FRAG) + lk_o * 4);
            d2[0] = d2[1] = *reinterpret_cast<const f32x2*>(ws + (G::BD - G::FRAG) + 16 + lk_o * 2);
            {
                const float* s0 = Mt + lk_o * MPL + cpix * 4;
                const float* s1 = Mt + (4 + (lk_o >> 1)) * MPL + cpix * 4 + (lk_o & 1) * 2;
                f32x4 w4[2][3];
                f32x2 w2[2][3];
#pragma unroll
                for (int iy = 0; iy < 4; ++iy) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        if (iy < 3) {
                            w4[iy & 1][kx] = *reinterpret_cast<const f32x4*>(ws + (G::TAP0 - G::FRAG) + (iy * 3 + kx) * 16 + lk_o * 4);
                            w2[iy & 1][kx] = *reinterpret_cast<const f32x2*>(ws + (G::TAP1 - G::FRAG) + (iy * 3 + kx) * 8 + lk_o * 2);
                        }
                        const int off = ((iy - 1) * MW + (kx - 1)) * 4;
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(s0 + off);
                        const f32x2 v2 = *reinterpret_cast<const f32x2*>(s1 + off);
                        if (iy < 3) { pk_fma4(d4[0], v4, w4[iy & 1][kx]); pk_fma2(d2[0], v2, w2[iy & 1][kx]); }
                        if (iy >= 1) { pk_fma4(d4[1], v4, w4[(iy - 1) & 1][kx]); pk_fma2(d2[1], v2, w2[(iy - 1) & 1][kx]); }
                    }
                }
            }
            const unsigned ylane = (unsigned)(li_o * C + lk_o * 4);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f32x4 acc[2];
                e1_project(d4[r], d2[r], wp, bpv, acc);
                float* yrow = Yc + ((long)(oy0 + row0 + r) * a.W + ox0) * C;   // uniform
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (nt == 1 && lk_o >= 2) continue;
                    const int qd = nt * 4 + lk_o;
                    const f32x4 v = acc[nt] + *reinterpret_cast<const f32x4*>(Mt + qd * MPL + (cpix + r * MW) * 4);
                    *reinterpret_cast<f32x4*>(yrow + (ylane + (unsigned)(nt * 16))) = v;
                }
            }
        }
        if (ti + 1 < tpw) e1_commit_tile<NA, NIT>(Xt, tid, xv);
    }
}

}  // namespace fear
