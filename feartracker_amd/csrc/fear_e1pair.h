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
#include <vector>

#ifndef E1P_ABL
#define E1P_ABL 0      // tools/kbench ablations: 1 no block A, 2 no block B, 4 no input loads, 8 stamps
#endif

namespace fear {

struct E1PairArgs {
    const float* X;      // [crops][H][W][24]
    float* Y;            // [crops][H][W][24]
    const float* Wpk;    // two blocks of E1PairGeom::WBLK floats (headchain-style host packing: e1pair_pack_block)
    int H, W, tiles_x, tiles_y;
};

struct E1PairGeom {
    static constexpr int C = 24, T = 16, XW = T + 4, MW = T + 2, NQ = C / 4;
    static constexpr int XPL = XW * XW * 4;                        // floats per plane of the input tile (1600 = 25 x 64)
    static constexpr int MPL = (MW * MW * 4 + 63) / 64 * 64;       // 1296 -> 1344
    static constexpr int NPIX_A = MW * MW, NMT_A = (NPIX_A + 15) / 16, MTA = (NMT_A + 7) / 8;
    // one block, packed: 4 projection fragments [chunk 0, out tile 0 | chunk 0, tile 1 | chunk 1, tile 0 | chunk 1, tile 1] of 256
    // floats (lane l: n = 16 nt + (l & 15); chunk 0: k = 4 (l >> 4) + i; chunk 1: k = 16 + 2 (l >> 4) + i for i < 2, zero else),
    // depthwise taps of chunk 0 [9][16] and of chunk 1 [9][8], depthwise bias [24], projection bias [24 -> 32]
    static constexpr int FRAG = 4 * 256, TAP0 = FRAG, TAP1 = TAP0 + 9 * 16, BD = TAP1 + 9 * 8, BP = BD + 24, WBLK = BP + 32;
    static constexpr int WS = WBLK - FRAG;                         // the part that lives in LDS (taps and biases: 272 floats)
    static_assert(WS % 4 == 0 && WBLK % 4 == 0, "16-byte granules");
    static constexpr int LDS_FLOATS = NQ * XPL + NQ * MPL + 2 * WS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;               // 72 832 B: two workgroups per CU
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

// The depthwise taps and bias of one block for this lane's channels (chunk 0: quad lk; chunk 1: channels 16 + 2 lk, + 1), read once
// per phase into registers: re-read per m-tile they were half of the kernel's LDS traffic, and that traffic — not HBM, not the ALU —
// was what bounded the first version (95 us against 108 for the two separate blocks).
struct E1Taps {
    f32x4 w0[9], b0;
    f32x2 w1[9], b1;
};
__device__ __forceinline__ void e1_load_taps(E1Taps& t, const float* __restrict__ ws, int lk) {
    using G = E1PairGeom;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        t.w0[k] = *reinterpret_cast<const f32x4*>(ws + (G::TAP0 - G::FRAG) + k * 16 + lk * 4);
        t.w1[k] = *reinterpret_cast<const f32x2*>(ws + (G::TAP1 - G::FRAG) + k * 8 + lk * 2);
    }
    t.b0 = *reinterpret_cast<const f32x4*>(ws + (G::BD - G::FRAG) + lk * 4);
    t.b1 = *reinterpret_cast<const f32x2*>(ws + (G::BD - G::FRAG) + 16 + lk * 2);
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

// depthwise 3x3 of NR vertically adjacent pixels (centres cpix, cpix + SW, ...) of a plane-layout tile with SW pixels per row:
// the (NR + 2) x 3 input reads are shared by the rows
template <int SW, int SPL, int NR>
__device__ __forceinline__ void e1_depthwise(const float* __restrict__ src, int cpix, const E1Taps& t, int lk, f32x4 (&d4)[NR], f32x2 (&d2)[NR]) {
    const float* s0 = src + lk * SPL + cpix * 4;
    const float* s1 = src + (4 + (lk >> 1)) * SPL + cpix * 4 + (lk & 1) * 2;
#pragma unroll
    for (int r = 0; r < NR; ++r) { d4[r] = t.b0; d2[r] = t.b1; }
#pragma unroll
    for (int iy = 0; iy < NR + 2; ++iy)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int off = ((iy - 1) * SW + (kx - 1)) * 4;
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(s0 + off);
            const f32x2 v2 = *reinterpret_cast<const f32x2*>(s1 + off);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int ky = iy - r;
                if (ky >= 0 && ky < 3) {
                    pk_fma4(d4[r], v4, t.w0[ky * 3 + kx]);
                    pk_fma2(d2[r], v2, t.w1[ky * 3 + kx]);
                }
            }
        }
}

__global__ __launch_bounds__(512, 4) void e1pair_kernel(E1PairArgs a) {
    using G = E1PairGeom;
    constexpr int T = G::T, XW = G::XW, MW = G::MW, XPL = G::XPL, MPL = G::MPL, NQ = G::NQ, C = G::C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Xt = lds;                   // [NQ][XPL]
    float* const Mt = lds + NQ * XPL;        // [NQ][MPL]
    float* const WSl = Mt + NQ * MPL;        // [2][WS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int tiles = a.tiles_x * a.tiles_y;
    const unsigned tix = xcd_tile_index(blockIdx.x, gridDim.x);
    const long crop = tix / tiles;
    const int tile = tix % tiles;
    const int ox0 = (tile % a.tiles_x) * T, oy0 = (tile / a.tiles_x) * T;
    const float* Xc = a.X + crop * a.H * a.W * C;
    float* Yc = a.Y + crop * a.H * a.W * C;

    // ---- the input tile: six planes of 400 pixels, 2400 float4 slots over 512 threads; a slot outside the map reads zeros (the
    //      buffer load's out-of-range value: the depthwise's zero padding), all loads in flight together
    constexpr int NSLOT = NQ * XW * XW, NIT = (NSLOT + 511) / 512;
    f32x4 xv[NIT];
    {
        const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xc), 0, a.H * a.W * C * 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int s = it * 512 + tid;                 // quad fastest: a wave's 64 loads are 1 KB of consecutive addresses
            const int pix = s / NQ, q = s - pix * NQ;
            const int py = pix / XW, px = pix - py * XW;
            const int gy = oy0 - 2 + py, gx = ox0 - 2 + px;
            const bool ok = s < NSLOT && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            if (E1P_ABL & 4) { xv[it] = (f32x4){1.f, 2.f, 3.f, 4.f}; continue; }
            xv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img, ok ? ((gy * a.W + gx) * C + q * 4) * 4 : (int)0x80000000, 0, 0));
        }
    }
    // the projection fragments of both blocks stay in registers, taps and biases go to LDS
    f32x4 wp[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int f = 0; f < 4; ++f) wp[b][f] = *reinterpret_cast<const f32x4*>(a.Wpk + b * G::WBLK + f * 256 + lane * 4);
    if (tid < 2 * G::WS / 4) {
        const int b = tid / (G::WS / 4), j = tid - b * (G::WS / 4);
        *reinterpret_cast<f32x4*>(WSl + b * G::WS + j * 4) = *reinterpret_cast<const f32x4*>(a.Wpk + b * G::WBLK + G::FRAG + j * 4);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int s = it * 512 + tid;
        const int pix = s / NQ, q = s - pix * NQ;
        if (s < NSLOT) *reinterpret_cast<f32x4*>(Xt + q * XPL + pix * 4) = xv[it];
    }
    __syncthreads();

    // ---- block A on the 18x18 region around the tile -> Mt (zero outside the map: block B's depthwise pads with zeros)
    if (!(E1P_ABL & 1)) {
        const float* ws = WSl;
        E1Taps taps;
        e1_load_taps(taps, ws, lk);
        f32x4 bpv[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bpv[nt] = *reinterpret_cast<const f32x4*>(ws + (G::BP - G::FRAG) + (nt * 4 + lk) * 4);
#pragma unroll
        for (int i = 0; i < G::MTA; ++i) {
            const int mt = wave + 8 * i;
            if (mt >= G::NMT_A) break;                                 // wave-uniform
            // m-tiles 0 .. 17 are the first 16 columns of a row (16 consecutive 16-byte units of a plane: conflict-free LDS reads and
            // stores), m-tiles 18 .. 20 the two remaining columns of the 18 rows, two pixels per row (row-major m-tiles over the
            // 18-wide region straddled a row end in almost every m-tile: 35 % of the kernel's LDS cycles were bank conflicts)
            const int j = (mt - MW) * 16 + li;                         // (mt >= 18)
            const bool valid = mt < MW || j < 2 * MW;
            const int my = mt < MW ? mt : (valid ? j >> 1 : 0), mx = mt < MW ? li : 16 + (j & 1);
            const int pc = my * MW + mx;
            const int cpix = (my + 1) * XW + mx + 1;
            f32x4 acc[2], d4[1];
            f32x2 d2[1];
            e1_depthwise<XW, XPL, 1>(Xt, cpix, taps, lk, d4, d2);
            e1_project(d4[0], d2[0], wp[0], bpv, acc);
            const int gy = oy0 - 1 + my, gx = ox0 - 1 + mx;
            const float inside = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? 1.f : 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (nt == 1 && lk >= 2) continue;                      // channels 24 .. 31 do not exist
                const int qd = nt * 4 + lk;
                const f32x4 v = (acc[nt] + *reinterpret_cast<const f32x4*>(Xt + qd * XPL + cpix * 4)) * inside;   // (finite values: x * 0 = 0)
                if (valid) *reinterpret_cast<f32x4*>(Mt + qd * MPL + pc * 4) = v;
            }
        }
    }
    __syncthreads();

    // ---- block B on the tile's 16 rows (two per wave, sharing their depthwise reads) -> global
    if (!(E1P_ABL & 2)) {
        const float* ws = WSl + G::WS;
        E1Taps taps;
        e1_load_taps(taps, ws, lk);
        f32x4 bpv[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bpv[nt] = *reinterpret_cast<const f32x4*>(ws + (G::BP - G::FRAG) + (nt * 4 + lk) * 4);
        const int row0 = wave * 2;
        const int cpix = (row0 + 1) * MW + li + 1;
        f32x4 d4[2];
        f32x2 d2[2];
        e1_depthwise<MW, MPL, 2>(Mt, cpix, taps, lk, d4, d2);
        const unsigned ylane = (unsigned)(li * C + lk * 4);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 acc[2];
            e1_project(d4[r], d2[r], wp[1], bpv, acc);
            float* yrow = Yc + ((long)(oy0 + row0 + r) * a.W + ox0) * C;   // uniform
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (nt == 1 && lk >= 2) continue;
                const int qd = nt * 4 + lk;
                const f32x4 v = acc[nt] + *reinterpret_cast<const f32x4*>(Mt + qd * MPL + (cpix + r * MW) * 4);
                *reinterpret_cast<f32x4*>(yrow + (ylane + (unsigned)(nt * 16))) = v;
            }
        }
    }
}

}  // namespace fear
