// kbench — standalone micro-bench for the fused 16x16 block kernels (development tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kbench tools/kbench.hip
// Run on the GPU box: tools/kbench [crops=256] [iters=20]   (weights are random: timing only)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef KHDR
#define KHDR "../feartracker_amd/csrc/fear_kernels.h"     // -DKHDR='"/path/to/variant.h"' benches a saved variant of the header
#endif
#include KHDR
#include "../feartracker_amd/csrc/fear_e1pair.h"
#include "../feartracker_amd/csrc/fear_chain32.h"
#ifdef FEAR_E1PAIR_REF
#include FEAR_E1PAIR_REF        // round 5's kernel with its symbols renamed (tools/_kb/e1pair_r5.h, made by: git show <rev>:...fear_e1pair.h | sed ...)
#endif

using namespace fear;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

template <typename K>
static double time_kernel(K k, int lds, int crops, int iters, Ir2Args a, int threads = 512) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(crops), dim3(threads), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(crops), dim3(threads), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / iters;
}

template <int CIN, int COUT>
static Ir2Args make_args(int crops, size_t packed_floats) {
    Ir2Args a{};
    a.ldx = CIN; a.ldr = COUT; a.ldy = COUT;
    a.X = dev_rand((size_t)crops * 256 * CIN, 2.f);
    a.Wpk = dev_rand(packed_floats, 0.01f);
    a.bp = dev_rand(COUT, 0.2f);
    float* y;
    CK(hipMalloc(&y, (size_t)crops * 256 * COUT * sizeof(float)));
    a.Y = y; a.relu_dw = 1; a.relu_out = 0;
    return a;
}

template <int CIN, int CEXP, int COUT, int KS, int ST, int TW, int TH, bool EXPAND, int MINW>
static void bench_tile(const char* tag, int crops, int iters, int hw) {
    using G = IrT2Geom<CIN, CEXP, COUT, KS, ST, TW, TH, EXPAND>;
    if (G::LDS_BYTES > 160 * 1024) { printf("%-24s skipped: LDS %d B\n", tag, G::LDS_BYTES); return; }
    const int ho = hw / ST;
    const double flops = 2.0 * ((EXPAND ? (double)CIN * CEXP * hw * hw : 0.0) + ((double)CEXP * KS * KS + (double)CEXP * COUT) * ho * ho) * crops;
    IrT2Args t{};
    Ir2Args& a = t.b;
    a.ldx = CIN; a.ldr = COUT; a.ldy = COUT;
    a.X = dev_rand((size_t)crops * hw * hw * CIN, 2.f);
    a.Wpk = dev_rand((size_t)G::NCHUNK * (G::AP + G::BP), 0.01f);
    a.bp = dev_rand(64, 0.2f);
    float* y;
    CK(hipMalloc(&y, (size_t)crops * ho * ho * COUT * sizeof(float)));
    a.Y = y; a.relu_dw = 1; a.relu_out = 0;
    if (!EXPAND && CIN == COUT && ST == 1) a.R = a.X;     // an e1 block's residual is its input
    float* dbg;
    CK(hipMalloc(&dbg, 80 * sizeof(float)));
    CK(hipMemset(dbg, 0, 80 * sizeof(float)));
    a.P_Y = dbg;
    t.H = hw; t.W = hw; t.tiles_x = ho / TW; t.tiles_y = ho / TH;
    auto k = ir_tile_v2_kernel<CIN, CEXP, COUT, KS, ST, TW, TH, EXPAND, MINW>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)crops * t.tiles_x * t.tiles_y);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(512), G::LDS_BYTES, 0, t);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(512), G::LDS_BYTES, 0, t);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    printf("%-24s fp32-tile  %8.1f us  %6.1f TF/s  (LDS %d B)\n", tag, us, flops / us * 1e-6, G::LDS_BYTES);
    if (FEAR_ABL & 4096) {
        float h[80];
        CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int w = 0; w < 8; w += 4)
            printf("   wave %d (block 1000): prologue %.2f | top barrier %.2f | phase A %.2f | mid barrier %.2f | phase B %.2f | phase C %.2f | end %.2f | epilogue %.2f us\n",
                   w, h[w * 10 + 6] * 0.01, h[w * 10] * 0.01, h[w * 10 + 1] * 0.01, h[w * 10 + 2] * 0.01, h[w * 10 + 3] * 0.01,
                   h[w * 10 + 4] * 0.01, h[w * 10 + 5] * 0.01, h[w * 10 + 7] * 0.01);
    }
}

// v2 against v4 (phase-overlapped) on the same tile: same inputs and packed weights, outputs compared, both timed
template <int CIN, int CEXP, int COUT, int KS, int ST, int TW, int TH, int MINW2, int MINW4>
static void bench_tile_v4(const char* tag, int crops, int iters, int hw) {
#ifdef FEAR_HAVE_V4
    using G = IrT2Geom<CIN, CEXP, COUT, KS, ST, TW, TH, true>;
    using G4 = IrT4Geom<CIN, CEXP, COUT, KS, ST, TW, TH>;
    if (G4::LDS_BYTES > 160 * 1024) { printf("%-24s skipped: LDS %d B\n", tag, G4::LDS_BYTES); return; }
    const int ho = hw / ST;
    const double flops = 2.0 * ((double)CIN * CEXP * hw * hw + ((double)CEXP * KS * KS + (double)CEXP * COUT) * ho * ho) * crops;
    IrT2Args t{};
    Ir2Args& a = t.b;
    a.ldx = CIN; a.ldr = COUT; a.ldy = COUT;
    a.X = dev_rand((size_t)crops * hw * hw * CIN, 2.f);
    a.Wpk = dev_rand((size_t)G::NCHUNK * (G::AP + G::BP), 0.3f);
    a.bp = dev_rand(64, 0.2f);
    const size_t ny = (size_t)crops * ho * ho * COUT;
    float *y2, *y4;
    CK(hipMalloc(&y2, ny * sizeof(float)));
    CK(hipMalloc(&y4, ny * sizeof(float)));
    CK(hipMemset(y2, 0, ny * sizeof(float)));
    CK(hipMemset(y4, 0xff, ny * sizeof(float)));
    a.relu_dw = 1; a.relu_out = 0;
    t.H = hw; t.W = hw; t.tiles_x = ho / TW; t.tiles_y = ho / TH;
    auto k2 = ir_tile_v2_kernel<CIN, CEXP, COUT, KS, ST, TW, TH, true, MINW2>;
    auto k4 = ir_tile_v4_kernel<CIN, CEXP, COUT, KS, ST, TW, TH, MINW4>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, G4::LDS_BYTES));
    const dim3 grid((unsigned)crops * t.tiles_x * t.tiles_y);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double us[2];
    for (int v = 0; v < 2; ++v) {
        a.Y = v ? y4 : y2;
        for (int i = 0; i < 3; ++i) {
            if (v) hipLaunchKernelGGL(k4, grid, dim3(512), G4::LDS_BYTES, 0, t);
            else hipLaunchKernelGGL(k2, grid, dim3(512), G::LDS_BYTES, 0, t);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) {
            if (v) hipLaunchKernelGGL(k4, grid, dim3(512), G4::LDS_BYTES, 0, t);
            else hipLaunchKernelGGL(k2, grid, dim3(512), G::LDS_BYTES, 0, t);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[v] = 1e3 * ms / iters;
    }
    std::vector<float> h2(ny), h4(ny);
    CK(hipMemcpy(h2.data(), y2, ny * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h4.data(), y4, ny * sizeof(float), hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    size_t bad = 0;
    for (size_t i = 0; i < ny; ++i) {
        const double dd = fabs((double)h2[i] - (double)h4[i]);
        if (!(dd <= 1e30)) { ++bad; continue; }
        if (dd > maxd) maxd = dd;
        if (fabs(h2[i]) > maxv) maxv = fabs(h2[i]);
    }
    printf("%-30s v2 %8.1f us %6.1f TF/s | v4 %8.1f us %6.1f TF/s (%+.1f%%) | max|v2-v4| %.3g of max|y| %.3g, non-finite %zu  (LDS %d B)\n", tag,
           us[0], flops / us[0] * 1e-6, us[1], flops / us[1] * 1e-6, 100.0 * (us[0] / us[1] - 1.0), maxd, maxv, bad, G4::LDS_BYTES);
    CK(hipFree(y2)); CK(hipFree(y4));
#else
    (void)tag; (void)crops; (void)iters; (void)hw;
#endif
}

template <int TW, int TH, int MINW>
static void bench_stem_shape(int crops, int iters) {
    using G = IrT2Geom<27, 16, 16, 3, 1, TW, TH, true>;
    const int hw = 128;
    const double flops = 2.0 * (27.0 * 16 + 16.0 * 9 + 16.0 * 16) * hw * hw * crops;
    IrT2Args t{};
    Ir2Args& a = t.b;
    a.ldx = 0; a.ldr = 16; a.ldy = 16;
    a.X = dev_rand((size_t)crops * 3 * 256 * 256, 2.f);
    a.Wpk = dev_rand((size_t)G::NCHUNK * (G::AP + G::BP), 0.01f);
    a.bp = dev_rand(64, 0.2f);
    float* y;
    CK(hipMalloc(&y, (size_t)crops * hw * hw * 16 * sizeof(float)));
    a.Y = y; a.relu_dw = 1; a.relu_out = 0;
    t.H = hw; t.W = hw; t.tiles_x = hw / TW; t.tiles_y = hw / TH;
    auto k = ir_tile_v2_kernel<27, 16, 16, 3, 1, TW, TH, true, MINW, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)crops * t.tiles_x * t.tiles_y);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(512), G::LDS_BYTES, 0, t);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(512), G::LDS_BYTES, 0, t);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    printf("stem_irt_3x16x16_hw256 %dx%d w%d fp32-stem  %8.1f us  %6.1f TF/s  (LDS %d B)\n", TW, TH, MINW, us, flops / us * 1e-6, G::LDS_BYTES);
    CK(hipFree(y));
}


// -DFEAR_STEM_CHECK: the stem tile against a CPU restatement of stem conv (3x3 s2, 3->16, ReLU) + e1 block (dw3x3 + ReLU, pw 16->16,
// + residual) on 2 crops — weights packed as fear_engine.hip's pack_fused16_host packs them (stem k order: step j = 4j + lane group).
static void check_stem() {
    using G = IrT2Geom<27, 16, 16, 3, 1, 32, 16, true>;
    const int crops = 2, hw = 128, IH = 256;
    std::vector<float> img((size_t)crops * 3 * IH * IH), We(16 * 27), be(16), Wd(16 * 9), bd(16), Wp(16 * 16), bp(16);
    auto rnd = [](float sc) { return sc * ((float)rand() / (float)RAND_MAX - 0.5f); };
    for (auto& v : img) v = rnd(2.f);
    for (auto& v : We) v = rnd(0.5f);
    for (auto& v : be) v = rnd(0.2f);
    for (auto& v : Wd) v = rnd(0.5f);
    for (auto& v : bd) v = rnd(0.2f);
    for (auto& v : Wp) v = rnd(0.5f);
    for (auto& v : bp) v = rnd(0.2f);
    std::vector<float> pk;
    for (int kg = 0; kg < 2; ++kg)
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int n = l & 15, k = 4 * (kg * 4 + i) + (l >> 4);
                pk.push_back(k < 27 ? We[n * 27 + k] : 0.f);
            }
    for (int c = 0; c < 16; ++c) pk.push_back(be[c]);
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) pk.push_back(Wp[(l & 15) * 16 + (l >> 4) * 4 + i]);
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < 16; ++c) pk.push_back(Wd[c * 9 + t]);
    for (int c = 0; c < 16; ++c) pk.push_back(bd[c]);
    if ((int)pk.size() != G::AP + G::BP) { printf("pack size %zu != %d\n", pk.size(), G::AP + G::BP); return; }
    // CPU
    std::vector<float> s0((size_t)crops * hw * hw * 16), ref((size_t)crops * hw * hw * 16);
    for (int b = 0; b < crops; ++b)
        for (int y = 0; y < hw; ++y)
            for (int x = 0; x < hw; ++x)
                for (int n = 0; n < 16; ++n) {
                    double acc = be[n];
                    for (int ci = 0; ci < 3; ++ci)
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
                                if (iy < 0 || iy >= IH || ix < 0 || ix >= IH) continue;
                                acc += (double)We[n * 27 + (ci * 3 + ky) * 3 + kx] * img[((size_t)(b * 3 + ci) * IH + iy) * IH + ix];
                            }
                    s0[((size_t)(b * hw + y) * hw + x) * 16 + n] = acc > 0 ? (float)acc : 0.f;
                }
    for (int b = 0; b < crops; ++b)
        for (int y = 0; y < hw; ++y)
            for (int x = 0; x < hw; ++x) {
                float d[16];
                for (int c = 0; c < 16; ++c) {
                    double acc = bd[c];
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int yy = y - 1 + ky, xx = x - 1 + kx;
                            if (yy < 0 || yy >= hw || xx < 0 || xx >= hw) continue;
                            acc += (double)Wd[c * 9 + ky * 3 + kx] * s0[((size_t)(b * hw + yy) * hw + xx) * 16 + c];
                        }
                    d[c] = acc > 0 ? (float)acc : 0.f;
                }
                for (int n = 0; n < 16; ++n) {
                    double acc = bp[n];
                    for (int c = 0; c < 16; ++c) acc += (double)Wp[n * 16 + c] * d[c];
                    ref[((size_t)(b * hw + y) * hw + x) * 16 + n] = (float)acc + s0[((size_t)(b * hw + y) * hw + x) * 16 + n];
                }
            }
    float *dX, *dW, *dB, *dY;
    CK(hipMalloc(&dX, img.size() * 4)); CK(hipMemcpy(dX, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dW, pk.size() * 4)); CK(hipMemcpy(dW, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dB, 64 * 4)); CK(hipMemset(dB, 0, 64 * 4)); CK(hipMemcpy(dB, bp.data(), 16 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dY, ref.size() * 4)); CK(hipMemset(dY, 0xff, ref.size() * 4));
    IrT2Args t{};
    Ir2Args& a = t.b;
    a.ldx = 0; a.ldr = 16; a.ldy = 16; a.X = dX; a.Wpk = dW; a.bp = dB; a.Y = dY; a.relu_dw = 1; a.relu_out = 0;
    t.H = hw; t.W = hw; t.tiles_x = hw / 32; t.tiles_y = hw / 16;
    auto k = ir_tile_v2_kernel<27, 16, 16, 3, 1, 32, 16, true, 4, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(crops * t.tiles_x * t.tiles_y), dim3(512), G::LDS_BYTES, 0, t);
    CK(hipDeviceSynchronize());
    std::vector<float> out(ref.size());
    CK(hipMemcpy(out.data(), dY, out.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0; size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < out.size(); ++i) {
        const double dd = fabs((double)out[i] - ref[i]);
        if (!(dd <= 1e-3)) { if (first == (size_t)-1) first = i; ++bad; }
        if (dd > maxd) maxd = dd;
    }
    printf("stem check: max |d| %.3g, %zu of %zu beyond 1e-3", maxd, bad, out.size());
    if (bad) {
        const size_t px = first / 16;
        printf("; first: crop %zu y %zu x %zu ch %zu got %g want %g", px / (hw * hw), px / hw % hw, px % hw, first % 16, out[first], ref[first]);
        // histogram of bad pixels by (y % 16, x % 32)
        int hy[16] = {0}, hx[32] = {0};
        for (size_t i = 0; i < out.size(); i += 16) {
            bool b = false;
            for (int c = 0; c < 16; ++c) b |= !(fabs((double)out[i + c] - ref[i + c]) <= 1e-3);
            if (b) { ++hy[i / 16 / hw % 16]; ++hx[i / 16 % 32]; }
        }
        printf("\n  bad pixels by y%%16:"); for (int i = 0; i < 16; ++i) printf(" %d", hy[i]);
        printf("\n  bad pixels by x%%32:"); for (int i = 0; i < 32; ++i) printf(" %d", hx[i]);
    }
    printf("\n");
}


static void bench_e1pair(int crops, int iters) {
    using G = E1PairGeom;
    const int hw = 64;
    E1PairArgs a{};
    a.X = dev_rand((size_t)crops * hw * hw * 24, 2.f);
    a.Wpk = dev_rand(2 * G::WBLK, 0.2f);
    const size_t ny = (size_t)crops * hw * hw * 24;
    float* y;
    CK(hipMalloc(&y, ny * sizeof(float)));
    a.Y = y; a.H = hw; a.W = hw; a.tiles_x = hw / 16; a.tiles_y = hw / 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(e1pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref(ny), out(ny);
#ifdef FEAR_E1PAIR_REF
    std::vector<float> r5(ny);
    {
        E1PairR5Args b{};
        b.X = a.X; b.Wpk = a.Wpk; b.Y = y; b.H = hw; b.W = hw; b.tiles_x = a.tiles_x; b.tiles_y = a.tiles_y;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(e1pair_r5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, E1PairR5Geom::LDS_BYTES));
        const dim3 grid((unsigned)crops * 16);
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(e1pair_r5_kernel, grid, dim3(512), E1PairR5Geom::LDS_BYTES, 0, b);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(e1pair_r5_kernel, grid, dim3(512), E1PairR5Geom::LDS_BYTES, 0, b);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("e1pair round-5 kernel                                         %8.1f us\n", 1e3 * ms / iters);
        }
        CK(hipMemcpy(r5.data(), y, ny * sizeof(float), hipMemcpyDeviceToHost));
    }
#endif
    // tiles per workgroup: 1 = one launch slot per tile (round 4's form), 8 = 512 workgroups at 256 crops
    for (int rep = 0; rep < 2; ++rep)
        for (int tpw : {1, 2, 4, 8, 16}) {
            if ((crops * 16) % tpw) continue;
            a.tpw = tpw;
            const dim3 grid((unsigned)crops * 16 / tpw);
            CK(hipMemset(y, 0xff, ny * sizeof(float)));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(e1pair_kernel, grid, dim3(512), G::LDS_BYTES, 0, a);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(e1pair_kernel, grid, dim3(512), G::LDS_BYTES, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(out.data(), y, ny * sizeof(float), hipMemcpyDeviceToHost));
            size_t bad = 0;
            if (tpw == 1) {
                ref = out;
#ifdef FEAR_E1PAIR_REF
                size_t b5 = 0;
                for (size_t i = 0; i < ny; ++i) b5 += memcmp(&r5[i], &out[i], 4) != 0;
                if (rep == 0) printf("   against the round-5 kernel: %s (%zu of %zu differ)\n", b5 ? "MISMATCH" : "bit-identical", b5, ny);
#endif
            }
            else for (size_t i = 0; i < ny; ++i) bad += memcmp(&ref[i], &out[i], 4) != 0;
            printf("e1pair 24ch hw64 (E1P_ABL=%d, E1P_SKEW=%d, E1P_NA=%d) tiles/workgroup %2d  %8.1f us  (LDS %d B)  %s\n", E1P_ABL, E1P_SKEW, E1P_NA, tpw, 1e3 * ms / iters,
                   G::LDS_BYTES, tpw == 1 ? "" : bad ? "MISMATCH vs 1 tile/workgroup" : "bit-identical to 1 tile/workgroup");
        }
}

static void bench_stem(int crops, int iters) {
#ifdef FEAR_STEM_CHECK
    check_stem();
#endif
    bench_stem_shape<32, 16, 4>(crops, iters);
#ifdef FEAR_STEM_SHAPES
    bench_stem_shape<32, 16, 6>(crops, iters);
    bench_stem_shape<32, 32, 2>(crops, iters);
    bench_stem_shape<32, 32, 4>(crops, iters);
    bench_stem_shape<32, 16, 4>(crops, iters);
#endif
}

template <int CIN, int COUT, int KS>
static void bench_sep(const char* tag, int crops, int iters) {
    const double flops = 2.0 * 256 * ((double)CIN * KS * KS + (double)CIN * COUT) * crops;
    using G = Sep16Geom<CIN, COUT, KS>;
    Ir2Args a = make_args<CIN, COUT>(crops, (size_t)G::NCHUNK * G::CST);
    const double us = time_kernel(sep16_kernel<CIN, COUT, KS>, G::LDS_BYTES, crops, iters, a);
    printf("%-24s fp32-mfma  %8.1f us  %6.1f TF/s\n", tag, us, flops / us * 1e-6);
}

template <int CIN, int CEXP, int COUT, int KS, bool EXPAND>
static void bench(const char* tag, int crops, int iters) {
    const double flops = 2.0 * 256 * ((EXPAND ? (double)CIN * CEXP : 0.0) + (double)CEXP * KS * KS + (double)CEXP * COUT) * crops;
    {
        using G = Ir2Geom<CIN, CEXP, COUT, KS, EXPAND>;
        Ir2Args a = make_args<CIN, COUT>(crops, (size_t)G::NCHUNK * (G::AP + G::BP));
        const double us = time_kernel(ir16v2_fused_kernel<CIN, CEXP, COUT, KS, EXPAND>, G::LDS_BYTES, crops, iters, a);
        printf("%-24s fp32-mfma  %8.1f us  %6.1f TF/s\n", tag, us, flops / us * 1e-6);
    }
    {
        using G = IrHGeom<CIN, CEXP, COUT, KS, EXPAND>;
        Ir2Args a = make_args<CIN, COUT>(crops, (size_t)G::NCHUNK * (G::AP + G::BP));
        const double us = time_kernel(ir16h_fused_kernel<CIN, CEXP, COUT, KS, EXPAND>, G::LDS_BYTES, crops, iters, a);
        printf("%-24s f16-split  %8.1f us  %6.1f TF/s\n", tag, us, flops / us * 1e-6);
    }
}

int main(int argc, char** argv) {
    const int crops = argc > 1 ? atoi(argv[1]) : 256;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const bool all = argc > 3;      // any third argument: also the tile-shape sweep and the 16x16 kernels
    printf("FEAR_ABL=%d\n", FEAR_ABL);
    // the product's tile table (fear_engine.hip kFusedTile), in plan order
#ifdef FEAR_E1PAIR_ONLY
    bench_e1pair(crops, iters);
    return 0;
#endif
#ifdef FEAR_C32_ONLY
    {   // the 32 x 32 stage as one chain kernel (random weights: timing only; 25.3 GFLOP per 256 crops)
        Chain32Args a{};
        a.ldx = 32; a.ldy = 64;
        a.X = dev_rand((size_t)crops * 1024 * 32, 2.f);
        float* y;
        CK(hipMalloc(&y, (size_t)crops * 256 * 64 * sizeof(float)));
        a.Y = y;
        a.Wpk[0] = dev_rand((size_t)Ir2Geom<32, 96, 32, 5, true>::NCHUNK * (Ir2Geom<32, 96, 32, 5, true>::AP + Ir2Geom<32, 96, 32, 5, true>::BP), 0.01f);
        a.Wpk[1] = dev_rand((size_t)Ir2Geom<32, 192, 32, 5, true>::NCHUNK * (Ir2Geom<32, 192, 32, 5, true>::AP + Ir2Geom<32, 192, 32, 5, true>::BP), 0.01f);
        a.Wpk[2] = dev_rand((size_t)Ir2Geom<32, 192, 32, 3, true>::NCHUNK * (Ir2Geom<32, 192, 32, 3, true>::AP + Ir2Geom<32, 192, 32, 3, true>::BP), 0.01f);
        a.Wpk[3] = dev_rand((size_t)Ir2Geom<32, 192, 64, 5, true>::NCHUNK * (Ir2Geom<32, 192, 64, 5, true>::AP + Ir2Geom<32, 192, 64, 5, true>::BP), 0.01f);
        for (int j = 0; j < 4; ++j) a.bp[j] = dev_rand(64, 0.2f);
        auto k = chain32_kernel<C32Blk<32, 96, 32, 5, 1, true>, C32Blk<32, 192, 32, 5, 1, true>, C32Blk<32, 192, 32, 3, 1, true>, C32Blk<32, 192, 64, 5, 2, false>>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, C32Geom::LDS_BYTES));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(crops), dim3(512), C32Geom::LDS_BYTES, 0, a);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(crops), dim3(512), C32Geom::LDS_BYTES, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / iters, gf = 25.3 * crops / 256.0;
            printf("chain32 (C32_D=%d D2=%d GS2=%d ABL=%d)  %8.1f us per %d crops  %6.1f TF/s\n", C32_D, C32_D2, C32_GS2, C32_ABL, us, crops, gf / us * 1e-3);
        }
    }
    return 0;
#endif
#ifdef FEAR_IR16H_ONLY
    // the bf16 matrix-pipe form of the two heaviest stride-16 blocks (FEAR-M's stage; 512 crops = configs[3])
    for (int rep = 0; rep < 2; ++rep) {
        {
            using G = IrHGeom<112, 672, 112, 5, true>;
            Ir2Args a = make_args<112, 112>(crops, (size_t)G::NCHUNK * (G::AP + G::BP));
            const double us = time_kernel(ir16h_fused_kernel<112, 672, 112, 5, true, 2>, G::LDS_BYTES, crops, iters, a);
            printf("ir16h_112x672x112_k5 bf16 (FEAR_ABL=%d, IR16H_D=%d)  %8.1f us per %d crops\n", FEAR_ABL, IR16H_D, us, crops);
        }
        {
            using G = IrHGeom<64, 384, 64, 5, true>;
            Ir2Args a = make_args<64, 64>(crops, (size_t)G::NCHUNK * (G::AP + G::BP));
            const double us = time_kernel(ir16h_fused_kernel<64, 384, 64, 5, true, 2>, G::LDS_BYTES, crops, iters, a);
            printf("ir16h_64x384x64_k5   bf16 (FEAR_ABL=%d, IR16H_D=%d)  %8.1f us per %d crops\n", FEAR_ABL, IR16H_D, us, crops);
        }
    }
    return 0;
#endif
#ifdef FEAR_IR16_ONLY
    printf("IR16_GS=%d IR16_D=%d\n", IR16_GS, IR16_D);
    for (int rep = 0; rep < 2; ++rep) {
        bench<112, 672, 112, 5, true>("ir16_112x672x112_k5", crops, iters);
        bench<64, 384, 64, 5, true>("ir16_64x384x64_k5", crops, iters);
    }
    return 0;
#endif
    bench_stem(crops, iters);
#ifdef FEAR_STEM_ONLY
    return 0;
#endif
    bench_tile<16, 96, 24, 3, 2, 16, 8, true, 4>("s2  irt_16x96x24_k3s2_hw128", crops, iters, 128);
    bench_tile<24, 32, 24, 3, 1, 16, 16, false, 4>("s45 irt_24x24x24_k3 e1 16x16", crops, iters, 64);
#ifdef FEAR_E1_SHAPES
    bench_tile<24, 32, 24, 3, 1, 16, 16, false, 8>("s45 e1 16x16 w8", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 16, 16, false, 6>("s45 e1 16x16 w6", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 32, 16, false, 4>("s45 e1 32x16 w4", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 16, 8, false, 8>("s45 e1 16x8 w8", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 16, 8, false, 4>("s45 e1 16x8 w4", crops, iters, 64);
#endif
    bench_tile<24, 144, 32, 5, 2, 16, 16, true, 2>("s6  irt_24x144x32_k5s2 16x16", crops, iters, 64);
    bench_tile<32, 96, 32, 5, 1, 16, 16, true, 2>("s7  irt_32x96x32_k5 16x16", crops, iters, 32);
    bench_tile<32, 192, 32, 5, 1, 16, 32, true, 2>("s8  irt_32x192x32_k5 16x32", crops, iters, 32);
    bench_tile<32, 192, 32, 3, 1, 32, 32, true, 2>("s9  irt_32x192x32_k3 32x32", crops, iters, 32);
    bench_tile<32, 192, 64, 5, 2, 16, 16, true, 2>("s10 irt_32x192x64_k5s2 16x16", crops, iters, 32);
#ifdef FEAR_HAVE_V4
    bench_tile_v4<24, 144, 32, 5, 2, 16, 16, 2, 2>("s6  v4 24x144x32_k5s2 16x16", crops, iters, 64);
    bench_tile_v4<32, 192, 32, 3, 1, 32, 32, 2, 2>("s9  v4 32x192x32_k3 32x32", crops, iters, 32);
    bench_tile_v4<32, 192, 64, 5, 2, 16, 16, 2, 2>("s10 v4 32x192x64_k5s2 16x16", crops, iters, 32);
    bench_tile_v4<32, 192, 32, 5, 1, 16, 32, 2, 2>("s8  v4 32x192x32_k5 16x32", crops, iters, 32);
    bench_tile_v4<32, 96, 32, 5, 1, 16, 16, 2, 2>("s7  v4 32x96x32_k5 16x16", crops, iters, 32);
    bench_tile_v4<16, 96, 24, 3, 2, 16, 8, 4, 2>("s2  v4 16x96x24_k3s2 16x8", crops, iters, 128);
#endif
#ifdef FEAR_SEP_ONLY
    bench_sep<256, 256, 3>("sep16_256x256_k3", crops, iters);
    bench_sep<320, 256, 3>("sep16_320x256_k3", crops, iters);
    return 0;
#endif
    if (!all) return 0;
    bench<112, 672, 112, 5, true>("ir16_112x672x112_k5", crops, iters);
    bench<64, 384, 64, 5, true>("ir16_64x384x64_k5", crops, iters);
    bench_tile<24, 144, 32, 5, 2, 16, 8, true, 4>("irt_24x144x32_k5s2 16x8", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 32, 16, false, 4>("irt_24x24x24_k3 e1 32x16", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 16, 32, false, 4>("irt_24x24x24_k3 e1 16x32", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 32, 32, false, 4>("irt_24x24x24_k3 e1 32x32", crops, iters, 64);
    bench_tile<24, 32, 24, 3, 1, 16, 8, false, 4>("irt_24x24x24_k3 e1 16x8", crops, iters, 64);
    bench_tile<32, 192, 32, 5, 1, 16, 16, true, 2>("irt_32x192x32_k5 16x16", crops, iters, 32);
    bench_tile<32, 192, 32, 5, 1, 32, 16, true, 2>("irt_32x192x32_k5 32x16", crops, iters, 32);
    bench_tile<32, 192, 32, 5, 1, 32, 32, true, 2>("irt_32x192x32_k5 32x32", crops, iters, 32);
    bench_tile<32, 192, 32, 3, 1, 16, 32, true, 2>("irt_32x192x32_k3 16x32", crops, iters, 32);
    bench_tile<32, 192, 32, 3, 1, 16, 16, true, 2>("irt_32x192x32_k3 16x16", crops, iters, 32);
    bench_tile<32, 192, 32, 3, 1, 32, 16, true, 2>("irt_32x192x32_k3 32x16", crops, iters, 32);
    bench_tile<32, 192, 64, 5, 2, 16, 8, true, 2>("irt_32x192x64_k5s2 16x8", crops, iters, 32);
    bench_tile<24, 144, 32, 5, 2, 32, 8, true, 2>("irt_24x144x32_k5s2 32x8", crops, iters, 64);
    bench_tile<16, 96, 24, 3, 2, 16, 16, true, 2>("irt_16x96x24_k3s2 16x16", crops, iters, 128);
    bench_tile<16, 96, 24, 3, 2, 32, 8, true, 2>("irt_16x96x24_k3s2 32x8", crops, iters, 128);
    bench_tile<16, 96, 24, 3, 2, 32, 8, true, 4>("irt_16x96x24_k3s2 32x8 w4", crops, iters, 128);
    bench_tile<32, 96, 32, 5, 1, 16, 32, true, 2>("irt_32x96x32_k5 16x32", crops, iters, 32);
    bench_tile<32, 96, 32, 5, 1, 32, 32, true, 2>("irt_32x96x32_k5 32x32", crops, iters, 32);
    bench_sep<256, 256, 3>("sep16_256x256_k3", crops, iters);
    bench_sep<320, 256, 3>("sep16_320x256_k3", crops, iters);
    return 0;
}
