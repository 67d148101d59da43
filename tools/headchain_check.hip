// headchain_check — development tool: headchain_kernel (one launch for the whole BoxTower) against the eight sep16 launches it
// replaces, on random weights: outputs compared bit for bit, both timed.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -o tools/headchain_check tools/headchain_check.hip
// Run on the GPU box: tools/headchain_check [crops=256] [iters=20]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../feartracker_amd/csrc/fear_kernels.h"
#include "../feartracker_amd/csrc/fear_headchain.h"

using namespace fear;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale, float offset = 0.f) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = offset + scale * ((float)rand() / RAND_MAX - 0.5f);
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}
static float* dev_alloc(size_t n, int fill = 0) {
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, fill, n * sizeof(float)));
    return d;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    using G = HeadChainGeom<3>;
    constexpr int C = 256, CC = 320;
    const int cin[4] = {C, CC, C, C};
    float* X = dev_rand((size_t)n * 256 * C, 2.f);
    HeadChainArgs ha{};
    ha.X = X; ha.ldx = C; ha.n_crops = n; ha.relu_dw = 0; ha.relu_out = 1;
    float* refY[2];
    float* cat[2];
    float* x1[2];
    float* x2[2];
    struct RefBranch { const float* Wpk[4]; const float* bp[4]; } rb[2];
    auto host_rand = [](size_t n, float scale) {
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / (float)RAND_MAX - 0.5f);
        return h;
    };
    auto to_dev = [](const std::vector<float>& h) {
        float* d;
        CK(hipMalloc(&d, h.size() * sizeof(float)));
        CK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        return d;
    };
    constexpr int CST = 16 * 256 + G::WDF;      // Sep16Geom's packed chunk
    for (int br = 0; br < 2; ++br) {
        HeadChainBranch& b = ha.br[br];
        std::vector<float> sep[4], bias[4];
        for (int l = 0; l < 4; ++l) {
            sep[l] = host_rand((size_t)(cin[l] / 16) * CST, 0.12f);
            bias[l] = host_rand(C, 0.2f);
            rb[br].Wpk[l] = to_dev(sep[l]);
            rb[br].bp[l] = to_dev(bias[l]);
        }
        for (int l = 0; l < 4; ++l) b.W[l] = to_dev(headchain_pack(sep[l].data(), cin[l], bias[l].data(), l < 3 ? sep[l + 1].data() : nullptr, 3));
        b.Wd0 = to_dev(headchain_pack_dw(sep[0].data(), 0, 16, 3));
        b.WdC = to_dev(headchain_pack_dw(sep[1].data(), 16, 4, 3));
        b.Z = dev_rand((size_t)n * C * 64, 0.1f);
        b.z_stride = C * 64;
        b.P_Wpk = dev_rand((size_t)16 * G::PCH, 0.2f);
        b.P_bp = dev_rand(4, 0.2f);
        b.pred_cout = br == 0 ? 1 : 4;
        b.pred_act = br == 0 ? 0 : 2;
        b.pred_stride = b.pred_cout * 256;
        b.P_Y = dev_alloc((size_t)n * b.pred_cout * 256, 0xff);
        b.D = dev_alloc((size_t)n * G::D_FLOATS, 0xff);
        refY[br] = dev_alloc((size_t)n * b.pred_cout * 256, 0xff);
        cat[br] = dev_alloc((size_t)n * 256 * CC, 0xff);
        x1[br] = dev_alloc((size_t)n * 256 * C, 0xff);
        x2[br] = dev_alloc((size_t)n * 256 * C, 0xff);
    }
    long long* dbg;
    CK(hipMalloc(&dbg, 256 * sizeof(long long)));
    CK(hipMemset(dbg, 0, 256 * sizeof(long long)));
    ha.dbg = dbg;
    auto kCorr = sep16_kernel<256, 256, 3, true>;
    auto k320 = sep16_kernel<320, 256, 3>;
    auto k256 = sep16_kernel<256, 256, 3>;
    auto kPred = sep16_kernel<256, 256, 3, false, true>;
    auto kChain = headchain_kernel<3>;
    const int ldsCorr = Sep16Geom<256, 256, 3, true>::LDS_BYTES, lds320 = Sep16Geom<320, 256, 3>::LDS_BYTES, lds256 = Sep16Geom<256, 256, 3>::LDS_BYTES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kCorr), hipFuncAttributeMaxDynamicSharedMemorySize, ldsCorr));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k320), hipFuncAttributeMaxDynamicSharedMemorySize, lds320));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k256), hipFuncAttributeMaxDynamicSharedMemorySize, lds256));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kPred), hipFuncAttributeMaxDynamicSharedMemorySize, lds256));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kChain), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    printf("headchain LDS %d B, D scratch %.1f MB\n", G::LDS_BYTES, 2.0 * n * G::D_FLOATS * 4e-6);

    auto run_ref = [&] {
        for (int br = 0; br < 2; ++br) {
            const HeadChainBranch& b = ha.br[br];
            const RefBranch& r = rb[br];
            Ir2Args a{};
            a.relu_dw = 0; a.relu_out = 1;
            a.X = X; a.ldx = C; a.Wpk = r.Wpk[0]; a.bp = r.bp[0]; a.Y = cat[br]; a.ldy = CC; a.Z = b.Z; a.z_stride = b.z_stride;
            hipLaunchKernelGGL(kCorr, dim3(n), dim3(512), ldsCorr, 0, a);
            a = Ir2Args{};
            a.relu_dw = 0; a.relu_out = 1;
            a.X = cat[br]; a.ldx = CC; a.Wpk = r.Wpk[1]; a.bp = r.bp[1]; a.Y = x1[br]; a.ldy = C;
            hipLaunchKernelGGL(k320, dim3(n), dim3(512), lds320, 0, a);
            a.X = x1[br]; a.ldx = C; a.Wpk = r.Wpk[2]; a.bp = r.bp[2]; a.Y = x2[br]; a.ldy = C;
            hipLaunchKernelGGL(k256, dim3(n), dim3(512), lds256, 0, a);
            a.X = x2[br]; a.ldx = C; a.Wpk = r.Wpk[3]; a.bp = r.bp[3]; a.Y = nullptr;
            a.pred_cout = b.pred_cout; a.pred_act = b.pred_act; a.P_Wpk = b.P_Wpk; a.P_bp = b.P_bp; a.P_Y = refY[br]; a.pred_stride = b.pred_stride;
            hipLaunchKernelGGL(kPred, dim3(n), dim3(512), lds256, 0, a);
        }
    };
    auto run_chain = [&] { hipLaunchKernelGGL(kChain, dim3(16 * ((n + 7) / 8)), dim3(512), G::LDS_BYTES, 0, ha); };

    run_ref();
    CK(hipDeviceSynchronize());
    run_chain();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    int bad_total = 0;
    // n crops of `count` floats each, at strides sref / snew
    auto compare = [&](const char* what, const float* dref, size_t sref, const float* dnew, size_t snew, size_t count) {
        std::vector<float> r(count), v(count);
        size_t bad = 0, first = 0;
        double maxd = 0, maxv = 0;
        for (int k = 0; k < n; ++k) {
            CK(hipMemcpy(r.data(), dref + k * sref, count * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(v.data(), dnew + k * snew, count * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < count; ++i) {
                if (memcmp(&r[i], &v[i], 4) != 0) { if (!bad) first = k * count + i; ++bad; }
                const double d = fabs((double)r[i] - (double)v[i]);
                if (d == d && d > maxd) maxd = d;
                if (fabs(r[i]) > maxv) maxv = fabs(r[i]);
            }
        }
        printf("  %-28s %zu values, %zu differ (first at %zu), max|diff| %.3g, max|ref| %.3g\n", what, count * n, bad, first, maxd, maxv);
        bad_total += bad != 0;
    };
    for (int br = 0; br < 2; ++br) {
        printf("branch %d\n", br);
        compare("prediction map", refY[br], ha.br[br].pred_cout * 256, ha.br[br].P_Y, ha.br[br].pred_cout * 256, ha.br[br].pred_cout * 256);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) run_ref();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_ref = 1e3 * ms / iters;
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) run_chain();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_new = 1e3 * ms / iters;
        const double flops = 2.0 * n * 2 * 256 * (3.0 * (256.0 * 9 + 256.0 * 256) + (320.0 * 9 + 320.0 * 256) + 256.0 * 64 + 256.0 * 9 + 256.0 * 2.5);
        printf("%d crops: 8 x sep16 %.1f us (%.1f TF/s) | headchain %.1f us (%.1f TF/s)  %+.1f%%\n", n, us_ref, flops / us_ref * 1e-6, us_new,
               flops / us_new * 1e-6, 100.0 * (us_ref / us_new - 1.0));
    }
    if (HC_ABL & 8) {
        long long h[80];
        CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; ++w) {
            printf("wave %d of workgroup (100, 0), us since kernel start [prologue | layer 0 | 1 | 2 | 3 done]:", w * 4);
            for (int i = 1; i < 40 && h[w * 40 + i]; ++i) printf(" %.2f", (h[w * 40 + i] - h[w * 40]) * 0.01);
            printf("\n");
        }
    }
    if (HC_ABL & 128) {
        long long h[256];
        CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        printf("layer 2, passes 2..4, s_memtime ticks since wave 0's first stamp:\n  [pass start | first fragments read | GEMM done | epilogue + tile written | barrier A passed | depthwise + scratch stores done | barrier B passed]\n");
        for (int w = 0; w < 8; ++w) {
            printf("  wave %d:", w);
            for (int i = 0; i < 21; ++i) printf("%s %lld", i % 7 == 0 ? " |" : "", h[80 + w * 21 + i] - h[80]);
            printf("\n");
        }
    }
    printf(HC_ABL & 7 ? "(ablation build: outputs not expected to match)\n" : bad_total ? "MISMATCH\n" : "bit-identical\n");
    return bad_total && !(HC_ABL & 7) ? 1 : 0;
}
