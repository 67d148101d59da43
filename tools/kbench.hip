// kbench — standalone ablation micro-bench for the fused block kernels (development tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kbench tools/kbench.hip
// Run on the GPU box: tools/kbench [crops=256] [iters=20]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../feartracker_amd/csrc/fear_kernels.h"

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

template <int CIN, int CEXP, int COUT, int KS, int CE, bool EXPAND, int ABL>
static double run(const char* tag, int crops, int iters, IrArgs a) {
    auto k = ir16_fused_kernel<CIN, CEXP, COUT, KS, CE, EXPAND, ABL>;
    const int lds = ir16_lds_bytes<CIN, CEXP, COUT, KS, CE, EXPAND>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(crops), dim3(512), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(crops), dim3(512), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    const double flops = 2.0 * 256 * ((EXPAND ? (double)CIN * CEXP : 0.0) + (double)CEXP * KS * KS + (double)CEXP * COUT) * crops;
    printf("%-34s abl=%2d  %8.1f us  %6.1f TF/s\n", tag, ABL, us, flops / us * 1e-6);
    return us;
}

template <int CIN, int CEXP, int COUT, int KS, int CE, bool EXPAND>
static void suite(const char* tag, int crops, int iters) {
    IrArgs a{};
    a.ldx = CIN; a.ldr = COUT; a.ldy = COUT;
    a.X = dev_rand((size_t)crops * 256 * CIN, 2.f);
    a.We = EXPAND ? dev_rand((size_t)CEXP * CIN, 0.2f) : nullptr;
    a.be = EXPAND ? dev_rand(CEXP, 0.2f) : nullptr;
    a.Wd = dev_rand((size_t)KS * KS * CEXP, 0.4f);
    a.bd = dev_rand(CEXP, 0.2f);
    a.Wp = dev_rand((size_t)COUT * CEXP, 0.2f);
    a.bp = dev_rand(COUT, 0.2f);
    a.R = nullptr;
    float* y;
    CK(hipMalloc(&y, (size_t)crops * 256 * COUT * sizeof(float)));
    a.Y = y;
    a.relu_dw = 1; a.relu_out = 0;
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 0>(tag, crops, iters, a);
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 1>(tag, crops, iters, a);
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 2>(tag, crops, iters, a);
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 4>(tag, crops, iters, a);
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 5>(tag, crops, iters, a);
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 7>(tag, crops, iters, a);
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 8>(tag, crops, iters, a);
    run<CIN, CEXP, COUT, KS, CE, EXPAND, 10>(tag, crops, iters, a);
}

int main(int argc, char** argv) {
    const int crops = argc > 1 ? atoi(argv[1]) : 256;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    suite<112, 672, 112, 5, 32, true>("ir16_112x672x112_k5", crops, iters);
    suite<64, 384, 64, 5, 32, true>("ir16_64x384x64_k5", crops, iters);
    suite<256, 256, 256, 3, 32, false>("sep16_256x256x256_k3", crops, iters);
    return 0;
}
