// sepcheck — correctness probe for sep16_kernel variants against a plain host loop (development tool).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -o tools/_kb/sepcheck tools/sepcheck.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#ifndef FEAR_DBG_DUMP
#define FEAR_DBG_DUMP 99
#endif
#include "../feartracker_amd/csrc/fear_kernels.h"
using namespace fear;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static std::vector<float> rnd(size_t n, float s) { std::vector<float> v(n); for (auto& x : v) x = s * ((float)rand() / (float)RAND_MAX - 0.5f); return v; }
template <class T> static T* up(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

template <int CIN, int COUT>
static void check(int crops) {
    using G = Sep16Geom<CIN, COUT, 3>;
    constexpr int NCH = CIN / 16, NTP = COUT / 16;
    auto X = rnd((size_t)crops * 256 * CIN, 2.f), W = rnd((size_t)NCH * G::CST, 0.2f), B = rnd(COUT, 0.5f);
    Ir2Args a{};
    a.X = up(X); a.ldx = CIN; a.Wpk = up(W); a.bp = up(B); a.ldy = COUT; a.relu_dw = 0; a.relu_out = 1;
    float* y; CK(hipMalloc(&y, (size_t)crops * 256 * COUT * 4)); a.Y = y;
    auto k = sep16_kernel<CIN, COUT, 3>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(crops), dim3(512), G::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> Y((size_t)crops * 256 * COUT);
    CK(hipMemcpy(Y.data(), y, Y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, ref_max = 0;
    for (int crop = 0; crop < crops; ++crop)
        for (int py = 0; py < 16; ++py) for (int px = 0; px < 16; ++px) {
            std::vector<double> d(CIN);
            for (int c = 0; c < CIN; ++c) {
                const int ch = c % 16, cc = c / 16;
                double s = W[(size_t)cc * G::CST + G::WPF + 9 * 16 + ch];
                for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                    const int yy = py + ky - 1, xx = px + kx - 1;
                    if (yy < 0 || yy > 15 || xx < 0 || xx > 15) continue;
                    s += (double)X[((size_t)crop * 256 + yy * 16 + xx) * CIN + c] * W[(size_t)cc * G::CST + G::WPF + (ky * 3 + kx) * 16 + ch];
                }
                d[c] = s;
            }
            for (int n = 0; n < COUT; ++n) {
                double s = B[n];
                for (int c = 0; c < CIN; ++c) {
                    const int cc = c / 16, kk = c % 16, nt = n / 16, l = (n % 16) + 16 * (kk / 4), i = kk % 4;
                    s += d[c] * W[(size_t)cc * G::CST + nt * 256 + l * 4 + i];
                }
                if (s < 0) s = 0;
                const double got = Y[((size_t)crop * 256 + py * 16 + px) * COUT + n];
                worst = fmax(worst, fabs(got - s)); ref_max = fmax(ref_max, fabs(s));
            }
        }
    printf("sep16<%d,%d> plain: max abs err %.3e (max |ref| %.3e)  LDS %d B\n", CIN, COUT, worst, ref_max, G::LDS_BYTES);
}
// layer output y = relu(sep(x)) on the host (double), [crop][256 px][COUT]
template <int CIN, int COUT>
static std::vector<double> host_layer(const std::vector<float>& X, const std::vector<float>& W, const std::vector<float>& B, int crops) {
    using G = Sep16Geom<CIN, COUT, 3>;
    std::vector<double> Y((size_t)crops * 256 * COUT);
    for (int crop = 0; crop < crops; ++crop)
        for (int py = 0; py < 16; ++py) for (int px = 0; px < 16; ++px) {
            std::vector<double> d(CIN);
            for (int c = 0; c < CIN; ++c) {
                const int ch = c % 16, cc = c / 16;
                double s = W[(size_t)cc * G::CST + G::WPF + 9 * 16 + ch];
                for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                    const int yy = py + ky - 1, xx = px + kx - 1;
                    if (yy < 0 || yy > 15 || xx < 0 || xx > 15) continue;
                    s += (double)X[((size_t)crop * 256 + yy * 16 + xx) * CIN + c] * W[(size_t)cc * G::CST + G::WPF + (ky * 3 + kx) * 16 + ch];
                }
                d[c] = s;
            }
            for (int n = 0; n < COUT; ++n) {
                double s = B[n];
                for (int c = 0; c < CIN; ++c) {
                    const int cc = c / 16, kk = c % 16, nt = n / 16, l = (n % 16) + 16 * (kk / 4), i = kk % 4;
                    s += d[c] * W[(size_t)cc * G::CST + nt * 256 + l * 4 + i];
                }
                Y[((size_t)crop * 256 + py * 16 + px) * COUT + n] = s < 0 ? 0 : s;
            }
        }
    return Y;
}

static void check_pred(int crops, int only_chunk = -1, int only_tap = -1) {
    constexpr int C = 256;
    using G = Sep16Geom<C, C, 3>;
    constexpr int PCH = 256 + 9 * 16 + 16;
    auto X = rnd((size_t)crops * 256 * C, 2.f), W = rnd((size_t)16 * G::CST, 0.2f), B = rnd(C, 0.5f);
    auto PW_ = rnd((size_t)16 * PCH, 0.3f), PB = rnd(4, 0.2f);
    if (only_chunk >= 0)
        for (int cc = 0; cc < 16; ++cc)
            for (int j = 0; j < PCH; ++j) {
                const bool is_tap = j >= 256 && j < 256 + 9 * 16;
                if (cc != only_chunk || (only_tap >= 0 && is_tap && (j - 256) / 16 != only_tap)) PW_[(size_t)cc * PCH + j] = 0.f;
            }
    Ir2Args a{};
    a.X = up(X); a.ldx = C; a.Wpk = up(W); a.bp = up(B); a.ldy = C; a.relu_dw = 0; a.relu_out = 1;
    a.P_Wpk = up(PW_); a.P_bp = up(PB); a.pred_cout = 4; a.pred_act = 2; a.pred_stride = 4 * 256;
    float* py_; CK(hipMalloc(&py_, (size_t)crops * 4 * 256 * 4)); a.P_Y = py_;
    auto k = sep16_kernel<C, C, 3, false, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(crops), dim3(512), G::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> O((size_t)crops * 4 * 256);
    CK(hipMemcpy(O.data(), py_, O.size() * 4, hipMemcpyDeviceToHost));
    auto Y = host_layer<C, C>(X, W, B, crops);
    double worst = 0, ref_max = 0;
    for (int crop = 0; crop < crops; ++crop)
        for (int py = 0; py < 16; ++py) for (int px = 0; px < 16; ++px)
            for (int n = 0; n < 4; ++n) {
                double s = PB[n];
                for (int c = 0; c < C; ++c) {
                    const int cc = c / 16, ch = c % 16;
                    double d = PW_[(size_t)cc * PCH + 256 + 9 * 16 + ch];
                    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                        const int yy = py + ky - 1, xx = px + kx - 1;
                        if (yy < 0 || yy > 15 || xx < 0 || xx > 15) continue;
                        d += Y[((size_t)crop * 256 + yy * 16 + xx) * C + c] * PW_[(size_t)cc * PCH + 256 + (ky * 3 + kx) * 16 + ch];
                    }
                    const int l = n + 16 * (ch / 4), i = ch % 4;
                    s += d * PW_[(size_t)cc * PCH + l * 4 + i];
                }
                const double ref = exp(s), got = O[((size_t)crop * 4 + n) * 256 + py * 16 + px];
                worst = fmax(worst, fabs(got - ref) / fabs(ref)); ref_max = fmax(ref_max, fabs(ref));
            }
    printf("sep16 PRED (chunk %d tap %d): max rel err %.3e (max |ref| %.3e)\n", only_chunk, only_tap, worst, ref_max);
}

static void check_corr(int crops) {
    constexpr int C = 256;
    using G = Sep16Geom<C, C, 3, true>;
    auto X = rnd((size_t)crops * 256 * C, 2.f), W = rnd((size_t)16 * G::CST, 0.2f), B = rnd(C, 0.5f), Z = rnd((size_t)crops * C * 64, 1.f);
    Ir2Args a{};
    a.X = up(X); a.ldx = C; a.Wpk = up(W); a.bp = up(B); a.ldy = 320; a.relu_dw = 0; a.relu_out = 1;
    a.Z = up(Z); a.z_stride = (long)C * 64;
    float* y; CK(hipMalloc(&y, (size_t)crops * 256 * 320 * 4)); a.Y = y;
    auto k = sep16_kernel<C, C, 3, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(crops), dim3(512), G::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> O((size_t)crops * 256 * 320);
    CK(hipMemcpy(O.data(), y, O.size() * 4, hipMemcpyDeviceToHost));
    auto Y = host_layer<C, C>(X, W, B, crops);
    double wf = 0, wc = 0, rc = 0;
    for (int crop = 0; crop < crops; ++crop)
        for (int m = 0; m < 256; ++m) {
            for (int n = 0; n < C; ++n) wf = fmax(wf, fabs(O[((size_t)crop * 256 + m) * 320 + n] - Y[((size_t)crop * 256 + m) * C + n]));
            for (int t = 0; t < 64; ++t) {
                double s = 0;
                for (int c = 0; c < C; ++c) s += Y[((size_t)crop * 256 + m) * C + c] * Z[((size_t)crop * C + c) * 64 + t];
                wc = fmax(wc, fabs(O[((size_t)crop * 256 + m) * 320 + C + t] - s)); rc = fmax(rc, fabs(s));
            }
        }
    printf("sep16 CORR: features max abs err %.3e, correlation max abs err %.3e (max |ref| %.3e)  LDS %d B\n", wf, wc, rc, G::LDS_BYTES);
}

static void count_probe(int ch) {
    constexpr int C = 256;
    using G = Sep16Geom<C, C, 3>;
    constexpr int PCH = 256 + 9 * 16 + 16;
    std::vector<float> X((size_t)256 * C, 0.f), W((size_t)16 * G::CST, 0.f), B(C, 0.f), PW_((size_t)16 * PCH, 0.f), PB(4, 0.f);
    B[ch] = 1.f;                                            // layer output: channel ch == 1 everywhere
    const int cc = ch / 16, k = ch % 16;
    for (int t = 0; t < 9; ++t) PW_[(size_t)cc * PCH + 256 + t * 16 + k] = (float)(1 << t);   // tap signature
    PW_[(size_t)cc * PCH + (0 + 16 * (k / 4)) * 4 + k % 4] = 1.f;                            // head pw: n = 0 <- channel ch
    Ir2Args a{};
    a.X = up(X); a.ldx = C; a.Wpk = up(W); a.bp = up(B); a.ldy = C; a.relu_dw = 0; a.relu_out = 1;
    a.P_Wpk = up(PW_); a.P_bp = up(PB); a.pred_cout = 4; a.pred_act = 0; a.pred_stride = 4 * 256;
    float* py_; CK(hipMalloc(&py_, (size_t)4 * 256 * 4)); a.P_Y = py_;
    auto kf = sep16_kernel<C, C, 3, false, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipLaunchKernelGGL(kf, dim3(1), dim3(512), G::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> O(4 * 256);
    CK(hipMemcpy(O.data(), py_, O.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int py = 0; py < 16; ++py) for (int px = 0; px < 16; ++px) {
        int want = 0;
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
            const int yy = py + ky - 1, xx = px + kx - 1;
            if (yy >= 0 && yy < 16 && xx >= 0 && xx < 16) want += 1 << (ky * 3 + kx);
        }
        if ((int)O[py * 16 + px] != want) { if (bad < 6) printf("   ch %d px (%d,%d): got %d want %d (diff %d)\n", ch, py, px, (int)O[py * 16 + px], want, (int)O[py * 16 + px] - want); ++bad; }
    }
    printf("count probe ch %3d: %d wrong pixels\n", ch, bad);
}

static void pos_probe(int ch) {
    constexpr int C = 256;
    using G = Sep16Geom<C, C, 3>;
    constexpr int PCH = 256 + 9 * 16 + 16;
    const int cc = ch / 16, k = ch % 16;
    for (int t = 0; t < 9; ++t) {
        std::vector<float> X((size_t)256 * C, 0.f), W((size_t)16 * G::CST, 0.f), B(C, 0.f), PW_((size_t)16 * PCH, 0.f), PB(4, 0.f);
        for (int px = 0; px < 256; ++px) X[(size_t)px * C + ch] = (float)(px + 1);
        W[(size_t)cc * G::CST + G::WPF + 4 * 16 + k] = 1.f;                                   // layer dw: centre tap
        W[(size_t)cc * G::CST + (ch / 16) * 256 + ((ch % 16) + 16 * (k / 4)) * 4 + k % 4] = 1.f;   // layer pw: n = ch <- k = ch
        PW_[(size_t)cc * PCH + 256 + t * 16 + k] = 1.f;                                       // head dw: tap t only
        PW_[(size_t)cc * PCH + (0 + 16 * (k / 4)) * 4 + k % 4] = 1.f;                         // head pw: n = 0 <- channel ch
        Ir2Args a{};
        a.X = up(X); a.ldx = C; a.Wpk = up(W); a.bp = up(B); a.ldy = C; a.relu_dw = 0; a.relu_out = 1;
        a.P_Wpk = up(PW_); a.P_bp = up(PB); a.pred_cout = 4; a.pred_act = 0; a.pred_stride = 4 * 256;
        float* py_; CK(hipMalloc(&py_, (size_t)4 * 256 * 4)); a.P_Y = py_;
        auto kf = sep16_kernel<C, C, 3, false, true>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        hipLaunchKernelGGL(kf, dim3(1), dim3(512), G::LDS_BYTES, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<float> O(4 * 256);
        CK(hipMemcpy(O.data(), py_, O.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int py = 0; py < 16; ++py) for (int px = 0; px < 16; ++px) {
            const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
            const int want = (yy >= 0 && yy < 16 && xx >= 0 && xx < 16) ? yy * 16 + xx + 1 : 0;
            const int got = (int)O[py * 16 + px];
            if (got != want) { if (bad < 4) printf("   ch %d tap %d px (%d,%d): read pixel %d (%d,%d) instead of %d\n", ch, t, py, px, got - 1, (got - 1) / 16, (got - 1) % 16, want - 1); ++bad; }
        }
        printf("pos probe ch %d tap %d: %d wrong\n", ch, t, bad);
    }
}

static void dump_probe() {
    constexpr int C = 256;
    using G = Sep16Geom<C, C, 3>;
    constexpr int PCH = 256 + 9 * 16 + 16;
    auto X = rnd((size_t)256 * C, 2.f), W = rnd((size_t)16 * G::CST, 0.2f), B = rnd(C, 0.5f);
    std::vector<float> PW_((size_t)16 * PCH), PB(4, 0.f);
    for (size_t i = 0; i < PW_.size(); ++i) PW_[i] = (float)i;
    Ir2Args a{};
    a.X = up(X); a.ldx = C; a.Wpk = up(W); a.bp = up(B); a.ldy = C; a.relu_dw = 0; a.relu_out = 1;
    a.P_Wpk = up(PW_); a.P_bp = up(PB); a.pred_cout = 4; a.pred_act = 0; a.pred_stride = 4 * 256;
    float* py_; CK(hipMalloc(&py_, (size_t)4 * 256 * 4)); a.P_Y = py_;
    float* dump; CK(hipMalloc(&dump, PW_.size() * 4)); a.Y = dump;
    auto kf = sep16_kernel<C, C, 3, false, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipLaunchKernelGGL(kf, dim3(1), dim3(512), G::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> D(PW_.size());
    CK(hipMemcpy(D.data(), dump, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < D.size(); ++i) if (D[i] != (float)i) { if (bad < 24) printf("  WP[%zu] = %g\n", i, D[i]); ++bad; }
    printf("dump probe (iteration %d): %d of %zu staged floats wrong\n", FEAR_DBG_DUMP, bad, D.size());
}

int main() { check_pred(2); check<256, 256>(3); check<320, 256>(2); check_pred(2); check_corr(2); return 0; }
