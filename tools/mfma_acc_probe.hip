// mfma_acc_probe — what does a projection-like MFMA stream (128 MFMAs over 32 distinct accumulators, 4 per visit) cost
// per MFMA, by accumulator register class and by who emits the instruction?  (development probe)
//   k_builtin_v : __builtin MFMA, accumulators in VGPRs (hipcc's choice for a kernel under 256 registers)
//   k_builtin_a : __builtin MFMA, function flipped to the AccVGPR form by an inline asm with an "a" operand
//   k_asm_a     : inline-asm MFMA, "+a" accumulators
//   k_asm_v     : inline-asm MFMA, "+v" accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>
__device__ __forceinline__ void body(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const float a = 0.5f + lane, b = 0.25f;
    f32x4 accp[2][16];
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) { accp[0][nt] = (f32x4){a, b, a, b} + (float)nt; accp[1][nt] = (f32x4){b, a, b, a} - (float)nt; }
    const f32x4 e0 = {a, b, a + 1.f, b + 1.f}, e1 = {b, a, b + 2.f, a + 2.f}, w = {a, b, b, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 2) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(accp[0][nt]) : "v"(w[i]), "v"(e0[i]));
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(accp[1][nt]) : "v"(w[i]), "v"(e1[i]));
                } else if (KIND == 3) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accp[0][nt]) : "v"(w[i]), "v"(e0[i]));
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accp[1][nt]) : "v"(w[i]), "v"(e1[i]));
                } else {
                    accp[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], e0[i], accp[0][nt], 0, 0, 0);
                    accp[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], e1[i], accp[1][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (KIND >= 2) asm volatile("s_nop 15\n\ts_nop 15");
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) r += accp[0][nt] + accp[1][nt];
    out[blockIdx.x * 512 + threadIdx.x] = r.x + r.y + r.z + r.w;
}
__global__ __launch_bounds__(512) void k_builtin_v(float* out, int iters) { body<0>(out, iters); }
__global__ __launch_bounds__(512) void k_builtin_a(float* out, int iters) {
    float zero = 0.f;
    asm volatile("; MFMA accumulators in AccVGPRs" : : "a"(zero));
    body<1>(out, iters);
}
__global__ __launch_bounds__(512) void k_asm_a(float* out, int iters) { body<2>(out, iters); }
__global__ __launch_bounds__(512) void k_asm_v(float* out, int iters) { body<3>(out, iters); }

int main() {
    float* out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    void (*ks[4])(float*, int) = {k_builtin_v, k_builtin_a, k_asm_a, k_asm_v};
    const char* names[4] = {"builtin, VGPR accumulators", "builtin, AccVGPR accumulators", "inline asm, AccVGPR", "inline asm, VGPR"};
    for (int k = 0; k < 4; ++k) {
        hipLaunchKernelGGL(ks[k], dim3(256), dim3(512), 0, 0, out, iters);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(ks[k], dim3(256), dim3(512), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-32s %8.1f us   %.1f cycles per MFMA (2 waves per SIMD, 2.39 GHz)\n", names[k], ms * 1e3, ms * 1e-3 * 2.39e9 / iters / 256);
    }
    return 0;
}
