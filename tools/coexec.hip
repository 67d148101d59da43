// coexec — does an MFMA wave overlap with a VALU/LDS wave on the same SIMD?  (development probe)
// 8 waves per block, 1 block per CU.  mode bit0: waves 0-3 run MFMA loop; bit1: waves 4-7 run LDS+VALU chain loop;
// mode 4: ALL waves alternate [MFMA burst][LDS/VALU burst] in lockstep (barrier per iteration);
// mode 5: same but waves 4-7 start with the LDS/VALU burst (staggered).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void mfma_burst(f32x4 (&acc)[4], float a, float b) {
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
}
__device__ __forceinline__ void valu_burst(f32x4& d0, f32x4& d1, const float* lds, int lane) {
#pragma unroll
    for (int i = 0; i < 30; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((i * 67 + lane) & 1023) * 4);
        const f32x4 w = *reinterpret_cast<const f32x4*>(lds + 4096 + (i & 15) * 4);
        d0 += v * w;
        d1 += v * w.yzwx;
    }
}

__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 512) lds[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[4] = {};
    f32x4 d0 = {}, d1 = {};
    const float a = 0.5f + lane, b = 0.25f;
    if (mode < 4) {
        if (wave < 4) { if (mode & 1) for (int it = 0; it < iters; ++it) { mfma_burst(acc, a, b); mfma_burst(acc, a, b); } }
        else          { if (mode & 2) for (int it = 0; it < iters; ++it) { valu_burst(d0, d1, lds, lane); valu_burst(d0, d1, lds, lane); } }
    } else {
        const bool first_valu = (mode == 5) && wave >= 4;
        for (int it = 0; it < iters; ++it) {
            if (first_valu) {
                valu_burst(d0, d1, lds, lane);
                __builtin_amdgcn_sched_barrier(0);
                mfma_burst(acc, a, b);
            } else {
                mfma_burst(acc, a, b);
                __builtin_amdgcn_sched_barrier(0);
                valu_burst(d0, d1, lds, lane);
            }
            __syncthreads();
        }
    }
    f32x4 r = acc[0] + acc[1] + acc[2] + acc[3] + d0 + d1;
    out[blockIdx.x * 512 + tid] = r.x + r.y + r.z + r.w;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int mode : {1, 2, 3, 4, 5}) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, iters, mode);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, iters, mode);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d: %.1f us  (%.0f cycles/iter @2.4GHz)\n", mode, ms * 1e3, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
