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

// modes 7..9: the interleaved step structure of the fused kernels, all 8 waves, barrier per iteration:
//   30 x [2 MFMA + 4 packed FMA] + 56 MFMA.  7: FMAs on registers only; 8: FMA operands from LDS, reads 4 steps ahead;
//   9: like 7 but the FMAs first and the MFMAs after (no alternation)
template <int MODE>
__device__ __forceinline__ void step_burst(f32x4 (&acc)[4], f32x4& d0, f32x4& d1, const float* lds, int lane, float a, float b) {
    constexpr int D = 4;
    f32x4 ev[D], wv[D];
    if (MODE == 8) {
#pragma unroll
        for (int t = 0; t < D; ++t) {
            ev[t] = *reinterpret_cast<const f32x4*>(lds + ((t * 67 + lane) & 1023) * 4);
            wv[t] = *reinterpret_cast<const f32x4*>(lds + 4096 + (t & 15) * 4 + (lane >> 4) * 64);
        }
    } else {
#pragma unroll
        for (int t = 0; t < D; ++t) { ev[t] = d0 + (float)t; wv[t] = d1 + (float)t; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 9) {
#pragma unroll
        for (int t = 0; t < 30; ++t) { d0 += ev[t % D] * wv[t % D]; d1 += ev[t % D] * wv[(t + 1) % D]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 60; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 1], 0, 0, 0);
    } else {
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            const f32x4 e = ev[t % D], w = wv[t % D];
            if (MODE == 8 && t + D < 30) {
                ev[t % D] = *reinterpret_cast<const f32x4*>(lds + (((t + D) * 67 + lane) & 1023) * 4);
                wv[t % D] = *reinterpret_cast<const f32x4*>(lds + 4096 + ((t + D) & 15) * 4 + (lane >> 4) * 64);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1], 0, 0, 0);
            d0 += e * w;
            d1 += e * w.yzwx;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[2 + (i & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2 + (i & 1)], 0, 0, 0);
}

__global__ __launch_bounds__(512) void probe(float* out, long long* clk, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 512) lds[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[4] = {};
    f32x4 d0 = {}, d1 = {};
    const long long c0 = clock64(), w0 = wall_clock64();      // shader clock vs the constant 100 MHz counter
    const float a = 0.5f + lane, b = 0.25f;
    if (mode == 10 || mode == 11) {      // issue rate of v_pk_fma_f32 (10) vs v_fma_f32 (11): 8 independent chains, 2 waves per SIMD
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 p[8];
        float q[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = (f32x2){a + i, b + i};
#pragma unroll
        for (int i = 0; i < 16; ++i) q[i] = a + i;
        const f32x2 m2 = (f32x2){1.0001f, 0.9999f}, c2 = (f32x2){0.5f, 0.25f};
        for (int it = 0; it < iters; ++it) {
            if (mode == 10) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(m2.x), "v"(c2.x));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) d0.x += p[i].x + p[i].y;
#pragma unroll
        for (int i = 0; i < 16; ++i) d0.y += q[i];
    } else if (mode >= 7) {
        for (int it = 0; it < iters; ++it) {
            if (mode == 7) step_burst<7>(acc, d0, d1, lds, lane, a, b);
            else if (mode == 8) step_burst<8>(acc, d0, d1, lds, lane, a, b);
            else step_burst<9>(acc, d0, d1, lds, lane, a, b);
            __syncthreads();
        }
    } else if (mode == 6) {                                         // every wave MFMA-bound: clock under full matrix load
        for (int it = 0; it < iters; ++it) { mfma_burst(acc, a, b); mfma_burst(acc, a, b); }
    } else if (mode < 4) {
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
    if (blockIdx.x == 0 && tid == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

int main() {
    float* out;
    long long* clk;
    CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int mode : {1, 6, 7, 9, 10, 11}) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, clk, iters, mode);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, clk, iters, mode);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[2];
        CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double mhz = (double)h[0] / ((double)h[1] / 100.0);
        printf("mode %d: %.1f us  (%.0f shader cycles/iter, shader clock %.0f MHz)\n", mode, ms * 1e3, (double)h[0] / iters, mhz);
    }
    return 0;
}
