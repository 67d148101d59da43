// Cost of a dependent kernel launch on MI355X: normal launch, hipLaunchCooperativeKernel, and the two alternating in one stream
// (24 workgroups of 512 threads, trivial kernel).  build: hipcc -O3 --offload-arch=gfx950 tools/coop_launch_probe.hip -o /tmp/coop_launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512) void k(float* p, int it) { p[(blockIdx.x * 512 + threadIdx.x) % 4096] = (float)it; }
int main() {
    float* d; CK(hipMalloc(&d, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < 200; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k, dim3(24), dim3(512), 0, s, d, it);
                else { void* args[] = {&d, &it}; CK(hipLaunchCooperativeKernel((void*)k, dim3(24), dim3(512), args, 0, s)); }
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s launch: %.2f us per dependent launch\n", mode ? "cooperative" : "normal", 1e3 * ms / 200);
        }
    // alternating normal / cooperative in one stream (what a plan would do)
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int it = 0; it < 200; ++it) {
            if (it & 1) hipLaunchKernelGGL(k, dim3(24), dim3(512), 0, s, d, it);
            else { void* args[] = {&d, &it}; CK(hipLaunchCooperativeKernel((void*)k, dim3(24), dim3(512), args, 0, s)); }
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("alternating: %.2f us per dependent launch\n", 1e3 * ms / 200);
    }
    return 0;
}
