// What a grid barrier and a split-K exchange cost INSIDE one kernel on MI355X — the building blocks of a cooperative batch-1
// kernel for the seven 16x16 blocks — against the kernel boundary they would replace.
//   1. workgroup id -> XCD (HW_REG_XCC_ID) for a 256-workgroup launch
//   2. N grid barriers (agent-scope release / acquire around an atomic counter, bounded spin), 24 workgroups of 512 threads:
//      (a) the 24 workgroups of a 24-workgroup grid (spread over the 8 XCDs), (b) 24 workgroups that all sit on XCD 0
//      (256-workgroup grid, the others leave at once)
//   3. the same with a split-K exchange per iteration: every workgroup stores a 64 KB partial, barrier, workgroup y sums slice y
//      of the 24 partials and stores it, barrier, every workgroup reads the 64 KB result — checked numerically
//   4. the same work as 2 x N back-to-back kernel launches (partial store | reduce), the way the engine does it today
// build: hipcc -O3 --offload-arch=gfx950 tools/coop_probe.hip -o tools/_kb/coop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ void xcc_map_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

constexpr long SPIN_LIMIT = 4000000;     // ~ a second: a barrier that cannot complete sets *err and falls through

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned nwg, unsigned& target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        target += nwg;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { *err = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// agent-scope coherent 16-byte accesses (what a relaxed agent-scope atomic compiles to, per dword): sc1 on the instruction,
// no cache-wide write-back / invalidate
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);   // raw, untyped
}
constexpr int AUX_SC1 = 16;       // gfx940+: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1
__device__ __forceinline__ void st_sc1(float* base, long off_floats, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_of(base), (int)(off_floats * 4), 0, AUX_SC1);
}
__device__ __forceinline__ f32x4 ld_sc1(const float* base, long off_floats) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(base), (int)(off_floats * 4), 0, AUX_SC1));
}
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// barrier without cache maintenance: the exchanged data is accessed with sc1 itself
__device__ __forceinline__ void grid_barrier_nofence(unsigned* ctr, unsigned nwg, unsigned& target, int* err) {
    wait_vm();
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nwg;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { *err = 1; break; }
        }
    }
    __syncthreads();
}

struct ProbeArgs {
    unsigned* ctr;       // barrier counter (zeroed)
    unsigned* ticket;    // role counter for the XCD-local placement (zeroed)
    int* err;
    float* P;            // [S][NF][512] float4 partials
    float* Y;            // [NF][512] float4 result
    float* check;        // [S] per-workgroup checksum of the last result read
    long long* clk;      // [2] wall clock of role 0 at start / end
    int S, iters, local, exchange;   // exchange 2 = sc1 accesses + barrier without fences
};

constexpr int NF = 8;    // float4 per thread: 512 threads x 8 x 16 B = 64 KB per workgroup (256 pixels x 64 channels fp32)

__global__ __launch_bounds__(512) void probe_kernel(ProbeArgs a) {
    __shared__ int role_s;
    int y = blockIdx.x;
    if (a.local) {
        if (threadIdx.x == 0) {
            int r = -1;
            if (xcc_id() == 0) {
                r = (int)__hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r >= a.S) r = -1;
            }
            role_s = r;
        }
        __syncthreads();
        y = role_s;
        if (y < 0) return;
    }
    const int tid = threadIdx.x;
    unsigned target = 0;
    long long t0 = 0;
    if (y == 0 && tid == 0) t0 = wall_clock64();
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < a.iters; ++it) {
        if (a.exchange == 2) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float v = (float)((it + 1) * (y + 1)) + 0.001f * (float)(f * 512 + tid);
                st_sc1(a.P, (((long)y * NF + f) * 512 + tid) * 4, (f32x4){v, v + 1.f, v + 2.f, v + 3.f});
            }
            grid_barrier_nofence(a.ctr, a.S, target, a.err);
            const int total = NF * 512, per = (total + a.S - 1) / a.S;
            const int i = y * per + tid;
            if (tid < per && i < total) {
                f32x4 pv[24];
#pragma unroll
                for (int w = 0; w < 24; ++w) if (w < a.S) pv[w] = ld_sc1(a.P, ((long)w * total + i) * 4);
                wait_vm();
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 24; ++w) if (w < a.S) s += pv[w];
                st_sc1(a.Y, (long)i * 4, s);
            }
            grid_barrier_nofence(a.ctr, a.S, target, a.err);
            f32x4 yv[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) yv[f] = ld_sc1(a.Y, ((long)f * 512 + tid) * 4);
            wait_vm();
#pragma unroll
            for (int f = 0; f < NF; ++f) acc += yv[f];
            continue;
        }
        if (a.exchange) {
            // partial of workgroup y: value depends on (it, y, f, tid) so stale data shows up in the checksum
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float v = (float)((it + 1) * (y + 1)) + 0.001f * (float)(f * 512 + tid);
                *reinterpret_cast<f32x4*>(a.P + (((long)y * NF + f) * 512 + tid) * 4) = (f32x4){v, v + 1.f, v + 2.f, v + 3.f};
            }
        }
        grid_barrier(a.ctr, a.S, target, a.err);
        if (a.exchange) {
            // slice y of the NF*512 float4: summed over the S partials in order
            const int total = NF * 512, per = (total + a.S - 1) / a.S;
            const int i = y * per + tid;
            if (tid < per && i < total) {
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < a.S; ++w) s += *reinterpret_cast<const f32x4*>(a.P + ((long)w * total + i) * 4);
                *reinterpret_cast<f32x4*>(a.Y + (long)i * 4) = s;
            }
            grid_barrier(a.ctr, a.S, target, a.err);
#pragma unroll
            for (int f = 0; f < NF; ++f) acc += *reinterpret_cast<const f32x4*>(a.Y + ((long)f * 512 + tid) * 4);
        }
    }
    if (y == 0 && tid == 0) { a.clk[0] = t0; a.clk[1] = wall_clock64(); }
    if (a.exchange) {
        // checksum of the LAST iteration's result as this workgroup read it
        f32x4 last = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 lv[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) lv[f] = ld_sc1(a.Y, ((long)f * 512 + tid) * 4);
        wait_vm();
#pragma unroll
        for (int f = 0; f < NF; ++f) last += lv[f];
        float s = last.x + last.y + last.z + last.w + 0.f * acc.x;
        __shared__ float red[512];
        red[tid] = s;
        __syncthreads();
        if (tid == 0) { double t = 0; for (int i = 0; i < 512; ++i) t += red[i]; a.check[y] = (float)t; }
    }
}

// today's way: one launch stores the partials, one launch reduces them
__global__ __launch_bounds__(512) void partial_kernel(float* P, int it) {
    const int y = blockIdx.x, tid = threadIdx.x;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const float v = (float)((it + 1) * (y + 1)) + 0.001f * (float)(f * 512 + tid);
        *reinterpret_cast<f32x4*>(P + (((long)y * NF + f) * 512 + tid) * 4) = (f32x4){v, v + 1.f, v + 2.f, v + 3.f};
    }
}
__global__ __launch_bounds__(256) void reduce_kernel(const float* P, float* Y, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x, total = NF * 512;
    if (i >= total) return;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < S; ++w) s += *reinterpret_cast<const f32x4*>(P + ((long)w * total + i) * 4);
    *reinterpret_cast<f32x4*>(Y + (long)i * 4) = s;
}

int main() {
    const int S = 24, ITERS = 200;
    unsigned* d_map; CK(hipMalloc(&d_map, 256 * 4));
    hipLaunchKernelGGL(xcc_map_kernel, dim3(256), dim3(64), 0, 0, d_map);
    std::vector<unsigned> map(256);
    CK(hipMemcpy(map.data(), d_map, 256 * 4, hipMemcpyDeviceToHost));
    printf("workgroup -> XCD (first 32):");
    for (int i = 0; i < 32; ++i) printf(" %u", map[i]);
    int rr = 1; for (int i = 0; i < 256; ++i) rr &= (map[i] == (unsigned)(i % 8));
    printf("\nround robin over 8 XCDs for all 256: %s\n", rr ? "yes" : "NO");

    ProbeArgs a{};
    CK(hipMalloc(&a.ctr, 4)); CK(hipMalloc(&a.ticket, 4)); CK(hipMalloc(&a.err, 4));
    CK(hipMalloc(&a.P, (size_t)S * NF * 512 * 16)); CK(hipMalloc(&a.Y, (size_t)NF * 512 * 16));
    CK(hipMalloc(&a.check, S * 4)); CK(hipMalloc(&a.clk, 16));
    a.S = S; a.iters = ITERS;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // expected checksum of the last iteration: sum over w of ((ITERS)*(w+1) + 0.001*(f*512+tid) + {0,1,2,3})
    double expect = 0;
    for (int w = 0; w < S; ++w)
        for (int i = 0; i < NF * 512; ++i)
            for (int k = 0; k < 4; ++k) expect += (double)(float)((float)(ITERS * (w + 1)) + 0.001f * (float)i) + k;
    for (int local = 0; local < 2; ++local)
        for (int exchange = 0; exchange < 3; ++exchange)
            for (int rep = 0; rep < 3; ++rep) {
                a.local = local; a.exchange = exchange;
                CK(hipMemset(a.ctr, 0, 4)); CK(hipMemset(a.ticket, 0, 4)); CK(hipMemset(a.err, 0, 4)); CK(hipMemset(a.check, 0, S * 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe_kernel, dim3(local ? 256 : S), dim3(512), 0, 0, a);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int err; long long clk[2]; std::vector<float> chk(S);
                CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(clk, a.clk, 16, hipMemcpyDeviceToHost));
                CK(hipMemcpy(chk.data(), a.check, S * 4, hipMemcpyDeviceToHost));
                int bad = 0;
                if (exchange) for (int w = 0; w < S; ++w) if (fabs(chk[w] - expect) > 1e-4 * expect) ++bad;
                printf("placement %-8s %-24s: %7.2f us per iteration (device clock %7.2f)  spin-limit hit: %d  wrong checksums: %d\n",
                       local ? "XCD 0" : "spread", exchange == 2 ? "2 barriers + sc1 exchange" : exchange ? "2 barriers + exchange" : "1 barrier", 1e3 * ms / ITERS,
                       (clk[1] - clk[0]) * 0.01 / ITERS, err, bad);
            }
    // kernel-boundary version of the exchange
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int it = 0; it < ITERS; ++it) {
            hipLaunchKernelGGL(partial_kernel, dim3(S), dim3(512), 0, 0, a.P, it);
            hipLaunchKernelGGL(reduce_kernel, dim3((NF * 512 + 255) / 256), dim3(256), 0, 0, a.P, a.Y, S);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("two launches per iteration (partial store | reduce): %7.2f us per iteration\n", 1e3 * ms / ITERS);
    }
    return 0;
}
