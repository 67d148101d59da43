// issue_probe — what one vector-ALU instruction costs on a gfx950 SIMD, alone and beside fp32 MFMAs (development tool).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/issue_probe tools/issue_probe.hip
// Every kernel is one straight-line block of N instructions (".rept") inside a short loop, timed with s_memtime by lane 0 of
// wave 0 of workgroup 0; launched with 1, 2 and 4 waves per SIMD (256 / 512 / 1024 threads, one workgroup per CU) the
// printed number is SIMD cycles per instruction group = elapsed * 1 / (iterations * groups * waves_per_simd).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ITERS 64

// body: an asm string that forms ONE group; it is repeated REPT times back to back
#define PROBE(NAME, REPT, BODY)                                                                                   \
    __global__ __launch_bounds__(1024) void NAME(long long* out, float* sink, int iters) {                        \
        float s0 = 1.f + threadIdx.x, s1 = 2.f, s2 = 3.f, s3 = 4.f, x = 1.0001f, y = 0.999f;                        \
        f32x2 p0 = {1.f, 2.f}, p1 = p0, p2 = p0, p3 = p0, p4 = {0.5f, 0.25f}, p5 = p4;                            \
        f32x4 q0 = {1.f, 2.f, 3.f, 4.f}, q1 = q0, q2 = q0, q3 = q0;                                               \
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;                                               \
        int q = threadIdx.x, i0 = 1, i1 = 2, i2 = 3, i3 = 4;                                                        \
        __shared__ float sh[8192];                                                                                  \
        for (int i = threadIdx.x; i < 8192; i += blockDim.x) sh[i] = x;                                             \
        __syncthreads();                                                                                            \
        const unsigned la = (unsigned)(size_t)sh + (threadIdx.x & 63) * 16;                                         \
        const long long t0 = __builtin_readcyclecounter();                                                          \
        for (int it = 0; it < iters; ++it) {                                                                        \
            asm volatile(".rept " #REPT "\n" BODY "\n.endr\n"                                                      \
                         : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3), [p0] "+v"(p0), [p1] "+v"(p1), \
                           [p2] "+v"(p2), [p3] "+v"(p3), [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2), [q3] "+v"(q3), \
                           [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [i0] "+v"(i0), [i1] "+v"(i1), \
                           [i2] "+v"(i2), [i3] "+v"(i3)                                                             \
                         : [x] "v"(x), [y] "v"(y), [p4] "v"(p4), [p5] "v"(p5), [q] "v"(q), [la] "v"(la)             \
                         : "memory");                                                                               \
        }                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        const long long t1 = __builtin_readcyclecounter();                                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;                                                  \
        if (x == 123.456f) sink[threadIdx.x] = s0 + s1 + s2 + s3 + p0.x + p1.x + p2.x + p3.x + q0.x + q1.x + q2.x + q3.x + c0.x + c1.x + c2.x + c3.x + i0 + i1 + i2 + i3; \
    }

#define MFMA1A "v_mfma_f32_16x16x4_f32 %[c0], %[x], %[y], %[c0]\n"
#define MFMA1B "v_mfma_f32_16x16x4_f32 %[c1], %[x], %[y], %[c1]\n"
#define MFMA1C "v_mfma_f32_16x16x4_f32 %[c2], %[x], %[y], %[c2]\n"
#define MFMA1D "v_mfma_f32_16x16x4_f32 %[c3], %[x], %[y], %[c3]\n"
#define MFMA4 MFMA1A MFMA1B MFMA1C MFMA1D
#define MAX1A "v_max_f32 %[s0], 0, %[s0]\n"
#define MAX1B "v_max_f32 %[s1], 0, %[s1]\n"
#define MAX1C "v_max_f32 %[s2], 0, %[s2]\n"
#define MAX1D "v_max_f32 %[s3], 0, %[s3]\n"
#define MAX4 MAX1A MAX1B MAX1C MAX1D
#define FMA4 "v_fma_f32 %[s0], %[x], %[y], %[s0]\n v_fma_f32 %[s1], %[x], %[y], %[s1]\n v_fma_f32 %[s2], %[x], %[y], %[s2]\n v_fma_f32 %[s3], %[x], %[y], %[s3]\n"
#define PKFMA1A "v_pk_fma_f32 %[p0], %[p4], %[p5], %[p0]\n"
#define PKFMA1B "v_pk_fma_f32 %[p1], %[p4], %[p5], %[p1]\n"
#define PKFMA1C "v_pk_fma_f32 %[p2], %[p4], %[p5], %[p2]\n"
#define PKFMA1D "v_pk_fma_f32 %[p3], %[p4], %[p5], %[p3]\n"
#define PKFMA4 PKFMA1A PKFMA1B PKFMA1C PKFMA1D
#define PKMUL4 "v_pk_mul_f32 %[p0], %[p4], %[p5]\n v_pk_mul_f32 %[p1], %[p4], %[p5]\n v_pk_mul_f32 %[p2], %[p4], %[p5]\n v_pk_mul_f32 %[p3], %[p4], %[p5]\n"
#define PKADD4 "v_pk_add_f32 %[p0], %[p4], %[p5]\n v_pk_add_f32 %[p1], %[p4], %[p5]\n v_pk_add_f32 %[p2], %[p4], %[p5]\n v_pk_add_f32 %[p3], %[p4], %[p5]\n"
#define MULLO4 "v_mul_lo_u32 %[i0], %[q], %[q]\n v_mul_lo_u32 %[i1], %[q], %[q]\n v_mul_lo_u32 %[i2], %[q], %[q]\n v_mul_lo_u32 %[i3], %[q], %[q]\n"
#define LSHLADD4 "v_lshl_add_u32 %[i0], %[q], 2, %[q]\n v_lshl_add_u32 %[i1], %[q], 2, %[q]\n v_lshl_add_u32 %[i2], %[q], 2, %[q]\n v_lshl_add_u32 %[i3], %[q], 2, %[q]\n"
#define MOV4 "v_mov_b32 %[i0], %[q]\n v_mov_b32 %[i1], %[q]\n v_mov_b32 %[i2], %[q]\n v_mov_b32 %[i3], %[q]\n"
#define MAXI4 "v_max_i32 %[i0], 0, %[i0]\n v_max_i32 %[i1], 0, %[i1]\n v_max_i32 %[i2], 0, %[i2]\n v_max_i32 %[i3], 0, %[i3]\n"
#define DSR1A "ds_read_b128 %[q0], %[la]\n"
#define DSR1B "ds_read_b128 %[q1], %[la] offset:1024\n"
#define DSR1C "ds_read_b128 %[q2], %[la] offset:2048\n"
#define DSR1D "ds_read_b128 %[q3], %[la] offset:3072\n"
#define DSR4 DSR1A DSR1B DSR1C DSR1D
#define DSW1 "ds_write_b128 %[la], %[q0] offset:4096\n"
#define WAITL "s_waitcnt lgkmcnt(0)\n"

PROBE(p_mfma, 16, MFMA4)                                  // 4 MFMAs per group, independent accumulators
PROBE(p_mfma_dep, 64, MFMA1A)                             // 1 MFMA per group, every one depends on the previous
PROBE(p_mfma_dep2, 32, MFMA1A MFMA1B)                     // two chains alternating
PROBE(p_max, 16, MAX4)
PROBE(p_maxi, 16, MAXI4)
PROBE(p_fma, 16, FMA4)
PROBE(p_pkfma, 16, PKFMA4)
PROBE(p_pkmul, 16, PKMUL4)
PROBE(p_pkadd, 16, PKADD4)
PROBE(p_mullo, 16, MULLO4)
PROBE(p_lshladd, 16, LSHLADD4)
PROBE(p_mov, 16, MOV4)
PROBE(p_dsr, 16, DSR4 WAITL)
PROBE(p_dsr_nowait, 16, DSR4)
// fillers beside MFMAs: group = 4 MFMAs (128 cycles of MFMA issue) + n fillers
PROBE(p_mfma_max4, 16, MFMA1A MAX1A MFMA1B MAX1B MFMA1C MAX1C MFMA1D MAX1D)
PROBE(p_mfma_max8, 16, MFMA1A MAX1A MAX1B MFMA1B MAX1C MAX1D MFMA1C MAX1A MAX1B MFMA1D MAX1C MAX1D)
PROBE(p_mfma_max16, 16, MFMA1A MAX4 MFMA1B MAX4 MFMA1C MAX4 MFMA1D MAX4)
PROBE(p_mfma_pk4, 16, MFMA1A PKFMA1A MFMA1B PKFMA1B MFMA1C PKFMA1A MFMA1D PKFMA1B)
PROBE(p_mfma_pk8, 16, MFMA1A PKFMA1A PKFMA1B MFMA1B PKFMA1A PKFMA1B MFMA1C PKFMA1A PKFMA1B MFMA1D PKFMA1A PKFMA1B)
PROBE(p_mfma_pk16, 16, MFMA1A PKFMA4 MFMA1B PKFMA4 MFMA1C PKFMA4 MFMA1D PKFMA4)
PROBE(p_mfma_fma16, 16, MFMA1A FMA4 MFMA1B FMA4 MFMA1C FMA4 MFMA1D FMA4)
PROBE(p_mfma_int16, 16, MFMA1A LSHLADD4 MFMA1B LSHLADD4 MFMA1C LSHLADD4 MFMA1D LSHLADD4)
PROBE(p_mfma_dsr4, 16, MFMA1A DSR1A MFMA1B DSR1B MFMA1C DSR1A MFMA1D DSR1B)
PROBE(p_mfma_dsr8, 16, MFMA1A DSR1A DSR1B MFMA1B DSR1A DSR1B MFMA1C DSR1A DSR1B MFMA1D DSR1A DSR1B)
PROBE(p_mfma_dsw4, 16, MFMA1A DSW1 MFMA1B DSW1 MFMA1C DSW1 MFMA1D DSW1)
// grouped: all 4 MFMAs, then all fillers (one MFMA<->VALU switch per group instead of 8)
PROBE(p_mfma_then_max16, 16, MFMA4 MAX4 MAX4 MAX4 MAX4)
PROBE(p_mfma_then_pk16, 16, MFMA4 PKFMA4 PKFMA4 PKFMA4 PKFMA4)

struct Probe { const char* name; void (*k)(long long*, float*, int); int groups; int insts_per_group; const char* what; };

int main() {
    long long* out; float* sink;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4096 * sizeof(float)));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    const Probe probes[] = {
        {"mfma x4 indep", p_mfma, 16, 4, "MFMA"}, {"mfma dependent", p_mfma_dep, 64, 1, "MFMA"}, {"mfma 2 chains", p_mfma_dep2, 32, 2, "MFMA"},
        {"v_max_f32", p_max, 16, 4, "inst"}, {"v_max_i32", p_maxi, 16, 4, "inst"}, {"v_fma_f32", p_fma, 16, 4, "inst"},
        {"v_pk_fma_f32", p_pkfma, 16, 4, "inst"}, {"v_pk_mul_f32", p_pkmul, 16, 4, "inst"}, {"v_pk_add_f32", p_pkadd, 16, 4, "inst"},
        {"v_mul_lo_u32", p_mullo, 16, 4, "inst"}, {"v_lshl_add_u32", p_lshladd, 16, 4, "inst"}, {"v_mov_b32", p_mov, 16, 4, "inst"},
        {"ds_read_b128 (wait/4)", p_dsr, 16, 4, "inst"}, {"ds_read_b128 (no wait)", p_dsr_nowait, 16, 4, "inst"},
        {"4 mfma + 4 v_max", p_mfma_max4, 16, 1, "group"}, {"4 mfma + 8 v_max", p_mfma_max8, 16, 1, "group"},
        {"4 mfma + 16 v_max", p_mfma_max16, 16, 1, "group"}, {"4 mfma + 4 pk_fma", p_mfma_pk4, 16, 1, "group"},
        {"4 mfma + 8 pk_fma", p_mfma_pk8, 16, 1, "group"}, {"4 mfma + 16 pk_fma", p_mfma_pk16, 16, 1, "group"},
        {"4 mfma + 16 v_fma", p_mfma_fma16, 16, 1, "group"}, {"4 mfma + 16 lshl_add", p_mfma_int16, 16, 1, "group"},
        {"4 mfma + 4 ds_read", p_mfma_dsr4, 16, 1, "group"}, {"4 mfma + 8 ds_read", p_mfma_dsr8, 16, 1, "group"},
        {"4 mfma + 4 ds_write", p_mfma_dsw4, 16, 1, "group"},
        {"4 mfma THEN 16 v_max", p_mfma_then_max16, 16, 1, "group"}, {"4 mfma THEN 16 pk_fma", p_mfma_then_pk16, 16, 1, "group"},
    };
    printf("%-26s %14s %14s %14s   (SIMD cycles per %s; a group of 4 MFMAs alone = 128)\n", "probe", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "unit");
    for (const Probe& p : probes) {
        printf("%-26s", p.name);
        for (int wps : {1, 2, 4}) {
            const int threads = 256 * wps;
            for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(p.k, dim3(cus), dim3(threads), 0, 0, out, sink, ITERS);
            CK(hipDeviceSynchronize());
            long long cyc;
            CK(hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost));
            // s_memtime counts at a fixed 100 MHz on some parts: report raw ticks per unit too if they look too small
            const double per = (double)cyc / ((double)ITERS * p.groups * p.insts_per_group * wps);
            printf(" %14.2f", per);
        }
        printf("   per %s\n", p.what);
    }
    // calibrate the counter: time a known-length MFMA kernel with events
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(p_mfma, dim3(cus), dim3(256), 0, 0, out, sink, ITERS * 64);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc; CK(hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost));
        printf("calibration: %lld counter ticks in %.3f ms -> %.1f MHz counter; %d MFMAs per wave -> %.2f ns per MFMA\n", cyc, ms,
               cyc / ms * 1e-3, ITERS * 64 * 64, ms * 1e6 / (ITERS * 64 * 64));
    }
    return 0;
}
