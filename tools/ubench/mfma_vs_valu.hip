// Micro-benchmark: what does ONE wave's instruction stream cost while the OTHER wave of its SIMD issues f64 MFMAs back to
// back?  512-thread workgroups (one per CU): waves 0-3 = MFMA loop (or idle), waves 4-7 time blocks of 64 instructions of
// one kind: integer VALU, f64 VALU add, LDS write+read pairs, v_cndmask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mfma_iters, int reps, int prio, unsigned long long* out, double* sink) {
    __shared__ double lds[8192];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wib < 4) {
        d4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = d4{0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        double s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        if (s == 1.2345) sink[0] = s;
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        unsigned long long t[4] = {0, 0, 0, 0};
        unsigned x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7;
        double f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
        for (int r = 0; r < reps; ++r) {
            unsigned long long c0 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < 16; ++i) {       // 64 independent-ish integer adds
                x0 += 0x9e3779b9u; x1 += x0; x2 += 0x7f4a7c15u; x3 += x2;
            }
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            unsigned long long c1 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < 16; ++i) {       // 64 f64 adds
                f0 += 1.5; f1 += f0; f2 += 2.5; f3 += f2;
            }
            asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
            unsigned long long c2 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < 8; ++i) {        // 8 LDS write -> read round trips
                lds[threadIdx.x + 512 * i] = f0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                f1 += lds[((threadIdx.x * 9) & 511) + 512 * i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            unsigned long long c3 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < 16; ++i) {       // 64 compares + selects
                x0 = x1 > x2 ? x0 : x3; x1 = x2 > x3 ? x1 : x0; x2 = x3 > x0 ? x2 : x1; x3 = x0 > x1 ? x3 : x2;
            }
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            unsigned long long c4 = __builtin_readcyclecounter();
            t[0] += c1 - c0; t[1] += c2 - c1; t[2] += c3 - c2; t[3] += c4 - c3;
        }
        if (x0 + x1 + x2 + x3 == 12345u || f0 + f1 + f2 + f3 == 1.2345) sink[1] = f1;
        if (lane == 0)
            for (int q = 0; q < 4; ++q) out[(blockIdx.x * 4 + wib - 4) * 4 + q] = t[q];
    }
}

int main() {
    unsigned long long* out;
    double* sink;
    CK(hipMalloc(&out, 256 * 4 * 4 * 8));
    CK(hipMalloc(&sink, 16));
    const int reps = 100;
    for (int prio : {0, 1})
        for (int mf : {0, 60000}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mf, reps, prio, out, sink);
                CK(hipDeviceSynchronize());
            }
            static unsigned long long h[256 * 4 * 4];
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            double m[4] = {0, 0, 0, 0};
            for (int i = 0; i < 1024; ++i)
                for (int q = 0; q < 4; ++q) m[q] += (double)h[4 * i + q] / 1024 / reps;
            printf("MFMA waves %s, probe prio %d: 64 int adds %.0f cyc | 64 f64 adds %.0f cyc | 8 LDS write+read round trips %.0f cyc | 128 cmp+select %.0f cyc\n",
                   mf ? "busy" : "idle", prio ? 3 : 0, m[0], m[1], m[2], m[3]);
        }
    return 0;
}
