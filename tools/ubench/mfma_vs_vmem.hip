// Micro-benchmark: global-load round trip seen by one wave of a SIMD while the OTHER wave of that SIMD issues f64 MFMAs
// back to back (the two-waves-per-SIMD regime of k_wide_lp / k_halfstep).  512-thread workgroups, one per CU:
// waves 0-3 run the MFMA loop (or idle), waves 4-7 time batches of 4 x global_load_dwordx4 from an L2-resident buffer,
// then batches of ds_write_b128 + ds_read_b64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mfma_iters, int probes, const double2* buf, int nbuf, unsigned long long* out, double* sink) {
    __shared__ double2 lds[4096];
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
        unsigned long long tv = 0, tl = 0;
        double2 r0, r1, r2, r3;
        double acc = 0;
        for (int p = 0; p < probes; ++p) {
            const int base = ((blockIdx.x * 131 + p * 977 + wib * 61) * 64 + lane) % (nbuf - 4096);
            const unsigned long long t0 = __builtin_readcyclecounter();
            r0 = buf[base];
            r1 = buf[base + 1024];
            r2 = buf[base + 2048];
            r3 = buf[base + 3072];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_readcyclecounter();
            lds[threadIdx.x] = r0;
            lds[threadIdx.x + 512] = r1;
            lds[threadIdx.x + 1024] = r2;
            lds[threadIdx.x + 1536] = r3;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += lds[(threadIdx.x * 7) & 2047].x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_readcyclecounter();
            tv += t1 - t0;
            tl += t2 - t1;
        }
        if (acc == 1.2345) sink[1] = acc;
        if (lane == 0) {
            out[(blockIdx.x * 4 + wib - 4) * 2] = tv;
            out[(blockIdx.x * 4 + wib - 4) * 2 + 1] = tl;
        }
    }
}

int main() {
    const int nbuf = 1 << 18;      // 4 MB of double2: L2 / MALL resident
    double2* buf;
    unsigned long long* out;
    double* sink;
    CK(hipMalloc(&buf, (size_t)nbuf * 16));
    CK(hipMemset(buf, 0, (size_t)nbuf * 16));
    CK(hipMalloc(&out, 256 * 4 * 2 * 8));
    CK(hipMalloc(&sink, 16));
    const int probes = 200;
    for (int mf : {0, 40000}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mf, probes, buf, nbuf, out, sink);
            CK(hipDeviceSynchronize());
        }
        unsigned long long h[256 * 4 * 2];
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double tv = 0, tl = 0;
        for (int i = 0; i < 1024; ++i) {
            tv += (double)h[2 * i] / 1024 / probes;
            tl += (double)h[2 * i + 1] / 1024 / probes;
        }
        printf("MFMA waves %s: 4 x global_load_dwordx4 issue->landed %.0f cycles | 4 ds_write_b128 + ds_read %.0f cycles\n",
               mf ? "busy" : "idle", tv, tl);
    }
    return 0;
}
