// Micro-benchmark: random 512-byte row gathers (2 rows per walker) in two lane layouts.
//  A: row layout   - 32 lanes x 16 B per row, 2 walkers per pass, 8 passes per wave (16 walkers)
//  B: quad layout  - lane (walker = lane&15, part = lane>>4) reads 128 contiguous bytes (8 x dwordx4)
//  C: like A but all 8 passes' loads issued before use (software prefetch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void kA(const double* X, const int* ii, const int* jj, double* out, int ns, int D) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int sub = lane >> 5, gl = lane & 31;
    const int t0 = wave * 16;
    if (t0 >= ns) return;
    double acc = 0;
    for (int p = 0; p < 8; ++p) {
        const int t = t0 + p * 2 + sub;
        const int i = ii[t], j = jj[t];
        const double2 a = *(const double2*)(X + (size_t)i * D + gl * 2);
        const double2 b = *(const double2*)(X + (size_t)j * D + gl * 2);
        double q0 = b.x - (b.x - a.x) * 1.3, q1 = b.y - (b.y - a.y) * 1.3;
        acc += q0 * q0 + q1 * q1;
    }
    for (int m = 16; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (gl == 0) out[t0 + sub] = acc;
}

__global__ __launch_bounds__(256) void kC(const double* X, const int* ii, const int* jj, double* out, int ns, int D) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int sub = lane >> 5, gl = lane & 31;
    const int t0 = wave * 16;
    if (t0 >= ns) return;
    double2 a[8], b[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int t = t0 + p * 2 + sub;
        const int i = ii[t], j = jj[t];
        a[p] = *(const double2*)(X + (size_t)i * D + gl * 2);
        b[p] = *(const double2*)(X + (size_t)j * D + gl * 2);
    }
    double acc = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        double q0 = b[p].x - (b[p].x - a[p].x) * 1.3, q1 = b[p].y - (b[p].y - a[p].y) * 1.3;
        acc += q0 * q0 + q1 * q1;
    }
    for (int m = 16; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (gl == 0) out[t0 + sub] = acc;
}

template <int INTER>
__global__ __launch_bounds__(256) void kB(const double* X, const int* ii, const int* jj, double* out, int ns, int D) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int am = INTER ? (lane & 15) : (lane >> 2), ak = INTER ? (lane >> 4) : (lane & 3);
    const int t = wave * 16 + am;
    if (wave * 16 >= ns) return;
    const int i = ii[t], j = jj[t];
    const double2* pa = (const double2*)(X + (size_t)i * D + ak * 16);
    const double2* pb = (const double2*)(X + (size_t)j * D + ak * 16);
    double2 a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = pa[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = pb[k];
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double q0 = b[k].x - (b[k].x - a[k].x) * 1.3, q1 = b[k].y - (b[k].y - a[k].y) * 1.3;
        acc += q0 * q0 + q1 * q1;
    }
    if (INTER) { acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64); }
    else { acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); }
    if (ak == 0) out[t] = acc;
}

int main() {
    const int N = 65536, D = 64, ns = N / 2;
    std::vector<double> h((size_t)N * D);
    for (auto& v : h) v = rand() / (double)RAND_MAX;
    std::vector<int> hi(ns), hj(ns);
    for (int t = 0; t < ns; ++t) { hi[t] = 2 * t + (rand() & 1); hj[t] = rand() % N; }
    double *X, *out; int *ii, *jj;
    CK(hipMalloc(&X, h.size() * 8)); CK(hipMalloc(&out, ns * 8)); CK(hipMalloc(&ii, ns * 4)); CK(hipMalloc(&jj, ns * 4));
    CK(hipMemcpy(X, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(ii, hi.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(jj, hj.data(), ns * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int waves = ns / 16, blocks = (waves + 3) / 4;
    auto run = [&](const char* name, auto kern) {
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, X, ii, jj, out, ns, D);
        CK(hipEventRecord(e0));
        const int R = 200;
        for (int w = 0; w < R; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, X, ii, jj, out, ns, D);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / R, bytes = (double)ns * 2 * D * 8;
        printf("%-28s %8.2f us/launch  %8.1f GB/s (row bytes only)\n", name, us, bytes / us / 1e3);
    };
    run("A row-layout serial passes", kA);
    run("C row-layout prefetched", kC);
    run("B quad-layout interleaved", kB<1>);
    run("B quad-layout contiguous", kB<0>);
    return 0;
}
