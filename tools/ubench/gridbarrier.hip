// Micro-benchmark: cost of a device-wide barrier inside one cooperative kernel vs a kernel boundary.
// Every spin is bounded (the kernel gives up and flags an error instead of hanging).
//   A: monotonic counter, one atomicAdd per block, thread 0 spins on a relaxed device-scope load
//   B: same, but blocks first meet per XCD (8 counters), then the 8 XCD leaders meet
//   L: K empty back-to-back launches of the same grid (the launch floor the barrier would replace)
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr long long SPIN_MAX = 4000000;

__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target) {
    for (long long s = 0; s < SPIN_MAX; ++s) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__global__ __launch_bounds__(512) void kA(unsigned* ctr, unsigned* err, int K, double* sink) {
    double acc = 0;
    for (int k = 1; k <= K; ++k) {
        acc += k;                                   // stand-in for work
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (!wait_ge(ctr, (unsigned)k * gridDim.x)) atomicOr(err, 1u);
            __threadfence();
        }
        __syncthreads();
    }
    if (acc == -1) sink[0] = acc;
}

__global__ __launch_bounds__(512) void kB(unsigned* xctr /*[8*32]*/, unsigned* gctr, unsigned* err, int K, double* sink) {
    // blockIdx -> XCD is round-robin on this part: block b runs on XCD b % 8
    const int xcd = blockIdx.x & 7;
    const int per = (gridDim.x + 7 - xcd) / 8;      // blocks on my XCD
    double acc = 0;
    for (int k = 1; k <= K; ++k) {
        acc += k;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned old = __hip_atomic_fetch_add(&xctr[xcd * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)k * per - 1) {     // last arriver of this XCD goes to the global counter
                __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!wait_ge(gctr, (unsigned)k * 8)) atomicOr(err, 1u);
            __threadfence();
        }
        __syncthreads();
    }
    if (acc == -1) sink[0] = acc;
}

__global__ __launch_bounds__(512) void kEmpty(double* sink) {
    if (threadIdx.x == 9999) sink[0] = 1;
}

int main() {
    unsigned *ctr, *err;
    double* sink;
    CK(hipMalloc(&ctr, 4096));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int blocks : {256, 512}) {
        for (int threads : {256, 512}) {
            const int K = 2000;
            float ms;
            // L: back-to-back empty launches
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kEmpty, dim3(blocks), dim3(threads), 0, st, sink);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < K; ++i) hipLaunchKernelGGL(kEmpty, dim3(blocks), dim3(threads), 0, st, sink);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("blocks %d threads %d: empty launch %.2f us", blocks, threads, ms * 1e3 / K);
            // A
            {
                CK(hipMemset(ctr, 0, 4096));
                CK(hipMemset(err, 0, 4));
                int k = K;
                void* args[] = {&ctr, &err, &k, &sink};
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel((void*)kA, dim3(blocks), dim3(threads), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned h;
                CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                printf(" | barrier A %.2f us%s", ms * 1e3 / K, h ? " (TIMEOUT)" : "");
            }
            // B
            {
                CK(hipMemset(ctr, 0, 4096));
                CK(hipMemset(err, 0, 4));
                int k = K;
                unsigned* x = ctr;
                unsigned* g = ctr + 512;
                void* args[] = {&x, &g, &err, &k, &sink};
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel((void*)kB, dim3(blocks), dim3(threads), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned h;
                CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                printf(" | barrier B (per-XCD first) %.2f us%s\n", ms * 1e3 / K, h ? " (TIMEOUT)" : "");
            }
        }
    }
    return 0;
}
