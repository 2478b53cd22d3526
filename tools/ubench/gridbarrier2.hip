// Micro-benchmark 2: device-wide barriers that do NOT poll the line the arrivals land on.
// (gridbarrier.hip measured 11-21 us for "atomicAdd + spin on the same counter"; the pollers were starving the arrivals.)
//   C: flag array -- block b stores flag[b] = k, one wave per block polls all the flags (256 flags = one dwordx4 per lane)
//   E: arrival counter + separate "go" word: the last arriver publishes go = k, everyone polls go (its own 128-B line)
//   F: as E but per-XCD arrival counters, the 8 last arrivers meet on a global counter, the last of those publishes go
//   G: as C but two-level: per-XCD flag rows, the XCD's block 0 polls its row then stores an XCD flag; all poll the 8 XCD flags
// each with and without the agent-scope release/acquire fences a real half-step boundary needs.
// Every spin is bounded by the wall clock (20 ms) -- the kernel flags an error and leaves instead of hanging.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool expired(unsigned long long t0) { return wall_clock64() - t0 > 2000000ull; }   // 100 MHz: 20 ms

template <bool FENCE>
__global__ __launch_bounds__(512) void kC(unsigned* flags, unsigned* err, int K, double* sink) {
    double acc = 0;
    const int nb = gridDim.x;
    for (int k = 1; k <= K; ++k) {
        acc += k;
        __syncthreads();
        if (threadIdx.x < 64) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (threadIdx.x == 0) st(&flags[blockIdx.x], (unsigned)k);
            const unsigned long long t0 = wall_clock64();
            bool ok = false;
            while (!ok) {
                bool mine = true;
                for (int f = threadIdx.x; f < nb; f += 64) mine &= (ld(&flags[f]) >= (unsigned)k);
                ok = __all(mine);
                if (!ok && expired(t0)) { if (threadIdx.x == 0) atomicOr(err, 1u); break; }
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (acc == -1) sink[0] = acc;
}

template <bool FENCE>
__global__ __launch_bounds__(512) void kE(unsigned* ctr, unsigned* go, unsigned* err, int K, double* sink) {
    double acc = 0;
    for (int k = 1; k <= K; ++k) {
        acc += k;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)k * gridDim.x - 1) st(go, (unsigned)k);
            const unsigned long long t0 = wall_clock64();
            while (ld(go) < (unsigned)k) {
                if (expired(t0)) { atomicOr(err, 1u); break; }
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (acc == -1) sink[0] = acc;
}

template <bool FENCE>
__global__ __launch_bounds__(512) void kF(unsigned* xctr /* 8 x 32 */, unsigned* gctr, unsigned* go, unsigned* err, int K, double* sink) {
    const int xcd = blockIdx.x & 7;
    const int per = (gridDim.x + 7 - xcd) / 8;
    double acc = 0;
    for (int k = 1; k <= K; ++k) {
        acc += k;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned old = __hip_atomic_fetch_add(&xctr[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)k * per - 1) {
                const unsigned o2 = __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o2 == (unsigned)k * 8 - 1) st(go, (unsigned)k);
            }
            const unsigned long long t0 = wall_clock64();
            while (ld(go) < (unsigned)k) {
                if (expired(t0)) { atomicOr(err, 1u); break; }
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (acc == -1) sink[0] = acc;
}

template <bool FENCE>
__global__ __launch_bounds__(512) void kG(unsigned* xflags /* 8 x 64 */, unsigned* xdone /* 8 x 32 */, unsigned* err, int K, double* sink) {
    const int xcd = blockIdx.x & 7, inx = blockIdx.x >> 3;
    const int per = (gridDim.x + 7 - xcd) / 8;
    double acc = 0;
    for (int k = 1; k <= K; ++k) {
        acc += k;
        __syncthreads();
        if (threadIdx.x < 64) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (threadIdx.x == 0) st(&xflags[xcd * 64 + inx], (unsigned)k);
            const unsigned long long t0 = wall_clock64();
            if (inx == 0) {      // the XCD's collector
                bool ok = false;
                while (!ok) {
                    const bool mine = (int)threadIdx.x < per ? ld(&xflags[xcd * 64 + threadIdx.x]) >= (unsigned)k : true;
                    ok = __all(mine);
                    if (!ok && expired(t0)) { if (threadIdx.x == 0) atomicOr(err, 1u); break; }
                }
                if (threadIdx.x == 0) st(&xdone[xcd * 32], (unsigned)k);
            }
            bool ok = false;
            while (!ok) {
                const bool mine = threadIdx.x < 8 ? ld(&xdone[threadIdx.x * 32]) >= (unsigned)k : true;
                ok = __all(mine);
                if (!ok && expired(t0)) { if (threadIdx.x == 0) atomicOr(err, 2u); break; }
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (acc == -1) sink[0] = acc;
}

__global__ __launch_bounds__(512) void kEmpty(double* sink) {
    if (threadIdx.x == 9999) sink[0] = 1;
}

template <typename F>
static void timeit(const char* name, hipStream_t s, unsigned* buf, unsigned* err, int K, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned h = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(buf, 0, 1 << 16, s));
        CK(hipMemsetAsync(err, 0, 4, s));
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        unsigned hh;
        CK(hipMemcpy(&hh, err, 4, hipMemcpyDeviceToHost));
        h |= hh;
    }
    printf("  %-34s %7.2f us per barrier%s\n", name, best * 1e3 / K, h ? "  (TIMEOUT)" : "");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool fine = argc > 1 && atoi(argv[1]) == 1;
    unsigned *buf, *err;
    double* sink;
    if (fine) CK(hipExtMallocWithFlags((void**)&buf, 1 << 16, hipDeviceMallocFinegrained));
    else CK(hipMalloc(&buf, 1 << 16));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&sink, 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    printf("flags/counters in %s memory\n", fine ? "fine-grained" : "ordinary (coarse-grained)");
    for (int threads : {512}) {
        const int blocks = 256, K = 2000;
        printf("blocks %d threads %d\n", blocks, threads);
        {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kEmpty, dim3(blocks), dim3(threads), 0, s, sink);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < K; ++i) hipLaunchKernelGGL(kEmpty, dim3(blocks), dim3(threads), 0, s, sink);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %-34s %7.2f us per launch\n", "empty back-to-back launches", ms * 1e3 / K);
        }
        int k = K;
        unsigned* flags = buf;               // C: 256 flags
        unsigned* ctr = buf + 1024;          // E
        unsigned* go = buf + 2048;
        unsigned* xctr = buf + 3072;         // F: 8 x 32
        unsigned* gctr = buf + 4096;
        unsigned* go2 = buf + 5120;
        unsigned* xflags = buf + 6144;       // G: 8 x 64
        unsigned* xdone = buf + 7168;        // 8 x 32
#define COOP(kern, ...)                                                                                      \
    [&] {                                                                                                    \
        void* args[] = {__VA_ARGS__};                                                                        \
        CK(hipLaunchCooperativeKernel((void*)kern, dim3(blocks), dim3(threads), args, 0, s));                \
    }
        timeit("C flags, no fence", s, buf, err, K, COOP(kC<false>, &flags, &err, &k, &sink));
        timeit("C flags, release+acquire", s, buf, err, K, COOP(kC<true>, &flags, &err, &k, &sink));
        timeit("E counter+go, no fence", s, buf, err, K, COOP(kE<false>, &ctr, &go, &err, &k, &sink));
        timeit("E counter+go, release+acquire", s, buf, err, K, COOP(kE<true>, &ctr, &go, &err, &k, &sink));
        timeit("F xcd counters+go, no fence", s, buf, err, K, COOP(kF<false>, &xctr, &gctr, &go2, &err, &k, &sink));
        timeit("F xcd counters+go, rel+acq", s, buf, err, K, COOP(kF<true>, &xctr, &gctr, &go2, &err, &k, &sink));
        timeit("G two-level flags, no fence", s, buf, err, K, COOP(kG<false>, &xflags, &xdone, &err, &k, &sink));
        timeit("G two-level flags, rel+acq", s, buf, err, K, COOP(kG<true>, &xflags, &xdone, &err, &k, &sink));
    }
    return 0;
}
