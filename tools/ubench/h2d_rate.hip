// Host -> device rates for the exact-mode plan uploads (round 5): pinned staging -> HBM by hipMemcpyAsync on one / two streams, and by a
// kernel that reads the pinned buffer itself (the k_plan_fetch form), at the chunk sizes a step's plan has (C2: 1.31 MB of raw stream
// words + 0.26 MB of order, or 1.57 MB of finished columns).
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/h2d_rate.hip -o /tmp/h2d_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_fetch(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_busy(double* x, int iters) {       // something on the compute side while the copies run
    double a = x[threadIdx.x];
    for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
    x[threadIdx.x] = a;
}
int main() {
    const size_t sizes[] = {262144, 1310720, 1572864, 8u << 20};
    const int reps = 200;
    char *h = nullptr, *d = nullptr;
    double* dx = nullptr;
    CK(hipHostMalloc((void**)&h, (size_t)reps * (2u << 20), hipHostMallocDefault));
    CK(hipMalloc((void**)&d, (size_t)reps * (2u << 20)));
    CK(hipMalloc((void**)&dx, 4096));
    memset(h, 1, (size_t)reps * (2u << 20));
    hipStream_t s[2], sc;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    for (int busy = 0; busy < 2; ++busy)
        for (size_t sz : sizes) {
            const size_t stride = sz > (2u << 20) ? 0 : (2u << 20);
            const int n = sz > (2u << 20) ? 40 : reps;
            for (int mode = 0; mode < 4; ++mode) {      // 0: one stream; 1: two streams; 2: kernel fetch, 64 workgroups; 3: kernel fetch, 8 workgroups
                CK(hipDeviceSynchronize());
                if (busy) hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, sc, dx, 4000000);
                const double t0 = now_us();
                for (int r = 0; r < n; ++r) {
                    char* src = h + (stride ? (size_t)r * stride : 0);
                    char* dst = d + (stride ? (size_t)r * stride : 0);
                    if (mode == 0) CK(hipMemcpyAsync(dst, src, sz, hipMemcpyHostToDevice, s[0]));
                    else if (mode == 1) CK(hipMemcpyAsync(dst, src, sz, hipMemcpyHostToDevice, s[r & 1]));
                    else hipLaunchKernelGGL(k_fetch, dim3(mode == 2 ? 64 : 8), dim3(256), 0, s[0], (const uint4*)src, (uint4*)dst, sz / 16);
                }
                CK(hipStreamSynchronize(s[0]));
                CK(hipStreamSynchronize(s[1]));
                const double t1 = now_us();
                printf("busy=%d chunk %8zu B  %-28s %7.1f us/chunk  %6.1f GB/s\n", busy, sz,
                       mode == 0 ? "memcpyAsync, one stream" : mode == 1 ? "memcpyAsync, two streams" : mode == 2 ? "kernel fetch, 64 x 256" : "kernel fetch, 8 x 256",
                       (t1 - t0) / n, (double)sz * n / (t1 - t0) * 1e-3);
                CK(hipDeviceSynchronize());
            }
        }
    return 0;
}
