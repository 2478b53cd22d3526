// How fast is ONE workgroup on an otherwise idle MI355X?  Dependent integer adds, barriers, DPP scans, LDS reads: ns per operation.
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/one_cu_clock.hip -o /tmp/one_cu_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(1024) void k_adds(uint32_t* out, int n, uint32_t seed) {
    uint32_t x = seed + threadIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = x * 3u + 1u;      // dependent v_mad / v_mul+add
    }
    out[threadIdx.x] = x;
}
__global__ __launch_bounds__(1024) void k_barriers(uint32_t* out, int n) {
    __shared__ uint32_t s[32];
    uint32_t x = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = x;
        __syncthreads();
        x += s[(i + 1) & 15];
        __syncthreads();
    }
    out[threadIdx.x] = x;
}
__global__ __launch_bounds__(1024) void k_dpp(uint32_t* out, int n) {
    uint32_t x = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    }
    out[threadIdx.x] = x;
}
__global__ __launch_bounds__(1024) void k_lds(uint32_t* out, int n) {
    __shared__ uint32_t s[1024 * 13];
    for (int k = 0; k < 13; ++k) s[k * 1024 + threadIdx.x] = threadIdx.x * 7 + k;
    __syncthreads();
    uint32_t x = 0, i = threadIdx.x * 13;
    for (int it = 0; it < n; ++it) {
        for (int j = 0; j < 13; ++j) x += s[i + j] & 0xffffu;
        i = (i + (x & 1)) % (1024 * 13 - 13);
    }
    out[threadIdx.x] = x;
}
template <typename F> static float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    uint32_t* out; hipMalloc(&out, 4096 * 4);
    for (int threads : {64, 256, 1024}) {
        const int n = 20000;
        float ms = timeit([&] { hipLaunchKernelGGL(k_adds, dim3(1), dim3(threads), 0, 0, out, n, 1u); });
        printf("threads %4d: dependent mul-add chain: %.2f ns per op (%d ops)\n", threads, ms * 1e6 / (n * 16.0), n * 16);
        ms = timeit([&] { hipLaunchKernelGGL(k_barriers, dim3(1), dim3(threads), 0, 0, out, n); });
        printf("threads %4d: LDS write + barrier + LDS read + barrier: %.1f ns per iteration\n", threads, ms * 1e6 / n);
        ms = timeit([&] { hipLaunchKernelGGL(k_dpp, dim3(1), dim3(threads), 0, 0, out, n); });
        printf("threads %4d: six-step DPP wave scan: %.1f ns\n", threads, ms * 1e6 / n);
        ms = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(1), dim3(threads), 0, 0, out, n / 10); });
        printf("threads %4d: 13 consecutive LDS words + update: %.1f ns per 13 words\n", threads, ms * 1e6 / (n / 10));
    }
    // the same with the whole chip busy on the side? (clock ramp)
    for (int blocks : {1, 256, 2048}) {
        const int n = 20000;
        float ms = timeit([&] { hipLaunchKernelGGL(k_adds, dim3(blocks), dim3(256), 0, 0, out, n, 1u); });
        printf("blocks %4d x 256 threads: dependent chain %.2f ns per op\n", blocks, ms * 1e6 / (n * 16.0));
    }
    return 0;
}
