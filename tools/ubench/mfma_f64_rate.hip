// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate (register operands only), 1 / 2 / 4 waves per SIMD, all CUs,
// short (10 us) and long (300 us) kernels -- what is the matrix-pipe floor k_wide_lp and k_halfstep are priced against?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k(int iters, double* out, unsigned long long* cyc) {
    d4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 1.2345) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cyc[0] = c1 - c0;
        cyc[1] = w1 - w0;
    }
}

int main() {
    double* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 8));
    CK(hipMalloc(&cyc, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4}) {
        for (int iters : {20, 2000}) {
            const int threads = 64 * 4 * wps, blocks = 256, NACC = 8;
            hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2];
            CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
            const double nm = (double)iters * 4 * NACC;           // MFMAs per wave
            const double flops = nm * 2048.0 * blocks * 4 * wps;
            printf("waves/SIMD %d, %6.0f MFMAs per wave: kernel %8.1f us  %6.1f TFLOP/s | wave 0: %.1f shader cycles per MFMA per SIMD, "
                   "shader clock %.2f GHz\n", wps, nm, ms * 1e3, flops / (ms * 1e-3) / 1e12, (double)h[0] / (nm * wps),
                   (double)h[0] / ((double)h[1] * 10.0));
        }
    }
    return 0;
}
