// A USER's batched log-probability written as a HIP kernel and plugged into libemx through emx_set_target_callback
// (include/emx.h): what a maintainer of a likelihood code would write to keep ensemble.py:486-487's "one call on the (Ns, ndim)
// block" on the GPU.  Test material (tests/test_gpu_device_callable.py compiles it with hipcc on the GPU box); not part of the
// product.  Target: the dense Gaussian -0.5 (x - mu)^T icov (x - mu), ndim <= 64, one wavefront per row.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct user_gauss {
    double* mu;       // device, ndim
    double* icov;     // device, ndim x ndim row-major
    int ndim;
    long long calls;  // host-side statistics for the test
    long long rows;
};

// One wavefront takes ROWS consecutive rows; lane d keeps COLUMN d of icov in registers (read once per wave, not once per
// row: a first version that re-read the matrix for every row was bound by L1 bandwidth at 61 us per launch), the residual of
// the current row is broadcast through LDS.
constexpr int ROWS = 16;

__global__ __launch_bounds__(256) void k_user_dense(const double* __restrict__ q, long long n, int D, const double* __restrict__ mu,
                                                    const double* __restrict__ icov, double* __restrict__ out) {
    __shared__ double r[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double col[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) col[j] = (j < D && lane < D) ? icov[j * D + lane] : 0.0;       // icov is symmetric
    const double m = lane < D ? mu[lane] : 0.0;
    const long long row0 = ((long long)blockIdx.x * 4 + w) * ROWS;
    for (int k = 0; k < ROWS; ++k) {
        const long long row = row0 + k;
        if (row >= n) break;                                  // wave-uniform
        const double rd = lane < D ? q[row * D + lane] - m : 0.0;
        r[w][lane] = rd;
        __builtin_amdgcn_wave_barrier();
        double y = 0.0;
#pragma unroll
        for (int j = 0; j < 64; ++j) y = fma(col[j], r[w][j], y);
        double part = rd * y;
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) out[row] = -0.5 * part;
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" {

void* user_setup(const double* mu_host, const double* icov_host, int ndim) {
    if (ndim < 1 || ndim > 64) return nullptr;
    user_gauss* u = new user_gauss();
    u->ndim = ndim;
    u->calls = u->rows = 0;
    if (hipMalloc((void**)&u->mu, ndim * 8) != hipSuccess || hipMalloc((void**)&u->icov, (size_t)ndim * ndim * 8) != hipSuccess) return nullptr;
    if (hipMemcpy(u->mu, mu_host, ndim * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(u->icov, icov_host, (size_t)ndim * ndim * 8, hipMemcpyHostToDevice) != hipSuccess)
        return nullptr;
    return u;
}

void user_stats(void* user, long long* calls, long long* rows) {
    user_gauss* u = (user_gauss*)user;
    *calls = u->calls;
    *rows = u->rows;
}

void user_teardown(void* user) {
    user_gauss* u = (user_gauss*)user;
    (void)hipFree(u->mu);
    (void)hipFree(u->icov);
    delete u;
}

// emx_device_log_prob_fn: enqueue on `hip_stream`, never synchronise
int user_log_prob(void* user, const double* coords_dev, int64_t n, int32_t ndim, double* log_prob_dev, void* hip_stream) {
    user_gauss* u = (user_gauss*)user;
    if (ndim != u->ndim) return 1;
    u->calls += 1;
    u->rows += n;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_user_dense, dim3((unsigned)((n + 4 * ROWS - 1) / (4 * ROWS))), dim3(256), 0, (hipStream_t)hip_stream, coords_dev, (long long)n, (int)ndim,
                       u->mu, u->icov, log_prob_dev);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
