// libemx, the two one-off operations around the hot loop that the reference does with NumPy on the host:
//   emx_autocorr              integrated autocorrelation time of the device-resident chain (autocorr.py:20-123)
//   emx_walkers_independent   the initial-state conditioning check (ensemble.py:653-663)
// Both sit behind the C ABI so that a consumer of include/emx.h that is not Python gets them too.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/emx.h"
#include "emx_internal.hpp"

namespace {

#define AUX_HIP(ctx, expr)                                                                                    \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) {                                                                               \
            char _b[384];                                                                                     \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return emx_internal_fail(ctx, -2, _b);                                                            \
        }                                                                                                     \
    } while (0)

// ---- hipFFT, resolved at run time (PyTorch bundles its own copy; the process should hold one) ----------------------
struct FftApi {
    void* h = nullptr;
    int (*PlanMany)(void**, int, int*, int*, int, int, int*, int, int, int, int) = nullptr;
    int (*SetStream)(void*, hipStream_t) = nullptr;
    int (*ExecD2Z)(void*, double*, void*) = nullptr;
    int (*ExecZ2D)(void*, void*, double*) = nullptr;
    int (*Destroy)(void*) = nullptr;
} g_fft;
constexpr int FFT_D2Z = 0x6a, FFT_Z2D = 0x6c;

int fft_load(const char* path, std::string& err) {
    if (g_fft.h) return 0;
    const char* cands[] = {path, getenv("EMX_HIPFFT_LIB"), "libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so"};
    void* h = nullptr;
    for (const char* cnd : cands) {
        if (!cnd || !*cnd) continue;
        h = dlopen(cnd, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        err = std::string("cannot load libhipfft: ") + (dlerror() ? dlerror() : "not found");
        return -5;
    }
    g_fft.PlanMany = (int (*)(void**, int, int*, int*, int, int, int*, int, int, int, int))dlsym(h, "hipfftPlanMany");
    g_fft.SetStream = (int (*)(void*, hipStream_t))dlsym(h, "hipfftSetStream");
    g_fft.ExecD2Z = (int (*)(void*, double*, void*))dlsym(h, "hipfftExecD2Z");
    g_fft.ExecZ2D = (int (*)(void*, void*, double*))dlsym(h, "hipfftExecZ2D");
    g_fft.Destroy = (int (*)(void*))dlsym(h, "hipfftDestroy");
    if (!g_fft.PlanMany || !g_fft.SetStream || !g_fft.ExecD2Z || !g_fft.ExecZ2D || !g_fft.Destroy) {
        err = "libhipfft lacks hipfftPlanMany / hipfftExecD2Z / hipfftExecZ2D";
        return -5;
    }
    g_fft.h = h;
    return 0;
}

// ---- autocorrelation kernels ----------------------------------------------------------------------------------------
// series s = (walker, dim) of the chunk, sample t = stored step t0 + t * thin.  The chain is [step][walker][dim]: threads
// that are consecutive in s read consecutive doubles.
__global__ __launch_bounds__(256) void k_series_mean(const double* __restrict__ chain, double* __restrict__ mean, int64_t s0,
                                                     int64_t nser, int64_t ND, int64_t t0, int64_t thin, int64_t nt) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nser) return;
    const double* p = chain + t0 * ND + s0 + s;
    double acc = 0.0;
    for (int64_t t = 0; t < nt; ++t) acc += p[t * thin * ND];
    mean[s] = acc / (double)nt;
}

// buf[s][t] = x[t][s] - mean[s] for t < nt, 0 up to L: a 64 x 64 tile goes through LDS so that both sides are coalesced
__global__ __launch_bounds__(256) void k_series_gather(const double* __restrict__ chain, const double* __restrict__ mean,
                                                       double* __restrict__ buf, int64_t s0, int64_t nser, int64_t ND,
                                                       int64_t t0, int64_t thin, int64_t nt, int64_t L) {
    __shared__ double tile[64][65];
    const int64_t sb = (int64_t)blockIdx.x * 64, tb = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
    for (int r = ty; r < 64; r += 4) {                               // r: time inside the tile, tx: series
        const int64_t t = tb + r, s = sb + tx;
        double v = 0.0;
        if (t < nt && s < nser) v = chain[(t0 + t * thin) * ND + s0 + s] - mean[s];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                               // r: series inside the tile, tx: time
        const int64_t s = sb + r, t = tb + tx;
        if (s < nser && t < L) buf[s * L + t] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void k_power(double2* __restrict__ f, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 z = f[i];
    f[i] = double2{z.x * z.x + z.y * z.y, 0.0};
}

// macf[d][t] += sum over the chunk's walkers of acf_w,d[t] / acf_w,d[0]   (autocorr.py:33-34, :93-96)
__global__ __launch_bounds__(256) void k_acf_accumulate(const double* __restrict__ buf, double* __restrict__ macf, int64_t nw,
                                                        int32_t D, int64_t nt, int64_t L) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (t >= nt) return;
    double acc = 0.0;
    for (int64_t w = 0; w < nw; ++w) {
        const double* row = buf + (w * D + d) * L;
        acc += row[t] / row[0];
    }
    macf[(int64_t)d * nt + t] += acc;
}

// ---- conditioning check kernels: the matrix lives transposed, one contiguous row of N values per coordinate --------
__global__ __launch_bounds__(256) void k_transpose_in(const double* __restrict__ x, double* __restrict__ ct, int64_t N, int32_t D) {
    __shared__ double tile[64][65];
    const int64_t nb = (int64_t)blockIdx.x * 64;
    const int db = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {                               // r: walker, tx: coordinate
        const int64_t n = nb + r;
        const int d = db + tx;
        tile[r][tx] = (n < N && d < D) ? x[n * D + d] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                               // r: coordinate, tx: walker
        const int d = db + r;
        const int64_t n = nb + tx;
        if (d < D && n < N) ct[(int64_t)d * N + n] = tile[tx][r];
    }
}

template <typename F>
__device__ __forceinline__ double block_reduce(double v, F op, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = sh[0];
    for (int k = 1; k < nw; ++k) r = op(r, sh[k]);
    return r;
}

// ensemble.py:654-662 for one coordinate per block: centre, divide by the max |.|, divide by the 2-norm.  flag: a column
// that does not vary (max == 0) or a non-finite value makes the check fail outright.
__global__ __launch_bounds__(1024) void k_condition_column(double* __restrict__ ct, int64_t N, int32_t* __restrict__ bad) {
    __shared__ double sh[16];
    double* c = ct + (int64_t)blockIdx.x * N;
    auto add = [](double a, double b) { return a + b; };
    auto mx = [](double a, double b) { return a > b ? a : b; };
    double s = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) s += c[n];
    const double mean = block_reduce(s, add, sh) / (double)N;
    double m = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {
        const double v = c[n] - mean;
        c[n] = v;
        m = mx(m, fabs(v));
    }
    const double amax = block_reduce(m, mx, sh);
    if (!(amax > 0.0) || !(amax <= 1.79769313486231570815e308)) {
        if (threadIdx.x == 0) *bad = 1;
        return;
    }
    double q = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {
        const double v = c[n] / amax;
        c[n] = v;
        q += v * v;
    }
    const double nrm = sqrt(block_reduce(q, add, sh));
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) c[n] = c[n] / nrm;
}

// Householder step k, part 1 (one block): the reflector of column k below the diagonal.  v is stored in place (column k,
// entries k..N-1), scal[0] = 2 / v^T v (0: nothing to reflect), R[k][k] = -sign(c_kk) * alpha.
__global__ __launch_bounds__(1024) void k_house_vector(double* __restrict__ ct, int64_t N, int32_t D, int32_t k,
                                                       double* __restrict__ R, double* __restrict__ scal) {
    __shared__ double sh[16];
    double* c = ct + (int64_t)k * N;
    auto add = [](double a, double b) { return a + b; };
    double q = 0.0;
    for (int64_t n = k + threadIdx.x; n < N; n += blockDim.x) q += c[n] * c[n];
    const double alpha = sqrt(block_reduce(q, add, sh));
    const double ckk = c[k];
    const double beta = ckk >= 0.0 ? -alpha : alpha;             // R[k][k]
    // v = c - beta e_k  =>  v^T v = alpha^2 - 2 beta c_kk + beta^2 = 2 alpha^2 - 2 beta c_kk
    const double vtv = 2.0 * alpha * alpha - 2.0 * beta * ckk;
    __syncthreads();
    if (threadIdx.x == 0) {
        c[k] = ckk - beta;
        R[(int64_t)k * D + k] = beta;
        scal[0] = vtv > 0.0 ? 2.0 / vtv : 0.0;
    }
}

// part 2 (one block per remaining column j > k): c_j -= (2 v^T c_j / v^T v) v over rows k..N-1; R[k][j] = updated c_j[k]
__global__ __launch_bounds__(1024) void k_house_apply(double* __restrict__ ct, int64_t N, int32_t D, int32_t k,
                                                      double* __restrict__ R, const double* __restrict__ scal) {
    __shared__ double sh[16];
    const int j = k + 1 + blockIdx.x;
    const double* v = ct + (int64_t)k * N;
    double* c = ct + (int64_t)j * N;
    auto add = [](double a, double b) { return a + b; };
    double w = 0.0;
    for (int64_t n = k + threadIdx.x; n < N; n += blockDim.x) w += v[n] * c[n];
    const double f = block_reduce(w, add, sh) * scal[0];
    for (int64_t n = k + threadIdx.x; n < N; n += blockDim.x) c[n] -= f * v[n];
    __syncthreads();
    if (threadIdx.x == 0) R[(int64_t)k * D + j] = c[k];
}

// ---- host: extreme singular values of an upper-triangular R by (inverse) power iteration on R^T R -------------------
bool solve_upper_t(const std::vector<double>& R, int D, std::vector<double>& x) {      // R^T y = x, in place
    for (int i = 0; i < D; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= R[(size_t)k * D + i] * x[k];
        const double d = R[(size_t)i * D + i];
        if (d == 0.0) return false;
        x[i] = s / d;
    }
    return true;
}
bool solve_upper(const std::vector<double>& R, int D, std::vector<double>& x) {        // R y = x, in place
    for (int i = D - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < D; ++k) s -= R[(size_t)i * D + k] * x[k];
        const double d = R[(size_t)i * D + i];
        if (d == 0.0) return false;
        x[i] = s / d;
    }
    return true;
}
double norm2(const std::vector<double>& x) {
    double s = 0.0;
    for (double v : x) s += v * v;
    return std::sqrt(s);
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int emx_fft_load(const char* libhipfft_path) {
    std::string err;
    const int rc = fft_load(libhipfft_path, err);
    if (rc) return emx_internal_fail(nullptr, rc, err.c_str());
    return 0;
}

int emx_autocorr(emx_ctx* c, int64_t discard, int64_t thin, double cwin, double* tau_out, int32_t* window_out,
                 int64_t* nsamples_out) {
    EmxChainView v;
    int rc = emx_internal_chain_view(c, &v);
    if (rc) return rc;
    if (thin < 1 || discard < 0) return emx_internal_fail(c, -1, "emx_autocorr: thin >= 1 and discard >= 0");
    if (!v.chain || v.stored <= 0) return emx_internal_fail(c, -1, "emx_autocorr: no device-resident chain (emx_chain_config + stored steps)");
    // Backend.get_value slice (backend.py:56): steps discard + thin - 1, + thin, ... < stored
    const int64_t t0 = discard + thin - 1;
    const int64_t nt = t0 < v.stored ? (v.stored - t0 + thin - 1) / thin : 0;
    if (nsamples_out) *nsamples_out = nt;
    if (nt < 1) return emx_internal_fail(c, -1, "emx_autocorr: the selection is empty");
    std::string err;
    if (fft_load(nullptr, err)) return emx_internal_fail(c, -5, err.c_str());
    AUX_HIP(c, hipSetDevice(v.device));
    int64_t n = 1;
    while (n < nt) n <<= 1;                              // autocorr.py:13-17 next_pow_two
    const int64_t L = 2 * n, LC = n + 1;
    const int64_t N = v.N, ND = v.N * (int64_t)v.D;
    const int D = v.D;
    // walkers per chunk: real buffer + spectrum within ~3 GB
    const int64_t per_walker = (int64_t)D * (L * 8 + LC * 16);
    int64_t wc = std::max<int64_t>(1, std::min<int64_t>(N, (3ll << 30) / per_walker));
    if ((int64_t)D * wc > 0x7fffffffll / 2) wc = std::max<int64_t>(1, (0x7fffffffll / 2) / D);
    double *buf = nullptr, *mean = nullptr, *macf = nullptr;
    void* spec = nullptr;
    void *plan_f = nullptr, *plan_b = nullptr;
    int64_t plan_batch = 0;
    auto cleanup = [&]() {
        if (plan_f) g_fft.Destroy(plan_f);
        if (plan_b) g_fft.Destroy(plan_b);
        if (buf) hipFree(buf);
        if (mean) hipFree(mean);
        if (macf) hipFree(macf);
        if (spec) hipFree(spec);
    };
#define AUX_TRY(expr)                   \
    do {                                \
        const int _rc = (expr);         \
        if (_rc) {                      \
            cleanup();                  \
            return _rc;                 \
        }                               \
    } while (0)
    auto hipok = [&](hipError_t e, const char* what) -> int {
        if (e == hipSuccess) return 0;
        char b[256];
        snprintf(b, sizeof(b), "emx_autocorr: %s: %s", what, hipGetErrorString(e));
        return emx_internal_fail(c, -2, b);
    };
    AUX_TRY(hipok(hipMalloc((void**)&buf, (size_t)wc * D * L * 8), "buffer allocation"));
    AUX_TRY(hipok(hipMalloc(&spec, (size_t)wc * D * LC * 16), "spectrum allocation"));
    AUX_TRY(hipok(hipMalloc((void**)&mean, (size_t)wc * D * 8), "allocation"));
    AUX_TRY(hipok(hipMalloc((void**)&macf, (size_t)D * nt * 8), "allocation"));
    AUX_TRY(hipok(hipMemsetAsync(macf, 0, (size_t)D * nt * 8, v.stream), "memset"));
    for (int64_t w0 = 0; w0 < N; w0 += wc) {
        const int64_t nw = std::min(wc, N - w0), nser = nw * D;
        if (plan_batch != nser) {
            if (plan_f) g_fft.Destroy(plan_f), plan_f = nullptr;
            if (plan_b) g_fft.Destroy(plan_b), plan_b = nullptr;
            int len = (int)L;
            int e = g_fft.PlanMany(&plan_f, 1, &len, nullptr, 1, (int)L, nullptr, 1, (int)LC, FFT_D2Z, (int)nser);
            if (!e) e = g_fft.PlanMany(&plan_b, 1, &len, nullptr, 1, (int)LC, nullptr, 1, (int)L, FFT_Z2D, (int)nser);
            if (!e) e = g_fft.SetStream(plan_f, v.stream);
            if (!e) e = g_fft.SetStream(plan_b, v.stream);
            if (e) {
                cleanup();
                char b[128];
                snprintf(b, sizeof(b), "emx_autocorr: hipFFT plan creation failed (code %d, length %lld, batch %lld)", e, (long long)L,
                         (long long)nser);
                return emx_internal_fail(c, -6, b);
            }
            plan_batch = nser;
        }
        const int64_t s0 = w0 * D;
        hipLaunchKernelGGL(k_series_mean, dim3((unsigned)((nser + 255) / 256)), dim3(256), 0, v.stream, v.chain, mean, s0, nser, ND, t0,
                           thin, nt);
        hipLaunchKernelGGL(k_series_gather, dim3((unsigned)((nser + 63) / 64), (unsigned)((L + 63) / 64)), dim3(256), 0, v.stream,
                           v.chain, mean, buf, s0, nser, ND, t0, thin, nt, L);
        AUX_TRY(hipok(hipGetLastError(), "gather launch"));
        int e = g_fft.ExecD2Z(plan_f, buf, spec);
        if (!e) {
            const int64_t ne = nser * LC;
            hipLaunchKernelGGL(k_power, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, v.stream, (double2*)spec, ne);
            e = g_fft.ExecZ2D(plan_b, spec, buf);
        }
        if (e) {
            cleanup();
            return emx_internal_fail(c, -6, "emx_autocorr: hipFFT execution failed");
        }
        hipLaunchKernelGGL(k_acf_accumulate, dim3((unsigned)((nt + 255) / 256), (unsigned)D), dim3(256), 0, v.stream, buf, macf, nw, D,
                           nt, L);
        AUX_TRY(hipok(hipGetLastError(), "accumulate launch"));
    }
    std::vector<double> h((size_t)D * nt);
    AUX_TRY(hipok(hipMemcpyAsync(h.data(), macf, h.size() * 8, hipMemcpyDeviceToHost, v.stream), "copy"));
    AUX_TRY(hipok(hipStreamSynchronize(v.stream), "synchronize"));
    cleanup();
#undef AUX_TRY
    // Sokal window (autocorr.py:36-46, :98-101): taus = 2 cumsum(f) - 1, window = first M with M >= c tau(M)
    for (int d = 0; d < D; ++d) {
        const double* f = &h[(size_t)d * nt];
        double cs = 0.0;
        int64_t win = -1;
        bool any_below = false;
        std::vector<double> taus((size_t)nt);
        for (int64_t m = 0; m < nt; ++m) {
            cs += f[m] / (double)N;
            taus[m] = 2.0 * cs - 1.0;
            const bool below = (double)m < cwin * taus[m];
            any_below |= below;
            if (!below && win < 0) win = m;
        }
        // auto_window: argmin(m) if any(m) else len - 1, with m = arange < c * taus  ("below"); every lag below -> argmin of
        // an all-True array is 0
        if (!any_below) win = nt - 1;
        if (win < 0) win = 0;
        tau_out[d] = taus[win];
        if (window_out) window_out[d] = (int32_t)win;
    }
    return 0;
}

// Scratch of the conditioning check, kept per device between calls (round-5 advisor: every continuation of run_mcmc on a resident
// State runs the check -- ensemble.py:316-323 -- and paid four hipMalloc / hipFree pairs for it; a loop of short run_mcmc(None, k)
// calls, a convergence check, saw them on its fast path).  One arena per device, grown when a larger ensemble asks, never shrunk.
struct CondArena {
    void* p = nullptr;
    size_t bytes = 0;
};
static std::mutex g_cond_mu;
static CondArena g_cond_arena[64];

// coords: the (N, D) matrix on the host -- or, with on_device, already in this device's memory (the resident state of a context)
static int walkers_independent_impl(int32_t device, const double* coords, bool on_device, int64_t N, int32_t D, int32_t* independent, double* cond_out) {
    if (!coords || !independent || N < 1 || D < 1) return emx_internal_fail(nullptr, -1, "emx_walkers_independent: bad arguments");
    *independent = 0;
    if (cond_out) *cond_out = INFINITY;
    if (!on_device)
        for (int64_t i = 0; i < N * (int64_t)D; ++i)
            if (!std::isfinite(coords[i])) return 0;                  // ensemble.py:654-655 (a resident state: the factor's entries below)
    if (N < D) return 0;                                              // fewer walkers than dimensions: rank deficient
    AUX_HIP(nullptr, hipSetDevice(device));
    double *x = nullptr, *ct = nullptr, *R = nullptr, *scal = nullptr;
    int32_t* bad = nullptr;
    std::lock_guard<std::mutex> arena_lock(g_cond_mu);                // (one check at a time per process: they share the arena and the NULL stream)
    auto cleanup = [&]() {};
    auto fail = [&](hipError_t e, const char* what) -> int {
        cleanup();
        char b[256];
        snprintf(b, sizeof(b), "emx_walkers_independent: %s: %s", what, hipGetErrorString(e));
        return emx_internal_fail(nullptr, -2, b);
    };
    hipError_t e;
    {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t nd = up((size_t)N * D * 8), need = (on_device ? 0 : nd) + nd + up((size_t)D * D * 8) + 256 + 256;
        CondArena& ar = g_cond_arena[device >= 0 && device < 64 ? device : 0];
        if (ar.bytes < need) {
            if (ar.p) hipFree(ar.p);
            ar.p = nullptr;
            ar.bytes = 0;
            if ((e = hipMalloc(&ar.p, need)) != hipSuccess) return fail(e, "allocation");
            ar.bytes = need;
        }
        char* q = static_cast<char*>(ar.p);
        if (!on_device) {
            x = reinterpret_cast<double*>(q);
            q += nd;
        }
        ct = reinterpret_cast<double*>(q);
        q += nd;
        R = reinterpret_cast<double*>(q);
        q += up((size_t)D * D * 8);
        scal = reinterpret_cast<double*>(q);
        q += 256;
        bad = reinterpret_cast<int32_t*>(q);
    }
    if (!on_device && (e = hipMemcpy(x, coords, (size_t)N * D * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload");
    hipMemset(bad, 0, 4);
    hipMemset(R, 0, (size_t)D * D * 8);
    hipLaunchKernelGGL(k_transpose_in, dim3((unsigned)((N + 63) / 64), (unsigned)((D + 63) / 64)), dim3(256), 0, 0, on_device ? coords : x, ct, N, D);
    hipLaunchKernelGGL(k_condition_column, dim3((unsigned)D), dim3(1024), 0, 0, ct, N, bad);
    // Householder QR of the (N, D) matrix, column by column; R's singular values are the matrix's (backward stable, like the
    // SVD the reference takes; a Gram matrix would square the condition number and could not resolve the 1e8 threshold)
    for (int k = 0; k < D; ++k) {
        hipLaunchKernelGGL(k_house_vector, dim3(1), dim3(1024), 0, 0, ct, N, D, k, R, scal);
        if (k + 1 < D) hipLaunchKernelGGL(k_house_apply, dim3((unsigned)(D - k - 1)), dim3(1024), 0, 0, ct, N, D, k, R, scal);
    }
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "kernel launch");
    std::vector<double> Rh((size_t)D * D);
    int32_t hb = 0;
    if ((e = hipMemcpy(Rh.data(), R, Rh.size() * 8, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "download");
    if ((e = hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "download");
    cleanup();
    if (hb) return 0;                                                 // a coordinate that does not vary (ensemble.py:657-658)
    for (double r : Rh)
        if (!std::isfinite(r)) return 0;
    // sigma_max: power iteration on R^T R; sigma_min: inverse iteration through the two triangular solves
    std::vector<double> xv((size_t)D), y((size_t)D);
    for (int i = 0; i < D; ++i) xv[i] = 1.0 + 0.37 * std::sin(1.0 + i);
    double smax = 0.0;
    for (int it = 0; it < 200; ++it) {
        for (int i = 0; i < D; ++i) {                                 // y = R x
            double s = 0.0;
            for (int k2 = i; k2 < D; ++k2) s += Rh[(size_t)i * D + k2] * xv[k2];
            y[i] = s;
        }
        const double ny = norm2(y);
        for (int j = 0; j < D; ++j) {                                 // x = R^T y
            double s = 0.0;
            for (int i = 0; i <= j; ++i) s += Rh[(size_t)i * D + j] * y[i];
            xv[j] = s;
        }
        const double nx = norm2(xv);
        if (!(nx > 0.0)) break;
        const double est = nx / (ny > 0.0 ? ny : 1.0);               // |R^T y| / |y| -> sigma_max
        for (double& q : xv) q /= nx;
        if (it > 8 && std::fabs(est - smax) <= 1e-13 * est) {
            smax = est;
            break;
        }
        smax = est;
    }
    for (int i = 0; i < D; ++i) xv[i] = 1.0 + 0.41 * std::cos(2.0 + i);
    double smin = 0.0;
    bool singular = false;
    for (int it = 0; it < 200; ++it) {
        const double n0 = norm2(xv);
        for (double& q : xv) q /= n0;
        // z = (R^T R)^-1 x: R^T u = x, then R z = u;  |x| / |u| ... the Rayleigh quotient below gives sigma_min^2
        if (!solve_upper_t(Rh, D, xv)) { singular = true; break; }
        const double nu = norm2(xv);                                   // |R^-T x|
        if (!solve_upper(Rh, D, xv)) { singular = true; break; }
        const double nz = norm2(xv);
        if (!std::isfinite(nz) || !(nz > 0.0)) { singular = true; break; }
        const double est = nu / nz;                                    // -> sigma_min (|R^-T x| / |R^-1 R^-T x|)
        if (it > 8 && std::fabs(est - smin) <= 1e-12 * est) {
            smin = est;
            break;
        }
        smin = est;
    }
    const double cond = (singular || !(smin > 0.0)) ? INFINITY : smax / smin;
    if (cond_out) *cond_out = cond;
    *independent = cond <= 1e8 ? 1 : 0;                               // ensemble.py:663
    return 0;
}

int emx_walkers_independent(int32_t device, const double* coords, int64_t N, int32_t D, int32_t* independent, double* cond_out) {
    return walkers_independent_impl(device, coords, false, N, D, independent, cond_out);
}

int emx_walkers_independent_resident(emx_ctx* c, int32_t* independent, double* cond_out) {
    const double* X = nullptr;
    int64_t N = 0;
    int32_t D = 0;
    int device = 0;
    const int rc = emx_internal_state_view(c, &X, &N, &D, &device);
    if (rc) return rc;
    return walkers_independent_impl(device, X, true, N, D, independent, cond_out);
}

}  // extern "C"
#pragma GCC visibility pop
