// Micro-benchmark: what a cross-CU flag hand-off costs while the SAME CU streams agent-scope (sc1) row loads -- polled through the
// vector memory pipeline (global_load sc1: queued behind the CU's own outstanding loads) or through the scalar one (s_load glc).
// 256 workgroups x 512 threads, one per CU.  Wave 0 of workgroup b plays ping-pong with wave 0 of workgroup b ^ 1 (words written with
// sc1 vector stores); waves 1 .. NS stream 1-KB rows (16-byte sc1 loads, 8 in flight per lane) from a 32 MB buffer until wave 0 is done.
//   usage: poll_path [uncached_flags 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned long long ull;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ ull poll_vec(const ull* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ ull poll_sc(const ull* p) {
    ull v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_pp(ull* flags, const u4* stream, unsigned nrows, int rounds, int ns, ull* out, unsigned* err) {
    __shared__ volatile int stop;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    const unsigned b = blockIdx.x;
    if (wib == 0) {
        ull* mine = flags + (size_t)b * 16;                  // one 128-byte line per workgroup
        const ull* theirs = flags + (size_t)(b ^ 1u) * 16;
        const ull t0 = wall_clock64();
        bool bad = false;
        for (int r = 1; r <= rounds && !bad; ++r) {
            if ((b & 1u) == 0u) {
                if (lane == 0) __hip_atomic_store(mine, (ull)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const ull w0 = wall_clock64();
            for (;;) {
                const ull v = MODE ? poll_sc(theirs) : (ull)__builtin_amdgcn_readfirstlane((int)poll_vec(theirs));
                if ((long long)(v - (ull)r) >= 0 && v < (1ull << 40)) break;
                if (wall_clock64() - w0 > 2000000ull) { bad = true; break; }      // 20 ms
            }
            if ((b & 1u) == 1u) {
                if (lane == 0) __hip_atomic_store(mine, (ull)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const ull t1 = wall_clock64();
        if (lane == 0) {
            out[b] = t1 - t0;
            if (bad) atomicOr(err, 1u);
            stop = 1;
        }
    } else if (wib <= ns) {
        unsigned x = b * 7919u + (unsigned)wib * 104729u + 12345u;
        u4 acc = {0, 0, 0, 0};
        const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, (int)(nrows * 1024u), 0x00020000);
        while (!stop) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                x = x * 1664525u + 1013904223u;
                const unsigned row = (x >> 8) % nrows;
                const u4 w = __builtin_amdgcn_raw_buffer_load_b128(R, (int)(row * 1024u + (unsigned)lane * 16u), 0, 16 /* sc1 */);
                acc ^= w;
            }
        }
        if (acc.x == 0x12345u && acc.y == 7u) out[1000 + b] = acc.z;
    }
}

int main(int argc, char** argv) {
    const bool unc = argc > 1 && atoi(argv[1]) == 1;
    ull *flags, *out;
    unsigned* err;
    u4* stream;
    const unsigned nrows = 32768;         // 32 MB
    if (unc) CK(hipExtMallocWithFlags((void**)&flags, 256 * 128, hipDeviceMallocUncached));
    else CK(hipMalloc(&flags, 256 * 128));
    CK(hipMalloc(&out, 4096 * 8));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&stream, (size_t)nrows * 1024));
    CK(hipMemset(stream, 1, (size_t)nrows * 1024));
    printf("flag words in %s memory; 256 workgroups x 512 threads; hop = one direction of the ping-pong\n", unc ? "UNCACHED (hipDeviceMallocUncached)" : "ordinary");
    const int rounds = 2000;
    for (int ns : {0, 1, 3, 7}) {
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipMemset(flags, 0, 256 * 128));
            CK(hipMemset(err, 0, 4));
            CK(hipMemset(out, 0, 4096 * 8));
            if (mode) hipLaunchKernelGGL(k_pp<1>, dim3(256), dim3(512), 0, 0, flags, stream, nrows, rounds, ns, out, err);
            else hipLaunchKernelGGL(k_pp<0>, dim3(256), dim3(512), 0, 0, flags, stream, nrows, rounds, ns, out, err);
            CK(hipDeviceSynchronize());
            ull h[256];
            unsigned e;
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
            double s = 0, mx = 0;
            for (int i = 0; i < 256; ++i) { s += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
            printf("  streaming waves per CU %d   poll %-22s hop mean %6.2f us  max %6.2f us%s\n", ns, mode ? "s_load glc (scalar)" : "global_load sc1",
                   s / 256 * 10e-3 / (2.0 * rounds), mx * 10e-3 / (2.0 * rounds), e ? "   (TIMEOUT: stale or stuck)" : "");
            fflush(stdout);
        }
    }
    return 0;
}
