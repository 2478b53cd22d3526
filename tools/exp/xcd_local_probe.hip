// Can the workgroups that land on ONE XCD (blockIdx & 7 == 0 of an 8x oversized grid) synchronise through that XCD's L2?
// A flag barrier -- workgroup g stores k into word g, lane l of the first wave polls word l -- with the store / load flavours as
// template parameters; reports which flavours ever complete, the rounds per microsecond, and the XCC_IDs seen.
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/xcd_local_probe.hip -o tools/exp/xcd_local_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
// CPOL bits of the buffer builtins: 1 = sc0, 2 = nt, 16 = sc1
template <int ST, int LD, bool INV>
__global__ __launch_bounds__(64) void k_flagbar(unsigned* flags, unsigned* out, int ngroups, int rounds, unsigned long long timeout) {
    if ((blockIdx.x & 7u) != 0u) return;
    const unsigned bid = blockIdx.x >> 3;
    const int lane = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const __amdgpu_buffer_rsrc_t Fr = __builtin_amdgcn_make_buffer_rsrc((void*)flags, 0, 64 * 4, 0x00020000);
    const unsigned long long t_begin = wall_clock64();
    int done = 0;
    for (int k = 1; k <= rounds; ++k) {
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)k, Fr, (int)bid * 4, 0, ST);
        const unsigned long long t0 = wall_clock64();
        bool ok_all = false;
        for (;;) {
            if (INV) asm volatile("buffer_inv sc0" ::: "memory");
            const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(Fr, lane * 4, 0, LD);
            const bool ok = lane < ngroups ? (int)(v - (unsigned)k) >= 0 : true;
            if (__ballot(ok) == ~0ull) { ok_all = true; break; }
            if (wall_clock64() - t0 > timeout) break;
        }
        if (!ok_all) break;
        ++done;
    }
    if (lane == 0) {
        out[bid * 4 + 0] = xcc & 0xf;
        out[bid * 4 + 1] = (unsigned)done;
        out[bid * 4 + 2] = (unsigned)(wall_clock64() - t_begin);
    }
}
template <int ST, int LD, bool INV>
static void run(const char* name, int ngroups) {
    unsigned *flags, *out;
    hipMalloc(&flags, 64 * 4);
    hipMalloc(&out, 64 * 16);
    hipMemset(flags, 0, 64 * 4);
    hipMemset(out, 0, 64 * 16);
    const int rounds = 1000;
    hipLaunchKernelGGL((k_flagbar<ST, LD, INV>), dim3(ngroups * 8), dim3(64), 0, 0, flags, out, ngroups, rounds, 20000000ull /* 0.2 s */);
    hipDeviceSynchronize();
    std::vector<unsigned> h(64 * 4);
    hipMemcpy(h.data(), out, ngroups * 16, hipMemcpyDeviceToHost);
    unsigned mind = rounds, maxt = 0, xmask = 0;
    for (int g = 0; g < ngroups; ++g) { mind = h[g * 4 + 1] < mind ? h[g * 4 + 1] : mind; maxt = h[g * 4 + 2] > maxt ? h[g * 4 + 2] : maxt; xmask |= 1u << h[g * 4]; }
    printf("%-44s groups %2d: rounds completed %4u of %d, %.3f us per round, XCC mask 0x%x\n", name, ngroups, mind, rounds, mind ? maxt * 0.01 / mind : 0.0, xmask);
    hipFree(flags); hipFree(out);
}
int main() {
    for (int ng : {8, 32}) {
        run<0, 0, true>("plain store, buffer_inv sc0 + plain load", ng);
        run<0, 1, false>("plain store, sc0 load", ng);
        run<1, 1, false>("sc0 store, sc0 load", ng);
        run<0, 16, false>("plain store, sc1 load", ng);
        run<16, 16, false>("sc1 store, sc1 load (the device-wide flavour)", ng);
        run<16, 0, true>("sc1 store, buffer_inv sc0 + plain load", ng);
        run<0, 2, false>("plain store, nt load", ng);
    }
    return 0;
}
