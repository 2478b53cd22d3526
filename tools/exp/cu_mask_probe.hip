// Which XCD does bit i of a CU-masked stream's mask select?  One single-bit stream per CU, one workgroup each, reading XCC_ID / HW_ID.
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/cu_mask_probe.hip -o tools/exp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_where(uint32_t* out, int slot) {
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) {
        out[(slot + blockIdx.x) * 2] = xcc;
        out[(slot + blockIdx.x) * 2 + 1] = hwid;
    }
}
int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    printf("%s: %d CUs\n", pr.name, ncu);
    uint32_t* out;
    hipMalloc(&out, 4096 * 8);
    hipMemset(out, 0xff, 4096 * 8);
    // unmasked: where do workgroups 0..63 of one launch go?
    hipLaunchKernelGGL(k_where, dim3(64), dim3(64), 0, 0, out, 0);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(4096 * 2);
    hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    printf("unmasked launch, workgroup -> xcc:");
    for (int i = 0; i < 64; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
    for (int bit = 0; bit < ncu; ++bit) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        mask[bit >> 5] = 1u << (bit & 31);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("bit %d: stream creation failed\n", bit); continue; }
        hipLaunchKernelGGL(k_where, dim3(1), dim3(64), 0, st, out, 64 + bit);
        hipStreamSynchronize(st);
        hipStreamDestroy(st);
    }
    hipMemcpy(h.data(), out, (64 + ncu) * 8, hipMemcpyDeviceToHost);
    printf("mask bit -> xcc (se, cu of HW_ID):\n");
    for (int bit = 0; bit < ncu; ++bit) {
        const uint32_t x = h[2 * (64 + bit)], w = h[2 * (64 + bit) + 1];
        printf(" %3d:%u(%u,%u)%s", bit, x & 0xf, (w >> 13) & 0x7, (w >> 8) & 0xf, (bit % 8 == 7) ? "\n" : "");
    }
    // a mask of the 32 bits that mapped to xcc 0: do 64 workgroups all land there?
    uint32_t m0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n0 = 0;
    for (int bit = 0; bit < ncu; ++bit)
        if ((h[2 * (64 + bit)] & 0xf) == 0) { m0[bit >> 5] |= 1u << (bit & 31); ++n0; }
    hipStream_t st;
    hipExtStreamCreateWithCUMask(&st, 8, m0);
    hipMemset(out, 0xff, 4096 * 8);
    hipLaunchKernelGGL(k_where, dim3(64), dim3(256), 0, st, out, 0);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    printf("mask of the %d bits of xcc 0, 64 workgroups -> xcc:", n0);
    for (int i = 0; i < 64; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
    return 0;
}
