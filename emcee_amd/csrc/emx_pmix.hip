// k_persist_mix: DEMove and DESnookerMove steps of a mixture in ONE persistent launch (emx_persist_mix.hpp) -- a translation
// unit of its own (the instantiations are large; the build compiles the units in parallel).
#include "emx_persist_mix.hpp"
#include "emx_launch.hpp"

namespace emx {

template <int CH, int DPB, bool LOCAL>
static hipError_t launch_mix(dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
    auto kern = k_persist_mix<8, 2, CH, DPB, LOCAL>;
    static size_t lds_granted[MAX_DEVICES] = {};
    int dev = 0;
    if (lds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_granted[dev] = lds;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, P);
    return hipGetLastError();
}

hipError_t launch_persist_mix(int dpb, int local, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
#define EMX_CASE(b, ch)                                                                      \
    if (dpb == b) return local ? launch_mix<ch, b, true>(grid, block, lds, st, P) : launch_mix<ch, b, false>(grid, block, lds, st, P);
    EMX_CASE(1, 1) EMX_CASE(2, 2) EMX_CASE(3, 4) EMX_CASE(4, 4)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

hipError_t persist_mix_occupancy(int dpb, int threads, size_t lds, int* per_cu) {
#define EMX_CASE(b, ch)                                                                                                  \
    if (dpb == b) {                                                                                                      \
        auto kern = k_persist_mix<8, 2, ch, b, false>;                                                             \
        if (lds > 48 * 1024) {                                                                                           \
            const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                               \
        }                                                                                                                \
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, threads, lds);                                 \
    }
    EMX_CASE(1, 1) EMX_CASE(2, 2) EMX_CASE(3, 4) EMX_CASE(4, 4)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

}  // namespace emx
