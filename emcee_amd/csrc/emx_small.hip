// Second translation unit of libemx: every instantiation of emx::k_small_run (one workgroup runs whole emx_run calls of
// a small ensemble; emx_kernels.hpp) and its launch dispatch.  Split from emx.hip only so that the two halves of the
// template instantiation work compile in parallel.
#include <hip/hip_runtime.h>

#include "emx_kernels.hpp"

using namespace emx;

namespace {

constexpr int MAX_DEVICES = 64;      // function attributes are per device: one process may drive several GPUs

constexpr int shape_g(int cols) { return cols <= 4 ? 4 : cols <= 32 ? 8 : cols <= 64 ? 16 : cols <= 128 ? 32 : 64; }
constexpr int shape_ch(int cols) { return cols <= 8 ? 1 : cols <= 16 ? 2 : cols <= 256 ? 4 : cols <= 512 ? 8 : 16; }

template <int G, int V, int CH, int MOVESEL, bool PLANNED, int DPB = 0>
hipError_t launch_small_move(int threads, size_t lds, hipStream_t st, const SmallRunArgs& a) {
    auto kern = k_small_run<G, V, CH, MOVESEL, PLANNED, DPB>;
    static size_t lds_granted[MAX_DEVICES] = {};
    int dev = 0;
    if (lds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_granted[dev] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, st, a);
    return hipGetLastError();
}

template <int G, int V, int CH>
hipError_t launch_small(int move, int threads, size_t lds, hipStream_t st, const SmallRunArgs& a) {
    const bool planned = a.plans != nullptr;
    switch (move) {
        case MOVE_STRETCH:
            return planned ? launch_small_move<G, V, CH, MOVE_STRETCH, true>(threads, lds, st, a)
                           : launch_small_move<G, V, CH, MOVE_STRETCH, false>(threads, lds, st, a);
        case MOVE_DE:
            return planned ? launch_small_move<G, V, CH, MOVE_DE, true>(threads, lds, st, a)
                           : launch_small_move<G, V, CH, MOVE_DE, false>(threads, lds, st, a);
        case MOVE_SNOOKER:
            return planned ? launch_small_move<G, V, CH, MOVE_SNOOKER, true>(threads, lds, st, a)
                           : launch_small_move<G, V, CH, MOVE_SNOOKER, false>(threads, lds, st, a);
        case MOVE_GAUSS:       // native mode only: the exact mode's normals come from the host
            return planned ? hipErrorInvalidValue : launch_small_move<G, V, CH, MOVE_GAUSS, false>(threads, lds, st, a);
        case SMALL_ANY_MOVE:
            return planned ? launch_small_move<G, V, CH, SMALL_ANY_MOVE, true>(threads, lds, st, a)
                           : launch_small_move<G, V, CH, SMALL_ANY_MOVE, false>(threads, lds, st, a);
    }
    return hipErrorInvalidValue;
}

// dense target in the one-workgroup kernel: a single stretch move, or any schedule (the kernel then carries all three)
template <int DPB, int V>
hipError_t launch_small_dense(int move, int threads, size_t lds, hipStream_t st, const SmallRunArgs& a) {
    constexpr int cols = DPB * 16 / V;
    constexpr int G = shape_g(cols), CH = shape_ch(cols);
    const bool planned = a.plans != nullptr;
    if (move == MOVE_STRETCH)
        return planned ? launch_small_move<G, V, CH, MOVE_STRETCH, true, DPB>(threads, lds, st, a)
                       : launch_small_move<G, V, CH, MOVE_STRETCH, false, DPB>(threads, lds, st, a);
    return planned ? launch_small_move<G, V, CH, SMALL_ANY_MOVE, true, DPB>(threads, lds, st, a)
                   : launch_small_move<G, V, CH, SMALL_ANY_MOVE, false, DPB>(threads, lds, st, a);
}

}  // namespace

// (G, V, CH): row layout picked by emx.hip's pick_shape; dpb > 0: dense Gaussian target with Dp = 16 dpb
// (declared inside emx.hip's extern "C" region: same unmangled name here; it is not part of the public ABI)
extern "C" hipError_t emx_small_dispatch(int G, int V, int CH, int dpb, int movesel, int threads, size_t lds, hipStream_t st,
                                         const SmallRunArgs& a) {
    hipError_t e = hipErrorInvalidValue;
    if (dpb > 0) {
#define EMX_DCASE(b, v) \
    if (dpb == b && V == v) e = launch_small_dense<b, v>(movesel, threads, lds, st, a);
        EMX_DCASE(1, 1) EMX_DCASE(2, 1) EMX_DCASE(3, 1) EMX_DCASE(4, 1) EMX_DCASE(5, 1) EMX_DCASE(6, 1) EMX_DCASE(7, 1)
        EMX_DCASE(1, 2) EMX_DCASE(2, 2) EMX_DCASE(3, 2) EMX_DCASE(4, 2) EMX_DCASE(5, 2) EMX_DCASE(6, 2) EMX_DCASE(7, 2)
#undef EMX_DCASE
    } else {
#define EMX_CASE(g, v, ch) \
    if (G == g && V == v && CH == ch) e = launch_small<g, v, ch>(movesel, threads, lds, st, a);
        EMX_CASE(4, 1, 1) EMX_CASE(8, 1, 1) EMX_CASE(8, 1, 2) EMX_CASE(8, 1, 4) EMX_CASE(16, 1, 4) EMX_CASE(32, 1, 4) EMX_CASE(64, 1, 4)
        EMX_CASE(4, 2, 1) EMX_CASE(8, 2, 1) EMX_CASE(8, 2, 2) EMX_CASE(8, 2, 4) EMX_CASE(16, 2, 4) EMX_CASE(32, 2, 4)
#undef EMX_CASE
    }
    return e;
}
