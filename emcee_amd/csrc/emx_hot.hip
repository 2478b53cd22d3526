// The headline kernel's translation unit (see emx_launch.hpp): k_halfstep<8, 2, 4, STRETCH, 4, LEAN> for LEAN = 1 (single
// replica) and 2 (block-ownership exchanges), compiled with the ILP instruction scheduler.
#include "emx_launch.hpp"

namespace emx {

hipError_t launch_hot_stretch_dense64(int lean, dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a) {
    if (lean == 2) return launch_one<8, 2, 4, MOVE_STRETCH, 4, 2>(grid, block, lds, st, a);
    if (lean == 3) return launch_one<8, 2, 4, MOVE_STRETCH, 4, 3>(grid, block, lds, st, a);
    if (lean == 4) return launch_one<8, 2, 4, MOVE_STRETCH, 4, 4>(grid, block, lds, st, a);
    return launch_one<8, 2, 4, MOVE_STRETCH, 4, 1>(grid, block, lds, st, a);
}

template <int G, int V, int CH, int DPB, int MOVE, bool LOCAL = false, bool ROWS_LATE = false>
static hipError_t launch_persist(dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
    auto kern = k_persist<G, V, CH, DPB, MOVE, LOCAL, ROWS_LATE>;
    static size_t lds_granted[MAX_DEVICES] = {};
    int dev = 0;
    if (lds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_granted[dev] = lds;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, P);
    return hipGetLastError();
}

// padded ndim 16 * dpb, even ndim (two coordinates per lane): rows of 8 lanes, dpb = 1 ... 4
// rows_late: the stretch move's device-wide form that asks for the next half-step's own rows behind the MFMA phase (launches that store
// chain rows: k_persist's ROWS_LATE)
hipError_t launch_hot_persist_dense(int dpb, int move, int local, int rows_late, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
    if (rows_late && move == MOVE_STRETCH) {
#define EMX_SCASE(b, ch)                                                                                       \
    if (dpb == b)                                                                                              \
        return local ? launch_persist<8, 2, ch, b, MOVE_STRETCH, true, true>(grid, block, lds, st, P)          \
                     : launch_persist<8, 2, ch, b, MOVE_STRETCH, false, true>(grid, block, lds, st, P);
        EMX_SCASE(1, 1) EMX_SCASE(2, 2) EMX_SCASE(3, 4) EMX_SCASE(4, 4)
#undef EMX_SCASE
        return hipErrorInvalidValue;
    }
    if (local) {          // the one-XCD form
#define EMX_LCASE(b, ch)                                                                                           \
    if (dpb == b)                                                                                                  \
        return move == MOVE_DE        ? launch_persist<8, 2, ch, b, MOVE_DE, true>(grid, block, lds, st, P)        \
               : move == MOVE_SNOOKER ? launch_persist<8, 2, ch, b, MOVE_SNOOKER, true>(grid, block, lds, st, P)   \
                                      : launch_persist<8, 2, ch, b, MOVE_STRETCH, true>(grid, block, lds, st, P);
        EMX_LCASE(1, 1) EMX_LCASE(2, 2) EMX_LCASE(3, 4) EMX_LCASE(4, 4)
#undef EMX_LCASE
        return hipErrorInvalidValue;
    }
#define EMX_CASE(b, ch)                                                                                            \
    if (dpb == b)                                                                                                  \
        return move == MOVE_DE        ? launch_persist<8, 2, ch, b, MOVE_DE>(grid, block, lds, st, P)              \
               : move == MOVE_SNOOKER ? launch_persist<8, 2, ch, b, MOVE_SNOOKER>(grid, block, lds, st, P)         \
                                      : launch_persist<8, 2, ch, b, MOVE_STRETCH>(grid, block, lds, st, P);
    EMX_CASE(1, 1) EMX_CASE(2, 2) EMX_CASE(3, 4) EMX_CASE(4, 4)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

// how many workgroups of that instantiation the runtime says a CU holds at once (block size, dynamic LDS): the co-residency check
template <int G, int V, int CH, int DPB, int MOVE>
static hipError_t persist_occupancy(int threads, size_t lds, int* per_cu) {
    auto kern = k_persist<G, V, CH, DPB, MOVE>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, threads, lds);
}

hipError_t hot_persist_occupancy(int dpb, int move, int threads, size_t lds, int* per_cu) {
#define EMX_CASE(b, ch)                                                                                     \
    if (dpb == b)                                                                                           \
        return move == MOVE_DE        ? persist_occupancy<8, 2, ch, b, MOVE_DE>(threads, lds, per_cu)       \
               : move == MOVE_SNOOKER ? persist_occupancy<8, 2, ch, b, MOVE_SNOOKER>(threads, lds, per_cu)  \
                                      : persist_occupancy<8, 2, ch, b, MOVE_STRETCH>(threads, lds, per_cu);
    EMX_CASE(1, 1) EMX_CASE(2, 2) EMX_CASE(3, 4) EMX_CASE(4, 4)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

// the Gaussian Metropolis move's persistent form (k_persist_gauss): no barrier, any grid
hipError_t launch_hot_persist_gauss(int dpb, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistGaussArgs& P) {
#define EMX_CASE(b, ch)                                                                                              \
    if (dpb == b) {                                                                                                  \
        auto kern = k_persist_gauss<8, 2, ch, b>;                                                                    \
        static size_t lds_granted[MAX_DEVICES] = {};                                                                 \
        int dev = 0;                                                                                                 \
        if (lds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) { \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                           \
            lds_granted[dev] = lds;                                                                                  \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, grid, block, lds, st, P);                                                           \
        return hipGetLastError();                                                                                    \
    }
    EMX_CASE(1, 1) EMX_CASE(2, 2) EMX_CASE(3, 4) EMX_CASE(4, 4)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

}  // namespace emx
