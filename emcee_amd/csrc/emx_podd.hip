// k_persist for ODD ndim (round 6; verdict round 5, item 6): the same kernel (emx_kernels.hpp) in the row layouts pick_shape gives an
// odd ndim -- ONE coordinate per lane and chunk, since rows of an odd number of doubles are 8-byte aligned only (load_row_agent /
// store_row_agent: 8-byte accesses): padded ndim 16 -> rows of 8 lanes x 2 chunks, 32 -> 8 x 4, 48 and 64 -> 16 lanes x 4 chunks.
// Stretch, DE and snooker moves, device-wide and one-XCD forms; a translation unit of its own (24 instantiations of a large kernel:
// the build compiles the units in parallel).  Until this round an ensemble of 4 096 walkers on a 33-dimensional dense Gaussian took
// the per-half-step launches (12-15 us/step) where its 32-dimensional sibling ran persistently.
#include "emx_launch.hpp"

namespace emx {

template <int G, int CH, int DPB, int MOVE, bool LOCAL>
static hipError_t launch_podd(dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
    auto kern = k_persist<G, 1, CH, DPB, MOVE, LOCAL>;
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

template <int G, int CH, int DPB, bool LOCAL>
static hipError_t launch_podd_move(int move, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
    return move == MOVE_DE        ? launch_podd<G, CH, DPB, MOVE_DE, LOCAL>(grid, block, lds, st, P)
           : move == MOVE_SNOOKER ? launch_podd<G, CH, DPB, MOVE_SNOOKER, LOCAL>(grid, block, lds, st, P)
                                  : launch_podd<G, CH, DPB, MOVE_STRETCH, LOCAL>(grid, block, lds, st, P);
}

// padded ndim 16 * dpb, odd ndim: (G, CH) = (8, 2), (8, 4), (16, 4), (16, 4) for dpb = 1 ... 4 (pick_shape with V = 1)
hipError_t launch_persist_dense_odd(int dpb, int move, int local, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
#define EMX_CASE(b, g, ch) \
    if (dpb == b) return local ? launch_podd_move<g, ch, b, true>(move, grid, block, lds, st, P) : launch_podd_move<g, ch, b, false>(move, grid, block, lds, st, P);
    EMX_CASE(1, 8, 2) EMX_CASE(2, 8, 4) EMX_CASE(3, 16, 4) EMX_CASE(4, 16, 4)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

template <int G, int CH, int DPB, int MOVE>
static hipError_t podd_occupancy(int threads, size_t lds, int* per_cu) {
    auto kern = k_persist<G, 1, CH, DPB, MOVE, false>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, threads, lds);
}

hipError_t persist_dense_odd_occupancy(int dpb, int move, int threads, size_t lds, int* per_cu) {
#define EMX_CASE(b, g, ch)                                                                              \
    if (dpb == b)                                                                                       \
        return move == MOVE_DE        ? podd_occupancy<g, ch, b, MOVE_DE>(threads, lds, per_cu)         \
               : move == MOVE_SNOOKER ? podd_occupancy<g, ch, b, MOVE_SNOOKER>(threads, lds, per_cu)    \
                                      : podd_occupancy<g, ch, b, MOVE_STRETCH>(threads, lds, per_cu);
    EMX_CASE(1, 8, 2) EMX_CASE(2, 8, 4) EMX_CASE(3, 16, 4) EMX_CASE(4, 16, 4)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

}  // namespace emx
