// The headline kernel's translation unit (see emx_launch.hpp): k_halfstep<8, 2, 4, STRETCH, 4, LEAN> for LEAN = 1 (single
// replica) and 2 (block-ownership exchanges), compiled with the ILP instruction scheduler.
#include "emx_launch.hpp"

namespace emx {

hipError_t launch_hot_stretch_dense64(int lean, dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a) {
    if (lean == 2) return launch_one<8, 2, 4, MOVE_STRETCH, 4, 2>(grid, block, lds, st, a);
    return launch_one<8, 2, 4, MOVE_STRETCH, 4, 1>(grid, block, lds, st, a);
}

}  // namespace emx
