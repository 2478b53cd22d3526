// Launch helper shared by the translation units that instantiate k_halfstep (emx.hip, emx_hot.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "emx_kernels.hpp"

namespace emx {

constexpr int MAX_DEVICES = 64;      // function attributes are per device: one process may drive several GPUs

template <int G, int V, int CH, int MOVE, int DPB, int LEAN = 0>
hipError_t launch_one(dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a) {
    auto kern = k_halfstep<G, V, CH, MOVE, DPB, LEAN>;
    static size_t lds_granted[MAX_DEVICES] = {};      // per instantiation and device: raise the dynamic-LDS limit once
    int dev = 0;
    if (lds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_granted[dev] = lds;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return hipGetLastError();
}

// The headline shape -- stretch move, dense Gaussian target, ndim 64 (k_halfstep<8,2,4,STRETCH,4,LEAN>) -- lives in a
// translation unit of its own (emx_hot.hip) built with -amdgpu-sched-strategy=max-ilp: this kernel runs two waves per SIMD
// in lock step, i.e. it lives on instruction-level parallelism inside a wave, and the ILP scheduler is worth +2.2 % there,
// while the same flag costs the element-wise kernels up to 6 % (C3), so it is not a global build flag.
hipError_t launch_hot_stretch_dense64(int lean, dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a);

}  // namespace emx
