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
// while the same flag costs the element-wise kernels up to 6 % (C3), so it is not a global build flag (nor one for C4's DE /
// snooker kernels of the same shape: 33.2 -> 33.9 us/step under it, round 3, profiles/r03/ab_hot_de_snooker.txt).
hipError_t launch_hot_stretch_dense64(int lean, dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a);
// ... and its persistent form (k_persist; padded ndim 16 ... 64 with two coordinates per lane, stretch or DE move): `P.niter`
// half-steps in one launch.  The grid must be co-resident
// (one workgroup per CU at most).
hipError_t launch_hot_persist_dense(int dpb, int move, int local, int rows_late, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P);
// workgroups of that k_persist instantiation a CU holds at once, by the runtime's occupancy calculator
hipError_t hot_persist_occupancy(int dpb, int move, int threads, size_t lds, int* per_cu);
// ... and the Gaussian Metropolis move's (k_persist_gauss: a wave keeps its walkers in registers; no barrier, any grid)
hipError_t launch_hot_persist_gauss(int dpb, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistGaussArgs& P);
// emx_pmix.hip: k_persist<..., MOVE_MIX> -- DEMove and DESnookerMove steps of a mixture in one launch (either form)
hipError_t launch_persist_mix(int dpb, int local, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P);
hipError_t persist_mix_occupancy(int dpb, int threads, size_t lds, int* per_cu);
// emx_pvalu.hip: the persistent kernel for the element-wise targets (one-XCD form)
hipError_t launch_persist_valu(int G, int V, int CH, int move, int local, dim3 grid, dim3 block, hipStream_t st, const PersistArgs& P);
// emx_slab.hip: the fused dense half-step at padded ndim 80 ... 128 with the proposals in registers and a 32-column LDS slab
hipError_t launch_slab_dense(int dpb, int move, dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a);
size_t slab_lds_bytes(int Dp, int waves);
// emx_podd.hip: k_persist for odd ndim up to 63 (one coordinate per lane and chunk)
hipError_t launch_persist_dense_odd(int dpb, int move, int local, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P);
hipError_t persist_dense_odd_occupancy(int dpb, int move, int threads, size_t lds, int* per_cu);
// emx_pslab.hip: its persistent form (k_persist_slab: device-wide and one-XCD; the stretch and DE moves) -- same LDS layout
// (odd: an odd ndim, 65 ... 127 -- 8-byte-granular row accesses in the same register layout)
hipError_t launch_persist_slab(int dpb, int move, int local, int odd, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P);
hipError_t persist_slab_occupancy(int dpb, int move, int odd, int threads, size_t lds, int* per_cu);

// Wide dense Gaussian targets (padded ndim > 112, emx_wide.hip): log-probs of a block of rows, and the decision + commit
// of a half-step whose proposals sit in qout / fout.
struct WideLpArgs {
    const double* rows;          // row of slot t: rows + (order ? order[pos0 + t] : t) * D
    const int32_t* order;
    const double* img;           // the target image of emx_set_target: L in B-fragment order, then the padded mean
    double* out;                 // out[t], or out[order[pos0 + t]] with `scatter`
    uint32_t* status;
    const int32_t* t_hi_dev;     // device-side slot count (block-ownership exchanges), or nullptr
    int32_t D, Dp, pos0, t_lo, t_hi, scatter, check_bad;
    int32_t single_role;         // 1: never the role-split kernel (tuning "dense_wide" = 2: parity tests of the two kernels)
};
struct WideCommitArgs {
    double* X;
    double* lp;
    uint8_t* acc;
    uint32_t* acc_count;
    double* chain;
    double* chain_lp;
    double* sendbuf;
    const double* qout;
    const double* fout;
    const double* newlp;
    const int32_t* order;
    const double* logu;
    const int32_t* t_hi_dev;
    uint32_t* status;            // sticky status flags: a NaN log-prob (ensemble.py:550-551) is reported and the proposal rejected
    double* declp;               // replay exchange: decision of slot t at declp[t - t_lo] (new log-prob, NaN when rejected), or nullptr
    int32_t D, pos0, t_lo, t_hi;
};
hipError_t launch_wide_lp(const WideLpArgs& a, int nrows_bound, int num_cu, hipStream_t st);
hipError_t launch_wide_commit(const WideCommitArgs& a, int nrows_bound, int num_cu, hipStream_t st);

}  // namespace emx
