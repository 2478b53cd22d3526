// k_persist_slab: the slab form of the fused dense-Gaussian half-step (emx_slab.hip: padded ndim 80 ... 128, the tile's proposals in
// registers, one 32-column LDS slab of R = Q - mu at a time, eight waves a CU) as a PERSISTENT kernel -- up to 32 half-steps per
// launch, a barrier where the kernel boundaries were (red_blue.py:85,104: split k + 1 sees every update of split k), in the two forms
// k_persist has: device-wide (agent-scope accesses, the arrival-counter barrier) and one-XCD (every working group on one XCD, plain
// stores, sc1 loads answered by that XCD's L2, the flag barrier) -- round 6, verdict items 5 and 6.
//
// What a launch per half-step paid and this form does not: the 73 KB image of the target staged into LDS by every workgroup of every
// half-step (256 x 73 KB = 18.7 MB of L2 reads in front of the first row load), the launch gap, and -- for ensembles between "fits
// one workgroup" and ~16 384 walkers -- two round trips to the memory side per half-step where one XCD's L2 will do.
//
// Same arithmetic, operation by operation, as k_halfstep_slab (make_proposal; Y = R L by v_mfma_f64_16x16x4_f64, k-steps ascending
// inside each column block; qf as the sum of squares in the same order; decision red_blue.py:99-100; commit move.py:33-34): the same
// bits (tests/test_gpu_persist_slab.py).  Every wave owns exactly one 16-walker tile of every half-step (the host's grid rule, as for
// k_persist), the stretch and DE moves (two splits).
#include "emx_launch.hpp"

namespace emx {

constexpr int PSLAB_RT = 34;         // slab row stride in doubles (emx_slab.hip: SLAB_RT)

// Row accesses of the (G = 16, V = 2, CH = 4) layout for an ODD ndim (65 ... 127): a row of an odd number of doubles is 8-byte aligned
// only, so the two coordinates of a lane and chunk travel as two 8-byte accesses instead of one of 16.  The values in the registers --
// and everything made from them -- are those of the even layout: the kernel body is one.  (The launch-per-half-step path runs such
// an ndim in rows of 32 lanes, one coordinate a lane: the same proposals element by element, the same tile in LDS, the same bits.)
typedef unsigned int pslab_u2 __attribute__((ext_vector_type(2)));
template <bool ODD, int CPOL>
__device__ __forceinline__ void pslab_load_row(Row<16, 2, 4>& r, __amdgpu_buffer_rsrc_t rsrc, int row, int D, int gl) {
    if constexpr (!ODD) {
        load_row_agent<16, 2, 4, CPOL>(r, rsrc, row, D, gl);
    } else {
        const int base = row * D;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = (c * 16 + gl) * 2;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (d + v < D) {
                    const pslab_u2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (base + d + v) * 8, 0, CPOL);
                    double t;
                    __builtin_memcpy(&t, &w, 8);
                    r.x[c][v] = t;
                } else {
                    r.x[c][v] = 0.0;
                }
            }
        }
    }
}
template <bool ODD, int CPOL>
__device__ __forceinline__ void pslab_store_row(const Row<16, 2, 4>& r, __amdgpu_buffer_rsrc_t rsrc, int row, int D, int gl) {
    if constexpr (!ODD) {
        store_row_agent<16, 2, 4, CPOL>(r, rsrc, row, D, gl);
    } else {
        const int base = row * D;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = (c * 16 + gl) * 2;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (d + v < D) {
                    pslab_u2 w;
                    const double t = r.x[c][v];
                    __builtin_memcpy(&w, &t, 8);
                    __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, (base + d + v) * 8, 0, CPOL);
                }
            }
        }
    }
}
template <bool ODD>
__device__ __forceinline__ void pslab_store_stream(const Row<16, 2, 4>& r, double* __restrict__ base, int D, int gl) {
    if constexpr (!ODD) {
        store_row_stream<16, 2, 4>(r, base, D, gl);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = (c * 16 + gl) * 2;
#pragma unroll
            for (int v = 0; v < 2; ++v)
                if (d + v < D) __builtin_nontemporal_store(r.x[c][v], base + d + v);
        }
    }
}

template <int DPB, int MOVE, bool LOCAL, bool ODD = false>
static __global__ __launch_bounds__(512) void k_persist_slab(const PersistArgs P) {
    static_assert(MOVE == MOVE_STRETCH || MOVE == MOVE_DE, "the slab form takes the stretch and DE moves");
    if (LOCAL && (blockIdx.x & 7u) != 0u) return;       // (k_persist: of an eight times larger grid every eighth workgroup works -- one XCD)
    const unsigned bid = LOCAL ? blockIdx.x >> 3 : blockIdx.x, ngroups = LOCAL ? gridDim.x >> 3 : gridDim.x;
    constexpr int CPOL = EMX_CPOL_SC1;                       // loads: agent scope in both forms (answered by the L2 in the one-XCD form)
    constexpr int CPOL_ST = LOCAL ? 0 : EMX_CPOL_SC1;       // stores: plain in the one-XCD form (in the L2 when acknowledged)
    constexpr int G = 16, V = 2, CH = 4;
    constexpr bool DE = MOVE == MOVE_DE;
    constexpr int WPW = 64 / G, PPT = 16 / WPW;        // 4 walkers a pass, 4 passes a tile
    constexpr int Dp = DPB * 16, KK = Dp / 4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const HalfStepArgs& A = P.base;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, sub = lane / G, gl = lane % G;
    const int D = A.D;
    double* Sfrag = smem;
    double* muS = smem + dense_img_doubles(Dp);
    double* tile = muS + Dp + (size_t)wib * (16 * PSLAB_RT + 32);
    double* facS = tile + 16 * PSLAB_RT;
    {   // the image of the target (emx_set_target): ONCE per launch, every load in flight before the first store (emx_slab.hip)
        constexpr int IMG2 = (dense_img_doubles(Dp) + Dp) / 2;
        constexpr int NG = 10;
        const double2* img = reinterpret_cast<const double2*>(A.tp1);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int base = threadIdx.x; base < IMG2; base += NG * blockDim.x) {
            double2 stg[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) stg[j] = img[min(base + j * (int)blockDim.x, IMG2 - 1)];
#pragma unroll
            for (int j = 0; j < NG; ++j) dst[min(base + j * (int)blockDim.x, IMG2 - 1)] = stg[j];
        }
    }
    // Skewed start (emx_slab.hip; HalfStepArgs::ablate bit 8, tuning "slab_skew"): every wave of the chip leaves the barrier at the
    // same moment, so the row loads of a half-step (2 KB per update at ndim 128) and its 144 MFMAs per tile would run back to back
    // chip-wide.  The second wave of every SIMD (wib >= 4) issues its rows only when its sibling's have arrived -- a word in the
    // sibling's LDS region carrying the half-step's number -- so one wave's MFMA chain covers the other's loads.
    const bool skew_on = (A.ablate & 256) != 0 && blockDim.x == 512;
    int* sigw = reinterpret_cast<int*>(muS + Dp + (size_t)(wib & 3) * (16 * PSLAB_RT + 32) + 16 * PSLAB_RT + 16);     // (facS uses 16 of the 32 spare doubles)
    if (skew_on && wib < 4 && lane == 0) *sigw = 0;
    if (!persist_handshake<LOCAL>(P)) return;           // (also the workgroup barrier behind the image)
    const int wave = (int)bid * (blockDim.x >> 6) + wib;
    const int t0 = wave * 16;                           // this wave's slots of every split
    const __amdgpu_buffer_rsrc_t Xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.X, 0, A.N * D * 8, 0x00020000);
    const int myrow = (lane >> 4) + 4 * (lane & 3);    // decision lanes: (lane & 15) < 4 decide tile row myrow
    const bool mine = (lane & 15) < 4;
    // plan entries of the first half-step.  The stretch move asks for the later ones one half-step ahead, under the MFMA chain; the
    // DE move's three rows a walker leave no registers for that (13-15 spilled at padded ndim 128): it reads them behind the barrier
    constexpr bool PRE = !DE;
    int wi[PPT], ja[PPT], jb[DE ? PPT : 1];
    double s0v[PPT];
    {
        const PersistCols I(P.it[0], (size_t)P.base.N);
        const int pbase = I.pos0 + t0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int pos = pbase + k * WPW + sub;
            wi[k] = I.order[pos];
            ja[k] = I.p0[pos];
            if constexpr (DE) jb[k] = I.p1[pos];
            s0v[k] = I.s0[pos];
        }
    }
    for (int n = 0; n < P.niter; ++n) {
        const PersistCols I(P.it[n], (size_t)P.base.N);
        const int mypos = I.pos0 + t0 + myrow;
        if (!PRE && n > 0) {
            const int pbase = I.pos0 + t0;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int pos = pbase + k * WPW + sub;
                wi[k] = I.order[pos];
                ja[k] = I.p0[pos];
                if constexpr (DE) jb[k] = I.p1[pos];
                s0v[k] = I.s0[pos];
            }
        }
        // -------- every row of the tile: own rows and the partners the previous half-step may have moved --------
        if (skew_on && wib >= 4)
            while (__hip_atomic_load(sigw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != n + 1) __builtin_amdgcn_s_sleep(2);
        Row<G, V, CH> xi[PPT], xa[PPT], xb[DE ? PPT : 1];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            pslab_load_row<ODD, CPOL>(xi[k], Xr, wi[k], D, gl);
            pslab_load_row<ODD, CPOL>(xa[k], Xr, ja[k], D, gl);
            if constexpr (DE) pslab_load_row<ODD, CPOL>(xb[k], Xr, jb[k], D, gl);
        }
        if (skew_on && wib < 4) {
            // most of this wave's rows are here: the sibling may load now (bits 9-10: all / three quarters / half / a quarter of them)
            constexpr int NL = (DE ? 3 : 2) * PPT * CH;        // row loads of a tile (16 bytes a lane each)
            switch ((A.ablate >> 9) & 3) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL / 4) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL / 2) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NL / 4) : "memory"); break;
            }
            if (lane == 0) __hip_atomic_store(sigw, n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // -------- proposals: kept in registers (R = Q - mu goes to LDS slab by slab below) --------
        Row<G, V, CH> qk[PPT];
        bool ok[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            double factor = 0.0;
            make_proposal<G, V, CH, MOVE>(xi[k], xa[k], xb[DE ? k : 0], xb[0], s0v[k], A.gammas, D, gl, qk[k], factor, ja[k]);
            bool bl = false;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) bl |= !(fabs(qk[k].x[c][v]) <= 1.79769313486231570815e308);
            const bool badq = group_any<G>(bl, sub);      // non-finite proposal -> sticky error (ensemble.py:476-479), rejected
            if (badq && gl == 0) raise_status(A.status, ST_BAD_COORD);
            ok[k] = !badq;
            if (gl == 0) facS[k * WPW + sub] = badq ? -__builtin_inf() : 0.0;       // (+ my_fac below: exact, x + 0 = x)
            // stored step: the current row goes out now (fire and forget); an accepted proposal overwrites it after the decision
            if (I.chain) pslab_store_stream<ODD>(xi[k], I.chain + (size_t)wi[k] * D, D, gl);
        }
        // the deciding lanes' own entries, and the next half-step's plan entries: asked for here, under the MFMA chain
        const int my_i = I.order[mypos];
        const double my_logu = I.logu[mypos];
        const double my_fac = I.fac[mypos];             // (stretch: (D - 1) ln z, DE: 0)
        const double my_lpo = load_agent(A.lp + my_i);
        const bool more = n + 1 < P.niter;
        int wi_n[PRE ? PPT : 1], ja_n[PRE ? PPT : 1];
        double s0_n[PRE ? PPT : 1];
        if constexpr (PRE) {
            const PersistCols J(P.it[more ? n + 1 : n], (size_t)P.base.N);
            const int pbase = J.pos0 + t0;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int pos = pbase + k * WPW + sub;
                wi_n[k] = J.order[pos];
                ja_n[k] = J.p0[pos];
                s0_n[k] = J.s0[pos];
            }
        }
        // -------- Y = R L, slab by slab; column block nb takes the k-steps kk >= 4 nb (L is lower triangular) --------
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 accv[DPB];
#pragma unroll
        for (int nb = 0; nb < DPB; ++nb) accv[nb] = d4{0.0, 0.0, 0.0, 0.0};
        const int am = lane & 15, ak = lane >> 4;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (32 * c < Dp) {
                EMX_WAVE_SYNC();                        // every lane has read the slab before
                const double2 muc = *reinterpret_cast<const double2*>(muS + 32 * c + 2 * gl);
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    double2 r;
                    r.x = ok[k] ? qk[k].x[c][0] - muc.x : 0.0;            // dead row: zero residual
                    r.y = ok[k] ? qk[k].x[c][1] - muc.y : 0.0;
                    *reinterpret_cast<double2*>(tile + (k * WPW + sub) * PSLAB_RT + gl * 2) = r;
                }
                EMX_WAVE_SYNC();                        // this wave's slab is visible to all of its lanes
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int kk = 8 * c + k8;
                    if (kk < KK) {
                        const double a = tile[am * PSLAB_RT + 4 * k8 + ak];     // A[i = lane & 15][k = lane >> 4]
#pragma unroll
                        for (int nb = 0; nb < DPB; ++nb)
                            if (4 * nb <= kk)
                                accv[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                    a, Sfrag[(dense_block(DPB, nb, kk >> 2) * 4 + (kk & 3)) * 64 + lane], accv[nb], 0, 0, 0);
                    }
                }
            }
        }
        double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int nb = 0; nb < DPB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[r] = fma(accv[nb][r], accv[nb][r], part[r]);       // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 r
        const double my_qf = row16_sum4(part[0], part[1], part[2], part[3], lane);
        // -------- decisions (red_blue.py:99-100) and commit (move.py:33-34) --------
        bool acc = false;
        if (mine) {
            const double lpn = -0.5 * my_qf;
            if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);
            const double lnpdiff = (facS[myrow] + my_fac) + lpn - my_lpo;
            acc = lnpdiff > my_logu;
            store_scope<LOCAL>(A.acc + my_i, (uint8_t)(acc ? 1 : 0));
            if (acc) store_scope<LOCAL>(A.lp + my_i, lpn);
            if (I.chain_lp) {
                I.chain_lp[my_i] = acc ? lpn : my_lpo;
                if (acc) store_scope<LOCAL>(A.acc_count + my_i, load_agent(A.acc_count + my_i) + 1u);
            }
        }
        const unsigned long long am64 = __ballot(acc);           // bit (row & 3) * 16 + (row >> 2) <-> tile row
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp) {
            const int row = pp * WPW + sub;
            const bool ac = (am64 >> ((row & 3) * 16 + (row >> 2))) & 1ull;
            if (ac) {
                pslab_store_row<ODD, CPOL_ST>(qk[pp], Xr, wi[pp], D, gl);
                if (I.chain) pslab_store_stream<ODD>(qk[pp], I.chain + (size_t)wi[pp] * D, D, gl);
            }
        }
        EMX_WAVE_SYNC();                                // (facS and the slab are rewritten by the next half-step)
        if (!more) break;
        if constexpr (LOCAL)
            persist_barrier_local(P, P.lepoch0 + (unsigned)n + 1u, bid, ngroups);
        else
            persist_barrier(P, P.epoch0 + (unsigned)n + 2u);       // (+ 1: the handshake was this launch's first barrier)
        if constexpr (PRE) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                wi[k] = wi_n[k];
                ja[k] = ja_n[k];
                s0v[k] = s0_n[k];
            }
        }
    }
}

template <int DPB, int MOVE, bool LOCAL, bool ODD>
static hipError_t launch_pslab(dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
    auto kern = k_persist_slab<DPB, MOVE, LOCAL, ODD>;
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

template <int DPB, bool ODD>
static hipError_t launch_pslab_pick(int move, int local, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
    if constexpr (ODD) {
        if (move == MOVE_DE) return hipErrorInvalidValue;       // (the DE move's third row a walker: 39 spilled registers at padded ndim 128 -- not instantiated; the host's rule keeps it off)
    } else {
        if (move == MOVE_DE) return local ? launch_pslab<DPB, MOVE_DE, true, ODD>(grid, block, lds, st, P) : launch_pslab<DPB, MOVE_DE, false, ODD>(grid, block, lds, st, P);
    }
    return local ? launch_pslab<DPB, MOVE_STRETCH, true, ODD>(grid, block, lds, st, P) : launch_pslab<DPB, MOVE_STRETCH, false, ODD>(grid, block, lds, st, P);
}

// odd: ndim is odd (65 ... 127): the 8-byte-granular row accesses
hipError_t launch_persist_slab(int dpb, int move, int local, int odd, dim3 grid, dim3 block, size_t lds, hipStream_t st, const PersistArgs& P) {
#define EMX_CASE(b) \
    if (dpb == b) return odd ? launch_pslab_pick<b, true>(move, local, grid, block, lds, st, P) : launch_pslab_pick<b, false>(move, local, grid, block, lds, st, P);
    EMX_CASE(5) EMX_CASE(6) EMX_CASE(7) EMX_CASE(8)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

// workgroups of the device-wide instantiation a CU holds at once (block size, dynamic LDS): the co-residency check of persist_grid_fits
hipError_t persist_slab_occupancy(int dpb, int move, int odd, int threads, size_t lds, int* per_cu) {
#define EMX_OCC(kern_)                                                                                                          \
    {                                                                                                                           \
        auto kern = kern_;                                                                                                      \
        if (lds > 48 * 1024) {                                                                                                  \
            const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            if (e != hipSuccess) return e;                                                                                      \
        }                                                                                                                       \
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, threads, lds);                                        \
    }
#define EMX_CASE(b)                                                                                      \
    if (dpb == b) {                                                                                      \
        if (move == MOVE_DE) {                                                                           \
            if (odd) return hipErrorInvalidValue;                                                        \
            EMX_OCC((k_persist_slab<b, MOVE_DE, false, false>))                                          \
        }                                                                                                \
        if (odd) EMX_OCC((k_persist_slab<b, MOVE_STRETCH, false, true>))                                 \
        EMX_OCC((k_persist_slab<b, MOVE_STRETCH, false, false>))                                         \
    }
    EMX_CASE(5) EMX_CASE(6) EMX_CASE(7) EMX_CASE(8)
#undef EMX_CASE
#undef EMX_OCC
    return hipErrorInvalidValue;
}

}  // namespace emx
