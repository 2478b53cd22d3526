// The persistent half-step kernel for the ELEMENT-WISE targets (isotropic / diagonal Gaussian, Rosenbrock, box), one-XCD form only.
//
// Between "fits one workgroup's LDS" (k_small_run) and ~10^4 walkers a step of an element-wise target was two launches of pure
// latency (~10 us).  k_persist (emx_kernels.hpp) removes the launches for the dense Gaussian; its one-XCD form (round 4) also removes
// the memory-side round trips: every workgroup of the grid runs on ONE XCD, plain stores are in that XCD's L2 when acknowledged, sc1
// loads of lines it wrote are answered by it, and a flag barrier costs 0.32 us (persist_barrier_local).  This kernel is that form for
// the targets that need no LDS image and no MFMA: per half-step a wave loads its 16 walkers' rows and their partners' (sc1), makes
// the proposals (make_proposal: stretch.py:26-33, de.py:40-64, de_snooker.py:31-46), evaluates the target from registers
// (eval_valu_target), decides (red_blue.py:96-101) and commits (move.py:29-45), then meets the other workgroups at the flag barrier
// -- up to 32 half-steps a launch.  Same device functions in the same order as k_halfstep, hence the same bits
// (tests/test_gpu_persist.py).  Ensembles of up to 8 192 walkers, Philox plans, one replica; row layouts of 8 lanes per walker
// (ndim <= 64 even, <= 32 odd) and of 16 lanes with one coordinate a lane (odd ndim 33 ... 63, round 6); everything else keeps the
// per-half-step launches.
#include "emx_launch.hpp"

namespace emx {

// a row with one coordinate per lane and chunk (odd ndim), agent scope
template <int G, int V, int CH, int CPOL>
__device__ __forceinline__ void pv_load_row(Row<G, V, CH>& r, __amdgpu_buffer_rsrc_t rsrc, int row, int D, int gl) {
    if constexpr (V == 2) {
        load_row_agent<G, V, CH, CPOL>(r, rsrc, row, D, gl);
    } else {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const int base = row * D;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int d = c * G + gl;
            if (d < D) {
                const u2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (base + d) * 8, 0, CPOL);
                double t;
                __builtin_memcpy(&t, &w, 8);
                r.x[c][0] = t;
            } else {
                r.x[c][0] = 0.0;
            }
        }
    }
}
template <int G, int V, int CH, int CPOL>
__device__ __forceinline__ void pv_store_row(const Row<G, V, CH>& r, __amdgpu_buffer_rsrc_t rsrc, int row, int D, int gl) {
    if constexpr (V == 2) {
        store_row_agent<G, V, CH, CPOL>(r, rsrc, row, D, gl);
    } else {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const int base = row * D;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int d = c * G + gl;
            if (d < D) {
                u2 w;
                const double t = r.x[c][0];
                __builtin_memcpy(&w, &t, 8);
                __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, (base + d) * 8, 0, CPOL);
            }
        }
    }
}

template <int G, int V, int CH, int MOVE, bool LOCAL = true>
static __global__ __launch_bounds__(512) void k_persist_valu(const PersistArgs P) {
    static_assert(MOVE == MOVE_STRETCH || MOVE == MOVE_DE || MOVE == MOVE_SNOOKER, "the red / blue moves");
    constexpr bool DE = MOVE == MOVE_DE || MOVE == MOVE_SNOOKER;
    constexpr bool SN = MOVE == MOVE_SNOOKER;
    constexpr int WPW = 64 / G, PF = 16 / WPW;          // a wave owns 16 plan slots of every split (as in k_persist)
    // LOCAL: the one-XCD form -- every eighth workgroup of an eight times larger grid, plain stores, that XCD's flag barrier.
    // !LOCAL: the device-wide form (ensembles beyond 8 192 walkers: about a workgroup per CU) -- agent-scope stores and the device-wide
    // barrier, as k_persist; in exact mode it takes the fetched plans of sixteen steps per launch where the per-half-step path pays an
    // upload and five API calls per step.
    if (LOCAL && (blockIdx.x & 7u) != 0u) return;
    const unsigned bid = LOCAL ? blockIdx.x >> 3 : blockIdx.x, ngroups = LOCAL ? gridDim.x >> 3 : gridDim.x;
    (void)ngroups;
    const HalfStepArgs& A = P.base;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, sub = lane / G, gl = lane % G;
    const int D = A.D;
    Row<G, V, CH> mu, iv;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) mu.x[c][v] = iv.x[c][v] = 0.0;
    if (A.target == TGT_DIAG) {
        load_row<G, V, CH>(mu, A.tp0, D, gl);
        load_row<G, V, CH>(iv, A.tp1, D, gl);
    }
    if (!persist_handshake<LOCAL>(P)) return;
    const int wave = (int)bid * (blockDim.x >> 6) + wib;
    const int t0 = wave * 16;
    const __amdgpu_buffer_rsrc_t Xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.X, 0, A.N * D * 8, 0x00020000);
    constexpr int LD = EMX_CPOL_SC1, ST = LOCAL ? 0 : EMX_CPOL_SC1;      // loads agent-scope (LOCAL: answered by this XCD's L2); stores plain / agent-scope
    int wi[PF], ja[PF], jb[DE ? PF : 1], jc[SN ? PF : 1];
    double s0v[PF], facv[PF], loguv[PF];
    auto plan_of = [&](const PersistCols& I, int (&w)[PF], int (&a)[PF], int (&b)[DE ? PF : 1], int (&c3)[SN ? PF : 1], double (&s)[PF],
                       double (&f)[PF], double (&lu)[PF]) {
        const int pbase = I.pos0 + t0;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int pos = pbase + k * WPW + sub;
            w[k] = I.order[pos];
            a[k] = I.p0[pos];
            if constexpr (DE) b[k] = I.p1[pos];
            if constexpr (SN) c3[k] = I.p2[pos];
            s[k] = SN ? 0.0 : I.s0[pos];
            f[k] = I.fac[pos];
            lu[k] = I.logu[pos];
        }
    };
    plan_of(PersistCols(P.it[0], (size_t)P.base.N), wi, ja, jb, jc, s0v, facv, loguv);
    for (int n = 0; n < P.niter; ++n) {
        const PersistCols I(P.it[n], (size_t)P.base.N);
        // -------- own rows, partner rows, current log-probs: one round trip behind the barrier --------
        Row<G, V, CH> xi[PF], xa[PF], xb[DE ? PF : 1], xc[SN ? PF : 1];
        double lpo[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            pv_load_row<G, V, CH, LD>(xi[k], Xr, wi[k], D, gl);
            pv_load_row<G, V, CH, LD>(xa[k], Xr, ja[k], D, gl);
            if constexpr (DE) pv_load_row<G, V, CH, LD>(xb[k], Xr, jb[k], D, gl);
            if constexpr (SN) pv_load_row<G, V, CH, LD>(xc[k], Xr, jc[k], D, gl);
            lpo[k] = load_agent(A.lp + wi[k]);
        }
        // -------- plan entries of the next half-step (written by the plan kernel before this launch) --------
        const bool more = n + 1 < P.niter;
        int wi_n[PF], ja_n[PF], jb_n[DE ? PF : 1], jc_n[SN ? PF : 1];
        double s0_n[PF], fac_n[PF], logu_n[PF];
        plan_of(PersistCols(P.it[more ? n + 1 : n], (size_t)P.base.N), wi_n, ja_n, jb_n, jc_n, s0_n, fac_n, logu_n);
        // -------- proposals, target, decision, commit --------
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            double factor = facv[k];
            Row<G, V, CH> q;
            make_proposal<G, V, CH, MOVE>(xi[k], xa[k], xb[DE ? k : 0], xc[SN ? k : 0], s0v[k], A.gammas, D, gl, q, factor, ja[k]);
            bool bl = false;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) bl |= !(fabs(q.x[c][v]) <= 1.79769313486231570815e308);
            const bool badq = group_any<G>(bl, sub);              // ensemble.py:476-479
            if (badq && gl == 0) raise_status(A.status, ST_BAD_COORD);
            const double lp_new = eval_valu_target<G, V, CH>(q, mu, iv, A.tp0, A.tp1, A.target, A.tscale, D, gl, lane);
            if (gl == 0 && (lp_new != lp_new)) raise_status(A.status, ST_NAN_LOGP);          // ensemble.py:550-551
            const double lnpdiff = factor + lp_new - lpo[k];         // red_blue.py:99
            const bool accept = !badq && (lnpdiff > loguv[k]);       // red_blue.py:100
            const int i = wi[k];
            if (accept) {
                pv_store_row<G, V, CH, ST>(q, Xr, i, D, gl);        // move.py:33
                if (gl == 0) store_scope<LOCAL>(A.lp + i, lp_new);   // move.py:34
            }
            if (gl == 0) {
                store_scope<LOCAL>(A.acc + i, (uint8_t)(accept ? 1 : 0));
                if (I.chain_lp) {
                    I.chain_lp[i] = accept ? lp_new : lpo[k];
                    if (accept) store_scope<LOCAL>(A.acc_count + i, load_agent(A.acc_count + i) + 1u);
                }
            }
            if (I.chain) store_row_stream<G, V, CH>(accept ? q : xi[k], I.chain + (size_t)i * D, D, gl);
        }
        if (!more) break;
        if constexpr (LOCAL)
            persist_barrier_local(P, P.lepoch0 + (unsigned)n + 1u, bid, ngroups);
        else
            persist_barrier(P, P.epoch0 + (unsigned)n + 2u);       // (+ 1: the handshake was this launch's first barrier)
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            wi[k] = wi_n[k];
            ja[k] = ja_n[k];
            if constexpr (DE) jb[k] = jb_n[k];
            if constexpr (SN) jc[k] = jc_n[k];
            s0v[k] = s0_n[k];
            facv[k] = fac_n[k];
            loguv[k] = logu_n[k];
        }
    }
}

template <int G, int V, int CH, bool LOCAL>
static hipError_t launch_pv(int move, dim3 grid, dim3 block, hipStream_t st, const PersistArgs& P) {
    if (move == MOVE_DE)
        hipLaunchKernelGGL((k_persist_valu<G, V, CH, MOVE_DE, LOCAL>), grid, block, 0, st, P);
    else if (move == MOVE_SNOOKER)
        hipLaunchKernelGGL((k_persist_valu<G, V, CH, MOVE_SNOOKER, LOCAL>), grid, block, 0, st, P);
    else
        hipLaunchKernelGGL((k_persist_valu<G, V, CH, MOVE_STRETCH, LOCAL>), grid, block, 0, st, P);
    return hipGetLastError();
}

// row layouts of 8 lanes per walker: V = 2 (even ndim 10 ... 64) / V = 1 (odd ndim 5 ... 32), CH = 1, 2, 4; of 4 lanes per walker:
// ndim <= 4 and even ndim <= 8 (pick_shape: the dimensions of most real-world fits); local: the one-XCD form (grid already x 8)
hipError_t launch_persist_valu(int G, int V, int CH, int move, int local, dim3 grid, dim3 block, hipStream_t st, const PersistArgs& P) {
#define EMX_CASE(g, v, c)                                                                  \
    if (G == g && V == v && CH == c)                                                       \
        return local ? launch_pv<g, v, c, true>(move, grid, block, st, P) : launch_pv<g, v, c, false>(move, grid, block, st, P);
    EMX_CASE(4, 2, 1) EMX_CASE(4, 1, 1)
    EMX_CASE(8, 2, 1) EMX_CASE(8, 2, 2) EMX_CASE(8, 2, 4) EMX_CASE(8, 1, 1) EMX_CASE(8, 1, 2) EMX_CASE(8, 1, 4)
    EMX_CASE(16, 1, 4)        // round 6: odd ndim 33 ... 63 (rows of 16 lanes, one coordinate a lane and chunk: 192-237 VGPRs.  Even ndim
                              // 66 ... 128 would be <16, 2, 4>: 33-242 spilled registers -- those shapes keep the per-half-step launches)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

}  // namespace emx
