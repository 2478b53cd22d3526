// k_persist_mix: k_persist (emx_kernels.hpp) for a schedule of DEMove (two splits) and DESnookerMove (four) steps in ONE launch.
// A copy of that kernel rather than another instantiation of it: what the mixture needs -- the move of a half-step as a runtime,
// wave-uniform value, waves without a tile in a half-step -- changes the register allocation of the headline instantiations when it
// lives in the same template (k_persist<8, 2, 4, 4, MOVE_STRETCH>: 240 -> 250 VGPRs, C2 21.1 -> 21.6 us/step).  Same device functions
// (make_proposal, the MFMA chain, the decision and commit code), same order of operations: the same bits as the per-half-step path
// (tests/test_gpu_persist.py).  Included by emx_pmix.hip only.
#pragma once
#include "emx_kernels.hpp"

namespace emx {

template <int G, int V, int CH, int DPB, bool LOCAL = false>
static __global__ __launch_bounds__(512) void k_persist_mix(const PersistArgs P) {
    constexpr int MOVE = MOVE_MIX;
    static_assert(MOVE == MOVE_STRETCH || MOVE == MOVE_DE || MOVE == MOVE_SNOOKER || MOVE == MOVE_MIX, "the red / blue moves");
    // LOCAL: the one-XCD form for small ensembles.  The dispatcher deals workgroups to the eight XCDs in turn (workgroup i -> XCD
    // i mod 8: tools/exp/cu_mask_probe.hip), so of an eight times larger grid only every eighth workgroup works -- all of them on one
    // XCD, whose L2 then keeps the walker state coherent without agent-scope accesses (3.0 + 2.5 us of partner round trip and barrier
    // per half-step in the device-wide form, profiles/r04/persist_phase_c2.txt).  The handshake checks that they really share one.
    if (LOCAL && (blockIdx.x & 7u) != 0u) return;
    const unsigned bid = LOCAL ? blockIdx.x >> 3 : blockIdx.x, ngroups = LOCAL ? gridDim.x >> 3 : gridDim.x;
    constexpr int CPOL = EMX_CPOL_SC1;                       // loads: agent scope in both forms (answered by the L2 in the one-XCD form)
    constexpr int CPOL_ST = LOCAL ? 0 : EMX_CPOL_SC1;       // stores: plain in the one-XCD form (in the L2 when acknowledged)
    // MOVE_MIX: a schedule of DEMove (two splits) and DESnookerMove (four) steps in ONE launch -- the move of a half-step is a field of
    // its PersistIter, the branches on it are wave-uniform; the grid is the DE move's (a wave per 16-walker tile of HALF the ensemble),
    // and in a snooker half-step (a quarter) every other wave works (PersistIter::shift) -- the others only fetch what their next
    // half-step needs.  A run of one move per launch left C4 with 5.9 half-steps a launch and ~10 us of launch gap + handshake + cold
    // first loads for each (profiles/r04/c4_mix.txt).
    constexpr bool MIX = MOVE == MOVE_MIX;
    constexpr bool DE = MOVE == MOVE_DE || MOVE == MOVE_SNOOKER || MIX;   // de.py:40-64: two partners, q = s + gamma (c[pair 1] - c[pair 0])
    constexpr bool SN = MOVE == MOVE_SNOOKER;          // de_snooker.py:31-46: three partners z, z1, z2 (one from each other set)
    constexpr bool SNA = SN || MIX;                    // (room for the third partner)
    constexpr bool DEFER = !DE;                        // chain rows one half-step later (the other forms have no registers to spare)
    constexpr int WPW = 64 / G;
    constexpr int PPT = 16 / WPW;
    constexpr int PF = PPT;        // every pass of the tile in one batch (k_halfstep's snooker form splits it: two dependent round trips)
    static_assert(EMX_OPT_RTILE && EMX_OPT_RED4, "the persistent kernel is the one-tile-per-batch form");
    constexpr int Dp = DPB * 16, KK = Dp / 4, RT = Dp + 2;
    static_assert(G * V * CH >= Dp, "row layout must cover the padded dimension");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const HalfStepArgs& A = P.base;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int sub = lane / G;
    const int gl = lane % G;
    const int D = A.D;
    double* Sfrag = smem;
    double* muS = smem + dense_img_doubles(Dp);
    double* tile = muS + Dp + (size_t)wib * (16 * RT + 32);
    double* qfS = tile + 16 * RT;
    double* facS = qfS + 16;
    {   // the image of the target (emx_set_target): once per launch
        constexpr int IMG2 = (dense_img_doubles(Dp) + Dp) / 2;
        const double2* img = reinterpret_cast<const double2*>(A.tp1);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int e = threadIdx.x; e < IMG2; e += blockDim.x) dst[e] = img[e];
    }
    Row<G, V, CH> mu;
    load_row<G, V, CH>(mu, A.tp0, D, gl);
    if (!persist_handshake<LOCAL>(P)) return;                   // (also the workgroup barrier behind the image load)
    const int wave = (int)bid * (blockDim.x >> 6) + wib;
    // this wave's slots of every split; MIX: of the current half-step's splits.  A half-step of 2^-shift as many tiles as waves is
    // taken by the FIRST wpb >> shift waves of every workgroup (one per SIMD: waves 0, 2, 4, 6 would share two SIMDs), or -- a
    // workgroup of fewer waves -- by the workgroups whose low `shift` bits are 0.
    const int wpb_ = (int)(blockDim.x >> 6);
    auto mix_tile = [&](int shift, bool& on) -> int {
        if (wpb_ >> shift) {
            on = wib < (wpb_ >> shift);
            return (int)bid * (wpb_ >> shift) + wib;
        }
        on = ((int)bid & ((1 << shift) - 1)) == 0 && wib == 0;      // (wpb_ == 1 here)
        return (int)bid >> shift;
    };
    bool act = true;                                                      // this wave has a tile in the current half-step
    int t0 = MIX ? mix_tile(P.it[0].shift, act) * 16 : wave * 16;
    bool sn = MIX ? P.it[0].kind == MOVE_SNOOKER : SN;                   // the current half-step's move
    // instrumented build only (tools/persist_phase_clock.py, -DEMX_OPT_STAMPS=1): where the first wave of every workgroup spends a
    // half-step -- ticks summed over the launch's half-steps: partner rows arrive | proposals + tile | MFMA + reductions |
    // decisions + commit issued | stores acknowledged | barrier
    unsigned long long pst[6] = {0, 0, 0, 0, 0, 0}, pt = 0;
    const bool prof = EMX_OPT_STAMPS && A.dbg && wib == 0;
    if (prof) {
        pt = __builtin_readcyclecounter();
        if (lane == 0) A.dbg[(size_t)bid * 16 + 11] = wall_clock64();
    }
#define EMX_PSTAMP(k_)                                                   \
    do {                                                                 \
        if (prof) {                                                      \
            const unsigned long long t_ = __builtin_readcyclecounter(); \
            pst[k_] += t_ - pt;                                          \
            pt = t_;                                                     \
        }                                                                \
    } while (0)
    const __amdgpu_buffer_rsrc_t Xr = __builtin_amdgcn_make_buffer_rsrc((void*)A.X, 0, A.N * D * 8, 0x00020000);
    const int myrow = (lane >> 4) + 4 * (lane & 3);             // decision lanes: (lane & 15) < 4 decide tile row myrow
    const bool mine = (lane & 15) < 4;

    int wi[PF], ja[PF], jb[DE ? PF : 1], jc[SNA ? PF : 1], my_i = 0;
    double s0v[PF], facv[PF], my_logu = 0.0, my_lpo = 0.0;
    Row<G, V, CH> xi[PF];
    if (act) {
        const PersistCols I(P.it[0], (size_t)P.base.N);
        const int pbase = I.pos0 + t0;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int pos = pbase + k * WPW + sub;
            wi[k] = I.order[pos];
            ja[k] = I.p0[pos];
            if constexpr (DE) jb[k] = I.p1[pos];
            if (SN || (MIX && sn)) jc[k] = I.p2[pos];
            s0v[k] = (SN || (MIX && sn)) ? 0.0 : I.s0[pos];
            facv[k] = I.fac[pos];
        }
        my_i = I.order[pbase + myrow];
        my_logu = I.logu[pbase + myrow];
#pragma unroll
        for (int k = 0; k < PF; ++k) load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi[k], D, gl);
        my_lpo = load_agent(A.lp + my_i);
    }
    // stored steps: the rows (and log-probs) of a half-step leave one half-step later
    Row<G, V, CH> crow[DEFER ? PF : 1];
    int cwi[PF], cmy_i = 0;
    double clp = 0.0;
    double *cchain = nullptr, *cchain_lp = nullptr;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        if (DEFER || k == 0) crow[DEFER ? k : 0] = xi[k];
        cwi[k] = 0;
    }
    for (int n = 0; n < P.niter; ++n) {
        const PersistCols I(P.it[n], (size_t)P.base.N);
        const bool more = n + 1 < P.niter;
        const PersistCols J(P.it[more ? n + 1 : n], (size_t)P.base.N);
        const bool pre = more && J.split != 0;                   // its own walkers are this half-step's complement
        const unsigned stamp = P.epoch0 + (unsigned)n + 1u;      // of this half-step (never 0 before the counters wrap)
        int wi_n[PF], ja_n[PF], jb_n[DE ? PF : 1], jc_n[SNA ? PF : 1], my_i_n = 0;
        double s0_n[PF], fac_n[PF], my_logu_n = 0.0, my_lpo_n = 0.0;
        bool act_n = true;
        const int t0_n = MIX ? mix_tile(J.shift, act_n) * 16 : t0;
        if (MIX && !more) act_n = false;
        const bool sn_n = MIX ? J.kind == MOVE_SNOOKER : SN;
        // what the NEXT half-step of this wave needs: its plan entries (written by the plan kernel before this launch) ...
        auto next_entries = [&]() {
            if (!act_n) return;
            const int pbase = J.pos0 + t0_n;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int pos = pbase + k * WPW + sub;
                wi_n[k] = J.order[pos];
                ja_n[k] = J.p0[pos];
                if constexpr (DE) jb_n[k] = J.p1[pos];
                if (SN || (MIX && sn_n)) jc_n[k] = J.p2[pos];
                s0_n[k] = (SN || (MIX && sn_n)) ? 0.0 : J.s0[pos];
                fac_n[k] = J.fac[pos];
            }
            my_i_n = J.order[pbase + myrow];
            my_logu_n = J.logu[pbase + myrow];
        };
        // ... and its own rows: in flight during the MFMA phase (speculative unless `pre`)
        auto next_rows = [&]() {
            if (!(more && act_n)) return;
#pragma unroll
            for (int k = 0; k < PF; ++k) load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi_n[k], D, gl);
            my_lpo_n = load_agent(A.lp + my_i_n);
        };
        if (act) {
        // -------- partner rows: the walkers the previous half-step updated --------
        Row<G, V, CH> xa[PF], xb[DE ? PF : 1], xc[SNA ? PF : 1];
        persist_stagger_wait(P.stagger, wib);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            load_row_agent<G, V, CH, CPOL>(xa[k], Xr, ja[k], D, gl);
            if constexpr (DE) load_row_agent<G, V, CH, CPOL>(xb[k], Xr, jb[k], D, gl);
            if (SN || (MIX && sn)) load_row_agent<G, V, CH, CPOL>(xc[k], Xr, jc[k], D, gl);
        }
        next_entries();
        // -------- proposals -> the wave's LDS tile (R = Q - mu), kept in registers for the commit --------
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        EMX_PSTAMP(0);       // partner rows (and the next half-step's plan entries) have arrived
        Row<G, V, CH> qk[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int srow = k * WPW + sub;
            double factor = facv[k];
            Row<G, V, CH> q;
            if constexpr (MIX) {
                if (sn)
                    make_proposal<G, V, CH, MOVE_SNOOKER>(xi[k], xa[k], xb[k], xc[k], s0v[k], I.gammas, D, gl, q, factor, ja[k]);
                else
                    make_proposal<G, V, CH, MOVE_DE>(xi[k], xa[k], xb[k], xc[k], s0v[k], A.gammas, D, gl, q, factor, ja[k]);
            } else {
                make_proposal<G, V, CH, MOVE>(xi[k], xa[k], xb[DE ? k : 0], xc[SN ? k : 0], s0v[k], A.gammas, D, gl, q, factor, ja[k]);
            }
            bool bl = false;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) bl |= !(fabs(q.x[c][v]) <= 1.79769313486231570815e308);
            const bool badq = group_any<G>(bl, sub);
            if (badq && gl == 0) raise_status(A.status, ST_BAD_COORD);
            const int trow = srow & 15;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int d = (c * G + gl) * V + v;
                    if (d < Dp) tile[trow * RT + d] = !badq ? q.x[c][v] - mu.x[c][v] : 0.0;
                }
            qk[k] = q;
            if (gl == 0) facS[trow] = badq ? -__builtin_inf() : factor;
        }
        // -------- stored steps: the rows of the half-step BEFORE go out now, next to the MFMA phase -- issued before its barrier
        //          their 17 MB would sit between the commits and the arrival (every store is acknowledged in order) --------
        if (DEFER && cchain) {
#pragma unroll
            for (int k = 0; k < PF; ++k) store_row_stream<G, V, CH>(crow[k], cchain + (size_t)cwi[k] * D, D, gl);
            if (mine) cchain_lp[cmy_i] = clp;
            cchain = nullptr;
        }
        if (I.chain) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                if constexpr (DEFER)
                    crow[k] = xi[k];          // an accepted proposal replaces it after the decision
                else
                    store_row_stream<G, V, CH>(xi[k], I.chain + (size_t)wi[k] * D, D, gl);      // fire and forget; overwritten on accept
            }
        }
        next_rows();
        EMX_WAVE_SYNC();
        EMX_PSTAMP(1);       // proposals made, tile written, chain rows and next own rows issued
        // -------- Y = R L by v_mfma_f64_16x16x4_f64, qf[w] = sum_n Y[w][n]^2 (as k_halfstep) --------
        double my_qf;
        {
            const int am = lane & 15, ak = lane >> 4;
            typedef double d4 __attribute__((ext_vector_type(4)));
            double afr[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) afr[kk] = tile[am * RT + 4 * kk + ak];
            double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int nb = 0; nb < DPB; ++nb) {
                d4 accv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 4 * nb; kk < KK; ++kk)
                    accv = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], Sfrag[(dense_block(DPB, nb, kk >> 2) * 4 + (kk & 3)) * 64 + lane], accv, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) part[r] = fma(accv[r], accv[r], part[r]);
            }
            my_qf = row16_sum4(part[0], part[1], part[2], part[3], lane);
        }
        if (prof) { asm volatile("s_nop 0" ::: "memory"); }
        EMX_PSTAMP(2);       // MFMA chain + reductions
        // -------- decisions (red_blue.py:99-100) and commit (move.py:33-34) --------
        bool acc = false;
        if (mine) {
            const double lpn = -0.5 * my_qf;
            if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);
            const double lnpdiff = facS[myrow] + lpn - my_lpo;
            acc = lnpdiff > my_logu;
            store_scope<LOCAL>(A.acc + my_i, (uint8_t)(acc ? 1 : 0));       // (a walker's mark is written by another XCD every step: write-through)
            if (acc) {
                store_scope<LOCAL>(A.lp + my_i, lpn);
                store_scope<LOCAL>(P.ver + my_i, stamp);
            }
            if (I.chain_lp) {
                if constexpr (DEFER)
                    clp = acc ? lpn : my_lpo;
                else
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
                store_row_agent<G, V, CH, CPOL_ST>(qk[pp], Xr, wi[pp], D, gl);
                if (I.chain) {
                    if constexpr (DEFER)
                        crow[pp] = qk[pp];
                    else
                        store_row_stream<G, V, CH>(qk[pp], I.chain + (size_t)wi[pp] * D, D, gl);
                }
            }
        }
        if (DEFER && I.chain) {
            cchain = I.chain;
            cchain_lp = I.chain_lp;
            cmy_i = my_i;
#pragma unroll
            for (int k = 0; k < PF; ++k) cwi[k] = wi[k];
        }
        EMX_WAVE_SYNC();
        EMX_PSTAMP(3);       // decisions made, commit stores issued
        if (prof) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            EMX_PSTAMP(4);   // stores acknowledged (and the speculative own rows of the next half-step in)
        }
        } else {             // (MIX: no tile in this half-step)
            next_entries();
            next_rows();
        }
        if (!more) break;
        if constexpr (LOCAL)
            persist_barrier_local(P, P.lepoch0 + (unsigned)n + 1u, bid, ngroups);
        else
            persist_barrier(P, P.epoch0 + (unsigned)n + 2u);       // (+ 1: the handshake was this launch's first barrier)
        EMX_PSTAMP(5);       // device-wide barrier
        // -------- roll over --------
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            wi[k] = wi_n[k];
            ja[k] = ja_n[k];
            if constexpr (DE) jb[k] = jb_n[k];
            if constexpr (SNA) jc[k] = jc_n[k];
            s0v[k] = s0_n[k];
            facv[k] = fac_n[k];
        }
        my_i = my_i_n;
        my_logu = my_logu_n;
        my_lpo = my_lpo_n;
        if constexpr (MIX) {
            t0 = t0_n;
            act = act_n;
            sn = sn_n;
        }
        if (!pre && act) {      // first split of a new step: the walkers that moved in the half-step before are loaded again
            unsigned vk[PF];
#pragma unroll
            for (int k = 0; k < PF; ++k) vk[k] = load_agent(P.ver + wi[k]);
            const unsigned vm = load_agent(P.ver + my_i);
#pragma unroll
            for (int k = 0; k < PF; ++k)
                if (vk[k] == stamp) load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi[k], D, gl);
            if (vm == stamp) my_lpo = load_agent(A.lp + my_i);
        }
    }
    if (DEFER && cchain) {      // the last half-step's rows
#pragma unroll
        for (int k = 0; k < PF; ++k) store_row_stream<G, V, CH>(crow[k], cchain + (size_t)cwi[k] * D, D, gl);
        if (mine) cchain_lp[cmy_i] = clp;
    }
    if (prof && lane == 0) {
        unsigned long long* o = A.dbg + (size_t)bid * 16;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = pst[k];
        o[6] = (unsigned long long)P.niter;
        o[12] = wall_clock64();
    }
#undef EMX_PSTAMP
}


}  // namespace emx
