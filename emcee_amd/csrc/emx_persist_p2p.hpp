// k_persist_p2p: the persistent half-step kernel of the headline shape (stretch move, fused dense Gaussian, Philox plans) WITHOUT a
// device-wide barrier between the half-steps and WITHOUT a bandwidth-bound load phase behind it.
//
// What k_persist (emx_kernels.hpp) pays per half-step at 65 536 x 64 (profiles/r04/persist_phase_c2.txt): the device-wide barrier
// 2.5 us, then -- only then -- the partner rows' sc1 round trip, 16.8 MB chip-wide, 3.0 us: 57 % of a 9.6 us half-step in two
// dependent latencies, every wave in the same phase.  Here
//   * a tile (= wave) that has committed a half-step publishes ONE 8-byte word: {launch, half-step | 16 accept bits}, after its
//     stores have been acknowledged (agent-scope stores: the device can see them);
//   * the plan tells every slot which tile's word decides about each row it reads -- the partner's slot in the half-step before
//     (stretch.py:30-32: the partner is member `rint` of the complement, so its tile is known when the plan is made), and for the
//     first split of a step the slots its own walker and its partner had in the LAST split of the step before (one evaluation of
//     that step's keyed permutation each; k_native_plan_batch, columns p1 / p2 of a lean stretch plan);
//   * ALL rows of the next half-step -- own rows and partner rows -- are loaded speculatively next to the MFMA phase, when the
//     memory system is idle; at the start of the next half-step a wave polls the <= 32 words it depends on (one round trip of a
//     few bytes), and re-loads exactly the rows whose accept bit is set (~17 % at the headline target's acceptance).
// So the dependent chain of a half-step is: commit acknowledged -> word visible -> poll -> a light re-load -> compute, and no wave
// waits for the slowest of 2 048.
// Two hazards the device-wide barrier excluded for free are excluded by ONE lagging gate (per-XCD arrival counters as before,
// but nobody waits where it arrives):
//   gate(n - 1) = "every workgroup has finished half-step n - 1", waited for in the MIDDLE of half-step n -- some 4 us after the
//   last arrival, i.e. met when asked in practice --
//   (1) before the speculative loads of half-step n + 1 are issued: a row that is NOT re-loaded must carry every commit up to
//       n - 1 (rows committed in n are named by the words);
//   (2) before the commits of half-step n: they overwrite rows in place that a lagging wave may still have to read as partners
//       of half-step n - 1 (write after read).
// The words are double-buffered by the parity of the half-step: a tile rewrites a word two half-steps later, which the gate puts
// behind every reader of the old value.  Deadlock-free: every wait is for something an EARLIER half-step of another wave
// produces, and the handshake (persist_handshake) has shown the whole grid resident.  Every spin is bounded by the wall clock
// and gives way to the `dead` mark like k_persist's.
// Same load_row / make_proposal / MFMA chain / reductions / decision / commit as k_persist<G, V, CH, DPB, MOVE_STRETCH> in the same
// order: the same bits (tests/test_gpu_persist.py).  Included by emx_hot.hip only (the ILP scheduler's translation unit).
#pragma once
#include "emx_kernels.hpp"

namespace emx {

#ifndef EMX_P2P_WSHIFT
#define EMX_P2P_WSHIFT P2P_WSHIFT         // a tile's word lives at index tile << 5: 256 bytes apart.  Sixteen words to a 128-byte line had every
#endif                           // wave's polls -- 32 K loads a round -- land on 128 lines of a few memory channels, in front of the
                                 // publishing stores: 9 us in the poll phase (profiles/r05/p2p_first_contact.txt)
#ifndef EMX_P2P_SLEEP
#define EMX_P2P_SLEEP 16         // s_sleep (x 64 clocks) between two polls of a word that was not there yet: 2 048 waves polling 16 words
#endif                           // each as fast as the loads return is a request rate of the order of the chip's whole memory system
#ifndef EMX_P2P_SPEC_PARTNER
#define EMX_P2P_SPEC_PARTNER 1   // 0: partner rows are loaded after the words have been seen (all of them), not speculatively
#endif

// bit of tile row r in a published word: the decision lanes are (lane & 15) < 4, lane 16 b + a decides row 4 a + b
__device__ __forceinline__ unsigned p2p_bit(unsigned r) { return 4u * (r & 3u) + (r >> 2); }

// CHAIN: some half-step of the launch stores its step (backend.py:229).  The instantiation without has no chain code at all; the one
// with writes a walker's old row when the half-step starts and the proposal over it when it is accepted (streaming stores, fire and
// forget: k_persist's deferred rows would cost this kernel the 32 registers its speculative partner rows live in).
template <int G, int V, int CH, int DPB, bool CHAIN>
static __global__ __launch_bounds__(512) void k_persist_p2p(const PersistArgs P) {
    constexpr int MOVE = MOVE_STRETCH;
    constexpr int CPOL = EMX_CPOL_SC1;
    constexpr int WPW = 64 / G;
    constexpr int PPT = 16 / WPW;
    constexpr int PF = PPT;
    static_assert(EMX_OPT_RTILE && EMX_OPT_RED4, "the persistent kernel is the one-tile-per-batch form");
    constexpr int Dp = DPB * 16, KK = Dp / 4, RT = Dp + 2;
    static_assert(G * V * CH >= Dp, "row layout must cover the padded dimension");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ unsigned wg_arrived;                              // waves of this workgroup that have finished a half-step (running count)
    const HalfStepArgs& A = P.base;
    const unsigned bid = blockIdx.x, ngroups = gridDim.x;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const unsigned nw = blockDim.x >> 6;
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
    if (threadIdx.x == 0) wg_arrived = 0u;
    Row<G, V, CH> mu;
    load_row<G, V, CH>(mu, A.tp0, D, gl);
    if (!persist_handshake<false>(P)) return;                   // (also the workgroup barrier behind the image load)
    const int wave = (int)bid * (int)nw + wib;                  // = this wave's tile of every half-step
    const int t0 = wave * 16;
    unsigned long long pst[6] = {0, 0, 0, 0, 0, 0}, pt = 0;      // instrumented build only (tools/persist_phase_clock.py)
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
    unsigned* const go = P.bar + 9 * 32;                         // [go | dead | ...] (persist_handshake): `dead` opens every wait
    const unsigned* const gate_ctr = P.pctr + (lane & 7) * 32;   // lanes 0 .. 7 watch the eight arrival counters
    const unsigned gate_per = (ngroups + 7u - (unsigned)(lane & 7)) / 8u;      // workgroups that arrive on counter (lane & 7)
    bool dead = false;                                           // a wait of this run has timed out: nothing waits any more (the run is void)

    int wi[PF], ja[PF], dpk[PF], dik[PF], my_i;
    double s0v[PF], facv[PF], my_logu, my_lpo;
    Row<G, V, CH> xi[PF], xa[PF];
    {
        const PersistIter& I = P.it[0];
        const int pbase = I.pos0 + t0;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int pos = pbase + k * WPW + sub;
            wi[k] = I.order[pos];
            ja[k] = I.p0[pos];
            dpk[k] = dik[k] = -1;                                // (the kernel boundary: nothing to wait for)
            s0v[k] = I.s0[pos];
            facv[k] = I.fac[pos];
        }
        my_i = I.order[pbase + myrow];
        my_logu = I.logu[pbase + myrow];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi[k], D, gl);
            load_row_agent<G, V, CH, CPOL>(xa[k], Xr, ja[k], D, gl);
        }
        my_lpo = load_agent(A.lp + my_i);
    }
    for (int n = 0; n < P.niter; ++n) {
        const PersistIter& I = P.it[n];
        unsigned gate_v = 0u;
        if (n > 0) {
            // -------- the words this tile depends on: the tiles that may have moved one of its rows in half-step n - 1 --------
            const unsigned long long want = ((unsigned long long)P.seq << 6) | (unsigned long long)(n - 1);
            const unsigned long long* const W = P.tw + (size_t)((n - 1) & 1) * P.tw_stride;
            unsigned long long wp[PF], wo[PF];
            if (lane < 8) gate_v = load_agent(gate_ctr);        // (consumed in the middle of the half-step)
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                wp[k] = wo[k] = want << 16;                      // (no dependency: ready, no bit set)
                if (dpk[k] >= 0) wp[k] = __hip_atomic_load(W + ((size_t)(dpk[k] >> 4) << EMX_P2P_WSHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dik[k] >= 0) wo[k] = __hip_atomic_load(W + ((size_t)(dik[k] >> 4) << EMX_P2P_WSHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned long long tw0 = 0ull;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < PF; ++k) ok = ok && (wp[k] >> 16) == want && (wo[k] >> 16) == want;
                if (__ballot(!ok) == 0ull || dead) break;
                if (tw0 == 0ull) tw0 = wall_clock64();
                if (EMX_P2P_SLEEP) __builtin_amdgcn_s_sleep(EMX_P2P_SLEEP);
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    if ((wp[k] >> 16) != want) wp[k] = __hip_atomic_load(W + ((size_t)(dpk[k] >> 4) << EMX_P2P_WSHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((wo[k] >> 16) != want) wo[k] = __hip_atomic_load(W + ((size_t)(dik[k] >> 4) << EMX_P2P_WSHIFT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned long long waited = wall_clock64() - tw0;
                if (waited > 2000ull) dead = __builtin_amdgcn_readfirstlane((int)load_agent(go + 1)) != 0;      // (20 us: ONE word for 2 048 waves -- not in the fast path)
                if (!dead && waited > P.timeout_ticks) {
                    if (lane == 0) {
                        raise_status(A.status, ST_EXCHANGE_TIMEOUT);
                        __hip_atomic_store(go + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // 2: in the middle of a launch
                    }
                    dead = true;
                }
            }
            // -------- the rows that moved: loaded again (the others are the speculative loads of half-step n - 1) --------
            unsigned long long bo[PF];
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const bool cp = !EMX_P2P_SPEC_PARTNER || (dpk[k] >= 0 && ((wp[k] >> p2p_bit((unsigned)dpk[k] & 15u)) & 1ull) != 0ull);
                const bool co = dik[k] >= 0 && ((wo[k] >> p2p_bit((unsigned)dik[k] & 15u)) & 1ull) != 0ull;
                if (cp) load_row_agent<G, V, CH, CPOL>(xa[k], Xr, ja[k], D, gl);
                if (co) load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi[k], D, gl);
                bo[k] = __ballot(co);
            }
            {   // the decision lane's own log-prob: moved with row `myrow` (pass myrow / WPW, group myrow % WPW)
                bool cl = false;
#pragma unroll
                for (int k = 0; k < PF; ++k)
                    if (myrow / WPW == k) cl = ((bo[k] >> ((myrow % WPW) * G)) & 1ull) != 0ull;
                if (mine && cl) my_lpo = load_agent(A.lp + my_i);
            }
        }
        // -------- plan entries of the next half-step (written by the plan kernel before this launch) --------
        const bool more = n + 1 < P.niter;
        const PersistIter& J = P.it[more ? n + 1 : n];
        int wi_n[PF], ja_n[PF], dp_n[PF], di_n[PF], my_i_n;
        double s0_n[PF], fac_n[PF], my_logu_n;
        {
            const int pbase = J.pos0 + t0;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int pos = pbase + k * WPW + sub;
                wi_n[k] = J.order[pos];
                ja_n[k] = J.p0[pos];
                dp_n[k] = J.p1[pos];                             // lean stretch plans with `deps` (k_native_plan_batch)
                di_n[k] = J.p2[pos];
                s0_n[k] = J.s0[pos];
                fac_n[k] = J.fac[pos];
            }
            my_i_n = J.order[pbase + myrow];
            my_logu_n = J.logu[pbase + myrow];
        }
        // -------- proposals -> the wave's LDS tile (R = Q - mu), kept in registers for the commit --------
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        EMX_PSTAMP(0);       // words seen, moved rows (and the next half-step's plan entries) have arrived
        Row<G, V, CH> qk[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int srow = k * WPW + sub;
            double factor = facv[k];
            Row<G, V, CH> q;
            make_proposal<G, V, CH, MOVE>(xi[k], xa[k], xa[k], xa[k], s0v[k], A.gammas, D, gl, q, factor, ja[k]);
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
        // -------- stored steps: the walker's row as it is; an accepted proposal overwrites it after the decision --------
        if (CHAIN && I.chain) {
#pragma unroll
            for (int k = 0; k < PF; ++k) store_row_stream<G, V, CH>(xi[k], I.chain + (size_t)wi[k] * D, D, gl);
        }
        // -------- the gate: every workgroup has finished half-step n - 1 (see the head of this file) --------
        if (n > 0) {
            unsigned long long tg0 = 0ull;
            for (;;) {
                const unsigned need = (P.pepoch0 + (unsigned)n) * gate_per;
                const bool ok = lane < 8 ? (int)(gate_v - need) >= 0 : true;
                if (__ballot(!ok) == 0ull || dead) break;
                if (tg0 == 0ull) tg0 = wall_clock64();
                if (lane < 8) gate_v = load_agent(gate_ctr);
                const unsigned long long waited = wall_clock64() - tg0;
                if (waited > 2000ull) dead = __builtin_amdgcn_readfirstlane((int)load_agent(go + 1)) != 0;      // (20 us: ONE word for 2 048 waves -- not in the fast path)
                if (!dead && waited > P.timeout_ticks) {
                    if (lane == 0) {
                        raise_status(A.status, ST_EXCHANGE_TIMEOUT);
                        __hip_atomic_store(go + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    dead = true;
                }
            }
        }
        // -------- every row of the next half-step, speculatively: in flight during the MFMA phase --------
        double my_lpo_n = 0.0;
        if (more) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi_n[k], D, gl);
                if (EMX_P2P_SPEC_PARTNER) load_row_agent<G, V, CH, CPOL>(xa[k], Xr, ja_n[k], D, gl);
            }
            my_lpo_n = load_agent(A.lp + my_i_n);
        }
        EMX_WAVE_SYNC();
        EMX_PSTAMP(1);       // proposals made, tile written, chain rows and the next half-step's rows issued
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
            store_agent(A.acc + my_i, (uint8_t)(acc ? 1 : 0));
            if (acc) store_agent(A.lp + my_i, lpn);
            if (CHAIN && I.chain_lp) {
                I.chain_lp[my_i] = acc ? lpn : my_lpo;
                if (acc) store_agent(A.acc_count + my_i, load_agent(A.acc_count + my_i) + 1u);
            }
        }
        const unsigned long long am64 = __ballot(acc);           // bit (row & 3) * 16 + (row >> 2) <-> tile row
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp) {
            const int row = pp * WPW + sub;
            const bool ac = (am64 >> ((row & 3) * 16 + (row >> 2))) & 1ull;
            if (ac) {
                store_row_agent<G, V, CH, CPOL>(qk[pp], Xr, wi[pp], D, gl);
                if (CHAIN && I.chain) store_row_stream<G, V, CH>(qk[pp], I.chain + (size_t)wi[pp] * D, D, gl);
            }
        }
        EMX_WAVE_SYNC();
        EMX_PSTAMP(3);       // decisions made, commit stores issued
        if (!more) break;
        // -------- publish: commits acknowledged (and the speculative rows in) -> this tile's word, this workgroup's arrival --------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        EMX_PSTAMP(4);       // stores acknowledged, the next half-step's rows have arrived
        if (lane == 0) {
            const unsigned m16 = (unsigned)(am64 & 0xfull) | (unsigned)((am64 >> 12) & 0xf0ull) | (unsigned)((am64 >> 24) & 0xf00ull) |
                                 (unsigned)((am64 >> 36) & 0xf000ull);          // bit 4 b + a <- lane 16 b + a
            const unsigned long long word = ((((unsigned long long)P.seq << 6) | (unsigned long long)n) << 16) | (unsigned long long)m16;
            __hip_atomic_store(P.tw + (size_t)(n & 1) * P.tw_stride + ((size_t)wave << EMX_P2P_WSHIFT), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned old = __hip_atomic_fetch_add(&wg_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old + 1u) % nw == 0u)       // the workgroup's last wave: every wave's stores were acknowledged before it counted
                __hip_atomic_fetch_add(P.pctr + (bid & 7u) * 32u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        EMX_PSTAMP(5);       // word and arrival issued
        // -------- roll over --------
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            wi[k] = wi_n[k];
            ja[k] = ja_n[k];
            dpk[k] = dp_n[k];
            dik[k] = di_n[k];
            s0v[k] = s0_n[k];
            facv[k] = fac_n[k];
        }
        my_i = my_i_n;
        my_logu = my_logu_n;
        my_lpo = my_lpo_n;
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
