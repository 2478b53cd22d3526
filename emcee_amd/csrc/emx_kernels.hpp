// gfx950 kernels for the red/blue split-ensemble half-step.
//
// One launch = one half-step (one `split` of moves/red_blue.py:81-104): every walker of the
// active sub-ensemble is proposed (stretch.py:26-33 | de.py:40-64 | de_snooker.py:31-46), its
// log-probability is evaluated (ensemble.py:458-553, here fused), Metropolis-tested
// (red_blue.py:96-101) and committed in place (move.py:29-45).  The complement is not written
// during a launch, so there is no intra-launch hazard; the kernel boundary is the grid barrier
// the parallel stretch move needs between splits (document/ms.tex:447-463).
//
// Work mapping (CDNA4, wave64):
//   * a wave owns `spw` consecutive *slots* (members of the active sub-ensemble);
//   * phase A is lane-parallel over slots: lane l resolves slot l's walker, partner(s) and
//     scalar draws (from the host plan in exact mode, from Philox in native mode) and pays the
//     two logs once per walker instead of once per lane;
//   * phase B walks the slots WPW = 64/G at a time: G lanes share one walker row, each lane
//     holding CH chunks of V contiguous doubles (V = 2 -> 16-byte global loads, a row is read
//     as full 128-byte lines); per-row reductions are xor-shuffles inside the G-lane group;
//   * dense Gaussian target: rows of 16 walkers are staged in an LDS tile and contracted with
//     the LDS-resident precision matrix by v_mfma_f64_16x16x4_f64.
// Compile with -ffp-contract=off: the proposal arithmetic must round exactly like NumPy's
// separate multiply / subtract; reductions use explicit fma().
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "emx_planlog.hpp"
#include "emx_rng.hpp"

// build-time experiment switches (tools/ab_variants.sh builds variants, tools/ab_bench.sh alternates them on one
// box); the defaults are the shipped configuration.  Measured on MI355X at C2 (27.9 us/step baseline):
// RED4 +1.2 %; timestamps -1 %.  Tried and dropped: mean fragments held in registers (16 more loads at kernel
// start) -4 %; plan-entry loads issued before the image loads -2.2 %; commit as a wave-wide copy of one accepted
// row per iteration -2 % (profiles/r01/ab_variants.txt).
#ifndef EMX_OPT_RED4
#define EMX_OPT_RED4 1
#endif
#ifndef EMX_OPT_RTILE
#define EMX_OPT_RTILE 1       // dense target, batch == tile: the LDS tile holds R = Q - mu (no mean reads at the A fragments), accepted
#endif                        // rows are committed from the registers that made them (no tile re-read)
#ifndef EMX_LEAN_FOLD_D
#define EMX_LEAN_FOLD_D 0
#endif
#ifndef EMX_LEAN_FOLD_SPW
#define EMX_LEAN_FOLD_SPW 1     // +0.7 % at C2
#endif
#ifndef EMX_OPT_SKEW
#define EMX_OPT_SKEW 1        // dense target, 8-wave workgroups: the upper four waves stage the whole LDS image before they issue
#endif                        // their row loads, so the two waves of a SIMD run out of phase (loads first for the lower four)
#ifndef EMX_SNOOKER_BUDGET
#define EMX_SNOOKER_BUDGET 48  // doubles of rows in flight per lane for the snooker move (64 measured slower, see prefetch_depth)
#endif
#ifndef EMX_STAMP_WAVE
#define EMX_STAMP_WAVE 0      // instrumented build: the wave of every workgroup whose phases are clocked
#endif
#ifndef EMX_OPT_STAMPS
#define EMX_OPT_STAMPS 0      // phase timestamps (tools/phase_clock.py builds its own copy with -DEMX_OPT_STAMPS=1: they cost 1 %)
#endif
#ifndef EMX_EXPERIMENTS
#define EMX_EXPERIMENTS 0     // the timing experiments' switches (tuning "ablate": skip-phase masks of k_halfstep and the plan kernel) are compiled
#endif                        // only into the experiments flavour of the library (EMX_BUILD_FLAVOUR=exp -> libemx_exp.so, tools/ablate.py)

namespace emx {

enum : int { MOVE_STRETCH = 0, MOVE_DE = 1, MOVE_SNOOKER = 2, MOVE_GAUSS = 3, MOVE_EVAL = 4, MOVE_MIX = 8 /* k_persist_mix: DE and snooker steps in one launch */ };
enum : int { GAUSS_VECTOR = 0, GAUSS_RANDOM = 1, GAUSS_SEQUENTIAL = 2 };
enum : int { TGT_NONE = 0, TGT_ISO = 1, TGT_DIAG = 2, TGT_DENSE = 3, TGT_ROSEN = 4, TGT_BOX = 5,
             TGT_REPLAY = 7 };      // (6 is EMX_TARGET_DEVICE_CALLBACK, a host-side three-pass target: never a kernel's)     // replay exchange: no target, no decision -- the slot is a peer's ACCEPTED update, its new log-prob comes with the plan
enum : uint32_t { ST_NAN_LOGP = 1u, ST_BAD_COORD = 2u, ST_EXCHANGE_OVERFLOW = 4u, ST_EXCHANGE_TIMEOUT = 8u,
                  ST_PLAN_PRODUCER = 16u };     // bit 4: the device producer of exact-mode plans stalled or under-ran (emx_mtdev_kernels.hpp)
constexpr int EMX_MAX_PEERS = 8;       // direct exchange: GPUs of one node

// The sticky status lives in mapped host memory, one 32-bit flag per condition (index = bit number): raising one is
// a plain idempotent store -- no read-modify-write across PCIe -- and only error paths ever execute it.
__device__ __forceinline__ void raise_status(uint32_t* flags, uint32_t bit) {
    __hip_atomic_store(&flags[__builtin_ctz(bit)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct NativeArgs {
    uint64_t seed;
    uint64_t step;
    PermKey pk;
};

// hipGraph replay: everything that changes from one replay of the captured 8-step block to the next
// lives in device memory, written by k_graph_advance (the first node of the graph).
struct StepDesc {
    NativeArgs nat;            // seed, step, per-step keyed permutation
    long long stored_idx;      // chain row this step appends to, -1: not stored
};
struct GraphCounters {
    unsigned long long step_base;
    long long stored_base;
};

struct HalfStepArgs {
    // ensemble state
    double* X;            // (N, D) row-major
    double* lp;           // (N)
    uint8_t* acc;         // (N) accept flag of the current step
    uint32_t* acc_count;  // (N) accumulated on stored steps (backend.py:229)
    double* chain;        // row block of the stored step (N, D) or nullptr
    double* chain_lp;     // (N) or nullptr
    uint32_t* status;     // sticky error bits
    // split-phase outputs (target == TGT_NONE): proposals in slot order
    double* qout;         // (ns, D)
    double* fout;         // (ns)   factors
    double* sendbuf;      // sharded runs: final [row | log_prob | accepted] of the owned slots,
                          // (t_hi - t_lo, D + 2) or nullptr
    // plan (exact / inputs mode), slot-indexed at position pos0 + t
    const int32_t* order;
    const int32_t* p0;
    const int32_t* p1;
    const int32_t* p2;
    const double* s0;
    const double* uacc;
    const double* logu;   // optional: log(uacc) precomputed by k_native_plan
    const double* fac;    // optional: (D-1) ln zz precomputed by k_native_plan
    // target
    const double* tp0;    // mu  (diag, dense)
    const double* tp1;    // ivar (diag) | icov (dense, (D, D) row-major)
    double tscale;        // rosenbrock temperature (20), box: unused
    // move parameters
    double a;             // stretch scale
    double sigma, g0;     // DE
    double gammas;        // snooker
    int32_t N, D, S, split;
    int32_t pos0, ns;     // first plan position / number of slots of this split
    int32_t t_lo, t_hi;   // slots updated by this rank
    int32_t spw;          // slots per wave
    int32_t target;
    int32_t Dp;           // dense: D rounded up to 16
    int32_t ablate;       // timing experiments only (tools/ablate.py): skip-phase bit mask, 0 in production
    // graph replay: chain row resolved on the device from desc->stored_idx
    const StepDesc* desc;
    double* chain_all;
    double* chain_lp_all;
    // pull exchange: the slot count of the (compacted) plan is only known on the device
    const int32_t* t_hi_dev;
    // Gaussian Metropolis move: (N, D) displacement rows, indexed by walker (k_gauss_*); nullptr in the
    // native mode, where the rows are generated in registers from (gseed, gstep, walker, coordinate pair)
    const double* disp;
    const double* gscale;          // (D) standard deviations or nullptr: isotropic gsigma
    double gsigma, gfac;           // gfac: this step's step-size factor
    unsigned long long gseed, gstep;
    // phase timestamps (tools/phase_clock.py): [block][16] s_memtime samples of wave 0, or nullptr
    unsigned long long* dbg;
    // direct exchange (walker-block ownership, the peers' coordinate arrays mapped into this device's address space):
    // a partner row is read from the replica of the rank that owns it, i.e. over xGMI from that GPU's HBM.  npeer = 0:
    // one replica (everything else).  peer_lo is ascending; own entry of peerX == X.
    const struct PeerTable* peers;     // device memory (not kernel arguments: 24 scalar registers the single-GPU path never needs)
    int32_t npeer;
    // replay exchange: the decision of every slot of this launch, 8 bytes each -- the new log-prob of an accepted proposal,
    // NaN for a rejected one (an accepted log-prob is never NaN: NaN > log u is false) -- slot t at declp[t - t_lo]
    double* declp;
    // ... and, with the peers' receive buffers mapped (device-side replay exchange), stored straight into every one of them
    // as well: element push_off + (t - t_lo) of push_peers->X[q], q < npush (remote stores over xGMI, fire and forget; the
    // barrier kernel that follows publishes them)
    const struct PeerTable* push_peers;
    long long push_off;
    int32_t npush;
    int32_t skew_sleep;            // EMX_OPT_SKEW experiments: extra delay of the staging waves, in s_sleep(8) units (0 in production)
};

struct PeerTable {
    const double* X[EMX_MAX_PEERS];
    int32_t lo[EMX_MAX_PEERS];
};

// coordinate array holding the current row of walker j (block ownership: rank q owns [lo[q], lo[q + 1]))
template <int LEAN>
__device__ __forceinline__ const double* partner_base(const HalfStepArgs& A, int j) {
    if (LEAN == 1 || LEAN == 3 || LEAN == 4 || A.npeer == 0) return A.X;      // launch-uniform
    const PeerTable* __restrict__ T = A.peers;      // uniform address: scalar loads
    const double* b = T->X[0];
#pragma unroll
    for (int q = 1; q < EMX_MAX_PEERS; ++q) b = (q < A.npeer && j >= T->lo[q]) ? T->X[q] : b;
    return b;
}

// ----------------------------------------------------------------------------------------
// group helpers
// ----------------------------------------------------------------------------------------
// Cross-lane sums without the LDS crossbar: DPP row operations inside a 16-lane row, then the
// gfx950 v_permlane16_swap / v_permlane32_swap pair sums.  Every lane of the (aligned, contiguous)
// G-lane group ends up with the group total.  Must be called from wave-uniform control flow.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double sum_swap16(double x) {   // x[l] + x[l ^ 16]
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

__device__ __forceinline__ double sum_swap32(double x) {   // x[l] + x[l ^ 32]
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

template <int G>
__device__ __forceinline__ double group_sum(double x) {
    if constexpr (G >= 2) x += dpp_f64<0xB1>(x);    // quad_perm [1,0,3,2]  : l ^ 1
    if constexpr (G >= 4) x += dpp_f64<0x4E>(x);    // quad_perm [2,3,0,1]  : l ^ 2
    if constexpr (G >= 8) x += dpp_f64<0x141>(x);   // row_half_mirror      : other quad of the 8
    if constexpr (G >= 16) x += dpp_f64<0x140>(x);  // row_mirror           : other half of the row
    if constexpr (G >= 32) x = sum_swap16(x);
    if constexpr (G >= 64) x = sum_swap32(x);
    return x;
}

// Four values summed over the 16 lanes of a DPP row at once ("transpose-reduce"): the first two steps halve
// the number of live values instead of carrying all four through four butterfly steps.  On return every lane
// holds the row total of value (lane & 3): 5 adds instead of 16.
__device__ __forceinline__ double row16_sum4(double v0, double v1, double v2, double v3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const double x = (b0 ? v1 : v0) + dpp_f64<0xB1>(b0 ? v0 : v1);    // value (lane & 1) over {l, l ^ 1}
    const double y = (b0 ? v3 : v2) + dpp_f64<0xB1>(b0 ? v2 : v3);    // value 2 + (lane & 1)
    double z = (b1 ? y : x) + dpp_f64<0x4E>(b1 ? x : y);              // value (lane & 3) over the quad
    z += dpp_f64<0x124>(z);                                           // row_ror:4  -- lanes with the same (lane & 3)
    z += dpp_f64<0x128>(z);                                           // row_ror:8
    return z;
}

// any-lane-in-group predicate from one ballot (no data movement)
template <int G>
__device__ __forceinline__ bool group_any(bool pred, int sub) {
    const unsigned long long m = __ballot(pred);
    if constexpr (G == 64) {
        return m != 0ull;
    } else {
        return ((m >> (sub * G)) & ((1ull << G) - 1ull)) != 0ull;
    }
}

// Native-mode draws for slot t of `split`: a pure function of (seed, step, walker).
template <int MOVE>
__host__ __device__ inline void native_slot(const NativeArgs& na, int N, int S, int split, int t, double a,
                                            double sigma, double g0, int& i, int& p0, int& p1, int& p2, double& s0,
                                            double& uacc) {
    const uint32_t k0 = (uint32_t)na.seed, k1 = (uint32_t)(na.seed >> 32);
    const uint32_t sl = (uint32_t)na.step, sh = (uint32_t)(na.step >> 32);
    i = (int)perm_inv((uint32_t)(t * S + split), na.pk);
    p0 = p1 = p2 = i;
    s0 = 0.0;
    const Philox4 A = philox4x32_10((uint32_t)i, 0u, sl, sh, k0, k1);
    uacc = u53(A.v[2], A.v[3]);
    if (MOVE == MOVE_EVAL) return;
    const Philox4 B = philox4x32_10((uint32_t)i, 1u, sl, sh, k0, k1);
    const SplitSizes sz = split_sizes(N, S);
    const int ns_own = sz.of(split);
    const int64_t Nc = (int64_t)N - ns_own;
    if (MOVE == MOVE_STRETCH) {
        const double u = u53(A.v[0], A.v[1]);
        const double tt_ = (a - 1.0) * u + 1.0;
        double inva;
        s0 = pow2_reciprocal(a, inva) ? tt_ * tt_ * inva : tt_ * tt_ / a;        // the same bits either way (emx_rng.hpp)
        const int64_t r = (int64_t)bounded64(B.v[0], B.v[1], (uint64_t)Nc);
        int j = 0, tt = 0;
        // inline comp_locate (host+device)
        int64_t rr = r;
        for (int s = 0; s < S; ++s) {
            if (s == split) continue;
            const int n = sz.of(s);
            if (rr < n) { j = s; tt = (int)rr; break; }
            rr -= n;
        }
        p0 = (int)perm_inv((uint32_t)(tt * S + j), na.pk);
    } else if (MOVE == MOVE_DE) {
        int64_t r1 = (int64_t)bounded64(B.v[0], B.v[1], (uint64_t)Nc);
        int64_t r2 = (int64_t)bounded64(B.v[2], B.v[3], (uint64_t)(Nc - 1));
        if (r2 >= r1) ++r2;
        int64_t rr[2] = {r1, r2};
        int pw[2] = {i, i};
        for (int q = 0; q < 2; ++q) {
            int64_t x = rr[q];
            for (int s = 0; s < S; ++s) {
                if (s == split) continue;
                const int n = sz.of(s);
                if (x < n) { pw[q] = (int)perm_inv((uint32_t)((int)x * S + s), na.pk); break; }
                x -= n;
            }
        }
        p0 = pw[0];
        p1 = pw[1];
        // Box-Muller normal for gamma = g0 (1 + sigma N(0,1))   (de.py:56)
        const Philox4 C = philox4x32_10((uint32_t)i, 2u, sl, sh, k0, k1);
        const double u1 = 1.0 - u53(C.v[0], C.v[1]);  // (0, 1]
        const double u2 = u53(C.v[2], C.v[3]);
        const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
        s0 = g0 * (1.0 + sigma * g);
    } else if (MOVE == MOVE_SNOOKER) {
        // one member from each of the first three complement sets, then a uniform permutation
        const Philox4 C = philox4x32_10((uint32_t)i, 2u, sl, sh, k0, k1);
        int w[3] = {i, i, i};
        uint32_t wa[3] = {B.v[0], B.v[2], C.v[0]}, wb[3] = {B.v[1], B.v[3], C.v[1]};
        int q = 0;
        for (int s = 0; s < S && q < 3; ++s) {
            if (s == split) continue;
            const int n = sz.of(s);
            const int tt = (int)bounded64(wa[q], wb[q], (uint64_t)n);
            w[q] = (int)perm_inv((uint32_t)(tt * S + s), na.pk);
            ++q;
        }
        const int pm = (int)(((uint64_t)C.v[2] * 6ull) >> 32);  // 0..5
        // permutations of (0,1,2) in lexicographic order
        const int P0[6] = {0, 0, 1, 1, 2, 2}, P1[6] = {1, 2, 0, 2, 0, 1}, P2[6] = {2, 1, 2, 0, 1, 0};
        p0 = w[P0[pm]];
        p1 = w[P1[pm]];
        p2 = w[P2[pm]];
    }
}

// ----------------------------------------------------------------------------------------
// row I/O in the (G, V, CH) layout: lane gl holds columns (c*G + gl)*V + v
// ----------------------------------------------------------------------------------------
template <int G, int V, int CH>
struct Row {
    double x[CH][V];
};

template <int G, int V, int CH>
__device__ __forceinline__ void load_row(Row<G, V, CH>& r, const double* __restrict__ base, int D, int gl) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int d = (c * G + gl) * V;
        if constexpr (V == 2) {
            if (d + 1 < D) {
                const double2 t = *reinterpret_cast<const double2*>(base + d);
                r.x[c][0] = t.x;
                r.x[c][1] = t.y;
            } else {
                r.x[c][0] = 0.0;
                r.x[c][1] = 0.0;
            }
        } else {
            r.x[c][0] = d < D ? base[d] : 0.0;
        }
    }
}

template <int G, int V, int CH>
__device__ __forceinline__ void store_row(const Row<G, V, CH>& r, double* __restrict__ base, int D, int gl) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int d = (c * G + gl) * V;
        if constexpr (V == 2) {
            if (d + 1 < D) {
                double2 t;
                t.x = r.x[c][0];
                t.y = r.x[c][1];
                *reinterpret_cast<double2*>(base + d) = t;
            }
        } else {
            if (d < D) base[d] = r.x[c][0];
        }
    }
}

// Chain rows are written once and read, if ever, by a later get_chain: streaming ("nt") stores, so that 33.5 MB of them per
// stored step do not push the ensemble itself out of the Infinity Cache.
#ifndef EMX_OPT_NT_CHAIN
#define EMX_OPT_NT_CHAIN 1
#endif
template <int G, int V, int CH>
__device__ __forceinline__ void store_row_stream(const Row<G, V, CH>& r, double* __restrict__ base, int D, int gl) {
#if EMX_OPT_NT_CHAIN
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int d = (c * G + gl) * V;
        if constexpr (V == 2) {
            if (d + 1 < D) {
                typedef double d2v __attribute__((ext_vector_type(2)));
                d2v t;
                t.x = r.x[c][0];
                t.y = r.x[c][1];
                __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(base + d));
            }
        } else {
            if (d < D) __builtin_nontemporal_store(r.x[c][0], base + d);
        }
    }
#else
    store_row<G, V, CH>(r, base, D, gl);
#endif
}

// ----------------------------------------------------------------------------------------
// element-wise targets evaluated from the register-resident proposal
// ----------------------------------------------------------------------------------------
template <int G, int V, int CH>
__device__ __forceinline__ double eval_valu_target(const Row<G, V, CH>& q, const Row<G, V, CH>& mu,
                                                   const Row<G, V, CH>& iv, const double* __restrict__ tp0,
                                                   const double* __restrict__ tp1, int target, double tscale, int D,
                                                   int gl, int lane) {
    double acc = 0.0;
    if (target == TGT_ISO) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) acc = fma(q.x[c][v], q.x[c][v], acc);
        return -0.5 * group_sum<G>(acc);
    }
    if (target == TGT_DIAG) {
        if constexpr (CH <= 4) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const double d = q.x[c][v] - mu.x[c][v];
                    acc = fma(iv.x[c][v] * d, d, acc);
                }
        } else {
            // wide rows: stream (mu, ivar) from L1/L2 chunk by chunk instead of pinning 4*CH*V registers
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int d0 = (c * G + gl) * V;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int dd = d0 + v;
                    const double m = dd < D ? tp0[dd] : 0.0;
                    const double w = dd < D ? tp1[dd] : 0.0;
                    const double d = q.x[c][v] - m;
                    acc = fma(w * d, d, acc);
                }
            }
        }
        return -0.5 * group_sum<G>(acc);
    }
    if (target == TGT_ROSEN) {
        // sum_{d < D-1} 100 (x_{d+1} - x_d^2)^2 + (1 - x_d)^2   
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            // first element of the next lane / of the next chunk (held by the group's first lane)
            double nxt_lane, nxt_chunk = 0.0;
            if constexpr (G <= 16) {
                nxt_lane = dpp_f64<0x101>(q.x[c][0]);                                   // row_shl:1  -> lane + 1
                if (c + 1 < CH) nxt_chunk = dpp_f64<0x110 + (G - 1)>(q.x[c + 1 < CH ? c + 1 : c][0]);   // row_shr:G-1 -> lane - (G-1)
            } else {
                nxt_lane = __shfl(q.x[c][0], lane + 1, 64);
                if (c + 1 < CH) nxt_chunk = __shfl(q.x[c + 1 < CH ? c + 1 : c][0], lane - gl, 64);
            }
            const double nxt = (gl == G - 1) ? nxt_chunk : nxt_lane;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int d = (c * G + gl) * V + v;
                const double xn = (v + 1 < V) ? q.x[c][v + 1 < V ? v + 1 : v] : nxt;
                if (d + 1 < D) {
                    const double a1 = xn - q.x[c][v] * q.x[c][v];
                    const double b1 = 1.0 - q.x[c][v];
                    acc = fma(100.0 * a1, a1, acc);
                    acc = fma(b1, b1, acc);
                }
            }
        }
        return -group_sum<G>(acc) / tscale;
    }
    if (target == TGT_BOX) {
        // tests/integration/test_proposal.py:25-28 uniform_log_prob, all dims in [0, 1]
        double bad = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int d = (c * G + gl) * V + v;
                if (d < D && (q.x[c][v] > 1.0 || q.x[c][v] < 0.0)) bad = 1.0;
            }
        return group_sum<G>(bad) > 0.0 ? -__builtin_inf() : 0.0;
    }
    return 0.0;
}

// ----------------------------------------------------------------------------------------
// The half-step kernel.
//   G lanes per walker row, V doubles per lane per chunk, CH chunks, MOVE,
//   DPB = Dp/16 for the dense-Gaussian (MFMA) target, 0 for element-wise targets.
// A wave prefetches the rows of PF passes (PF * 64/G walkers) before touching any of them, so a
// batch costs one memory round trip instead of PF.
// Dynamic LDS (dense only): Sfrag[non-zero blocks of L] (MFMA B-fragment order) | mu[Dp] | per wave: tile[16][Dp+2], qf[16], fac[16]
// ----------------------------------------------------------------------------------------
// The LDS image of a dense Gaussian target (fused kernel, k_small_run): only the non-zero 16 x 16 blocks of the lower
// triangular factor L, column block by column block, k blocks from the diagonal down --
//   block(nb, kb), kb >= nb:  img[((block * 4 + i) * 64) + l] = L[k = 16 kb + 4 i + (l >> 4)][n = 16 nb + (l & 15)]
// followed by the zero-padded mean.  (B (B + 1) / 2 blocks of 2 KB instead of B^2: 20 instead of 32 KB at ndim 64, 56 instead
// of 98 KB at ndim 112, which is what decides how many waves of tiles fit next to it.)
__host__ __device__ constexpr int dense_block(int B, int nb, int kb) { return nb * B - nb * (nb - 1) / 2 + (kb - nb); }
__host__ __device__ constexpr int dense_img_doubles(int Dp) { return (Dp / 16) * (Dp / 16 + 1) / 2 * 256; }

#define EMX_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// Gaussian Metropolis move (moves/gaussian.py, moves/mh.py): every walker is its own slot; p0 carries the
// coordinate that moves (-1: all of them), uacc the accept uniform.  Same Philox streams as native_slot.
__host__ __device__ inline void native_gauss_slot(const NativeArgs& na, int D, int mode, int seqcol, int t, int& i,
                                                  int& p0, int& p1, int& p2, double& s0, double& uacc) {
    const uint32_t k0 = (uint32_t)na.seed, k1 = (uint32_t)(na.seed >> 32);
    const uint32_t sl = (uint32_t)na.step, sh = (uint32_t)(na.step >> 32);
    i = t;
    p1 = p2 = t;
    s0 = 0.0;
    const Philox4 A = philox4x32_10((uint32_t)t, 0u, sl, sh, k0, k1);
    uacc = u53(A.v[2], A.v[3]);
    if (mode == GAUSS_RANDOM)
        p0 = (int)bounded64(A.v[0], A.v[1], (uint64_t)D);
    else
        p0 = mode == GAUSS_SEQUENTIAL ? seqcol : -1;
}

// Noise of the native Gaussian Metropolis proposal.  Pair p (coordinates 2p, 2p + 1) of walker w is one Box-Muller pair
// made from HALF a Philox4x32-7 block: block (w, 2 + (p >> 1)), words (2h, 2h + 1) with h = p & 1 -- a 32-bit uniform for
// the radius (converted to f32: small values keep all their bits, so the tail reaches 6.7 sigma) and a 24-bit one for the
// direction, all in f32 hardware transcendentals (v_log / v_sqrt / v_sin / v_cos).  A Metropolis proposal only has to be
// symmetric for the chain to be exact, which any Box-Muller pair is (the direction is uniform); round count and
// precision affect nothing but the cost: 7 instead of 10 rounds, two pairs per block, no f64 logarithm.
// One definition for every path that needs a normal.
__device__ __forceinline__ void gauss_from_words(uint32_t a, uint32_t b, double& n0, double& n1) {
    const float u = ((float)a + 0.5f) * 2.3283064365386963e-10f;            // (0, 1]
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));   // sqrt(-2 ln u), ln u = ln 2 * log2 u
    const float rev = (float)(b >> 8) * 5.9604644775390625e-8f;             // [0, 1) revolutions, 24 bits: exact in f32
    n0 = (double)(r * __builtin_amdgcn_cosf(rev));
    n1 = (double)(r * __builtin_amdgcn_sinf(rev));
}

__device__ __forceinline__ Philox4 gauss_block(uint64_t seed, uint64_t step, int w, int blk) {
    return philox4x32<7>((uint32_t)w, 2u + (uint32_t)blk, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

__device__ __forceinline__ void native_gauss_pair(uint64_t seed, uint64_t step, int w, int pair, double& n0, double& n1) {
    const Philox4 R = gauss_block(seed, step, w, pair >> 1);
    const int h = pair & 1;
    gauss_from_words(h ? R.v[2] : R.v[0], h ? R.v[3] : R.v[1], n0, n1);
}

template <int MOVE>
constexpr int rows_per_pass() {
    return (MOVE == MOVE_STRETCH || MOVE == MOVE_GAUSS) ? 2 : MOVE == MOVE_DE ? 3 : MOVE == MOVE_SNOOKER ? 4 : 1;
}

template <int G, int V, int CH, int MOVE, int DPB>
constexpr int prefetch_depth() {
    constexpr int WPW = 64 / G;
    // <= 48 doubles of rows in flight per lane.  (Round 3: 64 for the snooker move -- whose four rows per walker split a 16-row
    // MFMA tile at ndim 64 into two dependent memory round trips -- was measured: one round trip, but 254 VGPRs and a spill,
    // C4 33.55 -> 34.15 us/step; profiles/r03/ab_snooker_budget.txt.  EMX_SNOOKER_BUDGET stays for the record.)
    int pf = (MOVE == MOVE_SNOOKER ? EMX_SNOOKER_BUDGET : 48) / (rows_per_pass<MOVE>() * CH * V);
    pf = pf < 1 ? 1 : (pf > 8 ? 8 : pf);
    int p2 = 1;
    while (p2 * 2 <= pf) p2 *= 2;
    if (DPB > 0 && p2 > 16 / WPW) p2 = 16 / WPW;           // dense: a batch never spans two 16-row MFMA tiles
    return p2 < G ? p2 : G;                                // at most 64 slots per batch
}

// native Gaussian move: this lane's part of walker w's displacement row, (f * scale_d) * n(w, d)
// ARGS: anything with gseed, gstep, gfac, gsigma, gscale (HalfStepArgs; GaussGen in the one-workgroup kernel)
template <int G, int V, int CH, typename ARGS>
__device__ __forceinline__ void gauss_disp_row(Row<G, V, CH>& t, const ARGS& A, int w, int col, int D, int gl) {
    if (col >= 0) {
        // one-coordinate modes (launch-uniform branch): a single normal per walker, computed once by every lane
        // of the walker's group instead of once per chunk
        double n0, n1;
        native_gauss_pair(A.gseed, A.gstep, w, col >> 1, n0, n1);
        const double val = (A.gfac * (A.gscale ? A.gscale[col] : A.gsigma)) * ((col & 1) ? n1 : n0);
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) t.x[c][v] = ((c * G + gl) * V + v == col) ? val : 0.0;
        return;
    }
    if constexpr (V == 2 && (CH % 2) == 0) {
        // Lanes gl and gl ^ 1 hold pairs c G + gl and c G + (gl ^ 1): the two halves of ONE block.  Of every two chunks the
        // even lane computes the block of the first, the odd lane the block of the second, and they swap the halves they
        // do not need (DPP quad_perm [1,0,3,2]): one Philox block per lane per two chunks instead of two.
        const int b = gl & 1;
#pragma unroll
        for (int c2 = 0; c2 < CH; c2 += 2) {
            const int cm = c2 + b;                                 // the chunk whose block this lane computes
            const Philox4 R = gauss_block(A.gseed, A.gstep, w, (cm * G + gl) >> 1);
            // my half of my block serves chunk cm; the other half is the neighbour's for the same chunk
            const uint32_t keep0 = b ? R.v[2] : R.v[0], keep1 = b ? R.v[3] : R.v[1];
            const uint32_t give0 = b ? R.v[0] : R.v[2], give1 = b ? R.v[1] : R.v[3];
            const uint32_t got0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give0, 0xB1, 0xf, 0xf, false);
            const uint32_t got1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give1, 0xB1, 0xf, 0xf, false);
            // chunk c2 + b from the kept words, chunk c2 + (1 - b) from the received ones
            double k0, k1, r0, r1;
            gauss_from_words(keep0, keep1, k0, k1);
            gauss_from_words(got0, got1, r0, r1);
            const double e0 = b ? r0 : k0, e1 = b ? r1 : k1;       // chunk c2     (even chunk of the pair)
            const double o0 = b ? k0 : r0, o1 = b ? k1 : r1;       // chunk c2 + 1
            const int de = 2 * (c2 * G + gl), dod = 2 * ((c2 + 1) * G + gl);
            t.x[c2][0] = de < D ? (A.gfac * (A.gscale ? A.gscale[de] : A.gsigma)) * e0 : 0.0;
            t.x[c2][1] = de + 1 < D ? (A.gfac * (A.gscale ? A.gscale[de + 1] : A.gsigma)) * e1 : 0.0;
            t.x[c2 + 1][0] = dod < D ? (A.gfac * (A.gscale ? A.gscale[dod] : A.gsigma)) * o0 : 0.0;
            t.x[c2 + 1][1] = dod + 1 < D ? (A.gfac * (A.gscale ? A.gscale[dod + 1] : A.gsigma)) * o1 : 0.0;
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if constexpr (V == 2) {
            const int pr = c * G + gl, d0 = 2 * pr;
            t.x[c][0] = t.x[c][1] = 0.0;
            if (d0 < D) {
                double n0, n1;
                native_gauss_pair(A.gseed, A.gstep, w, pr, n0, n1);
                t.x[c][0] = (A.gfac * (A.gscale ? A.gscale[d0] : A.gsigma)) * n0;
                if (d0 + 1 < D) t.x[c][1] = (A.gfac * (A.gscale ? A.gscale[d0 + 1] : A.gsigma)) * n1;
            }
        } else {
            const int d = c * G + gl;
            t.x[c][0] = 0.0;
            if (d < D) {
                double n0, n1;
                native_gauss_pair(A.gseed, A.gstep, w, d >> 1, n0, n1);
                t.x[c][0] = (A.gfac * (A.gscale ? A.gscale[d] : A.gsigma)) * ((d & 1) ? n1 : n0);
            }
        }
    }
}

// proposal from the prefetched rows; rounding order as in stretch.py:33 / de.py:53-62 / de_snooker.py:41-46
template <int G, int V, int CH, int MOVE>
__device__ __forceinline__ void make_proposal(const Row<G, V, CH>& xi, const Row<G, V, CH>& xa,
                                              const Row<G, V, CH>& xb, const Row<G, V, CH>& xc, double s0,
                                              double gammas, int D, int gl, Row<G, V, CH>& q, double& factor,
                                              int col = -1) {
    if constexpr (MOVE == MOVE_GAUSS) {
        // xa = this walker's displacement row (factor * scale * normal, gaussian.py:87); col >= 0: only
        // that coordinate moves (gaussian.py:92-101), the others keep their exact value
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int d = (c * G + gl) * V + v;
                const double moved = xi.x[c][v] + xa.x[c][v];   // x0 + ...
                q.x[c][v] = (col < 0 || d == col) ? moved : xi.x[c][v];
            }
    } else if constexpr (MOVE == MOVE_STRETCH) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double diff = xa.x[c][v] - xi.x[c][v];   // c[rint] - s
                const double prod = diff * s0;                  // ... * zz
                q.x[c][v] = xa.x[c][v] - prod;                  // c[rint] - (...)
            }
    } else if constexpr (MOVE == MOVE_DE) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double diff = xb.x[c][v] - xa.x[c][v];   // c[pair[1]] - c[pair[0]]
                const double prod = s0 * diff;                  // gamma * diffs
                q.x[c][v] = xi.x[c][v] + prod;                  // s + ...
            }
    } else if constexpr (MOVE == MOVE_SNOOKER) {
        // xa = z, xb = z1, xc = z2.  de_snooker.py:41-45: delta = s - z, u = delta / |delta|, q = s + u gammas (u.z1 - u.z2),
        // metropolis = ln|q - z| - ln|delta|.  ONE pass and one round of group reductions (round 5; three dependent ones before): the
        // three sums |delta|^2, delta.z1, delta.z2 are taken together -- u.z = (delta.z) / |delta| -- and |q - z| needs none:
        // q - z = u (|delta| + gammas (u.z1 - u.z2)) with |u| = 1.  Same quantities as the reference's, rounded differently in the
        // last place (the snooker move's coordinates were never bit-equal to NumPy's pairwise sums: tolerance 1e-9, accept masks equal).
        double n2 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double dl = xi.x[c][v] - xa.x[c][v];     // delta = s[i] - z
                n2 = fma(dl, dl, n2);
                e1 = fma(dl, xb.x[c][v], e1);
                e2 = fma(dl, xc.x[c][v], e2);
            }
        n2 = group_sum<G>(n2);
        e1 = group_sum<G>(e1);
        e2 = group_sum<G>(e2);
        const double norm = sqrt(n2);
        const double dd = e1 / norm - e2 / norm;                // dot(u, z1) - dot(u, z2)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double u = (xi.x[c][v] - xa.x[c][v]) / norm;
                const double ug = u * gammas;                   // u * gammas
                const double prod = ug * dd;                    // * (dot(u,z1) - dot(u,z2))
                const int d = (c * G + gl) * V + v;
                q.x[c][v] = d < D ? xi.x[c][v] + prod : 0.0;
            }
        const double nq = fabs(norm + gammas * dd);             // |q - z|
        factor = ((double)D - 1.0) * (log(nq) - log(norm));
    } else {
        q = xi;
    }
}

// LEAN: the production instantiation of the shapes the bench configurations use.  The host selects it when none of the
// occasional features is in play (sharded send buffers, device-side slot counts, graph replay descriptors, materialised
// Gaussian displacements, peers, timing experiments): those kernel arguments then fold to constants instead of sitting in
// scalar registers for the whole kernel -- the full kernel spills 66 SGPRs, and 25 fewer spills were worth 1.3 % at C2.
// LEAN = 3 / 4 (round 4: the headline shape only) split LEAN = 2 by exchange -- 3: the device-side slot count alone (pull), 4: slot
// count + decision output (replay), 2: everything (direct: the peer table) -- because LEAN = 2 still spilled 39 scalar registers.
// LEAN = 2 keeps what the block-ownership exchanges need (the device-side slot count of the compact plan, the peer
// table of the direct exchange) and folds the rest: the sharded stretch runs of the same shapes.
template <int G, int V, int CH, int MOVE, int DPB, int LEAN = 0>
static __global__ __launch_bounds__(512) void k_halfstep(const HalfStepArgs A) {
    const int ablate_ = (EMX_EXPERIMENTS && !LEAN) ? A.ablate : 0;
    const StepDesc* const desc_ = LEAN ? nullptr : A.desc;
    double* const sendbuf_ = LEAN ? nullptr : A.sendbuf;
    const int32_t* const thidev_ = LEAN == 1 ? nullptr : A.t_hi_dev;        // LEAN 2: the block-ownership exchanges (pull, direct)
    const double* const disp_ = LEAN ? nullptr : A.disp;
    const int skewsl_ = LEAN ? 0 : A.skew_sleep;
    double* const declp_ = (LEAN == 0 || LEAN == 4) ? A.declp : nullptr;   // LEAN 4: + the replay exchange's own pass
    auto put_decision = [&](int idx, double v) {
        declp_[idx] = v;
        if (A.npush) {
            const PeerTable* __restrict__ T = A.push_peers;
#pragma nounroll
            for (int q = 0; q < A.npush; ++q)                     // (rolled: eight peer pointers held in scalar registers across the
                const_cast<double*>(T->X[q])[A.push_off + idx] = v;   // batch loop were half of the LEAN instantiations' spills)
        }
    };
    const int target_ = (LEAN && DPB > 0) ? (int)TGT_DENSE : A.target;
    static_assert(G >= 4 && G <= 64 && (64 % G) == 0, "G lanes per walker");
    constexpr bool DENSE = DPB > 0;
    constexpr int WPW = 64 / G;       // walkers per pass (<= 16)
    constexpr int PPT = 16 / WPW;     // passes per 16-row dense tile
    constexpr int PF = prefetch_depth<G, V, CH, MOVE, DPB>();
    constexpr int NR = rows_per_pass<MOVE>();
    constexpr int Dp = DPB * 16;      // dense: padded dimension
    constexpr int KK = Dp / 4;        // MFMA k-steps
    constexpr int RT = Dp + 2;        // tile row stride (doubles): conflict-free A-fragment reads
    constexpr bool RTILE = EMX_OPT_RTILE && DENSE && PF == PPT && MOVE != MOVE_EVAL;
    static_assert(!DENSE || G * V * CH >= Dp, "row layout must cover the padded dimension");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (ablate_ & 64) return;     // timing experiments: launch + dispatch floor
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int sub = lane / G;
    const int gl = lane % G;
    const int D = (LEAN && EMX_LEAN_FOLD_D) ? G * V * CH : A.D;          // folding the dimension too was measured 1.7 % SLOWER at C2 (scheduling)
    double* chain_ = A.chain;
    double* chain_lp_ = A.chain_lp;
    if (desc_) {      // hipGraph replay: this step's chain row comes from the device-side descriptor
        const long long sidx_ = desc_->stored_idx;
        chain_ = sidx_ >= 0 ? A.chain_all + (size_t)sidx_ * A.N * D : nullptr;
        chain_lp_ = sidx_ >= 0 ? A.chain_lp_all + (size_t)sidx_ * A.N : nullptr;
    }

    double* Sfrag = smem;
    double* muS = smem + dense_img_doubles(Dp);
    double* tile = muS + Dp + (size_t)wib * (16 * RT + 32);
    double* qfS = tile + 16 * RT;
    double* facS = qfS + 16;

    // Dense target: A.tp1 holds the image the MFMA stage wants in LDS -- the Cholesky factor L of the
    // precision matrix (icov = L L^T) in B-fragment order, zero padded, followed by the padded mean:
    //   the non-zero 16 x 16 blocks (dense_block above), then mu
    // (built once by emx_set_target).  Its global loads are issued first and written to LDS only after
    // the first batch's row loads are in flight; one workgroup barrier precedes the first MFMA stage.
    constexpr int IMG2 = (dense_img_doubles(Dp) + Dp) / 2;  // image size in double2
    constexpr int NSTG = 5;                                 // double2 per thread held in registers (first round; 9 = the whole image of a skewed stager in one round was measured 2.7 % slower)
    double2 stg0, stg1, stg2, stg3, stg4, stg5, stg6, stg7, stg8;
    stg0 = stg1 = stg2 = stg3 = stg4 = stg5 = stg6 = stg7 = stg8 = double2{0.0, 0.0};
#define EMX_IMAGE_LOADS()                                                                          \
    do {                                                                                           \
        const double2* img_ = reinterpret_cast<const double2*>(A.tp1);                             \
        const int bs_ = stg_bs, tx_ = stg_tx;                                                      \
        if (tx_ < IMG2) stg0 = img_[tx_];                                                          \
        if (tx_ + bs_ < IMG2) stg1 = img_[tx_ + bs_];                                              \
        if (tx_ + 2 * bs_ < IMG2) stg2 = img_[tx_ + 2 * bs_];                                      \
        if (tx_ + 3 * bs_ < IMG2) stg3 = img_[tx_ + 3 * bs_];                                      \
        if (tx_ + 4 * bs_ < IMG2) stg4 = img_[tx_ + 4 * bs_];                                      \
        if constexpr (NSTG > 5) {                                                                  \
            if (tx_ + 5 * bs_ < IMG2) stg5 = img_[tx_ + 5 * bs_];                                  \
            if (tx_ + 6 * bs_ < IMG2) stg6 = img_[tx_ + 6 * bs_];                                  \
            if (tx_ + 7 * bs_ < IMG2) stg7 = img_[tx_ + 7 * bs_];                                  \
            if (tx_ + 8 * bs_ < IMG2) stg8 = img_[tx_ + 8 * bs_];                                  \
        }                                                                                          \
    } while (0)
    // EMX_OPT_SKEW: in an 8-wave workgroup the image is staged by waves 4-7 alone, BEFORE their own row loads; waves 0-3 go
    // straight to their loads.  The two waves that share a SIMD (w, w + 4) then run about one image latency out of phase:
    // while one is in its MFMA chain the other is still loading / proposing instead of queueing for the same pipe.
    const bool skew = EMX_OPT_SKEW && DENSE && blockDim.x == 512;
    const bool stager = !skew || (threadIdx.x >> 6) >= 4;
    const int stg_bs = skew ? 256 : (int)blockDim.x;
    const int stg_tx = skew ? (stager ? (int)threadIdx.x - 256 : (1 << 30)) : (int)threadIdx.x;      // non-stagers: beyond the image
    if constexpr (DENSE) {
        // The image loads are the FIRST memory operations of the kernel: vector-memory loads return in
        // order, so they land before the (slower, bandwidth-bound) row loads issued below and the
        // workgroup barrier that publishes the image does not wait for any row.
        EMX_IMAGE_LOADS();
    }
    // per-lane diag-Gaussian parameters (narrow rows): same columns for every walker of the wave
    Row<G, V, CH> mu, iv;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) mu.x[c][v] = iv.x[c][v] = 0.0;
    if (!DENSE && CH <= 4 && target_ == TGT_DIAG) {
        load_row<G, V, CH>(mu, A.tp0, D, gl);
        load_row<G, V, CH>(iv, A.tp1, D, gl);
    }
    if constexpr (RTILE) load_row<G, V, CH>(mu, A.tp0, D, gl);        // the mean in the row layout: the tile receives q - mu

    const int wave = blockIdx.x * (blockDim.x >> 6) + wib;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
#define EMX_STAMP(k_)                                                                              \
    do {                                                                                           \
        if (EMX_OPT_STAMPS && A.dbg && wib == 0) {                                                                   \
            const unsigned long long t_ = __builtin_readcyclecounter();                            \
            if (lane == 0) A.dbg[(size_t)blockIdx.x * 16 + (k_)] = t_;                             \
        }                                                                                          \
    } while (0)
    EMX_STAMP(0);
    if (EMX_OPT_STAMPS && A.dbg && wib == 0 && lane == 0) A.dbg[(size_t)blockIdx.x * 16 + 11] = wall_clock64();   // 100 MHz reference
    const int spw = (LEAN && EMX_LEAN_FOLD_SPW) ? ((DENSE && PF * WPW < 16) ? 16 : PF * WPW) : A.spw;
    const int tlo_ = (LEAN && EMX_LEAN_FOLD_SPW) ? 0 : A.t_lo;
    bool stage_pending = DENSE;   // this wave still owes its share of the image and the workgroup barrier
#define EMX_STAGE_PUBLISH()                                                                         \
    do {                                                                                            \
        if constexpr (DENSE) {                                                                      \
            double2* dst_ = reinterpret_cast<double2*>(smem);                                       \
            const int bs_ = stg_bs, tx_ = stg_tx;                                                   \
            if (tx_ < IMG2) dst_[tx_] = stg0;                                                       \
            if (tx_ + bs_ < IMG2) dst_[tx_ + bs_] = stg1;                                           \
            if (tx_ + 2 * bs_ < IMG2) dst_[tx_ + 2 * bs_] = stg2;                                   \
            if (tx_ + 3 * bs_ < IMG2) dst_[tx_ + 3 * bs_] = stg3;                                   \
            if (tx_ + 4 * bs_ < IMG2) dst_[tx_ + 4 * bs_] = stg4;                                   \
            if constexpr (NSTG > 5) {                                                               \
                if (tx_ + 5 * bs_ < IMG2) dst_[tx_ + 5 * bs_] = stg5;                               \
                if (tx_ + 6 * bs_ < IMG2) dst_[tx_ + 6 * bs_] = stg6;                               \
                if (tx_ + 7 * bs_ < IMG2) dst_[tx_ + 7 * bs_] = stg7;                               \
                if (tx_ + 8 * bs_ < IMG2) dst_[tx_ + 8 * bs_] = stg8;                               \
            }                                                                                       \
            for (int e_ = tx_ + NSTG * bs_; e_ < IMG2; e_ += bs_)                                   \
                dst_[e_] = reinterpret_cast<const double2*>(A.tp1)[e_];   /* very wide targets */   \
            __syncthreads();                                                                        \
            stage_pending = false;                                                                  \
        }                                                                                           \
    } while (0)
    const int t_hi = thidev_ ? *thidev_ : A.t_hi;
    if (tlo_ + wave * spw >= t_hi) {        // idle wave: publish its share, meet the barrier, leave
        EMX_STAGE_PUBLISH();
        return;
    }
    for (int t0 = tlo_ + wave * spw; t0 < t_hi; t0 += nwaves * spw) {     // wave-uniform batch loop
        const int nslot = min(spw, t_hi - t0);
        const int npass = (nslot + WPW - 1) / WPW;
        const int pbase = A.pos0 + t0;          // plan position of the wave's first slot

        for (int pb = 0; pb < npass; pb += PF) {
            if (skew && stager && stage_pending) {                       // image first, own loads after the barrier
                EMX_STAGE_PUBLISH();
                for (int r_ = 0; r_ < skewsl_; ++r_) __builtin_amdgcn_s_sleep(8);
            }
            // -------- plan entries of the batch: every lane of a group reads its walker's entry
            //          (same address across the group: one request, broadcast) --------
            int wi[PF], ja[PF], jb[NR >= 3 ? PF : 1], jc[NR >= 4 ? PF : 1];
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int srow = (pb + k) * WPW + sub;
                const int pos = pbase + (srow < nslot ? srow : 0);
                wi[k] = A.order[pos];
                if constexpr (NR >= 2) ja[k] = A.p0[pos];
                if constexpr (NR >= 3) jb[k] = A.p1[pos];
                if constexpr (NR >= 4) jc[k] = A.p2[pos];
            }
            EMX_STAMP(1);      // plan loads issued
            if (EMX_OPT_STAMPS && A.dbg) {       // instrumented runs only: when do the plan entries actually arrive?
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                EMX_STAMP(13);
            }
            // -------- issue every row load of the batch (+ the per-walker scalars) --------
            Row<G, V, CH> xi[PF], xa[NR >= 2 ? PF : 1], xb[NR >= 3 ? PF : 1], xc[NR >= 4 ? PF : 1];
            Row<G, V, CH> qk[RTILE ? PF : 1];          // RTILE: the proposals stay in registers until the commit
            double s0v[PF], facv[PF], lpov[PF], loguv[PF];
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int srow = (pb + k) * WPW + sub;
                const int pos = pbase + (srow < nslot ? srow : 0);
                if (!(ablate_ & 32)) load_row<G, V, CH>(xi[k], A.X + (size_t)wi[k] * D, D, gl);
                if constexpr (MOVE == MOVE_GAUSS) {
                    if (disp_) load_row<G, V, CH>(xa[k], disp_ + (size_t)wi[k] * D, D, gl);
                } else if constexpr (NR >= 2) {
                    if (!(ablate_ & 32)) load_row<G, V, CH>(xa[k], partner_base<LEAN>(A, ja[k]) + (size_t)ja[k] * D, D, gl);
                }
                if constexpr (NR >= 3) load_row<G, V, CH>(xb[k], partner_base<LEAN>(A, jb[k]) + (size_t)jb[k] * D, D, gl);
                if constexpr (NR >= 4) load_row<G, V, CH>(xc[k], partner_base<LEAN>(A, jc[k]) + (size_t)jc[k] * D, D, gl);
                if constexpr (MOVE != MOVE_EVAL) {
                    s0v[k] = (MOVE == MOVE_SNOOKER) ? 0.0 : A.s0[pos];
                    facv[k] = A.fac[pos];
                    if (!DENSE || target_ == TGT_NONE) {
                        lpov[k] = A.lp[wi[k]];
                        loguv[k] = A.logu[pos];
                    } else {
                        lpov[k] = 0.0;
                        loguv[k] = 0.0;
                    }
                } else {
                    s0v[k] = facv[k] = lpov[k] = loguv[k] = 0.0;
                }
            }

            EMX_STAMP(2);      // row loads issued (the plan entries have arrived)
            if (stage_pending) EMX_STAGE_PUBLISH();   // rows are in flight; the barrier only waits for the image
            EMX_STAMP(3);      // image published, workgroup barrier passed

            if (ablate_ & 128) {          // timing experiments: consume the loads, skip everything else
                double sink = 0.0;
#pragma unroll
                for (int k = 0; k < PF; ++k) sink += xi[k].x[0][0] + xa[NR >= 2 ? k : 0].x[0][0] + lpov[k];
                if (sink == 1.2345e-300) A.fout[0] = sink;
                continue;
            }
            // -------- proposals (+ element-wise target, decision, commit) --------
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int p = pb + k;
                if (p < npass) {                                   // wave-uniform
                    const int srow = p * WPW + sub;                // slot of this group inside the wave's range
                    const bool live = srow < nslot;
                    const int i = wi[k];
                    double factor = facv[k];

                    Row<G, V, CH> q;
                    if constexpr (MOVE == MOVE_GAUSS)
                        if (!disp_) gauss_disp_row<G, V, CH>(xa[k], A, i, ja[k], D, gl);
                    make_proposal<G, V, CH, MOVE>(xi[k], xa[NR >= 2 ? k : 0], xb[NR >= 3 ? k : 0], xc[NR >= 4 ? k : 0],
                                                  s0v[k], A.gammas, D, gl, q, factor, NR >= 2 ? ja[k] : -1);

                    // non-finite proposal -> sticky error (ensemble.py:476-479); the proposal is rejected
                    bool badq = false;
                    if constexpr (MOVE != MOVE_EVAL) {
                        bool bl = false;
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int v = 0; v < V; ++v) bl |= !(fabs(q.x[c][v]) <= 1.79769313486231570815e308);
                        badq = group_any<G>(bl, sub);
                        if (live && badq && gl == 0) raise_status(A.status, ST_BAD_COORD);
                    }

                    if (target_ == TGT_NONE) {
                        // split-phase: hand the proposal to the host log-prob (red_blue.py:90-93)
                        if (live) {
                            const int t = t0 + srow;
                            store_row<G, V, CH>(q, A.qout + (size_t)t * D, D, gl);
                            if (gl == 0) A.fout[t] = factor;
                        }
                    } else if (LEAN != 1 && !DENSE && target_ == TGT_REPLAY) {      // (a replay launch reads its slot count on the device: never LEAN 1)
                        // a peer's accepted update, recomputed from the replica: the same rows, the same draws, the same
                        // instructions as on the rank that took the decision, hence the same bits (move.py:33-34)
                        if constexpr (MOVE != MOVE_EVAL) {
                            if (live) {
                                store_row<G, V, CH>(q, A.X + (size_t)i * D, D, gl);
                                if (gl == 0) A.lp[i] = loguv[k];         // the plan's logu column carries the new log-prob
                            }
                        }
                    } else if constexpr (!DENSE) {
                        const double lp_new = eval_valu_target<G, V, CH>(q, mu, iv, A.tp0, A.tp1, target_, A.tscale, D, gl, lane);
                        if (live && gl == 0 && (lp_new != lp_new)) raise_status(A.status, ST_NAN_LOGP);   // ensemble.py:550-551
                        if constexpr (MOVE == MOVE_EVAL) {
                            if (live && gl == 0) A.lp[i] = lp_new;
                        } else {
                            const double lp_old = lpov[k];
                            const double lnpdiff = factor + lp_new - lp_old;                  // red_blue.py:99
                            const bool accept = live && !badq && (lnpdiff > loguv[k]);        // red_blue.py:100
                            if (accept) {
                                store_row<G, V, CH>(q, A.X + (size_t)i * D, D, gl);          // move.py:33
                                if (gl == 0) A.lp[i] = lp_new;                                // move.py:34
                            }
                            if (live && gl == 0) {
                                A.acc[i] = accept ? 1 : 0;
                                if (chain_lp_) {
                                    chain_lp_[i] = accept ? lp_new : lp_old;
                                    if (accept) A.acc_count[i] += 1u;
                                }
                                if (declp_) put_decision(t0 + srow - tlo_, accept ? lp_new : __builtin_nan(""));
                            }
                            if (live && chain_) store_row_stream<G, V, CH>(accept ? q : xi[k], chain_ + (size_t)i * D, D, gl);
                            if (live && sendbuf_) {
                                double* sb = sendbuf_ + (size_t)(t0 + srow - tlo_) * (D + 2);
                                store_row<G, V, CH>(accept ? q : xi[k], sb, D, gl);
                                if (gl == 0) {
                                    sb[D] = accept ? lp_new : lp_old;
                                    sb[D + 1] = accept ? 1.0 : 0.0;
                                }
                            }
                        }
                    } else {
                        // dense: the proposal goes to the wave's LDS tile (MFMA A operand, and commit source)
                        const int trow = srow & 15;
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int v = 0; v < V; ++v) {
                                const int d = (c * G + gl) * V + v;
                                if constexpr (RTILE) {
                                    if (d < Dp && !(ablate_ & 4)) tile[trow * RT + d] = (live && !badq) ? q.x[c][v] - mu.x[c][v] : 0.0;
                                } else {
                                    if (d < Dp && !(ablate_ & 4)) tile[trow * RT + d] = (live && !badq) ? q.x[c][v] : muS[d];   // dead row: zero residual
                                }
                            }
                        if constexpr (RTILE) qk[k] = q;
                        if (gl == 0) facS[trow] = badq ? -__builtin_inf() : factor;
                        // stored step / sharded run: the current row goes out now (fire and forget); an accepted
                        // proposal overwrites it after the decision -- no reload of rejected rows in the commit
                        if constexpr (MOVE != MOVE_EVAL) {
                            if (live && chain_) store_row_stream<G, V, CH>(xi[k], chain_ + (size_t)i * D, D, gl);
                            if (live && sendbuf_)
                                store_row<G, V, CH>(xi[k], sendbuf_ + (size_t)(t0 + srow - tlo_) * (D + 2), D, gl);
                        }
                    }
                }
            }

            EMX_STAMP(4);      // proposals done, tile rows written (all row loads consumed)
            if constexpr (DENSE) {
                const int plast = (pb + PF < npass ? pb + PF : npass) - 1;          // last pass of this batch
                const bool tile_done = ((plast + 1) % PPT == 0) || (plast + 1 == npass);
                if (target_ != TGT_NONE && tile_done && !(ablate_ & 16)) {
                    // ---- one 16-row tile: Y = R Sinv by v_mfma_f64_16x16x4_f64 (R = Q - mu), qf[w] = sum_n Y[w][n] R[w][n] ----
                    const int tb = (plast / PPT) * 16;                  // first slot of the tile
                    // decision lanes: the 16 lanes with (lane & 15) < 4.  After the row reduction lane (am, ak) holds the
                    // totals of tile rows ak + 4 r, so lane (am = r, ak) decides row ak + 4 r from its own registers
                    // (no LDS round trip).  Their walker and decision scalars are L1/L2-hot lines.
                    const int myrow = (lane >> 4) + 4 * (lane & 3);
                    const bool mine = (lane & 15) < 4 && tb + myrow < nslot;
                    const int mypos = pbase + (tb + myrow < nslot ? tb + myrow : 0);
                    const int my_i = A.order[mypos];
                    double my_lpo = 0.0, my_logu = 0.0;
                    if constexpr (MOVE != MOVE_EVAL) {
                        my_lpo = A.lp[my_i];
                        my_logu = A.logu[mypos];
                    }
                    EMX_WAVE_SYNC();                     // this wave's tile rows are visible to all of its lanes
                    double my_qf = 0.0;
                    {
                        // Y = R L (R = Q - mu, 16 x Dp) by v_mfma_f64_16x16x4_f64; L is lower triangular, so the
                        // k-steps below the diagonal block of column block nb vanish; qf[w] = sum_n Y[w][n]^2
                        const int am = lane & 15, ak = lane >> 4;
                        typedef double d4 __attribute__((ext_vector_type(4)));
                        double afr[KK];
#pragma unroll
                        for (int kk = 0; kk < KK; ++kk) {                                        // A[i = lane&15][k = lane>>4]
                            if constexpr (RTILE)
                                afr[kk] = tile[am * RT + 4 * kk + ak];
                            else
                                afr[kk] = tile[am * RT + 4 * kk + ak] - muS[4 * kk + ak];
                        }
                        double part[4] = {0.0, 0.0, 0.0, 0.0};
                        if (EMX_OPT_STAMPS && A.dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
                        EMX_STAMP(5);      // A fragments in registers
                        if (!(ablate_ & 1))
#pragma unroll
                        for (int nb = 0; nb < DPB; ++nb) {
                            d4 accv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int kk = 4 * nb; kk < KK; ++kk)
                                accv = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[kk], Sfrag[(dense_block(DPB, nb, kk >> 2) * 4 + (kk & 3)) * 64 + lane], accv, 0, 0, 0);
                            // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
                            for (int r = 0; r < 4; ++r) part[r] = fma(accv[r], accv[r], part[r]);
                        }
                        if (EMX_OPT_STAMPS && A.dbg) { asm volatile("s_nop 0" ::: "memory"); }
                        EMX_STAMP(6);      // MFMA chain + squares done
                        // over the 16 columns held by this row of lanes: lane (am, ak) ends with the total of tile
                        // row ak + 4 (am & 3), which is what the decision lanes (am < 4) need
#if EMX_OPT_RED4
                        my_qf = row16_sum4(part[0], part[1], part[2], part[3], lane);
#else
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[r] = group_sum<16>(part[r]);
                        my_qf = part[0];
#pragma unroll
                        for (int r = 1; r < 4; ++r) my_qf = (am == r) ? part[r] : my_qf;
#endif
                    }
                    EMX_STAMP(7);      // reductions done
                    // ---- decisions for the (up to) 16 rows of this tile ----
                    bool acc = false;
                    double lp_fin = my_lpo;
                    if (mine) {
                        const double lpn = -0.5 * my_qf;
                        if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);
                        if constexpr (MOVE == MOVE_EVAL) {
                            A.lp[my_i] = lpn;
                        } else {
                            const double lnpdiff = facS[myrow] + lpn - my_lpo;
                            acc = lnpdiff > my_logu;
                            A.acc[my_i] = acc ? 1 : 0;
                            if (acc) {
                                A.lp[my_i] = lpn;
                                lp_fin = lpn;
                            }
                            if (chain_lp_) {
                                chain_lp_[my_i] = lp_fin;
                                if (acc) A.acc_count[my_i] += 1u;
                            }
                            if (declp_) put_decision(t0 + tb + myrow - tlo_, acc ? lpn : __builtin_nan(""));
                        }
                    }
                    EMX_STAMP(8);      // decisions made, flag / log-prob stores issued
                    if constexpr (MOVE != MOVE_EVAL) {
                        const unsigned long long am64 = __ballot(acc);       // bit (row & 3) * 16 + (row >> 2) <-> tile row
                        if (sendbuf_ && (lane & 15) < 4) qfS[myrow] = lp_fin;
                        if (sendbuf_) EMX_WAVE_SYNC();
                        // commit the tile's rows in the (G, V, CH) row layout: accepted rows come from LDS
#pragma unroll
                        for (int pp = 0; pp < PPT; ++pp) {
                            if (ablate_ & 8) break;
                            const int row = pp * WPW + sub;              // 0..15
                            const int sidx = tb + row;
                            const bool lv = sidx < nslot;
                            const bool ac = lv && ((am64 >> ((row & 3) * 16 + (row >> 2))) & 1ull);
                            if (!lv) continue;
                            if (sendbuf_ && gl == 0) {
                                double* sb = sendbuf_ + (size_t)(t0 + sidx - tlo_) * (D + 2);
                                sb[D] = qfS[row];
                                sb[D + 1] = ac ? 1.0 : 0.0;
                            }
                            if (!ac) continue;
                            // the walker index is still in registers when the batch is exactly this tile
                            int wi2;
                            if constexpr (PF == PPT) {
                                wi2 = wi[0];
#pragma unroll
                                for (int k2 = 1; k2 < PF; ++k2) wi2 = (pp == k2) ? wi[k2] : wi2;
                            } else {
                                wi2 = A.order[pbase + sidx];
                            }
                            Row<G, V, CH> rr;
                            if constexpr (RTILE) {
                                rr = qk[pp < PF ? pp : 0];               // batch == tile: pass pp of the tile is pass pp of the batch
                            } else {
#pragma unroll
                                for (int c = 0; c < CH; ++c)
#pragma unroll
                                    for (int v = 0; v < V; ++v) {
                                        const int d = (c * G + gl) * V + v;
                                        rr.x[c][v] = d < D ? tile[row * RT + d] : 0.0;
                                    }
                            }
                            store_row<G, V, CH>(rr, A.X + (size_t)wi2 * D, D, gl);
                            if (chain_) store_row_stream<G, V, CH>(rr, chain_ + (size_t)wi2 * D, D, gl);
                            if (sendbuf_) store_row<G, V, CH>(rr, sendbuf_ + (size_t)(t0 + sidx - tlo_) * (D + 2), D, gl);
                        }
                    }
                    EMX_STAMP(9);      // commit stores issued
                    EMX_WAVE_SYNC();
                }
            }
        }
    }   // batch loop
    if (EMX_OPT_STAMPS && A.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    EMX_STAMP(10);             // every store of this wave acknowledged
    if (EMX_OPT_STAMPS && A.dbg && wib == 0 && lane == 0) A.dbg[(size_t)blockIdx.x * 16 + 12] = wall_clock64();
}

// ----------------------------------------------------------------------------------------
// Persistent form of the fused dense half-step (stretch, DE or snooker: MOVE template parameter): up to PERSIST_MAX_ITERS
// consecutive half-steps (= 16 steps of two splits, 8 of four) in ONE launch, a device-wide barrier where the kernel boundaries were.  A wave owns the same 16 plan slots of every
// split (the grid is exactly one 16-walker tile per wave, all workgroups co-resident); the LDS image of the target is staged
// once.  Two things make it pay:
//   * no cache maintenance at the barrier.  A fenced device-wide barrier costs 5.0 us, 3.2 of them the L2 write-back and
//     invalidate (profiles/r02/gridbarrier2_ubench.txt).  Here every access to the walker state (X, lp, acc_count, the stamps)
//     is an agent-scope access instead -- sc1 loads that are served by memory, sc1 stores that are acknowledged when the
//     device can see them -- so the barrier is s_waitcnt + per-XCD arrival counters + one "go" word only the last arriver
//     writes (1.8 us);
//   * what the NEXT half-step reads is loaded while this one computes: its plan entries, and its own rows and log-probs, in
//     flight during the MFMA phase when the memory system is idle.  When it is the second split of the same step its walkers
//     are this half-step's untouched complement (red_blue.py:62-67) and those loads are final; when it is the first split of
//     a new step some of them are being moved right now, so every accepted update also writes the half-step's stamp to ver[]
//     and after the barrier the walkers whose stamp is this half-step's (8 % at C2's acceptance) are loaded again.
//     Only the partner rows (the walkers just updated) always wait for the barrier.  (Measured and dropped: loading them
//     speculatively too and re-loading the sixth that moved -- 94 % of the waves then need the second round trip after the
//     stamps: 20.4 -> 23.0 us/step; profiles/r03/persist_spec_partners.txt.)
// Same load_row / make_proposal / MFMA chain / reductions / decision as k_halfstep<G, V, CH, STRETCH, DPB, 1>, in the same
// order: the same bits (tests/test_gpu_persist.py).  Every spin is bounded by the wall clock: a barrier that is never met
// raises the exchange-timeout status bit and the kernel runs to its end unsynchronised (reported; never a hang).
// ----------------------------------------------------------------------------------------
// Loads and stores of the walker state inside the persistent kernel are AGENT-scope accesses (sc1).  An ordinary load may be
// served by a line the vector L1 or this XCD's L2 still holds from before another workgroup's commit -- even when the
// allocation is hipDeviceMallocUncached (measured there: with plain loads every run of 37 steps differs from the reference
// path in a few hundred rows; buffer_inv sc1 after the barrier repairs that at 13 us a time, sc1 on the loads costs nothing).
// An ordinary store is acknowledged once this XCD's L2 has taken it; an agent-scope store when the device can see it, which
// is what the s_waitcnt before the barrier arrival has to mean (with plain stores about one run in a hundred still differed).
// With both, ordinary device memory is as good as uncached memory (profiles/r03/persist_coherence.txt).
typedef unsigned int emx_u4 __attribute__((ext_vector_type(4)));
constexpr int EMX_CPOL_SC1 = 16;
// CPOL: EMX_CPOL_SC1 (agent scope) or 0 (plain: the one-XCD form's stores)
template <int G, int V, int CH, int CPOL = EMX_CPOL_SC1>
__device__ __forceinline__ void load_row_agent(Row<G, V, CH>& r, __amdgpu_buffer_rsrc_t rsrc, int row, int D, int gl) {
    static_assert(V == 2 || V == 1, "two coordinates per lane and chunk (even ndim), or one (odd ndim: rows are 8-byte aligned only)");
    const int base = row * D;
    if constexpr (V == 1) {
        typedef unsigned int emx_u2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int d = c * G + gl;
            if (d < D) {
                const emx_u2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (base + d) * 8, 0, CPOL);
                double t;
                __builtin_memcpy(&t, &w, 8);
                r.x[c][0] = t;
            } else {
                r.x[c][0] = 0.0;
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int d = (c * G + gl) * V;
        if (d + 1 < D) {
            const emx_u4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (base + d) * 8, 0, CPOL);
            double2 t;
            __builtin_memcpy(&t, &w, 16);
            r.x[c][0] = t.x;
            r.x[c][1] = t.y;
        } else {
            r.x[c][0] = 0.0;
            r.x[c][1] = 0.0;
        }
    }
}
template <int G, int V, int CH, int CPOL = EMX_CPOL_SC1>
__device__ __forceinline__ void store_row_agent(const Row<G, V, CH>& r, __amdgpu_buffer_rsrc_t rsrc, int row, int D, int gl) {
    static_assert(V == 2 || V == 1, "two coordinates per lane and chunk (even ndim), or one (odd ndim)");
    const int base = row * D;
    if constexpr (V == 1) {
        typedef unsigned int emx_u2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int d = c * G + gl;
            if (d < D) {
                emx_u2 w;
                const double t = r.x[c][0];
                __builtin_memcpy(&w, &t, 8);
                __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, (base + d) * 8, 0, CPOL);
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int d = (c * G + gl) * V;
        if (d + 1 < D) {
            const double2 t = {r.x[c][0], r.x[c][1]};
            emx_u4 w;
            __builtin_memcpy(&w, &t, 16);
            __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, (base + d) * 8, 0, CPOL);
        }
    }
}
template <typename T>
__device__ __forceinline__ void store_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned load_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the same with the scope as a template parameter (LOCAL: the one-XCD form -- plain accesses)
template <bool LOCAL, typename T>
__device__ __forceinline__ void store_scope(T* p, T v) {
    if constexpr (LOCAL)
        *p = v;               // (plain: ordered before the barrier's flag store by the s_waitcnt in front of it)
    else
        store_agent(p, v);
}

// Half-steps a persistent launch can hold.  Round 6: 40 (32 before) -- twenty stretch / DE steps, so that an emx_run of 20 steps (the
// block length the bench is driven with) is ONE launch, not 16 + 4; a half-step's descriptor shrank from 96 to 48 bytes for it (the
// plan's seven columns are one pointer: every plan slot is one block [order|p0] [s0|uacc] [p1|p2] [logu|fac]), the kernel arguments
// from 3.5 to 2.4 KB.
constexpr int PERSIST_MAX_ITERS = 40;
constexpr int PERSIST_BAR_STRIDE = 32;          // words between the blocks of PersistArgs::bar: a cache line each (4 KB or 64 KB apart -- other
                                                // channels of the memory side -- changes nothing: profiles/r06/barrier/bar_stride_ab.txt)
constexpr int PERSIST_BAR_WORDS = 12 * PERSIST_BAR_STRIDE;   // PersistArgs::bar
struct PersistIter {
    const char* plan;              // the plan slot's block: [order|p0] int32, [s0|uacc] f64, [p1|p2] int32, [logu|fac] f64, N entries a column
    double *chain, *chain_lp;      // this step's row of the stored chain (backend.py:229), or nullptr
    int32_t pos0, split;
    double gammas;                 // k_persist_mix: the snooker move's scale of THIS half-step (HalfStepArgs::gammas is the first captured step's)
    int32_t kind, shift;           // MOVE_MIX: the half-step's move; it has 2^-shift as many tiles as the grid has waves (k_persist_mix: mix_tile)
};
// ... as the kernels read it: the columns spelled out (p1: the DE move's second partner; p1, p2: the snooker move's z1, z2)
struct PersistCols {
    const int32_t *order, *p0, *p1, *p2;
    const double *s0, *logu, *fac;
    double *chain, *chain_lp;
    int32_t pos0, split;
    double gammas;
    int32_t kind, shift;
    __host__ __device__ __forceinline__ PersistCols(const PersistIter& it, size_t N)
        : order(reinterpret_cast<const int32_t*>(it.plan)), p0(reinterpret_cast<const int32_t*>(it.plan) + N),
          p1(reinterpret_cast<const int32_t*>(it.plan + N * 24)), p2(reinterpret_cast<const int32_t*>(it.plan + N * 24) + N),
          s0(reinterpret_cast<const double*>(it.plan + N * 8)), logu(reinterpret_cast<const double*>(it.plan + N * 32)),
          fac(reinterpret_cast<const double*>(it.plan + N * 32) + N), chain(it.chain), chain_lp(it.chain_lp), pos0(it.pos0), split(it.split),
          gammas(it.gammas), kind(it.kind), shift(it.shift) {}
};
struct PersistArgs {
    HalfStepArgs base;
    PersistIter it[PERSIST_MAX_ITERS];
    unsigned* bar;                 // [8][32] per-XCD arrival counters | [32] global counter | [32] go word (go, dead, XCD mask, seq, one-XCD launch's XCD)
                                   // | [32] [32] the one-XCD form's counter and go word
    unsigned* ver;                 // (N) stamp of the half-step that last moved the walker
    unsigned epoch0;               // barriers already passed on these counters
    unsigned lepoch0;              // the one-XCD form: barriers already passed on ITS flags (the handshake does not count there)
    unsigned hepoch0;              // the one-XCD form: handshakes already counted (they alone use the arrival counter)
    unsigned long long timeout_ticks;
    int32_t niter;
    unsigned seq;                  // number of this launch (left in the barrier block's fourth `go` word once its grid is known co-resident)
    unsigned* started_host;        // pinned host word (or null): `seq` again, for the host -- launch k + 1 has started, so launch k is over
    int32_t stagger;               // > 0: some waves issue their partner loads later than the others (persist_stagger_wait)
};
static_assert(sizeof(PersistArgs) <= 4096, "kernel arguments of the persistent kernels");

// The one-XCD form's barrier: every workgroup of the grid runs on ONE XCD (k_persist<..., LOCAL>), so its L2 is the point of
// coherence.  What that allows was measured flavour by flavour (tools/exp/xcd_local_probe.hip, profiles/r04/xcd_local_probe.txt):
// read-modify-write atomics are executed beyond the L2 (polls of the L2 never see them); plain or sc0 loads are served stale whatever
// is invalidated; but a PLAIN store is in the L2 when it is acknowledged, and an agent-scope (sc1) load of a line this XCD wrote is
// answered by that L2 -- 0.32 us for a whole flag barrier of 32 workgroups against 2.5 us device-wide.  So: workgroup g stores the
// barrier's index into word g of one line, the first wave of every workgroup polls the line with sc1 loads, one word a lane.
__device__ __forceinline__ void persist_barrier_local(const PersistArgs& P, unsigned k, unsigned bid, unsigned ngroups) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's commits have reached the L2
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const __amdgpu_buffer_rsrc_t Fr = __builtin_amdgcn_make_buffer_rsrc((void*)(P.bar + 10 * PERSIST_BAR_STRIDE), 0, 64 * 4, 0x00020000);
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(k, Fr, (int)bid * 4, 0, 0);
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            // words 0 .. 31: the workgroups' flags; word 32: set by a workgroup whose wait timed out (the run is void: pass)
            const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(Fr, lane * 4, 0, EMX_CPOL_SC1);
            const bool ok = lane < (int)ngroups ? (int)(v - k) >= 0 : true;
            const bool dead = lane == 32 && v != 0u;
            if (__ballot(ok) == ~0ull || __ballot(dead) != 0ull) break;
            if (wall_clock64() - t0 > P.timeout_ticks) {
                if (lane == 0) {
                    raise_status(P.base.status, ST_EXCHANGE_TIMEOUT);
                    __hip_atomic_store(P.bar + 9 * PERSIST_BAR_STRIDE + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // 2: in the middle of a launch
                    __builtin_amdgcn_raw_buffer_store_b32(1u, Fr, 32 * 4, 0, 0);
                }
                break;
            }
        }
    }
    __syncthreads();
}

// The device-wide barrier: two levels of arrival counters, then a poll of one word.  2.5-2.7 us of a 10 us half-step at the headline
// shape (profiles/r04/persist_phase_c2.txt, profiles/r06/persist_phase_hier.txt).  Round 6 built two forms that replace the
// read-modify-write atomics by words -- hierarchical (flag words polled inside an XCD's L2, one word per XCD across: 2.55 us, the
// step time unchanged within 0.5 %) and flat (a word per workgroup, everybody polls all of them: 4.6 us) -- both bit-equal, neither
// faster; they were removed again (profiles/r06/hier_barrier.md names the commits that hold them).
// What the barrier costs is three trips to the memory side whichever instruction makes them.  One trip -- arrive without waiting for the
// count, every workgroup polling the eight per-XCD counters themselves -- is slower still (65 536 x 64: 21.8 -> 25.2 us/step,
// profiles/r06/barrier/one_trip_bar_form_ab.txt): 256 pollers on the lines the arrivals are counted in delay the arrivals.  And TWO trips
// -- the last arriver of an XCD adds to `go` itself, which counts eight a barrier, nobody is elected a second time -- change nothing
// (20.37-20.53 against 20.24-20.31 us/step, barrier/bar_two_trip_ab.txt): what a workgroup waits for at the barrier is the slowest
// workgroup, not the mechanism (barrier/barrier_skew.txt).  A copy of `go` per XCD (32 pollers a line): c4 -0.7 %, c2 +5 % (bar_replicas_ab.txt).
__device__ __forceinline__ void persist_barrier(const PersistArgs& P, unsigned k, unsigned long long* wall = nullptr) {      // (wall: instrumented build, [arrived, released] on the 100 MHz clock)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's commits (agent-scope stores) are visible to the device
    __syncthreads();
    if (threadIdx.x == 0) {
        const int xcd = blockIdx.x & 7;
        const unsigned per = (gridDim.x + 7 - xcd) / 8;
        unsigned* xctr = P.bar + xcd * PERSIST_BAR_STRIDE;
        unsigned* gctr = P.bar + 8 * PERSIST_BAR_STRIDE;
        unsigned* go = P.bar + 9 * PERSIST_BAR_STRIDE;                            // [go | dead]: one 8-byte word, polled with one load
        if (wall) wall[0] = wall_clock64();
        const unsigned old = __hip_atomic_fetch_add(xctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == k * per - 1) {
            const unsigned o2 = __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 == k * 8 - 1) __hip_atomic_store(go, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            const unsigned long long v = __hip_atomic_load(reinterpret_cast<unsigned long long*>(go), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)((unsigned)v - k) >= 0 || (v >> 32) != 0ull) break;          // (the counters wrap) | a barrier of this run has timed out
            if (wall_clock64() - t0 > P.timeout_ticks) {
                // never met: report, and let every later barrier of the run through at once (the results are void anyway;
                // the host refuses further persistent launches until the status has been read)
                raise_status(P.base.status, ST_EXCHANGE_TIMEOUT);
                __hip_atomic_store(go + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // 2: in the middle of a launch
                break;
            }
        }
        if (wall) wall[1] = wall_clock64();
    }
    __syncthreads();
}

// After a barrier every wave of the grid asks for its partner rows at once.  Waves that ask a little later (PersistArgs::stagger = how *
// 256 + units of 64 clocks; how 0: the second wave of every SIMD -- waves 4-7 of the workgroup --, 1: the odd waves, 2: the waves of
// SIMDs 2 and 3, 4: the waves of SIMD k wait k units) leave the memory pipeline to the others first: 65 536 x 64 stretch 20.75-20.93 ->
// 20.13-20.21 us/step with the waves of SIMDs 2 and 3 waiting 256-384 clocks (one late wave a SIMD: 20.5-20.6; every wave at its own
// time, or the wait placed in front of the MFMA phase instead: nothing), with stored chain rows 27.7-28.0 -> 27.1-27.4 with the second
// wave of every SIMD waiting 512 (profiles/r06/stagger_ab.md).
__device__ __forceinline__ void persist_stagger_wait(int stagger, int wib) {
    const int how = stagger >> 8, cnt = stagger & 255;
    const int units = how == 0 ? (wib >> 2) : how == 1 ? (wib & 1) : how == 2 ? ((wib >> 1) & 1) : (wib & 3);
    for (int s = 0; s < cnt * units; ++s) __builtin_amdgcn_s_sleep(1);
}

// The first barrier of a launch, BEFORE anything is written: every workgroup of the grid reports in.  Once it has been passed the
// whole grid is resident (nothing leaves a CU before it exits), so the barriers between the half-steps can only be late, never
// unmet.  When it is NOT passed in time -- another process holds the CUs with a persistent grid of its own -- the launch gives
// up with the ensemble untouched: the dead mark (1: clean) sends home the workgroups that start later and every launch queued
// behind this one, and the host redoes those launches' steps on the per-half-step path (persist_recover, emx.hip).
// -> false: leave without a store.
// LOCAL (the one-XCD form): only the workgroups with blockIdx & 7 == 0 get here; they count in ONE agent-scope counter (once per
// launch: it need not be cheap), each adds the XCD it really runs on to a mask, and the last arriver opens the barrier only when the
// mask names a single XCD -- otherwise the launch gives up, untouched, exactly like a grid that could not become co-resident.
template <bool LOCAL = false>
__device__ __forceinline__ bool persist_handshake(const PersistArgs& P) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        const unsigned k = LOCAL ? P.hepoch0 + 1u : P.epoch0 + 1u;
        const int xcd = LOCAL ? 0 : (int)(blockIdx.x & 7);
        const unsigned per = LOCAL ? gridDim.x >> 3 : (gridDim.x + 7 - xcd) / 8;
        unsigned* xctr = P.bar + xcd * PERSIST_BAR_STRIDE;
        unsigned* gctr = P.bar + 8 * PERSIST_BAR_STRIDE;
        unsigned* go = P.bar + 9 * PERSIST_BAR_STRIDE;
        int ok = 1;
        const unsigned long long v0 = __hip_atomic_load(reinterpret_cast<unsigned long long*>(go), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v0 >> 32) != 0ull) {
            ok = 0;                                                 // an earlier launch gave up: not even counted
        } else {
            unsigned xcc = 0;
            if constexpr (LOCAL) {
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                __hip_atomic_fetch_or(go + 2, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the mark is in before this workgroup counts as arrived
            }
            const unsigned old = __hip_atomic_fetch_add(xctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == k * per - 1) {
                bool open = true;
                unsigned o2 = k * 8 - 1;
                if constexpr (LOCAL) {
                    const unsigned m = __hip_atomic_load(go + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    open = (m & (m - 1u)) == 0u;                                   // one XCD
                    if (open) {                                                    // (every workgroup of this launch has marked; the next launch starts clean)
                        __hip_atomic_store(go + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {                                                   // report it like a grid that never became co-resident
                        raise_status(P.base.status, ST_EXCHANGE_TIMEOUT);
                        __hip_atomic_store(go + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             // 1: nothing was written
                    }
                } else {
                    o2 = __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (open && o2 == k * 8 - 1) {
                    // where a one-XCD launch lives (XCC_ID + 1; 0: everywhere), for k_plan_fetch, which keeps off that XCD
                    __hip_atomic_store(go + 4, LOCAL ? (xcc & 15u) + 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(go + 3, P.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // this launch will run to its end
                    if (P.started_host) __hip_atomic_store(P.started_host, P.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(go, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                const unsigned long long v = __hip_atomic_load(reinterpret_cast<unsigned long long*>(go), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 32) != 0ull) {
                    ok = 0;
                    break;
                }
                if ((int)((unsigned)v - k) >= 0) break;
                if (wall_clock64() - t0 > P.timeout_ticks) {
                    raise_status(P.base.status, ST_EXCHANGE_TIMEOUT);
                    __hip_atomic_store(go + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // 1: nothing was written
                    ok = 0;
                    break;
                }
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

template <int G, int V, int CH, int DPB, int MOVE = MOVE_STRETCH, bool LOCAL = false, bool ROWS_LATE = false>
static __global__ __launch_bounds__(512) void k_persist(const PersistArgs P) {
    static_assert(MOVE == MOVE_STRETCH || MOVE == MOVE_DE || MOVE == MOVE_SNOOKER, "the red / blue moves");
    // LOCAL: the one-XCD form for small ensembles.  The dispatcher deals workgroups to the eight XCDs in turn (workgroup i -> XCD
    // i mod 8: tools/exp/cu_mask_probe.hip), so of an eight times larger grid only every eighth workgroup works -- all of them on one
    // XCD, whose L2 then keeps the walker state coherent without agent-scope accesses (3.0 + 2.5 us of partner round trip and barrier
    // per half-step in the device-wide form, profiles/r04/persist_phase_c2.txt).  The handshake checks that they really share one.
    if (LOCAL && (blockIdx.x & 7u) != 0u) return;
    const unsigned bid = LOCAL ? blockIdx.x >> 3 : blockIdx.x, ngroups = LOCAL ? gridDim.x >> 3 : gridDim.x;
    constexpr int CPOL = EMX_CPOL_SC1;                       // loads: agent scope in both forms (answered by the L2 in the one-XCD form)
    constexpr int CPOL_ST = LOCAL ? 0 : EMX_CPOL_SC1;       // stores: plain in the one-XCD form (in the L2 when acknowledged)
    constexpr bool DE = MOVE == MOVE_DE || MOVE == MOVE_SNOOKER;   // de.py:40-64: two partners, q = s + gamma (c[pair 1] - c[pair 0])
    constexpr bool SN = MOVE == MOVE_SNOOKER;          // de_snooker.py:31-46: three partners z, z1, z2 (one from each other set)
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
    const int t0 = wave * 16;                                   // this wave's slots of every split
    // instrumented build only (tools/persist_phase_clock.py, -DEMX_OPT_STAMPS=1): where the first wave of every workgroup spends a
    // half-step -- ticks summed over the launch's half-steps: partner rows arrive | proposals + tile | MFMA + reductions |
    // decisions + commit issued | stores acknowledged | barrier
    unsigned long long pst[6] = {0, 0, 0, 0, 0, 0}, pt = 0;
    const bool prof = EMX_OPT_STAMPS && A.dbg && wib == EMX_STAMP_WAVE;
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

    int wi[PF], ja[PF], jb[DE ? PF : 1], jc[SN ? PF : 1], my_i;
    double s0v[PF], facv[PF], my_logu, my_lpo;
    Row<G, V, CH> xi[PF];
    {
        const PersistCols I(P.it[0], (size_t)P.base.N);
        const int pbase = I.pos0 + t0;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int pos = pbase + k * WPW + sub;
            wi[k] = I.order[pos];
            ja[k] = I.p0[pos];
            if constexpr (DE) jb[k] = I.p1[pos];
            if constexpr (SN) jc[k] = I.p2[pos];
            s0v[k] = SN ? 0.0 : I.s0[pos];
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
        // -------- partner rows: the walkers the previous half-step updated --------
        Row<G, V, CH> xa[PF], xb[DE ? PF : 1], xc[SN ? PF : 1];
        persist_stagger_wait(P.stagger, wib);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            load_row_agent<G, V, CH, CPOL>(xa[k], Xr, ja[k], D, gl);
            if constexpr (DE) load_row_agent<G, V, CH, CPOL>(xb[k], Xr, jb[k], D, gl);
            if constexpr (SN) load_row_agent<G, V, CH, CPOL>(xc[k], Xr, jc[k], D, gl);
        }
        // -------- plan entries of the next half-step (written by the plan kernel before this launch) --------
        const bool more = n + 1 < P.niter;
        const PersistCols J(P.it[more ? n + 1 : n], (size_t)P.base.N);
        const bool pre = more && J.split != 0;                   // its own walkers are this half-step's complement
        const unsigned stamp = P.epoch0 + (unsigned)n + 1u;      // of this half-step (never 0 before the counters wrap)
        int wi_n[PF], ja_n[PF], jb_n[DE ? PF : 1], jc_n[SN ? PF : 1], my_i_n;
        double s0_n[PF], fac_n[PF], my_logu_n;
        {
            const int pbase = J.pos0 + t0;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int pos = pbase + k * WPW + sub;
                wi_n[k] = J.order[pos];
                ja_n[k] = J.p0[pos];
                if constexpr (DE) jb_n[k] = J.p1[pos];
                if constexpr (SN) jc_n[k] = J.p2[pos];
                s0_n[k] = SN ? 0.0 : J.s0[pos];
                fac_n[k] = J.fac[pos];
            }
            my_i_n = J.order[pbase + myrow];
            my_logu_n = J.logu[pbase + myrow];
        }
        // (the scheduler puts these loads BEHIND the wait for the partner rows.  Forced in front of it -- a scheduling barrier here -- the
        // step is 2.4 % slower, 20.17 -> 20.65 us: ten more requests in front of the rows; profiles/r06/stagger/entries_early_ab.txt)
        // -------- proposals -> the wave's LDS tile (R = Q - mu), kept in registers for the commit --------
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        EMX_PSTAMP(0);       // partner rows (and the next half-step's plan entries) have arrived
        Row<G, V, CH> qk[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int srow = k * WPW + sub;
            double factor = facv[k];
            Row<G, V, CH> q;
            make_proposal<G, V, CH, MOVE>(xi[k], xa[k], xb[DE ? k : 0], xc[SN ? k : 0], s0v[k], A.gammas, D, gl, q, factor, ja[k]);
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
        // -------- own rows of the next half-step: in flight during the MFMA phase (speculative unless `pre`) --------
        // ROWS_LATE (round 6; the instantiation of launches that store chain rows): they are asked for BEHIND the MFMA phase, under the
        // decisions, the commit and the barrier -- in front of it they share the memory pipeline with the 17 MB of chain rows of the
        // half-step before (65 536 x 64 with the chain stored every step: 27.6-27.9 -> 26.3-26.4 us/step; without stored rows the same
        // move costs 4 %: 20.25 -> 21.1, hence a template parameter; profiles/r06/stagger/ownrows_late_ab*.txt)
        double my_lpo_n = 0.0;
        if (!ROWS_LATE && more) {
#pragma unroll
            for (int k = 0; k < PF; ++k) load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi_n[k], D, gl);
            my_lpo_n = load_agent(A.lp + my_i_n);
        }
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
        if (ROWS_LATE && more) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < PF; ++k) load_row_agent<G, V, CH, CPOL>(xi[k], Xr, wi_n[k], D, gl);
            my_lpo_n = load_agent(A.lp + my_i_n);
            __builtin_amdgcn_sched_barrier(0);
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
        if (!more) break;
        if constexpr (LOCAL)
            persist_barrier_local(P, P.lepoch0 + (unsigned)n + 1u, bid, ngroups);
        else
            persist_barrier(P, P.epoch0 + (unsigned)n + 2u,       // (+ 1: the handshake was this launch's first barrier)
                            (EMX_OPT_STAMPS && A.dbg) ? A.dbg + 4096 + ((size_t)n * ngroups + bid) * 2 : nullptr);
        EMX_PSTAMP(5);       // device-wide barrier
        // -------- roll over --------
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            wi[k] = wi_n[k];
            ja[k] = ja_n[k];
            if constexpr (DE) jb[k] = jb_n[k];
            if constexpr (SN) jc[k] = jc_n[k];
            s0v[k] = s0_n[k];
            facv[k] = fac_n[k];
        }
        my_i = my_i_n;
        my_logu = my_logu_n;
        my_lpo = my_lpo_n;
        if (!pre) {      // first split of a new step: the walkers that moved in the half-step before are loaded again
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

// ----------------------------------------------------------------------------------------
// Small ensembles: the whole run in ONE workgroup.
// When the ensemble (coordinates, log-probs, one step's plan) fits the 160 KB LDS of a CU, a step is two
// launches of pure latency (~3.6 us each).  Here one workgroup keeps the ensemble in LDS and iterates
// plan -> half-step -> ... -> half-step with workgroup barriers where the general path has kernel
// boundaries: `nsteps` full steps per launch, HBM touched only for the stored chain rows.  Either RNG mode, any
// schedule of stretch / DE / snooker moves, element-wise targets; the same device functions as the general path (native_slot,
// make_proposal, eval_valu_target), hence the same bits (tests/test_gpu_small_run.py).
// ----------------------------------------------------------------------------------------
constexpr int SMALL_MAX_MOVES = 8;
struct GaussGen {          // what gauss_disp_row needs to generate a walker's displacement row in registers
    unsigned long long gseed, gstep;
    double gfac, gsigma;
    const double* gscale;
};
// ----------------------------------------------------------------------------------------
// The Gaussian Metropolis move (moves/gaussian.py + moves/mh.py) on the fused dense target, persistent and WITHOUT any
// synchronisation: the update of a walker reads nobody else's row (mh.py:57-77), so a wave keeps its 16 walkers' rows and
// log-probs in registers for every step of the launch -- per step it generates the displacement rows (gauss_disp_row), writes
// the proposal tile, runs the MFMA chain, decides, and replaces the accepted rows in registers.  Global memory sees the plan's
// accept uniforms (and the coordinate that moves), the chain rows of stored steps, and the final state once.  Same functions,
// same order as k_halfstep<G, V, CH, MOVE_GAUSS, DPB, 1>: the same bits (tests/test_gpu_persist.py).
// ----------------------------------------------------------------------------------------
constexpr int PERSIST_GAUSS_MAX_STEPS = 16;
struct PersistGaussStep {
    const int32_t* p0;             // the coordinate that moves, per walker (-1: all of them; gaussian.py:92-101)
    const double* logu;            // log of the accept uniform, per walker
    double *chain, *chain_lp;      // this step's row of the stored chain, or nullptr
    unsigned long long gstep;      // Philox step of the noise
    double gfac;                   // this step's step-size factor (gaussian.py:81-84)
};
struct PersistGaussArgs {
    HalfStepArgs base;
    PersistGaussStep st[PERSIST_GAUSS_MAX_STEPS];
    int32_t nsteps;
};

template <int G, int V, int CH, int DPB>
static __global__ __launch_bounds__(512) void k_persist_gauss(const PersistGaussArgs P) {
    constexpr int MOVE = MOVE_GAUSS;
    constexpr int WPW = 64 / G;
    constexpr int PPT = 16 / WPW;
    constexpr int Dp = DPB * 16, KK = Dp / 4, RT = Dp + 2;
    static_assert(EMX_OPT_RTILE && EMX_OPT_RED4 && G * V * CH >= Dp, "the one-tile-per-batch form");
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
    double* facS = tile + 16 * RT + 16;
    {
        constexpr int IMG2 = (dense_img_doubles(Dp) + Dp) / 2;
        const double2* img = reinterpret_cast<const double2*>(A.tp1);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int e = threadIdx.x; e < IMG2; e += blockDim.x) dst[e] = img[e];
    }
    Row<G, V, CH> mu;
    load_row<G, V, CH>(mu, A.tp0, D, gl);
    __syncthreads();
    const int wave = blockIdx.x * (blockDim.x >> 6) + wib;
    const int w0 = wave * 16;                                   // this wave's walkers (the Gaussian move's plan order is the identity)
    const int myrow = (lane >> 4) + 4 * (lane & 3);
    const bool mine = (lane & 15) < 4;
    const int my_i = w0 + myrow;
    Row<G, V, CH> xi[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) load_row<G, V, CH>(xi[k], A.X + (size_t)(w0 + k * WPW + sub) * D, D, gl);
    double my_lp = A.lp[my_i];
    bool my_acc = false;
    for (int s = 0; s < P.nsteps; ++s) {
        const PersistGaussStep& S = P.st[s];
        GaussGen gg;
        gg.gseed = A.gseed;
        gg.gstep = S.gstep;
        gg.gfac = S.gfac;
        gg.gsigma = A.gsigma;
        gg.gscale = A.gscale;
        const double my_logu = S.logu[my_i];
        Row<G, V, CH> qk[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = w0 + k * WPW + sub;
            const int col = S.p0[i];
            Row<G, V, CH> xa, q;
            gauss_disp_row<G, V, CH>(xa, gg, i, col, D, gl);
            double factor = 0.0;                                 // the plan's fac column of a Gaussian step (symmetric proposal)
            make_proposal<G, V, CH, MOVE>(xi[k], xa, xa, xa, 0.0, A.gammas, D, gl, q, factor, col);
            bool bl = false;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) bl |= !(fabs(q.x[c][v]) <= 1.79769313486231570815e308);
            const bool badq = group_any<G>(bl, sub);
            if (badq && gl == 0) raise_status(A.status, ST_BAD_COORD);
            const int trow = (k * WPW + sub) & 15;
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
        EMX_WAVE_SYNC();
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
        bool acc = false;
        if (mine) {
            const double lpn = -0.5 * my_qf;
            if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);
            const double lnpdiff = facS[myrow] + lpn - my_lp;      // mh.py:63-66
            acc = lnpdiff > my_logu;
            if (acc) my_lp = lpn;
            my_acc = acc;
            if (S.chain_lp) {
                S.chain_lp[my_i] = my_lp;
                if (acc) A.acc_count[my_i] += 1u;
            }
        }
        const unsigned long long am64 = __ballot(acc);
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp) {
            const int row = pp * WPW + sub;
            const bool ac = (am64 >> ((row & 3) * 16 + (row >> 2))) & 1ull;
            if (ac) xi[pp] = qk[pp];                               // move.py:33, in registers
            if (S.chain) store_row_stream<G, V, CH>(xi[pp], S.chain + (size_t)(w0 + row) * D, D, gl);
        }
        EMX_WAVE_SYNC();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) store_row<G, V, CH>(xi[k], A.X + (size_t)(w0 + k * WPW + sub) * D, D, gl);
    if (mine) {
        A.lp[my_i] = my_lp;
        A.acc[my_i] = my_acc ? 1 : 0;
    }
}

constexpr int SMALL_ANY_MOVE = 7;      // MOVESEL: the kernel carries all three split-ensemble moves and picks per step

// move of a native-mode step: one Philox draw against the cdf (the host's philox_move_choice)
__host__ __device__ inline int native_move_choice(uint64_t seed, uint64_t step, const double* cdf, int n) {
    const Philox4 r = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), 0x4d4f5645u /*'MOVE'*/, 0, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    const double u = u53(r.v[0], r.v[1]);
    int k = 0;
    while (k < n - 1 && u >= cdf[k]) ++k;
    return k;
}

struct SmallRunArgs {
    double* X;
    double* lp;
    uint8_t* acc;
    uint32_t* acc_count;
    uint32_t* status;
    double* chain;        // first chain row this launch may append to (nullptr: nothing stored)
    double* chain_lp;
    const double* tp0;
    const double* tp1;
    double tscale;
    // the move schedule (ensemble.py:115-129): up to SMALL_MAX_MOVES stretch / DE / snooker moves and their cdf
    double a[SMALL_MAX_MOVES], sigma[SMALL_MAX_MOVES], g0[SMALL_MAX_MOVES], gammas[SMALL_MAX_MOVES], cdf[SMALL_MAX_MOVES];
    int32_t kind[SMALL_MAX_MOVES], nsplits[SMALL_MAX_MOVES];
    // Gaussian Metropolis moves (native mode): mode, isotropic sigma / per-coordinate scale; per step of the launch the
    // step-size factor and the sequential mode's column (host-computed: both are functions of the step number alone)
    int32_t gmode[SMALL_MAX_MOVES];
    double gsigma[SMALL_MAX_MOVES];
    const double* gscale[SMALL_MAX_MOVES];
    const double* step_fac;
    const int32_t* step_col;
    int32_t nmoves;
    unsigned long long seed, step0;
    long long i0;         // index of the first step inside the emx_run call (thinning phase, ensemble.py:416)
    int32_t N, D, target, nsteps, thin_by, store;
    int32_t batch;        // steps whose plans are evaluated in one pass (batch * N plan entries live in LDS)
    // PLANNED instantiations (exact MT19937 mode): the host-made plans of the launch's steps, 32 N bytes per step in
    // the staging layout [order|p0|p1|p2] int32, [s0|uacc] f64, and the move the host's choice() picked for each step
    const char* plans;
    const int32_t* step_moves;
};

// one entry of a step's plan (k_native_plan_batch's arithmetic for this move)
template <int MOVE>
__device__ __forceinline__ void small_plan_entry(const NativeArgs& na, int N, int D, int S, int pos, double a, double sigma,
                                                 double g0, int& i, int& a0, int& a1, int& a2, double& z, double& lu, double& fc) {
    int split = 0, t = pos;
    const SplitSizes sz = split_sizes(N, S);
    for (int k = 0; k < S; ++k) {
        const int n = sz.of(k);
        if (t < n) { split = k; break; }
        t -= n;
    }
    double u;
    native_slot<MOVE>(na, N, S, split, t, a, sigma, g0, i, a0, a1, a2, z, u);
    lu = plan_log_uniform(u);
    fc = (MOVE == MOVE_STRETCH) ? ((double)D - 1.0) * plan_log(z) : 0.0;
}

// one group's (walker's) update inside a half-step: the general kernel's element-wise-target branch
// the proposal of one walker from the LDS-resident ensemble (+ the non-finite check of ensemble.py:476-479)
template <int G, int V, int CH, int MOVE>
__device__ __forceinline__ void small_propose(const SmallRunArgs& A, const double* Xs, bool live, int i, int j0, int j1, int j2,
                                              double s0, double fac, double gammas, int D, int gl, int sub, Row<G, V, CH>& q,
                                              double& factor, bool& badq, const GaussGen* gg = nullptr) {
    constexpr int NR = rows_per_pass<MOVE>();
    Row<G, V, CH> xi, xa, xb, xc;
    load_row<G, V, CH>(xi, Xs + (size_t)i * D, D, gl);
    if constexpr (MOVE == MOVE_GAUSS)
        gauss_disp_row<G, V, CH>(xa, *gg, i, j0, D, gl);                 // j0: the coordinate that moves, or -1
    else
        load_row<G, V, CH>(xa, Xs + (size_t)j0 * D, D, gl);
    if constexpr (NR >= 3) load_row<G, V, CH>(xb, Xs + (size_t)j1 * D, D, gl);
    if constexpr (NR >= 4) load_row<G, V, CH>(xc, Xs + (size_t)j2 * D, D, gl);
    factor = fac;
    make_proposal<G, V, CH, MOVE>(xi, xa, NR >= 3 ? xb : xa, NR >= 4 ? xc : xa, (MOVE == MOVE_SNOOKER) ? 0.0 : s0, gammas, D,
                                  gl, q, factor, MOVE == MOVE_GAUSS ? j0 : -1);
    bool bl = false;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) bl |= !(fabs(q.x[c][v]) <= 1.79769313486231570815e308);
    badq = group_any<G>(bl, sub);
    if (live && badq && gl == 0) raise_status(A.status, ST_BAD_COORD);
}

template <int G, int V, int CH, int MOVE>
__device__ __forceinline__ void small_update(const SmallRunArgs& A, double* Xs, double* lps, uint8_t* accs, bool live, int i,
                                             int j0, int j1, int j2, double s0, double fac, double logu, double gammas,
                                             const Row<G, V, CH>& mu, const Row<G, V, CH>& iv, int D, int gl, int sub, int lane,
                                             const GaussGen* gg = nullptr) {
    Row<G, V, CH> q;
    double factor;
    bool badq;
    small_propose<G, V, CH, MOVE>(A, Xs, live, i, j0, j1, j2, s0, fac, gammas, D, gl, sub, q, factor, badq, gg);
    const double lp_new = eval_valu_target<G, V, CH>(q, mu, iv, A.tp0, A.tp1, A.target, A.tscale, D, gl, lane);
    if (live && gl == 0 && (lp_new != lp_new)) raise_status(A.status, ST_NAN_LOGP);
    const double lp_old = lps[i];
    const double lnpdiff = factor + lp_new - lp_old;                  // red_blue.py:99
    const bool accept = live && !badq && (lnpdiff > logu);            // red_blue.py:100
    if (accept) {
        store_row<G, V, CH>(q, Xs + (size_t)i * D, D, gl);
        if (gl == 0) lps[i] = lp_new;
    }
    if (live && gl == 0) accs[i] = accept ? 1 : 0;
}

template <int G, int V, int CH, int MOVESEL, bool PLANNED, int DPB = 0>
static __global__ __launch_bounds__(1024) void k_small_run(const SmallRunArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int WPW = 64 / G;
    constexpr bool DENSE = DPB > 0;
    constexpr int Dp = DENSE ? DPB * 16 : 16, KK = Dp / 4, RT = Dp + 2, PPT = 16 / WPW;
    const int N = A.N, D = A.D, T = blockDim.x, tid = threadIdx.x, B = A.batch;
    const int lane = tid & 63, wv = tid >> 6, nwave = T >> 6, sub = lane / G, gl = lane % G;
    double* Xs = smem;
    double* lps = Xs + (size_t)N * D;
    double* s0s = lps + N;                               // plan arrays: B steps x N entries each
    double* logus = s0s + (size_t)B * N;
    double* facs = logus + (size_t)B * N;
    int* orders = reinterpret_cast<int*>(facs + (size_t)B * N);
    int* p0s = orders + (size_t)B * N;
    int* p1s = p0s + (size_t)B * N;
    int* p2s = p1s + (size_t)B * N;
    uint32_t* acnt = reinterpret_cast<uint32_t*>(p2s + (size_t)B * N);
    uint8_t* accs = reinterpret_cast<uint8_t*>(acnt + N);
    // dense target: the Cholesky image (as in k_halfstep) and one 16-row tile per wave, 16-byte aligned behind the rest
    double* Sfrag = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(accs + N) + 15) & ~(uintptr_t)15);
    double* muS = Sfrag + dense_img_doubles(Dp);
    double* tile = muS + Dp + (size_t)wv * (16 * RT + 16);
    double* facS = tile + 16 * RT;
    if constexpr (DENSE)
        for (int e = tid; e < dense_img_doubles(Dp) + Dp; e += T) Sfrag[e] = A.tp1[e];

    for (int e = tid; e < N * D; e += T) Xs[e] = A.X[e];
    for (int e = tid; e < N; e += T) {
        lps[e] = A.lp[e];
        acnt[e] = 0u;
        accs[e] = 0;
    }
    Row<G, V, CH> mu, iv;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) mu.x[c][v] = iv.x[c][v] = 0.0;
    if (CH <= 4 && A.target == TGT_DIAG) {
        load_row<G, V, CH>(mu, A.tp0, D, gl);
        load_row<G, V, CH>(iv, A.tp1, D, gl);
    }
    // the move of step s: the host's choice() (exact mode) or one Philox draw against the cdf (native mode)
    auto move_of = [&](int s) -> int {
        if (A.nmoves == 1) return 0;
        if constexpr (PLANNED) return A.step_moves[s];
        return native_move_choice(A.seed, A.step0 + (unsigned long long)s, A.cdf, A.nmoves);
    };

    int row = 0;                                     // stored rows appended by this launch
    for (int sb = 0; sb < A.nsteps; sb += B) {
        const int nb = min(B, A.nsteps - sb);
        __syncthreads();                             // the previous batch's plans are no longer read
        // ---- the plans of nb steps in one pass, one entry per thread: plans do not depend on the state, so the
        //      Philox rounds, keyed-permutation inversions and logs of many steps run side by side instead of sitting
        //      on every step's critical path ----
        for (int e = tid; e < nb * N; e += T) {
            const int b = e / N, pos = e - b * N;
            const int m = move_of(sb + b);
            const int kind = MOVESEL == SMALL_ANY_MOVE ? A.kind[m] : MOVESEL;
            if constexpr (PLANNED) {
                // exact mode: entries made by the host's MT19937 twin; the logs are k_plan_logs' arithmetic
                const char* base = A.plans + (size_t)(sb + b) * N * 32;
                const int32_t* hi = reinterpret_cast<const int32_t*>(base);
                const double* hd = reinterpret_cast<const double*>(base + (size_t)N * 16);
                const double z = hd[pos], u = hd[N + pos];
                orders[e] = hi[pos];
                p0s[e] = hi[N + pos];
                p1s[e] = hi[2 * N + pos];
                p2s[e] = hi[3 * N + pos];
                s0s[e] = z;
                logus[e] = plan_log(u);
                facs[e] = (kind == MOVE_STRETCH) ? ((double)D - 1.0) * plan_log(z) : 0.0;
                continue;
            }
            NativeArgs na;
            na.seed = A.seed;
            na.step = A.step0 + (unsigned long long)(sb + b);
            na.pk = make_perm_key((uint64_t)N, na.seed, na.step);
            int i = 0, a0 = 0, a1 = 0, a2 = 0;
            double z = 0.0, lu = 0.0, fc = 0.0;
            const int S = A.nsplits[m];
            if (MOVESEL == MOVE_GAUSS || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_GAUSS)) {
                double u;
                native_gauss_slot(na, D, A.gmode[m], A.step_col ? A.step_col[sb + b] : 0, pos, i, a0, a1, a2, z, u);
                lu = plan_log_uniform(u);
            } else
            if (MOVESEL == MOVE_STRETCH || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_STRETCH))
                small_plan_entry<MOVE_STRETCH>(na, N, D, S, pos, A.a[m], A.sigma[m], A.g0[m], i, a0, a1, a2, z, lu, fc);
            else if (MOVESEL == MOVE_DE || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_DE))
                small_plan_entry<MOVE_DE>(na, N, D, S, pos, A.a[m], A.sigma[m], A.g0[m], i, a0, a1, a2, z, lu, fc);
            else
                small_plan_entry<MOVE_SNOOKER>(na, N, D, S, pos, A.a[m], A.sigma[m], A.g0[m], i, a0, a1, a2, z, lu, fc);
            orders[e] = i;
            p0s[e] = a0;
            p1s[e] = a1;
            p2s[e] = a2;
            s0s[e] = z;
            logus[e] = lu;
            facs[e] = fc;
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            const int s = sb + b;
            const int m = move_of(s);                                            // workgroup-uniform
            const int kind = MOVESEL == SMALL_ANY_MOVE ? A.kind[m] : MOVESEL;
            const int S = A.nsplits[m];
            const double gam = A.gammas[m];
            GaussGen gg;
            gg.gseed = A.seed;
            gg.gstep = A.step0 + (unsigned long long)s;
            gg.gfac = A.step_fac ? A.step_fac[s] : 1.0;
            gg.gsigma = A.gsigma[m];
            gg.gscale = A.gscale[m];
            // ---- the half-steps: a barrier where the general path has a kernel boundary ----
            int pos0 = b * N;
            for (int split = 0; split < S; ++split) {
                const int ns = (N - split + S - 1) / S;
                if constexpr (DENSE) {
                    // dense Gaussian target: a wave takes 16 slots at a time -- proposals into its LDS tile, the f64 MFMA
                    // contraction and the decisions exactly as k_halfstep does them (same instructions, same order)
                    for (int base = wv * 16; base < ns; base += nwave * 16) {       // wave-uniform
                        const int nslot = min(16, ns - base);
#pragma unroll
                        for (int pp = 0; pp < PPT; ++pp) {
                            const int trow = pp * WPW + sub;
                            const bool live = trow < nslot;
                            const int pos = pos0 + base + (live ? trow : 0);
                            Row<G, V, CH> q;
                            double factor = 0.0;
                            bool badq = false;
                            const int i = orders[pos], j0 = p0s[pos], j1 = p1s[pos], j2 = p2s[pos];
                            if (!PLANNED && (MOVESEL == MOVE_GAUSS || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_GAUSS)))
                                small_propose<G, V, CH, MOVE_GAUSS>(A, Xs, live, i, j0, j1, j2, s0s[pos], facs[pos], gam, D, gl, sub, q, factor, badq, &gg);
                            else if (MOVESEL == MOVE_STRETCH || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_STRETCH))
                                small_propose<G, V, CH, MOVE_STRETCH>(A, Xs, live, i, j0, j1, j2, s0s[pos], facs[pos], gam, D, gl, sub, q, factor, badq);
                            else if (MOVESEL == MOVE_DE || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_DE))
                                small_propose<G, V, CH, MOVE_DE>(A, Xs, live, i, j0, j1, j2, s0s[pos], facs[pos], gam, D, gl, sub, q, factor, badq);
                            else
                                small_propose<G, V, CH, MOVE_SNOOKER>(A, Xs, live, i, j0, j1, j2, s0s[pos], facs[pos], gam, D, gl, sub, q, factor, badq);
#pragma unroll
                            for (int c = 0; c < CH; ++c)
#pragma unroll
                                for (int v = 0; v < V; ++v) {
                                    const int d = (c * G + gl) * V + v;
                                    if (d < Dp) tile[trow * RT + d] = (live && !badq) ? q.x[c][v] : muS[d];     // dead row: zero residual
                                }
                            if (gl == 0) facS[trow] = badq ? -__builtin_inf() : factor;
                        }
                        const int myrow = (lane >> 4) + 4 * (lane & 3);
                        const bool mine = (lane & 15) < 4 && myrow < nslot;
                        const int mypos = pos0 + base + (myrow < nslot ? myrow : 0);
                        const int my_i = orders[mypos];
                        const double my_lpo = lps[my_i], my_logu = logus[mypos];
                        EMX_WAVE_SYNC();
                        double my_qf;
                        {
                            const int am = lane & 15, ak = lane >> 4;
                            typedef double d4 __attribute__((ext_vector_type(4)));
                            double afr[KK];
#pragma unroll
                            for (int kk = 0; kk < KK; ++kk) afr[kk] = tile[am * RT + 4 * kk + ak] - muS[4 * kk + ak];
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
#if EMX_OPT_RED4
                            my_qf = row16_sum4(part[0], part[1], part[2], part[3], lane);
#else
#pragma unroll
                            for (int r = 0; r < 4; ++r) part[r] = group_sum<16>(part[r]);
                            my_qf = part[0];
#pragma unroll
                            for (int r = 1; r < 4; ++r) my_qf = (am == r) ? part[r] : my_qf;
#endif
                        }
                        bool acc = false;
                        if (mine) {
                            const double lpn = -0.5 * my_qf;
                            if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);
                            const double lnpdiff = facS[myrow] + lpn - my_lpo;
                            acc = lnpdiff > my_logu;
                            accs[my_i] = acc ? 1 : 0;
                            if (acc) lps[my_i] = lpn;
                        }
                        const unsigned long long am64 = __ballot(acc);       // bit (row & 3) * 16 + (row >> 2) <-> tile row
#pragma unroll
                        for (int pp = 0; pp < PPT; ++pp) {
                            const int row = pp * WPW + sub;
                            if (row < nslot && ((am64 >> ((row & 3) * 16 + (row >> 2))) & 1ull)) {
                                const int wrow = orders[pos0 + base + row];
#pragma unroll
                                for (int c = 0; c < CH; ++c)
#pragma unroll
                                    for (int v = 0; v < V; ++v) {
                                        const int d = (c * G + gl) * V + v;
                                        if (d < D) Xs[(size_t)wrow * D + d] = tile[row * RT + d];
                                    }
                            }
                        }
                        EMX_WAVE_SYNC();
                    }
                } else
                for (int base = wv * WPW; base < ns; base += nwave * WPW) {      // wave-uniform
                    const int t = base + sub;
                    const bool live = t < ns;
                    const int pos = pos0 + (live ? t : 0);
                    const int i = orders[pos], j0 = p0s[pos], j1 = p1s[pos], j2 = p2s[pos];
                    const double s0 = s0s[pos], fac = facs[pos], logu = logus[pos];
                    if (!PLANNED && (MOVESEL == MOVE_GAUSS || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_GAUSS)))
                        small_update<G, V, CH, MOVE_GAUSS>(A, Xs, lps, accs, live, i, j0, j1, j2, s0, fac, logu, gam, mu, iv, D, gl, sub, lane, &gg);
                    else if (MOVESEL == MOVE_STRETCH || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_STRETCH))
                        small_update<G, V, CH, MOVE_STRETCH>(A, Xs, lps, accs, live, i, j0, j1, j2, s0, fac, logu, gam, mu, iv, D, gl, sub, lane);
                    else if (MOVESEL == MOVE_DE || (MOVESEL == SMALL_ANY_MOVE && kind == MOVE_DE))
                        small_update<G, V, CH, MOVE_DE>(A, Xs, lps, accs, live, i, j0, j1, j2, s0, fac, logu, gam, mu, iv, D, gl, sub, lane);
                    else
                        small_update<G, V, CH, MOVE_SNOOKER>(A, Xs, lps, accs, live, i, j0, j1, j2, s0, fac, logu, gam, mu, iv, D, gl, sub, lane);
                }
                __syncthreads();
                pos0 += ns;
            }
            // ---- chain append (ensemble.py:416, backend.py:229) ----
            if (A.store && ((A.i0 + s + 1) % A.thin_by == 0)) {
                double* cr = A.chain + (size_t)row * N * D;
                double* cl = A.chain_lp + (size_t)row * N;
                for (int e = tid; e < N * D; e += T) cr[e] = Xs[e];
                for (int e = tid; e < N; e += T) {
                    cl[e] = lps[e];
                    acnt[e] += accs[e];
                }
                ++row;
                __syncthreads();                     // the next half-step overwrites what was just copied
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < N * D; e += T) A.X[e] = Xs[e];
    for (int e = tid; e < N; e += T) {
        A.lp[e] = lps[e];
        A.acc[e] = accs[e];
        A.acc_count[e] += acnt[e];
    }
}

// logs of a host-supplied plan (exact / inputs modes), full width: logu = ln(uacc),
// fac = (D-1) ln zz for the stretch move (stretch.py:31), 0 otherwise
static __global__ void k_plan_logs(int N, int D, int stretch, const double* __restrict__ s0, const double* __restrict__ uacc,
                            double* __restrict__ logu, double* __restrict__ fac) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= N) return;
    logu[pos] = plan_log(uacc[pos]);
    fac[pos] = stretch ? ((double)D - 1.0) * plan_log(s0[pos]) : 0.0;
}

// The plans of up to sixteen steps fetched in ONE launch straight from the pipeline's pinned staging buffers (exact mode with the
// persistent kernels: a copy + k_plan_logs + two event operations PER STEP cost the host 25 us a step, more than the kernels):
// device slot <- [order | p0] [s0 | uacc] ([p1 | p2]), logu and fac as k_plan_logs computes them.
struct PlanFetchArgs {
    const char* host[16];     // pinned staging buffers (HostPlan layout: [order|p0] int32, [s0|uacc] f64, [p1|p2] int32)
    char* dev[16];            // device slot blocks ([order|p0] [s0|uacc] [p1|p2] [logu|fac])
    int32_t stretch[16];      // fac = (D-1) ln zz
    int32_t peers[16];        // the [p1|p2] columns are part of the plan
    int32_t kind[16];         // 0: a finished plan (values; the logs are taken here); 1: a raw step -- generator words in the columns, copied as
                              // they are (k_plan_raw_batch converts them behind this kernel); 2: a regen step -- `order` and, in p0's place,
                              // nkey[b] words of generator states (k_plan_regen_batch, then k_plan_raw_batch)
    int32_t nkey[16];
    int32_t N, D, n;
    unsigned* arrived;        // device: [0] workgroups of this launch that are done, [1] tickets handed out (the last one resets both)
    const unsigned* avoid_xcc;        // device word: XCC_ID + 1 of the XCD a one-XCD persistent launch lives on (0: none), or null
    int32_t pieces_x;                 // the work: pieces_x * n pieces of 256 entries (tickets), whatever the grid
    unsigned long long* host_done;    // pinned host word: steps fetched so far, written by the LAST workgroup (an event query
    unsigned delay_ticks;             // (tests: the kernel idles this long first -- 100 MHz ticks -- to show that its consumer waits for it)
    unsigned long long done_value;    // shows the completion tens of us late: the staging buffers go back to the producers on this word)
};
static __device__ __forceinline__ void plan_fetch_rows(const PlanFetchArgs& A, int b, int pos);
// The work -- gridDim.x * gridDim.y pieces of 256 entries -- is handed out by TICKET (A.arrived[1]), not by blockIdx, so that a
// workgroup may decline: one that finds itself on the XCD a one-XCD persistent launch lives on (A.avoid_xcc, written by that
// launch's handshake) leaves at once.  This kernel runs on the upload stream while the consumer's launches start and end, and a
// persistent workgroup takes every vector register of its CU (k_persist_mix: 8 waves x 249): a CU that holds as much as one wave
// of this kernel -- waiting 2 us and more for pinned memory -- cannot take it, and the launch's first barrier waits for the whole
// fetch (measured: 4 096 walkers, DE + snooker, persistent launches 209 us each against 178 with Philox plans).  The last
// workgroup to arrive does whatever tickets are left (none, unless every other one declined) before it tells the host.
// Beside a DEVICE-WIDE launch (no XCD to stay off) the host starts few workgroups instead (tuning "fetch_blocks"): each of them
// -- four waves of 32 registers, one a SIMD -- fits next to a k_persist workgroup (464 of a SIMD's 512 registers), two on one CU do
// not, and a grid of a thousand workgroups keeps arriving on whatever CU has room until the fetch is over: the launch then starts
// when the fetch ends (16 384 x 64, exact mode: 6.3 MB of plans per launch, 126 us over PCIe).  (The copy engine instead -- one
// hipMemcpyAsync per step, the logarithms from the device copies -- was measured too: slower than 64 workgroups,
// profiles/r05/exact_mix_probe.txt.)
static __global__ __launch_bounds__(256) void k_plan_fetch(const PlanFetchArgs A) {
    __shared__ unsigned tk_s;
    const unsigned total = (unsigned)A.pieces_x * (unsigned)A.n, nwg = gridDim.x;      // tickets | workgroups (a 1-D grid)
    if (A.delay_ticks) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < A.delay_ticks) __builtin_amdgcn_s_sleep(8);
    }
    bool decline = false;
    if (A.avoid_xcc && nwg >= 16u) {                            // uniform
        const unsigned want = __hip_atomic_load(A.avoid_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        decline = want != 0u && (xcc & 15u) + 1u == want;
    }
    bool last = false;
    for (;;) {
        if (!decline) {
            for (;;) {
                if (threadIdx.x == 0) tk_s = atomicAdd(A.arrived + 1, 1u);
                __syncthreads();
                const unsigned tk = tk_s;
                __syncthreads();
                if (tk >= total) break;
                const int pos = (int)(tk % (unsigned)A.pieces_x) * (int)blockDim.x + (int)threadIdx.x;
                if (pos < A.N) plan_fetch_rows(A, (int)(tk / (unsigned)A.pieces_x), pos);
            }
        }
        if (last) break;
        // every load of this workgroup has returned (its values were stored): count it; the last one tells the host
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            tk_s = atomicAdd(A.arrived, 1u) == nwg - 1u ? 1u : 0u;
        }
        __syncthreads();
        last = tk_s != 0u;
        __syncthreads();
        if (!last) return;
        decline = false;                                        // the last arriver sweeps up (normally: no ticket left)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        A.arrived[0] = 0u;
        A.arrived[1] = 0u;
        __hip_atomic_store(A.host_done, A.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static __device__ __forceinline__ void plan_fetch_rows(const PlanFetchArgs& A, int b, int pos) {
    const size_t N = (size_t)A.N;
    const int32_t* hi = reinterpret_cast<const int32_t*>(A.host[b]);
    const double* hd = reinterpret_cast<const double*>(A.host[b] + N * 8);
    int32_t* di = reinterpret_cast<int32_t*>(A.dev[b]);
    double* dd = reinterpret_cast<double*>(A.dev[b] + N * 8);
    if (A.kind[b] == 2) {                       // (uniform per step)
        di[pos] = hi[pos];
        if (pos < A.nkey[b]) di[N + pos] = hi[N + pos];
        return;
    }
    if (A.kind[b] == 1) {
        di[pos] = hi[pos];
        di[N + pos] = hi[N + pos];
        reinterpret_cast<uint2*>(dd)[pos] = reinterpret_cast<const uint2*>(hd)[pos];            // (words, not doubles: no floating-point move)
        reinterpret_cast<uint2*>(dd)[N + pos] = reinterpret_cast<const uint2*>(hd)[N + pos];
        return;
    }
    const double z = hd[pos], u = hd[N + pos];
    di[pos] = hi[pos];
    di[N + pos] = hi[N + pos];
    dd[pos] = z;
    dd[N + pos] = u;
    if (A.peers[b]) {
        const int32_t* hp = reinterpret_cast<const int32_t*>(A.host[b] + N * 24);
        int32_t* dp = reinterpret_cast<int32_t*>(A.dev[b] + N * 24);
        dp[pos] = hp[pos];
        dp[N + pos] = hp[N + pos];
    }
    double* dl = reinterpret_cast<double*>(A.dev[b] + N * 32);
    dl[pos] = plan_log(u);
    dl[N + pos] = A.stretch[b] ? ((double)A.D - 1.0) * plan_log(z) : 0.0;
}

// Exact mode, large ensembles (round 6): the fixed-length draws of a stretch step MADE AGAIN on the device.  The 5 N words behind the
// shuffle -- per split rand(Ns), randint(Nc, Ns) with Nc a power of two, Ns x rand(): stretch.py:30-32, red_blue.py:100, contiguous in
// the reference's stream -- used to cross from the generator's core to the tokenizer's, into the pinned staging buffer and over PCIe
// (1.3 MB a step at 65 536 walkers: what bounded the host pipeline, profiles/r05/exact_c2.md).  Now the tokenizer steps over them and
// passes on the generator STATE at every REGEN_SB-th block of the region (the plan's p0 column holds them: 624 words each); one
// workgroup per state runs MT19937's recurrence REGEN_SB - 1 blocks forward and drops every word where k_plan_raw expects it,
// untempered: the pairs of rand() in the s0 / uacc columns, the randint words in the p1 column (p0 is being read).  Same words, same
// places as the host finisher's copies: the plans stay bit-identical (tests/test_gpu_regen.py).
constexpr int REGEN_SB = 8;            // (= PIPE_REGEN_SB, emx_mtpipe.hpp)
struct PlanRegenArgs {
    char* dev;                         // device slot block ([order|p0] [s0|uacc] [p1|p2] [logu|fac])
    int32_t N, ns0;                    // walkers; members of split 0 (split 1: N - ns0)
    int32_t off;                       // the region's first word inside the first state's block
    int32_t nseg;
};
static __device__ __forceinline__ uint32_t regen_mix(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);
}
static __device__ __forceinline__ void plan_regen_body(const PlanRegenArgs& A, uint32_t (*key)[624]) {
    const int tid = threadIdx.x, seg = blockIdx.x;
    const size_t N = (size_t)A.N;
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(A.dev + N * 4);
    for (int i = tid; i < 624; i += 256) key[0][i] = keys[(size_t)seg * 624 + i];
    __syncthreads();
    uint32_t* wz = reinterpret_cast<uint32_t*>(A.dev + N * 8);        // s0 column: two words a slot
    uint32_t* wu = reinterpret_cast<uint32_t*>(A.dev + N * 16);       // uacc column
    uint32_t* wr = reinterpret_cast<uint32_t*>(A.dev + N * 24);       // p1 column: one word a slot
    const long long F0 = 5ll * A.ns0, F = 5ll * (long long)A.N;
    int cur = 0;
    for (int b = 0; b < REGEN_SB; ++b) {
        const long long r0 = ((long long)seg * REGEN_SB + b) * 624 - A.off;     // region index of this block's first word
        if (r0 >= F) break;                                                      // (uniform)
        for (int i = tid; i < 624; i += 256) {
            const long long r = r0 + i;
            if (r < 0 || r >= F) continue;
            const int split = r >= F0;
            const long long q = split ? r - F0 : r;
            const int base = split ? A.ns0 : 0, ns = split ? A.N - A.ns0 : A.ns0;
            const uint32_t w = key[cur][i];
            if (q < 2ll * ns)
                wz[2 * (size_t)base + (size_t)q] = w;                  // stretch.py:30 rand(Ns)
            else if (q < 3ll * ns)
                wr[(size_t)base + (size_t)(q - 2ll * ns)] = w;         // stretch.py:32 randint(Nc, Ns)
            else
                wu[2 * (size_t)base + (size_t)(q - 3ll * ns)] = w;     // red_blue.py:100 rand() per walker
        }
        if (b + 1 < REGEN_SB) {
            // one twist: three dependent phases of <= 227 words (word kk needs the NEW word kk - 227)
            const uint32_t* o = key[cur];
            uint32_t* n = key[cur ^ 1];
            if (tid < 227) n[tid] = o[tid + 397] ^ regen_mix(o[tid], o[tid + 1]);
            __syncthreads();
            if (tid < 227) {
                const int kk = tid + 227;
                n[kk] = n[kk - 227] ^ regen_mix(o[kk], o[kk + 1]);
            }
            __syncthreads();
            if (tid < 169) {
                const int kk = tid + 454;
                n[kk] = n[kk - 227] ^ regen_mix(o[kk], o[kk + 1]);
            } else if (tid == 169) {
                n[623] = n[396] ^ regen_mix(o[623], n[0]);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
}
static __global__ __launch_bounds__(256) void k_plan_regen(const PlanRegenArgs A) {
    __shared__ uint32_t key[2][624];
    plan_regen_body(A, key);
}
// ... of up to sixteen steps at once (the persistent launches' fetch: blockIdx.y = step; a step that is no regen step has nseg = 0)
struct PlanRegenBatchArgs {
    PlanRegenArgs st[16];
};
static __global__ __launch_bounds__(256) void k_plan_regen_batch(const PlanRegenBatchArgs B) {
    __shared__ uint32_t key[2][624];
    const PlanRegenArgs& A = B.st[blockIdx.y];
    if ((int)blockIdx.x >= A.nseg) return;          // (uniform)
    plan_regen_body(A, key);
}

// Device finish of an exact-mode stretch plan (round 5; csrc/emx_mtpipe.hpp, PipeStepInfo::raw).  The host pipeline hands over what
// only it can make -- `order` (the shuffled split, red_blue.py:76-85) and, where the complement's size is not a power of two, the
// accepted randint values -- and passes every fixed-length draw on as it left the generator: MT19937 STATE words, copied into the
// plan's own columns (the two words of a walker's stretch uniform where its s0 will stand, those of its accept uniform in uacc, the
// randint word in p0).  The upload is the one copy it always was; this kernel then tempers, converts exactly like
// RandomState.random_sample, resolves the partner through `order` (stretch.py:27,32) and writes the logs k_plan_logs would -- in place.
// Same arithmetic as the host finisher (emx_mtpipe.cpp: convert_pairs, convert_pairs_zz): IEEE multiply / add / divide, no contraction.
constexpr int PLAN_RAW_SPLITS = 8;
struct PlanRawArgs {
    char* dev;                         // device slot block ([order|p0] [s0|uacc] [p1|p2] [logu|fac])
    double a;                          // the stretch scale (stretch.py:30)
    int32_t off[PLAN_RAW_SPLITS + 1];
    int32_t N, D, S, wr_words;         // wr_words: p0 holds generator words (power-of-two complement), else accepted randint values
    int32_t wr_p1;                     // 1: the randint words stand in the p1 column (k_plan_regen put them there: p0 held the generator states)
};
static __device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static __device__ __forceinline__ double mt_pair_double(uint32_t w0, uint32_t w1) {
    const int32_t a = (int32_t)(mt_temper(w0) >> 5), b = (int32_t)(mt_temper(w1) >> 6);
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
static __device__ __forceinline__ void plan_raw_body(const PlanRawArgs& A) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= A.N) return;
    const size_t N = (size_t)A.N;
    int s = 0;
    while (s + 1 < A.S && pos >= A.off[s + 1]) ++s;
    const int base = A.off[s], ns = A.off[s + 1] - base;
    int32_t* di = reinterpret_cast<int32_t*>(A.dev);
    double* dd = reinterpret_cast<double*>(A.dev + N * 8);
    double* dl = reinterpret_cast<double*>(A.dev + N * 32);
    const uint2 wz = reinterpret_cast<const uint2*>(dd)[pos], wu = reinterpret_cast<const uint2*>(dd)[N + pos];
    const double u = mt_pair_double(wz.x, wz.y);
    const double tt = (A.a - 1.0) * u + 1.0;                    // stretch.py:30  ((a - 1) * rand + 1) ** 2 / a
    const double zz = tt * tt / A.a;
    const double ua = mt_pair_double(wu.x, wu.y);               // red_blue.py:100
    const uint32_t w = A.wr_p1 ? reinterpret_cast<const uint32_t*>(A.dev + N * 24)[pos] : (uint32_t)di[N + pos];
    const int32_t r = A.wr_words ? (int32_t)(mt_temper(w) & (uint32_t)(A.N - ns - 1)) : (int32_t)w;          // stretch.py:32 randint(Nc)
    di[N + pos] = r < base ? di[r] : di[r + ns];                // stretch.py:27: c = the other sets' members in plan order
    dd[pos] = zz;
    dd[N + pos] = ua;
    dl[pos] = plan_log(ua);
    dl[N + pos] = ((double)A.D - 1.0) * plan_log(zz);
}
static __global__ __launch_bounds__(256) void k_plan_raw(const PlanRawArgs A) { plan_raw_body(A); }
// ... of up to sixteen steps at once (blockIdx.y = step; a step that was handed over finished has N = 0)
struct PlanRawBatchArgs {
    PlanRawArgs st[16];
};
static __global__ __launch_bounds__(256) void k_plan_raw_batch(const PlanRawBatchArgs B) { plan_raw_body(B.st[blockIdx.y]); }

// ----------------------------------------------------------------------------------------
// Split-phase accept/commit (target evaluated on the host: arbitrary Python log_prob_fn).
// new_lp[t], fout[t], qout[t] are slot-indexed; red_blue.py:96-104.
// ----------------------------------------------------------------------------------------
struct AcceptArgs {
    double* X;
    double* lp;
    uint8_t* acc;
    uint32_t* acc_count;
    double* chain;
    double* chain_lp;
    uint32_t* status;
    const double* qout;
    const double* fout;
    const double* new_lp;
    const int32_t* order;
    const double* uacc;
    int32_t N, D, S, split, pos0, ns, move;
};

static __global__ __launch_bounds__(256) void k_accept(const AcceptArgs A) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= A.ns) return;
    const int i = A.order[A.pos0 + t];
    const double uacc = A.uacc[A.pos0 + t];
    const double nlp = A.new_lp[t];
    const double lp_old = A.lp[i];
    if (nlp != nlp) raise_status(A.status, ST_NAN_LOGP);
    const double lnpdiff = A.fout[t] + nlp - lp_old;
    const bool accept = lnpdiff > plan_log(uacc);
    const double* q = A.qout + (size_t)t * A.D;
    double* xr = A.X + (size_t)i * A.D;
    if (accept)
        for (int d = lane; d < A.D; d += 64) xr[d] = q[d];
    if (A.chain)
        for (int d = lane; d < A.D; d += 64) __builtin_nontemporal_store(accept ? q[d] : xr[d], &A.chain[(size_t)i * A.D + d]);
    if (lane == 0) {
        if (accept) A.lp[i] = nlp;
        A.acc[i] = accept ? 1 : 0;
        if (A.chain_lp) {
            A.chain_lp[i] = accept ? nlp : lp_old;
            if (accept) A.acc_count[i] += 1u;
        }
    }
}

struct GaussDispArgs {
    double* disp;             // (N, D)
    const double* scale;      // (D) standard deviations, or nullptr: isotropic `sigma`
    const int32_t* col;       // plan p0 (slot == walker), used when mode != GAUSS_VECTOR
    double sigma, f;          // f: this step's step-size factor (gaussian.py:81-84), 1 when unused
    uint64_t seed, step;
    int32_t N, D, mode;
};

// native draws: disp[w][d] = (f * scale_d) * n(w, d).  One thread per (walker, coordinate pair) in the
// vector mode; in the one-coordinate modes one thread per walker writes just the coordinate that moves.
static __global__ __launch_bounds__(256) void k_gauss_disp(const GaussDispArgs A) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int npair = (A.D + 1) / 2;
    if (A.mode == GAUSS_VECTOR) {
        if (tid >= (long long)A.N * npair) return;
        const int w = (int)(tid / npair), pr = (int)(tid - (long long)w * npair);
        double n0, n1;
        native_gauss_pair(A.seed, A.step, w, pr, n0, n1);
        const int d0 = 2 * pr, d1 = d0 + 1;
        const double s0 = A.f * (A.scale ? A.scale[d0] : A.sigma);
        A.disp[(size_t)w * A.D + d0] = s0 * n0;
        if (d1 < A.D) {
            const double s1 = A.f * (A.scale ? A.scale[d1] : A.sigma);
            A.disp[(size_t)w * A.D + d1] = s1 * n1;
        }
    } else {
        if (tid >= A.N) return;
        const int w = (int)tid, d = A.col[w];
        double n0, n1;
        native_gauss_pair(A.seed, A.step, w, d >> 1, n0, n1);
        const double sc = A.f * (A.scale ? A.scale[d] : A.sigma);
        A.disp[(size_t)w * A.D + d] = sc * ((d & 1) ? n1 : n0);
    }
}

// host-supplied normals (exact / inputs modes): in place, disp = (f * scale_d) * n -- gaussian.py:87's
// left-to-right product
static __global__ __launch_bounds__(256) void k_gauss_scale(const GaussDispArgs A) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)A.N * A.D) return;
    const int d = (int)(tid % A.D);
    const double sc = A.f * (A.scale ? A.scale[d] : A.sigma);
    A.disp[tid] = sc * A.disp[tid];
}

// Native-mode plans, evaluated full width (one lane per walker and step): the Philox rounds, the keyed-
// permutation inversions and the two f64 logs are paid once per walker here, not on the few lanes a
// half-step group would spare.  Up to 8 consecutive steps per launch (grid.y = step): one step is only N
// lanes of latency-bound work, so batching fills the machine and amortises the launch.  emx_plan_get
// returns these arrays to the parity tests.
#ifndef EMX_NATIVE_BATCH
#define EMX_NATIVE_BATCH 16     // 8 -> 16: +0.9 % at C2 (the plan kernel's launch is amortised over twice the steps); 32: +0.6 % more for twice the ring memory
#endif
constexpr int NATIVE_BATCH_MAX = EMX_NATIVE_BATCH;      // native plans evaluated per k_native_plan_batch launch
struct NativeBatchArgs {
    NativeArgs nat[NATIVE_BATCH_MAX];
    int32_t* order[NATIVE_BATCH_MAX];
    int32_t* p0[NATIVE_BATCH_MAX];
    int32_t* p1[NATIVE_BATCH_MAX];
    int32_t* p2[NATIVE_BATCH_MAX];
    double* s0[NATIVE_BATCH_MAX];
    double* uacc[NATIVE_BATCH_MAX];
    double* logu[NATIVE_BATCH_MAX];
    double* fac[NATIVE_BATCH_MAX];
    double a[NATIVE_BATCH_MAX], sigma[NATIVE_BATCH_MAX], g0[NATIVE_BATCH_MAX];
    int32_t move[NATIVE_BATCH_MAX], S[NATIVE_BATCH_MAX];
    int32_t gmode[NATIVE_BATCH_MAX], gcol[NATIVE_BATCH_MAX];   // Gaussian move: mode, the sequential mode's column
    int32_t N, D, nb;
    int32_t lean;           // 1: only the columns the fused half-step kernel of the step's move reads are written (below)
    int32_t ablate;         // timing experiments only (tuning "ablate" bits 8..): 1 no logs, 2 no stores, 4 return at once; 0 in production
    const StepDesc* desc;   // graph replay: per-step NativeArgs from device memory instead of nat[]
};

template <bool ONLY_STRETCH>
static __device__ __forceinline__ void native_plan_batch_body(const NativeBatchArgs& B) {
    const int b = blockIdx.y;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= B.N || (EMX_EXPERIMENTS && !ONLY_STRETCH && (B.ablate & 4))) return;
    const int N = B.N, S = B.S[b];
    int split = 0, t = pos;
    const SplitSizes sz = split_sizes(N, S);
    for (int s = 0; s < S; ++s) {
        const int n = sz.of(s);
        if (t < n) { split = s; break; }
        t -= n;
    }
    int i, a0, a1, a2;
    double z, u;
    const int mv = ONLY_STRETCH ? MOVE_STRETCH : B.move[b];
    const NativeArgs nat = (!ONLY_STRETCH && B.desc) ? B.desc[b].nat : B.nat[b];      // (graph replay: descriptors in device memory)
    if (mv == MOVE_GAUSS)
        native_gauss_slot(nat, B.D, B.gmode[b], B.gcol[b], pos, i, a0, a1, a2, z, u);
    else if (mv == MOVE_STRETCH)
        native_slot<MOVE_STRETCH>(nat, N, S, split, t, B.a[b], B.sigma[b], B.g0[b], i, a0, a1, a2, z, u);
    else if (mv == MOVE_DE)
        native_slot<MOVE_DE>(nat, N, S, split, t, B.a[b], B.sigma[b], B.g0[b], i, a0, a1, a2, z, u);
    else
        native_slot<MOVE_SNOOKER>(nat, N, S, split, t, B.a[b], B.sigma[b], B.g0[b], i, a0, a1, a2, z, u);
    // This kernel is bound by what it WRITES (48 bytes per entry at ~3 TB/s: 16.8 us per 16 steps at 65 536 walkers, 75 us at
    // 262 144), not by its arithmetic.  The fused half-step reads order, p0 (+ p1 for DE, + p1, p2 for the snooker move), s0
    // (not the snooker move), logu and fac -- never uacc, whose logarithm it is handed.  A lean plan leaves the other columns
    // unwritten: 32 instead of 48 bytes per entry for the stretch move.  Whoever wants them (emx_plan_get: the parity tests; the
    // split-phase and sharded paths) gets a full plan.  (Streaming stores for these columns -- written once, read once -- were
    // measured: the half-step then fetches its plan entries from HBM instead of the Infinity Cache, C2 23.61 -> 24.19 us/step,
    // C3 stored 57.5 -> 61.5; profiles/r03/ab_nt_plan_stores.txt.)
    const bool full = ONLY_STRETCH ? false : !B.lean;      // (the stretch-only form: lean plans, no timing switches)
    if (EMX_EXPERIMENTS && !ONLY_STRETCH && (B.ablate & 2)) {                                  // timing experiments: keep the arithmetic alive, write one word
        if (i + a0 + a1 + a2 == -12345 && z + u == 1.2345e-300) B.order[b][pos] = i;
        return;
    }
    if (EMX_EXPERIMENTS && !ONLY_STRETCH && (B.ablate & 1)) {
        B.order[b][pos] = i;
        B.p0[b][pos] = a0;
        B.s0[b][pos] = z;
        B.logu[b][pos] = u;
        B.fac[b][pos] = z;
        return;
    }
    B.order[b][pos] = i;
    B.p0[b][pos] = a0;
    if (full || mv == MOVE_DE || mv == MOVE_SNOOKER) B.p1[b][pos] = a1;
    if (full || mv == MOVE_SNOOKER) B.p2[b][pos] = a2;
    if (full || mv != MOVE_SNOOKER) B.s0[b][pos] = z;
    if (full) B.uacc[b][pos] = u;
    B.logu[b][pos] = plan_log_uniform(u);
    B.fac[b][pos] = (mv == MOVE_STRETCH) ? ((double)B.D - 1.0) * plan_log(z) : 0.0;
}

static __global__ __launch_bounds__(256) void k_native_plan_batch(const NativeBatchArgs B) { native_plan_batch_body<false>(B); }

// The same plans for a batch of lean stretch steps alone, without the other moves' branches and the
// descriptor path: 32 vector registers against 80.  (Round 5 also ran it on a low-priority stream of its own NEXT to the resident
// persistent launch that reads the batch before -- it fits the 48 registers per SIMD a CU has left beside a k_persist workgroup --
// with launches ending at batch boundaries: the kernel stretches from 14 to 59 us, the persistent launches slow down by what it
// no longer costs in front of them, C2 21.07-21.11 against 20.94-21.09 us/step; profiles/r05/plan_side_ab.txt.  Dropped.)
static __global__ __launch_bounds__(256) void k_native_plan_batch_stretch(const NativeBatchArgs B) { native_plan_batch_body<true>(B); }

static __global__ void k_graph_set(GraphCounters* ctr, unsigned long long step_base, long long stored_base) {
    ctr->step_base = step_base;
    ctr->stored_base = stored_base;
}

// First node of the captured 8-step graph: derive this replay's per-step descriptors from the device
// counters and advance them (one block; every lane reads the counters before lane 0 updates them).
static __global__ void k_graph_advance(GraphCounters* ctr, StepDesc* desc, unsigned long long seed, int N, int nb, int store) {
    const int b = threadIdx.x;
    const unsigned long long base = ctr->step_base;
    const long long sb = ctr->stored_base;
    __syncthreads();
    if (b < nb) {
        desc[b].nat.seed = seed;
        desc[b].nat.step = base + (unsigned long long)b;
        desc[b].nat.pk = make_perm_key((uint64_t)N, seed, base + (unsigned long long)b);
        desc[b].stored_idx = store ? sb + b : -1;
    }
    if (b == 0) {
        ctr->step_base = base + (unsigned long long)nb;
        if (store) ctr->stored_base = sb + nb;
    }
}

// sharded runs: write the all-gathered [row | log_prob | accepted] records of the other ranks'
// slots into the local replica (and the stored chain step, if any)
struct ScatterArgs {
    double* X;
    double* lp;
    uint8_t* acc;
    uint32_t* acc_count;
    double* chain;
    double* chain_lp;
    const double* gathered;   // virtual (ns, D + 2) array in slot order
    const int32_t* order;
    int32_t N, D, S, split, pos0, t_lo, t_hi;
};

static __global__ __launch_bounds__(256) void k_scatter_rows(const ScatterArgs A) {
    const int lane = threadIdx.x & 63;
    const int t = A.t_lo + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= A.t_hi) return;
    const int i = A.order[A.pos0 + t];
    const double* src = A.gathered + (size_t)t * (A.D + 2);
    double* dst = A.X + (size_t)i * A.D;
    for (int d = lane; d < A.D; d += 64) dst[d] = src[d];
    if (A.chain)
        for (int d = lane; d < A.D; d += 64) A.chain[(size_t)i * A.D + d] = src[d];
    if (lane == 0) {
        const double l = src[A.D];
        const bool ac = src[A.D + 1] != 0.0;
        A.lp[i] = l;
        A.acc[i] = ac ? 1 : 0;
        if (A.chain_lp) {
            A.chain_lp[i] = l;
            if (ac) A.acc_count[i] += 1u;
        }
    }
}

// ----------------------------------------------------------------------------------------
// Pull exchange (walker-block ownership).  Rank r owns walkers [N r / G, N (r+1) / G); the RNG
// plan is replicated, so every rank can tell, without asking, which of ITS rows the walkers
// other ranks update in this half-step will read.  Per half-step:
//   k_pull_plan    -> compact plan of the slots whose walker this rank owns (+ device-side count) and, per
//                     destination rank, up to cap records [global row index | row] (unused records: NaN index)
//   (all-to-all, cap records per pair)
//   k_pull_scatter -> received rows into the local replica at their global index; re-arms the send records and counters
//   k_halfstep     -> over the compact plan
// ----------------------------------------------------------------------------------------
__host__ __device__ inline int block_owner(int w, int N, int G) {     // r with N r / G <= w < N (r+1) / G
    return (int)((((long long)w + 1) * G - 1) / N);
}

struct PullPlanArgs {
    const int32_t *order, *p0, *p1, *p2;      // this step's plan (slot-indexed), already offset to the split
    const double *s0, *uacc, *logu, *fac;
    int32_t *corder, *cp0, *cp1, *cp2;        // compact plan of this rank's active walkers
    double *cs0, *cuacc, *clogu, *cfac;
    int32_t* counts;                          // this half-step's counters: [0] own active slots; [1 + q] records for rank q
    const double* X;                          // own rows are copied straight into the send records
    double* rec;                              // [G][cap] records of D + 1 doubles: [row index | row]; unused records keep a NaN index
    uint32_t* status;
    int32_t N, D, G, rank, ns, npart, cap;
};

// One launch per half-step does what k_pull_plan + k_pull_pack used to: the compact plan of the own slots and the records
// of the rows the peers will read, copied by the whole wave, one row per iteration.  `cap` is the same for every
// half-step of a context (the largest any installed move needs), so a record always sits at the same address: unused
// records carry a NaN index, which k_pull_scatter restores after the exchange (no memset on the stream).
static __global__ __launch_bounds__(256) void k_pull_plan(const PullPlanArgs A) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = t < A.ns;
    int i = 0, pj[3] = {0, 0, 0}, oi = -1;
    if (live) {
        i = A.order[t];
        pj[0] = A.p0[t];
        pj[1] = A.p1[t];
        pj[2] = A.p2[t];
        oi = block_owner(i, A.N, A.G);
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    const size_t recw = (size_t)A.D + 1;
    // (a) slots of walkers this rank owns -> compact plan, one atomic per wave
    {
        const bool mine = live && oi == A.rank;
        const unsigned long long m = __ballot(mine);
        if (m) {
            int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(&A.counts[0], __popcll(m));
            base = __shfl(base, leader);
            if (mine) {
                const int e = base + __popcll(m & below);
                A.corder[e] = i;
                A.cp0[e] = pj[0];
                A.cp1[e] = pj[1];
                A.cp2[e] = pj[2];
                A.cs0[e] = A.s0[t];
                A.cuacc[e] = A.uacc[t];
                A.clogu[e] = A.logu[t];
                A.cfac[e] = A.fac[t];
            }
        }
    }
    // (b) partners owned here of walkers updated elsewhere -> a [row index | row] record in the owner's block
    for (int j = 0; j < A.npart; ++j) {
        const bool here = live && oi != A.rank && block_owner(pj[j], A.N, A.G) == A.rank;
        unsigned long long any = __ballot(here);
        if (!any) continue;
        for (int q = 0; q < A.G; ++q) {                         // wave-uniform
            const bool hit = here && oi == q;
            unsigned long long m = __ballot(hit);
            if (!m) continue;
            int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(&A.counts[1 + q], __popcll(m));
            base = __shfl(base, leader);
            const int e = base + __popcll(m & below);
            if (hit && e >= A.cap) raise_status(A.status, ST_EXCHANGE_OVERFLOW);
            while (m) {                                         // one requested row per iteration, copied by the whole wave
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int eb = __shfl(e, b), idx = __shfl(pj[j], b);
                if (eb >= A.cap) continue;
                double* dst = A.rec + ((size_t)q * A.cap + eb) * recw;
                const double* src = A.X + (size_t)idx * A.D;
                if (lane == 0) dst[0] = (double)idx;
                for (int d = lane; d < A.D; d += 64) dst[1 + d] = src[d];
            }
        }
    }
}

struct PullRowsArgs {
    double* X;
    const double* rec;           // received: [G][cap] records, the peers' rows this rank's walkers will read
    double* sent;                // this rank's send records: their index fields go back to NaN for the next half-step
    int32_t* counts_next;        // the other counter buffer: zeroed here for the next half-step
    int32_t N, D, G, rank, cap;
};

// 16 lanes per record
static __global__ __launch_bounds__(256) void k_pull_scatter(const PullRowsArgs A) {
    const int r = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k <= A.G; k += blockDim.x) A.counts_next[k] = 0;      // world sizes up to EMX_MAX_RANKS
    if (r >= A.G * A.cap) return;
    const size_t o = (size_t)r * (A.D + 1);
    if (r / A.cap != A.rank) {
        const double h = A.rec[o];
        if (h >= 0.0) {                                       // NaN: unused record
            double* dst = A.X + (size_t)(long long)h * A.D;
            for (int d = l; d < A.D; d += 16) dst[d] = A.rec[o + 1 + d];
        }
    }
    if (l == 0) A.sent[o] = __builtin_nan("");
}

// initial state of the send records (and after the installed moves changed the capacity)
static __global__ __launch_bounds__(256) void k_pull_reset(double* sent, int nrec, int D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrec) sent[(size_t)r * (D + 1)] = __builtin_nan("");
}

// ----------------------------------------------------------------------------------------
// Direct exchange: nothing is packed or sent.  Every rank owns a walker block; the half-step kernel reads partner rows
// straight from the owner's HBM (HalfStepArgs::peerX).  What remains per step is
//   k_own_plan      -> per split, the compact plan of the slots whose walker this rank owns (counts on the device)
//   k_peer_barrier  -> before every half-step: "my previous half-step is complete and visible" to every peer, then wait for
//                      the same from every peer.  It orders both hazards: a partner row must carry its owner's last
//                      update, and nobody may start writing split k+1 rows while a peer still reads them as partners.
// ----------------------------------------------------------------------------------------
struct OwnPlanArgs {
    const int32_t *order, *p0, *p1, *p2;      // this step's full plan
    const double *s0, *uacc, *logu, *fac;
    int32_t *corder, *cp0, *cp1, *cp2;        // compact plan: split s at offset off[s], counts[s] entries
    double *cs0, *cuacc, *clogu, *cfac;
    int32_t* counts;                          // [S]
    int32_t off[66];
    int32_t N, G, rank, S;
};

static __global__ __launch_bounds__(256) void k_own_plan(const OwnPlanArgs A) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = t < A.N;
    int i = 0, s = -1;
    if (live) {
        i = A.order[t];
        if (block_owner(i, A.N, A.G) == A.rank) {
            s = 0;
            while (s + 1 < A.S && t >= A.off[s + 1]) ++s;
        }
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(s >= 0);
    while (todo) {                                                  // wave-uniform: one pass per split present in the wave
        const int leader = __ffsll((long long)todo) - 1;
        const int sp = __shfl(s, leader);
        const bool hit = s == sp;
        const unsigned long long m = __ballot(hit);
        int base = 0;
        if (lane == leader) base = atomicAdd(&A.counts[sp], __popcll(m));
        base = __shfl(base, leader);
        if (hit) {
            const int e = A.off[sp] + base + __popcll(m & below);
            A.corder[e] = i;
            A.cp0[e] = A.p0[t];
            A.cp1[e] = A.p1[t];
            A.cp2[e] = A.p2[t];
            A.cs0[e] = A.s0[t];
            A.cuacc[e] = A.uacc[t];
            A.clogu[e] = A.logu[t];
            A.cfac[e] = A.fac[t];
        }
        todo &= ~m;
    }
}

// ----------------------------------------------------------------------------------------
// Replay exchange (full replicas, slot-range ownership): what a half-step sends is its DECISIONS -- 8 bytes per walker-update,
// never a coordinate.  Rank r proposes / evaluates / accepts the slots [ns r / G, ns (r+1) / G) of the split (k_halfstep, fused)
// and writes for each the new log-prob, or NaN when the proposal was rejected; one all-gather replicates those decisions; then
// every rank REPLAYS the accepted updates of the others on its own replica: the proposal of slot t is a function of the rows
// of the replica (identical on every rank before the half-step), of the replicated plan and of nothing else, so recomputing it
// gives the bits the owner committed.  Extra HBM traffic per rank: the accepted fraction of the other ranks' slots
// (24 D + 8 bytes each) -- the price of never touching xGMI with a row.
//   k_replay_compact -> the compact plan of the accepted slots of the other ranks (its logu column = their new log-probs),
//                       the accepted flags of ALL foreign slots, and the count on the device
//   k_halfstep<..., target = TGT_REPLAY> over the compact plan
// ----------------------------------------------------------------------------------------
struct ReplayCompactArgs {
    const int32_t *order, *p0, *p1, *p2;      // this split's plan (already offset to the split)
    const double* s0;
    const double* gathered;                   // [G][rows]: block q = rank q's decisions, its slot lo_q first
    int32_t *corder, *cp0, *cp1, *cp2;        // compact plan of the accepted foreign slots
    double *cs0, *clogu;
    int32_t* count;                           // this half-step's counter (zero on entry)
    int32_t* count_next;                      // the other counter: zeroed here for the next half-step
    uint8_t* acc;
    int32_t ns, G, rank, rows;
};

static __global__ __launch_bounds__(256) void k_replay_compact(const ReplayCompactArgs A) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) *A.count_next = 0;
    bool hit = false;
    double v = 0.0;
    int i = 0;
    if (t < A.ns) {
        const int q = block_owner(t, A.ns, A.G);                       // the rank whose slot range holds t (shard_range)
        if (q != A.rank) {
            const int lo = (int)((long long)A.ns * q / A.G);
            v = A.gathered[(size_t)q * A.rows + (t - lo)];
            hit = !(v != v);
            i = A.order[t];
            A.acc[i] = hit ? 1 : 0;
        }
    }
    const unsigned long long m = __ballot(hit);
    if (!m) return;
    int base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(A.count, __popcll(m));
    base = __shfl(base, leader);
    if (hit) {
        const int e = base + __popcll(m & ((1ull << lane) - 1ull));
        A.corder[e] = i;
        A.cp0[e] = A.p0[t];
        A.cp1[e] = A.p1[t];
        A.cp2[e] = A.p2[t];
        A.cs0[e] = A.s0[t];
        A.clogu[e] = v;
    }
}

// The stretch move's replay in ONE launch (the configurations that are measured at N > 1 all use it): a wave scans 64 slots of the
// split -- one decision per lane --, sets the accepted flags of the foreign ones and replays the accepted among them 64 / G at a
// time, picking its plan entries straight from the full plan (no compact plan, no device-side count, one launch less on every
// half-step's dependency chain).  Same load_row / make_proposal / store_row as the kernel that took the decisions: same bits.
struct ReplayFusedArgs {
    double* X;
    double* lp;
    uint8_t* acc;
    const int32_t *order, *p0;                // this split's plan (already offset to the split)
    const double* s0;
    const double* gathered;                   // this half-step's receive buffer: [G][rows]
    int32_t ns, G, rank, rows, D;
};

template <int G, int V, int CH>
static __global__ __launch_bounds__(256) void k_replay_stretch(const ReplayFusedArgs A) {
    constexpr int WPW = 64 / G;
    const int lane = threadIdx.x & 63, sub = lane / G, gl = lane % G;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
    const int D = A.D;
    for (int chunk = wave; chunk * 64 < A.ns; chunk += nwaves) {       // wave-uniform
        const int t = chunk * 64 + lane;
        bool hit = false;
        double v = 0.0;
        if (t < A.ns) {
            const int q = block_owner(t, A.ns, A.G);
            if (q != A.rank) {
                const int lo = (int)((long long)A.ns * q / A.G);
                v = A.gathered[(size_t)q * A.rows + (t - lo)];
                hit = !(v != v);
                A.acc[A.order[t]] = hit ? 1 : 0;
            }
        }
        const unsigned long long m = __ballot(hit);
        const int cnt = __popcll(m);
        const int myrank = hit ? __popcll(m & ((1ull << lane) - 1ull)) : -1;
        for (int base = 0; base < cnt; base += WPW) {                  // wave-uniform
            int src = -1;
#pragma unroll
            for (int s = 0; s < WPW; ++s) {
                const unsigned long long b = __ballot(myrank == base + s);
                if (sub == s) src = b ? __ffsll((long long)b) - 1 : -1;
            }
            const bool live = src >= 0;
            const int ts = __shfl(t, live ? src : 0, 64);
            const double vs = __shfl(v, live ? src : 0, 64);
            const int i = A.order[ts], j = A.p0[ts];
            const double z = A.s0[ts];
            Row<G, V, CH> xi, xa, qrow;
            load_row<G, V, CH>(xi, A.X + (size_t)i * D, D, gl);
            load_row<G, V, CH>(xa, A.X + (size_t)j * D, D, gl);
            double factor = 0.0;
            make_proposal<G, V, CH, MOVE_STRETCH>(xi, xa, xa, xa, z, 0.0, D, gl, qrow, factor);
            if (live) {
                store_row<G, V, CH>(qrow, A.X + (size_t)i * D, D, gl);
                if (gl == 0) A.lp[i] = vs;
            }
        }
    }
}

// Replay exchange without a collective library: every rank stores its decisions straight into every peer's receive buffer
// (mapped like the direct exchange's arrays: hipIpc between processes) -- 8 bytes per own walker-update to each peer -- and the
// one-wave barrier kernel (k_peer_barrier: system-scope release, a flag into every peer's array, spin, acquire) tells everybody
// that everybody's decisions have landed.  Two receive buffers alternate by half-step: a peer can only be writing half-step
// h + 2's decisions once this rank has pushed h + 1's, i.e. after it finished reading h's.
struct PushArgs {
    const double* src;                     // this rank's decisions of the half-step: rows doubles
    double* peer[EMX_MAX_PEERS];           // [q]: rank q's receive buffers as mapped here (own entry: the local one)
    long long off;                         // this half-step's buffer (0 or the buffer size) + rank * rows
    int32_t rows, npeer;
};

static __global__ __launch_bounds__(256) void k_push_decisions(const PushArgs A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y;
    if (e < A.rows && q < A.npeer) A.peer[q][A.off + e] = A.src[e];
}

// stored step of the replay exchange: the chain row is the replica after the step (backend.py:225-231)
struct StoreStepArgs {
    const double* X;
    const double* lp;
    const uint8_t* acc;
    uint32_t* acc_count;
    double* chain;
    double* chain_lp;
    long long nx;      // N * D
    int32_t N;
};

static __global__ __launch_bounds__(256) void k_store_step(const StoreStepArgs A) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < A.nx; e += stride) __builtin_nontemporal_store(A.X[e], &A.chain[e]);
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < A.N; w += (long long)gridDim.x * blockDim.x) {
        A.chain_lp[w] = A.lp[w];
        A.acc_count[w] += A.acc[w];
    }
}

struct PeerBarrierArgs {
    unsigned long long* peer_flags[EMX_MAX_PEERS];   // [q]: rank q's flag array (one slot per rank) as mapped here
    unsigned long long* my_flags;                     // this rank's own array: slot q is written by rank q
    int32_t* dead;                                    // set once a wait timed out: the run is invalid, later barriers return at once
    uint32_t* status;
    unsigned long long epoch, timeout_ticks;          // wall_clock64 ticks (100 MHz)
    int32_t rank, npeer;
};

// one wave; lane q talks to rank q
static __global__ __launch_bounds__(64) void k_peer_barrier(const PeerBarrierArgs A) {
    const int q = threadIdx.x;
    // everything this device wrote in earlier kernels (the previous half-step's commits) leaves its L2
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (q < A.npeer && q != A.rank) {
        __hip_atomic_store(&A.peer_flags[q][A.rank], A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        const bool dead = *A.dead != 0;
        // a barrier that timed out earlier: the epochs of the ranks no longer line up, nothing after it is ordered.  Say so on
        // EVERY later barrier (the host refuses further half-steps as well) until the peers are attached again
        if (dead) raise_status(A.status, ST_EXCHANGE_TIMEOUT);
        while (!dead && __hip_atomic_load(&A.my_flags[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < A.epoch) {
            if (wall_clock64() - t0 > A.timeout_ticks) {           // a peer never arrived: never hang the GPU
                raise_status(A.status, ST_EXCHANGE_TIMEOUT);
                *A.dead = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    // stale copies of the peers' rows in this device's caches are dropped before the half-step reads them
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// Replica sync of the pull exchange: every rank's block as [row | log_prob | accepted | accepted count] records
struct BlockArgs {
    double* X;
    double* lp;
    uint8_t* acc;
    uint32_t* acc_count;
    double* rec;          // pack: own block -> rec[0 .. bmax); unpack: rec[G][bmax] -> replica
    int32_t N, D, G, rank, bmax;
};

static __global__ __launch_bounds__(256) void k_block_pack(const BlockArgs A) {
    const int e = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    const int lo = (int)((long long)A.N * A.rank / A.G), hi = (int)((long long)A.N * (A.rank + 1) / A.G);
    if (e >= hi - lo) return;
    const int w = lo + e;
    double* dst = A.rec + (size_t)e * (A.D + 3);
    for (int d = l; d < A.D; d += 16) dst[d] = A.X[(size_t)w * A.D + d];
    if (l == 0) {
        dst[A.D] = A.lp[w];
        dst[A.D + 1] = (double)A.acc[w];
        dst[A.D + 2] = (double)A.acc_count[w];
    }
}

static __global__ __launch_bounds__(256) void k_block_unpack(const BlockArgs A) {
    const int r = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    if (r >= A.G * A.bmax) return;
    const int q = r / A.bmax, e = r - q * A.bmax;
    if (q == A.rank) return;
    const int lo = (int)((long long)A.N * q / A.G), hi = (int)((long long)A.N * (q + 1) / A.G);
    if (e >= hi - lo) return;
    const int w = lo + e;
    const double* src = A.rec + (size_t)r * (A.D + 3);
    for (int d = l; d < A.D; d += 16) A.X[(size_t)w * A.D + d] = src[d];
    if (l == 0) {
        A.lp[w] = src[A.D];
        A.acc[w] = src[A.D + 1] != 0.0 ? 1 : 0;
        A.acc_count[w] = (uint32_t)src[A.D + 2];
    }
}

}  // namespace emx
