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

#include "emx_rng.hpp"

namespace emx {

enum : int { MOVE_STRETCH = 0, MOVE_DE = 1, MOVE_SNOOKER = 2, MOVE_EVAL = 3 };
enum : int { TGT_NONE = 0, TGT_ISO = 1, TGT_DIAG = 2, TGT_DENSE = 3, TGT_ROSEN = 4, TGT_BOX = 5 };
enum : uint32_t { ST_NAN_LOGP = 1u, ST_BAD_COORD = 2u };

struct NativeArgs {
    uint64_t seed;
    uint64_t step;
    PermKey pk;
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
    // target
    const double* tp0;    // mu  (diag, dense)
    const double* tp1;    // ivar (diag) | icov (dense, (D, D) row-major)
    double tscale;        // rosenbrock temperature (20), box: unused
    // move parameters
    double a;             // stretch scale
    double sigma, g0;     // DE
    double gammas;        // snooker
    NativeArgs nat;
    int32_t N, D, S, split;
    int32_t pos0, ns;     // first plan position / number of slots of this split
    int32_t t_lo, t_hi;   // slots updated by this rank
    int32_t spw;          // slots per wave
    int32_t native;       // 1: derive the plan from Philox in flight
    int32_t target;
    int32_t Dp;           // dense: D rounded up to 16
};

// ----------------------------------------------------------------------------------------
// group helpers
// ----------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
    for (int m = G >> 1; m > 0; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

__device__ __forceinline__ int set_size(int N, int S, int j) { return (N - j + S - 1) / S; }

// complement position r (sets != split, concatenated in set order) -> (set j, member tt)
__device__ __forceinline__ void comp_locate(int N, int S, int split, int64_t r, int& j, int& tt) {
    j = 0;
    for (int s = 0; s < S; ++s) {
        if (s == split) continue;
        const int n = set_size(N, S, s);
        if (r < n) {
            j = s;
            tt = (int)r;
            return;
        }
        r -= n;
    }
    j = (split == S - 1) ? S - 2 : S - 1;  // unreachable for r < Nc
    tt = 0;
}

struct Slot {
    int i;            // walker to update
    int p0, p1, p2;   // partner walkers
    double s0;        // zz (stretch) | gamma (DE)
    double logu;      // log of the accept uniform
    double lp_old;
    double factor;    // (D-1) ln zz for stretch, else filled later
};

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
    const int ns_own = (N - split + S - 1) / S;
    const int64_t Nc = (int64_t)N - ns_own;
    if (MOVE == MOVE_STRETCH) {
        const double u = u53(A.v[0], A.v[1]);
        const double tt_ = (a - 1.0) * u + 1.0;
        s0 = tt_ * tt_ / a;
        const int64_t r = (int64_t)bounded64(B.v[0], B.v[1], (uint64_t)Nc);
        int j = 0, tt = 0;
        // inline comp_locate (host+device)
        int64_t rr = r;
        for (int s = 0; s < S; ++s) {
            if (s == split) continue;
            const int n = (N - s + S - 1) / S;
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
                const int n = (N - s + S - 1) / S;
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
            const int n = (N - s + S - 1) / S;
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

// ----------------------------------------------------------------------------------------
// element-wise targets evaluated from the register-resident proposal
// ----------------------------------------------------------------------------------------
template <int G, int V, int CH>
__device__ __forceinline__ double eval_valu_target(const Row<G, V, CH>& q, const Row<G, V, CH>& mu,
                                                   const Row<G, V, CH>& iv, int target, double tscale, int D, int gl,
                                                   int lane) {
    double acc = 0.0;
    if (target == TGT_ISO) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) acc = fma(q.x[c][v], q.x[c][v], acc);
        return -0.5 * group_sum<G>(acc);
    }
    if (target == TGT_DIAG) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double d = q.x[c][v] - mu.x[c][v];
                acc = fma(iv.x[c][v] * d, d, acc);
            }
        return -0.5 * group_sum<G>(acc);
    }
    if (target == TGT_ROSEN) {
        // sum_{d < D-1} 100 (x_{d+1} - x_d^2)^2 + (1 - x_d)^2   
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            // first element of the next lane / next chunk
            const double nxt_lane = __shfl(q.x[c][0], lane + 1, 64);
            double nxt_chunk = 0.0;
            if (c + 1 < CH) nxt_chunk = __shfl(q.x[c + 1 < CH ? c + 1 : c][0], lane - gl, 64);
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
//   G lanes per walker row, V doubles per lane per chunk, CH chunks, MOVE, DENSE target.
// Dynamic LDS (DENSE only): Sinv[Dp][Dp+16] | mu[Dp] | per wave: tile[16][Dp+2], qf[16], fac[16]
// ----------------------------------------------------------------------------------------
#define EMX_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

template <int G, int V, int CH, int MOVE, bool DENSE>
__global__ __launch_bounds__(256) void k_halfstep(const HalfStepArgs A) {
    static_assert(G >= 4 && G <= 64 && (64 % G) == 0, "G lanes per walker");
    constexpr int WPW = 64 / G;       // walkers per pass (<= 16)
    constexpr int PPT = 16 / WPW;     // passes per 16-row dense tile
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int sub = lane / G;
    const int gl = lane % G;
    const int D = A.D;
    const int Dp = A.Dp;
    const int RS = Dp + (((Dp >> 4) & 1) ? 0 : 16);   // Sinv row stride == 16 (mod 32): conflict-free B reads
    const int RT = Dp + 2;    // tile row stride: conflict-free A-fragment reads

    double* Sinv = smem;
    double* muS = smem + (size_t)Dp * RS;
    double* tile = muS + Dp + (size_t)wib * (16 * RT + 32);
    double* qfS = tile + 16 * RT;
    double* facS = qfS + 16;

    if constexpr (DENSE) {
        // stage the precision matrix (zero padded) and the mean once per workgroup
        for (int e = threadIdx.x; e < Dp * Dp; e += blockDim.x) {
            const int r = e / Dp, c = e - r * Dp;
            Sinv[r * RS + c] = (r < D && c < D) ? A.tp1[(size_t)r * D + c] : 0.0;
        }
        for (int e = threadIdx.x; e < Dp; e += blockDim.x) muS[e] = e < D ? A.tp0[e] : 0.0;
        __syncthreads();
    }

    // per-lane target parameters (diag Gaussian): same columns for every walker of the wave
    Row<G, V, CH> mu, iv;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) mu.x[c][v] = iv.x[c][v] = 0.0;
    if (!DENSE && A.target == TGT_DIAG) {
        load_row<G, V, CH>(mu, A.tp0, D, gl);
        load_row<G, V, CH>(iv, A.tp1, D, gl);
    }

    const int wave = blockIdx.x * (blockDim.x >> 6) + wib;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int spw = A.spw;
    for (int t0 = A.t_lo + wave * spw; t0 < A.t_hi; t0 += nwaves * spw) {   // wave-uniform batch loop
    const int nslot = min(spw, A.t_hi - t0);

    // ---------------- phase A: lane l <-> slot t0 + l ----------------
    Slot sl;
    {
        const bool valid = lane < nslot;
        const int t = t0 + (valid ? lane : 0);
        double uacc;
        if (A.native) {
            native_slot<MOVE>(A.nat, A.N, A.S, A.split, t, A.a, A.sigma, A.g0, sl.i, sl.p0, sl.p1, sl.p2, sl.s0, uacc);
        } else {
            const int pos = A.pos0 + t;
            sl.i = A.order[pos];
            sl.p0 = (MOVE == MOVE_EVAL) ? sl.i : A.p0[pos];
            sl.p1 = (MOVE == MOVE_DE || MOVE == MOVE_SNOOKER) ? A.p1[pos] : sl.i;
            sl.p2 = (MOVE == MOVE_SNOOKER) ? A.p2[pos] : sl.i;
            sl.s0 = (MOVE == MOVE_STRETCH || MOVE == MOVE_DE) ? A.s0[pos] : 1.0;
            uacc = (MOVE == MOVE_EVAL) ? 0.5 : A.uacc[pos];
        }
        sl.logu = log(uacc);
        sl.lp_old = (MOVE == MOVE_EVAL) ? 0.0 : A.lp[sl.i];
        sl.factor = (MOVE == MOVE_STRETCH) ? ((double)D - 1.0) * log(sl.s0) : 0.0;   // stretch.py:31
    }

    const int npass = (nslot + WPW - 1) / WPW;

    for (int p = 0; p < npass; ++p) {
        const int srow = p * WPW + sub;             // slot (= phase-A lane) handled by this group
        const bool live = srow < nslot;
        const int src = live ? srow : 0;
        const int i = __shfl(sl.i, src, 64);
        const int j0 = __shfl(sl.p0, src, 64);
        const double s0 = __shfl(sl.s0, src, 64);
        double factor = __shfl(sl.factor, src, 64);
        const double lp_old = __shfl(sl.lp_old, src, 64);
        const double logu = __shfl(sl.logu, src, 64);

        Row<G, V, CH> xi, q;
        load_row<G, V, CH>(xi, A.X + (size_t)i * D, D, gl);

        if constexpr (MOVE == MOVE_STRETCH) {
            Row<G, V, CH> xj;
            load_row<G, V, CH>(xj, A.X + (size_t)j0 * D, D, gl);
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const double diff = xj.x[c][v] - xi.x[c][v];   // c[rint] - s
                    const double prod = diff * s0;                  // ... * zz
                    q.x[c][v] = xj.x[c][v] - prod;                  // c[rint] - (...)   (stretch.py:33)
                }
        } else if constexpr (MOVE == MOVE_DE) {
            const int j1 = __shfl(sl.p1, src, 64);
            Row<G, V, CH> x1, x2;
            load_row<G, V, CH>(x1, A.X + (size_t)j0 * D, D, gl);   // pair[0]
            load_row<G, V, CH>(x2, A.X + (size_t)j1 * D, D, gl);   // pair[1]
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const double diff = x2.x[c][v] - x1.x[c][v];   // np.diff(c[pairs], axis=1)  (de.py:53)
                    const double prod = s0 * diff;                  // gamma * diffs
                    q.x[c][v] = xi.x[c][v] + prod;                  // s + ...               (de.py:62)
                }
        } else if constexpr (MOVE == MOVE_SNOOKER) {
            const int j1 = __shfl(sl.p1, src, 64);
            const int j2 = __shfl(sl.p2, src, 64);
            Row<G, V, CH> z, z1, z2;
            load_row<G, V, CH>(z, A.X + (size_t)j0 * D, D, gl);
            load_row<G, V, CH>(z1, A.X + (size_t)j1 * D, D, gl);
            load_row<G, V, CH>(z2, A.X + (size_t)j2 * D, D, gl);
            double n2 = 0.0;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const double dl = xi.x[c][v] - z.x[c][v];       // delta = s[i] - z   (de_snooker.py:41)
                    n2 = fma(dl, dl, n2);
                }
            const double norm = sqrt(group_sum<G>(n2));
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const double u = (xi.x[c][v] - z.x[c][v]) / norm;
                    d1 = fma(u, z1.x[c][v], d1);
                    d2 = fma(u, z2.x[c][v], d2);
                }
            d1 = group_sum<G>(d1);
            d2 = group_sum<G>(d2);
            const double dd = d1 - d2;
            double m2 = 0.0;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const double u = (xi.x[c][v] - z.x[c][v]) / norm;
                    const double ug = u * A.gammas;                 // u * gammas
                    const double prod = ug * dd;                    // * (dot(u,z1) - dot(u,z2))
                    const int d = (c * G + gl) * V + v;
                    q.x[c][v] = d < D ? xi.x[c][v] + prod : 0.0;    // (de_snooker.py:44)
                    const double e = q.x[c][v] - z.x[c][v];
                    m2 = fma(e, e, m2);
                }
            const double nq = sqrt(group_sum<G>(m2));
            factor = ((double)D - 1.0) * (log(nq) - log(norm));     // (de_snooker.py:45-46)
        } else {  // MOVE_EVAL
            q = xi;
        }

        // non-finite proposal -> sticky error (ensemble.py:476-479); the proposal is rejected
        bool badq = false;
        if constexpr (MOVE != MOVE_EVAL) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) badq |= !(fabs(q.x[c][v]) <= 1.79769313486231570815e308);
            badq = group_sum<G>(badq ? 1.0 : 0.0) > 0.0;
            if (live && badq && gl == 0) atomicOr(A.status, ST_BAD_COORD);
        }

        if (A.target == TGT_NONE) {
            // split-phase: hand the proposal to the host log-prob (red_blue.py:90-93)
            if (live) {
                const int t = t0 + srow;
                store_row<G, V, CH>(q, A.qout + (size_t)t * D, D, gl);
                if (gl == 0) A.fout[t] = factor;
            }
            continue;
        }

        if constexpr (!DENSE) {
            const double lp_new = eval_valu_target<G, V, CH>(q, mu, iv, A.target, A.tscale, D, gl, lane);
            if (live && gl == 0 && (lp_new != lp_new)) atomicOr(A.status, ST_NAN_LOGP);   // ensemble.py:550-551
            if constexpr (MOVE == MOVE_EVAL) {
                if (live && gl == 0) A.lp[i] = lp_new;
            } else {
                const double lnpdiff = factor + lp_new - lp_old;                  // red_blue.py:99
                const bool accept = live && !badq && (lnpdiff > logu);            // red_blue.py:100
                if (accept) {
                    store_row<G, V, CH>(q, A.X + (size_t)i * D, D, gl);          // move.py:33
                    if (gl == 0) A.lp[i] = lp_new;                                // move.py:34
                }
                if (live && gl == 0) {
                    A.acc[i] = accept ? 1 : 0;
                    if (A.chain_lp) {
                        A.chain_lp[i] = accept ? lp_new : lp_old;
                        if (accept) A.acc_count[i] += 1u;
                    }
                }
                if (live && A.chain) store_row<G, V, CH>(accept ? q : xi, A.chain + (size_t)i * D, D, gl);
                if (live && A.sendbuf) {
                    double* sb = A.sendbuf + (size_t)(t0 + srow - A.t_lo) * (D + 2);
                    store_row<G, V, CH>(accept ? q : xi, sb, D, gl);
                    if (gl == 0) {
                        sb[D] = accept ? lp_new : lp_old;
                        sb[D + 1] = accept ? 1.0 : 0.0;
                    }
                }
            }
        } else {
            // ---- stage q into the wave's LDS tile; every PPT passes (16 rows) contract with Sinv ----
            const int trow = srow & 15;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int d = (c * G + gl) * V + v;
                    if (d < Dp) tile[trow * RT + d] = (live && !badq) ? q.x[c][v] : muS[d];   // dead rows: zero residual
                }
            if (gl == 0) facS[trow] = badq ? -__builtin_inf() : factor;
            const bool tile_done = ((p + 1) % PPT == 0) || (p + 1 == npass);
            if (!tile_done) continue;
            EMX_WAVE_SYNC();
            {
                // Y = (Q - mu) Sinv  (16 x Dp) by v_mfma_f64_16x16x4_f64; qf[w] = sum_n Y[w][n] (Q - mu)[w][n]
                const int am = lane & 15, ak = lane >> 4;
                double part[4] = {0.0, 0.0, 0.0, 0.0};
                typedef double d4 __attribute__((ext_vector_type(4)));
                for (int nb = 0; nb < Dp / 16; ++nb) {
                    d4 accv = {0.0, 0.0, 0.0, 0.0};
                    for (int kk = 0; kk < Dp / 4; ++kk) {
                        const int k = 4 * kk + ak;
                        const double av = tile[am * RT + k] - muS[k];       // A[i = lane & 15][k = lane >> 4]
                        const double bv = Sinv[k * RS + 16 * nb + am];      // B[k = lane >> 4][j = lane & 15]
                        accv = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, accv, 0, 0, 0);
                    }
                    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
                    const int n = 16 * nb + am;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int w = ak + 4 * r;
                        const double dv = tile[w * RT + n] - muS[n];
                        part[r] = fma(accv[r], dv, part[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int m = 8; m > 0; m >>= 1) part[r] += __shfl_xor(part[r], m, 64);
                    if (am == 0) qfS[ak + 4 * r] = part[r];
                }
            }
            EMX_WAVE_SYNC();
            // ---- decisions for the (up to) 16 slots of this tile, lane-parallel: lane l <-> slot l ----
            const int tb = (p / PPT) * 16;                     // first slot of the tile
            const bool mine = lane >= tb && lane < tb + 16 && lane < nslot;
            bool acc = false;
            double lp_fin = sl.lp_old;
            if (mine) {
                const double lpn = -0.5 * qfS[lane - tb];
                if (lpn != lpn) atomicOr(A.status, ST_NAN_LOGP);
                if constexpr (MOVE == MOVE_EVAL) {
                    A.lp[sl.i] = lpn;
                } else {
                    const double lnpdiff = facS[lane - tb] + lpn - sl.lp_old;
                    acc = lnpdiff > sl.logu;
                    A.acc[sl.i] = acc ? 1 : 0;
                    if (acc) A.lp[sl.i] = lpn;
                    if (acc) lp_fin = lpn;
                    if (A.chain_lp) {
                        A.chain_lp[sl.i] = acc ? lpn : sl.lp_old;
                        if (acc) A.acc_count[sl.i] += 1u;
                    }
                }
            }
            if constexpr (MOVE != MOVE_EVAL) {
                const unsigned long long am64 = __ballot(acc);
                // commit the tile's rows in the (G, V, CH) row layout: accepted rows come from LDS
                for (int pp = 0; pp < PPT; ++pp) {
                    const int row = pp * WPW + sub;              // 0..15
                    const int sidx = tb + row;
                    const bool lv = sidx < nslot;
                    const int wi = __shfl(sl.i, lv ? sidx : 0, 64);
                    const double lpf = __shfl(lp_fin, lv ? sidx : 0, 64);
                    const bool ac = lv && ((am64 >> sidx) & 1ull);
                    if (!lv) continue;
                    Row<G, V, CH> rr;
                    if (ac) {
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int v = 0; v < V; ++v) {
                                const int d = (c * G + gl) * V + v;
                                rr.x[c][v] = d < D ? tile[row * RT + d] : 0.0;
                            }
                        store_row<G, V, CH>(rr, A.X + (size_t)wi * D, D, gl);
                    }
                    if (A.chain || A.sendbuf) {
                        if (!ac) load_row<G, V, CH>(rr, A.X + (size_t)wi * D, D, gl);
                        if (A.chain) store_row<G, V, CH>(rr, A.chain + (size_t)wi * D, D, gl);
                        if (A.sendbuf) {
                            double* sb = A.sendbuf + (size_t)(t0 + sidx - A.t_lo) * (D + 2);
                            store_row<G, V, CH>(rr, sb, D, gl);
                            if (gl == 0) {
                                sb[D] = lpf;
                                sb[D + 1] = ac ? 1.0 : 0.0;
                            }
                        }
                    }
                }
            }
            EMX_WAVE_SYNC();
        }
    }
    }   // batch loop
}

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
    NativeArgs nat;
    int32_t N, D, S, split, pos0, ns, native, move;
};

__global__ __launch_bounds__(256) void k_accept(const AcceptArgs A) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= A.ns) return;
    int i;
    double uacc;
    if (A.native) {
        int a0, a1, a2;
        double s0;
        // only i and uacc are needed; they do not depend on the move
        native_slot<MOVE_EVAL>(A.nat, A.N, A.S, A.split, t, 2.0, 0.0, 0.0, i, a0, a1, a2, s0, uacc);
    } else {
        i = A.order[A.pos0 + t];
        uacc = A.uacc[A.pos0 + t];
    }
    const double nlp = A.new_lp[t];
    const double lp_old = A.lp[i];
    if (nlp != nlp) atomicOr(A.status, ST_NAN_LOGP);
    const double lnpdiff = A.fout[t] + nlp - lp_old;
    const bool accept = lnpdiff > log(uacc);
    const double* q = A.qout + (size_t)t * A.D;
    double* xr = A.X + (size_t)i * A.D;
    if (accept)
        for (int d = lane; d < A.D; d += 64) xr[d] = q[d];
    if (A.chain)
        for (int d = lane; d < A.D; d += 64) A.chain[(size_t)i * A.D + d] = accept ? q[d] : xr[d];
    if (lane == 0) {
        if (accept) A.lp[i] = nlp;
        A.acc[i] = accept ? 1 : 0;
        if (A.chain_lp) {
            A.chain_lp[i] = accept ? nlp : lp_old;
            if (accept) A.acc_count[i] += 1u;
        }
    }
}

// Dump the native-mode plan of one step (parity tests replay it through the oracle).
template <int MOVE>
__global__ void k_native_plan(NativeArgs nat, int N, int S, double a, double sigma, double g0, int32_t* order,
                              int32_t* p0, int32_t* p1, int32_t* p2, double* s0, double* uacc) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= N) return;
    int split = 0, t = pos;
    for (int s = 0; s < S; ++s) {
        const int n = (N - s + S - 1) / S;
        if (t < n) { split = s; break; }
        t -= n;
    }
    int i, a0, a1, a2;
    double z, u;
    native_slot<MOVE>(nat, N, S, split, t, a, sigma, g0, i, a0, a1, a2, z, u);
    order[pos] = i;
    p0[pos] = a0;
    p1[pos] = a1;
    p2[pos] = a2;
    s0[pos] = z;
    uacc[pos] = u;
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
    NativeArgs nat;
    int32_t N, D, S, split, pos0, t_lo, t_hi, native;
};

__global__ __launch_bounds__(256) void k_scatter_rows(const ScatterArgs A) {
    const int lane = threadIdx.x & 63;
    const int t = A.t_lo + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= A.t_hi) return;
    int i;
    if (A.native)
        i = (int)perm_inv((uint32_t)(t * A.S + A.split), A.nat.pk);
    else
        i = A.order[A.pos0 + t];
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

}  // namespace emx
