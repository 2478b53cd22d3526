// Dense Gaussian targets too wide for the LDS-resident precision matrix (padded ndim > 128): the half-step is
//   k_halfstep (propose only: q, factor -> qout / fout)  ->  k_wide_lp  ->  k_wide_commit
// all on the device, on the context's stream.  The reference evaluates the same thing as one vectorised log_prob_fn
// call on the proposal block (ensemble.py:486-501, compute_log_prob) between get_proposal and the accept loop
// (red_blue.py:90-101).
//
// k_wide_lp is f64-MFMA-bound (D^2 flop against 8 D bytes per row):  Y = R L with R = Q - mu (rows x Dp) and L the lower
// Cholesky factor of the precision matrix, log-prob = -0.5 * rowsum(Y^2) -- the contraction of k_halfstep's dense stage, in
// the same association order (column blocks ascending, k ascending from the diagonal block, squares folded per column
// block), so the two paths agree bit for bit on an ndim both can take.  One wave owns a 16-row tile and walks the
// non-zero 16x16 blocks of L in 128-column macro blocks (8 accumulators); the workgroup's waves share each 16 x 128 slab of L
// through LDS (double-buffered, one barrier per slab = 32 MFMAs per wave); a wave's rows arrive as 128-byte pieces (four
// lanes per row), pass through a 16 x 16 LDS tile of its own and come back as A fragments.  Every load is issued a whole
// slab before its first use, and nothing touches a loaded register in between (a select or a compare on a loaded value
// there cost a vmcnt(0) in the issue phase: 23 % of the kernel, measured).
//
// Three forms of that kernel, one fold order (per 128-column macro block a chain of squares from zero, the chains' totals added
// macro block by macro block), hence the same bits:
//   k_wide_lp_ms   few row tiles per CU: one tile per workgroup, its macro blocks on different waves, no barrier in the loop
//   k_wide_lp<W>   in between: W tiles per workgroup share every slab of L through LDS
//   k_wide_lp_ws   >= 8 tiles per CU: four waves multiply two tiles each, four waves stage for all eight
#include <algorithm>

#include "emx_launch.hpp"

namespace emx {

namespace {

constexpr int SLAB = 32 * 64;        // doubles per slab: 4 k-steps x 8 column blocks x 64 lanes (B-fragment order)
typedef double double4_t __attribute__((ext_vector_type(4)));

// one slab against the first NC column blocks of the macro block
template <int NC, int I0, int I1>
__device__ __forceinline__ void slab_mfma(double4_t (&acc)[8], const double (&afr)[4], const double* bs, int lane) {
    double b[2][NC];          // the B fragments of k-step i + 1 are read while the MFMAs of k-step i are issued
#pragma unroll
    for (int j = 0; j < NC; ++j) b[I0 & 1][j] = bs[(j * 4 + I0) * 64 + lane];
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        if (i + 1 < I1) {
#pragma unroll
            for (int j = 0; j < NC; ++j) b[(i + 1) & 1][j] = bs[(j * 4 + i + 1) * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[i], b[i & 1][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int ART = 18;              // A-tile row stride (doubles): conflict-free fragment reads, 16-byte aligned rows
typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));

__host__ __device__ constexpr size_t wide_lds_bytes(int Dp, int W) {
    return ((size_t)2 * SLAB + (size_t)Dp + (size_t)W * 16 * ART) * sizeof(double);
}

// Staging at the end of the slab costs the matrix pipe about 1 100 of every 5 300 cycles at ndim 512 (MFMA issue is in order
// and blocks its wave; the two waves of a SIMD stage at the same time).  Three ways of hiding it were measured on MI355X and
// dropped (profiles/r02/wide_dense.txt): two four-wave groups half an iteration apart on two barriers per slab (+18 %
// time: a single wave issues f64 MFMAs at 70, not 64, cycles and both barriers are exposed), staging skewed into the MFMA
// stream after k-step 1 / 3 (+-1 %), two co-resident 4-wave workgroups (they are not co-scheduled: +40 %).
template <int W>
__global__ __launch_bounds__(64 * W) void k_wide_lp(const WideLpArgs A) {
    constexpr int NT = 64 * W;
    constexpr int NLD = SLAB / 2 / NT;            // double2 per thread and slab
    extern __shared__ __attribute__((aligned(16))) double dsm[];      // slab[2][SLAB] | mu[Dp] | per wave: A tile [16][ART]
    typedef double4_t d4;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, tx = threadIdx.x;
    double* Bs = dsm;
    const int am = lane & 15, ak = lane >> 4;             // MFMA A-fragment coordinates
    const int arow = lane >> 2, aseg = lane & 3;          // row loads: four lanes cover one 128-byte piece of a row
    const int D = A.D, Dp = A.Dp, DPB = Dp / 16, KK = Dp / 4;
    const int t_hi = A.t_hi_dev ? *A.t_hi_dev : A.t_hi;
    const int ntiles = (t_hi - A.t_lo + 15) / 16;
    const double2* img2 = reinterpret_cast<const double2*>(A.img);
    const int nmacro = (DPB + 7) / 8;
    double* muS = dsm + (size_t)2 * SLAB;
    double* At = muS + Dp + wib * 16 * ART;
    for (int d = tx; d < Dp; d += NT) muS[d] = A.img[(size_t)Dp * Dp + d];      // zero padded beyond D
    __syncthreads();
    // this thread's pieces of a slab: double2 number e = tx + r NT of the 32 chunks (column block j, k-step i) x 32
    int boff[NLD], bj[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int e = tx + r * NT, chunk = e >> 5, within = e & 31;
        bj[r] = chunk >> 2;
        boff[r] = ((chunk >> 2) * KK + (chunk & 3)) * 32 + within;
    }

    for (int base = blockIdx.x * W; base < ntiles; base += gridDim.x * W) {      // workgroup-uniform
        const int tile = base + wib;
        const int t = A.t_lo + tile * 16 + arow;
        const bool rowlive = tile < ntiles && t < t_hi;
        const double* rowp = A.rows + (size_t)(rowlive ? (A.order ? A.order[A.pos0 + t] : t) : 0) * D;
        double part[4] = {0.0, 0.0, 0.0, 0.0};
        bool bad = false;
        d4 acc[8];
        double2 bn[NLD];
        double afr[4], xn[4];
        int kn = 0;

        // slab (nbb, sp): k rows 16 (8 nbb + sp) ..+16, columns 128 nbb ..+128.  Raw loads only: their first use
        // (consume) comes after the current slab's MFMAs.
        auto issue = [&](int nbb, int sp) {
            const int ncb = min(8, DPB - 8 * nbb), kk0 = 4 * (8 * nbb + sp);
            const double2* src = img2 + ((size_t)(8 * nbb) * KK + kk0) * 32;           // wave-uniform
#pragma unroll
            for (int r = 0; r < NLD; ++r) bn[r] = bj[r] < ncb ? src[boff[r]] : double2{0.0, 0.0};
            kn = 4 * kk0 + 4 * aseg;
            if (16 * (8 * nbb + sp) + 16 <= D) {               // wave-uniform: the whole k block lies inside the row
                const double* rp = rowlive ? rowp + kn : A.rows;        // dead rows read row 0 (finite or not, never used)
                const double2_a8 lo = *reinterpret_cast<const double2_a8*>(rp);
                const double2_a8 hi = *reinterpret_cast<const double2_a8*>(rp + 2);
                xn[0] = lo.x;                                  // raw: masked in consume (no use of a loaded value here)
                xn[1] = lo.y;
                xn[2] = hi.x;
                xn[3] = hi.y;
            } else {                                           // the ragged last k block of a row (ndim not a multiple of 16)
#pragma unroll
                for (int e = 0; e < 4; ++e) xn[e] = rowp[min(kn + e, D - 1)];
            }
        };
        // The raw row values go into the wave's A tile (masked: dead rows and the padding columns are zero) and come back
        // as this lane's four A fragments R = Q - mu of the next slab.  The finiteness test is an integer test of the
        // exponent field.
        auto consume = [&]() {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xn[e] = (rowlive && kn + e < D) ? xn[e] : 0.0;
                bad |= (__double2hiint(xn[e]) & 0x7ff00000) == 0x7ff00000;
            }
            double2* dst = reinterpret_cast<double2*>(At + arow * ART + 4 * aseg);
            dst[0] = double2{xn[0], xn[1]};
            dst[1] = double2{xn[2], xn[3]};
            EMX_WAVE_SYNC();
            const int kf = kn - 4 * aseg + ak;                 // first k of this lane's fragments
#pragma unroll
            for (int i = 0; i < 4; ++i) afr[i] = At[am * ART + 4 * i + ak] - muS[kf + 4 * i];
            EMX_WAVE_SYNC();
        };
        auto publish = [&](int buf) {
            double2* dst = reinterpret_cast<double2*>(Bs + (size_t)buf * SLAB);
#pragma unroll
            for (int r = 0; r < NLD; ++r) dst[tx + r * NT] = bn[r];
        };

        // slab order: macro block by macro block, k blocks from the diagonal down
        auto next_of = [&](int& nb_, int& sp_) {
            if (++sp_ == DPB - 8 * nb_) {
                ++nb_;
                sp_ = 0;
            }
        };
        int nbb = 0, sp = 0, buf = 0;
        int nbl = 0, spl = 0;                                  // the slab whose loads are in flight
        issue(0, 0);
        publish(0);
        consume();
        next_of(nbl, spl);
        if (nbl < nmacro) issue(nbl, spl);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
        __syncthreads();
        for (;;) {
            const int ncb = min(8, DPB - 8 * nbb), nsl = DPB - 8 * nbb;
            const bool last = (sp == nsl - 1) && nbb + 1 >= nmacro;
            const int jlim = min(ncb - 1, sp);                 // column block j starts at its diagonal block: k block >= j
            const double* bs = Bs + (size_t)buf * SLAB;
#define EMX_SLAB_STEPS(I0, I1)                                                  \
    switch (jlim) {                                                            \
        case 7: slab_mfma<8, I0, I1>(acc, afr, bs, lane); break;               \
        case 6: slab_mfma<7, I0, I1>(acc, afr, bs, lane); break;               \
        case 5: slab_mfma<6, I0, I1>(acc, afr, bs, lane); break;               \
        case 4: slab_mfma<5, I0, I1>(acc, afr, bs, lane); break;               \
        case 3: slab_mfma<4, I0, I1>(acc, afr, bs, lane); break;               \
        case 2: slab_mfma<3, I0, I1>(acc, afr, bs, lane); break;               \
        case 1: slab_mfma<2, I0, I1>(acc, afr, bs, lane); break;               \
        default: slab_mfma<1, I0, I1>(acc, afr, bs, lane); break;              \
    }
            // the next slab's operands: loaded during the previous slab, to LDS now, and the loads of the slab after issued
            auto stage = [&]() {
                if (last) return;
                __builtin_amdgcn_sched_barrier(0);
                publish(buf ^ 1);
                consume();
                next_of(nbl, spl);
                if (nbl < nmacro) issue(nbl, spl);
                __builtin_amdgcn_sched_barrier(0);
            };
            EMX_SLAB_STEPS(0, 4);
            stage();
#undef EMX_SLAB_STEPS
            if (sp == nsl - 1) {
                // macro block complete: its squares folded column block by column block (one chain from zero), the chain's
                // total added to the row sums -- macro block by macro block, so that the macro blocks of a tile can also be
                // computed by different waves (k_wide_lp_ms) and give the same bits
                double pm[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < ncb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pm[r] = fma(acc[j][r], acc[j][r], pm[r]);
                    }
                    acc[j] = d4{0.0, 0.0, 0.0, 0.0};
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) part[r] += pm[r];
            }
            __syncthreads();                                   // slab s + 1 is published; nobody reads slab s any more
            if (last) break;
            next_of(nbb, sp);
            buf ^= 1;
        }

        // lane (am, ak) ends with the total of tile row ak + 4 (am & 3)
        const double qf = row16_sum4(part[0], part[1], part[2], part[3], lane);
        const int myrow = (lane >> 4) + 4 * (lane & 3);
        const int tt = A.t_lo + tile * 16 + myrow;
        const unsigned long long bm = __ballot(bad);          // lanes 4 r .. 4 r + 3 loaded row r
        const bool rowbad = A.check_bad && ((bm >> (4 * myrow)) & 0xFull);
        if ((lane & 15) < 4 && tile < ntiles && tt < t_hi) {
            const double lpn = rowbad ? -__builtin_inf() : -0.5 * qf;        // a non-finite proposal is rejected (ensemble.py:476-479 raised already)
            if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);             // ensemble.py:550-551
            A.out[A.scatter ? (A.order ? A.order[A.pos0 + tt] : tt) : tt] = lpn;
        }
    }
}

// ---- the same contraction with the work split by ROLE (8 tiles per workgroup: ensembles of >= 8 tiles per CU) ----------------
// Waves 0-3 (one per SIMD) only multiply, two row tiles each: per slab 8 A-fragment reads, 32 B-fragment reads (each feeds
// two MFMAs), 64 MFMAs.  Waves 4-7 (the other wave of each SIMD) only stage: they load the next slab of L and the next
// 16 x 16 piece of all eight row tiles (a slab ahead, in registers), subtract the mean, test finiteness and write both into
// the other LDS buffers.  One workgroup barrier per slab hands the buffers over.  In the single-role kernel above every wave
// carries ~150 staging instructions per slab in front of the same barrier as its 32 MFMAs, and MFMA issue blocks its wave:
// the matrix pipe idled for a fifth of the kernel whatever the order.
constexpr int WS_TILES = 8, WS_CONS = 4, WS_LOAD = 4, WS_NT = 64 * (WS_CONS + WS_LOAD), WS_NLT = 64 * WS_LOAD;

__host__ __device__ constexpr size_t wide_ws_lds_bytes(int Dp) {
    return ((size_t)2 * SLAB + (size_t)2 * WS_TILES * 16 * ART + (size_t)Dp) * sizeof(double) + WS_TILES * 16 * 8;
}

// one slab against the first NC column blocks for TWO row tiles: every B fragment feeds two MFMAs
template <int NC>
__device__ __forceinline__ void slab_mfma2(double4_t (&acc0)[8], double4_t (&acc1)[8], const double (&a0)[4], const double (&a1)[4],
                                           const double* bs, int lane) {
    double b[2][NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) b[0][j] = bs[(j * 4) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) {
#pragma unroll
            for (int j = 0; j < NC; ++j) b[(i + 1) & 1][j] = bs[(j * 4 + i + 1) * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            acc0[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b[i & 1][j], acc0[j], 0, 0, 0);
            acc1[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b[i & 1][j], acc1[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// (Round 5 built the stretch proposal INTO the loader waves -- one launch fewer per half-step -- bit-equal and 16-19 % slower: the
// gathers of partner pieces all fall into the first macro block's pass; profiles/r05/wide_fuse_ab.txt.  Removed in round 6; git
// history has it: `git show 8adb33a:emcee_amd/csrc/emx_wide.hip`.)
__global__ __launch_bounds__(WS_NT) void k_wide_lp_ws(const WideLpArgs A) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];      // slab[2][SLAB] | A tiles [2][8][16][ART] | mu[Dp] | bad[8][16][8]
    typedef double4_t d4;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, tx = threadIdx.x;
    const int D = A.D, Dp = A.Dp, DPB = Dp / 16, KK = Dp / 4;
    const int t_hi = A.t_hi_dev ? *A.t_hi_dev : A.t_hi;
    const int ntiles = (t_hi - A.t_lo + 15) / 16;
    const int nmacro = (DPB + 7) / 8;
    double* Bs = dsm;
    double* As = dsm + (size_t)2 * SLAB;
    double* muS = As + (size_t)2 * WS_TILES * 16 * ART;
    unsigned char* badS = reinterpret_cast<unsigned char*>(muS + Dp);
    for (int d = tx; d < Dp; d += WS_NT) muS[d] = A.img[(size_t)Dp * Dp + d];      // zero padded beyond D
    __syncthreads();
    auto next_of = [&](int& nb_, int& sp_) {                  // slab order: macro block by macro block, k blocks from the diagonal down
        if (++sp_ == DPB - 8 * nb_) {
            ++nb_;
            sp_ = 0;
        }
    };
    int nslabs = 0;
    for (int m = 0; m < nmacro; ++m) nslabs += DPB - 8 * m;

    for (int base = blockIdx.x * WS_TILES; base < ntiles; base += gridDim.x * WS_TILES) {      // workgroup-uniform
        if (wib >= WS_CONS) {
            // ------------------------------------------------ loader waves ------------------------------------------------
            const int lt = tx - 64 * WS_CONS;                  // 0 .. 255
            const double2* img2 = reinterpret_cast<const double2*>(A.img);
            // slab of L: double2 number e = lt + r * 256 of the 32 chunks (column block j, k-step i) x 32
            int boff[4], bj[4];
            // row pieces: (tile w, row, 16-byte piece) number e of 8 x 16 x 8
            const double* rbase[4];
            bool rlive[4];
            int aoff[4], apc[4];
            bool bad[4] = {false, false, false, false};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = lt + r * WS_NLT, chunk = e >> 5, within = e & 31;
                bj[r] = chunk >> 2;
                boff[r] = ((chunk >> 2) * KK + (chunk & 3)) * 32 + within;
                const int w = e >> 7, row = (e >> 3) & 15, pc = e & 7;
                const int tile = base + w, t = A.t_lo + tile * 16 + row;
                rlive[r] = tile < ntiles && t < t_hi;
                rbase[r] = A.rows + (size_t)(rlive[r] ? (A.order ? A.order[A.pos0 + t] : t) : 0) * D;
                aoff[r] = (w * 16 + row) * ART + 2 * pc;
                apc[r] = 2 * pc;
            }
            double2 bn[4], xn[4];
            auto issue = [&](int nbb, int sp) {                // raw loads only; first use one slab later
                const int ncb = min(8, DPB - 8 * nbb), kb = 8 * nbb + sp;
                const double2* src = img2 + ((size_t)(8 * nbb) * KK + 4 * kb) * 32;
#pragma unroll
                for (int r = 0; r < 4; ++r) bn[r] = bj[r] < ncb ? src[boff[r]] : double2{0.0, 0.0};
                const int k0 = 16 * kb;
                if (k0 + 16 <= D) {                            // uniform: the whole k block lies inside the rows
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double2_a8 v = *reinterpret_cast<const double2_a8*>(rbase[r] + k0 + apc[r]);
                        xn[r] = double2{v.x, v.y};
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        xn[r] = double2{rbase[r][min(k0 + apc[r], D - 1)], rbase[r][min(k0 + apc[r] + 1, D - 1)]};
                }
            };
            auto publish = [&](int kb, int buf) {              // R = Q - mu, masked; the finiteness test on the raw values
                double2* bdst = reinterpret_cast<double2*>(Bs + (size_t)buf * SLAB);
                double* adst = As + (size_t)buf * WS_TILES * 16 * ART;
                const int k0 = 16 * kb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bdst[lt + r * WS_NLT] = bn[r];
                    const int k = k0 + apc[r];
                    const double x0 = (rlive[r] && k < D) ? xn[r].x : 0.0, x1 = (rlive[r] && k + 1 < D) ? xn[r].y : 0.0;
                    bad[r] |= (__double2hiint(x0) & 0x7ff00000) == 0x7ff00000 || (__double2hiint(x1) & 0x7ff00000) == 0x7ff00000;
                    *reinterpret_cast<double2*>(adst + aoff[r]) = double2{x0 - muS[k], x1 - muS[k + 1]};
                }
            };
            int nbl = 0, spl = 0;
            issue(0, 0);
            publish(0, 0);
            next_of(nbl, spl);
            if (nbl < nmacro) issue(nbl, spl);
            __syncthreads();                                   // slab 0 is in buffer 0
            for (int s = 0; s < nslabs; ++s) {
                if (s + 1 < nslabs) {
                    publish(8 * nbl + spl, (s + 1) & 1);       // nobody reads that buffer any more
                    next_of(nbl, spl);
                    if (nbl < nmacro) issue(nbl, spl);
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                badS[lt + r * WS_NLT] = bad[r] ? 1 : 0;
            }
            __syncthreads();                                   // the flags are visible to the deciding lanes
            __syncthreads();                                   // ... and read before the next pass overwrites anything
        } else {
            // ----------------------------------------------- consumer waves -----------------------------------------------
            const int am = lane & 15, ak = lane >> 4;
            double part0[4] = {0.0, 0.0, 0.0, 0.0}, part1[4] = {0.0, 0.0, 0.0, 0.0};
            d4 acc0[8], acc1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc0[j] = acc1[j] = d4{0.0, 0.0, 0.0, 0.0};
            int s = 0;
            __syncthreads();                                   // slab 0 is in buffer 0
            // one slab: A fragments of both tiles, the MFMAs against the first NC column blocks, hand the buffers over
#define EMX_WS_SLAB(NC)                                                                              \
    do {                                                                                             \
        const double* bs = Bs + (size_t)(s & 1) * SLAB;                                              \
        const double* at = As + ((size_t)(s & 1) * WS_TILES + 2 * wib) * 16 * ART;                   \
        double a0[4], a1[4];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
            a0[i] = at[am * ART + 4 * i + ak];                                                       \
            a1[i] = at[16 * ART + am * ART + 4 * i + ak];                                            \
        }                                                                                            \
        slab_mfma2<NC>(acc0, acc1, a0, a1, bs, lane);                                                \
        ++s;                                                                                         \
        __syncthreads();                                                                             \
    } while (0)
            for (int m = 0; m < nmacro; ++m) {
                const int ncb = min(8, DPB - 8 * m), nsl = DPB - 8 * m;
                // the slabs across the diagonal blocks: column block j starts at its diagonal block (k block >= j); the column
                // blocks a short last macro block does not have hold zeros (the loaders write them) and are never folded
                if (nsl > 0) EMX_WS_SLAB(1);
                if (nsl > 1) EMX_WS_SLAB(2);
                if (nsl > 2) EMX_WS_SLAB(3);
                if (nsl > 3) EMX_WS_SLAB(4);
                if (nsl > 4) EMX_WS_SLAB(5);
                if (nsl > 5) EMX_WS_SLAB(6);
                if (nsl > 6) EMX_WS_SLAB(7);
                for (int sp = 7; sp < nsl; ++sp) EMX_WS_SLAB(8);
                // macro block complete: the chain of its squares (from zero), then the chain's total into the row sums
                double pm0[4] = {0.0, 0.0, 0.0, 0.0}, pm1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < ncb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pm0[r] = fma(acc0[j][r], acc0[j][r], pm0[r]);
                            pm1[r] = fma(acc1[j][r], acc1[j][r], pm1[r]);
                        }
                    }
                    acc0[j] = acc1[j] = d4{0.0, 0.0, 0.0, 0.0};
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    part0[r] += pm0[r];
                    part1[r] += pm1[r];
                }
            }
#undef EMX_WS_SLAB
            __syncthreads();                                   // the loaders' finiteness flags
            const int myrow = (lane >> 4) + 4 * (lane & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double qf = h ? row16_sum4(part1[0], part1[1], part1[2], part1[3], lane)
                                    : row16_sum4(part0[0], part0[1], part0[2], part0[3], lane);
                const int w = 2 * wib + h, tile = base + w;
                const int tt = A.t_lo + tile * 16 + myrow;
                if ((lane & 15) < 4 && tile < ntiles && tt < t_hi) {
                    const unsigned long long fl = *reinterpret_cast<const unsigned long long*>(badS + (w * 16 + myrow) * 8);
                    const bool rowbad = A.check_bad && fl != 0ull;
                    const double lpn = rowbad ? -__builtin_inf() : -0.5 * qf;    // a non-finite proposal is rejected (ensemble.py:476-479 raised already)
                    if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);         // ensemble.py:550-551
                    A.out[A.scatter ? (A.order ? A.order[A.pos0 + tt] : tt) : tt] = lpn;
                }
            }
            __syncthreads();
        }
    }
}

// ---- the same contraction for ensembles of FEW row tiles: the macro blocks of one tile on different waves ----------------------
// A workgroup owns one 16-row tile; wave w computes macro blocks w, w + 4, ... (128 columns of Y each) on its own: A fragments
// through a wave-private LDS tile as above, B fragments straight from the image in L2 (nothing to share: every wave reads a
// different part of L), double-buffered a slab ahead in registers, no workgroup barrier inside the loop.  Each macro block's
// chain of squares goes to LDS; wave 0 adds them in macro order -- the order of the other two kernels, so the bits are the same
// -- and decides.  With the tiles of a small ensemble on one wave each (k_wide_lp<1>) a 512-dimensional log-prob took 80
// slabs in sequence per tile; here the longest wave takes 32.
constexpr int MS_W = 4;

__host__ __device__ constexpr size_t wide_ms_lds_bytes(int Dp) {
    return ((size_t)Dp + (size_t)MS_W * 16 * ART + (size_t)((Dp / 16 + 7) / 8) * 256) * sizeof(double);
}

// the B fragments of slab (m, sp) for the first NC column blocks: 4 k-steps x NC loads of 8 bytes per lane
template <int NC>
__device__ __forceinline__ void ms_load_b(double (&b)[4][8], const double* img, int KK, int m, int kb, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) b[i][j] = img[((size_t)(8 * m + j) * KK + 4 * kb + i) * 64 + lane];
}

template <int NC>
__device__ __forceinline__ void ms_mfma(double4_t (&acc)[8], const double (&afr)[4], const double (&b)[4][8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[i], b[i][j], acc[j], 0, 0, 0);
}

__global__ __launch_bounds__(64 * MS_W) void k_wide_lp_ms(const WideLpArgs A) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];      // mu[Dp] | per wave: A tile [16][ART] | per macro block: chain totals [64][4]
    typedef double4_t d4;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, tx = threadIdx.x;
    const int am = lane & 15, ak = lane >> 4;
    const int arow = lane >> 2, aseg = lane & 3;
    const int D = A.D, Dp = A.Dp, DPB = Dp / 16, KK = Dp / 4;
    const int t_hi = A.t_hi_dev ? *A.t_hi_dev : A.t_hi;
    const int ntiles = (t_hi - A.t_lo + 15) / 16;
    const int nmacro = (DPB + 7) / 8;
    double* muS = dsm;
    double* At = muS + Dp + wib * 16 * ART;
    double* pmS = muS + Dp + MS_W * 16 * ART;
    for (int d = tx; d < Dp; d += 64 * MS_W) muS[d] = A.img[(size_t)Dp * Dp + d];
    __syncthreads();

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {          // workgroup-uniform
        const int t = A.t_lo + tile * 16 + arow;
        const bool rowlive = t < t_hi;
        const double* rowp = A.rows + (size_t)(rowlive ? (A.order ? A.order[A.pos0 + t] : t) : 0) * D;
        bool bad = false;
        for (int m = wib; m < nmacro; m += MS_W) {
            const int ncb = min(8, DPB - 8 * m), nsl = DPB - 8 * m;
            d4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
            double afr[4], xn[4], bc[4][8], bn[4][8];
            // the A fragments of k block kb: raw loads (issue), then mask / finiteness / tile / R = Q - mu (consume)
            auto issue_a = [&](int kb) {
                const int kn = 16 * kb + 4 * aseg;
                if (16 * kb + 16 <= D) {
                    const double* rp = rowlive ? rowp + kn : A.rows;
                    const double2_a8 lo = *reinterpret_cast<const double2_a8*>(rp);
                    const double2_a8 hi = *reinterpret_cast<const double2_a8*>(rp + 2);
                    xn[0] = lo.x;
                    xn[1] = lo.y;
                    xn[2] = hi.x;
                    xn[3] = hi.y;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xn[e] = rowp[min(kn + e, D - 1)];
                }
            };
            auto consume_a = [&](int kb) {
                const int kn = 16 * kb + 4 * aseg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xn[e] = (rowlive && kn + e < D) ? xn[e] : 0.0;
                    bad |= (__double2hiint(xn[e]) & 0x7ff00000) == 0x7ff00000;
                }
                double2* dst = reinterpret_cast<double2*>(At + arow * ART + 4 * aseg);
                dst[0] = double2{xn[0], xn[1]};
                dst[1] = double2{xn[2], xn[3]};
                EMX_WAVE_SYNC();
#pragma unroll
                for (int i = 0; i < 4; ++i) afr[i] = At[am * ART + 4 * i + ak] - muS[16 * kb + 4 * i + ak];
                EMX_WAVE_SYNC();
            };
            // one slab with NC column blocks; the operands of the next one (NCN column blocks) are loaded meanwhile
#define EMX_MS_SLAB(NC, NCN)                                                   \
    do {                                                                       \
        const int kb_ = 8 * m + sp_;                                           \
        const bool more_ = sp_ + 1 < nsl;                                      \
        if (more_) {                                                           \
            ms_load_b<NCN>(bn, A.img, KK, m, kb_ + 1, lane);                   \
            issue_a(kb_ + 1);                                                  \
        }                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                     \
        ms_mfma<NC>(acc, afr, bc);                                             \
        __builtin_amdgcn_sched_barrier(0);                                     \
        if (more_) {                                                           \
            consume_a(kb_ + 1);                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                   \
                _Pragma("unroll") for (int j_ = 0; j_ < NCN; ++j_) bc[i_][j_] = bn[i_][j_]; \
        }                                                                      \
        ++sp_;                                                                 \
    } while (0)
            int sp_ = 0;
            ms_load_b<1>(bc, A.img, KK, m, 8 * m, lane);
            issue_a(8 * m);
            consume_a(8 * m);
            if (nsl > 0) EMX_MS_SLAB(1, 2);
            if (nsl > 1) EMX_MS_SLAB(2, 3);
            if (nsl > 2) EMX_MS_SLAB(3, 4);
            if (nsl > 3) EMX_MS_SLAB(4, 5);
            if (nsl > 4) EMX_MS_SLAB(5, 6);
            if (nsl > 5) EMX_MS_SLAB(6, 7);
            if (nsl > 6) EMX_MS_SLAB(7, 8);
            while (sp_ < nsl) EMX_MS_SLAB(8, 8);
#undef EMX_MS_SLAB
            double pm[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < ncb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pm[r] = fma(acc[j][r], acc[j][r], pm[r]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) pmS[((size_t)m * 64 + lane) * 4 + r] = pm[r];
        }
        const unsigned long long bm = __ballot(bad);          // lanes 4 r .. 4 r + 3 loaded row r; wave 0 saw every k (macro block 0)
        __syncthreads();
        if (wib == 0) {
            double part[4] = {0.0, 0.0, 0.0, 0.0};
            for (int m = 0; m < nmacro; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[r] += pmS[((size_t)m * 64 + lane) * 4 + r];
            const double qf = row16_sum4(part[0], part[1], part[2], part[3], lane);
            const int myrow = (lane >> 4) + 4 * (lane & 3);
            const int tt = A.t_lo + tile * 16 + myrow;
            const bool rowbad = A.check_bad && ((bm >> (4 * myrow)) & 0xFull);
            if ((lane & 15) < 4 && tt < t_hi) {
                const double lpn = rowbad ? -__builtin_inf() : -0.5 * qf;        // a non-finite proposal is rejected (ensemble.py:476-479 raised already)
                if (lpn != lpn) raise_status(A.status, ST_NAN_LOGP);             // ensemble.py:550-551
                A.out[A.scatter ? (A.order ? A.order[A.pos0 + tt] : tt) : tt] = lpn;
            }
        }
        __syncthreads();                                       // the chain totals are consumed before the next tile overwrites them
    }
}

// decision + commit of one half-step from (qout, fout, newlp): red_blue.py:96-101, move.py:33-34.  GL lanes per slot: a whole
// wave for the wide rows this file is about, 8-32 for the narrow rows of a device-callback target (emx_set_target_callback;
// at ndim 64 a wave per slot was 32 768 waves of four dependent loads each: 8.8 us per launch, round 3).
template <int GL>
__global__ __launch_bounds__(256) void k_wide_commit(const WideCommitArgs A) {
    constexpr int SPW = 64 / GL;                       // slots per wave
    const int lane = threadIdx.x & 63, gl = lane % GL, sub = lane / GL;
    const int t_hi = A.t_hi_dev ? *A.t_hi_dev : A.t_hi;
    const int D = A.D;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    for (int t = A.t_lo + wave * SPW + sub; t < t_hi; t += nwave * SPW) {
        const int pos = A.pos0 + t;
        const int i = A.order[pos];
        const double nlp = A.newlp[t], lp_old = A.lp[i];
        if (A.status && gl == 0 && nlp != nlp) raise_status(A.status, ST_NAN_LOGP);      // ensemble.py:550-551 (a caller's callback may return one)
        const double lnpdiff = A.fout[t] + nlp - lp_old;                     // red_blue.py:99
        const bool accept = lnpdiff > A.logu[pos];                          // red_blue.py:100
        const double* q = A.qout + (size_t)t * D;
        double* xr = A.X + (size_t)i * D;
        double* ch = A.chain ? A.chain + (size_t)i * D : nullptr;
        double* sb = A.sendbuf ? A.sendbuf + (size_t)(t - A.t_lo) * (D + 2) : nullptr;
        if (accept || ch || sb) {               // a rejected proposal of an unstored, unsharded step touches no row at all
            for (int d = gl; d < D; d += GL) {
                const double v = accept ? q[d] : xr[d];
                if (accept) xr[d] = v;
                if (ch) __builtin_nontemporal_store(v, &ch[d]);          // chain rows stream past the Infinity Cache (store_row_stream)
                if (sb) sb[d] = v;
            }
        }
        if (gl == 0) {
            const double lp_fin = accept ? nlp : lp_old;
            if (accept) A.lp[i] = nlp;
            A.acc[i] = accept ? 1 : 0;
            if (A.chain_lp) {
                A.chain_lp[i] = lp_fin;
                if (accept) A.acc_count[i] += 1u;
            }
            if (sb) {
                sb[D] = lp_fin;
                sb[D + 1] = accept ? 1.0 : 0.0;
            }
            if (A.declp) A.declp[t - A.t_lo] = accept ? nlp : __builtin_nan("");
        }
    }
}

}  // namespace

namespace {

template <int W>
hipError_t launch_lp(const WideLpArgs& a, dim3 grid, hipStream_t st) {
    const size_t lds = wide_lds_bytes(a.Dp, W);               // up to 66 KB: above the 64 KB a kernel gets without asking
    static size_t lds_granted[MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)k_wide_lp<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_granted[dev] = lds;
    }
    hipLaunchKernelGGL(k_wide_lp<W>, grid, dim3(64 * W), lds, st, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_wide_lp(const WideLpArgs& a, int nrows_bound, int num_cu, hipStream_t st) {
    if (nrows_bound <= 0) return hipSuccess;
    const int ntiles = (nrows_bound + 15) / 16;
    // wide workgroups share each slab of L among more rows; few tiles spread over more CUs instead
    const int W = ntiles >= 8 * num_cu ? 8 : ntiles >= 4 * num_cu ? 4 : ntiles >= 2 * num_cu ? 2 : 1;
    const int nblocks = (ntiles + W - 1) / W;
    const dim3 grid((unsigned)std::min(nblocks, 4 * num_cu));
    const int nmacro = (a.Dp / 16 + 7) / 8;
    if (!a.single_role && nmacro >= 2 && ntiles < num_cu * (nmacro >= 4 ? 5 : 3)) {
        // few row tiles per CU, several macro blocks: the macro blocks of a tile on different waves (a wave's longest chain is
        // DPB slabs of ~2 500 cycles against nslabs ~ DPB (nmacro + 1) / 2 slabs of ~5 300 in the single-role kernel)
        const size_t lds = wide_ms_lds_bytes(a.Dp);
        static size_t lds_granted[MAX_DEVICES] = {};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
            const hipError_t e = hipFuncSetAttribute((const void*)k_wide_lp_ms, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            lds_granted[dev] = lds;
        }
        hipLaunchKernelGGL(k_wide_lp_ms, dim3((unsigned)std::min(ntiles, 4 * num_cu)), dim3(64 * MS_W), lds, st, a);
        return hipGetLastError();
    }
    if (W == 8 && !a.single_role) {
        const size_t lds = wide_ws_lds_bytes(a.Dp);
        static size_t lds_granted[MAX_DEVICES] = {};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
            const hipError_t e = hipFuncSetAttribute((const void*)k_wide_lp_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            lds_granted[dev] = lds;
        }
        hipLaunchKernelGGL(k_wide_lp_ws, grid, dim3(WS_NT), lds, st, a);
        return hipGetLastError();
    }
    switch (W) {
        case 8: return launch_lp<8>(a, grid, st);
        case 4: return launch_lp<4>(a, grid, st);
        case 2: return launch_lp<2>(a, grid, st);
        default: return launch_lp<1>(a, grid, st);
    }
}

hipError_t launch_wide_commit(const WideCommitArgs& a, int nrows_bound, int num_cu, hipStream_t st) {
    if (nrows_bound <= 0) return hipSuccess;
    const int gl = a.D <= 64 ? 8 : a.D <= 128 ? 16 : a.D <= 256 ? 32 : 64;           // lanes per slot
    const int per_block = 4 * (64 / gl);
    const int nblocks = std::min((nrows_bound + per_block - 1) / per_block, 64 * num_cu);
    if (gl == 8) hipLaunchKernelGGL(k_wide_commit<8>, dim3((unsigned)nblocks), dim3(256), 0, st, a);
    else if (gl == 16) hipLaunchKernelGGL(k_wide_commit<16>, dim3((unsigned)nblocks), dim3(256), 0, st, a);
    else if (gl == 32) hipLaunchKernelGGL(k_wide_commit<32>, dim3((unsigned)nblocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_wide_commit<64>, dim3((unsigned)nblocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace emx
