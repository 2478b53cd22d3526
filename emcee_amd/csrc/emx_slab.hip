// Fused dense-Gaussian half-step for padded ndim 80 ... 128 (even ndim 66 ... 128): the "slab" form.
//
// What it replaces: k_halfstep<16, 2, 4, MOVE, DPB, 1> at DPB = 5 ... 8.  That instantiation keeps a full 16 x (Dp + 2) tile per
// wave in LDS next to the 72 KB image of the Cholesky factor (Dp = 128): 4 x 16.6 KB of tiles is all that fits, so a CU runs ONE
// 4-wave workgroup -- one wave per SIMD -- and that wave runs load -> propose -> 144 dependent MFMAs -> decide -> commit strictly in
// sequence: 0.38 of the HBM roofline at 65 536 x 128 (profiles/r03/dense_128_fused.txt).
//
// Here the 16 proposals of a tile stay in REGISTERS (they are needed again for the commit anyway) and the LDS tile holds one
// 32-column SLAB of R = Q - mu at a time -- exactly the values q.x[c][*] of chunk c in the (G = 16, V = 2, CH = 4) row layout: 16 rows
// x 34 doubles = 4.3 KB per wave instead of 16.6.  Eight waves then fit next to the image (109 KB): two waves per SIMD, twice the
// loads in flight.  The MFMA loop runs slab by slab with the k-steps outermost, so the DPB column blocks are DPB INDEPENDENT
// accumulator chains issued back to back instead of one dependent chain after another.
//
// Same arithmetic, same order of operations per output element as k_halfstep (make_proposal; Y = R L by v_mfma_f64_16x16x4_f64 with
// the k-steps ascending inside each column block; qf = sum of squares in the same order; decision red_blue.py:99-100), hence the same
// bits: tests/test_gpu_wide_dense.py holds it against the wide-target path and the per-tile kernel.
// Scope: the single-replica LEAN case (no send buffers, no device-side slot counts, no timing switches); the stretch and DE moves
// (the snooker move's four rows per walker -- 256 VGPRs of rows a tile -- do not fit two waves per SIMD: it keeps the per-tile kernel).
#include "emx_launch.hpp"

namespace emx {

constexpr int SLAB_RT = 34;          // slab row stride in doubles (32 columns + 2: conflict-free A-fragment reads, as Dp + 2)

size_t slab_lds_bytes(int Dp, int waves) {
    return ((size_t)dense_img_doubles(Dp) + Dp + (size_t)waves * (16 * SLAB_RT + 32)) * sizeof(double);
}

template <int DPB, int MOVE>
static __global__ __launch_bounds__(512) void k_halfstep_slab(const HalfStepArgs A) {
    constexpr int G = 16, V = 2, CH = 4;
    constexpr bool DE = MOVE == MOVE_DE || MOVE == MOVE_SNOOKER;
    constexpr bool SN = MOVE == MOVE_SNOOKER;
    constexpr int WPW = 64 / G, PPT = 16 / WPW;        // 4 walkers a pass, 4 passes a tile
    constexpr int Dp = DPB * 16, KK = Dp / 4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, sub = lane / G, gl = lane % G;
    const int D = A.D;
    double* Sfrag = smem;
    double* muS = smem + dense_img_doubles(Dp);
    double* tile = muS + Dp + (size_t)wib * (16 * SLAB_RT + 32);
    double* facS = tile + 16 * SLAB_RT;
    // Skewed start (A.ablate bit 8, tuning "slab_skew"): with one tile per wave -- 65 536 walkers on 256 CUs -- every wave of the chip
    // loads, then every wave multiplies: the HBM phase (2 KB of rows per update at ndim 128) and the MFMA phase (144 f64 MFMAs per
    // tile) run back to back chip-wide.  The second wave of every SIMD (wib >= 4) therefore issues its first tile's row loads only
    // when its sibling's rows have arrived (a word in the sibling's LDS region), so one wave's MFMA chain covers the other's loads.
    const bool skew_on = (A.ablate & 256) != 0 && blockDim.x == 512;
    {   // The image of the target (emx_set_target), once per workgroup.  Its loads are all in flight before the first is written
        // to LDS: as `for (e ...) dst[e] = img[e]` the compiler made a load - wait - store loop, TEN dependent L2 round trips in
        // front of every half-step's first row load (rounds 4 and 5 until this).  With the skewed start the second waves -- which
        // have nothing else to do until their siblings' rows are back -- stage all of it (19 double2 a lane, registers no row
        // occupies yet); the first waves go straight to their plan entries and rows.
        constexpr int IMG2 = (dense_img_doubles(Dp) + Dp) / 2;
        constexpr int NB = (IMG2 + 255) / 256;
        const double2* img = reinterpret_cast<const double2*>(A.tp1);
        double2* dst = reinterpret_cast<double2*>(smem);
        if (skew_on) {
            if (wib >= 4) {
                const int tx = (int)threadIdx.x - 256;
                double2 stg[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) stg[j] = img[min(tx + j * 256, IMG2 - 1)];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (tx + j * 256 < IMG2) dst[tx + j * 256] = stg[j];
            }
        } else {
            constexpr int NG = 10;                      // any block size, no skew: ten loads in flight per round
            for (int base = threadIdx.x; base < IMG2; base += NG * blockDim.x) {
                double2 stg[NG];
#pragma unroll
                for (int j = 0; j < NG; ++j) stg[j] = img[min(base + j * (int)blockDim.x, IMG2 - 1)];
#pragma unroll
                for (int j = 0; j < NG; ++j)            // straight-line: beyond the end the last element is rewritten with itself
                    dst[min(base + j * (int)blockDim.x, IMG2 - 1)] = stg[j];
            }
        }
    }
    // (mu is read from its LDS image slab by slab -- muS[j] = mu[j], zero beyond ndim, exactly what load_row gives: sixteen VGPRs
    // the DE instantiations need, round 5: 72-104 B of scratch before)
    const int wave = blockIdx.x * (blockDim.x >> 6) + wib, nwaves = gridDim.x * (blockDim.x >> 6);
    const int nslot = A.t_hi - A.t_lo;                 // (t_lo == 0: lean_kind)
    const int ntile = (nslot + 15) / 16;
    const int myrow = (lane >> 4) + 4 * (lane & 3);    // decision lanes: (lane & 15) < 4 decide tile row myrow
    int* sigw = reinterpret_cast<int*>(muS + Dp + (size_t)(wib & 3) * (16 * SLAB_RT + 32) + 16 * SLAB_RT + 16);     // (facS uses 16 of the 32 spare doubles)
    if (skew_on && wib < 4 && lane == 0) *sigw = wave < ntile ? 0 : 1;      // a wave without a tile never holds its sibling
    bool staged = false;
    for (int T = wave; T < ntile; T += nwaves) {
        const int tb = T * 16, pbase = A.pos0 + tb;
        // -------- plan entries, then every row of the tile --------
        int wi[PPT], ja[PPT], jb[DE ? PPT : 1], jc[SN ? PPT : 1];
        double s0v[PPT];
        bool live[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int srow = k * WPW + sub;
            live[k] = tb + srow < nslot;
            const int pos = pbase + (live[k] ? srow : 0);
            wi[k] = A.order[pos];
            ja[k] = A.p0[pos];
            if constexpr (DE) jb[k] = A.p1[pos];
            if constexpr (SN) jc[k] = A.p2[pos];
            s0v[k] = SN ? 0.0 : A.s0[pos];
        }
        const bool mine = (lane & 15) < 4 && tb + myrow < nslot;
        const int mypos = pbase + (tb + myrow < nslot ? myrow : 0);
        if (skew_on && wib >= 4 && !staged) {
            __syncthreads();                            // (the image; the sibling's word is initialised)
            staged = true;
            while (__hip_atomic_load(sigw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(4);
        }
        Row<G, V, CH> xi[PPT], xa[PPT], xb[DE ? PPT : 1], xc[SN ? PPT : 1];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            load_row<G, V, CH>(xi[k], A.X + (size_t)wi[k] * D, D, gl);
            load_row<G, V, CH>(xa[k], A.X + (size_t)ja[k] * D, D, gl);
            if constexpr (DE) load_row<G, V, CH>(xb[k], A.X + (size_t)jb[k] * D, D, gl);
            if constexpr (SN) load_row<G, V, CH>(xc[k], A.X + (size_t)jc[k] * D, D, gl);
        }
        if (!staged) {                                  // the rows are in flight; the barrier only waits for the image
            __syncthreads();
            staged = true;
            if (skew_on && wib < 4) {
                // most of this wave's rows are here: the sibling may load now.  Not "all": its first row would then arrive a whole
                // memory latency after this wave's last -- returns are in order, so requests queued behind the tail keep the pipe full
                // (A.ablate bits 9-10, tuning "slab_skew" 1 ... 4: the sibling starts when all / three quarters / half / a quarter
                // of this wave's row loads have returned)
                constexpr int NL = (DE ? 3 : 2) * PPT * CH;        // row loads of a tile (16 bytes a lane each)
                switch ((A.ablate >> 9) & 3) {
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL / 4) : "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL / 2) : "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NL / 4) : "memory"); break;
                }
                if (lane == 0) __hip_atomic_store(sigw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // -------- proposals: kept in registers (R = Q - mu goes to LDS slab by slab below) --------
        Row<G, V, CH> qk[PPT];
        bool ok[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            double factor = 0.0;
            make_proposal<G, V, CH, MOVE>(xi[k], xa[k], xb[DE ? k : 0], xc[SN ? k : 0], s0v[k], A.gammas, D, gl, qk[k], factor, ja[k]);
            bool bl = false;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int v = 0; v < V; ++v) bl |= !(fabs(qk[k].x[c][v]) <= 1.79769313486231570815e308);
            const bool badq = group_any<G>(bl, sub);      // non-finite proposal -> sticky error (ensemble.py:476-479), rejected
            if (live[k] && badq && gl == 0) raise_status(A.status, ST_BAD_COORD);
            ok[k] = live[k] && !badq;
            if (gl == 0) facS[k * WPW + sub] = badq ? -__builtin_inf() : 0.0;       // (+ my_fac below: exact, x + 0 = x)
            // stored step: the current row goes out now (fire and forget); an accepted proposal overwrites it after the decision
            if (live[k] && A.chain) store_row_stream<G, V, CH>(xi[k], A.chain + (size_t)wi[k] * D, D, gl);
        }
        // the deciding lanes' own entries: asked for here, under the MFMA chain (seven VGPRs the row phase does not have to spare)
        const int my_i = A.order[mypos];
        const double my_logu = A.logu[mypos];
        const double my_fac = A.fac[mypos];             // (stretch: (D - 1) ln z, DE: 0 -- neither move changes it: read by the deciding lane alone)
        const double my_lpo = A.lp[my_i];
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
                    *reinterpret_cast<double2*>(tile + (k * WPW + sub) * SLAB_RT + gl * 2) = r;
                }
                EMX_WAVE_SYNC();                        // this wave's slab is visible to all of its lanes
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int kk = 8 * c + k8;
                    if (kk < KK) {
                        const double a = tile[am * SLAB_RT + 4 * k8 + ak];     // A[i = lane & 15][k = lane >> 4]
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
            A.acc[my_i] = acc ? 1 : 0;
            if (acc) A.lp[my_i] = lpn;
            if (A.chain_lp) {
                A.chain_lp[my_i] = acc ? lpn : my_lpo;
                if (acc) A.acc_count[my_i] += 1u;
            }
        }
        const unsigned long long am64 = __ballot(acc);           // bit (row & 3) * 16 + (row >> 2) <-> tile row
#pragma unroll
        for (int pp = 0; pp < PPT; ++pp) {
            const int row = pp * WPW + sub;
            const bool ac = (am64 >> ((row & 3) * 16 + (row >> 2))) & 1ull;
            if (ac) {
                store_row<G, V, CH>(qk[pp], A.X + (size_t)wi[pp] * D, D, gl);
                if (A.chain) store_row_stream<G, V, CH>(qk[pp], A.chain + (size_t)wi[pp] * D, D, gl);
            }
        }
        EMX_WAVE_SYNC();                                // (facS and the slab are rewritten by the next tile)
    }
    if (!staged) __syncthreads();                       // idle wave: meet the workgroup's barrier
}

template <int DPB, int MOVE>
static hipError_t launch_slab(dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a) {
    auto kern = k_halfstep_slab<DPB, MOVE>;
    static size_t lds_granted[MAX_DEVICES] = {};
    int dev = 0;
    if (lds > 48 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES && lds > lds_granted[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_granted[dev] = lds;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return hipGetLastError();
}

hipError_t launch_slab_dense(int dpb, int move, dim3 grid, dim3 block, size_t lds, hipStream_t st, const HalfStepArgs& a) {
#define EMX_CASE(b)                                                                                    \
    if (dpb == b)                                                                                      \
        return move == MOVE_DE ? launch_slab<b, MOVE_DE>(grid, block, lds, st, a) : launch_slab<b, MOVE_STRETCH>(grid, block, lds, st, a);
    EMX_CASE(5) EMX_CASE(6) EMX_CASE(7) EMX_CASE(8)
#undef EMX_CASE
    return hipErrorInvalidValue;
}

}  // namespace emx
