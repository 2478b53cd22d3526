// Kernels of the device-side exact-plan producer (see emx_mtdev.hpp for the design).  gfx950, wave64.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "emx_mtjump.hpp"      // MT_N

namespace emx {
namespace mtdev {

constexpr int WIN_BLOCKS = 33;                   // 33 * 624 = 20 592 >= 19 937 + 624 words: the window a jump reads
constexpr int JUMP_SPLIT = 8;                    // workgroups sharing one jump polynomial (78 polynomial words each)
constexpr int SEG_BLOCKS = 1024;                 // blocks per segment: the stride of the jump polynomials
constexpr unsigned long long SEG_WORDS = (unsigned long long)SEG_BLOCKS * MT_N;
constexpr int PMAX = 32;                         // segments per round at most
constexpr int TOK_T = 1024;                      // threads of the tokenizer workgroup
constexpr int TOK_WPT = 48;                      // words per thread of a full window (49 152 words)
constexpr int FIN_T = 1024;
constexpr unsigned ST_MT_UNDERRUN = 4u;          // status bit 2 (shared with the pull exchange's overflow): the stream ran out -- the run is void

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
// the recurrence's contribution of (x[k], x[k + 1]):  y = (x[k] & UPPER) | (x[k + 1] & LOWER);  (y >> 1) ^ (y odd ? MATRIX : 0)
__device__ __forceinline__ uint32_t mt_mix(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);
}

// one twist of a block held in LDS by 256 threads: three dependent phases of <= 227 words (word kk needs the NEW word kk - 227)
__device__ __forceinline__ void twist_lds(const uint32_t* o, uint32_t* n, int tid) {
    if (tid < 227) n[tid] = o[tid + 397] ^ mt_mix(o[tid], o[tid + 1]);
    __syncthreads();
    if (tid < 227) {
        const int kk = tid + 227;
        n[kk] = n[kk - 227] ^ mt_mix(o[kk], o[kk + 1]);
    }
    __syncthreads();
    if (tid < 169) {
        const int kk = tid + 454;                                    // 454 .. 622
        n[kk] = n[kk - 227] ^ mt_mix(o[kk], o[kk + 1]);
    } else if (tid == 169) {
        n[623] = n[396] ^ mt_mix(o[623], n[0]);
    }
    __syncthreads();
}

// ---- the window: WIN_BLOCKS blocks after the base state, untempered; optionally the base block itself into the stream ----------
static __global__ __launch_bounds__(256) void k_mt_window(const uint32_t* __restrict__ base_key, uint32_t* __restrict__ xwin,
                                                          uint32_t* __restrict__ stream, unsigned long long smask,
                                                          unsigned long long base_word, int write_base_block) {
    __shared__ uint32_t key[2][MT_N];
    const int tid = threadIdx.x;
    for (int i = tid; i < MT_N; i += 256) {
        const uint32_t k = base_key[i];
        key[0][i] = k;
        if (write_base_block) stream[(base_word + (unsigned long long)i) & smask] = mt_temper(k);
    }
    __syncthreads();
    int cur = 0;
    for (int b = 0; b < WIN_BLOCKS; ++b) {
        twist_lds(key[cur], key[cur ^ 1], tid);
        cur ^= 1;
        for (int i = tid; i < MT_N; i += 256) xwin[b * MT_N + i] = key[cur][i];
    }
}

// ---- the jumps: partial[k][q][j] = XOR over the set bits i of polynomial k inside word range q of window[i + j] ------------------
static __global__ __launch_bounds__(640) void k_mt_jump(const uint32_t* __restrict__ polys, const uint32_t* __restrict__ xwin,
                                                        uint32_t* __restrict__ partial) {
    constexpr int WPS = MT_N / JUMP_SPLIT;                // polynomial words per workgroup (78 -> 2 496 coefficients)
    __shared__ uint32_t Xs[WPS * 32 + MT_N];
    const int q = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const int i0 = q * WPS * 32;
    for (int e = tid; e < WPS * 32 + MT_N; e += 640) Xs[e] = xwin[i0 + e];         // (the last index read is 33 * 624 - 1)
    __syncthreads();
    if (tid < MT_N) {
        uint32_t acc = 0;
        const uint32_t* g = polys + (size_t)k * MT_N + q * WPS;
        for (int w = 0; w < WPS; ++w) {
            uint32_t gw = __builtin_amdgcn_readfirstlane(g[w]);                    // uniform: the loop below is scalar control flow
            const uint32_t* xs = Xs + w * 32 + tid;
            while (gw) {
                const int b = __builtin_ctz(gw);
                gw &= gw - 1u;
                acc ^= xs[b];
            }
        }
        partial[((size_t)k * JUMP_SPLIT + q) * MT_N + tid] = acc;
    }
}

// ---- the stream: workgroup k twists + tempers segment k into the ring -----------------------------------------------------------
struct GenArgs {
    const uint32_t* xwin;
    const uint32_t* partial;
    uint32_t* stream;
    unsigned long long smask;          // ring capacity in words - 1 (a power of two)
    unsigned long long first_word;     // absolute word index of segment 0's first block
    uint32_t* next_base;               // [624]: key of the round's last block (the next round's base state)
    int32_t last_seg;
    int32_t blocks_per_seg;
};
static __global__ __launch_bounds__(256) void k_mt_gen(const GenArgs A) {
    __shared__ uint32_t key[2][MT_N];
    const int tid = threadIdx.x, k = blockIdx.x;
    for (int i = tid; i < MT_N; i += 256) {
        uint32_t v;
        if (k == 0) {
            v = A.xwin[i];                                                          // the block right after the base state
        } else {
            v = 0;
            for (int q = 0; q < JUMP_SPLIT; ++q) v ^= A.partial[((size_t)(k - 1) * JUMP_SPLIT + q) * MT_N + i];
        }
        key[0][i] = v;
    }
    __syncthreads();
    int cur = 0;
    const unsigned long long w0 = A.first_word + (unsigned long long)k * (unsigned long long)A.blocks_per_seg * MT_N;
    for (int b = 0; b < A.blocks_per_seg; ++b) {
        const unsigned long long wb = w0 + (unsigned long long)b * MT_N;
        for (int i = tid; i < MT_N; i += 256) A.stream[(wb + (unsigned long long)i) & A.smask] = mt_temper(key[cur][i]);
        if (b + 1 < A.blocks_per_seg) {
            twist_lds(key[cur], key[cur ^ 1], tid);
            cur ^= 1;
        }
    }
    if (k == A.last_seg)
        for (int i = tid; i < MT_N; i += 256) A.next_base[i] = key[cur][i];
}

// ---- the tokenizer ----------------------------------------------------------------------------------------------------------------
struct TokArgs {
    const uint32_t* stream;
    unsigned long long smask;
    unsigned long long* pos;           // [1] absolute stream position: in = where the batch starts, out = where it ends
    unsigned long long avail_end;      // words below this index have been generated
    uint32_t* status;                  // the context's sticky status flags (mapped host memory)
    unsigned* err;                     // [1] device flag: 1 = the stream ran out (the batch is void)
    uint32_t* J;                       // [nb][N]: J[i] = accepted Fisher-Yates target of index i (i >= 1)
    uint32_t* rint;                    // [nb][N]: accepted randint values of a non-power-of-two complement, plan order
    unsigned long long* tokpos;        // [nb][S][3]: positions of the z words, the randint words, the accept words of every split
    unsigned long long* step_end;      // [nb]: position after the step
    unsigned long long* nwindows;      // [1] statistics
    int32_t N, S, nb, randomize;
};

// exclusive prefix sum of v over the 1024-thread workgroup and the grand total.  `wsum` (16 words of LDS) must not be written again
// before every thread has left the loop below: callers alternate two buffers.
__device__ __forceinline__ uint32_t wg_exscan1024(uint32_t v, volatile uint32_t* wsum, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TOK_T / 64; ++w) {
        const uint32_t s = wsum[w];
        off += (w < wave) ? s : 0u;
        tot += s;
    }
    total = tot;
    return off + x - v;
}

// expected number of accepts among the first t words of a Fisher-Yates scan that starts at index i0 (a guess: any value would do)
__device__ __forceinline__ uint32_t shuffle_guess(uint32_t i0, uint32_t t) {
    if (t == 0 || i0 == 0) return 0;
    uint32_t m = 0xffffffffu >> __builtin_clz(i0);
    float i = (float)i0, trem = (float)t, a = 0.0f;
    for (int it = 0; it < 33; ++it) {
        const float M = (float)m + 1.0f, lo = (float)(m >> 1);
        const float tb = M * __logf((i + 1.0f) / (lo + 1.0f));            // words this band takes
        if (trem <= tb || m <= 1u) {
            a += (i + 1.0f) * (1.0f - __expf(-trem / M));
            break;
        }
        a += i - lo;
        trem -= tb;
        i = lo;
        m >>= 1;
    }
    const float cap = (float)(t < i0 ? t : i0);
    a = a < 0.0f ? 0.0f : (a > cap ? cap : a);
    return (uint32_t)(a + 0.5f);
}

static __global__ __launch_bounds__(TOK_T) void k_mt_tok(const TokArgs A) {
    __shared__ uint32_t wsum[2][TOK_T / 64];
    __shared__ uint32_t wflag[2][TOK_T / 64];
    __shared__ unsigned long long s_end;          // words consumed by the window that finished the scan
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = A.N, S = A.S;
    unsigned long long p = *A.pos;
    unsigned long long nwin = 0;
    bool dead = false;
    for (int b = 0; b < A.nb && !dead; ++b) {
        uint32_t* Jb = A.J + (size_t)b * N;
        uint32_t* Rb = A.rint + (size_t)b * N;
        p += 2;                                                     // ensemble.py:406 choice(moves, p=weights): one uniform (one move: index 0)
        if (A.randomize) {
            // ---- red_blue.py:80 random.shuffle(inds): for i = N-1 .. 1: j = random_interval(i) ----
            uint32_t i0 = (uint32_t)(N - 1);
            int guard = 0;
            while (i0 > 0) {
                if (++guard > (1 << 20)) {                          // (cannot happen: a window of >= 1024 words without one accept)
                    dead = true;
                    break;
                }
                uint32_t est = (uint32_t)(1.45f * (float)i0) + 256u;
                est = (est + TOK_T - 1) / TOK_T;
                const int wpt = (int)(est > (uint32_t)TOK_WPT ? (uint32_t)TOK_WPT : est);
                const unsigned long long nw = (unsigned long long)wpt * TOK_T;
                if (p + nw > A.avail_end) {
                    dead = true;
                    break;
                }
                ++nwin;
                const unsigned long long pw = p + (unsigned long long)tid * wpt;
                uint32_t base = shuffle_guess(i0, (uint32_t)tid * (uint32_t)wpt);
                uint32_t cnt = 0;
                int mpos = 0, mneg = 0;                             // (0: compute on the first pass)
                uint32_t base_used = 0;
                bool have = false;
                int par = 0;
                for (int iter = 0;; ++iter) {
                    if (iter > 2 * TOK_T) {                         // (cannot happen: thread k is exact after k + 1 rounds)
                        dead = true;
                        break;
                    }
                    const int dlt = (int)(base - base_used);
                    if (!have || dlt >= mpos || -dlt >= mneg) {
                        int i = (int)i0 - (int)base;
                        cnt = 0;
                        mpos = 0x7fffffff;
                        mneg = 0x7fffffff;
                        for (int j = 0; j < wpt; ++j) {
                            const uint32_t w = A.stream[(pw + (unsigned long long)j) & A.smask];
                            int dp, dn;
                            if (i > 0) {
                                const uint32_t m = 0xffffffffu >> __builtin_clz((uint32_t)i);
                                const uint32_t v = w & m;
                                const int lo = (int)(m >> 1);
                                if (v <= (uint32_t)i) {
                                    const int vv = v > 1u ? (int)v : 1;
                                    dp = i - vv + 1;
                                    dp = dp < i - lo ? dp : i - lo;
                                    dn = (int)m - i + 1;
                                    ++cnt;
                                    --i;
                                } else {
                                    dp = i - lo;
                                    dn = (int)v - i;
                                }
                            } else {
                                dp = 0x7fffffff;
                                dn = 1 - i;
                            }
                            mpos = mpos < dp ? mpos : dp;
                            mneg = mneg < dn ? mneg : dn;
                        }
                        base_used = base;
                        have = true;
                    }
                    uint32_t total;
                    const uint32_t nb_ = wg_exscan1024(cnt, wsum[par], total);
                    const bool changed = nb_ != base;
                    base = nb_;
                    const unsigned long long bal = __ballot(changed);
                    if (lane == 0) wflag[par][wave] = bal != 0ull;
                    __syncthreads();
                    uint32_t any = 0;
#pragma unroll
                    for (int w = 0; w < TOK_T / 64; ++w) any |= wflag[par][w];
                    par ^= 1;
                    if (!any) break;
                }
                if (dead) break;
                // the counts are now those of the serial scan.  Emit: J[i] = v for every accepted word; the thread that takes i
                // to 0 ends the shuffle
                if (tid == 0) s_end = nw;
                __syncthreads();
                {
                    int i = (int)i0 - (int)base;
                    for (int j = 0; j < wpt && i > 0; ++j) {
                        const uint32_t w = A.stream[(pw + (unsigned long long)j) & A.smask];
                        const uint32_t m = 0xffffffffu >> __builtin_clz((uint32_t)i);
                        const uint32_t v = w & m;
                        if (v <= (uint32_t)i) {
                            Jb[i] = v;
                            --i;
                            if (i == 0) s_end = (unsigned long long)tid * wpt + (unsigned long long)j + 1ull;
                        }
                    }
                }
                uint32_t total;
                (void)wg_exscan1024(cnt, wsum[par], total);          // (the barrier inside also publishes s_end)
                par ^= 1;
                p += s_end;
                i0 -= total;                                        // total <= i0: a thread stops accepting at i == 0
                __syncthreads();                                    // s_end is rewritten by the next window
            }
            if (dead) break;
        }
        // ---- per split: stretch.py:30 rand(Ns) | stretch.py:32 randint(Nc, Ns) | red_blue.py:100 rand() x Ns ----
        int off = 0;
        for (int s = 0; s < S && !dead; ++s) {
            const int ns = (N - s + S - 1) / S;
            const uint32_t nc = (uint32_t)(N - ns);
            unsigned long long* tp = A.tokpos + ((size_t)b * S + s) * 3;
            const unsigned long long pz = p;
            p += 2ull * (unsigned long long)ns;
            const unsigned long long pr = p;
            const uint32_t rng = nc - 1u;
            if (rng == 0u) {
                // one complement member: randint draws nothing (mt19937_legacy.hpp randint)
            } else if ((nc & rng) == 0u) {
                p += (unsigned long long)ns;                        // a power of two: one word per value, none rejected
            } else {
                // masked rejection: the ns accepted values, in order; the position after the ns-th
                const uint32_t m = 0xffffffffu >> __builtin_clz(rng);
                uint32_t got = 0;
                int guard = 0;
                while (got < (uint32_t)ns) {
                    if (++guard > (1 << 20)) {
                        dead = true;
                        break;
                    }
                    uint32_t est = (uint32_t)(((unsigned long long)((uint32_t)ns - got) * (unsigned long long)(m + 1ull)) / (rng + 1ull)) + 256u;
                    est += est >> 5;
                    est = (est + TOK_T - 1) / TOK_T;
                    const int wpt = (int)(est > (uint32_t)TOK_WPT ? (uint32_t)TOK_WPT : est);
                    const unsigned long long nw = (unsigned long long)wpt * TOK_T;
                    if (p + nw > A.avail_end) {
                        dead = true;
                        break;
                    }
                    ++nwin;
                    const unsigned long long pw = p + (unsigned long long)tid * wpt;
                    uint32_t cnt = 0;
                    for (int j = 0; j < wpt; ++j) cnt += ((A.stream[(pw + (unsigned long long)j) & A.smask] & m) <= rng) ? 1u : 0u;
                    uint32_t total;
                    uint32_t idx = got + wg_exscan1024(cnt, wsum[0], total);
                    if (tid == 0) s_end = nw;
                    __syncthreads();
                    for (int j = 0; j < wpt && idx < (uint32_t)ns; ++j) {
                        const uint32_t v = A.stream[(pw + (unsigned long long)j) & A.smask] & m;
                        if (v <= rng) {
                            Rb[off + idx] = v;
                            ++idx;
                            if (idx == (uint32_t)ns) s_end = (unsigned long long)tid * wpt + (unsigned long long)j + 1ull;
                        }
                    }
                    __syncthreads();
                    p += s_end;
                    got = got + total < (uint32_t)ns ? got + total : (uint32_t)ns;
                    __syncthreads();
                }
            }
            const unsigned long long pu = p;
            p += 2ull * (unsigned long long)ns;
            if (p > A.avail_end) dead = true;
            if (tid == 0) {
                tp[0] = pz;
                tp[1] = pr;
                tp[2] = pu;
            }
            off += ns;
        }
        if (tid == 0 && !dead) A.step_end[b] = p;
    }
    if (tid == 0) {
        if (dead) {
            *A.err = 1u;
            __hip_atomic_store(&A.status[__builtin_ctz(ST_MT_UNDERRUN)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            *A.pos = p;
        }
        if (A.nwindows) *A.nwindows += nwin;
    }
}

// ---- the finisher ---------------------------------------------------------------------------------------------------------------
struct FinArgs {
    const uint32_t* stream;
    unsigned long long smask;
    const uint32_t* J;                 // [nb][N]
    const uint32_t* rint;              // [nb][N]
    const unsigned long long* tokpos;  // [nb][S][3]
    const unsigned* err;               // the tokenizer's flag: a void batch is not finished
    uint32_t* scratch;                 // [nb][5][N]: cnt | slot | start | bucket | bmin (then labels)
    int32_t* order[16];
    int32_t* p0[16];
    double* s0[16];
    double* uacc[16];
    double* logu[16];
    double* fac[16];
    const unsigned long long* step_end;// [nb]
    uint32_t* blk_words;               // [nb][624]: the (tempered) block each step ends in -- the generator state after that step
    double a;                          // StretchMove.a
    int32_t N, D, S, randomize;
};

__device__ __forceinline__ uint32_t ld_agent_u32(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive scan of in[0, n) (read through `get`) into out, whole workgroup; returns nothing (total unused by the callers)
template <typename GET, typename PUT>
__device__ __forceinline__ void wg_scan_array(int n, GET get, PUT put, volatile uint32_t* wsum) {
    const int tid = threadIdx.x;
    const int chunk = (n + FIN_T - 1) / FIN_T;
    const int lo = tid * chunk < n ? tid * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    uint32_t sum = 0;
    for (int e = lo; e < hi; ++e) sum += get(e);
    uint32_t total;
    uint32_t run = wg_exscan1024(sum, wsum, total);
    for (int e = lo; e < hi; ++e) {
        const uint32_t v = get(e);
        put(e, run, v);
        run += v;
    }
    __syncthreads();
}

static __global__ __launch_bounds__(FIN_T) void k_mt_fin(const FinArgs A) {
    __shared__ uint32_t wsum[2][FIN_T / 64];
    if (*A.err) return;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = A.N, S = A.S;
    const uint32_t* J = A.J + (size_t)b * N;
    uint32_t* cnt = A.scratch + (size_t)b * 5 * N;
    uint32_t* slot = cnt + N;
    uint32_t* start = slot + N;
    uint32_t* bucket = start + N;
    uint32_t* bmin = bucket + N;
    uint32_t* label = cnt;                       // (cnt is dead once `start` exists)
    int32_t* order = A.order[b];
    constexpr uint32_t INF = 0xffffffffu;
    {   // RandomState.get_state() after this step (ensemble.py:410): the block the position stands in, kept outside the ring
        const unsigned long long a = A.step_end[b];
        const unsigned long long w0 = (a > 0 ? (a - 1ull) / MT_N : 0ull) * MT_N;
        for (int i = tid; i < MT_N; i += FIN_T) A.blk_words[(size_t)b * MT_N + i] = A.stream[(w0 + (unsigned long long)i) & A.smask];
    }
    if (A.randomize) {
        // red_blue.py:78-80: labels i % S, then the swaps (i, J[i]) for i = N-1 .. 1.  Position i is final once step i has run
        // and then holds what position J[i] held just before; that value was put there by the latest earlier step -- the
        // SMALLEST i' > i with J[i'] = J[i] -- from position i', which in turn ... : a chain through the buckets of swap targets
        // that ends at a position no earlier step has hit, whose initial label (origin % S) is the answer.
        for (int e = tid; e < N; e += FIN_T) {
            cnt[e] = 0;
            bmin[e] = INF;
        }
        __syncthreads();
        for (int i = 1 + tid; i < N; i += FIN_T) {
            const uint32_t j = J[i];
            if (j < (uint32_t)i) {
                slot[i] = atomicAdd(&cnt[j], 1u);
                atomicMin(&bmin[j], (uint32_t)i);
            }
        }
        __syncthreads();
        wg_scan_array(N, [&](int e) { return ld_agent_u32(cnt + e); }, [&](int e, uint32_t pre, uint32_t) { start[e] = pre; }, wsum[0]);
        for (int i = 1 + tid; i < N; i += FIN_T) {
            const uint32_t j = J[i];
            if (j < (uint32_t)i) bucket[start[j] + slot[i]] = (uint32_t)i;
        }
        __syncthreads();
        // (cnt is read once more below -- bucket sizes -- before `label` overwrites it: labels go to `slot` first)
        for (int i = tid; i < N; i += FIN_T) {
            const uint32_t j = i == 0 ? 0u : J[i];
            uint32_t cand = INF;
            const uint32_t s0 = start[j], n0 = ld_agent_u32(cnt + j);
            for (uint32_t e = 0; e < n0; ++e) {
                const uint32_t h = bucket[s0 + e];
                if (h > (uint32_t)i && h < cand) cand = h;
            }
            uint32_t q = j;
            if (cand != INF) {
                q = cand;
                for (;;) {
                    const uint32_t h = ld_agent_u32(bmin + q);
                    if (h == INF) break;
                    q = h;
                }
            }
            slot[i] = q % (uint32_t)S;
        }
        __syncthreads();
        for (int e = tid; e < N; e += FIN_T) label[e] = slot[e];
        __syncthreads();
    } else {
        for (int e = tid; e < N; e += FIN_T) label[e] = (uint32_t)(e % S);
        __syncthreads();
    }
    // red_blue.py:85 boolean-mask order: ascending walker index inside each set, sets in label order
    {
        int off = 0;
        for (int s = 0; s < S; ++s) {
            const int ns = (N - s + S - 1) / S;
            wg_scan_array(N, [&](int e) { return label[e] == (uint32_t)s ? 1u : 0u; },
                          [&](int e, uint32_t pre, uint32_t v) {
                              if (v) order[off + (int)pre] = e;
                          },
                          wsum[s & 1]);
            off += ns;
        }
    }
    __syncthreads();
    // stretch.py:30-33 and red_blue.py:100: partners resolved against the complement (the sets before the split, then after),
    // zz = ((a - 1) u + 1)^2 / a, the accept uniforms; logs as k_plan_logs takes them
    {
        const double a = A.a, dm1 = (double)A.D - 1.0;
        int off = 0;
        for (int s = 0; s < S; ++s) {
            const int ns = (N - s + S - 1) / S;
            const uint32_t nc = (uint32_t)(N - ns);
            const unsigned long long* tp = A.tokpos + ((size_t)b * S + s) * 3;
            const unsigned long long pz = tp[0], pr = tp[1], pu = tp[2];
            const uint32_t rng = nc - 1u;
            const bool pow2 = rng != 0u && (nc & rng) == 0u;
            for (int t = tid; t < ns; t += FIN_T) {
                uint32_t r;
                if (rng == 0u)
                    r = 0u;
                else if (pow2)
                    r = A.stream[(pr + (unsigned long long)t) & A.smask] & rng;
                else
                    r = A.rint[(size_t)b * N + off + t];
                A.p0[b][off + t] = (int)r < off ? order[r] : order[r + ns];
                {
                    const uint32_t w0 = A.stream[(pz + 2ull * t) & A.smask], w1 = A.stream[(pz + 2ull * t + 1ull) & A.smask];
                    const double u = ((double)(int)(w0 >> 5) * 67108864.0 + (double)(int)(w1 >> 6)) / 9007199254740992.0;
                    const double tt = (a - 1.0) * u + 1.0;
                    const double zz = tt * tt / a;
                    A.s0[b][off + t] = zz;
                    A.fac[b][off + t] = dm1 * log(zz);
                }
                {
                    const uint32_t w0 = A.stream[(pu + 2ull * t) & A.smask], w1 = A.stream[(pu + 2ull * t + 1ull) & A.smask];
                    const double u = ((double)(int)(w0 >> 5) * 67108864.0 + (double)(int)(w1 >> 6)) / 9007199254740992.0;
                    A.uacc[b][off + t] = u;
                    A.logu[b][off + t] = log(u);
                }
            }
            off += ns;
        }
    }
}

}  // namespace mtdev
}  // namespace emx
