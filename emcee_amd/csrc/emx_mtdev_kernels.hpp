// Kernels of the device-side exact-plan producer (see emx_mtdev.hpp for the design).  gfx950, wave64.
//
// The generator's and the finisher's kernels are many small workgroups (256 threads, a few KB of LDS); the tokenizer is ONE
// 1024-thread workgroup with two 52 KB windows of LDS: it needs a CU of its own, which a persistent consumer (k_persist holds one
// 103 KB workgroup on every CU for a whole batch of steps) does not leave -- exact mode runs the per-half-step launches.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "emx_mtjump.hpp"      // MT_N
#include "emx_planlog.hpp"

namespace emx {
namespace mtdev {

constexpr int WIN_BLOCKS = 33;                   // 33 * 624 = 20 592 >= 19 937 + 624 words: the window a jump reads
constexpr int JUMP_SPLIT = 8;                    // workgroups sharing one jump polynomial (78 polynomial words each)
constexpr int SEG_BLOCKS = 128;                  // blocks per segment: the stride of the jump polynomials
constexpr unsigned long long SEG_WORDS = (unsigned long long)SEG_BLOCKS * MT_N;
constexpr int PMAX = 128;                        // segments per round at most (a full round: 10.2 M words, ~24 steps of 65 536 walkers)
constexpr int FIN_T = 256;
constexpr int FIN_CHUNK = 4096;                  // elements per workgroup of the finisher's scans (16 per thread)
constexpr int FIN_MAX_S = 64;
constexpr unsigned ST_MT_PRODUCER = 16u;         // status bit 4 (ST_PLAN_PRODUCER of emx_kernels.hpp): the stream ran out under the tokenizer, or a stage
                                                 // waited for another beyond the bound -- either way the steps taken from the producer are void

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
// ---- cross-stream dependencies without events --------------------------------------------------------------------------
// The producer's stages live on streams of their own and used to wait for each other through hipStreamWaitEvent.  Measured
// (profiles/r04/mtdev_gates.txt, and emx.hip: pipe_fetch_deferred): a wait on an event recorded several launches earlier does NOT
// resolve to that record -- the runtime learns of completions lazily and orders the waiting stream behind the other stream's LATEST
// work -- so "the tokenizer of batch n + 2 waits for the finisher of batch n" became "... of batch n + 1", the generator waited for
// that finisher too, and the three stages ran one after the other (tokenizer 1 100 + finisher 285 + generator 196 us a batch).
// Instead every stage counts what it has completed in a device word (k_gate_signal, the last kernel of the stage's work on its
// in-order stream) and whoever depends on it starts with k_gate_wait: one wave that polls the word.  The kernel boundaries on
// either side do the release / acquire.  A wait that is never met -- a stage that died -- raises bit 2 of the error word after
// `timeout_ticks` (100 MHz), status bit 4 of the context with it, and lets its stream go on: the run is void and says so
// (emx_status; every call that retires the producer returns the error as well).
static __global__ void k_gate_signal(unsigned long long* word, unsigned long long value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
static __global__ void k_gate_wait(const unsigned long long* word, unsigned long long value, unsigned long long timeout_ticks, unsigned* err,
                                   unsigned long long* waited, uint32_t* status) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
        if (wall_clock64() - t0 > timeout_ticks) {
            atomicOr(err, 4u);           // (the finisher kernels return at once from here on: emx_mtdev.hip reads it in finish())
            __hip_atomic_store(&status[__builtin_ctz(ST_MT_PRODUCER)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // the context's sticky status
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    if (waited) {          // (statistics: 10 ns ticks this site has waited, and how often)
        atomicAdd(waited, wall_clock64() - t0);
        atomicAdd(waited + 1, 1ull);
    }
}

// the end positions of a tokenised batch, straight into pinned host memory, then the batch's number into a pinned word: the host
// follows the tokenizer by reading that word (hipEventSynchronize on an event of this stream waited for the stream's LATEST
// work -- the tokenizer of the batch just enqueued, 1.1 ms -- and kept the host from enqueueing ahead)
static __global__ void k_pos_publish(const unsigned long long* __restrict__ step_end, unsigned long long* __restrict__ h_end, int n,
                                     unsigned long long* h_done, unsigned long long value) {
    if ((int)threadIdx.x < n) h_end[threadIdx.x] = step_end[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_done, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

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
// 256 threads, three output words each (j = tid, tid + 256, tid + 512 < 624)
static __global__ __launch_bounds__(256) void k_mt_jump(const uint32_t* __restrict__ polys, const uint32_t* __restrict__ xwin,
                                                        uint32_t* __restrict__ partial) {
    constexpr int WPS = MT_N / JUMP_SPLIT;                // polynomial words per workgroup (78 -> 2 496 coefficients)
    __shared__ uint32_t Xs[WPS * 32 + MT_N + 256];        // (+ 256: the third output word of threads >= 112 reads, and discards, past the end)
    const int q = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const int i0 = q * WPS * 32;
    for (int e = tid; e < WPS * 32 + MT_N + 256; e += 256) Xs[e] = e < WPS * 32 + MT_N ? xwin[i0 + e] : 0u;   // (the last index read is 33 * 624 - 1)
    __syncthreads();
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    const uint32_t* g = polys + (size_t)k * MT_N + q * WPS;
    for (int w = 0; w < WPS; ++w) {
        uint32_t gw = __builtin_amdgcn_readfirstlane(g[w]);                    // uniform: the loop below is scalar control flow
        const uint32_t* xs = Xs + w * 32 + tid;
        while (gw) {
            const int b = __builtin_ctz(gw);
            gw &= gw - 1u;
            a0 ^= xs[b];
            a1 ^= xs[b + 256];
            a2 ^= xs[b + 512];
        }
    }
    uint32_t* out = partial + ((size_t)k * JUMP_SPLIT + q) * MT_N;
    out[tid] = a0;
    out[tid + 256] = a1;
    if (tid + 512 < MT_N) out[tid + 512] = a2;
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
// ONE workgroup walks the stream in the reference's draw order and does only what decides the position: the masked rejection tests
// of random.shuffle (red_blue.py:80) and of a non-power-of-two randint (stretch.py:32).
//
// A window of 1024 x wpt words is decided at once: every thread runs its wpt consecutive words exactly, from a guessed count of
// accepts before them; the counts are prefix-summed and the guesses replaced until nothing changes -- a fixed point of that iteration
// IS the serial result (thread 0 is exact at once, thread k once the threads before it are).  A thread recomputes only when its
// count of earlier accepts has moved further than the smallest distance of any of its tests from the other decision (its margin).
// How fast this settles is a matter of how many tests a unit shift flips: 2 x window / mask.  So the window shrinks with the mask
// (wpt = 13, 7, 3, 1 -- odd: conflict-free LDS rows; wshift 11: 13 words a thread down to mask 2^15, 7 at 2^14, 3 at 2^13), and the last few thousand indices, where even 1024 words are too many, are
// walked by ONE wave, 64 words at a time, the same fixed point taken with ballots instead of barriers.
//
// What it leaves behind for the (parallel) finisher are WALK RECORDS: for every window, where it starts and, per thread, the state
// that thread's words start from -- the Fisher-Yates index i (a shuffle window), or the count of values accepted so far (a randint
// window).  Replaying a thread's words from that state is exact and independent of every other thread, so writing out the accepted
// targets J[i] / values is the finisher's work, not the serial path's.  (The one-wave tail writes its few J[i] itself.)
constexpr int TOK_T = 1024;                      // threads of the tokenizer workgroup
constexpr int TOK_WPT_MAX = 13;                  // consecutive words per thread at most
constexpr int TOK_W_MAX = TOK_T * TOK_WPT_MAX;   // words per window at most (13 312); two windows of LDS: 104 KB

struct WalkRec {
    unsigned long long pos;            // absolute stream position of the window's first word
    uint32_t kind;                     // 0: shuffle; 1 + s: the randint of split s
    uint32_t wpt;                      // words per thread
    uint32_t state[TOK_T];
};

struct TokArgs {
    const uint32_t* stream;
    unsigned long long smask;
    unsigned long long* pos;           // [1] absolute stream position: in = where the batch starts, out = where it ends
    unsigned long long avail_end;      // words below this index have been generated
    uint32_t* status;                  // the context's sticky status flags (mapped host memory)
    unsigned* err;                     // [1] device flag: 1 = the stream ran out (the batch is void)
    WalkRec* recs;                     // [nb][maxrec]
    uint32_t* nrec;                    // [nb]
    uint32_t* J;                       // [nb][N]: the tail's accepted Fisher-Yates targets (the finisher writes the rest)
    unsigned long long* tokpos;        // [nb][S][3]: positions of the z words, the randint words, the accept words of every split
    unsigned long long* step_end;      // [nb]: position after the step
    unsigned long long* stats;         // [8] windows | fixed-point rounds | tail groups | tail rounds | 10 ns ticks: waiting for a window | deciding chunk windows | the tail | whole kernel
    int32_t N, S, nb, randomize, maxrec;
    int32_t wshift;                    // words per thread = (mask + 1) >> wshift (default 11), rounded down to 13 / 7 / 3 / 1
    int32_t tail;                      // indices at or below this are walked by one wave
};

// inclusive prefix sum over the wave (DPP: four shifts inside the rows of 16, two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return x;
}

// exclusive prefix sum of v (< 2^31 in total) over the 1024-thread workgroup and the grand total; the OR of `flag` over the workgroup
// comes back in any_flag.  ONE barrier; `buf` (16 words of LDS) must not be written again before every thread has read it: callers
// alternate two.  The sixteen wave totals are read once (lane l takes total l mod 16) and scanned in registers -- read one after the
// other as 32 volatile words they cost 1.5 us a round, more than everything else in it.
__device__ __forceinline__ uint32_t wg_exscan1024(uint32_t v, bool flag, uint32_t* buf, uint32_t& total, bool& any_flag) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t x = wave_incl_scan(v);
    const unsigned long long bal = __ballot(flag);
    if (lane == 63) buf[wave] = x | (bal != 0ull ? 0x80000000u : 0u);
    __syncthreads();
    const uint32_t pv = buf[lane & 15];
    uint32_t inc = pv & 0x7fffffffu;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);      // lanes 0 .. 15: inclusive scan of the wave totals
    total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15);
    const uint32_t off = wave == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)inc, wave - 1);
    any_flag = (__ballot((pv >> 31) != 0u) & 0xffffull) != 0ull;
    return off + x - v;
}

__device__ __forceinline__ uint32_t mask_of(uint32_t i) { return 0xffffffffu >> __builtin_clz(i | 1u); }      // smallest 2^k - 1 >= i (i >= 1)

__device__ __forceinline__ int tok_wpt_of(uint32_t i0, int wshift) {
    const uint32_t w = (mask_of(i0) + 1u) >> wshift;
    return w >= 13u ? 13 : w >= 7u ? 7 : w >= 3u ? 3 : 1;
}

// One thread's WPT consecutive words of a shuffle window, from index i (<= 0: the scan is over).  -> accepts; `margin` = how far the
// start index may move, either way, with every decision (and so the count) staying what it is: recomputing is needed only past it.
template <int WPT>
__device__ __forceinline__ uint32_t tok_walk_count_t(const uint32_t* win, int i, int& margin) {
    if (i <= 0) {
        margin = 0x7fffffff;              // (a move of the start index to a positive value is caught by the caller)
        return 0;
    }
    const uint32_t m = mask_of((uint32_t)i);
    const int lo = (int)(m >> 1);
    if (i - WPT > lo) {
        // the whole walk stays inside one mask band and cannot reach 0: six instructions a word.  x = v - i - 1 is negative exactly
        // for an accept; as unsigned numbers, x is the distance (less one) of a reject from acceptance and ~x = i - v that of an accept
        // from rejection, and whichever does not apply is huge: one three-way minimum tracks both
        uint32_t w[WPT];
#pragma unroll
        for (int j = 0; j < WPT; ++j) w[j] = win[j];
        uint32_t mg = 0x7fffffffu;
        int i1 = i + 1;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int x = (int)(w[j] & m) - i1;
            const uint32_t ux = (uint32_t)x;
            mg = mg < ux ? mg : ux;
            mg = mg < ~ux ? mg : ~ux;
            i1 += x >> 31;
        }
        int mgi = (int)mg + 1;
        const int up = (int)m - i + 1, dn = i - WPT - lo;         // the start index leaves the band | the walk could leave it
        mgi = mgi < up ? mgi : up;
        margin = mgi < dn ? mgi : dn;
        return (uint32_t)(i + 1 - i1);
    }
    uint32_t cnt = 0;
    for (int j = 0; j < WPT && i > 0; ++j) {
        const uint32_t v = win[j] & mask_of((uint32_t)i);
        if (v <= (uint32_t)i) {
            ++cnt;
            --i;
        }
    }
    margin = 1;                                                      // any move of the start index recomputes
    return cnt;
}
__device__ __forceinline__ uint32_t tok_walk_count(const uint32_t* win, int wpt, int i, int& margin) {
    switch (wpt) {                                                   // (wave-uniform)
        case 13: return tok_walk_count_t<13>(win, i, margin);
        case 7: return tok_walk_count_t<7>(win, i, margin);
        case 3: return tok_walk_count_t<3>(win, i, margin);
        default: return tok_walk_count_t<1>(win, i, margin);
    }
}

// `nwords` (a multiple of 1024) words of the stream from position p into LDS, asynchronously and past the registers
// (global_load_lds: every wave's 64 consecutive words land at the wave-uniform LDS address + 4 * lane); tok_window_wait() before the
// barrier that publishes them.  Ring indices are 32-bit (the ring holds at most 2^32 words).
__device__ __forceinline__ void tok_window_load(const uint32_t* __restrict__ stream, uint32_t smask32, unsigned long long p, uint32_t* win, int rows) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // (scalar: the LDS addresses stay out of the VGPRs)
    const uint32_t o = (uint32_t)p + (uint32_t)tid;
#pragma nounroll
    for (int k = 0; k < rows; ++k)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + ((o + (uint32_t)(k * TOK_T)) & smask32)),
                                         (__attribute__((address_space(3))) void*)(win + k * TOK_T + wave * 64), 4, 0, 0);
}
__device__ __forceinline__ void tok_window_wait() { __builtin_amdgcn_s_waitcnt(0x0f70); }       // vmcnt(0)

static __global__ __launch_bounds__(TOK_T) void k_mt_tok(const TokArgs A) {
    __shared__ uint32_t win2[2][TOK_W_MAX];
    __shared__ uint32_t sbuf[2][16];
    __shared__ uint32_t sflag[2][16];
    __shared__ unsigned long long s_end;          // words consumed by the window that finished a scan
    __shared__ uint32_t s_i0;                     // the tail's index when it leaves a window
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = A.N, S = A.S;
    const uint32_t smask32 = (uint32_t)A.smask;
    unsigned long long p = *A.pos;
    unsigned long long nwin = 0, nround = 0, ngroup = 0, ntround = 0;
    unsigned long long t_wait = 0, t_chunk = 0, t_tail = 0;
    const unsigned long long t_begin = wall_clock64();
#ifdef EMX_TOK_PROFILE
    unsigned long long pf[6] = {0, 0, 0, 0, 0, 0}, pq = 0;
#define TOKP(k_) do { const unsigned long long t_ = wall_clock64(); pf[k_] += t_ - pq; pq = t_; } while (0)
#else
#define TOKP(k_) do { } while (0)
#endif
#ifdef EMX_TOK_PROFILE
    unsigned long long t_walk0 = 0, n_walk0 = 0, t_scan0 = 0;
#endif
    bool dead = false;
    int par = 0;
    for (int b = 0; b < A.nb && !dead; ++b) {
        WalkRec* recs = A.recs + (size_t)b * A.maxrec;
        uint32_t* Jb = A.J + (size_t)b * N;
        int nrec = 0;
        p += 2;                                                     // ensemble.py:406 choice(moves, p=weights): one uniform (one move: index 0)
        if (A.randomize) {
            // ---- red_blue.py:80 random.shuffle(inds): for i = N-1 .. 1: j = random_interval(i) ----
            uint32_t i0 = (uint32_t)(N - 1);
            int cur = 0;
            int rows = tok_wpt_of(i0, A.wshift);                    // rows of 1024 words in flight into win2[cur]
            tok_window_load(A.stream, smask32, p, win2[0], rows);
            while (i0 > 0) {
                const int wpt = tok_wpt_of(i0, A.wshift);           // (<= rows: the index only falls)
                const int W = wpt * TOK_T;
                if (p + (unsigned long long)W > A.avail_end || nrec >= A.maxrec) {
                    dead = true;
                    break;
                }
                ++nwin;
                const unsigned long long tw0 = wall_clock64();
                tok_window_wait();
                __syncthreads();                                    // this window is in; every thread is done with the previous one
                const unsigned long long tw1 = wall_clock64();
                t_wait += tw1 - tw0;
                // a window inside the shuffle is consumed whole unless it ends the scan (then the prefetch is dropped; words not yet
                // generated are never used: the check above comes first).  The next window is no wider than this one.
                tok_window_load(A.stream, smask32, p + (unsigned long long)W, win2[cur ^ 1], wpt);
                rows = wpt;
                const uint32_t* win = win2[cur];
                cur ^= 1;
                if (i0 <= (uint32_t)A.tail) {
                    // ---- the last indices: one wave, 64 words at a time, the fixed point with ballots ----
                    if (tid < 64) {
                        uint32_t i = i0;
                        int g = 0;
                        uint32_t w = win[lane];
                        for (; g < W && i > 0; g += 64) {
                            const uint32_t wn = g + 64 < W ? win[g + 64 + lane] : 0u;         // the next group's words, on their way
                            ++ngroup;
                            unsigned long long acc;
                            const uint32_t m = mask_of(i);
                            if (i > 64u + (m >> 1)) {
                                // the whole group inside one mask band, no index reaches 0: v is fixed, a round is
                                // count-below / subtract / compare / ballot
                                const uint32_t v = w & m;
                                acc = __ballot(v <= i);
                                for (;;) {
                                    ++ntround;
                                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(acc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc, 0u));
                                    const unsigned long long nacc = __ballot(v <= i - below);
                                    if (nacc == acc) break;
                                    acc = nacc;
                                }
                            } else {
                                acc = 0ull;
                                for (;;) {
                                    ++ntround;
                                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(acc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc, 0u));
                                    const int ik = (int)i - (int)below;
                                    const uint32_t v = w & mask_of((uint32_t)(ik > 0 ? ik : 1));
                                    const unsigned long long nacc = __ballot(ik > 0 && v <= (uint32_t)ik);
                                    if (nacc == acc) break;
                                    acc = nacc;
                                }
                            }
                            if ((acc >> lane) & 1ull) {
                                const uint32_t ik = i - __builtin_amdgcn_mbcnt_hi((uint32_t)(acc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc, 0u));
                                Jb[ik] = w & mask_of(ik);
                                if (ik == 1u) s_end = (unsigned long long)(g + lane + 1);      // the accept that takes i to 0 ends the shuffle
                            }
                            i -= (uint32_t)__popcll(acc);
                            w = wn;
                        }
                        if (lane == 0) {
                            s_i0 = i;
                            if (i > 0) s_end = (unsigned long long)W;
                        }
                    }
                    __syncthreads();
                    i0 = s_i0;
                    p += s_end;
                    t_tail += wall_clock64() - tw1;
                    continue;
                }
#ifdef EMX_TOK_PROFILE
                pq = wall_clock64();
#endif
                const uint32_t* mine = win + tid * wpt;
                // a guess of the accepts before this thread's words: rate (i0 + 1) / (m + 1), falling as i does
                uint32_t base;
                {
                    const float M = (float)mask_of(i0) + 1.0f, x = (float)(tid * wpt), a0 = ((float)i0 + 1.0f) / M;
                    float g = a0 * x * (1.0f - 0.5f * x / M);
                    g = g < 0.0f ? 0.0f : g;
                    base = (uint32_t)(g + 0.5f);
                    base = base < i0 ? base : i0;
                }
                uint32_t cnt = 0, base_used = 0;
                int margin = 0;
                bool have = false;
                uint32_t total = 0;
                for (int iter = 0;; ++iter) {
                    if (iter > TOK_T + 4) {                         // (cannot happen: thread k is exact after k + 1 rounds)
                        dead = true;
                        break;
                    }
                    ++nround;
                    TOKP(0);          // guess / loop overhead
                    {
                        const int dlt = (int)(base - base_used);
                        const int adl = dlt < 0 ? -dlt : dlt;
                        if (!have || (adl != 0 && (adl >= margin || (int)i0 - (int)base_used <= 0))) {
                            cnt = tok_walk_count(mine, wpt, (int)i0 - (int)base, margin);
                            base_used = base;
                            have = true;
                        }
                    }
                    TOKP(1);          // walk (this wave's)
                    bool dummy;
                    base = wg_exscan1024(cnt, false, sbuf[par], total, dummy);
                    TOKP(2);          // scan + barrier (waits for the slowest wave's walk)
                    par ^= 1;
                    // Done when every thread's count still holds for the base the scan just gave it (inside its margin): the counts
                    // then ARE those of these bases -- the fixed point -- and no further scan is needed to see it.
                    const int dlt = (int)(base - base_used);
                    const int adl = dlt < 0 ? -dlt : dlt;
                    const bool again = adl != 0 && (adl >= margin || (int)i0 - (int)base_used <= 0);
                    const unsigned long long bal = __ballot(again);
                    if (lane == 0) sflag[par][tid >> 6] = bal != 0ull;
                    __syncthreads();
                    const bool any = (__ballot(sflag[par][lane & 15] != 0u) & 0xffffull) != 0ull;
                    TOKP(3);          // margin check + barrier
                    if (!any) break;
                }
                if (dead) break;
                // the counts are now those of the serial scan
                const int istart = (int)i0 - (int)base > 0 ? (int)i0 - (int)base : 0;
                recs[nrec].state[tid] = (uint32_t)istart;
                if (tid == 0) {
                    recs[nrec].pos = p;
                    recs[nrec].kind = 0u;
                    recs[nrec].wpt = (uint32_t)wpt;
                    s_end = (unsigned long long)W;
                }
                ++nrec;
                if (total >= i0) {
                    // the scan ends in this window: the thread that takes i to 0 says where
                    __syncthreads();
                    if (istart > 0 && (uint32_t)istart == cnt) {
                        int i = istart;
                        for (int j = 0; j < wpt; ++j) {
                            const uint32_t v = mine[j] & mask_of((uint32_t)i);
                            if (v <= (uint32_t)i && --i == 0) {
                                s_end = (unsigned long long)(tid * wpt + j + 1);
                                break;
                            }
                        }
                    }
                    __syncthreads();
                    p += s_end;
                    i0 = 0;
                } else {
                    p += (unsigned long long)W;
                    i0 -= total;
                }
                TOKP(4);              // record, bookkeeping
                t_chunk += wall_clock64() - tw1;
            }
            tok_window_wait();                                      // (the dropped prefetch has landed before its buffer is used again)
            if (dead) break;
        }
        // ---- per split: stretch.py:30 rand(Ns) | stretch.py:32 randint(Nc, Ns) | red_blue.py:100 rand() x Ns ----
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
                constexpr int wpt = TOK_WPT_MAX, W = TOK_W_MAX;
                while (got < (uint32_t)ns) {
                    if (p + (unsigned long long)W > A.avail_end || nrec >= A.maxrec) {
                        dead = true;
                        break;
                    }
                    ++nwin;
                    __syncthreads();
                    tok_window_load(A.stream, smask32, p, win2[0], wpt);
                    tok_window_wait();
                    __syncthreads();
                    const uint32_t* mine = win2[0] + tid * wpt;
                    uint32_t cnt = 0;
                    for (int j = 0; j < wpt; ++j) cnt += ((mine[j] & m) <= rng) ? 1u : 0u;
                    uint32_t total;
                    bool any;
                    const uint32_t idx = got + wg_exscan1024(cnt, false, sbuf[par], total, any);
                    par ^= 1;
                    recs[nrec].state[tid] = idx;
                    if (tid == 0) {
                        recs[nrec].pos = p;
                        recs[nrec].kind = 1u + (uint32_t)s;
                        recs[nrec].wpt = (uint32_t)wpt;
                        s_end = (unsigned long long)W;
                    }
                    ++nrec;
                    if (got + total >= (uint32_t)ns) {
                        __syncthreads();
                        if (idx < (uint32_t)ns && idx + cnt >= (uint32_t)ns) {
                            uint32_t k = idx;
                            for (int j = 0; j < wpt; ++j)
                                if ((mine[j] & m) <= rng && ++k == (uint32_t)ns) {
                                    s_end = (unsigned long long)(tid * wpt + j + 1);
                                    break;
                                }
                        }
                        __syncthreads();
                        p += s_end;
                        got = (uint32_t)ns;
                    } else {
                        p += (unsigned long long)W;
                        got += total;
                    }
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
        }
        if (tid == 0 && !dead) {
            A.step_end[b] = p;
            A.nrec[b] = (uint32_t)nrec;
        }
    }
    if (tid == 0) {
        if (dead) {
            *A.err = 1u;
            __hip_atomic_store(&A.status[__builtin_ctz(ST_MT_PRODUCER)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            *A.pos = p;
        }
        if (A.stats) {
            A.stats[0] += nwin;
            A.stats[1] += nround;
            A.stats[2] += ngroup;
            A.stats[3] += ntround;
            A.stats[4] += t_wait;
            A.stats[5] += t_chunk;
            A.stats[6] += t_tail;
            A.stats[7] += wall_clock64() - t_begin;
#ifdef EMX_TOK_PROFILE
            for (int k = 0; k < 5; ++k) A.stats[8 + k] += pf[k];
#endif
#ifdef EMX_TOK_PROFILE
            A.stats[8] += n_walk0;
            A.stats[9] += t_walk0;
            A.stats[10] += t_scan0;
#endif
        }
    }
}

// ---- the finisher: a batch's 16 steps side by side, every kernel a grid of (chunks, steps) ----------------------------------------
struct FinArgs {
    const uint32_t* stream;
    unsigned long long smask;
    const WalkRec* recs;               // [nb][maxrec]
    const uint32_t* nrec;              // [nb]
    uint32_t* J;                       // [nb][N]: J[i] = accepted Fisher-Yates target of index i (i >= 1)
    uint32_t* rint;                    // [nb][N]: accepted randint values of a non-power-of-two complement, plan order
    const unsigned long long* tokpos;  // [nb][S][3]
    const unsigned* err;               // the tokenizer's flag: a void batch is not finished
    uint32_t* scratch;                 // [nb][5][N]: cnt | slot (then labels) | start | bucket | bmin
    uint32_t* partial;                 // [nb][nchunk]: chunk sums of cnt
    uint32_t* hist;                    // [nb][nchunk][S]: labels per chunk
    int32_t* order[16];
    int32_t* p0[16];
    double* s0[16];
    double* uacc[16];
    double* logu[16];
    double* fac[16];
    const unsigned long long* step_end;// [nb]
    uint32_t* blk_words;               // [nb][624]: the (tempered) block each step ends in -- the generator state after that step
    double a;                          // StretchMove.a
    int32_t N, D, S, randomize, maxrec, nchunk;
};

__device__ __forceinline__ uint32_t* fin_cnt(const FinArgs& A, int b) { return A.scratch + (size_t)b * 5 * A.N; }
__device__ __forceinline__ uint32_t* fin_slot(const FinArgs& A, int b) { return fin_cnt(A, b) + A.N; }
__device__ __forceinline__ uint32_t* fin_start(const FinArgs& A, int b) { return fin_cnt(A, b) + 2 * (size_t)A.N; }
__device__ __forceinline__ uint32_t* fin_bucket(const FinArgs& A, int b) { return fin_cnt(A, b) + 3 * (size_t)A.N; }
__device__ __forceinline__ uint32_t* fin_bmin(const FinArgs& A, int b) { return fin_cnt(A, b) + 4 * (size_t)A.N; }
constexpr uint32_t FIN_INF = 0xffffffffu;

// replay the walk records: J[i] of every accepted Fisher-Yates word, the accepted randint values; also the hit counters' reset
static __global__ __launch_bounds__(FIN_T) void k_fin_walk(const FinArgs A) {
    if (*A.err) return;
    const int r = blockIdx.x >> 2, b = blockIdx.y, t = (blockIdx.x & 3) * FIN_T + threadIdx.x;        // four blocks per 1024-thread record
    const int N = A.N, S = A.S;
    if (r >= (int)A.nrec[b]) return;
    const WalkRec& R = A.recs[(size_t)b * A.maxrec + r];
    const int wpt = (int)R.wpt;
    const unsigned long long pw = R.pos + (unsigned long long)(t * wpt);
    if (R.kind == 0u) {
        uint32_t* J = A.J + (size_t)b * N;
        int i = (int)R.state[t];
        for (int j = 0; j < wpt && i > 0; ++j) {
            const uint32_t v = A.stream[(pw + (unsigned long long)j) & A.smask] & mask_of((uint32_t)i);
            if (v <= (uint32_t)i) {
                J[i] = v;
                --i;
            }
        }
    } else {
        const int s = (int)R.kind - 1;
        const int q = N / S, rem = N % S;
        const int ns = q + (s < rem ? 1 : 0), off = s * q + (s < rem ? s : rem);
        const uint32_t rng = (uint32_t)(N - ns) - 1u, m = 0xffffffffu >> __builtin_clz(rng);
        uint32_t* Rb = A.rint + (size_t)b * N + off;
        uint32_t idx = R.state[t];
        for (int j = 0; j < wpt && idx < (uint32_t)ns; ++j) {
            const uint32_t v = A.stream[(pw + (unsigned long long)j) & A.smask] & m;
            if (v <= rng) Rb[idx++] = v;
        }
    }
}

static __global__ __launch_bounds__(FIN_T) void k_fin_init(const FinArgs A) {
    if (*A.err) return;
    const int b = blockIdx.y, N = A.N;
    uint32_t *cnt = fin_cnt(A, b), *bmin = fin_bmin(A, b);
    for (int e = blockIdx.x * FIN_CHUNK + threadIdx.x; e < N && e < (int)(blockIdx.x + 1) * FIN_CHUNK; e += FIN_T) {
        cnt[e] = 0;
        bmin[e] = FIN_INF;
    }
    if (blockIdx.x == 0) {
        // RandomState.get_state() after this step (ensemble.py:410): the block the position stands in, kept outside the ring
        const unsigned long long a = A.step_end[b];
        const unsigned long long w0 = (a > 0 ? (a - 1ull) / MT_N : 0ull) * MT_N;
        for (int i = threadIdx.x; i < MT_N; i += FIN_T) A.blk_words[(size_t)b * MT_N + i] = A.stream[(w0 + (unsigned long long)i) & A.smask];
    }
}

// red_blue.py:78-80: labels i % S, then the swaps (i, J[i]) for i = N-1 .. 1.  Position i is final once step i has run and then
// holds what position J[i] held just before; that value was put there by the latest earlier step -- the SMALLEST i' > i with
// J[i'] = J[i] -- from position i', which in turn ... : a chain through the buckets of swap targets that ends at a position no
// earlier step has hit, whose initial label (origin % S) is the answer.
static __global__ __launch_bounds__(FIN_T) void k_fin_hits(const FinArgs A) {
    if (*A.err) return;
    const int b = blockIdx.y, N = A.N;
    const uint32_t* J = A.J + (size_t)b * N;
    uint32_t *cnt = fin_cnt(A, b), *slot = fin_slot(A, b), *bmin = fin_bmin(A, b);
    for (int i = blockIdx.x * FIN_CHUNK + threadIdx.x; i < N && i < (int)(blockIdx.x + 1) * FIN_CHUNK; i += FIN_T) {
        if (i == 0) continue;
        const uint32_t j = J[i];
        if (j < (uint32_t)i) {
            slot[i] = atomicAdd(&cnt[j], 1u);
            atomicMin(&bmin[j], (uint32_t)i);
        }
    }
}

// exclusive prefix sum of v over a 256-thread workgroup and the grand total (one barrier; buf: 4 words of LDS, not reused by the caller)
__device__ __forceinline__ uint32_t wg_exscan256(uint32_t v, volatile uint32_t* buf, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t x = wave_incl_scan(v);
    if (lane == 63) buf[wave] = x;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t s = buf[w];
        off += (w < wave) ? s : 0u;
        tot += s;
    }
    total = tot;
    return off + x - v;
}

// sum of v over the 256-thread workgroup (every thread gets it); buf: 4 words of LDS
__device__ __forceinline__ uint32_t wg_sum256(uint32_t v, volatile uint32_t* buf) {
    uint32_t x = v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = x;
    __syncthreads();
    const uint32_t t = buf[0] + buf[1] + buf[2] + buf[3];
    __syncthreads();
    return t;
}

static __global__ __launch_bounds__(FIN_T) void k_fin_sum(const FinArgs A) {
    __shared__ uint32_t buf[4];
    if (*A.err) return;
    const int b = blockIdx.y, c = blockIdx.x, N = A.N;
    const uint32_t* cnt = fin_cnt(A, b);
    uint32_t s = 0;
    for (int e = c * FIN_CHUNK + threadIdx.x; e < N && e < (c + 1) * FIN_CHUNK; e += FIN_T) s += cnt[e];
    const uint32_t t = wg_sum256(s, buf);
    if (threadIdx.x == 0) A.partial[(size_t)b * A.nchunk + c] = t;
}

// start[e] = number of hits on positions below e (exclusive scan of cnt): the chunk sums before this chunk, then the chunk itself
static __global__ __launch_bounds__(FIN_T) void k_fin_scan(const FinArgs A) {
    __shared__ uint32_t buf[4];
    __shared__ uint32_t sb[4];
    if (*A.err) return;
    const int b = blockIdx.y, c = blockIdx.x, N = A.N, tid = threadIdx.x;
    const uint32_t* cnt = fin_cnt(A, b);
    uint32_t* start = fin_start(A, b);
    uint32_t pre = 0;
    for (int k = tid; k < c; k += FIN_T) pre += A.partial[(size_t)b * A.nchunk + k];
    uint32_t run = wg_sum256(pre, buf);
    // 16 consecutive elements per thread
    const int e0 = c * FIN_CHUNK + tid * (FIN_CHUNK / FIN_T);
    uint32_t v[FIN_CHUNK / FIN_T];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < FIN_CHUNK / FIN_T; ++k) {
        v[k] = e0 + k < N ? cnt[e0 + k] : 0u;
        s += v[k];
    }
    uint32_t total;
    run += wg_exscan256(s, sb, total);
#pragma unroll
    for (int k = 0; k < FIN_CHUNK / FIN_T; ++k) {
        if (e0 + k < N) start[e0 + k] = run;
        run += v[k];
    }
}

static __global__ __launch_bounds__(FIN_T) void k_fin_bucket(const FinArgs A) {
    if (*A.err) return;
    const int b = blockIdx.y, N = A.N;
    const uint32_t* J = A.J + (size_t)b * N;
    const uint32_t *slot = fin_slot(A, b), *start = fin_start(A, b);
    uint32_t* bucket = fin_bucket(A, b);
    for (int i = blockIdx.x * FIN_CHUNK + threadIdx.x; i < N && i < (int)(blockIdx.x + 1) * FIN_CHUNK; i += FIN_T) {
        if (i == 0) continue;
        const uint32_t j = J[i];
        if (j < (uint32_t)i) bucket[start[j] + slot[i]] = (uint32_t)i;
    }
}

// labels after the shuffle (into `slot`, dead by now) and how many of each label the chunk holds
static __global__ __launch_bounds__(FIN_T) void k_fin_label(const FinArgs A) {
    __shared__ uint32_t h[FIN_MAX_S];
    if (*A.err) return;
    const int b = blockIdx.y, c = blockIdx.x, N = A.N, S = A.S, tid = threadIdx.x;
    const uint32_t* J = A.J + (size_t)b * N;
    const uint32_t *cnt = fin_cnt(A, b), *start = fin_start(A, b), *bucket = fin_bucket(A, b), *bmin = fin_bmin(A, b);
    uint32_t* label = fin_slot(A, b);
    if (tid < FIN_MAX_S) h[tid] = 0;
    __syncthreads();
    for (int i = c * FIN_CHUNK + tid; i < N && i < (c + 1) * FIN_CHUNK; i += FIN_T) {
        uint32_t q;
        if (A.randomize) {
            const uint32_t j = i == 0 ? 0u : J[i];
            uint32_t cand = FIN_INF;
            const uint32_t s0 = start[j], n0 = cnt[j];
            for (uint32_t e = 0; e < n0; ++e) {
                const uint32_t hh = bucket[s0 + e];
                if (hh > (uint32_t)i && hh < cand) cand = hh;
            }
            q = j;
            if (cand != FIN_INF) {
                q = cand;
                for (;;) {
                    const uint32_t hh = bmin[q];
                    if (hh == FIN_INF) break;
                    q = hh;
                }
            }
        } else {
            q = (uint32_t)i;
        }
        const uint32_t lab = q % (uint32_t)S;
        label[i] = lab;
        atomicAdd(&h[lab], 1u);
    }
    __syncthreads();
    if (tid < S) A.hist[((size_t)b * A.nchunk + c) * S + tid] = h[tid];
}

// red_blue.py:85 boolean-mask order: ascending walker index inside each set, sets in label order
static __global__ __launch_bounds__(FIN_T) void k_fin_order(const FinArgs A) {
    __shared__ uint32_t run[FIN_MAX_S];                // next free place of every label's set, in plan order
    __shared__ uint32_t wcount[4][FIN_MAX_S];
    __shared__ uint32_t buf[4];
    if (*A.err) return;
    const int b = blockIdx.y, c = blockIdx.x, N = A.N, S = A.S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t* label = fin_slot(A, b);
    int32_t* order = A.order[b];
    for (int s = 0; s < S; ++s) {
        uint32_t pre = 0;
        for (int k = tid; k < c; k += FIN_T) pre += A.hist[((size_t)b * A.nchunk + k) * S + s];
        const uint32_t t = wg_sum256(pre, buf);
        if (tid == 0) {
            const int q = N / S, rem = N % S;
            run[s] = (uint32_t)(s * q + (s < rem ? s : rem)) + t;
        }
    }
    __syncthreads();
    for (int r0 = c * FIN_CHUNK; r0 < N && r0 < (c + 1) * FIN_CHUNK; r0 += FIN_T) {
        const int e = r0 + tid;
        const bool live = e < N;
        const uint32_t lab = live ? label[e] : 0xffffffffu;
        if (tid < FIN_MAX_S) {
            wcount[0][tid] = 0;
            wcount[1][tid] = 0;
            wcount[2][tid] = 0;
            wcount[3][tid] = 0;
        }
        __syncthreads();
        // rank inside the wave among the lanes with the same label
        uint32_t rank = 0;
        {
            unsigned long long todo = __ballot(live);
            while (todo) {
                const int leader = __builtin_ctzll(todo);
                const uint32_t s = (uint32_t)__shfl((int)lab, leader, 64);
                const unsigned long long same = __ballot(live && lab == s);
                if (live && lab == s) rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                if (lane == leader) wcount[wave][s] = (uint32_t)__popcll(same);
                todo &= ~same;
            }
        }
        __syncthreads();
        if (live) {
            uint32_t off = run[lab];
            for (int w = 0; w < wave; ++w) off += wcount[w][lab];
            order[off + rank] = e;
        }
        __syncthreads();
        if (tid < S) run[tid] += wcount[0][tid] + wcount[1][tid] + wcount[2][tid] + wcount[3][tid];
        __syncthreads();
    }
}

// stretch.py:30-33 and red_blue.py:100: partners resolved against the complement (the sets before the split, then after),
// zz = ((a - 1) u + 1)^2 / a, the accept uniforms; logs as k_plan_logs takes them
static __global__ __launch_bounds__(FIN_T) void k_fin_plan(const FinArgs A) {
    if (*A.err) return;
    const int b = blockIdx.y, N = A.N, S = A.S;
    const int t = blockIdx.x * FIN_T + threadIdx.x;
    if (t >= N) return;
    const int q = N / S, rem = N % S;
    const int s = t < rem * (q + 1) ? t / (q + 1) : rem + (t - rem * (q + 1)) / q;
    const int ns = q + (s < rem ? 1 : 0), off = s * q + (s < rem ? s : rem), tl = t - off;
    const uint32_t nc = (uint32_t)(N - ns), rng = nc - 1u;
    const unsigned long long* tp = A.tokpos + ((size_t)b * S + s) * 3;
    const unsigned long long pz = tp[0], pr = tp[1], pu = tp[2];
    const int32_t* order = A.order[b];
    uint32_t r;
    if (rng == 0u)
        r = 0u;
    else if ((nc & rng) == 0u)
        r = A.stream[(pr + (unsigned long long)tl) & A.smask] & rng;
    else
        r = A.rint[(size_t)b * N + t];
    A.p0[b][t] = (int)r < off ? order[r] : order[r + ns];
    const double a = A.a, dm1 = (double)A.D - 1.0;
    {
        const uint32_t w0 = A.stream[(pz + 2ull * tl) & A.smask], w1 = A.stream[(pz + 2ull * tl + 1ull) & A.smask];
        const double u = ((double)(int)(w0 >> 5) * 67108864.0 + (double)(int)(w1 >> 6)) / 9007199254740992.0;
        const double tt = (a - 1.0) * u + 1.0;
        const double zz = tt * tt / a;
        A.s0[b][t] = zz;
        A.fac[b][t] = dm1 * plan_log(zz);
    }
    {
        const uint32_t w0 = A.stream[(pu + 2ull * tl) & A.smask], w1 = A.stream[(pu + 2ull * tl + 1ull) & A.smask];
        const double u = ((double)(int)(w0 >> 5) * 67108864.0 + (double)(int)(w1 >> 6)) / 9007199254740992.0;
        A.uacc[b][t] = u;
        A.logu[b][t] = plan_log(u);
    }
}

}  // namespace mtdev
}  // namespace emx
