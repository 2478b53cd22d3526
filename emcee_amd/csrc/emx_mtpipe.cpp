// Exact-mode plan pipeline (see emx_mtpipe.hpp).  Plain host C++: compiled with -ffp-contract=off so that the few
// floating-point expressions (zz, polar radius, gamma) round exactly like NumPy's separate operations.
#include "emx_mtpipe.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#include <sys/prctl.h>
#endif

namespace emx {
namespace {

#if defined(__x86_64__) && defined(__clang__)
#define EMX_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define EMX_CLONES
#endif

constexpr int BLK = 624;
constexpr uint64_t AHEAD_BLKS = 256;     // the generator runs at most this far ahead of the tokenizer: 160 k words (640 KB) stay cache resident between the two threads
constexpr uint64_t RING_BLKS = AHEAD_BLKS + 32;      // the ring: the lead + the bursts in flight
constexpr uint64_t GEN_BURST = 8;        // blocks per publication (generator_main)
constexpr int RING_MIRROR = 16;          // words [0, 16) of the ring repeated behind its end: a word pair or vector that straddles the wrap reads on
constexpr int MAX_SPLITS = 64;

inline uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The threads hand each other megabytes per step through the cache hierarchy (the word ring, the token and staging
// buffers).  Spread over the CCDs / sockets of a large host by the scheduler they run 6x slower than the calling thread
// alone (measured on a 2 x EPYC 9575F: 0.66 ms/step against 0.105 ms/step inside one L3 domain), so they are confined
// to the CPUs that share the last-level cache with the CPU the caller runs on (within the process's own affinity
// mask).  EMX_PIPE_AFFINITY=off disables it, EMX_PIPE_AFFINITY=<cpu list, e.g. 0-7,128-135> overrides it.
#if defined(__linux__)
struct CpuSet {
    cpu_set_t set;
    bool valid = false;
};

bool parse_cpu_list(const char* txt, cpu_set_t& out) {
    CPU_ZERO(&out);
    int count = 0;
    const char* p = txt;
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
        if (!*p) break;
        char* e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p) return false;
        long b = a;
        p = e;
        if (*p == '-') {
            b = strtol(p + 1, &e, 10);
            if (e == p + 1) return false;
            p = e;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            CPU_SET((int)c, &out);
            ++count;
        }
    }
    return count > 0;
}

CpuSet pipeline_cpus() {
    CpuSet r;
    const char* env = getenv("EMX_PIPE_AFFINITY");
    if (env && (!strcmp(env, "off") || !strcmp(env, "0") || !*env)) return r;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return r;
    cpu_set_t want;
    if (env) {
        if (!parse_cpu_list(env, want)) return r;
    } else {
        const int cpu = sched_getcpu();
        if (cpu < 0) return r;
        char path[128], buf[512];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        FILE* f = fopen(path, "r");
        if (!f) return r;
        const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        buf[n] = 0;
        if (!parse_cpu_list(buf, want)) return r;
    }
    CPU_AND(&r.set, &want, &allowed);
    r.valid = CPU_COUNT(&r.set) >= 2;        // a single CPU for five busy threads would be worse than no pinning
    return r;
}

// One logical CPU per physical core of the set (the lowest SMT sibling), the caller's own core last: generator,
// tokenizer and finishers each get a core of their own when there are enough -- two of them sharing a core through SMT
// halves both (the generator and the tokenizer are the critical pair).
std::vector<int> distinct_cores(const CpuSet& cs) {
    std::vector<int> reps;
    if (!cs.valid) return reps;
    const int self = sched_getcpu();
    int self_rep = -1;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &cs.set)) continue;
        char path[128], buf[256];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        int rep = c;
        if (FILE* f = fopen(path, "r")) {
            const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
            fclose(f);
            buf[n] = 0;
            cpu_set_t sib;
            if (parse_cpu_list(buf, sib)) {
                for (int k = 0; k < CPU_SETSIZE; ++k)
                    if (CPU_ISSET(k, &sib) && CPU_ISSET(k, &cs.set)) {
                        rep = k;
                        break;
                    }
                if (self >= 0 && CPU_ISSET(self, &sib)) self_rep = rep;
            }
        }
        if (std::find(reps.begin(), reps.end(), rep) == reps.end()) reps.push_back(rep);
    }
    if (self_rep >= 0) {                      // the caller keeps its core to itself as long as there are others
        auto it = std::find(reps.begin(), reps.end(), self_rep);
        if (it != reps.end()) {
            reps.erase(it);
            reps.push_back(self_rep);
        }
    }
    return reps;
}

void confine(std::thread& t, const CpuSet& cs, const std::vector<int>& cores, int slot) {
    if (!cs.valid) return;
    if ((int)cores.size() >= 3 && slot >= 0) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cores[(size_t)slot % cores.size()], &one);
        if (pthread_setaffinity_np(t.native_handle(), sizeof(one), &one) == 0) return;
    }
    pthread_setaffinity_np(t.native_handle(), sizeof(cs.set), &cs.set);
}
#else
struct CpuSet {
    bool valid = false;
};
CpuSet pipeline_cpus() { return CpuSet(); }
std::vector<int> distinct_cores(const CpuSet&) { return {}; }
void confine(std::thread&, const CpuSet&, const std::vector<int>&, int) {}
#endif

// Waiting for a neighbouring stage: spin on PAUSE for a while, then sleep in short naps.  The consumer may take its steps in bursts
// (sixteen per launch of the persistent kernels, 80-150 us apart): a stage that went to sleep in between woke 100+ us late (40 us + the
// default 50 us timer slack), three stages in a row 250 us -- more than a burst takes.  Hence: the spin covers the gap between two
// bursts, the naps are 20 us with the threads' timer slack set to 1 us (stage_thread_setup).  PAUSE, not sched_yield: on a host whose
// logical CPUs are SMT pairs (the 16-CPU boxes) six finishers yielding in a loop next to the generator and the tokenizer cost those a
// third of their rate (C2 exact mode 54 -> 87 us/step, profiles/r04/exact_mid.txt); PAUSE hands the core's issue slots to the sibling.
// The spin is for a consumer that takes its steps in bursts only (the constructor's bursty_consumer: the persistent kernels); with a
// consumer that takes a step at a time and orders its uploads with events, spinning stage threads made it SLOWER (26.8 -> 48.8 us/step
// at 4 096 walkers, 68 -> 84 at 65 536: profiles/r04/exact_mid.txt) -- there the window is 0.  EMX_PIPE_SPIN_US overrides both.
struct Backoff {
    int n = 0;
    uint64_t t0 = 0;
    uint64_t spin = 0;            // the owning pipeline's spin window (a field of the pipeline, not of the process: round 5)
    explicit Backoff(uint64_t spin_window_ns) : spin(spin_window_ns) {}
    uint64_t spin_ns() const { return spin; }
    inline void pause() {
        if (n < 256) {
            ++n;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            return;
        }
        const uint64_t spin = spin_ns();
        if (spin == 0 && n < 512) {          // (a consumer that takes a step at a time: yield a little, then 40 us naps, as in rounds 1-3)
            ++n;
            std::this_thread::yield();
            return;
        }
        const uint64_t now = now_ns();
        if (n == 256 || (spin == 0 && n == 512)) {
            ++n;
            t0 = now;
        }
        const uint64_t waited = now - t0;
        if (waited < spin) {
#if defined(__x86_64__)
            for (int k = 0; k < 32; ++k) __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        } else if (waited < 150000000ull) {
            std::this_thread::sleep_for(std::chrono::microseconds(spin ? 20 : 40));
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(500));      // a pipeline left idle between calls costs next to nothing
        }
    }
};
inline void stage_thread_setup(uint64_t spin_window_ns) {
#if defined(__linux__)
    if (spin_window_ns != 0) prctl(PR_SET_TIMERSLACK, 1000ul, 0, 0, 0);       // (ns; the default 50 us is added to every sleep_for above)
#endif
}

// ---- the MT19937 recurrence, out of place so that every loop is a plain vectorisable map -------------------------
EMX_CLONES void twist_block(const uint32_t* __restrict o, uint32_t* __restrict n) {
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
    for (int kk = 0; kk < 227; ++kk) {
        const uint32_t y = (o[kk] & UPPER) | (o[kk + 1] & LOWER);
        n[kk] = o[kk + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
    }
    for (int kk = 227; kk < 454; ++kk) {                     // reads n[0, 227): written above
        const uint32_t y = (o[kk] & UPPER) | (o[kk + 1] & LOWER);
        n[kk] = n[kk - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
    }
    for (int kk = 454; kk < 623; ++kk) {                     // reads n[227, 396)
        const uint32_t y = (o[kk] & UPPER) | (o[kk + 1] & LOWER);
        n[kk] = n[kk - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
    }
    const uint32_t y = (o[623] & UPPER) | (n[0] & LOWER);
    n[623] = n[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
}

// MT19937's output tempering.  The ring holds the generator's STATE words (round 5): the recurrence is the one serial thing in this
// file and tempering is half of the vector work of a block, so it moved to the consumers -- the tokenizer tempers the words it
// tests, the finishers (K threads) the words they convert.
inline uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

#if defined(__x86_64__) && defined(__clang__)
#include <immintrin.h>
#define EMX_HAVE_AVX512_GEN 1
__attribute__((target("avx512f"))) inline __m512i temper_v(__m512i r) {
    const __m512i TB = _mm512_set1_epi32((int)0x9d2c5680u), TC = _mm512_set1_epi32((int)0xefc60000u);
    __m512i t = _mm512_xor_si512(r, _mm512_srli_epi32(r, 11));
    t = _mm512_xor_si512(t, _mm512_and_si512(_mm512_slli_epi32(t, 7), TB));
    t = _mm512_xor_si512(t, _mm512_and_si512(_mm512_slli_epi32(t, 15), TC));
    return _mm512_xor_si512(t, _mm512_srli_epi32(t, 18));
}
// The whole block in registers: 39 vectors of 16 words.  Word kk of the new block needs new word kk - 227 (for
// kk >= 227), i.e. a vector that starts 13 words into new vector i - 15: instead of storing the new words and
// re-loading them unaligned (a load that straddles two just-issued stores cannot be forwarded and waits for both to
// retire -- the dependency that holds the auto-vectorised loops at 0.16 ns/word), the last 15 result vectors stay in
// registers and the straddling vector is made with one valignd.  The new state words go to `n` (the private buffer the next block is
// made from) and to `out` (64-byte aligned).  STREAM: with non-temporal stores -- for a ring too large to stay in the last-level
// cache, whose lines are cold whenever the generator comes round again and would first be READ from memory by a plain store (the
// caller fences before it publishes).  A ring that does stay cached takes plain stores: measured on the MI355X host, the
// non-temporal form held the generator at 53 us per step of 65 536 walkers (32 GB/s of write-combining), plain stores into a
// cache-resident ring at 37.
template <bool STREAM>
__attribute__((target("avx512f"))) void twist_avx512(const uint32_t* __restrict o, uint32_t* __restrict n, uint32_t* __restrict out) {
    const __m512i UPPER = _mm512_set1_epi32((int)0x80000000u), ONE = _mm512_set1_epi32(1), MATRIX = _mm512_set1_epi32((int)0x9908b0dfu);
    __m512i N[39];
    const __m512i Olast = _mm512_loadu_si512(o + 608);
#pragma unroll
    for (int i = 0; i < 39; ++i) {
        const __m512i cur = _mm512_loadu_si512(o + 16 * i);
        __m512i nxt;
        if (i < 38)
            nxt = _mm512_loadu_si512(o + 16 * i + 1);
        else
            nxt = _mm512_alignr_epi32(N[0], cur, 1);                      // word 623 pairs with the NEW word 0
        __m512i m;
        if (i <= 13)
            m = _mm512_loadu_si512(o + 16 * i + 397);                     // old words kk + 397
        else if (i == 14)
            m = _mm512_alignr_epi32(N[0], Olast, 13);                     // old 621..623, then new 0..12
        else
            m = _mm512_alignr_epi32(N[i - 14], N[i - 15], 13);            // new words kk - 227
        // y = (cur & UPPER) | (nxt & LOWER);  r = m ^ (y >> 1) ^ (y & 1 ? MATRIX : 0)   (y & 1 == nxt & 1)
        const __m512i y = _mm512_ternarylogic_epi32(UPPER, cur, nxt, 0xCA);    // UPPER ? cur : nxt, bitwise
        __m512i r = _mm512_xor_si512(m, _mm512_srli_epi32(y, 1));
        const __mmask16 odd = _mm512_test_epi32_mask(nxt, ONE);
        r = _mm512_mask_xor_epi32(r, odd, r, MATRIX);
        N[i] = r;
        _mm512_storeu_si512(n + 16 * i, r);
        if (STREAM) _mm512_stream_si512(reinterpret_cast<__m512i*>(out + 16 * i), r);
    }
    // (not streamed: the block is copied out in one piece.  Stores into the ring from inside the loop -- into lines the tokenizer's
    // core is reading -- held the generator at 78-87 us per step of 65 536 walkers against 36 this way: the loop's loads and
    // ALU work wait behind a store buffer full of ownership requests)
    if (!STREAM) std::memcpy(out, n, BLK * 4);
}
#endif
void twist_generic(const uint32_t* o, uint32_t* n, uint32_t* out) {
    twist_block(o, n);
    std::memcpy(out, n, BLK * 4);
}

// one generator step: the new state words from the old ones (out of place)
using TwistFn = void (*)(const uint32_t*, uint32_t*, uint32_t*);

// the register-resident AVX-512 version where the CPU has it and it reproduces the generic one on a test block
TwistFn pick_twist(bool stream) {
#ifdef EMX_HAVE_AVX512_GEN
    if (__builtin_cpu_supports("avx512f") && !getenv("EMX_PIPE_NO_AVX512")) {
        alignas(64) uint32_t o[BLK + 16], a[BLK + 16], b[BLK + 16];
        uint32_t x = 0x9e3779b9u;
        for (int i = 0; i < BLK + 16; ++i) {
            x ^= x << 13;
            x ^= x >> 17;
            x ^= x << 5;
            o[i] = x;
        }
        alignas(64) uint32_t oa[BLK + 16], ob[BLK + 16];
        twist_generic(o, a, oa);
        twist_avx512<true>(o, b, ob);
        _mm_sfence();
        if (!std::memcmp(a, b, BLK * 4) && !std::memcmp(oa, ob, BLK * 4)) return stream ? twist_avx512<true> : twist_avx512<false>;
    }
#endif
    return twist_generic;
}

// random_sample() of consecutive word pairs (w: state words, tempered here)
EMX_CLONES void convert_pairs(const uint32_t* __restrict w, double* __restrict dst, int64_t n) {
    for (int64_t e = 0; e < n; ++e) {
        const int32_t a = (int32_t)(temper(w[2 * e]) >> 5), b = (int32_t)(temper(w[2 * e + 1]) >> 6);
        dst[e] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
}

// stretch.py:30  zz = ((a - 1) * rand(Ns) + 1) ** 2 / a, straight from the word pairs
EMX_CLONES void convert_pairs_zz(const uint32_t* __restrict w, double* __restrict dst, int64_t n, double a) {
    for (int64_t e = 0; e < n; ++e) {
        const int32_t hi = (int32_t)(temper(w[2 * e]) >> 5), lo = (int32_t)(temper(w[2 * e + 1]) >> 6);
        const double u = (hi * 67108864.0 + lo) / 9007199254740992.0;
        const double tt = (a - 1.0) * u + 1.0;
        dst[e] = tt * tt / a;
    }
}

#ifdef EMX_HAVE_AVX512_GEN
// Fisher-Yates targets, 16 stream words at a time.  Word k of a vector is accepted iff v_k <= i - (accepted before it): a
// word <= i - 16 is accepted and a word > i rejected whatever the others do, and that is nearly every word (the rest,
// 16 values out of the mask range, send the vector to the scalar loop).  Accepted values are compressed in stream order
// into jr[(n - 1) - i ...]: the reversed layout makes their addresses ascend.  Returns the words consumed.
// (p: state words, tempered here.)
// Compaction of the accepted lanes of one vector to dst, in lane order.  Two forms: vpcompressd (one instruction, but tens of cycles
// on some cores) and two 8-lane table permutes (vpermd through a 256-entry table of lane lists).  pick_compaction() times both once.
struct CompactLut {
    alignas(32) uint32_t idx[256][8];
    CompactLut() {
        for (int m = 0; m < 256; ++m) {
            int k = 0;
            for (int b = 0; b < 8; ++b)
                if (m & (1 << b)) idx[m][k++] = (uint32_t)b;
            for (; k < 8; ++k) idx[m][k] = 0;
        }
    }
};
static const CompactLut g_compact_lut;
static std::atomic<int> g_compaction{-1};      // 0: vpcompressd, 1: table permutes, 2: none (tools/ubench/mt_scan_bench.cpp only: timing without it)
template <int MODE>
__attribute__((target("avx512f,avx512vl,avx512bw,avx2,popcnt"))) inline int compact_store(uint32_t* dst, __mmask16 a, __m512i v) {
    if (MODE == 0) {
        _mm512_storeu_si512(dst, _mm512_maskz_compress_epi32(a, v));
        return __builtin_popcount((unsigned)a);
    } else if (MODE == 1) {
        const unsigned ml = (unsigned)a & 0xffu, mh = ((unsigned)a >> 8) & 0xffu;
        const int cl = __builtin_popcount(ml);
        const __m256i lo8 = _mm512_castsi512_si256(v), hi8 = _mm512_extracti64x4_epi64(v, 1);
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst), _mm256_permutevar8x32_epi32(lo8, _mm256_load_si256(reinterpret_cast<const __m256i*>(g_compact_lut.idx[ml]))));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + cl), _mm256_permutevar8x32_epi32(hi8, _mm256_load_si256(reinterpret_cast<const __m256i*>(g_compact_lut.idx[mh]))));
        return cl + __builtin_popcount(mh);
    } else {
        return __builtin_popcount((unsigned)a);
    }
}

template <int MODE>
__attribute__((target("avx512f,avx512vl,avx512bw,avx2,popcnt"))) size_t shuffle_scan_avx512_t(const uint32_t* p, size_t navail, uint32_t mask, int64_t& i,
                                                                                              int64_t lo, uint32_t* jr, int64_t nm1) {
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    size_t used = 0;
    // Four vectors per trip while the band is wide (round 5), tested against thresholds made from the value i had one trip EARLIER:
    // the loop used to carry i through broadcast -> compare -> mask -> popcount -> subtract, some 15-20 cycles a trip whatever its
    // width.  With i' >= i the i of the trip before, a word > i' is rejected and a word <= i' - 128 accepted whatever the others do
    // (i' - 64 <= i at the start of this trip, at most 63 accepted before a word inside it); a word in between -- 128 values out of
    // the mask range -- sends the trip to the one-vector loop below.  The compares of a trip then depend on nothing the trip before
    // it computes, and consecutive trips overlap.
    if (mask >= (1u << 14)) {
        int64_t stale = i;
        while (navail - used >= 64 && i - 64 > lo) {
            const __m512i hi = _mm512_set1_epi32((int)(uint32_t)stale), lo128 = _mm512_set1_epi32((int)(uint32_t)(stale - 128));
            stale = i;
            const __m512i v0 = _mm512_and_si512(temper_v(_mm512_loadu_si512(p + used)), vmask);
            const __m512i v1 = _mm512_and_si512(temper_v(_mm512_loadu_si512(p + used + 16)), vmask);
            const __m512i v2 = _mm512_and_si512(temper_v(_mm512_loadu_si512(p + used + 32)), vmask);
            const __m512i v3 = _mm512_and_si512(temper_v(_mm512_loadu_si512(p + used + 48)), vmask);
            __mmask16 a0 = _mm512_cmple_epu32_mask(v0, lo128), a1 = _mm512_cmple_epu32_mask(v1, lo128),
                      a2 = _mm512_cmple_epu32_mask(v2, lo128), a3 = _mm512_cmple_epu32_mask(v3, lo128);
            const __mmask16 r0 = _mm512_cmpgt_epu32_mask(v0, hi), r1 = _mm512_cmpgt_epu32_mask(v1, hi), r2 = _mm512_cmpgt_epu32_mask(v2, hi),
                            r3 = _mm512_cmpgt_epu32_mask(v3, hi);
            if ((__mmask16)((a0 | r0) & (a1 | r1) & (a2 | r2) & (a3 | r3)) != (__mmask16)0xffff) {
                // some lane lies between the thresholds (one trip in eight at the top of the range): decided exactly here, in stream
                // order -- the lanes before it in its vector are all certain -- unless a vector has two of them (the loop below)
                unsigned am[4] = {(unsigned)a0, (unsigned)a1, (unsigned)a2, (unsigned)a3};
                const unsigned un[4] = {(unsigned)(__mmask16)~(a0 | r0), (unsigned)(__mmask16)~(a1 | r1), (unsigned)(__mmask16)~(a2 | r2), (unsigned)(__mmask16)~(a3 | r3)};
                alignas(64) uint32_t lanes[64];
                _mm512_store_si512(lanes, v0);
                _mm512_store_si512(lanes + 16, v1);
                _mm512_store_si512(lanes + 32, v2);
                _mm512_store_si512(lanes + 48, v3);
                int64_t ic = i;
                bool two = false;
                for (int k = 0; k < 4; ++k) {
                    if (un[k]) {
                        if (un[k] & (un[k] - 1u)) {
                            two = true;
                            break;
                        }
                        const int l = __builtin_ctz(un[k]);
                        const int before = __builtin_popcount(am[k] & ((1u << l) - 1u));
                        if ((int64_t)lanes[16 * k + l] <= ic - before) am[k] |= 1u << l;
                    }
                    ic -= __builtin_popcount(am[k]);
                }
                if (two) break;
                a0 = (__mmask16)am[0];
                a1 = (__mmask16)am[1];
                a2 = (__mmask16)am[2];
                a3 = (__mmask16)am[3];
            }
            uint32_t* dst = jr + (nm1 - i);                   // (16 slots of slack behind jr: a later store overwrites the tail of the one before it)
            const int c0 = compact_store<MODE>(dst, a0, v0);
            const int c1 = compact_store<MODE>(dst + c0, a1, v1);
            const int c2 = compact_store<MODE>(dst + c0 + c1, a2, v2);
            const int c3 = compact_store<MODE>(dst + c0 + c1 + c2, a3, v3);
            i -= (int64_t)(c0 + c1 + c2 + c3);
            used += 64;
        }
    }
    // The same idea one vector per trip for the narrower bands (ensembles of a few thousand walkers never reach the loop above):
    // thresholds from the i of the trip before -- a word > i' is rejected, a word <= i' - 32 accepted (i' - 16 <= i at the start of
    // this trip, at most 15 accepted in front of a word inside it), one word in between is decided in place, two leave the loop.
    if (mask >= (1u << 9)) {
        int64_t stale = i;
        while (navail - used >= 16 && i - 16 > lo) {
            const __m512i hi = _mm512_set1_epi32((int)(uint32_t)stale), lo32 = _mm512_set1_epi32((int)(uint32_t)(stale - 32));
            stale = i;
            const __m512i v = _mm512_and_si512(temper_v(_mm512_loadu_si512(p + used)), vmask);
            __mmask16 acc = _mm512_cmple_epu32_mask(v, lo32);
            const __mmask16 rej = _mm512_cmpgt_epu32_mask(v, hi);
            if ((__mmask16)(acc | rej) != (__mmask16)0xffff) {
                const unsigned un = (unsigned)(__mmask16)~(acc | rej);
                if (un & (un - 1u)) break;
                alignas(64) uint32_t lanes[16];
                _mm512_store_si512(lanes, v);
                const int l = __builtin_ctz(un);
                if ((int64_t)lanes[l] <= i - __builtin_popcount((unsigned)acc & ((1u << l) - 1u))) acc = (__mmask16)((unsigned)acc | (1u << l));
            }
            i -= (int64_t)compact_store<MODE>(jr + (nm1 - i), acc, v);
            used += 16;
        }
    }
    while (navail - used >= 16 && i - 16 > lo) {
        const __m512i v = _mm512_and_si512(temper_v(_mm512_loadu_si512(p + used)), vmask);
        __mmask16 acc = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)(i - 16)));
        const __mmask16 rej = _mm512_cmpgt_epu32_mask(v, _mm512_set1_epi32((int)(uint32_t)i));
        if ((__mmask16)(acc | rej) != (__mmask16)0xffff) {                  // a word in (i - 16, i]: order matters
            const unsigned un = (unsigned)(__mmask16)~(acc | rej);
            if (un & (un - 1u)) break;                                      // two of them: scalar
            alignas(64) uint32_t lanes[16];
            _mm512_store_si512(lanes, v);
            const int l = __builtin_ctz(un);
            if ((int64_t)lanes[l] <= i - __builtin_popcount((unsigned)acc & ((1u << l) - 1u))) acc = (__mmask16)((unsigned)acc | (1u << l));
        }
        i -= (int64_t)compact_store<MODE>(jr + (nm1 - i), acc, v);          // 16 slots of slack behind jr
        used += 16;
    }
    return used;
}
// which compaction this CPU does faster: both forms over the same 64 K pseudo-random words, once per process
int pick_compaction() {
    // (contexts may be created concurrently: an atomic, and a measurement taken twice gives the same kind of answer)
    const int known = g_compaction.load(std::memory_order_relaxed);
    if (known >= 0) return known;
    if (const char* e = getenv("EMX_PIPE_COMPACTION")) {
        const int v = atoi(e) ? 1 : 0;
        g_compaction.store(v, std::memory_order_relaxed);
        return v;
    }
    std::vector<uint32_t> w(65536 + 64), out(65536 + 64);
    uint32_t x = 0x2545f491u;
    for (auto& v : w) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        v = x;
    }
    uint64_t best[2] = {~0ull, ~0ull};
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            int64_t i = 60000;
            const uint64_t t0 = now_ns();
            if (mode == 0)
                shuffle_scan_avx512_t<0>(w.data(), 40000, 65535u, i, 32767, out.data(), 65535);
            else
                shuffle_scan_avx512_t<1>(w.data(), 40000, 65535u, i, 32767, out.data(), 65535);
            best[mode] = std::min<uint64_t>(best[mode], now_ns() - t0);
        }
    const int pick = best[1] < best[0] ? 1 : 0;
    g_compaction.store(pick, std::memory_order_relaxed);
    return pick;
}
// randint(0, rng + 1) values by masked rejection, 16 stream words at a time (the acceptance test does not depend on the position:
// plain order-preserving compaction); stops in front of the vector that could overshoot n.  Returns the words consumed.
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt"))) size_t randint_scan_avx512(const uint32_t* p, size_t navail, uint32_t mask, uint32_t rng,
                                                                                      int32_t* dst, int64_t& k, int64_t n) {
    const __m512i vmask = _mm512_set1_epi32((int)mask), vr = _mm512_set1_epi32((int)rng);
    size_t used = 0;
    while (navail - used >= 16 && k + 16 <= n) {
        const __m512i v = _mm512_and_si512(temper_v(_mm512_loadu_si512(p + used)), vmask);
        const __mmask16 acc = _mm512_cmple_epu32_mask(v, vr);
        const int c = __builtin_popcount((unsigned)acc);
        _mm512_mask_storeu_epi32(dst + k, (__mmask16)((1u << c) - 1u), _mm512_maskz_compress_epi32(acc, v));
        k += c;
        used += 16;
    }
    return used;
}
// legacy_gauss candidates (RandomState.randn, de.py:56): x1 = 2 double - 1, x2 = 2 double - 1, r2 = x1 x1 + x2 x2, kept when
// 0 < r2 < 1 -- four stream words a candidate whatever its fate, so four candidates a vector and an order-preserving compaction.
// An accepted candidate gives two normals, f x2 first and the cached f x1 behind it (f = sqrt(-2 ln r2 / r2), made by the
// finishers): the tokens (x2, x1) go to gx[t], gx[t + 1], r2 to both gr2.  Every operation is the scalar loop's, rounded alike
// (products and sum separately; 2 d - 1 is exact either way).  Stops in front of the vector that could pass n entries.
__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq,popcnt"))) size_t polar_scan_avx512(const uint32_t* p, size_t navail, double* gx, double* gr2,
                                                                                             int64_t& t, int64_t n) {
    const __m512i lo32 = _mm512_set1_epi64(0xffffffffll);
    const __m512d sc = _mm512_set1_pd(0x1p-52), one = _mm512_set1_pd(1.0), zero = _mm512_setzero_pd();
    size_t used = 0;
    while (navail - used >= 16 && t + 8 <= n) {
        const __m512i q = temper_v(_mm512_loadu_si512(p + used));              // qword k = word 2 k | word 2 k + 1 << 32: one double
        const __m512i a = _mm512_srli_epi64(_mm512_and_si512(q, lo32), 5), b = _mm512_srli_epi64(q, 38);
        const __m512d d = _mm512_cvtepi64_pd(_mm512_or_si512(_mm512_slli_epi64(a, 26), b));      // (a 2^26 + b) < 2^53: exact
        const __m512d x = _mm512_sub_pd(_mm512_mul_pd(d, sc), one);            // lanes (x1, x2) of candidate 0, 1, 2, 3
        const __m512d sq = _mm512_mul_pd(x, x);
        const __m512d r2 = _mm512_add_pd(sq, _mm512_permute_pd(sq, 0x55));     // both lanes of a pair: x1 x1 + x2 x2
        const __mmask8 m = (__mmask8)(_mm512_cmp_pd_mask(r2, one, _CMP_LT_OQ) & _mm512_cmp_pd_mask(r2, zero, _CMP_NEQ_OQ));
        _mm512_storeu_pd(gx + t, _mm512_maskz_compress_pd(m, _mm512_permute_pd(x, 0x55)));      // (x2, x1); t + 8 <= n: room for all
        _mm512_storeu_pd(gr2 + t, _mm512_maskz_compress_pd(m, r2));
        t += __builtin_popcount((unsigned)m);
        used += 16;
    }
    return used;
}
// The snooker move's draws (de_snooker.py:37-40), per walker: randint(n0), randint(n1), randint(n2), then shuffle(w) =
// random_interval(2), random_interval(1): five masked-rejection draws whose ranges come round in a fixed order.  A window of 64
// tempered words and one 64-bit acceptance mask per range; a walker's draws are the first set bits of the masks taken in turn, each
// behind the one before (a dozen walkers a window: the vector work -- tempering, sixteen compares -- is shared among them, the rest
// is five shift / and / count-trailing-zeros steps a walker).  A walker whose draws do not end inside the window starts the next
// one; one that does not fit a fresh window is left to the scalar loop.  r / m: the three ranges (n - 1) and their masks.
__attribute__((target("avx512f,avx512vl,avx512bw,popcnt,bmi"))) size_t snooker_scan_avx512(const uint32_t* p, size_t navail, const uint32_t* r, const uint32_t* m,
                                                                                         int32_t* p0, int32_t* p1, int32_t* p2, uint8_t* perm, int64_t& t,
                                                                                         int64_t n) {
    const __m512i m0 = _mm512_set1_epi32((int)m[0]), m1 = _mm512_set1_epi32((int)m[1]), m2 = _mm512_set1_epi32((int)m[2]), m3 = _mm512_set1_epi32(3);
    const __m512i r0 = _mm512_set1_epi32((int)r[0]), r1 = _mm512_set1_epi32((int)r[1]), r2 = _mm512_set1_epi32((int)r[2]), r3 = _mm512_set1_epi32(2);
    alignas(64) uint32_t tw[64];
    const bool pow2 = r[0] == m[0] && r[1] == m[1] && r[2] == m[2];      // set sizes that are powers of two: their masks reject nothing
    size_t used = 0;
    while (t < n && navail - used >= 64) {
        uint64_t A0 = 0, A1 = 0, A2 = 0, A3 = 0;
        for (int q = 0; q < 4; ++q) {
            const __m512i v = temper_v(_mm512_loadu_si512(p + used + 16 * q));
            _mm512_store_si512(tw + 16 * q, v);
            A0 |= (uint64_t)_mm512_cmple_epu32_mask(_mm512_and_si512(v, m0), r0) << (16 * q);
            A1 |= (uint64_t)_mm512_cmple_epu32_mask(_mm512_and_si512(v, m1), r1) << (16 * q);
            A2 |= (uint64_t)_mm512_cmple_epu32_mask(_mm512_and_si512(v, m2), r2) << (16 * q);
            A3 |= (uint64_t)_mm512_cmple_epu32_mask(_mm512_and_si512(v, m3), r3) << (16 * q);
        }
        unsigned o = 0;                           // words of the window already consumed (a walker needs five at least)
        if (pow2) {
            // nothing but random_interval(2) rejects: a walker's first three words are its partners, one dependent
            // shift - count - add per walker instead of four
            while (t < n && o <= 59) {
                const uint64_t a = A3 >> (o + 3);
                if (!a) break;
                const unsigned i3 = o + 3 + (unsigned)__builtin_ctzll(a), i4 = i3 + 1;
                if (i4 > 63) break;
                p0[t] = (int32_t)(tw[o] & m[0]);
                p1[t] = (int32_t)(tw[o + 1] & m[1]);
                p2[t] = (int32_t)(tw[o + 2] & m[2]);
                perm[t] = (uint8_t)((tw[i3] & 3u) | ((tw[i4] & 1u) << 2));
                ++t;
                o = i4 + 1;
            }
        }
        while (!pow2 && t < n && o <= 59) {
            uint64_t a = A0 & (~0ull << o);
            if (!a) break;
            const unsigned i0 = (unsigned)__builtin_ctzll(a);
            if (i0 > 59) break;
            a = A1 & (~0ull << (i0 + 1));
            if (!a) break;
            const unsigned i1 = (unsigned)__builtin_ctzll(a);
            if (i1 > 60) break;
            a = A2 & (~0ull << (i1 + 1));
            if (!a) break;
            const unsigned i2 = (unsigned)__builtin_ctzll(a);
            if (i2 > 61) break;
            a = A3 & (~0ull << (i2 + 1));
            if (!a) break;
            const unsigned i3 = (unsigned)__builtin_ctzll(a), i4 = i3 + 1;    // random_interval(1): mask 1 rejects nothing
            if (i4 > 63) break;
            p0[t] = (int32_t)(tw[i0] & m[0]);
            p1[t] = (int32_t)(tw[i1] & m[1]);
            p2[t] = (int32_t)(tw[i2] & m[2]);
            perm[t] = (uint8_t)((tw[i3] & 3u) | ((tw[i4] & 1u) << 2));
            ++t;
            o = i4 + 1;
        }
        if (o == 0) break;                        // not even one walker fits this window
        used += o;
    }
    return used;
}
// The DE move's pair codes decoded eight at a time (de.py:44-52, de_pair in emx_mtpipe.hpp): code k of [0, nc (nc - 1)) is the k-th
// ordered pair of distinct complement members -- k < T = nc (nc - 1) / 2: (i, j) with j < i the kk-th pair of the lower triangle, else
// (j, i) for kk = k - T -- and the two members go through the complement's order (stretch.py:27 comp).  i is the one integer with
// i (i - 1) / 2 <= kk < (i + 1) i / 2: a square root and one correction either way find it, the inequality is checked, and a vector
// with a lane that fails it (none seen) is left to the scalar decoder -- so the result is de_pair's whatever the root's rounding.
// codes32: the codes in p0's own slots (32-bit draws); else codes64.  Returns the entries done (a multiple of 8).
__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq"))) int64_t de_decode_avx512(const uint64_t* codes64, bool codes32, int64_t n, uint64_t nc, int64_t base,
                                                                                     int64_t ns, const int32_t* order, int32_t* p0, int32_t* p1) {
    const __m512i T = _mm512_set1_epi64((long long)(nc * (nc - 1) / 2)), one = _mm512_set1_epi64(1);
    const __m512i vbase = _mm512_set1_epi64((long long)base), vns = _mm512_set1_epi64((long long)ns);
    const __m512d d1 = _mm512_set1_pd(1.0), d8 = _mm512_set1_pd(8.0), dh = _mm512_set1_pd(0.5);
#define EMX_TRI(i_) _mm512_srli_epi64(_mm512_mullo_epi64((i_), _mm512_sub_epi64((i_), one)), 1)      /* i (i - 1) / 2 */
    int64_t t = 0;
    for (; t + 8 <= n; t += 8) {
        const __m512i k = codes32 ? _mm512_cvtepu32_epi64(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p0 + t))) : _mm512_loadu_si512(codes64 + t);
        const __mmask8 lower = _mm512_cmplt_epu64_mask(k, T);
        const __m512i kk = _mm512_mask_sub_epi64(k, (__mmask8)~lower, k, T);
        const __m512d x = _mm512_mul_pd(_mm512_add_pd(d1, _mm512_sqrt_pd(_mm512_add_pd(d1, _mm512_mul_pd(d8, _mm512_cvtepu64_pd(kk))))), dh);
        __m512i i = _mm512_cvttpd_epu64(x);
        i = _mm512_mask_sub_epi64(i, _mm512_cmpgt_epu64_mask(EMX_TRI(i), kk), i, one);
        const __m512i up = _mm512_add_epi64(i, one);
        i = _mm512_mask_mov_epi64(i, _mm512_cmple_epu64_mask(EMX_TRI(up), kk), up);
        const __m512i lo = EMX_TRI(i);
        if ((__mmask8)(_mm512_cmple_epu64_mask(lo, kk) & _mm512_cmpgt_epu64_mask(EMX_TRI(_mm512_add_epi64(i, one)), kk)) != (__mmask8)0xff) break;
        const __m512i j = _mm512_sub_epi64(kk, lo);
        const __m512i f = _mm512_mask_blend_epi64(lower, j, i), sd = _mm512_mask_blend_epi64(lower, i, j);
        const __m512i fi = _mm512_mask_add_epi64(f, _mm512_cmpge_epu64_mask(f, vbase), f, vns);
        const __m512i si = _mm512_mask_add_epi64(sd, _mm512_cmpge_epu64_mask(sd, vbase), sd, vns);
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(p0 + t), _mm512_i64gather_epi32(fi, order, 4));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(p1 + t), _mm512_i64gather_epi32(si, order, 4));
    }
    return t;
#undef EMX_TRI
}
inline size_t shuffle_scan_avx512(const uint32_t* p, size_t navail, uint32_t mask, int64_t& i, int64_t lo, uint32_t* jr, int64_t nm1) {
    switch (g_compaction.load(std::memory_order_relaxed)) {
        case 1: return shuffle_scan_avx512_t<1>(p, navail, mask, i, lo, jr, nm1);
        case 2: return shuffle_scan_avx512_t<2>(p, navail, mask, i, lo, jr, nm1);
        default: return shuffle_scan_avx512_t<0>(p, navail, mask, i, lo, jr, nm1);
    }
}
#endif

// The ring of state words (round 5: consumers temper): small enough to stay in the generator's and the tokenizer's caches.
// (Measured and dropped, round 5: a ring of several steps that the finishers read the fixed-length draws from directly, so that
// the tokenizer need not copy them.  Plain stores into such a ring while six other cores read it held the generator at 90 us per
// step of 65 536 walkers, non-temporal stores at 53 -- against 36 into this one, which only the tokenizer reads;
// profiles/r05/exact_c2.md.)
struct WordStream {
    uint32_t* ring = nullptr;        // nblk * BLK + RING_MIRROR state words
    uint64_t nblk = 0;
    std::vector<uint32_t> own;       // (the ring's memory when the caller gave none)
    alignas(64) std::atomic<uint64_t> produced{0};     // blocks produced
    alignas(64) std::atomic<uint64_t> keep{0};         // lowest block anybody may still read (the oldest step not retired)
    std::atomic<uint64_t> rdblk{0};                    // block the tokenizer stands in
    alignas(64) std::atomic<bool> stop{false};
    std::atomic<uint64_t> gen_wait_ns{0}, gen_blocks{0}, rd_wait_ns{0};     // read by stage_times() while the threads run
    uint64_t spin_ns = 0;            // Backoff's spin window for this pipeline's threads
    uint64_t words() const { return nblk * BLK; }
};

void generator_main(WordStream* ws, const uint32_t* start_key) {
    stage_thread_setup(ws->spin_ns);
    // block 0 is the block the caller's generator currently stands in (no twist)
    const uint64_t NBLK = ws->nblk;
    const TwistFn twist = pick_twist(NBLK * BLK * 4 > (24ull << 20));
    std::memcpy(&ws->ring[0], start_key, BLK * 4);
    std::memcpy(&ws->ring[NBLK * BLK], start_key, RING_MIRROR * 4);
    ws->produced.store(1, std::memory_order_release);
    // The recurrence runs in a private double buffer and every block goes out to the ring as well: making block b from block b - 1
    // IN the ring -- one store per vector -- was 1.6x SLOWER (124 against 75 ms per 400 steps of 65 536 walkers on the build host):
    // the lines of block b - 1 are being pulled by the tokenizer's core just then.
    alignas(64) uint32_t pkey[2][BLK + 16];
    std::memcpy(pkey[0], start_key, BLK * 4);
    Backoff bo(ws->spin_ns);
    // A block is 50-150 ns of work; the shared words (`keep`, the reader's block, this thread's `produced`) each cost a cache-line
    // transfer between cores when touched, so they are touched once per GEN_BURST blocks: space for a burst is checked once, the
    // burst is published once.  (Round 5; before, every block loaded `keep`, published `produced` and wrote a sequentially
    // consistent statistics word -- about as long as the twist itself.)
    uint64_t keep_seen = ws->keep.load(std::memory_order_acquire), rd_seen = ws->rdblk.load(std::memory_order_acquire);
    for (uint64_t b = 1; !ws->stop.load(std::memory_order_relaxed);) {
        const auto blocked = [&] { return b + GEN_BURST - keep_seen >= NBLK || (b > rd_seen && b - rd_seen > AHEAD_BLKS); };
        if (blocked()) {
            keep_seen = ws->keep.load(std::memory_order_acquire);
            rd_seen = ws->rdblk.load(std::memory_order_acquire);
            if (blocked()) {
                const uint64_t t0 = now_ns();
                bo.pause();
                ws->gen_wait_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
                continue;
            }
        }
        bo.n = 0;
        for (uint64_t e = b + GEN_BURST; b < e; ++b) {
            twist(pkey[(b - 1) & 1], pkey[b & 1], &ws->ring[(b % NBLK) * BLK]);
            if ((b % NBLK) == 0) std::memcpy(&ws->ring[NBLK * BLK], pkey[b & 1], RING_MIRROR * 4);
        }
#if defined(__x86_64__)
        __builtin_ia32_sfence();                           // the non-temporal stores above are ordered before the publication below
#endif
        ws->gen_blocks.store(b - 1, std::memory_order_relaxed);
        ws->produced.store(b, std::memory_order_release);
    }
}

// the tokenizer's view of the stream
struct Reader {
    WordStream* ws;
    const uint32_t *base = nullptr, *cur = nullptr, *end = nullptr;
    uint64_t a_base = 0;          // absolute index of *base
    const std::atomic<bool>* stop;
    bool dead = false;
    bool vec_ok = false;          // the AVX-512 scans may be used (MtPlanPipeline::Impl::vec_scan)
    bool vec_dq = false;          // ... and the one that needs AVX-512 DQ (polar_scan_avx512: 64-bit integer -> double)

    uint64_t pos() const { return a_base + (uint64_t)(cur - base); }

    void seek(uint64_t a) {
        a_base = a;
        base = cur = end = nullptr;
    }

    // what the generator may overwrite and how far it may run ahead
    void publish(uint64_t a) {
        ws->keep.store(a > 0 ? (a - 1) / BLK : 0, std::memory_order_release);
        ws->rdblk.store(a / BLK, std::memory_order_release);
    }

    // wait until the words below position `a` exist
    bool wait_produced(uint64_t a) {
        if (ws->produced.load(std::memory_order_acquire) * BLK >= a) return true;
        Backoff bo(ws->spin_ns);
        const uint64_t t0 = now_ns();
        // The pipeline fails loudly instead of hanging when the generator can make no progress at all (a ring that cannot take this
        // step: step_words_bound exceeded, which does not happen).  "No progress" is asked of the GENERATOR, not of the wall clock
        // (round-5 advisor: a process stopped under a debugger, a stalled VM or an oversubscribed host turned ten seconds of wall
        // clock into a failed run): the limit -- EMX_PIPE_STALL_MS, default 10 s -- restarts whenever a block appears, and a
        // sample taken later than a second after the one before it (this thread was not running either) restarts it as well.
        static const uint64_t stall_ns = [] {
            const char* e = getenv("EMX_PIPE_STALL_MS");
            const long long ms = e ? atoll(e) : 10000;
            return (uint64_t)(ms > 0 ? ms : 10000) * 1000000ull;
        }();
        uint64_t seen = ws->produced.load(std::memory_order_acquire), t_seen = t0, t_prev = t0;
        while (seen * BLK < a) {
            if (stop->load(std::memory_order_relaxed)) {
                dead = true;
                break;
            }
            bo.pause();
            const uint64_t now = now_ns(), prod = ws->produced.load(std::memory_order_acquire);
            if (prod != seen || now - t_prev > 1000000000ull) {
                seen = prod;
                t_seen = now;
            } else if (now - t_seen > stall_ns) {
                dead = true;
                break;
            }
            t_prev = now;
        }
        ws->rd_wait_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
        return !dead;
    }

    // make at least one word available at the current position (waits for the generator)
    void refill() {
        const uint64_t a = pos();
        publish(a);
        if (dead || !wait_produced(a + 1)) {
            static const uint32_t zeros[2] = {0, 0};
            base = cur = zeros;
            end = zeros + 2;
            a_base = a;
            return;
        }
        const uint64_t prod = ws->produced.load(std::memory_order_acquire);
        const uint64_t ring_words = ws->words();
        const uint64_t off = a % ring_words;
        // (at most 32 blocks at a time: the shared words are published here only, and a reader that took everything produced kept the
        // generator out of the ring until it had consumed all of it: the two took turns)
        const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(prod * BLK - a, ring_words - off), 32ull * BLK);
        base = cur = &ws->ring[off];
        end = cur + n;
        a_base = a;
    }
    inline size_t avail() {
        if (cur == end) refill();
        return (size_t)(end - cur);
    }
    inline uint32_t next32() {
        if (__builtin_expect(cur == end, 0)) refill();
        return temper(*cur++);
    }
    inline uint64_t next64() {
        const uint64_t hi = next32();
        const uint64_t lo = next32();
        return (hi << 32) | lo;
    }
    inline double next_double() {
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // distributions.c random_interval(): masked rejection in [0, max]
    inline uint64_t random_interval(uint64_t max) {
        if (max == 0) return 0;
        uint64_t mask = max, value;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        mask |= mask >> 32;
        if (max <= 0xffffffffull) {
            while ((value = (next32() & mask)) > max && !dead) {
            }
        } else {
            while ((value = (next64() & mask)) > max && !dead) {
            }
        }
        return value;
    }
    // the smallest all-ones mask covering v
    static inline uint64_t mask_of(uint64_t v) {
        v |= v >> 1;
        v |= v >> 2;
        v |= v >> 4;
        v |= v >> 8;
        v |= v >> 16;
        v |= v >> 32;
        return v;
    }
    // randint(0, rng + 1) / random_interval(rng) with the mask made once by the caller (rng <= 2^32 - 2, mask = mask_of(rng)):
    // the per-walker loops of the DE and snooker moves draw from the same few ranges over and over
    inline uint32_t masked32(uint32_t rng, uint32_t mask) {
        uint32_t val;
        do {
            val = next32() & mask;
        } while (val > rng && !dead);
        return val;
    }
    // RandomState.randint(0, n) element
    inline uint64_t randint(uint64_t n) {
        const uint64_t rng = n - 1;
        if (rng == 0) return 0;
        if (rng <= 0xffffffffull) {
            if (rng == 0xffffffffull) return next32();
            uint64_t mask = rng;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            uint32_t val;
            do {
                val = next32() & (uint32_t)mask;
            } while (val > rng && !dead);
            return val;
        }
        if (rng == 0xffffffffffffffffull) return next64();
        uint64_t mask = rng, val;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        mask |= mask >> 32;
        do {
            val = next64() & mask;
        } while (val > rng && !dead);
        return val;
    }
    // n consecutive random_sample() values; zz_a != 0: transformed to the stretch factor on the fly
    void fill_doubles(double* dst, int64_t n, double zz_a = 0.0) {
        int64_t k = 0;
        while (k < n && !dead) {
            const size_t av = avail();
            if (av < 2) {                    // the window ends in the middle of a pair
                const double u = next_double();
                if (zz_a != 0.0) {
                    const double tt = (zz_a - 1.0) * u + 1.0;
                    dst[k++] = tt * tt / zz_a;
                } else {
                    dst[k++] = u;
                }
                continue;
            }
            const int64_t take = std::min<int64_t>((int64_t)(av / 2), n - k);
            if (zz_a != 0.0)
                convert_pairs_zz(cur, dst + k, take, zz_a);
            else
                convert_pairs(cur, dst + k, take);
            cur += 2 * take;
            k += take;
        }
    }
    // n consecutive randint(0, bound) values (32-bit results: bound <= 2^32)
    void fill_randint32(int32_t* dst, int64_t n, uint64_t bound) {
        const uint64_t rng = bound - 1;
        if (rng == 0) {
            for (int64_t k = 0; k < n; ++k) dst[k] = 0;
            return;
        }
        if (rng >= 0xffffffffull) {
            for (int64_t k = 0; k < n; ++k) dst[k] = (int32_t)randint(bound);
            return;
        }
        uint32_t mask = (uint32_t)rng;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        int64_t k = 0;
        if (mask == (uint32_t)rng) {         // power-of-two bound: no rejection, a masked copy
            while (k < n && !dead) {
                const int64_t take = std::min<int64_t>((int64_t)avail(), n - k);
                for (int64_t e = 0; e < take; ++e) dst[k + e] = (int32_t)(temper(cur[e]) & mask);
                cur += take;
                k += take;
            }
            return;
        }
        while (k < n && !dead) {
            const size_t av = avail();
            const uint32_t* p = cur;
            const uint32_t* pe = cur + av;
#ifdef EMX_HAVE_AVX512_GEN
            if (vec_ok) p += randint_scan_avx512(p, av, mask, (uint32_t)rng, dst, k, n);
#endif
            int budget = 64;                  // (the tail, and the words between two refills; then the vector path is tried again)
            while (p < pe && k < n && budget-- > 0) {
                const uint32_t v = temper(*p++) & mask;
                dst[k] = (int32_t)v;          // branch-free compaction: a rejected value is overwritten by the next
                k += (v <= (uint32_t)rng);
            }
            cur = p;
        }
    }
    // RandomState.shuffle of n items: the accepted swap target of every i = n-1 .. 1, stored as jr[(n - 1) - i]
    // (jr has 16 slots of slack)
    void shuffle_targets(uint32_t* jr, int64_t n, bool vec) {
        const int64_t nm1 = n - 1;
        int64_t i = n - 1;
        while (i > 0 && (uint64_t)i > 0xffffffffull) {
            jr[nm1 - i] = (uint32_t)random_interval((uint64_t)i);     // unreachable for int32 walker counts
            --i;
        }
        while (i > 0 && !dead) {
            uint32_t mask = (uint32_t)i;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            const int64_t lo = (int64_t)(mask >> 1);           // i in (lo, mask] share this mask
            while (i > lo && !dead) {
                const size_t av = avail();
                const uint32_t* p = cur;
                const uint32_t* pe = cur + av;
#ifdef EMX_HAVE_AVX512_GEN
                if (vec) p += shuffle_scan_avx512(p, av, mask, i, lo, jr, nm1);
#endif
                int budget = 16;                                // one vector's worth, then the fast path is tried again
                while (p < pe && i > lo && budget-- > 0) {
                    const uint32_t v = temper(*p++) & mask;
                    jr[nm1 - i] = v;                            // a rejected draw is overwritten by the next one
                    i -= (int64_t)(v <= (uint32_t)i);
                }
                cur = p;
            }
        }
        (void)vec;
    }
    // n consecutive randn() draws of the legacy polar method as tokens (RawStep::gx / gr2): the cached second normal of an earlier
    // call first, then pairs from accepted candidates; a last odd draw leaves its pair's second normal cached
    void fill_polar(double* gx, double* gr2, int64_t n, int& has_gauss, double& gauss) {
        int64_t t = 0;
        if (has_gauss && n > 0) {
            gx[0] = gauss;
            gr2[0] = -1.0;
            has_gauss = 0;
            gauss = 0.0;
            t = 1;
        }
        while (t < n && !dead) {
#ifdef EMX_HAVE_AVX512_GEN
            if (vec_dq && t + 8 <= n) {
                const size_t av = avail();
                cur += polar_scan_avx512(cur, av, gx, gr2, t, n);
                if (t >= n) break;                       // (an even count met exactly: nothing is cached)
            }
#endif
            // one accepted candidate the scalar way: the words between two refills, and the end of the draws
            double x1, x2, r2;
            do {
                x1 = 2.0 * next_double() - 1.0;
                x2 = 2.0 * next_double() - 1.0;
                r2 = x1 * x1 + x2 * x2;
            } while ((r2 >= 1.0 || r2 == 0.0) && !dead);
            gx[t] = x2;                                  // this call returns f * x2 ...
            gr2[t] = r2;
            ++t;
            if (t < n) {                                 // ... and the next one the cached f * x1
                gx[t] = x1;
                gr2[t] = r2;
                ++t;
            } else {
                const double f = std::sqrt(-2.0 * std::log(r2) / r2);
                gauss = f * x1;
                has_gauss = 1;
            }
        }
    }
    // the snooker move's five draws per walker, n walkers (ranges r[k] = n_k - 1 <= 2^32 - 2 with masks m[k]; perm = j2 | j1 << 2)
    void fill_snooker(int32_t* p0, int32_t* p1, int32_t* p2, uint8_t* perm, int64_t n, const uint32_t* r, const uint32_t* m) {
        int64_t t = 0;
        while (t < n && !dead) {
#ifdef EMX_HAVE_AVX512_GEN
            if (vec_ok) {
                const size_t av = avail();
                cur += snooker_scan_avx512(cur, av, r, m, p0, p1, p2, perm, t, n);
                if (t >= n) break;
            }
#endif
            p0[t] = (int32_t)masked32(r[0], m[0]);
            p1[t] = (int32_t)masked32(r[1], m[1]);
            p2[t] = (int32_t)masked32(r[2], m[2]);
            const uint32_t j2 = masked32(2u, 3u);                                              // random_interval(2)
            const uint32_t j1 = next32() & 1u;                                                 // random_interval(1): the mask rejects nothing
            perm[t] = (uint8_t)(j2 | (j1 << 2));
            ++t;
        }
    }
    // The next F words are fixed-length draws the consumer makes again from the generator's STATE (PipeStepInfo::regen): the state --
    // the ring's block itself: it holds state words -- at every SB-th block of the region goes to `keys`, the words are stepped over.
    // Nothing but those blocks crosses to this core.  -> false: the pipeline is dead.
    bool skip_region_keys(uint64_t F, int SB, uint32_t* keys, int32_t& off, int32_t& nseg) {
        const uint64_t p = pos();
        const uint64_t b0 = p / BLK, b1 = (p + F - 1) / BLK;
        off = (int32_t)(p - b0 * BLK);
        nseg = 0;
        for (uint64_t b = b0; b <= b1 && !dead; b += (uint64_t)SB) {
            publish(std::max<uint64_t>(p, b * BLK));          // block b stays in the ring; the generator may run its lead beyond it
            if (!wait_produced((b + 1) * BLK)) return false;
            std::memcpy(keys + (size_t)nseg * BLK, &ws->ring[(b % ws->nblk) * BLK], (size_t)BLK * 4);
            ++nseg;
        }
        // the region's last block exists -- and stays -- before anybody takes the generator state behind it (tokenizer_main's snapshot)
        publish(p + F);
        if (dead || !wait_produced(p + F)) return false;
        seek(p + F);
        return true;
    }
    // n state words, verbatim (fixed-length draws are tempered and converted by the finishers -- or by the consumer's kernel)
    void copy_words(uint32_t* dst, int64_t n) {
        int64_t k = 0;
        while (k < n && !dead) {
            const int64_t take = std::min<int64_t>((int64_t)avail(), n - k);
            std::memcpy(dst + k, cur, (size_t)take * 4);
            cur += take;
            k += take;
        }
    }
};

// de.py:49: a DE step's pair codes are drawn from [0, pop), pop = nc (nc - 1); below 2^32 they are 32-bit draws and travel in the
// sink's p0 column from the tokenizer to the finisher
inline bool de_codes_fit_32(uint64_t pop) { return pop - 1 != 0 && pop - 1 < 0xffffffffull; }

// randint(0, bound) draws exactly one word per value when bound is a power of two (the mask rejects nothing)
inline bool pow2_bound(uint64_t bound) { return bound >= 2 && bound <= 0x80000000ull && (bound & (bound - 1)) == 0; }

struct RawStep {               // tokens of one step that do not already sit in the plan sink
    std::vector<uint32_t> j;   // Fisher-Yates targets, reversed: j[(N - 1) - i] for i = N-1 .. 1 (+ 16 slots of slack)
    std::vector<uint32_t> wz, wr, wu;   // state words of the fixed-length draws, converted by the finisher: rand(Ns) of the stretch factor
                                        // (2 per walker), power-of-two randint (1), the accept uniforms (2).  (A raw step's go straight
                                        // into the sink: PipeStepInfo::raw.)
    std::vector<uint64_t> k64; // DE: pair codes (de.py:49)
    std::vector<double> gx, gr2;  // DE: polar-method tokens of randn (de.py:56): normal = gx * sqrt(-2 ln r2 / r2), r2 < 0: gx itself
    std::vector<uint8_t> perm; // snooker: shuffle(w) draws, j2 | j1 << 2
};

}  // namespace

struct MtPlanPipeline::Impl {
    int64_t N = 0, nsteps = 0;
    int32_t D = 0;
    std::vector<emx_move_desc> moves;
    std::vector<double> cdf;
    std::vector<PlanSink> sinks;
    int nsinks = 0, K = 1, NR = 0, NSNAP = 0;
    MT19937Legacy start;
    WordStream ws;
    std::vector<RawStep> raws;
    std::vector<PipeStepInfo> infos;               // [nsinks]
    std::vector<MT19937Legacy> snaps;              // [NSNAP]
    // sequence flags (step numbers), one cache line each
    struct alignas(64) Seq {
        std::atomic<int64_t> v{0};
    };
    std::vector<Seq> raw_ready, raw_done, sink_ready;
    alignas(64) std::atomic<int64_t> released{0};  // steps whose sink may be rewritten: [0, released)
    std::atomic<bool> stop{false};
    std::atomic<bool> failed{false};
    std::thread gen, tok;
    std::vector<std::thread> fin;
    bool joined = false;
    uint64_t t_start = 0, tok_done_ns = 0;
    std::atomic<uint64_t> tok_wait_sink_ns{0}, tok_busy_ns{0};                 // read by stage_times() while the threads run
    bool vec_scan = false, vec_dq = false, stats = false, fill_unused = false, device_finish = false;
    int64_t regen_min = 0;                         // > 0: eligible steps go out as generator states (PipeStepInfo::regen)
    uint64_t tok_shuffle_ns = 0;
    std::atomic<int64_t> tok_steps{0};             // steps tokenised so far
    std::vector<uint64_t> fin_wait_ns;
    std::vector<std::atomic<uint64_t>> fin_busy_ns;

    void tokenizer_main();
    void finisher_main(int id);
    void tokenize(Reader& rd, int64_t n, int& has_gauss, double& gauss);
    void finish_step(int64_t n, std::vector<uint8_t>& labels);
    void join_all();
};

bool MtPlanPipeline::supports(int32_t nmoves, const emx_move_desc* moves) {
    for (int i = 0; i < nmoves; ++i) {
        const int k = moves[i].kind;
        if (k != EMX_MOVE_STRETCH && k != EMX_MOVE_DE && k != EMX_MOVE_SNOOKER) return false;
        if (moves[i].nsplits < 2 || moves[i].nsplits > MAX_SPLITS) return false;
    }
    return nmoves >= 1;
}

MtPlanPipeline::MtPlanPipeline(const MT19937Legacy& start, int64_t N, int32_t D, int32_t nmoves, const emx_move_desc* moves,
                               const double* cdf, int64_t nsteps, const PlanSink* sinks, int32_t nsinks, int32_t nworkers,
                               bool fill_unused_fields, bool device_finish, bool bursty_consumer, int64_t regen_min_walkers)
    : impl_(new Impl()) {
    Impl& m = *impl_;
    m.regen_min = device_finish ? regen_min_walkers : 0;
    {
        const char* e = getenv("EMX_PIPE_SPIN_US");
        m.ws.spin_ns = e ? (uint64_t)atoll(e) * 1000ull : (bursty_consumer ? 150000ull : 0ull);
    }
    m.N = N;
    m.D = D;
    m.nsteps = nsteps;
    m.moves.assign(moves, moves + nmoves);
    m.cdf.assign(cdf, cdf + nmoves);
    m.sinks.assign(sinks, sinks + nsinks);
    m.nsinks = nsinks;
    int K = nworkers;
    const CpuSet cs = pipeline_cpus();
    if (K <= 0) {
        // CPUs the threads may actually run on: the L3 domain they are confined to, else what the machine reports.  Six finishers
        // where there is room (round 3, MI355X host, 16 CPUs in the domain: 0.078-0.082 -> 0.063 ms/step at 65 536 walkers, eight
        // no better: 0.061-0.063; the finishers' writes into cache-cold pinned memory were the slowest stage at four, 84 us per
        // thread and step against 69 for the generator) -- generator + tokenizer + finishers + the caller must fit.
#if defined(__linux__)
        const unsigned hw = cs.valid ? (unsigned)CPU_COUNT(&cs.set) : std::thread::hardware_concurrency();
#else
        const unsigned hw = std::thread::hardware_concurrency();
#endif
        K = hw >= 14 ? 6 : hw >= 10 ? 4 : hw >= 6 ? 3 : hw >= 4 ? 2 : 1;
        // device finish leaves the finishers the swaps and `order` only (170 against 250-290 us of work per step at 65 536 walkers):
        // five keep up, and generator + tokenizer + five + the caller are the eight threads an eight-core domain has cores for -- no
        // outlier in five fresh processes each (45.7-47.1 us/step; six: 45.3-46.5 and one 60.7 with the generator on a busy core's
        // sibling; four: 46.6-48.3; profiles/r05/exact_c2_finisher_count.txt)
        // (up to 131 072 walkers: beyond, the finishers' swaps bound the step and six are faster -- 262 144 walkers 214 against 324 us/step)
        // (a bursty consumer -- the persistent launches: one enqueue per sixteen steps -- leaves the caller's core idle: the sixth finisher
        // has it; with regen steps the finishers are what bounds the step there, 171 us of swaps and `order` per step of 65 536 walkers)
        if (device_finish && K == 6 && hw <= 16 && N <= 131072 && !bursty_consumer) K = 5;
    }
    K = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(K, nsinks), nsteps));
    m.K = K;
    m.NR = K + 2;
    m.NSNAP = nsinks + 68;          // (generator states behind the last steps: emx.hip's persist_settle takes the pipeline back up to 60 steps)
    m.start = start;
    m.device_finish = device_finish;
    m.ws.nblk = RING_BLKS;
    m.ws.own.resize(m.ws.nblk * BLK + RING_MIRROR + 16);
    m.ws.ring = reinterpret_cast<uint32_t*>(((uintptr_t)m.ws.own.data() + 63) & ~(uintptr_t)63);      // (the vector twist stores whole lines)
    m.raws.resize(m.NR);
    bool any_shuffle = false, any_de = false, any_sn = false, any_stretch = false;
    for (auto& mv : m.moves) {
        any_shuffle |= mv.randomize_split != 0;
        any_stretch |= mv.kind == EMX_MOVE_STRETCH;
        any_de |= mv.kind == EMX_MOVE_DE;
        any_sn |= mv.kind == EMX_MOVE_SNOOKER;
    }
    for (auto& r : m.raws) {
        if (any_shuffle) r.j.resize((size_t)N + 16);
        r.wu.resize((size_t)2 * N);
        if (any_stretch) {
            r.wz.resize((size_t)2 * N);
            r.wr.resize((size_t)N);
        }
        if (any_de) {
            r.k64.resize((size_t)N);
            r.gx.resize((size_t)N);
            r.gr2.resize((size_t)N);
        }
        if (any_sn) r.perm.resize((size_t)N);
    }
    m.infos.resize(nsinks);
    m.snaps.resize(m.NSNAP);
    m.raw_ready = std::vector<Impl::Seq>(m.NR);
    m.raw_done = std::vector<Impl::Seq>(m.NR);
    m.sink_ready = std::vector<Impl::Seq>(nsinks);
    for (int s = 0; s < m.NR; ++s) {
        m.raw_ready[s].v.store(-1);
        m.raw_done[s].v.store((int64_t)s - m.NR);     // "step s - NR is done": slot s is free for step s
    }
    for (int s = 0; s < nsinks; ++s) m.sink_ready[s].v.store(-1);
#ifdef EMX_HAVE_AVX512_GEN
    m.vec_scan = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx2") && !getenv("EMX_PIPE_NO_AVX512");
    m.vec_dq = m.vec_scan && __builtin_cpu_supports("avx512dq");
    if (m.vec_scan) pick_compaction();
#endif
    m.stats = getenv("EMX_PIPE_STATS") != nullptr;
    m.fill_unused = fill_unused_fields;
    m.fin_wait_ns.assign(K, 0);
    m.fin_busy_ns = std::vector<std::atomic<uint64_t>>(K);
    for (auto& b : m.fin_busy_ns) b.store(0, std::memory_order_relaxed);
    m.t_start = now_ns();
    // One physical core per thread is the fastest placement on an idle host (0.080 vs 0.100 ms/step at 65 536 walkers) and the
    // slowest when another tenant of the machine occupies one of the chosen cores (0.19 seen): opt-in (EMX_PIPE_CORE_PINNING=1);
    // by default the threads may move inside the L3 domain.
    const std::vector<int> cores = getenv("EMX_PIPE_CORE_PINNING") ? distinct_cores(cs) : std::vector<int>();
    m.gen = std::thread(generator_main, &m.ws, m.start.key);
    confine(m.gen, cs, cores, 0);
    m.tok = std::thread([&m] { m.tokenizer_main(); });
    confine(m.tok, cs, cores, 1);
    for (int k = 0; k < K; ++k) {
        m.fin.emplace_back([&m, k] { m.finisher_main(k); });
        confine(m.fin.back(), cs, cores, (int)cores.size() >= 3 + k ? 2 + k : -1);     // out of cores: anywhere in the L3 domain
    }
}

int MtPlanPipeline::workers() const { return impl_->K; }

// Where the pipeline's time goes, per step PRODUCED so far (plain counters of the stage threads, read while they run: a
// snapshot for reporting, not a synchronisation).
void MtPlanPipeline::stage_times(double out[6], int64_t* steps) const {
    const Impl& m = *impl_;
    const int64_t n = std::max<int64_t>(1, m.tok_steps.load(std::memory_order_relaxed));
    const double elapsed = (double)(now_ns() - m.t_start);
    uint64_t fin = 0;
    for (const auto& b : m.fin_busy_ns) fin += b.load(std::memory_order_relaxed);
    out[0] = elapsed * 1e-3 / n;                                   // wall clock per produced step (it runs ahead: an upper bound)
    out[1] = (elapsed - (double)m.ws.gen_wait_ns) * 1e-3 / n;      // generator: twist + temper
    out[2] = (double)m.tok_busy_ns * 1e-3 / n;                     // tokenizer: rejection tests, raw words handed on
    out[3] = (double)fin * 1e-3 / n;                               // finishers, summed over the K threads
    out[4] = (double)m.ws.rd_wait_ns * 1e-3 / n;                   // tokenizer waited for the generator
    out[5] = (double)m.tok_wait_sink_ns * 1e-3 / n;                // tokenizer waited for a free staging buffer (the consumer)
    if (steps) *steps = m.tok_steps.load(std::memory_order_relaxed);
}

void MtPlanPipeline::Impl::join_all() {
    if (joined) return;
    stop.store(true);
    ws.stop.store(true);
    if (gen.joinable()) gen.join();
    if (tok.joinable()) tok.join();
    for (auto& t : fin)
        if (t.joinable()) t.join();
    joined = true;
    if (getenv("EMX_PIPE_STATS")) {
        const double tot = (now_ns() - t_start) * 1e-6;
        fprintf(stderr, "[emx pipe] N=%lld steps=%lld workers=%d: %.3f ms total; generator %llu blocks, waited %.3f ms for space; tokenizer finished at "
                        "%.3f ms (shuffle scans %.3f ms), waited %.3f ms for words and %.3f ms for sinks/raw slots;",
                (long long)N, (long long)nsteps, K, tot, (unsigned long long)ws.gen_blocks, ws.gen_wait_ns * 1e-6, tok_done_ns * 1e-6,
                tok_shuffle_ns * 1e-6, ws.rd_wait_ns * 1e-6, tok_wait_sink_ns * 1e-6);
        for (int k = 0; k < K; ++k) fprintf(stderr, " fin%d busy %.3f wait %.3f;", k, fin_busy_ns[k] * 1e-6, fin_wait_ns[k] * 1e-6);
        fprintf(stderr, "\n");
    }
}

MtPlanPipeline::~MtPlanPipeline() {
    impl_->join_all();
    delete impl_;
}

// ---- tokenizer: one pass over the stream in the reference's draw order --------------------------------------------
void MtPlanPipeline::Impl::tokenize(Reader& rd, int64_t n, int& has_gauss, double& gauss) {
    const int nm = (int)moves.size();
    // ensemble.py:406  move = self._random.choice(self._moves, p=self._weights): one uniform against the cdf
    int mi;
    {
        const double u = rd.next_double();
        int lo = 0, hi = nm;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (u < cdf[mid])
                hi = mid;
            else
                lo = mid + 1;
        }
        mi = lo < nm ? lo : nm - 1;
    }
    const emx_move_desc& mv = moves[mi];
    const int S = mv.nsplits;
    PipeStepInfo& info = infos[n % nsinks];
    info.move = mi;
    info.S = S;
    info.off[0] = 0;
    for (int s = 0; s < S; ++s) info.off[s + 1] = info.off[s] + (int32_t)((N - s + S - 1) / S);   // label counts survive the shuffle
    RawStep& raw = raws[n % NR];
    const PlanSink& sk = sinks[n % nsinks];
    info.raw = (device_finish && mv.kind == EMX_MOVE_STRETCH && S <= PIPE_RAW_SPLITS) ? 1 : 0;
    info.wr_ring = 1;
    const uint64_t ts0 = stats ? now_ns() : 0;
    if (mv.randomize_split) rd.shuffle_targets(raw.j.data(), N, vec_scan);                        // red_blue.py:80
    if (stats) tok_shuffle_ns += now_ns() - ts0;
    info.regen = 0;
    if (info.raw && regen_min > 0 && N >= regen_min && S == 2 && (N % 2) == 0 && pow2_bound((uint64_t)(N / 2))) {
        // both splits' draws are fixed-length and contiguous: 2 x (2 Ns + Ns + 2 Ns) = 5 N words.  The states must fit the p0 column.
        const uint64_t F = 5ull * (uint64_t)N;
        const uint64_t nblk_max = F / BLK + 2, nseg_max = (nblk_max + PIPE_REGEN_SB - 1) / PIPE_REGEN_SB;
        if (nseg_max * BLK <= (uint64_t)N) {
            info.regen = 1;
            if (!rd.skip_region_keys(F, PIPE_REGEN_SB, reinterpret_cast<uint32_t*>(sk.p0), info.regen_off, info.regen_nseg)) return;
            return;
        }
    }
    for (int split = 0; split < S && !rd.dead; ++split) {
        const int64_t base = info.off[split], ns = info.off[split + 1] - info.off[split], nc = N - ns;
        if (mv.kind == EMX_MOVE_STRETCH) {
            // stretch.py:30 rand(Ns): fixed length.  stretch.py:32 randint(Nc, Ns): fixed length too where Nc is a power of two (no
            // rejection), else the rejection decides the position.  A raw step's words go where their values will stand.
            rd.copy_words(info.raw ? reinterpret_cast<uint32_t*>(sk.s0 + base) : raw.wz.data() + 2 * base, 2 * ns);
            if (pow2_bound((uint64_t)nc))
                rd.copy_words(info.raw ? reinterpret_cast<uint32_t*>(sk.p0 + base) : raw.wr.data() + base, ns);
            else
                rd.fill_randint32(sk.p0 + base, ns, (uint64_t)nc);
        } else if (mv.kind == EMX_MOVE_DE) {
            const uint64_t pop = (uint64_t)nc * (uint64_t)(nc - 1);
            if (de_codes_fit_32(pop)) {
                // de.py:49: the pair codes are randint(0, pop) draws, 32-bit here: they wait in the sink's p0 column (which the
                // finisher overwrites with the partners they decode to), made 16 words at a time
                rd.fill_randint32(sk.p0 + base, ns, pop);
            } else {
                for (int64_t t = 0; t < ns; ++t) raw.k64[base + t] = rd.randint(pop);
            }
            // de.py:56 randn(ns, 1): legacy polar method, the second value of a pair is cached for the next call
            rd.fill_polar(raw.gx.data() + base, raw.gr2.data() + base, ns, has_gauss, gauss);
        } else {   // snooker: de_snooker.py:37-40 per walker
            int cs[3], q = 0;
            for (int s = 0; s < S && q < 3; ++s)
                if (s != split) cs[q++] = s;
            const uint64_t nj[3] = {(uint64_t)(info.off[cs[0] + 1] - info.off[cs[0]]), (uint64_t)(info.off[cs[1] + 1] - info.off[cs[1]]),
                                    (uint64_t)(info.off[cs[2] + 1] - info.off[cs[2]])};
            const bool small = nj[0] >= 2 && nj[1] >= 2 && nj[2] >= 2 && nj[0] < 0xffffffffull && nj[1] < 0xffffffffull && nj[2] < 0xffffffffull;
            if (small) {                                                                          // (the masks made once per split)
                const uint32_t rr[3] = {(uint32_t)(nj[0] - 1), (uint32_t)(nj[1] - 1), (uint32_t)(nj[2] - 1)};
                const uint32_t mm[3] = {(uint32_t)Reader::mask_of(rr[0]), (uint32_t)Reader::mask_of(rr[1]), (uint32_t)Reader::mask_of(rr[2])};
                rd.fill_snooker(sk.p0 + base, sk.p1 + base, sk.p2 + base, raw.perm.data() + base, ns, rr, mm);
            } else {
                for (int64_t t = 0; t < ns && !rd.dead; ++t) {
                    sk.p0[base + t] = (int32_t)rd.randint(nj[0]);
                    sk.p1[base + t] = (int32_t)rd.randint(nj[1]);
                    sk.p2[base + t] = (int32_t)rd.randint(nj[2]);
                    const uint32_t j2 = (uint32_t)rd.random_interval(2);
                    const uint32_t j1 = (uint32_t)rd.random_interval(1);
                    raw.perm[base + t] = (uint8_t)(j2 | (j1 << 2));
                }
            }
        }
        rd.copy_words(info.raw ? reinterpret_cast<uint32_t*>(sk.uacc + base) : raw.wu.data() + 2 * base, 2 * ns);      // red_blue.py:100 rand() x Ns
        if (info.raw && !pow2_bound((uint64_t)nc)) info.wr_ring = 0;       // (S > 2 with N % S != 0: set sizes differ, so may this)
    }
}

void MtPlanPipeline::Impl::tokenizer_main() {
    stage_thread_setup(ws.spin_ns);
    Reader rd;
    rd.ws = &ws;
    rd.stop = &stop;
    rd.seek((uint64_t)start.pos);
    rd.vec_ok = vec_scan;
    rd.vec_dq = vec_dq;
    int has_gauss = start.has_gauss;
    double gauss = start.gauss;
    for (int64_t n = 0; n < nsteps; ++n) {
        Backoff bo(ws.spin_ns);
        // the sink of step n is free once step n - nsinks has been uploaded; the raw slot once its finisher is done
        if (n >= released.load(std::memory_order_acquire) + nsinks || raw_done[n % NR].v.load(std::memory_order_acquire) != n - NR) {
            const uint64_t t0 = now_ns();
            while ((n >= released.load(std::memory_order_acquire) + nsinks || raw_done[n % NR].v.load(std::memory_order_acquire) != n - NR) &&
                   !stop.load(std::memory_order_relaxed))
                bo.pause();
            tok_wait_sink_ns += now_ns() - t0;
        }
        if (stop.load(std::memory_order_relaxed)) return;
        {
            const uint64_t t0 = now_ns();
            const uint64_t w0 = ws.rd_wait_ns.load(std::memory_order_relaxed);
            tokenize(rd, n, has_gauss, gauss);
            tok_busy_ns.fetch_add((now_ns() - t0) - (ws.rd_wait_ns.load(std::memory_order_relaxed) - w0), std::memory_order_relaxed);         // without the time it waited for words
        }
        if (rd.dead) return;
        // generator state after this step (NumPy get_state() semantics: a block consumed to its end reports pos = 624)
        {
            const uint64_t a = rd.pos();
            const uint64_t blk = a > 0 ? (a - 1) / BLK : 0;
            MT19937Legacy& sn = snaps[n % NSNAP];
            // the block is still retained: keep <= (a - 1) / BLK and the generator never overwrites blocks >= keep
            sn.set_state(&ws.ring[(blk % ws.nblk) * BLK], (int)(a - blk * BLK), has_gauss, gauss);      // (the ring holds the state words themselves)
        }
        raw_ready[n % NR].v.store(n, std::memory_order_release);
        tok_steps.store(n + 1, std::memory_order_relaxed);
    }
    tok_done_ns = now_ns() - t_start;
}

// ---- finisher: one whole step, independent of every other step ----------------------------------------------------
void MtPlanPipeline::Impl::finish_step(int64_t n, std::vector<uint8_t>& labels) {
    const PipeStepInfo& info = infos[n % nsinks];
    const emx_move_desc& mv = moves[info.move];
    const int S = info.S;
    const RawStep& raw = raws[n % NR];
    const PlanSink& sk = sinks[n % nsinks];
    uint8_t* x = labels.data();
    for (int64_t i = 0; i < N; ++i) x[i] = (uint8_t)(i % S);                 // red_blue.py:78
    if (mv.randomize_split) {
        const uint32_t* jr = raw.j.data();
        for (int64_t i = N - 1; i > 0; --i) {                               // red_blue.py:80 (the swaps of RandomState.shuffle)
            const uint32_t jj = jr[(N - 1) - i];
            const uint8_t t = x[i];
            x[i] = x[jj];
            x[jj] = t;
        }
    }
    // boolean-mask gather order (red_blue.py:85): ascending walker index inside each set
    int32_t cur[MAX_SPLITS + 1];
    for (int s = 0; s < S; ++s) cur[s] = info.off[s];
    int32_t* order = sk.order;
    for (int64_t i = 0; i < N; ++i) order[cur[x[i]]++] = (int32_t)i;
    for (int split = 0; split < S; ++split) {
        const int64_t base = info.off[split], ns = info.off[split + 1] - info.off[split], nc = N - ns;
        auto comp = [&](uint64_t r) -> int32_t { return (int64_t)r < base ? order[r] : order[r + ns]; };   // stretch.py:27
        if (info.regen) continue;          // (the consumer makes the draws again from the generator states in the sink: k_plan_regen)
        if (info.raw) {
            // device finish: the tokenizer put the words where their values will stand (PipeStepInfo::raw); only a step whose splits
            // drew their partners both ways (set sizes that are and are not powers of two) has values to make here
            if (!info.wr_ring && pow2_bound((uint64_t)nc)) {
                const uint32_t msk = (uint32_t)(nc - 1);
                for (int64_t t = 0; t < ns; ++t) sk.p0[base + t] = (int32_t)(temper((uint32_t)sk.p0[base + t]) & msk);
            }
            continue;
        }
        convert_pairs(raw.wu.data() + 2 * base, sk.uacc + base, ns);                             // red_blue.py:100
        if (mv.kind == EMX_MOVE_STRETCH) {
            convert_pairs_zz(raw.wz.data() + 2 * base, sk.s0 + base, ns, mv.a);                    // stretch.py:30
            if (pow2_bound((uint64_t)nc)) {
                const uint32_t msk = (uint32_t)(nc - 1);
                const uint32_t* wr = raw.wr.data() + base;
                for (int64_t t = 0; t < ns; ++t) sk.p0[base + t] = (int32_t)(temper(wr[t]) & msk);
            }
            // p1 / p2 carry no information for a stretch step (one partner): the kernels never read them and the upload
            // stops before them; they are filled only for consumers that compare whole plans (fill_unused)
            if (fill_unused)
                for (int64_t t = 0; t < ns; ++t) sk.p1[base + t] = sk.p2[base + t] = order[base + t];
            for (int64_t t = 0; t < ns; ++t) sk.p0[base + t] = comp((uint32_t)sk.p0[base + t]);
        } else if (mv.kind == EMX_MOVE_DE) {
            const bool k32 = de_codes_fit_32((uint64_t)nc * (uint64_t)(nc - 1));
            int64_t t0 = 0;
#ifdef EMX_HAVE_AVX512_GEN
            if (vec_dq && nc < (1ll << 31))
                t0 = de_decode_avx512(raw.k64.data() + base, k32, ns, (uint64_t)nc, base, ns, order, sk.p0 + base, sk.p1 + base);
#endif
            for (int64_t t = t0; t < ns; ++t) {
                uint64_t f, s;
                de_pair(k32 ? (uint64_t)(uint32_t)sk.p0[base + t] : raw.k64[base + t], (uint64_t)nc, f, s);
                sk.p0[base + t] = comp(f);
                sk.p1[base + t] = comp(s);
            }
            double last_r2 = -1.0, last_fac = 0.0;                 // the two normals of a polar pair share r2, hence the factor
            for (int64_t t = 0; t < ns; ++t) {
                sk.p2[base + t] = order[base + t];
                const double r2 = raw.gr2[base + t];
                double g = raw.gx[base + t];
                if (r2 >= 0.0) {
                    if (r2 != last_r2) {
                        last_fac = std::sqrt(-2.0 * std::log(r2) / r2);
                        last_r2 = r2;
                    }
                    g = last_fac * g;
                }
                sk.s0[base + t] = mv.g0 * (1.0 + mv.sigma * g);
            }
        } else {
            int cs[3], q = 0;
            for (int s = 0; s < S && q < 3; ++s)
                if (s != split) cs[q++] = s;
            for (int64_t t = 0; t < ns; ++t) {
                int32_t w[3] = {order[info.off[cs[0]] + sk.p0[base + t]], order[info.off[cs[1]] + sk.p1[base + t]],
                                order[info.off[cs[2]] + sk.p2[base + t]]};
                const int j2 = raw.perm[base + t] & 3, j1 = (raw.perm[base + t] >> 2) & 1;
                std::swap(w[2], w[j2]);
                std::swap(w[1], w[j1]);
                sk.p0[base + t] = w[0];
                sk.p1[base + t] = w[1];
                sk.p2[base + t] = w[2];
                sk.s0[base + t] = 0.0;
            }
        }
    }
}

void MtPlanPipeline::Impl::finisher_main(int id) {
    stage_thread_setup(ws.spin_ns);
    std::vector<uint8_t> labels((size_t)N);
    for (int64_t n = id; n < nsteps; n += K) {
        Backoff bo(ws.spin_ns);
        const uint64_t t0 = now_ns();
        while (raw_ready[n % NR].v.load(std::memory_order_acquire) != n && !stop.load(std::memory_order_relaxed)) bo.pause();
        if (stop.load(std::memory_order_relaxed)) return;
        const uint64_t t1 = now_ns();
        fin_wait_ns[id] += t1 - t0;
        finish_step(n, labels);
        fin_busy_ns[id] += now_ns() - t1;
        raw_done[n % NR].v.store(n, std::memory_order_release);
        sink_ready[n % nsinks].v.store(n, std::memory_order_release);
    }
}

bool MtPlanPipeline::wait_ready(int64_t n, PipeStepInfo& info, void (*poll)(void*), void* poll_arg) {
    Impl& m = *impl_;
    if (n < 0 || n >= m.nsteps) return false;
    Backoff bo(m.ws.spin_ns);
    int spins = 0;
    while (m.sink_ready[n % m.nsinks].v.load(std::memory_order_acquire) != n) {
        if (m.failed.load() || m.stop.load()) return false;
        if (poll && ((++spins & 15) == 0)) poll(poll_arg);
        bo.pause();
    }
    info = m.infos[n % m.nsinks];
    return true;
}

void MtPlanPipeline::release(int64_t n) {
    Impl& m = *impl_;
    int64_t cur = m.released.load(std::memory_order_relaxed);
    if (n + 1 > cur) m.released.store(n + 1, std::memory_order_release);
}

void MtPlanPipeline::finish(int64_t steps_consumed, MT19937Legacy& out) {
    Impl& m = *impl_;
    m.join_all();
    if (steps_consumed <= 0)
        out = m.start;
    else
        out = m.snaps[(steps_consumed - 1) % m.NSNAP];
}

}  // namespace emx
