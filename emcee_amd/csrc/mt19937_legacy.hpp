// MT19937 + the NumPy *legacy* RandomState algorithms the reference draws through.
//
// Reference call sites (relative to /root/reference/src/emcee):
//   ensemble.py:166-167,406  RandomState(), choice(moves, p=weights)
//   moves/red_blue.py:80,100 shuffle(inds), rand()
//   moves/stretch.py:30,32   rand(Ns), randint(Nc, size=Ns)
//   moves/de.py:49,56        choice(n, size, replace=True), randn(ns, 1)
//   moves/de_snooker.py:38-39 randint(Nc[j]), shuffle(w)
// NumPy is an un-vendored, unpinned dependency of the reference (setup.py:25); its legacy
// stream is frozen, and the algorithms below restate numpy/random/src (mt19937.c,
// legacy-distributions.c, distributions.c: random_interval, bounded_masked_uint32/64).
// tests/test_mt19937_exact.py checks every routine word-for-word against numpy.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace emx {

struct MT19937Legacy {
    uint32_t key[624];      // NumPy's (untempered) state words
    uint32_t out[624];      // the same block, tempered: what next32() hands out
    int pos = 624;
    int has_gauss = 0;
    double gauss = 0.0;

    static inline uint32_t temper(uint32_t y) {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }

    void set_state(const uint32_t* k, int p, int hg, double g) {
        std::memcpy(key, k, sizeof(key));
        for (int i = 0; i < 624; ++i) out[i] = temper(key[i]);
        pos = p;
        has_gauss = hg;
        gauss = g;
    }

    // Regenerate the whole 624-word block.  Each segment only reads words that are still "old"
    // inside one vector (key[kk+1] is read before key[kk..] is written), so the loops vectorise.
    void twist() {
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
        uint32_t* __restrict k = key;
        int kk;
#if defined(__clang__)
#pragma clang loop vectorize(enable) interleave(enable)
#endif
        for (kk = 0; kk < 624 - 397; kk++) {
            const uint32_t y = (k[kk] & UPPER) | (k[kk + 1] & LOWER);
            k[kk] = k[kk + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
        }
        // the second segment reads words written 227 positions earlier: chunks of <= 227 are independent
        for (int base = 624 - 397; base < 623; base += 224) {
            const int end = base + 224 < 623 ? base + 224 : 623;
#if defined(__clang__)
#pragma clang loop vectorize(enable) interleave(enable)
#endif
            for (kk = base; kk < end; kk++) {
                const uint32_t y = (k[kk] & UPPER) | (k[kk + 1] & LOWER);
                k[kk] = k[kk - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
            }
        }
        {
            const uint32_t y = (k[623] & UPPER) | (k[0] & LOWER);
            k[623] = k[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
        }
#if defined(__clang__)
#pragma clang loop vectorize(enable) interleave(enable)
#endif
        for (int i = 0; i < 624; ++i) out[i] = temper(k[i]);
        pos = 0;
    }

    inline uint32_t next32() {
        if (__builtin_expect(pos == 624, 0)) twist();
        return out[pos++];
    }

    inline uint64_t next64() {
        uint64_t hi = next32();
        uint64_t lo = next32();
        return (hi << 32) | lo;
    }

    // random_sample(): 53-bit double from two words.
    inline double next_double() {
        int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }

    // distributions.c random_interval(): masked rejection in [0, max].
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
            while ((value = (next32() & mask)) > max) {
            }
        } else {
            while ((value = (next64() & mask)) > max) {
            }
        }
        return value;
    }

    // RandomState.randint(0, n) element (dtype int64, masked): value in [0, n).
    inline uint64_t randint(uint64_t n) {
        uint64_t rng = n - 1;
        if (rng == 0) return 0;  // no draw consumed
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
            } while (val > rng);
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
        } while (val > rng);
        return val;
    }

    // legacy-distributions.c legacy_gauss(): Marsaglia polar with a cached second value.
    inline double next_gauss() {
        if (has_gauss) {
            const double tmp = gauss;
            gauss = 0.0;
            has_gauss = 0;
            return tmp;
        }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = 1;
        return f * x2;
    }

    // RandomState.shuffle on a 1-d array of n items: for i = n-1..1: j = random_interval(i); swap.
    // Same draws as the loop above them in NumPy, restructured: all i that share a rejection mask are served from
    // the tempered block in one tight loop (no per-element call, no mask recomputation).
    template <typename T>
    inline void shuffle(T* x, int64_t n) {
        int64_t i = n - 1;
        while (i > 0 && (uint64_t)i > 0xffffffffull) {           // 64-bit draws: astronomically large arrays only
            const int64_t j = (int64_t)random_interval((uint64_t)i);
            const T t = x[i];
            x[i] = x[j];
            x[j] = t;
            --i;
        }
        while (i > 0) {
            uint32_t mask = (uint32_t)i;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            const int64_t lo = (int64_t)(mask >> 1);             // i in (lo, mask] share this mask
            while (i > lo) {
                if (pos == 624) twist();
                int p = pos;
                const uint32_t* w = out;
                while (p < 624 && i > lo) {
                    // branch-free: a rejected draw (one in four on average, unpredictable) swaps x[i] with itself
                    const uint32_t v = w[p++] & mask;
                    const bool ok = v <= (uint32_t)i;
                    const int64_t j = ok ? (int64_t)v : i;
                    const T t = x[i];
                    x[i] = x[j];
                    x[j] = t;
                    i -= (int64_t)ok;
                }
                pos = p;
            }
        }
    }

    // n consecutive randint(0, bound) values (RandomState.randint(bound, size=n)), into `dst` through `map`
    template <typename F>
    inline void fill_randint(int64_t n, uint64_t bound, F&& put) {
        const uint64_t rng = bound - 1;
        if (rng == 0 || rng >= 0xffffffffull) {                  // no draw / whole words / 64-bit: the scalar routine
            for (int64_t k = 0; k < n; ++k) put(k, randint(bound));
            return;
        }
        uint32_t mask = (uint32_t)rng;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        int64_t k = 0;
        while (k < n) {
            if (pos == 624) twist();
            int p = pos;
            while (p < 624 && k < n) {
                const uint32_t v = out[p++] & mask;
                if (v <= (uint32_t)rng) put(k++, (uint64_t)v);
            }
            pos = p;
        }
    }

    // n consecutive random_sample() values (RandomState.rand(n))
    inline void fill_doubles(double* dst, int64_t n) {
        int64_t k = 0;
        while (k < n) {
            if (pos >= 623) {                    // fewer than two words left in the block: the slow path handles the seam
                dst[k++] = next_double();
                continue;
            }
            int p = pos;
            const int64_t take = (int64_t)((624 - p) / 2) < n - k ? (int64_t)((624 - p) / 2) : n - k;
            for (int64_t e = 0; e < take; ++e, p += 2) {
                const int32_t a = (int32_t)(out[p] >> 5), b = (int32_t)(out[p + 1] >> 6);
                dst[k + e] = (a * 67108864.0 + b) / 9007199254740992.0;
            }
            pos = p;
            k += take;
        }
    }

    // RandomState.choice(len(cdf), p=...) scalar: one random_sample, searchsorted(side='right').
    inline int choice_cdf(const double* cdf, int n) {
        const double u = next_double();
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (u < cdf[mid])
                hi = mid;
            else
                lo = mid + 1;
        }
        return lo < n ? lo : n - 1;
    }
};

}  // namespace emx
