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
    template <typename T>
    inline void shuffle(T* x, int64_t n) {
        for (int64_t i = n - 1; i > 0; --i) {
            const int64_t j = (int64_t)random_interval((uint64_t)i);
            const T t = x[i];
            x[i] = x[j];
            x[j] = t;
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
