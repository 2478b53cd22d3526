// Host checks of the integer helpers the native plans use (csrc/emx_rng.hpp): the closed-form inverse of the keyed permutation against
// the loop form and against perm_fwd, the narrow bounded64 against the 128-bit product, pow2_reciprocal, split_sizes.
// Built and run by tests/test_plan_log.py.
#include <cstdio>
#include <initializer_list>
#include <cstdlib>
#include <cmath>
#include "emx_rng.hpp"
using namespace emx;
static uint32_t unmix_ref(uint32_t x, const PermKey& k) {
    x = unxorshift(x, k.s1, k.bits); x = ((x - k.c3) * k.m3inv) & k.mask;
    x = unxorshift(x, k.s2, k.bits); x = ((x - k.c2) * k.m2inv) & k.mask;
    x = unxorshift(x, k.s1, k.bits); x = ((x - k.c1) * k.m1inv) & k.mask; return x; }
int main() {
    long bad = 0, n = 0;
    for (uint64_t N : {1ull, 2ull, 3ull, 4ull, 5ull, 7ull, 8ull, 33ull, 100ull, 1000ull, 4096ull, 65536ull, 65537ull, 100003ull, 1048576ull, 16777216ull, 16777217ull, 40000000ull, 2147483648ull})
        for (uint64_t step = 0; step < 3; ++step) {
            PermKey k = make_perm_key(N, 42 + step, step);
            const uint64_t stride = N > 2000000 ? N / 1000003 + 1 : 1;
            for (uint64_t w = 0; w < N; w += stride) {
                uint32_t p = perm_fwd((uint32_t)w, k);
                if (perm_inv(p, k) != (uint32_t)w) ++bad;
                if (perm_unmix(p, k) != unmix_ref(p, k)) ++bad;
                ++n;
            }
        }
    // bounded64 narrow form against the 128-bit product; pow2 reciprocal
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 20000000; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t a = (uint32_t)s, b = (uint32_t)(s >> 32); uint64_t nn = (s >> 17) % 0xffffffffull + 1;
        if (i % 3 == 0) nn = 0xffffffffull; if (i % 5 == 0) { a = 0xffffffffu; b = 0xffffffffu; }
        const uint64_t r = ((uint64_t)a << 32) | b;
        if (bounded64(a, b, nn) != (uint64_t)(((unsigned __int128)r * nn) >> 64)) ++bad;
        ++n;
    }
    for (int e = -1000; e <= 1000; ++e) { double a = std::ldexp(1.0, e), inv; if (!pow2_reciprocal(a, inv) || inv != 1.0 / a) ++bad; }
    double inv; if (pow2_reciprocal(3.0, inv) || pow2_reciprocal(-2.0, inv) || pow2_reciprocal(0.0, inv) || pow2_reciprocal(INFINITY, inv)) ++bad;
    for (int N = 1; N < 300; ++N) for (int S = 1; S <= 9; ++S) { SplitSizes z = split_sizes(N, S); for (int q = 0; q < S; ++q) if (z.of(q) != (N - q + S - 1) / S) ++bad; }
    printf("checked %ld, bad %ld\n", n, bad);
    return bad != 0;
}
