// Counter-based RNG (Philox4x32-10) and the keyed walker permutation used by the native
// (EMX_RNG_PHILOX) mode.  Every function is __host__ __device__ so that the host can
// reproduce, bit for bit, the plan a kernel derives in flight (used by the parity tests).
//
// The native mode replaces the reference's serial MT19937 consumer
// (red_blue.py:80 shuffle, stretch.py:30-32, red_blue.py:100) with draws that are a pure
// function of (seed, step, walker): same distributions, different stream.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define EMX_HD __host__ __device__ __forceinline__
#else
#define EMX_HD inline
#endif

namespace emx {

EMX_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

// Both halves of a 32 x 32 -> 64 bit product.  (Round 3 measured ONE v_mad_u64_u32 in place of the v_mul_hi_u32 + v_mul_lo_u32
// pair the compiler emits for this: no difference -- C2 23.68 against 23.68, C3 38.61 against 38.67 us/step,
// profiles/r03/ab_mad64_spinsync.txt -- the wide multiply costs what the two it replaces cost.  Plain C it stays.)
EMX_HD void mul_hilo32(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    const uint64_t r = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(r >> 32);
    lo = (uint32_t)r;
}

// a ^ b ^ c: one v_bitop3_b32 on the device (the compiler leaves two v_xor_b32 when one operand is a scalar: 40 of a stretch
// plan entry's ~470 vector instructions)
EMX_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

struct Philox4 {
    uint32_t v[4];
};

// Philox4x32-R (Salmon et al. 2011), key = 64-bit seed, counter = 128 bits.  R = 10 is the recommended strength and what
// every decision-bearing draw uses; R = 7 is the smallest round count that passes BigCrush (ibid., table 2) and serves the
// bulk noise of the Gaussian Metropolis proposal, where the integer multiplies were the bottleneck.
template <int ROUNDS>
EMX_HD Philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mul_hilo32(M0, c0, hi0, lo0);
        mul_hilo32(M1, c2, hi1, lo1);
        c0 = xor3(hi1, c1, k0);
        c1 = lo1;
        c2 = xor3(hi0, c3, k1);
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0;
    o.v[1] = c1;
    o.v[2] = c2;
    o.v[3] = c3;
    return o;
}

EMX_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}

// 53-bit uniform in [0,1) from two words (same construction as MT19937 random_sample).
EMX_HD double u53(uint32_t a, uint32_t b) {
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// x / a for a power of two a is x * (1 / a) in every bit (both are the one rounding of the same real number); the reciprocal
// comes from the exponent field -- integer work on a launch-uniform value, scalar on the device -- and replaces the ~25
// instructions of an f64 division per plan entry (StretchMove's default a = 2).  False: a is not a power of two whose
// reciprocal is normal.
EMX_HD bool pow2_reciprocal(double a, double& inv) {
    uint64_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (uint64_t)__double_as_longlong(a);
#else
    __builtin_memcpy(&b, &a, 8);
#endif
    const uint64_t e = b >> 52;                         // sign included: a negative a fails the range test
    if ((b & 0x000fffffffffffffull) != 0 || e < 1 || e > 2045) return false;
    const uint64_t ib = (2046 - e) << 52;
#if defined(__HIP_DEVICE_COMPILE__)
    inv = __longlong_as_double((long long)ib);
#else
    __builtin_memcpy(&inv, &ib, 8);
#endif
    return true;
}

// Sizes of the S sub-ensembles of N walkers, `arange(N) % S` (red_blue.py:78): set s has (N - s + S - 1) / S = N / S + (s < N % S)
// members.  One division per plan entry instead of one per set looked at (the device has no scalar divide: each is ~25
// instructions through the vector unit and back).
struct SplitSizes {
    int q, r;
    EMX_HD int of(int s) const { return q + (s < r ? 1 : 0); }
};

EMX_HD SplitSizes split_sizes(int N, int S) {
    SplitSizes z;
    if (S == 2) {
        z.q = N >> 1;
        z.r = N & 1;
    } else {
        z.q = N / S;
        z.r = N - z.q * S;
    }
    return z;
}

// Unbiased-enough bounded integer in [0, n): 64-bit multiply-high (bias < n / 2^64).
EMX_HD uint64_t bounded64(uint32_t a, uint32_t b, uint64_t n) {
    // every caller's n is a walker count (< 2^32; emx_create refuses more than 2^31 walkers): the high 64 bits of the 64 x 64
    // product are then (a n + ((b n) >> 32)) >> 32 -- no overflow: a n + 2^32 - 1 <= 2^64 - 2^32 -- two wide multiplies
    // instead of the four of a general 64 x 64 multiply-high
    if ((n >> 32) == 0) {
        const uint64_t n32 = n;
        return ((uint64_t)a * n32 + (((uint64_t)b * n32) >> 32)) >> 32;
    }
    const uint64_t r = ((uint64_t)a << 32) | (uint64_t)b;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(r, n);
#else
    return (uint64_t)(((unsigned __int128)r * (unsigned __int128)n) >> 64);
#endif
}

// ---------------------------------------------------------------------------------------
// Keyed bijection on [0, n): the per-step random split.  Walker w belongs to
// sub-ensemble  perm(w) % nsplits  and is member number  perm(w) / nsplits  of it, which
// gives exactly the set sizes of `arange(n) % nsplits` shuffled (red_blue.py:78-80).
// Built from invertible k-bit mixers (xorshift / odd multiply-add), cycle-walked for
// non-power-of-two n.
// ---------------------------------------------------------------------------------------
struct PermKey {
    uint64_t n;
    uint32_t bits;       // k = ceil(log2(n)), >= 1
    uint32_t mask;       // 2^k - 1   (n <= 2^31)
    uint32_t m1, c1, m2, c2, m3, c3;         // odd multipliers / offsets
    uint32_t m1inv, m2inv, m3inv;            // inverses mod 2^k
    uint32_t s1, s2;                          // xorshift amounts
};

EMX_HD uint32_t perm_mix(uint32_t x, const PermKey& k) {
    x = (x * k.m1 + k.c1) & k.mask;
    x ^= x >> k.s1;
    x = (x * k.m2 + k.c2) & k.mask;
    x ^= x >> k.s2;
    x = (x * k.m3 + k.c3) & k.mask;
    x ^= x >> k.s1;
    return x;
}

EMX_HD uint32_t unxorshift(uint32_t y, uint32_t s, uint32_t bits) {
    uint32_t x = y;
    for (uint32_t sh = s; sh < bits; sh += s) x ^= y >> sh;
    return x;
}

// the low 32 bits of a product whose low k <= 24 bits are all that is kept: v_mul_u32_u24 (full rate; it reads the low 24 bits
// of each operand, which decide the low 24 of the product) in place of the quarter-rate v_mul_lo_u32
template <bool SMALL>
EMX_HD uint32_t mul_kept_bits(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (SMALL) return __umul24(a, b);
#endif
    return a * b;
}

// make_perm_key's shifts are s1 = ceil(bits / 2) and s2 = ceil(bits / 3), so unxorshift's loop has at most one resp. two trips
// and a value below 2^bits shifted by 2 s1 (3 s2) is zero: the closed forms below are the loop, without the loop -- as a loop
// with a launch-uniform, unknown trip count each of the six inversions of a plan entry cost ~40 scalar instructions (a
// division for the unroller's trip count among them) around its two vector ones.
template <bool SMALL>
EMX_HD uint32_t perm_unmix_t(uint32_t x, const PermKey& k) {
    x ^= x >> k.s1;                                                   // == unxorshift(x, k.s1, k.bits)
    x = mul_kept_bits<SMALL>(x - k.c3, k.m3inv) & k.mask;
    x = x ^ (x >> k.s2) ^ (x >> (2u * k.s2));                         // == unxorshift(x, k.s2, k.bits)   (2 s2 <= 22)
    x = mul_kept_bits<SMALL>(x - k.c2, k.m2inv) & k.mask;
    x ^= x >> k.s1;
    x = mul_kept_bits<SMALL>(x - k.c1, k.m1inv) & k.mask;
    return x;
}

EMX_HD uint32_t perm_unmix(uint32_t x, const PermKey& k) { return perm_unmix_t<false>(x, k); }

EMX_HD uint32_t perm_fwd(uint32_t w, const PermKey& k) {
    uint32_t x = perm_mix(w, k);
    while (x >= k.n) x = perm_mix(x, k);
    return x;
}

EMX_HD uint32_t perm_inv(uint32_t p, const PermKey& k) {
    if (k.bits <= 24) {                                   // uniform: ensembles of up to 2^24 walkers
        uint32_t x = perm_unmix_t<true>(p, k);
        while (x >= k.n) x = perm_unmix_t<true>(x, k);
        return x;
    }
    uint32_t x = perm_unmix_t<false>(p, k);
    while (x >= k.n) x = perm_unmix_t<false>(x, k);
    return x;
}

EMX_HD uint32_t modinv_pow2(uint32_t a) {  // a odd; inverse mod 2^32 (Newton)
    uint32_t x = a;
    for (int i = 0; i < 5; ++i) x *= 2u - a * x;
    return x;
}

EMX_HD PermKey make_perm_key(uint64_t n, uint64_t seed, uint64_t step) {
    PermKey k{};
    k.n = n;
    uint32_t bits = 1;
    while ((1ull << bits) < n) ++bits;
    k.bits = bits;
    k.mask = bits >= 32 ? 0xffffffffu : (uint32_t)((1ull << bits) - 1);
    const Philox4 a = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), 0x5045524du /*'PERM'*/, 0, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    const Philox4 b = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), 0x5045524du, 1, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    k.m1 = a.v[0] | 1u;
    k.c1 = a.v[1];
    k.m2 = a.v[2] | 1u;
    k.c2 = a.v[3];
    k.m3 = b.v[0] | 1u;
    k.c3 = b.v[1];
    // multipliers ~ golden-ratio-like: force some high/low structure so that tiny k still mixes
    k.m1 = (k.m1 & ~6u) | 4u | 1u;  // == 5 mod 8: maximal multiplicative order
    k.m2 = (k.m2 & ~6u) | 4u | 1u;
    k.m3 = (k.m3 & ~6u) | 4u | 1u;
    k.m1inv = modinv_pow2(k.m1);
    k.m2inv = modinv_pow2(k.m2);
    k.m3inv = modinv_pow2(k.m3);
    k.s1 = bits > 1 ? (bits + 1) / 2 : 1;
    k.s2 = bits > 2 ? (bits + 2) / 3 : 1;
    return k;
}

}  // namespace emx
