// GF(2) polynomial arithmetic behind the MT19937 jump-ahead (see emx_mtjump.hpp).  Plain host C++.
#include "emx_mtjump.hpp"

#include <cstring>
#include <map>
#include <mutex>

namespace emx {
namespace {

constexpr int W64 = (MT_DEG + 63) / 64 + 1;          // 64-bit words of a polynomial of degree <= 19937 (313)
constexpr int W64_2 = 2 * W64 + 2;                   // ... of a product of two of them

using Poly = std::vector<uint64_t>;

inline bool bit(const uint64_t* p, int i) { return (p[i >> 6] >> (i & 63)) & 1u; }
inline void flip(uint64_t* p, int i) { p[i >> 6] ^= 1ull << (i & 63); }

// dst ^= src << s, src of nsrc words, dst long enough
inline void xor_shifted(uint64_t* dst, const uint64_t* src, int nsrc, int s) {
    const int ws = s >> 6, bs = s & 63;
    if (bs == 0) {
        for (int k = 0; k < nsrc; ++k) dst[ws + k] ^= src[k];
        return;
    }
    uint64_t carry = 0;
    for (int k = 0; k < nsrc; ++k) {
        dst[ws + k] ^= (src[k] << bs) | carry;
        carry = src[k] >> (64 - bs);
    }
    dst[ws + nsrc] ^= carry;
}

struct Modulus {
    Poly phi;                        // W64 words, degree MT_DEG
    std::vector<int> support;        // exponents of phi below MT_DEG when phi is sparse (else empty)
    // p (W64_2 words, degree < 2 * MT_DEG) -> p mod phi in the low W64 words
    void reduce(uint64_t* p) const {
        if (!support.empty()) {
            // t^d = sum_{e in support} t^(d - DEG + e): clear the high bits from the top, word by word
            for (int d = 2 * MT_DEG; d >= MT_DEG; --d) {
                if (!bit(p, d)) continue;
                flip(p, d);
                const int base = d - MT_DEG;
                for (int e : support) flip(p, base + e);
            }
            return;
        }
        for (int d = 2 * MT_DEG; d >= MT_DEG; --d)
            if (bit(p, d)) xor_shifted(p, phi.data(), W64, d - MT_DEG);
    }
};

void mul(const uint64_t* a, const uint64_t* b, uint64_t* out /* W64_2, zeroed here */) {
    std::memset(out, 0, sizeof(uint64_t) * W64_2);
    // 64 shifted copies of b would save the per-bit shifting; the set bits of a are processed word-wise instead: for every bit
    // position s inside a word, gather the words of a that have bit s set and add b << s at those word offsets
    std::vector<uint64_t> bs((size_t)W64 + 1);
    for (int s = 0; s < 64; ++s) {
        if (s == 0) {
            for (int k = 0; k < W64; ++k) bs[k] = b[k];
            bs[W64] = 0;
        } else {
            uint64_t carry = 0;
            for (int k = 0; k < W64; ++k) {
                bs[k] = (b[k] << s) | carry;
                carry = b[k] >> (64 - s);
            }
            bs[W64] = carry;
        }
        for (int w = 0; w < W64; ++w)
            if ((a[w] >> s) & 1u) {
                uint64_t* d = out + w;
                const uint64_t* src = bs.data();
                for (int k = 0; k <= W64; ++k) d[k] ^= src[k];
            }
    }
}

void square(const uint64_t* a, uint64_t* out /* W64_2 */) {
    std::memset(out, 0, sizeof(uint64_t) * W64_2);
    for (int w = 0; w < W64; ++w) {
        uint64_t x = a[w], lo = 0, hi = 0;
        // spread the 64 bits of x over 128: bit i -> bit 2 i
        auto spread32 = [](uint64_t v) {
            v &= 0xffffffffull;
            v = (v | (v << 16)) & 0x0000ffff0000ffffull;
            v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
            v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
            v = (v | (v << 2)) & 0x3333333333333333ull;
            v = (v | (v << 1)) & 0x5555555555555555ull;
            return v;
        };
        lo = spread32(x);
        hi = spread32(x >> 32);
        out[2 * w] = lo;
        out[2 * w + 1] = hi;
    }
}

// MT19937 output bit sequence for Berlekamp-Massey: bit 0 of the untempered words of a generator started from a fixed state
struct Gen {
    uint32_t key[2][MT_N];
    int cur = 0, pos = MT_N;
    Gen() {
        uint32_t s = 19650218u;                         // init_genrand (any non-degenerate state does)
        for (int i = 0; i < MT_N; ++i) {
            key[0][i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1);
        }
    }
    uint32_t next() {
        if (pos == MT_N) {
            mt_twist_block(key[cur], key[cur ^ 1]);
            cur ^= 1;
            pos = 0;
        }
        return key[cur][pos++];
    }
};

std::mutex g_mu;
bool g_phi_done = false, g_phi_ok = false;
Modulus g_mod;

bool char_poly_locked() {
    if (g_phi_done) return g_phi_ok;
    g_phi_done = true;
    const int NB = 2 * MT_DEG + 64;
    std::vector<uint8_t> s((size_t)NB);
    {
        Gen g;
        for (int i = 0; i < 2 * MT_N; ++i) g.next();     // past the seed block: every word from here on is a product of the recurrence
        for (int i = 0; i < NB; ++i) s[i] = (uint8_t)(g.next() & 1u);
    }
    // Berlekamp-Massey over GF(2); C, B as bit sets; win bit i = s[n - i]
    const int WW = 2 * ((NB >> 6) + 2) + 4;           // room for B << m whatever the (bounded by n) degrees are
    Poly C((size_t)WW, 0), B((size_t)WW, 0), T((size_t)WW, 0), win((size_t)WW, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < NB; ++n) {
        // win <<= 1; win |= s[n]
        {
            const int used = (n >> 6) + 2 < WW ? (n >> 6) + 2 : WW;
            uint64_t carry = s[n];
            for (int k = 0; k < used; ++k) {
                const uint64_t nx = win[k] >> 63;
                win[k] = (win[k] << 1) | carry;
                carry = nx;
            }
        }
        const int lw = (L >> 6) + 1;
        uint64_t acc = 0;
        for (int k = 0; k < lw; ++k) acc ^= C[k] & win[k];
        const int d = __builtin_parityll(acc);
        if (!d) {
            ++m;
        } else if (2 * L <= n) {
            T = C;
            xor_shifted(C.data(), B.data(), (n >> 6) + 2, m);
            L = n + 1 - L;
            B = T;
            m = 1;
        } else {
            xor_shifted(C.data(), B.data(), (n >> 6) + 2, m);
            ++m;
        }
    }
    if (L != MT_DEG) return g_phi_ok = false;
    // connection polynomial C (s_n = sum_{i>=1} c_i s_{n-i}) -> characteristic polynomial phi_i = c_{L - i}
    g_mod.phi.assign((size_t)W64, 0);
    int terms = 0;
    for (int i = 0; i <= L; ++i)
        if (bit(C.data(), L - i)) {
            flip(g_mod.phi.data(), i);
            ++terms;
        }
    g_mod.support.clear();
    if (terms <= 2048)
        for (int i = 0; i < L; ++i)
            if (bit(g_mod.phi.data(), i)) g_mod.support.push_back(i);
    return g_phi_ok = true;
}

struct JumpTable {
    int count = 0;
    std::vector<uint32_t> words;       // count x MT_POLY_WORDS
};
std::map<uint64_t, JumpTable> g_tables;

void to_words32(const uint64_t* p, uint32_t* out) {
    for (int w = 0; w < MT_POLY_WORDS; ++w) out[w] = (uint32_t)(p[w >> 1] >> ((w & 1) * 32));
}

}  // namespace

void mt_twist_block(const uint32_t* o, uint32_t* n) {
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
    for (int kk = 0; kk < 227; ++kk) {
        const uint32_t y = (o[kk] & UPPER) | (o[kk + 1] & LOWER);
        n[kk] = o[kk + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
    }
    for (int kk = 227; kk < 623; ++kk) {
        const uint32_t y = (o[kk] & UPPER) | (o[kk + 1] & LOWER);
        n[kk] = n[kk - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
    }
    const uint32_t y = (o[623] & UPPER) | (n[0] & LOWER);
    n[623] = n[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX);
}

bool mt_char_poly(std::vector<uint32_t>& phi) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!char_poly_locked()) return false;
    phi.assign((size_t)MT_POLY_WORDS + 1, 0);
    for (int i = 0; i <= MT_DEG; ++i)
        if (bit(g_mod.phi.data(), i)) phi[i >> 5] |= 1u << (i & 31);
    return true;
}

bool mt_jump_polys(uint64_t stride_words, int count, const uint32_t** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!char_poly_locked() || count < 1 || stride_words < 1) return false;
    JumpTable& tb = g_tables[stride_words];
    if (tb.count >= count) {
        *out = tb.words.data();
        return true;
    }
    Poly g1((size_t)W64, 0), prod((size_t)W64_2, 0);
    if (tb.count >= 1) {
        for (int w = 0; w < MT_POLY_WORDS; ++w) g1[w >> 1] |= (uint64_t)tb.words[w] << ((w & 1) * 32);
    } else {
        // t^stride mod phi by square and multiply (multiplying by t is a shift)
        Poly r((size_t)W64, 0);
        r[0] = 1;
        int top = 63;
        while (top > 0 && !((stride_words >> top) & 1u)) --top;
        for (int b = top; b >= 0; --b) {
            square(r.data(), prod.data());
            g_mod.reduce(prod.data());
            std::memcpy(r.data(), prod.data(), sizeof(uint64_t) * W64);
            if ((stride_words >> b) & 1u) {
                std::memset(prod.data(), 0, sizeof(uint64_t) * W64_2);
                xor_shifted(prod.data(), r.data(), W64, 1);
                g_mod.reduce(prod.data());
                std::memcpy(r.data(), prod.data(), sizeof(uint64_t) * W64);
            }
        }
        g1 = r;
    }
    std::vector<uint32_t> words((size_t)count * MT_POLY_WORDS);
    if (tb.count > 0) std::memcpy(words.data(), tb.words.data(), sizeof(uint32_t) * (size_t)tb.count * MT_POLY_WORDS);
    Poly cur((size_t)W64, 0);
    int have = tb.count;
    if (have == 0) {
        to_words32(g1.data(), words.data());
        have = 1;
    }
    for (int w = 0; w < MT_POLY_WORDS; ++w) cur[w >> 1] |= (uint64_t)words[(size_t)(have - 1) * MT_POLY_WORDS + w] << ((w & 1) * 32);
    for (; have < count; ++have) {
        mul(cur.data(), g1.data(), prod.data());
        g_mod.reduce(prod.data());
        std::memcpy(cur.data(), prod.data(), sizeof(uint64_t) * W64);
        to_words32(cur.data(), words.data() + (size_t)have * MT_POLY_WORDS);
    }
    tb.words.swap(words);
    tb.count = count;
    *out = tb.words.data();
    return true;
}

void mt_apply_jump(const uint32_t* g, const uint32_t* window, uint32_t* out) {
    for (int j = 0; j < MT_N; ++j) out[j] = 0;
    for (int i = 0; i < MT_DEG; ++i)
        if ((g[i >> 5] >> (i & 31)) & 1u)
            for (int j = 0; j < MT_N; ++j) out[j] ^= window[i + j];
}

}  // namespace emx
