// MT19937 jump-ahead: the GF(2) polynomials that let many workgroups generate disjoint segments of ONE NumPy-legacy stream.
//
// Reference emcee draws everything from one serial MT19937 stream (ensemble.py:166-167,406, moves/red_blue.py:80,100,
// moves/stretch.py:30-32).  The recurrence x[a + 624] = x[a + 397] ^ A(x[a] | x[a + 1]) is linear over GF(2), so the
// generator's state `steps` words further on is a fixed linear function of the state now: with phi(t) the characteristic
// polynomial of the recurrence (degree 19937) and g(t) = t^steps mod phi(t) = sum g_i t^i,
//
//     x[a + steps + j] = XOR over { i : g_i = 1 } of x[a + i + j]          (every a past the seed block, every j)
//
// -- a convolution of g with a window of 19937 + 624 consecutive (untempered) words.  (Haramoto, Matsumoto, Nishimura, Panneton,
// L'Ecuyer, "Efficient jump ahead for F2-linear random number generators", INFORMS J. Comput. 20 (2008) -- the sliding-window
// form of their polynomial method; phi is found here with Berlekamp-Massey from the generator's own output rather than quoted.)
// The device evaluates the convolution (k_mt_jump, emx_mtdev.hip); this file computes the polynomials once per process, on the
// host: g_k = t^(k * stride) mod phi for k = 1 .. count.  tests/test_mtdev_cpu.py checks jumped states against the stepped
// generator word for word.
#pragma once
#include <cstdint>
#include <vector>

namespace emx {

constexpr int MT_N = 624;
constexpr int MT_DEG = 19937;                       // degree of the characteristic polynomial
constexpr int MT_POLY_WORDS = (MT_DEG + 31) / 32;   // 624 32-bit words hold a polynomial of degree < 19937 (bit i of word i / 32 = g_i)
constexpr int MT_WINDOW = MT_DEG + MT_N;            // words of the window the convolution reads (20 561)

// one twist of a whole block, out of place (the same map as MT19937Legacy::twist)
void mt_twist_block(const uint32_t* old_key, uint32_t* new_key);

// phi(t), bit i of word i / 32 = coefficient of t^i, MT_DEG + 1 bits (MT_POLY_WORDS + 1 words).  Computed once (Berlekamp-Massey
// over 2 * 19937 + 64 output bits), cached for the process.  Returns false if the linear complexity found is not 19937.
bool mt_char_poly(std::vector<uint32_t>& phi);

// g_k = t^(k * stride_words) mod phi for k = 1 .. count, MT_POLY_WORDS words each, consecutive in `out` (resized).
// Cached per (stride_words, count' >= count) for the process; thread safe.  Returns false on an internal inconsistency.
bool mt_jump_polys(uint64_t stride_words, int count, const uint32_t** out);

// host reference of the device convolution: state `steps` words after the window start.  window[0 .. MT_WINDOW) are consecutive
// untempered words (all produced by the recurrence, i.e. not the seed block); out[j] = x[steps + j], j < 624
void mt_apply_jump(const uint32_t* g, const uint32_t* window, uint32_t* out);

}  // namespace emx
