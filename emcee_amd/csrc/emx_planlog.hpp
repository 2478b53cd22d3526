// plan_log: the natural logarithm the plan entries take -- log(u) of the acceptance uniform (red_blue.py:100) and log(zz) of the
// stretch factor (stretch.py:33).  plan_log_tab: positive, finite, NORMAL arguments and zero (log 0 = -inf; a uniform can be 0).
//
// The device library's log spends ~84 vector instructions (double-double arithmetic for arguments it never sees here:
// subnormals, negatives, NaN) and k_native_plan_batch is bound by instruction issue with a third of it in its two logarithms.
// This one takes ~32: a 128-interval table (emx_logtab.hpp, made by tools/gen_logtab.py) and a short series.
//
//   x = 2^k z,  z in [0.6875, 1.375);   i = the top seven mantissa bits behind that offset;   r = z invc_i - 1 = rh + rl EXACTLY
//   log x = (k LN2HI + logc_hi_i)  +  rh  +  [ rl + (k LN2LO + logc_lo_i) + r^2 (-1/2 + r/3 - ... ) ]
//
// k LN2HI + logc_hi is exact (both on the 2^-42 grid, |sum| < 2^10); its sum with rh is carried as hi + lo; |r| <= 2^-8.4 except
// in the two intervals around 1, where invc = 1, log c = 0, |r| < 2^-7 and the series is the result.  r is kept exact (product
// and its fma remainder) because next to those intervals the result is as small as r itself: one rounding of r there is a
// quarter ulp of the result.  The series runs to r^8 / 8 (first omitted term: 2^-59 of the result in the unit intervals,
// nothing elsewhere).  Measured against the 80-bit logl: tests/test_plan_log.py (it asks for < 0.55 ulp).
//
// Every kernel that makes a plan entry's logarithms calls this (the native plan kernel, the in-kernel plans of the small-ensemble
// kernels, the exact mode's conversions, the split-phase commit), so that two routes to the same step agree in every bit.
#pragma once
#include <cstdint>
#include <cstring>

#include "emx_logtab.hpp"

#if defined(__HIPCC__)
#define EMX_PL_HD __host__ __device__ __forceinline__
#else
#define EMX_PL_HD inline
#endif

namespace emx {

struct alignas(32) LogRow {
    double invc, logc_hi, logc_lo, pad;
};

#if defined(__HIPCC__)
static __device__ const LogRow g_plan_log_rows[128] = {EMX_LOGTAB_ROWS};
#endif
static const LogRow h_plan_log_rows[128] = {EMX_LOGTAB_ROWS};

EMX_PL_HD double plan_log_tab(double x, const LogRow* __restrict__ tab) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t xl = (uint32_t)__double2loint(x), xh = (uint32_t)__double2hiint(x);
#else
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    const uint32_t xl = (uint32_t)bits, xh = (uint32_t)(bits >> 32);
#endif
    const uint32_t tmp = xh - 0x3fe60000u;
    const int i = (int)((tmp >> 13) & 127u);
    const int k = (int32_t)tmp >> 20;
    const uint32_t zh = xh - (tmp & 0xfff00000u);
#if defined(__HIP_DEVICE_COMPILE__)
    const double z = __hiloint2double((int)zh, (int)xl);
#else
    const uint64_t zb = ((uint64_t)zh << 32) | xl;
    double z;
    std::memcpy(&z, &zb, 8);
#endif
    const LogRow row = tab[i];
    const double kd = (double)k;
    const double ph = z * row.invc;
    const double rl = __builtin_fma(z, row.invc, -ph);                      // z invc = ph + rl
    const double rh = ph - 1.0;                                             // exact (ph in [0.99, 1.01])
    const double r = rh + rl;
    const double w = __builtin_fma(kd, EMX_LOGTAB_LN2HI, row.logc_hi);      // exact
    const double hi = w + rh;
    const double lo = (w - hi) + rh;                                        // exact: |w| >= |rh| or w == 0
    const double r2 = r * r;
    double p = __builtin_fma(r, -1.0 / 8.0, 1.0 / 7.0);
    p = __builtin_fma(r, p, -1.0 / 6.0);
    p = __builtin_fma(r, p, 1.0 / 5.0);
    p = __builtin_fma(r, p, -1.0 / 4.0);
    p = __builtin_fma(r, p, 1.0 / 3.0);
    p = __builtin_fma(r, p, -0.5);
    const double tail = __builtin_fma(kd, EMX_LOGTAB_LN2LO, row.logc_lo) + rl;
    const double y = hi + (__builtin_fma(r2, p, lo) + tail);
    return x == 0.0 ? -__builtin_inf() : y;
}

#if defined(__HIPCC__)
// a uniform of [0, 1) on the 2^-53 grid (u53, mt_pair_double): zero or normal by construction
__device__ __forceinline__ double plan_log_uniform(double u) { return plan_log_tab(u, g_plan_log_rows); }

// anything else a plan may carry (a caller's own plan, a stretch factor of an odd `a`): negative, subnormal, infinite and NaN
// arguments go to the device library's log, whose answers for them are the reference's (NaN, -inf, inf)
__device__ __forceinline__ double plan_log(double x) {
    const uint32_t xh = (uint32_t)__double2hiint(x);
    if (xh - 0x00100000u < 0x7fe00000u) return plan_log_tab(x, g_plan_log_rows);      // positive, normal, finite
    return log(x);
}
#endif

}  // namespace emx
