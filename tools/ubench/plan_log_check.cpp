// Accuracy of plan_log (csrc/emx_planlog.hpp) against the 80-bit logl, in ulps of a double: acceptance uniforms, stretch factors, every
// exponent, the table intervals' boundaries, the neighbourhood of 1.  usage: plan_log_check [samples per family]; exit 1 when the worst error
// reaches 0.55 ulp.  Built and run by tests/test_plan_log.py (g++ -O2 -ffp-contract=off -mfma).
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <random>
#include "emx_planlog.hpp"
using namespace emx;
static double worst = 0, worst_x = 0; static long n = 0;
static void chk(double x) {
    const double y = plan_log_tab(x, h_plan_log_rows);
    const long double t = logl((long double)x);
    if (x == 0) { if (!(std::isinf(y) && y < 0)) { printf("log 0 wrong\n"); } return; }
    // error in ulps of the double nearest the true value
    int ex; (void)frexpl(t, &ex);
    const double ul = t == 0 ? 4.9e-324 : std::ldexp(1.0, ex - 53);          // spacing of doubles in t's binade
    const double e = (double)(fabsl((long double)y - t) / (long double)ul);
    ++n;
    if (e > worst) { worst = e; worst_x = x; }
}
int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 20000000;
    std::mt19937_64 g(12345);
    for (long i = 0; i < N; ++i) {                       // acceptance uniforms
        const double u = (double)(g() >> 11) * (1.0 / 9007199254740992.0);
        chk(u);
        const double a = (i & 1) ? 2.0 : 1.0 + (double)(g() >> 11) * (4.0 / 9007199254740992.0);
        const double t = (a - 1.0) * u + 1.0;            // stretch factors
        chk(t * t / a);
    }
    printf("uniforms + stretch: n %ld worst %.4f ulp at %a\n", n, worst, worst_x);
    for (long i = 0; i < N; ++i) {                       // any exponent
        const int e = (int)(g() % 2000) - 1000;
        const double m = 1.0 + (double)(g() >> 12) * (1.0 / 4503599627370496.0);
        chk(std::ldexp(m, e));
    }
    printf("+ all exponents: n %ld worst %.4f ulp at %a\n", n, worst, worst_x);
    for (int i = 0; i <= 128; ++i)                       // interval boundaries, both binades, and their neighbours
        for (int s = -2000; s <= 2000; ++s) {
            double b = i < 80 ? 0.6875 + i / 256.0 : 1.0 + (i - 80) / 128.0;
            double x = b;
            for (int q = 0; q < std::abs(s); ++q) x = std::nextafter(x, s > 0 ? 4.0 : 0.0);
            for (int e = -3; e <= 3; ++e) chk(std::ldexp(x, e));
        }
    printf("+ boundaries: n %ld worst %.4f ulp at %a\n", n, worst, worst_x);
    for (long i = 0; i < N; ++i) {                       // near 1, log-uniform distance
        const double d = std::ldexp(1.0 + (double)(g() >> 12) * (1.0 / 4503599627370496.0), -(int)(g() % 52) - 1);
        chk(1.0 + d);
        chk(1.0 - d);
    }
    chk(0.0); chk(1.0); chk(0x1p-53); chk(1.0 - 0x1p-53); chk(0x1p-1022); chk(1.7976931348623157e308);
    printf("+ near one: n %ld worst %.4f ulp at %a\n", n, worst, worst_x);
    printf("log(1) = %a  log(2^-53) = %.17g (libm %.17g)\n", plan_log_tab(1.0, h_plan_log_rows), plan_log_tab(0x1p-53, h_plan_log_rows), std::log(0x1p-53));
    return worst < 0.55 ? 0 : 1;
}
