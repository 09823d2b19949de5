// Accuracy of the portable libm (dynearthsol_amd/csrc/des_libm.hpp) against x87 long double
// (64-bit significand: the yardstick's own error is < 2^-10 ulp of a double), with glibc's
// double functions measured beside it.
//   g++ -O2 -ffp-contract=off -Idynearthsol_amd/csrc tools/libm_accuracy.cpp -o /tmp/libm_accuracy && /tmp/libm_accuracy [samples]
#include "des_libm.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <random>

static double ulp_err(double got, long double want)
{
    if (std::isnan(got) && std::isnan((double)want)) return 0;
    if (std::isinf(got) || std::isinf((double)want)) return got == (double)want ? 0 : 1e9;
    int e;
    std::frexp((double)want, &e);                      // want = m 2^e, 0.5 <= |m| < 1
    if (e < -1021) e = -1021;                          // subnormal range: fixed spacing
    long double u = std::ldexp(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / u);
}

struct Stat { double mine = 0, libc = 0; double ax = 0, ay = 0; long same = 0, total = 0; };

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 4000000;
    std::mt19937_64 rng(12345);
    auto uni = [&](double a, double b) { return a + (b - a) * (rng() >> 11) * 0x1p-53; };
    auto logu = [&](double a, double b) { return std::exp(uni(std::log(a), std::log(b))); };
    Stat s;
    auto rep = [&](const char *name) {
        printf("%-34s max ulp error: portable %.4f (at %.17g, %.17g)  glibc %.4f  same bits as glibc: %.3f %%\n", name, s.mine, s.ax, s.ay, s.libc,
               100.0 * s.same / s.total);
        s = Stat();
    };
    auto acc = [&](double got, double libc, long double want, double x, double y) {
        double e = ulp_err(got, want);
        if (e > s.mine) { s.mine = e; s.ax = x; s.ay = y; }
        s.total++;
        if (got == libc) s.same++;
        double l = ulp_err(libc, want);
        if (l > s.libc) s.libc = l;
    };
    // pow on the creep-law ranges: strain-rate invariant 1e-25..1e-8, exponent 1/n - 1
    for (long i = 0; i < n; ++i) { double x = logu(1e-25, 1e-8), y = uni(-1, 0); acc(deslibm::pow(x, y), std::pow(x, y), powl(x, y), x, y); }
    rep("pow(edot, 1/n-1)");
    for (long i = 0; i < n; ++i) { double x = logu(1e-40, 1e10), y = uni(-2, 2); acc(deslibm::pow(x, y), std::pow(x, y), powl(x, y), x, y); }
    rep("pow(1e-40..1e10, -2..2)");
    for (long i = 0; i < n; ++i) { double x = logu(1e-300, 1e300), y = uni(-1.02, 1.02); acc(deslibm::pow(x, y), std::pow(x, y), powl(x, y), x, y); }
    rep("pow(1e-300..1e300, -1..1)");
    for (long i = 0; i < n; ++i) { double x = uni(0.99, 1.01), y = logu(1, 6e4) * (i & 1 ? 1 : -1); acc(deslibm::pow(x, y), std::pow(x, y), powl(x, y), x, y); }
    rep("pow(0.99..1.01, +-1..6e4)");
    for (long i = 0; i < n; ++i) { double x = uni(0.5, 2.0), y = uni(-1000, 1000); acc(deslibm::pow(x, y), std::pow(x, y), powl(x, y), x, y); }
    rep("pow(0.5..2, -1000..1000)");
    for (long i = 0; i < n; ++i) { double x = logu(4.9e-324, 2.2e-308), y = uni(-1, 1); acc(deslibm::pow(x, y), std::pow(x, y), powl(x, y), x, y); }
    rep("pow(subnormal, -1..1)");
    for (long i = 0; i < n; ++i) { double x = logu(1e-5, 1e5), y = uni(-150, 150); acc(deslibm::pow(x, y), std::pow(x, y), powl(x, y), x, y); }
    rep("pow(1e-5..1e5, -150..150) incl. over/underflow");
    for (long i = 0; i < n; ++i) { double x = uni(0, 200); acc(deslibm::exp(x), std::exp(x), expl(x), x, 0); }
    rep("exp(0..200)  [E/nRT]");
    for (long i = 0; i < n; ++i) { double x = uni(-745, 709.7); acc(deslibm::exp(x), std::exp(x), expl(x), x, 0); }
    rep("exp(-745..709.7)");
    for (long i = 0; i < n; ++i) { double x = logu(1e-20, 1) * (i & 1 ? 1 : -1); acc(deslibm::exp(x), std::exp(x), expl(x), x, 0); }
    rep("exp(+-1e-20..1)");
    for (long i = 0; i < n; ++i) { double x = uni(-3.2, 3.2); acc(deslibm::sin(x), std::sin(x), sinl(x), x, 0); }
    rep("sin(-pi..pi)");
    for (long i = 0; i < n; ++i) { double x = uni(-3.2, 3.2); acc(deslibm::cos(x), std::cos(x), cosl(x), x, 0); }
    rep("cos(-pi..pi)");
    for (long i = 0; i < n; ++i) { double x = uni(-1e5, 1e5); acc(deslibm::sin(x), std::sin(x), sinl(x), x, 0); }
    rep("sin(-1e5..1e5)");
    for (long i = 0; i < n; ++i) { double x = uni(-1e5, 1e5); acc(deslibm::cos(x), std::cos(x), cosl(x), x, 0); }
    rep("cos(-1e5..1e5)");
    for (long i = 0; i < n; ++i) { double x = logu(1e-30, 1); acc(deslibm::sin(x), std::sin(x), sinl(x), x, 0); }
    rep("sin(1e-30..1)");
    for (long i = 0; i < n; ++i) { double x = uni(0, 1.5533); acc(deslibm::tan(x), std::tan(x), tanl(x), x, 0); }
    rep("tan(0..89 deg)");
    for (long i = 0; i < n; ++i) { double y = uni(-10, 10), x = uni(-10, 10); acc(deslibm::atan2(y, x), std::atan2(y, x), atan2l(y, x), y, x); }
    rep("atan2(-10..10, -10..10)");
    for (long i = 0; i < n; ++i) { double y = logu(1e-30, 1e30), x = logu(1e-30, 1e30) * (i & 1 ? 1 : -1); acc(deslibm::atan2(y, x), std::atan2(y, x), atan2l(y, x), y, x); }
    rep("atan2(1e-30..1e30, +-1e-30..1e30)");
    // special values
    const double inf = INFINITY, nan = NAN;
    double sp[] = {0.0, -0.0, 1.0, -1.0, 0.5, 2.0, inf, -inf, nan, 1e-310, 1e308, 3.0};
    int bad = 0;
    for (double x : sp) for (double y : sp) {
        double a = deslibm::pow(x, y), b = std::pow(x, y);
        bool same = (std::isnan(a) && std::isnan(b)) || a == b;
        if (!same && !(x < 0 || (x == 0 && std::signbit(x)))) { printf("pow(%g, %g): portable %g glibc %g\n", x, y, a, b); ++bad; }
        a = deslibm::atan2(x, y); b = std::atan2(x, y);
        same = (std::isnan(a) && std::isnan(b)) || (a == b && std::signbit(a) == std::signbit(b));
        if (!same && ulp_err(a, atan2l(x, y)) > 1.5) { printf("atan2(%g, %g): portable %.17g glibc %.17g\n", x, y, a, b); ++bad; }
    }
    for (double x : sp) {
        double a = deslibm::exp(x), b = std::exp(x);
        if (!((std::isnan(a) && std::isnan(b)) || a == b)) { printf("exp(%g): portable %g glibc %g\n", x, a, b); ++bad; }
    }
    printf("special values: %d mismatches (x >= 0)\n", bad);
    return bad != 0;
}
