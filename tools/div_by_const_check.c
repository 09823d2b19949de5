/* tools/div_by_const_check.c -- could the fp64 divisions by the constants 3 and 6 on the hot path (tet volume / 6, traces / 3,
 * second invariants: ~9 per element in the evp stress update, ~12 instructions each on gfx950) be replaced by q = x * RN(1/c),
 * r = fma(-c, q, x), q' = fma(r, RN(1/c), q) (with q kept where r == 0: the sign of zero) and still return the correctly rounded
 * quotient the CPU reference computes?  6e8 bit patterns per constant, every exponent range incl. denormals, tiny and huge:
 *   c = 3: no mismatch;  c = 6: 15.6M mismatches, all with DENORMAL quotients (double rounding) -- so a range guard is needed,
 *   and with it the sequence saves 3-4 of 12 instructions per division: ~1 % of a step.  Not built.
 *   gcc -O2 -mfma -ffp-contract=off -o /tmp/divc tools/div_by_const_check.c -lm && /tmp/divc        (any x86-64 CPU with FMA, ~80 s)
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline double divc(double x, double c, double y) { double q = x * y; double r = fma(-c, q, x); double q1 = fma(r, y, q); return r == 0.0 ? q : q1; }
static uint64_t s = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main() {
    const double cs[] = {3.0, 6.0};
    for (int k = 0; k < 2; ++k) {
        const double c = cs[k], y = 1.0 / c;
        long bad = 0, n = 0;
        // every bit pattern class: random 64-bit patterns (all exponents incl. denormals, inf, nan), denormals exhaustively sampled
        for (long i = 0; i < 600000000L; ++i) {
            uint64_t b = rnd();
            if ((i & 3) == 1) b &= 0x800fffffffffffffull;                       // denormals / zero
            if ((i & 3) == 2) b = (b & 0x800fffffffffffffull) | ((uint64_t)(rnd() % 8) << 52);   // tiny normals
            if ((i & 3) == 3) b = (b & 0x800fffffffffffffull) | ((uint64_t)(2040 + rnd() % 7) << 52);   // huge
            double x; memcpy(&x, &b, 8);
            double t = x / c, d = divc(x, c, y);
            if (isnan(t) && isnan(d)) continue;
            ++n;
            if (memcmp(&t, &d, 8)) { if (bad < 6) printf("c=%g x=%a true %a got %a\n", c, x, t, d); ++bad; }
        }
        double z = -0.0, t = z / c, d = divc(z, c, y);
        printf("c = %g: %ld mismatches of %ld; -0: %s\n", c, bad, n, memcmp(&t, &d, 8) ? "BAD" : "ok");
    }
    return 0;
}
