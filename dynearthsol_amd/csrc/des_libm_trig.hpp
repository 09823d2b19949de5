// des_libm_trig.hpp -- sin, cos, tan, atan2 that return the bits of the C library the CPU reference runs on.
//
// Third-party notice: the routines below restate sysdeps/ieee754/dbl-64/s_sin.c, s_sincos.c, s_tan.c and e_atan2.c of the
// GNU C Library 2.35 -- the IBM Accurate Mathematical Library, written by International Business Machines Corp.,
// Copyright (C) 2001-2022 Free Software Foundation, Inc., licensed under the GNU Lesser General Public License,
// version 2.1 or (at your option) any later version.  This file and des_libm_trig_tables.hpp are a derived work of
// those sources and are distributed under the same licence (LGPL-2.1-or-later); see THIRD_PARTY_NOTICES.md at the
// repository root.  No warranty; see the licence for details.
//
// Like exp / pow in des_libm.hpp these are NOT designs of our own: they restate, operation by operation, the
// routines of glibc 2.35 (sysdeps/ieee754/dbl-64/s_sin.c, s_tan.c, e_atan2.c -- the IBM Accurate Mathematical
// Library, in 2.35 without its multi-precision slow paths) in the form the C library executes on an x86-64 host
// with FMA (__sin_fma, __cos_fma, __tan_fma, __ieee754_atan2_fma, which the ifunc resolver picks on any CPU with
// FMA + AVX2): where gcc contracted a product and a sum of the C source into one fused operation there is an
// explicit fma below, read off the shipped object code (libm-2.35.a), and nowhere else.  The tables are the C
// library's own numbers (des_libm_trig_tables.hpp).  The reference calls these functions in plastic_props
// (matprops.cxx:589-606: sin, tan of the friction / dilation angle) and in the Kopp solver (3x3-C/dsyevc3.c:60-70:
// atan2, cos, sin); with them the device returns what the CPU build computes also for models that yield.
//
// Range: sin / cos for |x| < 105414350 (s_sin.c's limit of its own reduction; beyond it glibc calls __branred),
// tan for |x| <= 1e8 (s_tan.c's own reductions, the longer one beyond 25 included; past 1e8 glibc calls __branred),
// atan2 for finite non-zero arguments whose ratio needs no rescaling; des_libm.hpp falls back to its own routines outside (never reached from the stress update:
// angles below pi/2, Cardano's phi in [0, pi/3]).  tests/test_libm.py sweeps them against the host's libm.
#pragma once

#include "des_libm_trig_tables.hpp"

namespace deslibm {

// a * b + c as the object code has it: one fused operation in the *_fma builds (F), a rounded product and a sum in the
// plain build (the double-precision sincos has no FMA variant in glibc 2.35: libm.a holds s_sincos.o only)
template <bool F> DES_LIBM_FN double g_mad(double a, double b, double c) { return F ? fma_(a, b, c) : a * b + c; }

// ---- s_sin.c ---------------------------------------------------------------------------------------------------
// do_sin / do_cos: sin, cos of x + dx from the table entry of the nearest k/128 and short Taylor series of the rest
template <bool F> DES_LIBM_FN double g_do_sin(double x, double dx)
{
    const double xold = x;
    const double big = 52776558133248.0;
    const double sn3 = -0.16666666666666488, sn5 = 0.008333332142857223;
    const double cs2 = 0.5, cs4 = -0.04166666666666644, cs6 = 0.001388888740079376;
    if (!(x > 0)) dx = -dx;                                  // if (x <= 0) dx = -dx
    const double ax = __builtin_fabs(x);
    const double u = big + ax;
    x = ax - (u - big);
    const int k = (int)(uint32_t)bits(u) * 4;
    const double xx = x * x;
    const double s = x + g_mad<F>(x * xx, g_mad<F>(sn5, xx, sn3), dx);
    const double c = g_mad<F>(x, dx, xx * g_mad<F>(g_mad<F>(cs6, xx, cs4), xx, cs2));
    const double sn = des_sincostab[k], ssn = des_sincostab[k + 1], cs = des_sincostab[k + 2], ccs = des_sincostab[k + 3];
    const double cor = g_mad<F>(s, cs, g_mad<F>(-c, sn, g_mad<F>(s, ccs, ssn)));
    return __builtin_copysign(sn + cor, xold);
}

template <bool F> DES_LIBM_FN double g_do_cos(double x, double dx)
{
    const double big = 52776558133248.0;
    const double sn3 = -0.16666666666666488, sn5 = 0.008333332142857223;
    const double cs2 = 0.5, cs4 = -0.04166666666666644, cs6 = 0.001388888740079376;
    if (x < 0) dx = -dx;
    const double ax = __builtin_fabs(x);
    const double u = big + ax;
    x = (ax - (u - big)) + dx;
    const int k = (int)(uint32_t)bits(u) * 4;
    const double xx = x * x;
    const double s = g_mad<F>(x * xx, g_mad<F>(sn5, xx, sn3), x);
    const double c = xx * g_mad<F>(g_mad<F>(cs6, xx, cs4), xx, cs2);
    const double sn = des_sincostab[k], ssn = des_sincostab[k + 1], cs = des_sincostab[k + 2], ccs = des_sincostab[k + 3];
    const double cor = g_mad<F>(-s, sn, g_mad<F>(-c, cs, g_mad<F>(-s, ssn, ccs)));
    return cs + cor;
}

// TAYLOR_SIN: |x| < 0.126
template <bool F> DES_LIBM_FN double g_taylor_sin(double xx, double x, double dx)
{
    const double s1 = -0.16666666666666666, s2 = 0.008333333333332329, s3 = -0.00019841269834414642,
                 s4 = 2.755729806860771e-06, s5 = -2.5022014848318398e-08;
    const double p = g_mad<F>(g_mad<F>(g_mad<F>(g_mad<F>(s5, xx, s4), xx, s3), xx, s2), xx, s1);
    const double t = g_mad<F>(g_mad<F>(p, x, -(0.5 * dx)), xx, dx);
    return x + t;
}

// do_sin with its small-argument branch (s_sin.c: do_sin / do_sincos)
template <bool F> DES_LIBM_FN double g_sin_of(double a, double da)
{
    if (__builtin_fabs(a) < 0.126) return g_taylor_sin<F>(a * a, a, da);
    return g_do_sin<F>(a, da);
}

// reduce_sincos: x = n pi/2 + (a + da), |x| < 105414350
template <bool F> DES_LIBM_FN int g_reduce_sincos(double x, double *a, double *da)
{
    const double hpinv = 0.6366197723675814, toint = 6755399441055744.0;
    const double mp1 = 1.5707963407039642, mp2 = -1.3909067564377153e-08, pp3 = -4.97899623147991e-17, pp4 = -1.9034889620193266e-25;
    const double t = g_mad<F>(x, hpinv, toint);
    const double xn = t - toint;
    const double y = g_mad<F>(-xn, mp2, g_mad<F>(-xn, mp1, x));
    const int n = (int)(uint32_t)bits(t) & 3;
    const double t2 = g_mad<F>(-xn, pp3, y);                     // y - xn pp3, the product not rounded
    double db = g_mad<F>(-pp3, xn, y - t2);
    const double b = g_mad<F>(-xn, pp4, t2);
    db += g_mad<F>(-xn, pp4, t2 - b);
    *a = b; *da = db;
    return n;
}

// the range the restatement covers (the rest: des_libm.hpp's own routines)
DES_LIBM_FN bool g_sincos_in_range(double x) { return ((bits(x) >> 32) & 0x7fffffff) < 0x419921FB; }

DES_LIBM_FN double g_sin(double x)
{
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffff;
    if (k < 0x3e500000) return x;                            // |x| < 2^-26
    if (k < 0x3feb6000) return g_sin_of<true>(x, 0.0);             // |x| < 0.855469
    if (k < 0x400368fd) {                                    // |x| < 2.426265
        const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
        const double t = hp0 - __builtin_fabs(x);
        return __builtin_copysign(g_do_cos<true>(t, hp1), x);
    }
    double a, da;
    const int n = g_reduce_sincos<true>(x, &a, &da);
    const double r = (n & 1) ? g_do_cos<true>(a, da) : g_sin_of<true>(a, da);
    return (n & 2) ? -r : r;
}

DES_LIBM_FN double g_cos(double x)
{
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffff;
    if (k < 0x3e400000) return 1.0;                          // |x| < 2^-27
    if (k < 0x3feb6000) return g_do_cos<true>(x, 0.0);
    if (k < 0x400368fd) {
        const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
        const double y = hp0 - __builtin_fabs(x);
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        return g_sin_of<true>(a, da);
    }
    double a, da;
    const int n = g_reduce_sincos<true>(x, &a, &da) + 1;
    const double r = (n & 1) ? g_do_cos<true>(a, da) : g_sin_of<true>(a, da);
    return (n & 2) ? -r : r;
}

// ---- s_tan.c (__tan_fma) ---------------------------------------------------------------------------------------
// tan(a + da) from the reduced argument, n odd: -cot.  |a| <= pi/4 (+ a hair).
DES_LIBM_FN double g_tan_reduced(double a, double da, int n)
{
    const double g2 = 0.060799986124038696;
    const double d3 = 0.3333333333333333, d5 = 0.1333333333332673, d7 = 0.05396825411095433, d9 = 0.021869380598167253,
                 d11 = 0.008896546349534924;
    const double e0 = 0.3333333333332254, e1 = 0.13333354920669913;
    double ya, yya, sy;
    if (a < 0.0) { ya = -a; yya = -da; sy = -1.0; } else { ya = a; yya = da; sy = 1.0; }
    if (ya <= g2) {
        // (VII) 1e-7 < |y| <= 0.0608 (2.35 has no separate branch below 1e-7 any more)
        const double a2 = a * a;
        double t2 = fma_(fma_(fma_(fma_(d11, a2, d9), a2, d7), a2, d5), a2, d3);
        t2 = fma_(a * a2, t2, da);
        const double b = a + t2;
        if (!n) return b;                                    // tan
        // -cot: EADD (a, t2, b, db); DIV2 (1, 0, b, db, c, dc); -(c + dc)
        const double db = (__builtin_fabs(a) > __builtin_fabs(t2)) ? ((a - b) + t2) : ((t2 - b) + a);
        const double c = 1.0 / b;
        const double u = c * b;
        const double uu = fma_(c, b, -u);
        const double cc = fma_(-db, c, ((1.0 - u) - uu) + 0.0) / b;
        const double z = c + cc;
        const double zz = (c - z) + cc;
        return -(zz + z);
    }
    // (VIII) 0.0608 < |y| <= 0.787
    const int i = (int)fma_(256.0, ya, -15.5);
    const double z = (ya - des_tan_xfg[4 * i]) + yya;
    const double z2 = z * z;
    const double pz = fma_(z * z2, fma_(z2, e1, e0), z);
    const double fi = des_tan_xfg[4 * i + 1], gi = des_tan_xfg[4 * i + 2];
    if (n) return (gi - (fi + gi) * pz / (pz + fi)) * -sy;   // -cot
    return ((fi + gi) * pz / (gi - pz) + fi) * sy;           // tan
}

DES_LIBM_FN bool g_tan_in_range(double x) { return __builtin_fabs(x) <= 100000000.0; }      // (NaN: false)

DES_LIBM_FN double g_tan(double x)
{
    const double g1 = 1.2589993048095494e-08, g2 = 0.060799986124038696, g3 = 0.7869997024536133, g4 = 25.0;
    const double w = (x < 0.0) ? -x : x;
    if (w <= g1) return x;                                   // (I)
    if (w <= g2) {                                           // (II)
        const double d3 = 0.3333333333333333, d5 = 0.1333333333332673, d7 = 0.05396825411095433, d9 = 0.021869380598167253,
                     d11 = 0.008896546349534924;
        const double x2 = x * x;
        const double t2 = fma_(fma_(fma_(fma_(d11, x2, d9), x2, d7), x2, d5), x2, d3);
        return fma_(x * x2, t2, x);
    }
    if (w <= g3) {                                           // (III)
        const double e0 = 0.3333333333332254, e1 = 0.13333354920669913;
        const int i = (int)fma_(256.0, w, -15.5);
        const double z = w - des_tan_xfg[4 * i];
        const double z2 = z * z;
        const double s = (x < 0.0) ? -1.0 : 1.0;
        const double pz = fma_(z * z2, fma_(z2, e1, e0), z);
        const double fi = des_tan_xfg[4 * i + 1], gi = des_tan_xfg[4 * i + 2];
        return ((fi + gi) * pz / (gi - pz) + fi) * s;
    }
    const double hpinv = 0.6366197723675814, toint = 6755399441055744.0;
    const double mp1 = 1.5707963407039642, mp2 = -1.3909067564377153e-08;
    const double t = fma_(x, hpinv, toint);
    const double xn = t - toint;
    const int n = (int)(uint32_t)bits(t) & 1;
    const double t1 = fma_(-xn, mp2, fma_(-xn, mp1, x));
    if (w <= g4) {                                           // 0.787 < |x| <= 25: pi/2 in three pieces
        const double mp3 = -4.9789962505147994e-17;
        const double a = fma_(-xn, mp3, t1);
        const double da = fma_(-xn, mp3, t1 - a);
        return g_tan_reduced(a, da, n);
    }
    // 25 < |x| <= 1e8: four pieces
    const double pp3 = -4.97899623147991e-17, pp4 = -1.9034889620193266e-25;
    const double tt = fma_(-xn, pp3, t1);
    double da = fma_(-xn, pp3, t1 - tt);
    double a = fma_(-xn, pp4, tt);
    da = da + fma_(-xn, pp4, tt - a);
    const double s1 = a + da;                                // EADD (a, da, t1, t2)
    const double s2 = (__builtin_fabs(a) > __builtin_fabs(da)) ? ((a - s1) + da) : ((da - s1) + a);
    return g_tan_reduced(s1, s2, n);
}

// ---- e_atan2.c (__ieee754_atan2_fma) ---------------------------------------------------------------------------
DES_LIBM_FN bool g_atan2_in_range(double, double) { return true; }       // every argument pair

// atan polynomial of e_atan2.c for u < 1/16: u^2 -> d3 + v (d5 + ... + v d13)
DES_LIBM_FN double g_atan_poly(double v)
{
    const double d3 = -0.3333333333333333, d5 = 0.19999999999998855, d7 = -0.14285714283953163, d9 = 0.11111109821886427,
                 d11 = -0.09090424391727987, d13 = 0.07601836584380736;
    return fma_(fma_(fma_(fma_(fma_(d13, v, d11), v, d9), v, d7), v, d5), v, d3);
}
// cij row of u >= 1/16 and its polynomial c2 + v (c3 + v (c4 + v (c5 + v c6)))
DES_LIBM_FN int g_atan_row(double u) { const double two52 = 4503599627370496.0; return ((int)(fma_(u, 256.0, two52) - two52) - 16) * 7; }
DES_LIBM_FN double g_atan_cpoly(const double *c, double v)
{
    return fma_(fma_(fma_(fma_(c[6], v, c[5]), v, c[4]), v, c[3]), v, c[2]);
}

DES_LIBM_FN double g_atan2(double y, double x)
{
    const double hpi = 1.5707963267948966, hpi1 = 6.123233995736766e-17, opi = 3.141592653589793, opi1 = 1.2246467991473532e-16;
    const double inv16 = 0.0625, twom500 = 3.054936363499605e-151, two500 = 3.273390607896142e+150;
    const uint64_t bx = bits(x), by = bits(y);
    const bool xneg = (bx >> 63) != 0, yneg = (by >> 63) != 0;
    if (x != x || y != y) return x + y;                                                // NaN
    const bool xinf = (bx << 1) == 0xffe0000000000000ULL, yinf = (by << 1) == 0xffe0000000000000ULL;
    if ((by << 1) == 0) return yneg ? (xneg ? -opi : -0.0) : (xneg ? opi : 0.0);       // y = +-0
    if (x == 0.0) return yneg ? -hpi : hpi;                                           // x = +-0
    if (xinf) {                                                                       // x = +-inf
        const double qpi = 0.7853981633974483, tqpi = 2.356194490192345;
        if (yinf) return xneg ? (yneg ? -tqpi : tqpi) : (yneg ? -qpi : qpi);
        return xneg ? (yneg ? -opi : opi) : (yneg ? -0.0 : 0.0);
    }
    if (yinf) return yneg ? -hpi : hpi;                                               // y = +-inf
    double ax = xneg ? -x : x, ay = yneg ? -y : y;
    const int de = (int)((uint32_t)(by >> 32) & 0x7ff00000) - (int)((uint32_t)(bx >> 32) & 0x7ff00000);
    if (de >= 59768832) return (y > 0) ? hpi : -hpi;                                  // |y / x| > 2^57
    if (de <= -59768832) {                                                           // |y / x| < 2^-57
        if (x > 0) return __builtin_copysign(ay / ax, y);
        return (y > 0) ? opi : -opi;
    }
    if (ax < twom500 || ay < twom500) { ax *= two500; ay *= two500; }
    if (ax > two500 || ay > two500) { ax *= twom500; ay *= twom500; }
    double u, du, z;
    if (ay < ax) { u = ay / ax; const double v = ax * u; const double vv = fma_(ax, u, -v); du = ((ay - v) - vv) / ax; }
    else         { u = ax / ay; const double v = ay * u; const double vv = fma_(ay, u, -v); du = ((ax - v) - vv) / ay; }
    if (x > 0) {
        if (ay < ax) {                                       // (i) atan(ay / ax)
            if (u < inv16) {
                const double v = u * u;
                z = u + fma_(u * v, g_atan_poly(v), du);
            } else {
                const double *c = des_atan_cij + g_atan_row(u);
                const double t3 = u - c[0];
                const double v = du + t3;                    // EADD (t3, du, v, dv)
                const double dv = (__builtin_fabs(t3) > __builtin_fabs(du)) ? ((t3 - v) + du) : ((du - v) + t3);
                const double t2 = c[2];
                const double p3 = fma_(fma_(fma_(c[6], v, c[5]), v, c[4]), v, c[3]);
                z = fma_(v, t2, fma_(dv, t2, (v * v) * p3)) + c[1];
            }
        } else {                                             // (ii) pi/2 - atan(ax / ay)
            if (u < inv16) {
                const double v = u * u;
                const double zz = (u * v) * g_atan_poly(v);
                const double t2 = hpi - u;                   // ESUB (hpi, u, t2, cor)
                const double cor = (hpi > __builtin_fabs(u)) ? ((hpi - t2) - u) : (hpi - (u + t2));
                z = ((((cor + hpi1) - du) - zz)) + t2;
            } else {
                const double *c = des_atan_cij + g_atan_row(u);
                const double v = (u - c[0]) + du;
                z = (hpi - c[1]) + fma_(-g_atan_cpoly(c, v), v, hpi1);
            }
        }
    } else if (ay > ax) {                                    // (iii) x < 0: pi/2 + atan(ax / ay)
        if (u < inv16) {
            const double v = u * u;
            const double zz = (v * u) * g_atan_poly(v);
            const double t2 = u + hpi;                       // EADD (hpi, u, t2, cor)
            const double cor = (hpi > __builtin_fabs(u)) ? ((hpi - t2) + u) : ((u - t2) + hpi);
            z = (((cor + hpi1) + du) + zz) + t2;
        } else {
            const double *c = des_atan_cij + g_atan_row(u);
            const double v = (u - c[0]) + du;
            z = (hpi + c[1]) + fma_(g_atan_cpoly(c, v), v, hpi1);
        }
    } else {                                                 // (iv) x < 0: pi - atan(ay / ax)
        if (u < inv16) {
            const double v = u * u;
            const double zz = (v * u) * g_atan_poly(v);
            const double t2 = opi - u;                       // ESUB (opi, u, t2, cor)
            const double cor = (opi > __builtin_fabs(u)) ? ((opi - t2) - u) : (opi - (u + t2));
            z = (((cor + opi1) - du) - zz) + t2;
        } else {
            const double *c = des_atan_cij + g_atan_row(u);
            const double v = (u - c[0]) + du;
            z = (opi - c[1]) + fma_(-g_atan_cpoly(c, v), v, opi1);
        }
    }
    return __builtin_copysign(z, y);
}

// s_sincos.c (no FMA variant): what a compiler turns sin(x), cos(x) of one argument into -- gcc does at -O1 and up, so
// the Kopp solver's cos(phi), sin(phi) (3x3-C/dsyevc3.c:66-67) reach the C library as ONE sincos call.  Not the same
// bits as sin() and cos(): no contraction, and in [0.855469, 2.426265) the sine comes from do_cos(a, da) of the
// two-term difference instead of do_cos(hp0 - |x|, hp1).
DES_LIBM_FN void g_sincos(double x, double *sinx, double *cosx)
{
    const uint32_t k = (uint32_t)(bits(x) >> 32) & 0x7fffffff;
    if (k < 0x400368fd) {
        if (k < 0x3e400000) { *sinx = x; *cosx = 1.0; return; }
        if (k < 0x3feb6000) { *sinx = g_sin_of<false>(x, 0.0); *cosx = g_do_cos<false>(x, 0.0); return; }
        const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
        const double y = hp0 - __builtin_fabs(x);
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        *sinx = __builtin_copysign(g_do_cos<false>(a, da), x);
        *cosx = g_sin_of<false>(a, da);
        return;
    }
    double a, da;
    const int n = g_reduce_sincos<false>(x, &a, &da);
    const double r0 = (n & 1) ? g_do_cos<false>(a, da) : g_sin_of<false>(a, da);
    *sinx = (n & 2) ? -r0 : r0;
    const int m = n + 1;
    const double r1 = (m & 1) ? g_do_cos<false>(a, da) : g_sin_of<false>(a, da);
    *cosx = (m & 2) ? -r1 : r1;
}

}  // namespace deslibm
