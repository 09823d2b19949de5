// Portable libm for the stress update: pow, exp, sin, cos, tan, atan2 written in IEEE-754
// double operations only (+, -, *, /, fma, integer bit moves, table look-ups), so that the
// SAME source gives the SAME bits on gfx950 and on a CPU (both sides are compiled with
// -ffp-contract=off; every fma below is explicit).
//
// Why it exists: the reference calls std::pow / std::exp / std::sin / std::tan (rheology.cxx:
// 260-300 creep viscosity, 330-420 plastic_props) and the Kopp solver calls atan2 / cos / sin
// (3x3-C/dsyevc3.c:60-70); glibc and ROCm's ocml round those differently in the last 1-2 ulp,
// which is the ONLY source of device-vs-CPU differences on the path.  The
// engine uses these functions (its default; DES_LIBM=ocml selects ROCm's instead), the CPU
// checker can be switched to the same ones, and the two then agree to the bit for every rheology.
//
// pow and exp go one step further: they return the bits of the C library the CPU reference runs
// on (glibc 2.35, x86-64 with FMA), see "exp / pow" below -- they are the only libm calls that
// reach the state of a model whose elements do not yield (the creep law), so on such a model the
// device in portable mode equals the CPU run with the *C library's* libm, bit for bit.
//
// Accuracy (tools/libm_accuracy.cpp against long double / tests/test_libm.py against mpmath):
// pow, exp <= 0.52 ulp (glibc's); sin, cos <= 0.8 ulp for |x| <= 1e5; tan = sin/cos <= 2 ulp;
// atan2 <= 1.5 ulp.  sin/cos use a three-term Cody-Waite reduction that is exact for
// |x| < 2^20*pi/2 and lose accuracy (never determinism) beyond.
//
// Third-party notice: exp / pow below restate sysdeps/ieee754/dbl-64/e_exp.c and e_pow.c of the GNU C Library 2.35
// (from Arm's Optimized Routines; Copyright (C) 2018-2022 Free Software Foundation, Inc., LGPL-2.1-or-later; upstream
// Arm Optimized Routines: MIT OR Apache-2.0 WITH LLVM-exception) and regenerate their tables from the recipe those
// sources document; that part of this file is distributed under LGPL-2.1-or-later, see THIRD_PARTY_NOTICES.md.
//
// Tables: tools/gen_libm_tables.py (the pow / exp tables are glibc's, recomputed from the recipe
// its sources document; the sin / cos / atan coefficients are our own fits).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DES_LIBM_FN static __device__ __forceinline__
#define DES_LIBM_TAB static __device__ const
#else
#define DES_LIBM_FN static inline
#define DES_LIBM_TAB static const
#endif

#include "des_libm_tables.hpp"

namespace deslibm {

// Table access.  A device kernel may stage the five 128-entry tables in LDS first
// (lds_stage*(), then every call of that wavefront reads its LDS copy instead of global memory:
// E2 93 -> 85 us at 1.1M tets); kernels that do not stage them, and the CPU build, read the arrays.
#if defined(__HIPCC__) && defined(DES_LIBM_LDS_TABLES)
// One private copy per wavefront (DES_LIBM_LDS_WAVES of them per workgroup): a wave runs in
// lockstep, so its copy needs no workgroup barrier -- a barrier would line up the load / compute
// / store phases of all waves of the workgroup, which costs more than the staging itself.
#ifndef DES_LIBM_LDS_WAVES
#define DES_LIBM_LDS_WAVES 4
#endif
static __shared__ double lds_tab[DES_LIBM_LDS_WAVES][5 * 128];
// (workgroups of more than DES_LIBM_LDS_WAVES wavefronts share the copies: every wavefront writes the whole of its copy itself
//  before it reads -- the same values whoever writes them)
DES_LIBM_FN double *lds_mine() { return lds_tab[(threadIdx.x >> 6) % DES_LIBM_LDS_WAVES]; }
// lds_stage_begin() copies (every lane of the wave calls it, at the top of the kernel);
// lds_stage_end() orders the copy before the first deslibm:: call of the wave.
DES_LIBM_FN void lds_stage_begin()
{
    double *t = lds_mine();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = lane + 64 * u;
        t[j] = des_log_invc[j];
        t[128 + j] = des_log_chi[j];
        t[256 + j] = des_log_clo[j];
        t[384 + j] = des_exp_hi[j];
        t[512 + j] = des_exp_tail[j];
    }
}
DES_LIBM_FN void lds_stage_end() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
DES_LIBM_FN void lds_stage() { lds_stage_begin(); lds_stage_end(); }
#define DES_T_INVC(i) lds_mine()[i]
#define DES_T_CHI(i)  lds_mine()[128 + (i)]
#define DES_T_CLO(i)  lds_mine()[256 + (i)]
#define DES_T_EHI(i)  lds_mine()[384 + (i)]
#define DES_T_ETL(i)  lds_mine()[512 + (i)]
#else
#define DES_T_INVC(i) des_log_invc[i]
#define DES_T_CHI(i)  des_log_chi[i]
#define DES_T_CLO(i)  des_log_clo[i]
#define DES_T_EHI(i)  des_exp_hi[i]
#define DES_T_ETL(i)  des_exp_tail[i]
#endif

DES_LIBM_FN uint64_t bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
DES_LIBM_FN double   dbl(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
DES_LIBM_FN double   fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// s = fl(a + b), *e = (a + b) - s exactly (Knuth; no ordering assumption)
DES_LIBM_FN double two_sum(double a, double b, double *e)
{
    const double s = a + b;
    const double bb = s - a;
    *e = (a - (s - bb)) + (b - bb);
    return s;
}

// round-to-nearest integer of |t| < 2^51 by the shift trick; *ki = bits of the shifted value
// (its low 52 bits hold 2^51 + n)
DES_LIBM_FN double round_shift(double t, uint64_t *ki)
{
    const double shift = 6755399441055744.0;              // 1.5 * 2^52
    double kd = t + shift;
    *ki = bits(kd);
    return kd - shift;
}

// ---- exp / pow: the C library's bits --------------------------------------------------------
// These two are NOT designs of our own: they restate, operation by operation, the exp and pow of
// glibc 2.35 (sysdeps/ieee754/dbl-64/e_exp.c, e_pow.c -- Szabolcs Nagy's ARM optimized routines)
// in the form the C library executes on an x86-64 host with FMA (__ieee754_exp_fma /
// __ieee754_pow_fma, selected by sysdeps/x86_64/fpu/multiarch/ifunc-fma4.h whenever the CPU has
// FMA + AVX2; built -mfma -mavx2, so gcc contracts a*b + c where the source allows it -- the
// contraction pattern below was read off the shipped object code, every fma_ is one vfmadd).
// That libm is the one the CPU reference / oracle calls for the creep law (matprops.cxx:333-377),
// the only libm use that reaches the state of a model that does not yield -- so with
// this libm (the engine's default) the device reproduces the CPU run to the bit, not merely to 1 ulp per call.
// tests/test_libm.py compares both functions with the host's own on 2e7 arguments (zero
// mismatches); a host without FMA runs glibc's non-fma variant, which differs in the last bit of
// a few calls per thousand (the test says so and is skipped there).
// errno and the floating-point exception flags are not reproduced.
DES_LIBM_FN uint32_t top12(double x) { return (uint32_t)(bits(x) >> 52); }

// e_exp.c / e_pow.c specialcase(): the scale 2^(k/N) alone would over- or underflow
DES_LIBM_FN double exp_special(double tmp, uint64_t sbits, uint64_t ki)
{
    if ((ki & 0x80000000ULL) == 0) {
        sbits -= 1009ULL << 52;                             // k > 0: the exponent of scale may have overflowed by <= 460
        const double scale = dbl(sbits);
        return dbl(0x7f00000000000000ULL) * fma_(scale, tmp, scale);       // 0x1p1009
    }
    sbits += 1022ULL << 52;                                 // k < 0: care in the subnormal range
    const double scale = dbl(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (__builtin_fabs(y) < 1.0) {
        // round y to the right precision before scaling it into the subnormal range
        const double one = (y < 0.0) ? -1.0 : 1.0;
        double lo = (scale - y) + st;
        const double hi = one + y;
        lo = ((one - hi) + y) + lo;
        y = (hi + lo) - one;
        if (y == 0.0) y = dbl(sbits & 0x8000000000000000ULL);
    }
    return dbl(0x0010000000000000ULL) * y;                               // 0x1p-1022
}

// exp(x + xtail) * (-1)^(sign_bias != 0); e_pow.c exp_inline(), and e_exp.c __exp with POW = 0
#define DES_SIGN_BIAS (0x800u << 7)
template <int POW>
DES_LIBM_FN double exp_inline(double x, double xtail, uint32_t sign_bias)
{
    uint32_t abstop = top12(x) & 0x7ff;
    if (__builtin_expect(abstop - 0x3c9u >= 0x3fu, 0)) {      // |x| < 2^-54 or >= 512
        if (abstop - 0x3c9u >= 0x80000000u) {
            const double one = 1.0 + x;                     // tiny: avoid spurious underflow
            return sign_bias ? -one : one;
        }
        if (abstop >= 0x409u) {                              // |x| >= 1024
            if (!POW) {
                if (bits(x) == 0xfff0000000000000ULL) return 0.0;
                if (abstop >= 0x7ffu) return 1.0 + x;
            }
            const double r = (bits(x) >> 63) ? 0.0 : __builtin_inf();         // __math_uflow / __math_oflow
            return sign_bias ? -r : r;
        }
        abstop = 0;                                          // large x is special-cased below
    }
    // x = ln2/N k + r, exp(x) = 2^(k/N) exp(r)
    double kd = fma_(des_exp_invln2N, x, 6755399441055744.0);     // z + Shift, Shift = 0x1.8p52
    const uint64_t ki = bits(kd);
    kd -= 6755399441055744.0;
    double r = fma_(kd, des_exp_negln2loN, fma_(kd, des_exp_negln2hiN, x));
    if (POW) r += xtail;
    const int j = (int)(ki & 127);
    const uint64_t top = (ki + sign_bias) << 45;
    const double tail = DES_T_ETL(j);
    const uint64_t sbits = bits(DES_T_EHI(j)) - ((uint64_t)j << 45) + top;   // T[idx + 1] + top
    const double r2 = r * r;
    const double p23 = fma_(des_exp_C[1], r, des_exp_C[0]);
    const double p45 = fma_(r, des_exp_C[3], des_exp_C[2]);
    const double tmp = fma_(r2 * r2, p45, fma_(p23, r2, tail + r));
    if (__builtin_expect(abstop == 0, 0)) return exp_special(tmp, sbits, ki);
    const double scale = dbl(sbits);
    return fma_(scale, tmp, scale);
}

DES_LIBM_FN double exp(double x) { return exp_inline<0>(x, 0.0, 0); }

// e_pow.c log_inline(): log(x) = k ln2 + log(c) + log1p(z/c - 1) as y + *tail, ~2^-68 relative
DES_LIBM_FN double log_inline(uint64_t ix, double *tail)
{
    const uint64_t OFF = 0x3fe6955500000000ULL;
    const uint64_t tmp = ix - OFF;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const double z = dbl(ix - (tmp & (0xfffULL << 52)));
    const double kd = (double)k;
    const double r = fma_(z, DES_T_INVC(i), -1.0);           // exact: invc has 8 bits
    const double t1 = fma_(kd, des_ln2hi, DES_T_CHI(i));     // exact by construction of the table
    const double t2 = t1 + r;
    const double lo1 = fma_(kd, des_ln2lo, DES_T_CLO(i));
    const double lo2 = (t1 - t2) + r;
    const double ar = des_log_A[0] * r;                      // A[0] = -0.5
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma_(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double q12 = fma_(r, des_log_A[2], des_log_A[1]);
    const double q34 = fma_(r, des_log_A[4], des_log_A[3]);
    const double q56 = fma_(r, des_log_A[6], des_log_A[5]);
    const double q = fma_(ar2, fma_(q56, ar2, q34), q12);
    const double lo = fma_(ar3, q, ((lo1 + lo2) + lo3) + lo4);
    const double y = hi + lo;
    *tail = (hi - y) + lo;
    return y;
}

// 0: y is not an integer, 1: odd, 2: even (e_pow.c checkint)
DES_LIBM_FN int pow_checkint(uint64_t iy)
{
    const int e = (int)((iy >> 52) & 0x7ff);
    if (e < 0x3ff) return 0;
    if (e > 0x3ff + 52) return 2;
    if (iy & ((1ULL << (0x3ff + 52 - e)) - 1)) return 0;
    if (iy & (1ULL << (0x3ff + 52 - e))) return 1;
    return 2;
}

DES_LIBM_FN bool pow_zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ULL - 1; }

DES_LIBM_FN double pow(double x, double y)
{
    uint32_t sign_bias = 0;
    uint64_t ix = bits(x);
    const uint64_t iy = bits(y);
    uint32_t topx = top12(x);
    const uint32_t topy = top12(y);
    if (__builtin_expect(topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu, 0)) {
        // x < 0x1p-1022 or inf or nan, or |y| < 0x1p-65 or |y| >= 0x1p63 or nan
        const uint64_t one = 0x3ff0000000000000ULL, inf = 0x7ff0000000000000ULL;
        if (pow_zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0;                     // (signalling NaNs are not told apart)
            if (ix == one) return 1.0;
            if (2 * ix > 2 * inf || 2 * iy > 2 * inf) return x + y;
            if (2 * ix == 2 * one) return 1.0;
            if ((2 * ix < 2 * one) == !(iy >> 63)) return 0.0;      // |x| < 1 && y == inf, or |x| > 1 && y == -inf
            return y * y;
        }
        if (pow_zeroinfnan(ix)) {
            double x2 = x * x;
            if ((ix >> 63) && pow_checkint(iy) == 1) x2 = -x2;
            return (iy >> 63) ? 1 / x2 : x2;
        }
        if (ix >> 63) {                                      // finite x < 0
            const int yint = pow_checkint(iy);
            if (yint == 0) return dbl(0x7ff8000000000000ULL) ;       // __math_invalid
            if (yint == 1) sign_bias = DES_SIGN_BIAS;
            ix &= 0x7fffffffffffffffULL;
            topx &= 0x7ff;
        }
        if ((topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
            if (ix == one) return 1.0;
            if ((topy & 0x7ff) < 0x3beu) return ix > one ? 1.0 + y : 1.0 - y;        // |y| < 2^-65
            return ((ix > one) == (topy < 0x800u)) ? __builtin_inf() : 0.0;     // __math_oflow / __math_uflow
        }
        if (topx == 0) {                                     // subnormal x: normalise
            ix = bits(x * 4503599627370496.0);                 // 0x1p52
            ix &= 0x7fffffffffffffffULL;
            ix -= 52ULL << 52;
        }
    }
    double lo;
    const double hi = log_inline(ix, &lo);
    const double ehi = y * hi;
    const double elo = fma_(y, lo, fma_(y, hi, -ehi));
    return exp_inline<1>(ehi, elo, sign_bias);
}

// ---- sin / cos ------------------------------------------------------------------------
// x = n pi/2 + (r + rt), |r| <= pi/4 (+ a hair); returns n mod 4.
DES_LIBM_FN int rem_pio2(double x, double *r, double *rt)
{
    // no shortcut for |x| <= pi/4: fn = 0 there and every step below returns x unchanged
    uint64_t ki;
    const double fn = round_shift(x * des_invpio2, &ki);
    // three 33-bit pieces of pi/2: fn*piece exact for |fn| < 2^20
    const double a = fma_(fn, -des_pio2[0], x);                 // exact (Sterbenz)
    double e1, e2, e3;
    const double b = two_sum(a, -(fn * des_pio2[1]), &e1);
    const double c = two_sum(b, -(fn * des_pio2[2]), &e2);
    const double e = (e1 + e2) - fn * des_pio2[3];
    *r = two_sum(c, e, &e3);
    *rt = e3;
    return (int)(ki & 3);
}

DES_LIBM_FN double sin_kernel(double x, double t)
{
    const double z = x * x;
    const double p = des_sin_S[1] + z * (des_sin_S[2] + z * (des_sin_S[3] + z * (des_sin_S[4] + z * des_sin_S[5])));
    const double v = z * x;
    // x + t + x^3 (S0 + z p), with the tail's first-order effect t (1 - z/2)
    return x + (v * des_sin_S[0] + (t - z * (0.5 * t - v * p)));
}

DES_LIBM_FN double cos_kernel(double x, double t)
{
    const double z = x * x;
    const double p = z * (des_cos_C[0] + z * (des_cos_C[1] + z * (des_cos_C[2] + z * (des_cos_C[3] + z * (des_cos_C[4] + z * des_cos_C[5])))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * p - x * t));
}

DES_LIBM_FN void sincos_own(double x, double *s, double *c)
{
    double r, t;                                     // inf / NaN turn into NaN on their own (inf - inf)
    const int n = rem_pio2(x, &r, &t);
    const double sk = sin_kernel(r, t), ck = cos_kernel(r, t);
    const double sv = (n & 1) ? ck : sk, cv = (n & 1) ? sk : ck;
    *s = (n & 2) ? -sv : sv;
    *c = ((n + 1) & 2) ? -cv : cv;
}

DES_LIBM_FN double sin_own(double x) { double s, c; sincos_own(x, &s, &c); return s; }
DES_LIBM_FN double cos_own(double x) { double s, c; sincos_own(x, &s, &c); return c; }
DES_LIBM_FN double tan_own(double x) { double s, c; sincos_own(x, &s, &c); return s / c; }

// ---- atan2 ----------------------------------------------------------------------------
DES_LIBM_FN double atan_pos(double x)              // x >= 0 (inf allowed)
{
    // reduction to |t| <= 7/16 around 0, 1/2, 1, 3/2, inf: t = num/den picked by selects, one division
    const bool b0 = x < 0.4375, b1 = x < 0.6875, b2 = x < 1.1875, b3 = x < 2.4375;
    const double num = b0 ? x : b1 ? 2.0 * x - 1.0 : b2 ? x - 1.0 : b3 ? x - 1.5 : -1.0;
    const double den = b0 ? 1.0 : b1 ? 2.0 + x : b2 ? x + 1.0 : b3 ? 1.0 + 1.5 * x : x;
    const double hi = b0 ? 0.0 : b1 ? des_atan_hi[0] : b2 ? des_atan_hi[1] : b3 ? des_atan_hi[2] : des_atan_hi[3];
    const double lo = b0 ? 0.0 : b1 ? des_atan_lo[0] : b2 ? des_atan_lo[1] : b3 ? des_atan_lo[2] : des_atan_lo[3];
    const double t = num / den;
    const double z = t * t;
    const double w = z * z;
    const double pe = des_atan_T[0] + w * (des_atan_T[2] + w * (des_atan_T[4] + w * (des_atan_T[6] + w * (des_atan_T[8] + w * des_atan_T[10]))));
    const double po = des_atan_T[1] + w * (des_atan_T[3] + w * (des_atan_T[5] + w * (des_atan_T[7] + w * des_atan_T[9])));
    const double q = z * (pe + z * po);            // atan(t)/t - 1
    return hi + ((t * q + lo) + t);
}

DES_LIBM_FN double atan2_own(double y, double x)
{
    const bool sy = (bits(y) >> 63) != 0, sx = (bits(x) >> 63) != 0;
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    const double inf = __builtin_inf();
    // y/0 = inf -> pi/2, 0/x = 0 and y/inf = 0 -> 0 or pi, NaN stays NaN: the main line covers them
    const double a = atan_pos(ay / ax);
    double r = sx ? des_pi_hi - (a - des_pi_lo) : a;
    r = (ay == 0.0 && x == x) ? (sx ? des_pi_hi : 0.0) : r;                                    // includes 0/0
    r = (ax == inf && ay == inf) ? (sx ? 3.0 * des_atan_hi[1] : des_atan_hi[1]) : r;
    return sy ? -r : r;
}

}  // namespace deslibm

// sin / cos / tan / atan2 with the C library's bits (glibc 2.35), where its own range reduction applies
#include "des_libm_trig.hpp"

namespace deslibm {
DES_LIBM_FN double sin(double x) { return g_sincos_in_range(x) ? g_sin(x) : sin_own(x); }
DES_LIBM_FN double cos(double x) { return g_sincos_in_range(x) ? g_cos(x) : cos_own(x); }
DES_LIBM_FN void sincos(double x, double *s, double *c) { if (g_sincos_in_range(x)) g_sincos(x, s, c); else sincos_own(x, s, c); }
DES_LIBM_FN double tan(double x) { return g_tan_in_range(x) ? g_tan(x) : tan_own(x); }
DES_LIBM_FN double atan2(double y, double x) { return g_atan2_in_range(y, x) ? g_atan2(y, x) : atan2_own(y, x); }
}  // namespace deslibm
