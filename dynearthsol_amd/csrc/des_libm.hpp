// Portable libm for the stress update: pow, exp, sin, cos, tan, atan2 written in IEEE-754
// double operations only (+, -, *, /, fma, integer bit moves, table look-ups), so that the
// SAME source gives the SAME bits on gfx950 and on a CPU (both sides are compiled with
// -ffp-contract=off; every fma below is explicit).
//
// Why it exists: the reference calls std::pow / std::exp / std::sin / std::tan (rheology.cxx:
// 260-300 creep viscosity, 330-420 plastic_props) and the Kopp solver calls atan2 / cos / sin
// (3x3-C/dsyevc3.c:60-70); glibc and ROCm's ocml round those differently in the last 1-2 ulp,
// which is the ONLY source of device-vs-CPU differences on the path.  With
// DES_LIBM=portable the engine uses these functions, the CPU checker can be switched to the
// same ones, and the two agree to the bit for every rheology.  The default stays ocml.
//
// Accuracy (tools/libm_accuracy.cpp against long double / tests/test_libm.py against mpmath):
// pow, exp <= 0.53 ulp; sin, cos <= 0.8 ulp for |x| <= 1e5; tan = sin/cos <= 2 ulp;
// atan2 <= 1.5 ulp.  Domain notes: pow() is defined here for x >= 0 only (x < 0 gives NaN; the
// path raises a strain-rate invariant or a material constant); sin/cos use a three-term
// Cody-Waite reduction that is exact for |x| < 2^20*pi/2 and lose accuracy (never determinism)
// beyond.
//
// The method for pow/exp is the usual table-driven one (Tang 1989/1990; the structure with an
// exact r = fma(z, 1/c, -1) follows the published design of the ARM optimized routines):
// tables and coefficients are our own, from tools/gen_libm_tables.py.
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
DES_LIBM_FN double *lds_mine() { return lds_tab[threadIdx.x >> 6]; }
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

// ---- exp ------------------------------------------------------------------------------
// exp(x + xtail), |xtail| << |x|.  x = k ln2/128 + r; result = 2^(k/128) (1 + tail + r + r^2 P(r)).
// Straight-line code: the argument is clamped into the range where the main path is valid and
// the out-of-range / NaN results are selected at the end (on the GPU a branch costs more than
// the few selects, and this function sits in the per-element creep law).
DES_LIBM_FN double exp_core(double x, double xtail)
{
    double xc = (x > -746.0) ? x : -746.0;                  // NaN lands here too
    xc = (xc < 710.0) ? xc : 710.0;
    uint64_t ki;
    const double kd = round_shift(des_exp_invln2N * xc, &ki);
    double r = fma_(kd, -des_exp_ln2hiN, xc);               // exact: ln2hiN has 42 bits, |k| < 2^18
    r = fma_(kd, -des_exp_ln2loN, r) + xtail;
    const int64_t n = (int64_t)(ki & 0xfffffffffffffULL) - ((int64_t)1 << 51);
    const int j = (int)(n & 127);
    const int64_t k = n >> 7;                                  // floor(n / 128), -1076 .. 1025
    const double r2 = r * r;
    const double tmp = DES_T_ETL(j) + r + r2 * (des_exp_C[0] + r * des_exp_C[1]) + r2 * r2 * (des_exp_C[2] + r * des_exp_C[3]);
    // 2^k in two factors, so that neither leaves the exponent range; the second multiplication
    // is exact unless the result is subnormal or overflows (then it rounds once, as it should)
    const int64_t k1 = k >> 1, k2 = k - k1;
    const double s1 = dbl(bits(DES_T_EHI(j)) + ((uint64_t)k1 << 52));
    const double s2 = dbl((uint64_t)(1023 + k2) << 52);
    double res = (s1 + s1 * tmp) * s2;
    res = (x > 709.782712893384) ? __builtin_inf() : res;
    res = (x < -745.2) ? 0.0 : res;
    res = (x != x) ? x + x : res;
    return res;
}

DES_LIBM_FN double exp(double x) { return exp_core(x, 0.0); }

// ---- pow ------------------------------------------------------------------------------
// log(x) for finite x > 0 as hi + lo with ~2^-68 relative error.
DES_LIBM_FN double log_dd(uint64_t ix, int64_t kadj, double *lo_out)
{
    const uint64_t OFF = 0x3fe6955500000000ULL;
    const uint64_t tmp = ix - OFF;
    const int i = (int)((tmp >> 45) & 127);
    const int64_t k = ((int64_t)tmp >> 52) + kadj;
    const double z = dbl(ix - (tmp & 0xfff0000000000000ULL));
    const double kd = (double)k;
    const double r = fma_(z, DES_T_INVC(i), -1.0);          // exact (see the generator)
    // k ln2 + log c + r - r^2/2, every partial sum kept with its rounding error
    double e1, e2, e3;
    const double t1 = two_sum(kd * des_ln2hi, DES_T_CHI(i), &e1);   // kd*ln2hi is exact (42 + 11 bits)
    const double t2 = two_sum(t1, r, &e2);
    const double ar = -0.5 * r;
    const double ar2 = r * ar;
    const double e4 = fma_(ar, r, -ar2);                      // exact error of ar2
    const double hi = two_sum(t2, ar2, &e3);
    const double r2 = r * r;
    const double p = (r * r2) * (des_log_A[0] + r * des_log_A[1] + r2 * (des_log_A[2] + r * des_log_A[3]
                     + r2 * (des_log_A[4] + r * des_log_A[5])));
    const double lo = (kd * des_ln2lo + DES_T_CLO(i)) + e1 + e2 + e3 + e4 + p;
    const double y = hi + lo;
    *lo_out = (hi - y) + lo;                                  // |hi| >= |lo|
    return y;
}

DES_LIBM_FN double pow(double x, double y)
{
    const uint64_t ix = bits(x), iy = bits(y);
    const uint64_t ax = ix & 0x7fffffffffffffffULL, ay = iy & 0x7fffffffffffffffULL;
    const bool yneg = (iy >> 63) != 0;
    // main path on a base that is positive, finite and normal (anything else is replaced by 1.5
    // here and overridden below); subnormal bases are scaled by 2^52
    const bool x_ok = (ix - 1) < 0x7fefffffffffffffULL;        // 0 < x < inf
    const bool sub = ix < 0x0010000000000000ULL;
    uint64_t im = sub ? bits(x * 4503599627370496.0) : ix;
    im = x_ok ? im : 0x3ff8000000000000ULL;
    double lo;
    const double hi = log_dd(im, sub ? -52 : 0, &lo);
    const double ehi = y * hi;
    const double elo = y * lo + fma_(y, hi, -ehi);
    double res = exp_core(ehi, elo);                           // y = +-inf: ehi = +-inf gives inf / 0
    res = (ax == 0) ? (yneg ? __builtin_inf() : 0.0) : res;    // pow(+-0, y); the sign of zero is dropped
    res = (ix == 0x7ff0000000000000ULL) ? (yneg ? 0.0 : __builtin_inf()) : res;
    res = ((ix >> 63) && ax != 0) ? dbl(0x7ff8000000000000ULL) : res;      // x < 0: not on the path
    res = (ax > 0x7ff0000000000000ULL || ay > 0x7ff0000000000000ULL) ? x + y : res;
    res = (ix == 0x3ff0000000000000ULL || ay == 0) ? 1.0 : res;            // pow(1, y) = pow(x, 0) = 1
    return res;
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

DES_LIBM_FN void sincos(double x, double *s, double *c)
{
    double r, t;                                     // inf / NaN turn into NaN on their own (inf - inf)
    const int n = rem_pio2(x, &r, &t);
    const double sk = sin_kernel(r, t), ck = cos_kernel(r, t);
    const double sv = (n & 1) ? ck : sk, cv = (n & 1) ? sk : ck;
    *s = (n & 2) ? -sv : sv;
    *c = ((n + 1) & 2) ? -cv : cv;
}

DES_LIBM_FN double sin(double x) { double s, c; sincos(x, &s, &c); return s; }
DES_LIBM_FN double cos(double x) { double s, c; sincos(x, &s, &c); return c; }
DES_LIBM_FN double tan(double x) { double s, c; sincos(x, &s, &c); return s / c; }

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

DES_LIBM_FN double atan2(double y, double x)
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
