// des_kernels.hpp -- per-element / per-node device functions of the explicit time step,
// written for gfx950 (wave64).  Every function restates one piece of the reference's hot
// path (file:line cited) with the reference's operation order, so that with
// -ffp-contract=off the libm-free parts agree with the CPU build to the bit.
#ifndef DES_KERNELS_HPP
#define DES_KERNELS_HPP

#include <hip/hip_runtime.h>
#include <float.h>

#include "des_params.h"
#include "des_libm.hpp"
#if defined(DES_LIBM_LDS_TABLES)
static_assert(DES_LIBM_LDS_WAVES * 64 == 256, "one LDS table copy per wavefront of a DES_BLOCK workgroup");
#endif

#define DES_BLOCK 256

namespace desk {

struct alignas(32) d4 { double x, y, z, w; };

__device__ __forceinline__ d4 ld4(const d4 *p) { return *p; }

// Spatially contiguous chunks of the mesh stay on one XCD: hardware deals consecutive
// workgroups round-robin over the 8 XCDs (blocks b and b+8 share an L2), so logical block
// L = (b % 8) * ceil(nb / 8) + b / 8 keeps each XCD sweeping its own 1/8 of the renumbered
// mesh and the nodal gathers of neighbouring blocks hit the same 4 MiB L2.  Speed only:
// every logical block is executed exactly once whatever the placement.
__device__ __forceinline__ int logical_block(int nblocks)
{
    const int per = (nblocks + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}

// fields.cxx:11-38 -- shape-function gradients of a linear tet from its 4 corner points
__device__ __forceinline__ void shape_fn(const d4 c[4], double vol, double sx[4], double sy[4], double sz[4])
{
    double iv = 1.0 / (6.0 * vol);

    double x01 = c[0].x - c[1].x, x02 = c[0].x - c[2].x, x03 = c[0].x - c[3].x;
    double x12 = c[1].x - c[2].x, x13 = c[1].x - c[3].x, x23 = c[2].x - c[3].x;
    double y01 = c[0].y - c[1].y, y02 = c[0].y - c[2].y, y03 = c[0].y - c[3].y;
    double y12 = c[1].y - c[2].y, y13 = c[1].y - c[3].y, y23 = c[2].y - c[3].y;
    double z01 = c[0].z - c[1].z, z02 = c[0].z - c[2].z, z03 = c[0].z - c[3].z;
    double z12 = c[1].z - c[2].z, z13 = c[1].z - c[3].z, z23 = c[2].z - c[3].z;

    sx[0] = iv * (y13*z12 - y12*z13);
    sx[1] = iv * (y02*z23 - y23*z02);
    sx[2] = iv * (y13*z03 - y03*z13);
    sx[3] = iv * (y01*z02 - y02*z01);

    sy[0] = iv * (z13*x12 - z12*x13);
    sy[1] = iv * (z02*x23 - z23*x02);
    sy[2] = iv * (z13*x03 - z03*x13);
    sy[3] = iv * (z01*x02 - z02*x01);

    sz[0] = iv * (x13*y12 - x12*y13);
    sz[1] = iv * (x02*y23 - x23*y02);
    sz[2] = iv * (x13*y03 - x03*y13);
    sz[3] = iv * (x01*y02 - x02*y01);
}

// geometry.cxx:36-56
__device__ __forceinline__ double tet_volume(const d4 c[4])
{
    double x01 = c[0].x - c[1].x, x12 = c[1].x - c[2].x, x23 = c[2].x - c[3].x;
    double y01 = c[0].y - c[1].y, y12 = c[1].y - c[2].y, y23 = c[2].y - c[3].y;
    double z01 = c[0].z - c[1].z, z12 = c[1].z - c[2].z, z23 = c[2].z - c[3].z;
    return (x01*(y23*z12 - y12*z23) +
            x12*(y01*z23 - y23*z01) +
            x23*(y12*z01 - y01*z12)) / 6;
}

// geometry.cxx:77-107 (THREED)
__device__ __forceinline__ double tri_area(const d4 &a, const d4 &b, const d4 &c)
{
    double ab0 = b.x - a.x, ab1 = b.y - a.y, ab2 = b.z - a.z;
    double ac0 = c.x - a.x, ac1 = c.y - a.y, ac2 = c.z - a.z;
    double d0 = ab1*ac2 - ab2*ac1;
    double d1 = ab2*ac0 - ab0*ac2;
    double d2 = ab0*ac1 - ab1*ac0;
    return sqrt(d0*d0 + d1*d1 + d2*d2) / 2;
}
// the radicand alone: sqrt is monotone and correctly rounded, so max_i sqrt(x_i) = sqrt(max_i x_i) to the bit --
// the largest of several areas costs one square root
__device__ __forceinline__ double tri_area_sq4(const d4 &a, const d4 &b, const d4 &c)
{
    double ab0 = b.x - a.x, ab1 = b.y - a.y, ab2 = b.z - a.z;
    double ac0 = c.x - a.x, ac1 = c.y - a.y, ac2 = c.z - a.z;
    double d0 = ab1*ac2 - ab2*ac1;
    double d1 = ab2*ac0 - ab0*ac2;
    double d2 = ab0*ac1 - ab1*ac0;
    return d0*d0 + d1*d1 + d2*d2;
}

__device__ __forceinline__ double trace3(const double *s) { return s[0] + s[1] + s[2]; }

// utils.hpp:222-231
__device__ __forceinline__ double second_invariant2(const double *t)
{
    double a = (t[0] + t[1] + t[2]) / 3;
    return (0.5 * ((t[0]-a)*(t[0]-a) + (t[1]-a)*(t[1]-a) + (t[2]-a)*(t[2]-a))
            + t[3]*t[3] + t[4]*t[4] + t[5]*t[5]);
}

// ---------------------------------------------------------------------------------
// Material properties (matprops.cxx).  `mk` points at the element's marker counts.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double harmonic_mean(const double *s, const int *n, int nmat)
{
    if (nmat == 1) return s[0];               // matprops.cxx:136
    double result = 0; int m = 0;
    for (int i = 0; i < nmat; i++) { if (n[i] == 0) continue; result += n[i] / s[i]; m += n[i]; }
    return m / result;
}

__device__ __forceinline__ double arithmetic_mean(const double *s, const int *n, int nmat)
{
    if (nmat == 1) return s[0];               // matprops.cxx:118
    double result = 0; int m = 0;
    for (int i = 0; i < nmat; i++) { if (n[i] == 0) continue; result += n[i] * s[i]; m += n[i]; }
    return result / m;
}

// The marker counts of one element as the property functions see them.  Most elements hold
// markers of ONE material: for those the engine keeps (material, count) in a single int
// (des_dev.hip: mono[]) instead of reading the nmat counts; count(m) then answers from registers.
// The loops still run over every material with a wave-uniform m (so the material constants stay
// scalar loads) and do exactly the reference's arithmetic.
struct Mix {
    const int *mk;          // [nmat] counts of a mixed element, or NULL
    int mat, cnt;           // single-material element: its material and marker count
    __device__ __forceinline__ int count(int m) const { return mk ? mk[m] : (m == mat ? cnt : 0); }
};

// matprops.cxx:642-664; T = mean nodal temperature of the element
__device__ __forceinline__ double mat_rho(const des_params *p, const Mix &mx, double T)
{
    const double celsius0 = 273;
    double TinCelsius = T - celsius0;
    double result = 0;
    int n = 0;
    if (!mx.mk && p->nmat > 1) {
        // single-material element: the other terms are x * 0 added to a positive sum
        result += p->rho0[mx.mat] * (1 - p->alpha[mx.mat] * TinCelsius) * mx.cnt;
        return result / mx.cnt;
    }
    for (int m = 0; m < p->nmat; m++) {
        const int k = mx.count(m);
        result += p->rho0[m] * (1 - p->alpha[m] * TinCelsius) * k;
        n += k;
    }
    return result / n;
}

// The six libm functions of the stress update, as a policy: ROCm's ocml (default) or the
// portable set of des_libm.hpp, which a CPU build reproduces to the bit (DES_LIBM=portable).
struct MathOcml {
    static __device__ __forceinline__ void stage_begin() {}
    static __device__ __forceinline__ void stage_end() {}
    static __device__ __forceinline__ double pow(double a, double b) { return ::pow(a, b); }
    static __device__ __forceinline__ double exp(double a) { return ::exp(a); }
    static __device__ __forceinline__ double sin(double a) { return ::sin(a); }
    static __device__ __forceinline__ double cos(double a) { return ::cos(a); }
    static __device__ __forceinline__ double tan(double a) { return ::tan(a); }
    static __device__ __forceinline__ void sincos(double a, double *s, double *c) { ::sincos(a, s, c); }
    static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
};
struct MathPortable {
    static __device__ __forceinline__ void stage_begin() { deslibm::lds_stage_begin(); }   // whole workgroup, top of the kernel
    static __device__ __forceinline__ void stage_end() { deslibm::lds_stage_end(); }       // barrier before the first call
    static __device__ __forceinline__ double pow(double a, double b) { return deslibm::pow(a, b); }
    static __device__ __forceinline__ double exp(double a) { return deslibm::exp(a); }
    static __device__ __forceinline__ double sin(double a) { return deslibm::sin(a); }
    static __device__ __forceinline__ double cos(double a) { return deslibm::cos(a); }
    static __device__ __forceinline__ double tan(double a) { return deslibm::tan(a); }
    static __device__ __forceinline__ void sincos(double a, double *s, double *c) { deslibm::sincos(a, s, c); }
    static __device__ __forceinline__ double atan2(double y, double x) { return deslibm::atan2(y, x); }
};

struct ViscTerms { double pow_edot[DES_MAX_MAT], coef_term[DES_MAX_MAT], nR[DES_MAX_MAT]; };

// matprops.cxx:333-377
template <class M>
__device__ __forceinline__ double mat_visc(const des_params *p, const ViscTerms *vt, const Mix &mx,
                                           double T, const double *s, const double *edot6)
{
    const double min_strain_rate = 1e-30;
    double s0 = trace3(s) / 3;
    double edot = sqrt(second_invariant2(edot6));
    edot = fmax(edot, min_strain_rate);
    double result = 0;
    int n = 0;
    for (int m = 0; m < p->nmat; m++) {
        const int marker_count = mx.count(m);
        if (marker_count == 0) continue;
        double visc0 = 0.25 * M::pow(edot, vt->pow_edot[m]) * vt->coef_term[m]
            * M::exp((p->visc_activation_energy[m] + p->visc_activation_volume[m] * s0)
                  / (vt->nR[m] * T)) * 1e6;
        result += marker_count / visc0;
        n += marker_count;
    }
    double visc = n / result;
    visc = fmin(fmax(visc, p->visc_min), p->visc_max);
    return visc;
}

// matprops.cxx:380-418 + 589-606
// pptab (may be null): the five results for a single-material element by (material, marker count, weakening regime),
// [nmat][DES_PPTAB_CNT][3][5], filled once per engine by THIS function (des_dev.hip: k_pptab) -- outside the linear
// weakening range the results depend on nothing else, so every such element reads five doubles instead of running
// sin, tan, a square root and five divisions.  Regimes: 0 pls < pls0; 1 pls == pls0 (the interpolation at q = 0:
// the values of regime 0, but its hardening modulus); 2 pls >= pls1.  Anything else is computed below.
#define DES_PPTAB_CNT 64
template <class M>
__device__ __forceinline__ void plastic_props(const des_params *p, const Mix &mx, double pls,
                                              double &amc, double &anphi, double &anpsi,
                                              double &hardn, double &ten_max, const double *__restrict__ pptab = nullptr)
{
    if (pptab && !mx.mk && mx.cnt > 0 && mx.cnt < DES_PPTAB_CNT) {
        // the material's weakening range through wave-uniform (scalar) loads and selects: indexing the two arrays with the
        // lane's own material is two dependent vector loads in front of the table row's (-2 us of 72 at 1M tets)
        double p0 = p->pls0[0], p1 = p->pls1[0];
        for (int m = 1; m < p->nmat; m++)
            if (mx.mat == m) { p0 = p->pls0[m]; p1 = p->pls1[m]; }
        const int regime = (pls < p0) ? 0 : ((pls < p1) ? ((pls == p0) ? 1 : -1) : 2);
        if (regime >= 0) {
            const double *t = pptab + (((size_t)mx.mat * DES_PPTAB_CNT + mx.cnt) * 3 + regime) * 5;
            amc = t[0]; anphi = t[1]; anpsi = t[2]; hardn = t[3]; ten_max = t[4];
            return;
        }
    }
    double c = 0, f = 0, d = 0, h = 0;
    int n = 0;
    for (int m = 0; m < p->nmat; m++) {
        int k = mx.count(m);
        if (k == 0) continue;
        n += k;
        if (pls < p->pls0[m]) {
            c += p->cohesion0[m] * k;
            f += p->friction_angle0[m] * k;
            d += p->dilation_angle0[m] * k;
            h += 0;
        } else if (pls < p->pls1[m]) {
            double q = (pls - p->pls0[m]) / (p->pls1[m] - p->pls0[m]);
            c += (p->cohesion0[m] + q * (p->cohesion1[m] - p->cohesion0[m])) * k;
            f += (p->friction_angle0[m] + q * (p->friction_angle1[m] - p->friction_angle0[m])) * k;
            d += (p->dilation_angle0[m] + q * (p->dilation_angle1[m] - p->dilation_angle0[m])) * k;
            h += (p->cohesion1[m] - p->cohesion0[m]) / (p->pls1[m] - p->pls0[m]) * k;
        } else {
            c += p->cohesion1[m] * k;
            f += p->friction_angle1[m] * k;
            d += p->dilation_angle1[m] * k;
            h += 0;
        }
    }
    double cohesion = c / n, phi = f / n, psi = d / n;
    hardn = h / n;

    const double DEG2RAD = M_PI / 180;
    // the reference's own calls: sin of both angles, tan of the friction angle (matprops.cxx:598-605) -- with the
    // portable libm they return the C library's bits (des_libm_trig.hpp); sin(0) is exactly 0 there, so the usual
    // zero dilation angle needs no call at all
    double sphi = M::sin(phi * DEG2RAD);
    double spsi = (psi == 0) ? 0.0 : M::sin(psi * DEG2RAD);
    anphi = (1 + sphi) / (1 - sphi);
    anpsi = (1 + spsi) / (1 - spsi);
    amc = 2 * cohesion * sqrt(anphi);
    ten_max = (phi == 0) ? p->tension_max : fmin(p->tension_max, cohesion / M::tan(phi * DEG2RAD));
}

// ---------------------------------------------------------------------------------
// 3x3 symmetric eigen-solvers (J. Kopp, arXiv:physics/0610206; the reference vendors the
// C version under 3x3-C/).  Register-only: matrices are unrolled into scalars so nothing
// is indexed at run time (no scratch).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double sqr(double x) { return x * x; }

// 3x3-C/dsyevc3.c:31-80.  a = {A00, A11, A22, A01, A02, A12}
template <class M>
__device__ __forceinline__ void dsyevc3(const double *a, double w[3])
{
    const double sqrt3 = 1.73205080756887729352744634151;
    const double A00 = a[0], A11 = a[1], A22 = a[2], A01 = a[3], A02 = a[4], A12 = a[5];
    double de = A01 * A12;
    double dd = sqr(A01);
    double ee = sqr(A12);
    double ff = sqr(A02);
    double m  = A00 + A11 + A22;
    double c1 = (A00*A11 + A00*A22 + A11*A22) - (dd + ee + ff);
    double c0 = A22*dd + A00*ee + A11*ff - A00*A11*A22 - 2.0 * A02*de;

    double p = sqr(m) - 3.0*c1;
    double q = m*(p - (3.0/2.0)*c1) - (27.0/2.0)*c0;
    double sqrt_p = sqrt(fabs(p));

    double phi = 27.0 * (0.25*sqr(c1)*(p - c1) + c0*(q + 27.0/4.0*c0));
    phi = (1.0/3.0) * M::atan2(sqrt(fabs(phi)), q);

    // cos(phi) and sin(phi) of ONE argument reach the C library as one sincos call in any optimised build of the
    // reference (gcc turns the pair into it from -O1 on; 3x3-C/Makefile: -O3 -ffast-math), and glibc's sincos is not
    // bit for bit its sin and cos (des_libm_trig.hpp: g_sincos)
    double sn_phi, cs_phi;
    M::sincos(phi, &sn_phi, &cs_phi);
    double c = sqrt_p*cs_phi;
    double s = (1.0/sqrt3)*sqrt_p*sn_phi;

    w[1]  = (1.0/3.0)*(m - c);
    w[2]  = w[1] + s;
    w[0]  = w[1] + c;
    w[1] -= s;
}

// Givens update of two eigenvector columns inside the QL sweep (dsyevq3.c:329-335)
#define DES_QL_ROT(QA, QB) { t = QB; QB = s*QA + c*t; QA = c*QA - s*t; }

// 3x3-C/dsytrd3.c:379-455 + dsyevq3.c:245-350: Householder tridiagonalisation, then QL
// with implicit shifts.  Q columns are returned in q[r][col].  Rarely taken (degenerate
// or near-degenerate tensors).  It takes / returns everything BY VALUE -- with pointer arguments the caller's
// eigenvector matrix had to live in memory, and the Mohr-Coulomb return (which calls this once in a blue moon)
// kept it in scratch for every element it handles (~40 scratch accesses of 16 B per call) -- and indexes
// nothing at run time (below), so it and the return mapping are inlined into the stress update without a
// scratch frame anywhere: the one-pass update went 127-138 -> 115-119 us with 2-10 % of 1.6M tets yielding.
#ifndef DES_RM_INLINE
#define DES_RM_INLINE __forceinline__
#endif
struct Eig3 { double q[3][3]; double w[3]; int rc; };
__device__ DES_RM_INLINE Eig3 dsyevq3_core(double A00, double A11, double A22, double A01, double A02, double A12)
{
    Eig3 R;
    double *w = R.w;
    double e0, e1, e2 = 0;
    double Q11 = 1, Q12 = 0, Q21 = 0, Q22 = 1;     // Q00 = 1, Q01 = Q02 = Q10 = Q20 = 0
    {
        double h = sqr(A01) + sqr(A02);
        double g = (A01 > 0) ? -sqrt(h) : sqrt(h);
        e0 = g;
        double f = g * A01;
        double u1 = A01 - g, u2 = A02;
        double omega = h - f;
        if (omega > 0.0) {
            omega = 1.0 / omega;
            double K = 0.0;
            f = A11 * u1 + A12 * u2;
            double q1 = omega * f;
            K += u1 * f;
            f = A12 * u1 + A22 * u2;
            double q2 = omega * f;
            K += u2 * f;
            K *= 0.5 * sqr(omega);
            q1 = q1 - K * u1;
            q2 = q2 - K * u2;
            w[0] = A00;
            w[1] = A11 - 2.0*q1*u1;
            w[2] = A22 - 2.0*q2*u2;
            f = omega * u1;
            Q11 = Q11 - f*u1;
            Q21 = Q21 - f*u2;
            f = omega * u2;
            Q12 = Q12 - f*u1;
            Q22 = Q22 - f*u2;
            e1 = A12 - q1*u2 - u1*q2;
        } else {
            w[0] = A00; w[1] = A11; w[2] = A22;
            e1 = A12;
        }
    }
    // Everything below is indexed by l, m, i in {0, 1, 2} that depend on the data: w, e and the columns of q live
    // in scalars and are picked by selects (DES_G3 / DES_P3), so that nothing is addressed at run time -- no
    // scratch frame, and the whole solver can be inlined into the stress update.
    double w0 = w[0], w1 = w[1], w2 = w[2];
    double q00 = 1, q01 = 0, q02 = 0, q10 = 0, q11 = Q11, q12 = Q12, q20 = 0, q21 = Q21, q22 = Q22;
#define DES_G3(a, i) ((i) == 0 ? a##0 : ((i) == 1 ? a##1 : a##2))
#define DES_P3(a, i, v) { const double v_ = (v); if ((i) == 0) a##0 = v_; else if ((i) == 1) a##1 = v_; else a##2 = v_; }
    double g, r, p, f, b, s, c, t;
    int rc = 0;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        int nIter = 0;
        while (rc == 0) {
            int m;
            for (m = l; m <= 1; m++) {
                g = fabs(DES_G3(w, m)) + fabs(DES_G3(w, m + 1));
                if (fabs(DES_G3(e, m)) + g == g) break;
            }
            if (m == l) break;
            if (nIter++ >= 30) { rc = -1; break; }

            const double wl = DES_G3(w, l), el = DES_G3(e, l);
            g = (DES_G3(w, l + 1) - wl) / (el + el);
            r = sqrt(sqr(g) + 1.0);
            if (g > 0) g = DES_G3(w, m) - wl + el/(g + r);
            else       g = DES_G3(w, m) - wl + el/(g - r);

            s = c = 1.0;
            p = 0.0;
            for (int i = m-1; i >= l; i--) {
                const double ei = DES_G3(e, i);
                f = s * ei;
                b = c * ei;
                if (fabs(f) > fabs(g)) {
                    c      = g / f;
                    r      = sqrt(sqr(c) + 1.0);
                    DES_P3(e, i + 1, f * r);
                    c     *= (s = 1.0/r);
                } else {
                    s      = f / g;
                    r      = sqrt(sqr(s) + 1.0);
                    DES_P3(e, i + 1, g * r);
                    s     *= (c = 1.0/r);
                }
                g = DES_G3(w, i + 1) - p;
                r = (DES_G3(w, i) - g)*s + 2.0*c*b;
                p = s * r;
                DES_P3(w, i + 1, g + p);
                g = c*r - b;
                if (i == 0) { DES_QL_ROT(q00, q01); DES_QL_ROT(q10, q11); DES_QL_ROT(q20, q21); }
                else        { DES_QL_ROT(q01, q02); DES_QL_ROT(q11, q12); DES_QL_ROT(q21, q22); }
            }
            DES_P3(w, l, DES_G3(w, l) - p);
            DES_P3(e, l, g);
            DES_P3(e, m, 0.0);
        }
    }
#undef DES_G3
#undef DES_P3
    w[0] = w0; w[1] = w1; w[2] = w2;
    R.q[0][0] = q00; R.q[0][1] = q01; R.q[0][2] = q02;
    R.q[1][0] = q10; R.q[1][1] = q11; R.q[1][2] = q12;
    R.q[2][0] = q20; R.q[2][1] = q21; R.q[2][2] = q22;
    R.rc = rc;
    return R;
}
__device__ __forceinline__ int dsyevq3(const double *a, double Q[3][3], double w[3])
{
    const Eig3 R = dsyevq3_core(a[0], a[1], a[2], a[3], a[4], a[5]);
    w[0] = R.w[0]; w[1] = R.w[1]; w[2] = R.w[2];           // (the QL sweep works on w in place, converged or not)
    if (R.rc != 0) return R.rc;                          // no convergence: Q is not written (dsyevq3.c:283)
    Q[0][0] = R.q[0][0]; Q[0][1] = R.q[0][1]; Q[0][2] = R.q[0][2];
    Q[1][0] = R.q[1][0]; Q[1][1] = R.q[1][1]; Q[1][2] = R.q[1][2];
    Q[2][0] = R.q[2][0]; Q[2][1] = R.q[2][1]; Q[2][2] = R.q[2][2];
    return 0;
}

// 3x3-C/dsyevh3.c:112-215
// *fallback (diagnostics, des_dev_eigen_eval) is set when the QL solver took over
template <class M>
__device__ __forceinline__ void dsyevh3(const double *a, double Q[3][3], double w[3], int *fallback = nullptr)
{
    const double A00 = a[0], A11 = a[1], A01 = a[3], A02 = a[4], A12 = a[5];
    dsyevc3<M>(a, w);

    double t = fabs(w[0]), u;
    if ((u = fabs(w[1])) > t) t = u;
    if ((u = fabs(w[2])) > t) t = u;
    if (t < 1.0) u = t;
    else         u = sqr(t);
    double error = 256.0 * DBL_EPSILON * sqr(u);

    Q[0][1] = A01*A12 - A02*A11;
    Q[1][1] = A02*A01 - A12*A00;
    Q[2][1] = sqr(A01);

    Q[0][0] = Q[0][1] + A02*w[0];
    Q[1][0] = Q[1][1] + A12*w[0];
    Q[2][0] = (A00 - w[0]) * (A11 - w[0]) - Q[2][1];
    double norm = sqr(Q[0][0]) + sqr(Q[1][0]) + sqr(Q[2][0]);

    if (norm <= error) { if (fallback) *fallback = 1; dsyevq3(a, Q, w); return; }
    norm = sqrt(1.0 / norm);
    Q[0][0] = Q[0][0] * norm; Q[1][0] = Q[1][0] * norm; Q[2][0] = Q[2][0] * norm;

    Q[0][1] = Q[0][1] + A02*w[1];
    Q[1][1] = Q[1][1] + A12*w[1];
    Q[2][1] = (A00 - w[1]) * (A11 - w[1]) - Q[2][1];
    norm = sqr(Q[0][1]) + sqr(Q[1][1]) + sqr(Q[2][1]);
    if (norm <= error) { if (fallback) *fallback = 1; dsyevq3(a, Q, w); return; }
    norm = sqrt(1.0 / norm);
    Q[0][1] = Q[0][1] * norm; Q[1][1] = Q[1][1] * norm; Q[2][1] = Q[2][1] * norm;

    Q[0][2] = Q[1][0]*Q[2][1] - Q[2][0]*Q[1][1];
    Q[1][2] = Q[2][0]*Q[0][1] - Q[0][0]*Q[2][1];
    Q[2][2] = Q[0][0]*Q[1][1] - Q[1][0]*Q[0][1];
}

// rheology.cxx:23-45: compare-swap network (0,1),(1,2),(0,1), fully unrolled
#define DES_SWAP_P(i, j) if (p[i] > p[j]) { double tmp_ = p[i]; p[i] = p[j]; p[j] = tmp_; }
#define DES_SWAP_PV(i, j) if (p[i] > p[j]) { double tmp_ = p[i]; p[i] = p[j]; p[j] = tmp_; \
    for (int r_ = 0; r_ < 3; ++r_) { double b_ = v[r_][i]; v[r_][i] = v[r_][j]; v[r_][j] = b_; } }

// rheology.cxx:63-71
template <class M>
__device__ __forceinline__ void principal_values3(const double *s, double p[3])
{
    dsyevc3<M>(s, p);
    DES_SWAP_P(0, 1) DES_SWAP_P(1, 2) DES_SWAP_P(0, 1)
}

// rheology.cxx:76-84
template <class M>
__device__ __forceinline__ void principal_stresses3(const double *s, double p[3], double v[3][3], int *fallback = nullptr)
{
    dsyevh3<M>(s, v, p, fallback);
    DES_SWAP_PV(0, 1) DES_SWAP_PV(1, 2) DES_SWAP_PV(0, 1)
}

// rheology.cxx:248-260
__device__ __forceinline__ void elastic(double bulkm, double shearm, const double *de, double *s)
{
    double lambda = bulkm - 2. / 3 * shearm;
    double dev = trace3(de);
    for (int i = 0; i < 3; ++i) s[i] += 2 * shearm * de[i] + lambda * dev;
    for (int i = 3; i < 6; ++i) s[i] += 2 * shearm * de[i];
}

// rheology.cxx:277-295
__device__ __forceinline__ void maxwell(double bulkm, double shearm, double viscosity, double dt,
                                        double dv, const double *de, double *s)
{
    double tmp = 0.5 * dt * shearm / viscosity;
    double f1 = 1 - tmp;
    double f2 = 1 / (1 + tmp);
    double dev = trace3(de) / 3;
    double s0 = trace3(s) / 3;
    for (int i = 0; i < 3; ++i)
        s[i] = ((s[i] - s0) * f1 + 2 * shearm * (de[i] - dev)) * f2 + s0 + bulkm * dv;
    for (int i = 3; i < 6; ++i)
        s[i] = (s[i] * f1 + 2 * shearm * de[i]) * f2;
}

// rheology.cxx:298-310
__device__ __forceinline__ void viscous(double bulkm, double viscosity, double total_dv,
                                        const double *edot, double *s)
{
    double dev = trace3(edot) / 3;
    for (int i = 0; i < 3; ++i) s[i] = 2 * viscosity * (edot[i] - dev) + bulkm * total_dv;
    for (int i = 3; i < 6; ++i) s[i] = 2 * viscosity * edot[i];
}

// The Mohr-Coulomb return after the pre-filter said "maybe yielding" (rheology.cxx:363-475).
// ~0.2 % of the elements get here.  Inlined (DES_RM_INLINE; round 1 kept it out of line, which cost every
// caller a scratch frame for the call); the stress travels BY VALUE (in registers): passing a pointer to
// the caller's array would force that array -- the stress every element works on -- into scratch memory.
// mode: the reference's failure_mode (0 none, 1 tensile, 10 shear; a local there, rheology.cxx:322)
// + 100 when dsyevh3 fell back to dsyevq3 -- only read by des_dev_elasto_plastic_eval.
struct Stress7 { double s0, s1, s2, s3, s4, s5, depls; int mode; };

template <class M>
__device__ DES_RM_INLINE Stress7 mohr_coulomb_return(double bulkm, double shearm, double amc, double anphi,
                                                    double anpsi, double hardn, double ten_max, Stress7 io)
{
    double s[6] = {io.s0, io.s1, io.s2, io.s3, io.s4, io.s5};
    io.depls = 0;
    double p[3], v[3][3];
    int ql = 0;
    principal_stresses3<M>(s, p, v, &ql);
    io.mode = 100 * ql;

    double fs = p[0] - p[2] * anphi + amc;
    double ft = p[2] - ten_max;
    if (fs > 0 && ft < 0) return io;

    double pa = sqrt(1 + anphi*anphi) + anphi;
    double ps = ten_max * anphi - amc;
    double h = p[2] - ten_max + pa * (p[0] - ps);
    double a1 = bulkm + 4. / 3 * shearm;
    double a2 = bulkm - 2. / 3 * shearm;

    double alam, depls;
    io.mode += (h < 0) ? 10 : 1;
    if (h < 0) {
        alam = fs / (a1 - a2*anpsi + a1*anphi*anpsi - a2*anphi + 2*sqrt(anphi)*hardn);
        p[0] -= alam * (a1 - a2 * anpsi);
        p[1] -= alam * (a2 - a2 * anpsi);
        p[2] -= alam * (a2 - a1 * anpsi);
        depls = fabs(alam) * sqrt((7 + 4*anpsi + 7*anpsi*anpsi) / 18);
    } else {
        alam = ft / a1;
        p[0] -= alam * a2;
        p[1] -= alam * a2;
        p[2] -= alam * a1;
        depls = fabs(alam) * sqrt(7. / 18);
    }

    double ss[3][3] = {{0,0,0},{0,0,0},{0,0,0}};
    for (int m = 0; m < 3; m++)
        for (int n = m; n < 3; n++)
            for (int k = 0; k < 3; k++)
                ss[m][n] += v[m][k] * v[n][k] * p[k];
    io.s0 = ss[0][0]; io.s1 = ss[1][1]; io.s2 = ss[2][2];
    io.s3 = ss[0][1]; io.s4 = ss[0][2]; io.s5 = ss[1][2];
    io.depls = depls;
    return io;
}

// rheology.cxx:312-484 (THREED).  DEFER = 1: an element that gets past the pre-filter is not
// returned to the yield surface here; *defer is set and the caller hands the element to the
// second pass (the return mapping with dsyevh3 / dsyevq3 needs 40 more registers than the rest
// of the stress update and is taken by well under 1 % of the elements of a typical model).
template <class M, int DEFER = 0>
__device__ __forceinline__ double elasto_plastic(double bulkm, double shearm, double amc, double anphi,
                                                 double anpsi, double hardn, double ten_max,
                                                 const double *de, double *s, bool *defer = nullptr, int *mode = nullptr)
{
    elastic(bulkm, shearm, de, s);
    const double YIELD_PREFILTER_MARGIN = 1e-2;            // rheology.cxx:18
    {
        // Bound test before the reference's eigenvalue pre-filter.  Every eigenvalue of s lies
        // in [q - r, q + r] with q = tr(s)/3 and r = (2/3) sqrt(p), p = tr(s)^2 - 3 c1 =
        // (3/2)|dev s|^2 (the same p dsyevc3 forms).  If both pre-filter inequalities hold for
        // that whole interval with TWICE the reference's band, they hold for the eigenvalues
        // the reference computes (Cardano error <= 6.6e-4 max|lambda|, band >= 1e-2 max|lambda|,
        // rheology.cxx:14-18): the reference returns here too, with the same trial stress.
        // Saves atan2/cos/sin for the ~99 % of elements that are nowhere near yield;
        // anything not provably safe falls through to the reference's own test below.
        const double m = s[0] + s[1] + s[2];
        const double c1 = (s[0]*s[1] + s[0]*s[2] + s[1]*s[2]) - (s[3]*s[3] + s[4]*s[4] + s[5]*s[5]);
        const double q = m / 3;
        const double r = (2.0 / 3.0) * sqrt(fabs(m*m - 3.0*c1));
        const double lo = q - r, hi = q + r;
        const double amax = fmax(fabs(lo), fabs(hi));
        const double band2 = 2 * YIELD_PREFILTER_MARGIN * (amax + anphi * amax + fabs(amc));
        if (anphi >= 0 && lo - hi * anphi + amc > band2 && hi - ten_max < -band2)
            return 0;
    }
    {
        double pf[3];
        principal_values3<M>(s, pf);
        const double band = YIELD_PREFILTER_MARGIN * (fabs(pf[0]) + anphi * fabs(pf[2]) + fabs(amc));
        if (pf[0] - pf[2] * anphi + amc > band && pf[2] - ten_max < -band)
            return 0;
    }
    if (defer) *defer = true;                  // past the pre-filter (counted; DEFER: handed on)
    if (DEFER) return 0.0;
    Stress7 io = {s[0], s[1], s[2], s[3], s[4], s[5], 0.0, 0};
    io = mohr_coulomb_return<M>(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max, io);
    if (mode) *mode = io.mode;
    s[0] = io.s0; s[1] = io.s1; s[2] = io.s2; s[3] = io.s3; s[4] = io.s4; s[5] = io.s5;
    return io.depls;
}

// fields.cxx:791-805
__device__ __forceinline__ void jaumann_rate_3d(double *s, double dt, double w3, double w4, double w5)
{
    double s_inc[6];
    s_inc[0] = -2.0 * s[3] * w3 - 2.0 * s[4] * w4;
    s_inc[1] =  2.0 * s[3] * w3 - 2.0 * s[5] * w5;
    s_inc[2] =  2.0 * s[4] * w4 + 2.0 * s[5] * w5;
    s_inc[3] = s[0] * w3 - s[1] * w3 - s[4] * w5 - s[5] * w4;
    s_inc[4] = s[0] * w4 - s[2] * w4 + s[3] * w5 - s[5] * w3;
    s_inc[5] = s[1] * w5 - s[2] * w5 + s[3] * w4 + s[4] * w3;
    for (int i = 0; i < 6; ++i) s[i] += dt * s_inc[i];
}

// matprops.cxx:12-101, 153-174.  Table in kilobar (1e8 Pa); rows: PREM, modified PREM.
__device__ __forceinline__ double ref_pressure(const des_params *p, double z)
{
    double depth = -z;
    if (p->ref_pressure_option == 0)
        return p->rho0[p->mattype_ref] * p->gravity * depth;
    const double ref_depth[46] = {
        0e3, 3e3, 15e3, 24.4e3, 40e3, 60e3, 80e3, 115e3, 150e3, 185e3, 220e3, 265e3, 310e3,
        355e3, 400e3, 450e3, 500e3, 550e3, 600e3, 635e3, 670e3, 721e3, 771e3, 871e3, 971e3,
        1071e3, 1171e3, 1271e3, 1371e3, 1471e3, 1571e3, 1671e3, 1771e3, 1871e3, 1971e3,
        2071e3, 2171e3, 2271e3, 2371e3, 2471e3, 2571e3, 2671e3, 2741e3, 2771e3, 2871e3, 2891e3 };
    const double ref_p[46] = {
        0e8, 0.3e8, 3.3e8, 6.0e8, 11.2e8, 17.8e8, 24.5e8, 36.1e8, 47.8e8, 59.4e8, 71.1e8,
        86.4e8, 102.0e8, 117.7e8, 133.5e8, 152.2e8, 171.3e8, 190.7e8, 210.4e8, 224.3e8,
        238.3e8, 260.7e8, 282.9e8, 327.6e8, 372.8e8, 418.6e8, 464.8e8, 511.6e8, 558.9e8,
        606.8e8, 655.2e8, 704.1e8, 753.5e8, 803.6e8, 854.3e8, 905.6e8, 957.6e8, 1010.3e8,
        1063.8e8, 1118.2e8, 1173.4e8, 1229.7e8, 1269.7e8, 1287.0e8, 1345.6e8, 1357.5e8 };
    if (depth <= 0) return 0;
    int n;
    for (n = 1; n < 46; n++)
        if (depth <= ref_depth[n]) break;
    double p0 = ref_p[n-1], p1 = ref_p[n];
    if (p->ref_pressure_option == 2) {
        const double mod[4] = {0e8, 0.82e8, 4.1e8, 6.7e8};
        if (n-1 < 4) p0 = mod[n-1];
        if (n < 4) p1 = mod[n];
    }
    return p0 + (p1 - p0) * (depth - ref_depth[n-1]) / (ref_depth[n] - ref_depth[n-1]);
}

// ---------------------------------------------------------------------------------
// wave64 / block reductions
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
// fixed-shape butterfly: the same lanes always meet in the same order -> reproducible sum
__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// min over IEEE doubles of either sign with a 64-bit CAS (one call per block per quantity)
// min / max of NON-NEGATIVE doubles (lengths, time-step bounds, speeds): their bit patterns order
// like unsigned integers, so one native 64-bit integer atomic does it -- no compare-and-swap
// retry loop under the contention of thousands of workgroups.  The plain read in front drops
// most calls (and NaN, as fmin/fmax would); a stale read can only let a useless atomic through.
__device__ __forceinline__ void atomic_min_double(double *addr, double val)
{
    if (!(val < *(volatile double *)addr)) return;
    atomicMin((unsigned long long *)addr, (unsigned long long)__double_as_longlong(val));
}
__device__ __forceinline__ void atomic_max_double(double *addr, double val)
{
    if (!(val > *(volatile double *)addr)) return;
    atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(val));
}

} // namespace desk

#endif
