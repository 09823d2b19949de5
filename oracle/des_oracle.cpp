/* des_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the DynEarthSol explicit time step, kernel by kernel and in the
 * reference's own order (dynearthsol.cxx:768-894), on flat SoA arrays.  It exists to
 * check the HIP path: nothing under dynearthsol_amd/ may include, link or call it.
 *
 * Each function cites the reference file:line it restates.  Arithmetic keeps the
 * reference's operation order and association so that, compiled without FMA
 * contraction, the libm-free kernels are reproducible to the bit.
 * ONE deliberate deviation from the reference's loop order (listed in DESIGN.md section 4): the
 * sum of calculate_residual_force (fields.cxx:700-722 -- an OpenMP reduction there, its order open;
 * one running sum over (i, j) on one thread) is formed per block of 64 global node ids + a fixed
 * tree, as the engine forms it, so that the pseudo-transient loop takes the same decision at any
 * rank count; tests/test_reference_anchors.py holds it against the serial association.
 *
 * Pinning (see DESIGN.md "Oracle"): the reference's C++ translation units cannot be
 * built in this image (parameters.hpp:10 needs nanoflann, input.cxx:8 needs boost);
 * the vendored, self-contained 3x3-C solver can, and oracle/_ref builds it to pin the
 * eigen-solver restatement below.  Whole-step behaviour is pinned against the run
 * anchors recorded in SURVEY.md Appendix A.
 *
 * Dimension: a compile-time switch like the reference's own (-DTHREED, constants.hpp:12-25):
 * the default build restates the THREED branches (libdes_oracle.so), -DDES_NDIMS=2 the 2-D
 * (triangle) ones (libdes_oracle2d.so, same exports).
 */
#include "des_oracle.h"
#include "../dynearthsol_amd/csrc/des_libm.hpp"   /* the portable libm, for des_oracle_set_libm(1) only */

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

/* The libm calls of the path (rheology.cxx creep law / plastic_props, 3x3-C/dsyevc3.c).
 * Default: the C library's, as in the reference.  des_oracle_set_libm(1) switches them to
 * des_libm.hpp -- the functions the HIP engine uses under DES_LIBM=portable -- so that a test
 * can show that device and CPU agree to the bit once the libm is the same on both sides. */
int g_portable_libm = 0;
inline double m_pow(double a, double b) { return g_portable_libm ? deslibm::pow(a, b) : std::pow(a, b); }
inline double m_exp(double a) { return g_portable_libm ? deslibm::exp(a) : std::exp(a); }
inline double m_sin(double a) { return g_portable_libm ? deslibm::sin(a) : std::sin(a); }
inline double m_cos(double a) { return g_portable_libm ? deslibm::cos(a) : std::cos(a); }
inline double m_tan(double a) { return g_portable_libm ? deslibm::tan(a) : std::tan(a); }
// cos and sin of one argument: ONE sincos call, as an optimised build of 3x3-C/dsyevc3.c:66-67 makes it (gcc's
// sincos transformation, -O1 and up) -- written out so that it does not depend on this file's optimisation level
inline void m_sincos(double a, double *s, double *c) { if (g_portable_libm) deslibm::sincos(a, s, c); else ::sincos(a, s, c); }
inline double m_atan2(double y, double x) { return g_portable_libm ? deslibm::atan2(y, x) : std::atan2(y, x); }

#ifndef DES_NDIMS
#define DES_NDIMS 3
#endif
const int ND = DES_NDIMS;    // NDIMS, constants.hpp:12-16
const int NPE = ND + 1;      // NODES_PER_ELEM, constants.hpp:19
const int NSTR = ND * (ND + 1) / 2;   // constants.hpp:25
const int NPF = ND;          // NODES_PER_FACET, constants.hpp:60
// constants.hpp:64-76
#if DES_NDIMS == 3
const int NODE_OF_FACET[4][3] = {{1,2,3},{0,3,2},{0,1,3},{0,2,1}};
#else
const int NODE_OF_FACET[3][2] = {{1,2},{2,0},{0,1}};
#endif
const double YEAR2SEC = 365.2422 * 86400;   // constants.hpp:76
const unsigned BOUNDZ0 = 1u << 4, BOUNDZ1 = 1u << 5;
const unsigned BOUND_ANY = 0x3ffu;
const int iboundx0 = 0, iboundz0 = 4, iboundz1 = 5, iboundn0 = 6, iboundn3 = 9;
const double DEG2RAD = M_PI / 180;     // constants.hpp:77

typedef std::vector<double> dvec;
typedef std::vector<int> ivec;

} // namespace

struct des_oracle {
    des_params p;
    int nn, ne;
    int iso = 0;                        // inside isostasy_adjustment (dynearthsol.cxx:496-544)
    bool pt_jump = false;               // Param::control.PT_jump while the pseudo-transient loop runs
    bool body_force_adjustment = false; // Param::ic.has_body_force_adjustment while initial_body_force_adjustment runs (fields.cxx:690)
    long long n_pt_iterations = 0;      // iterations taken since the last des_oracle_step call began
    int g0 = 0;                         // global id of the first owned node (des_halo::owned_global_begin)
    std::vector<double> res_blocks;     // [global residual blocks] partials of the partition-independent residual (des_params.h)
    int n_return_mapping = 0;           // elements past the yield pre-filter in the last update_stress
    // topology
    ivec conn;                          // [NPE][ne]
    ivec sup_idx, sup_arr, sup_lidx;
    std::vector<unsigned> bcflag;
    ivec bf_elem[DES_NBDRY], bf_facet[DES_NBDRY], bnodes[DES_NBDRY];
    dvec bnormals, edge_vec;
    int edge_slot[DES_NBDRY * DES_NBDRY];
    int ntop, etop, ntop_elems;
    ivec top_nodes, elem_and_nodes, conn_surf, ssup_idx, ssup_arr, top_elems;
    // nodal fields
    dvec coord, vel, force, force_residual, coord0, temperature, volume_n, mass, tmass,
         hmass, ymass, dhacc, ntmp, total_dx, total_slope;
    // element fields
    dvec stressyy;                      // plane-strain out-of-plane stress (fields.cxx:75); 2-D only
    dvec stress, strain, strain_rate, plstrain, delta_plstrain, viscosity, volume,
         volume_old, dpressure, edvoldt, radiogenic, etmp, tmp_result;
    ivec elemmarkers, etmp_int;
    dvec dh, edvacc_surf, dh_n;
    bool wall_given = false; double wall[3] = {0, 0, 0};   // des_oracle_wall_set: the x0 wall's extent / the lowest node of the WHOLE mesh (2-D, decomposed)
    // Output::average_fields state (output.hpp:30-36)
    dvec stress_avg, dplstrain_avg, strain0, coord_avg0;
    double avg_time0;
    int o0, o1;                         // owned nodes [o0, o1) (whole mesh unless decomposed): reductions
    int c0, c1;                         // nodes the node loops run over: every local node
    double dt_part[6];
    double l2_part;
    int nn_global;                  // compute_dt partials: minl, dt_maxwell, dt_diffusion, gdt_min, -max_vem, -max_surf_vel
    // matprops cache (matprops.cxx:259-303): bulkm, shearm, phi, cp, k per element
    dvec c_bulkm, c_shearm, c_phi, c_cp, c_k;
    bool markers_dirty;
    // visc() material-only terms (matprops.cxx:237-250)
    double visc_pow_edot[DES_MAX_MAT], visc_coef_term[DES_MAX_MAT], visc_nR[DES_MAX_MAT];
    // clock
    double dt, time, l2_residual, max_surf_vel, max_global_vel_mag, global_dt_min;
    long long steps;
    int status;
};

namespace {

// ---------------------------------------------------------------------------------
// 3x3 symmetric eigen-solvers: restatement of J. Kopp's algorithms as vendored under
// 3x3-C/ (arXiv:physics/0610206).  The reference compiles them -O3 -ffast-math
// (3x3-C/Makefile:3-4); here they are plain IEEE.
// ---------------------------------------------------------------------------------
inline double sqr(double x) { return x * x; }

// 3x3-C/dsyevc3.c:31-80 -- Cardano, eigenvalues only
int dsyevc3(const double A[3][3], double w[3])
{
    const double sqrt3 = 1.73205080756887729352744634151;
    double de = A[0][1] * A[1][2];
    double dd = sqr(A[0][1]);
    double ee = sqr(A[1][2]);
    double ff = sqr(A[0][2]);
    double m  = A[0][0] + A[1][1] + A[2][2];
    double c1 = (A[0][0]*A[1][1] + A[0][0]*A[2][2] + A[1][1]*A[2][2]) - (dd + ee + ff);
    double c0 = A[2][2]*dd + A[0][0]*ee + A[1][1]*ff - A[0][0]*A[1][1]*A[2][2]
                - 2.0 * A[0][2]*de;

    double p = sqr(m) - 3.0*c1;
    double q = m*(p - (3.0/2.0)*c1) - (27.0/2.0)*c0;
    double sqrt_p = std::sqrt(std::fabs(p));

    double phi = 27.0 * (0.25*sqr(c1)*(p - c1) + c0*(q + 27.0/4.0*c0));
    phi = (1.0/3.0) * m_atan2(std::sqrt(std::fabs(phi)), q);

    double sn_phi, cs_phi;
    m_sincos(phi, &sn_phi, &cs_phi);
    double c = sqrt_p*cs_phi;
    double s = (1.0/sqrt3)*sqrt_p*sn_phi;

    w[1]  = (1.0/3.0)*(m - c);
    w[2]  = w[1] + s;
    w[0]  = w[1] + c;
    w[1] -= s;
    return 0;
}

// 3x3-C/dsytrd3.c:379-455 -- Householder reduction to tridiagonal form
void dsytrd3(const double A[3][3], double Q[3][3], double d[3], double e[2])
{
    const int n = 3;
    double u[3], q[3];
    double omega, f, K, h, g;

    for (int i = 0; i < n; i++) {
        Q[i][i] = 1.0;
        for (int j = 0; j < i; j++)
            Q[i][j] = Q[j][i] = 0.0;
    }

    h = sqr(A[0][1]) + sqr(A[0][2]);
    if (A[0][1] > 0) g = -std::sqrt(h);
    else             g = std::sqrt(h);
    e[0] = g;
    f    = g * A[0][1];
    u[1] = A[0][1] - g;
    u[2] = A[0][2];

    omega = h - f;
    if (omega > 0.0) {
        omega = 1.0 / omega;
        K = 0.0;
        for (int i = 1; i < n; i++) {
            f    = A[1][i] * u[1] + A[i][2] * u[2];
            q[i] = omega * f;
            K   += u[i] * f;
        }
        K *= 0.5 * sqr(omega);

        for (int i = 1; i < n; i++)
            q[i] = q[i] - K * u[i];

        d[0] = A[0][0];
        d[1] = A[1][1] - 2.0*q[1]*u[1];
        d[2] = A[2][2] - 2.0*q[2]*u[2];

        for (int j = 1; j < n; j++) {
            f = omega * u[j];
            for (int i = 1; i < n; i++)
                Q[i][j] = Q[i][j] - f*u[i];
        }
        e[1] = A[1][2] - q[1]*u[2] - u[1]*q[2];
    } else {
        for (int i = 0; i < n; i++)
            d[i] = A[i][i];
        e[1] = A[1][2];
    }
}

// 3x3-C/dsyevq3.c:245-350 -- QL with implicit shifts
int dsyevq3(const double A[3][3], double Q[3][3], double w[3])
{
    const int n = 3;
    double e[3];
    double g, r, p, f, b, s, c, t;
    int nIter, m;

    dsytrd3(A, Q, w, e);

    for (int l = 0; l < n-1; l++) {
        nIter = 0;
        while (1) {
            for (m = l; m <= n-2; m++) {
                g = std::fabs(w[m]) + std::fabs(w[m+1]);
                if (std::fabs(e[m]) + g == g)
                    break;
            }
            if (m == l)
                break;

            if (nIter++ >= 30)
                return -1;

            g = (w[l+1] - w[l]) / (e[l] + e[l]);
            r = std::sqrt(sqr(g) + 1.0);
            if (g > 0) g = w[m] - w[l] + e[l]/(g + r);
            else       g = w[m] - w[l] + e[l]/(g - r);

            s = c = 1.0;
            p = 0.0;
            for (int i = m-1; i >= l; i--) {
                f = s * e[i];
                b = c * e[i];
                if (std::fabs(f) > std::fabs(g)) {
                    c      = g / f;
                    r      = std::sqrt(sqr(c) + 1.0);
                    e[i+1] = f * r;
                    c     *= (s = 1.0/r);
                } else {
                    s      = f / g;
                    r      = std::sqrt(sqr(s) + 1.0);
                    e[i+1] = g * r;
                    s     *= (c = 1.0/r);
                }

                g = w[i+1] - p;
                r = (w[i] - g)*s + 2.0*c*b;
                p = s * r;
                w[i+1] = g + p;
                g = c*r - b;

                for (int k = 0; k < n; k++) {
                    t = Q[k][i+1];
                    Q[k][i+1] = s*Q[k][i] + c*t;
                    Q[k][i]   = c*Q[k][i] - s*t;
                }
            }
            w[l] -= p;
            e[l]  = g;
            e[m]  = 0.0;
        }
    }
    return 0;
}

// 3x3-C/dsyevh3.c:112-215 -- Cardano eigenvalues + cross-product eigenvectors,
// QL fallback when the error estimate says so
int dsyevh3(const double A[3][3], double Q[3][3], double w[3])
{
    double norm, error, t, u;

    dsyevc3(A, w);

    t = std::fabs(w[0]);
    if ((u = std::fabs(w[1])) > t) t = u;
    if ((u = std::fabs(w[2])) > t) t = u;
    if (t < 1.0) u = t;
    else         u = sqr(t);
    error = 256.0 * DBL_EPSILON * sqr(u);

    Q[0][1] = A[0][1]*A[1][2] - A[0][2]*A[1][1];
    Q[1][1] = A[0][2]*A[0][1] - A[1][2]*A[0][0];
    Q[2][1] = sqr(A[0][1]);

    Q[0][0] = Q[0][1] + A[0][2]*w[0];
    Q[1][0] = Q[1][1] + A[1][2]*w[0];
    Q[2][0] = (A[0][0] - w[0]) * (A[1][1] - w[0]) - Q[2][1];
    norm    = sqr(Q[0][0]) + sqr(Q[1][0]) + sqr(Q[2][0]);

    if (norm <= error)
        return dsyevq3(A, Q, w);
    norm = std::sqrt(1.0 / norm);
    for (int j = 0; j < 3; j++)
        Q[j][0] = Q[j][0] * norm;

    Q[0][1] = Q[0][1] + A[0][2]*w[1];
    Q[1][1] = Q[1][1] + A[1][2]*w[1];
    Q[2][1] = (A[0][0] - w[1]) * (A[1][1] - w[1]) - Q[2][1];
    norm    = sqr(Q[0][1]) + sqr(Q[1][1]) + sqr(Q[2][1]);
    if (norm <= error)
        return dsyevq3(A, Q, w);
    norm = std::sqrt(1.0 / norm);
    for (int j = 0; j < 3; j++)
        Q[j][1] = Q[j][1] * norm;

    Q[0][2] = Q[1][0]*Q[2][1] - Q[2][0]*Q[1][1];
    Q[1][2] = Q[2][0]*Q[0][1] - Q[0][0]*Q[2][1];
    Q[2][2] = Q[0][0]*Q[1][1] - Q[1][0]*Q[0][1];
    return 0;
}

// rheology.cxx:23-45 -- 3-swap sorting network, columns of v follow
void sort_principal3(double p[3], double (*v)[3])
{
    const int pairs[3][2] = {{0,1},{1,2},{0,1}};
    for (int k = 0; k < 3; ++k) {
        const int i = pairs[k][0], j = pairs[k][1];
        if (p[i] > p[j]) {
            const double tmp = p[i]; p[i] = p[j]; p[j] = tmp;
            if (v) {
                for (int r = 0; r < 3; ++r) {
                    const double b = v[r][i]; v[r][i] = v[r][j]; v[r][j] = b;
                }
            }
        }
    }
}

// rheology.cxx:48-58
void unflatten_stress3(const double *s, double a[3][3])
{
    a[0][0] = s[0]; a[1][1] = s[1]; a[2][2] = s[2];
    a[0][1] = s[3]; a[0][2] = s[4]; a[1][2] = s[5];
    a[1][0] = a[2][0] = a[2][1] = 0;   // lower triangle is never read
}

// rheology.cxx:63-71
void principal_values3(const double *s, double p[3])
{
    double a[3][3];
    unflatten_stress3(s, a);
    dsyevc3(a, p);
    sort_principal3(p, nullptr);
}

// rheology.cxx:76-84
void principal_stresses3(const double *s, double p[3], double v[3][3])
{
    double a[3][3];
    unflatten_stress3(s, a);
    dsyevh3(a, v, p);
    sort_principal3(p, v);
}

// utils.hpp:211-219
#if DES_NDIMS == 3
inline double trace3(const double *s) { return s[0] + s[1] + s[2]; }
#else
inline double trace3(const double *s) { return s[0] + s[1]; }
#endif

// utils.hpp:222-231
inline double second_invariant2(const double *t)
{
#if DES_NDIMS == 3
    double a = (t[0] + t[1] + t[2]) / 3;
    return (0.5 * ((t[0]-a)*(t[0]-a) + (t[1]-a)*(t[1]-a) + (t[2]-a)*(t[2]-a))
            + t[3]*t[3] + t[4]*t[4] + t[5]*t[5]);
#else
    return 0.25*(t[0]-t[1])*(t[0]-t[1]) + t[2]*t[2];
#endif
}

#if DES_NDIMS == 2
// rheology.cxx:86-119: Mohr circle of {XX, ZZ, XZ}; p[0] <= p[1]
void principal_stresses2(const double *s, double p[2], double &cos2t, double &sin2t)
{
    double s0 = 0.5 * (s[0] + s[1]);
    double rad = std::sqrt(second_invariant2(s));     // second_invariant, utils.hpp:234-241
    p[0] = s0 - rad;
    p[1] = s0 + rad;
    const double eps = 1e-15;
    double a = 0.5 * (s[0] - s[1]);
    double b = - rad;
    if (b < -eps) {
        cos2t = a / b;
        sin2t = s[2] / b;
    } else {
        cos2t = 1;
        sin2t = 0;
    }
}
#endif

// rheology.cxx:248-260
void elastic(double bulkm, double shearm, const double *de, double *s)
{
    double lambda = bulkm - 2. / 3 * shearm;
    double dev = trace3(de);
    for (int i = 0; i < ND; ++i)
        s[i] += 2 * shearm * de[i] + lambda * dev;
    for (int i = ND; i < NSTR; ++i)
        s[i] += 2 * shearm * de[i];
}

// rheology.cxx:277-295
void maxwell(double bulkm, double shearm, double viscosity, double dt, double dv,
             const double *de, double *s)
{
    double tmp = 0.5 * dt * shearm / viscosity;
    double f1 = 1 - tmp;
    double f2 = 1 / (1 + tmp);
    double dev = trace3(de) / ND;
    double s0 = trace3(s) / ND;
    for (int i = 0; i < ND; ++i)
        s[i] = ((s[i] - s0) * f1 + 2 * shearm * (de[i] - dev)) * f2 + s0 + bulkm * dv;
    for (int i = ND; i < NSTR; ++i)
        s[i] = (s[i] * f1 + 2 * shearm * de[i]) * f2;
}

// rheology.cxx:298-310
void viscous(double bulkm, double viscosity, double total_dv, const double *edot, double *s)
{
    double dev = trace3(edot) / ND;
    for (int i = 0; i < ND; ++i)
        s[i] = 2 * viscosity * (edot[i] - dev) + bulkm * total_dv;
    for (int i = ND; i < NSTR; ++i)
        s[i] = 2 * viscosity * edot[i];
}

const double YIELD_PREFILTER_MARGIN = 1e-2;   // rheology.cxx:18

// rheology.cxx:312-484 (has_hydraulic_diffusion == false)
void elasto_plastic(double bulkm, double shearm, double amc, double anphi, double anpsi,
                    double hardn, double ten_max, const double *de, double &depls,
                    double *s, int &failure_mode, int *past_prefilter = nullptr)
{
    elastic(bulkm, shearm, de, s);
    depls = 0;
    failure_mode = 0;

#if DES_NDIMS == 2
    // rheology.cxx:364-369, 371-483 with NDIMS = 2: pure 2-D Mohr-Coulomb in the X-Z plane
    double p[2];
    double cos2t, sin2t;
    if (past_prefilter) *past_prefilter = 1;       // no pre-filter in 2-D: every element reaches the yield test
    principal_stresses2(s, p, cos2t, sin2t);

    double fs = p[0] - p[1] * anphi + amc;
    double ft = p[1] - ten_max;
    if (fs > 0 && ft < 0)
        return;

    double pa = std::sqrt(1 + anphi*anphi) + anphi;
    double ps = ten_max * anphi - amc;
    double h = p[1] - ten_max + pa * (p[0] - ps);
    double a1 = bulkm + 4. / 3 * shearm;
    double a2 = bulkm - 2. / 3 * shearm;

    double alam;
    if (h < 0) {
        failure_mode = 10;
        alam = fs / (a1 - a2*anpsi + a1*anphi*anpsi - a2*anphi + 2*std::sqrt(anphi)*hardn);
        p[0] -= alam * (a1 - a2 * anpsi);
        p[1] -= alam * (a2 - a1 * anpsi);
        depls = std::fabs(alam) * std::sqrt((3 + 2*anpsi + 3*anpsi*anpsi) / 8);
    } else {
        failure_mode = 1;
        alam = ft / a1;
        p[0] -= alam * a2;
        p[1] -= alam * a1;
        depls = std::fabs(alam) * std::sqrt(3. / 8);
    }
    double dc2 = (p[0] - p[1]) * cos2t;
    double dss = p[0] + p[1];
    s[0] = 0.5 * (dss + dc2);
    s[1] = 0.5 * (dss - dc2);
    s[2] = 0.5 * (p[0] - p[1]) * sin2t;
#else
    double p[3];
    double v[3][3];
    {
        // eigenvalue-only pre-filter, rheology.cxx:354-361
        double pf[3];
        principal_values3(s, pf);
        const double band = YIELD_PREFILTER_MARGIN
            * (std::fabs(pf[0]) + anphi * std::fabs(pf[2]) + std::fabs(amc));
        if (pf[0] - pf[2] * anphi + amc > band && pf[2] - ten_max < -band)
            return;
    }
    if (past_prefilter) *past_prefilter = 1;       // statistics only (des_scalars::n_return_mapping)
    principal_stresses3(s, p, v);

    double fs = p[0] - p[2] * anphi + amc;
    double ft = p[2] - ten_max;
    if (fs > 0 && ft < 0)
        return;

    double pa = std::sqrt(1 + anphi*anphi) + anphi;
    double ps = ten_max * anphi - amc;
    double h = p[2] - ten_max + pa * (p[0] - ps);
    double a1 = bulkm + 4. / 3 * shearm;
    double a2 = bulkm - 2. / 3 * shearm;

    double alam;
    if (h < 0) {
        failure_mode = 10;   // shear
        alam = fs / (a1 - a2*anpsi + a1*anphi*anpsi - a2*anphi + 2*std::sqrt(anphi)*hardn);
        p[0] -= alam * (a1 - a2 * anpsi);
        p[1] -= alam * (a2 - a2 * anpsi);
        p[2] -= alam * (a2 - a1 * anpsi);
        depls = std::fabs(alam) * std::sqrt((7 + 4*anpsi + 7*anpsi*anpsi) / 18);
    } else {
        failure_mode = 1;    // tensile
        alam = ft / a1;
        p[0] -= alam * a2;
        p[1] -= alam * a2;
        p[2] -= alam * a1;
        depls = std::fabs(alam) * std::sqrt(7. / 18);
    }

    // rotate back, rheology.cxx:460-475
    double ss[3][3] = {{0,0,0},{0,0,0},{0,0,0}};
    for (int m = 0; m < 3; m++)
        for (int n = m; n < 3; n++)
            for (int k = 0; k < 3; k++)
                ss[m][n] += v[m][k] * v[n][k] * p[k];
    s[0] = ss[0][0]; s[1] = ss[1][1]; s[2] = ss[2][2];
    s[3] = ss[0][1]; s[4] = ss[0][2]; s[5] = ss[1][2];
#endif
}

#if DES_NDIMS == 2
// rheology.cxx:486-701 (plane strain: three principal stresses, two principal strains;
// has_hydraulic_diffusion == false)
void elasto_plastic2d(double bulkm, double shearm, double amc, double anphi, double anpsi,
                      double hardn, double ten_max, const double *de, double &depls,
                      double *s, double &syy, int &failure_mode)
{
    depls = 0;
    failure_mode = 0;

    double a1 = bulkm + 4. / 3 * shearm;
    double a2 = bulkm - 2. / 3 * shearm;
    double sxx = s[0] + de[1]*a2 + de[0]*a1;
    double szz = s[1] + de[0]*a2 + de[1]*a1;
    double sxz = s[2] + de[2]*2*shearm;
    syy += (de[0] + de[1]) * a2;

    double p[3];
    double cos2t, sin2t;
    int n1, n2, n3;
    {
        double s0 = 0.5 * (sxx + szz);
        double rad = 0.5 * std::sqrt((sxx-szz)*(sxx-szz) + 4*sxz*sxz);
        double si = s0 - rad;
        double sii = s0 + rad;
        const double eps = 1e-15;
        if (rad > eps) {
            cos2t = 0.5 * (szz - sxx) / rad;
            sin2t = -sxz / rad;
        } else {
            cos2t = 1;
            sin2t = 0;
        }
        if (syy > sii) {
            n1 = 0; n2 = 1; n3 = 2;
            p[0] = si; p[1] = sii; p[2] = syy;
        } else if (syy < si) {
            n1 = 1; n2 = 2; n3 = 0;
            p[0] = syy; p[1] = si; p[2] = sii;
        } else {
            n1 = 0; n2 = 2; n3 = 1;
            p[0] = si; p[1] = syy; p[2] = sii;
        }
    }

    if (p[0] >= ten_max) {
        s[0] = s[1] = syy = ten_max;
        s[2] = 0.0;
        failure_mode = 1;
        return;
    }
    if (p[1] >= ten_max) {
        p[1] = p[2] = ten_max;
        failure_mode = 2;
    } else if (p[2] >= ten_max) {
        p[2] = ten_max;
        failure_mode = 3;
    }

    double fs = p[0] - p[2] * anphi + amc;
    if (fs >= 0.0) {
        s[0] = sxx;
        s[1] = szz;
        s[2] = sxz;
        return;
    }

    failure_mode += 10;

    const double alams = fs / (a1 - a2*anpsi + a1*anphi*anpsi - a2*anphi + hardn);
    p[0] -= alams * (a1 - a2 * anpsi);
    p[1] -= alams * (a2 - a2 * anpsi);
    p[2] -= alams * (a2 - a1 * anpsi);

    depls = 0.5 * std::fabs(alams + alams * anpsi);

    if (p[0] >= ten_max) {
        s[0] = s[1] = syy = ten_max;
        s[2] = 0.0;
        failure_mode += 20;
        return;
    }
    if (p[1] >= ten_max) {
        p[1] = p[2] = ten_max;
        failure_mode += 20;
    } else if (p[2] >= ten_max) {
        p[2] = ten_max;
        failure_mode += 20;
    }

    {
        double dc2 = (p[n1] - p[n2]) * cos2t;
        double dss = p[n1] + p[n2];
        s[0] = 0.5 * (dss + dc2);
        s[1] = 0.5 * (dss - dc2);
        s[2] = 0.5 * (p[n1] - p[n2]) * sin2t;
        syy = p[n3];
    }
}
#endif

// ---------------------------------------------------------------------------------
// PREM reference pressure, matprops.cxx:12-101, 153-174
// ---------------------------------------------------------------------------------
const double prem_depth[46] = {
    0e3, 3e3, 15e3, 24.4e3, 40e3, 60e3, 80e3, 115e3, 150e3, 185e3, 220e3, 265e3, 310e3,
    355e3, 400e3, 450e3, 500e3, 550e3, 600e3, 635e3, 670e3, 721e3, 771e3, 871e3, 971e3,
    1071e3, 1171e3, 1271e3, 1371e3, 1471e3, 1571e3, 1671e3, 1771e3, 1871e3, 1971e3,
    2071e3, 2171e3, 2271e3, 2371e3, 2471e3, 2571e3, 2671e3, 2741e3, 2771e3, 2871e3, 2891e3 };
const double prem_p[46] = {
    0e8, 0.3e8, 3.3e8, 6.0e8, 11.2e8, 17.8e8, 24.5e8, 36.1e8, 47.8e8, 59.4e8, 71.1e8,
    86.4e8, 102.0e8, 117.7e8, 133.5e8, 152.2e8, 171.3e8, 190.7e8, 210.4e8, 224.3e8,
    238.3e8, 260.7e8, 282.9e8, 327.6e8, 372.8e8, 418.6e8, 464.8e8, 511.6e8, 558.9e8,
    606.8e8, 655.2e8, 704.1e8, 753.5e8, 803.6e8, 854.3e8, 905.6e8, 957.6e8, 1010.3e8,
    1063.8e8, 1118.2e8, 1173.4e8, 1229.7e8, 1269.7e8, 1287.0e8, 1345.6e8, 1357.5e8 };

double prem_pressure(double depth, bool modified)
{
    if (depth <= 0) return 0;
    int n;
    for (n = 1; n < 46; n++)
        if (depth <= prem_depth[n]) break;
    // the "modified" table differs in entries 1..3 only (matprops.cxx:77)
    double p0 = prem_p[n-1], p1 = prem_p[n];
    if (modified) {
        const double mod[4] = {0e8, 0.82e8, 4.1e8, 6.7e8};
        if (n-1 < 4) p0 = mod[n-1];
        if (n < 4) p1 = mod[n];
    }
    return p0 + (p1 - p0) * (depth - prem_depth[n-1]) / (prem_depth[n] - prem_depth[n-1]);
}

double ref_pressure(const des_params &p, double z)
{
    double depth = -z;
    double pr = 0;
    if (p.ref_pressure_option == 0)
        pr = p.rho0[p.mattype_ref] * p.gravity * depth;
    else if (p.ref_pressure_option == 1)
        pr = prem_pressure(depth, false);
    else if (p.ref_pressure_option == 2)
        pr = prem_pressure(depth, true);
    return pr;
}

// ---------------------------------------------------------------------------------
// MatProps accessors
// ---------------------------------------------------------------------------------
struct Mat {
    const des_oracle &o;
    explicit Mat(const des_oracle &o_) : o(o_) {}
    const int *markers(int e) const { return &o.elemmarkers[(size_t)e * o.p.nmat]; }

    double bulkm(int e) const { return o.c_bulkm[e]; }     // matprops.cxx:321-324
    double shearm(int e) const { return o.c_shearm[e]; }   // matprops.cxx:327-330
    double phi(int e) const { return o.c_phi[e]; }
    double cp(int e) const { return o.c_cp[e]; }
    double k(int e) const { return o.c_k[e]; }

    double elemT(int e) const {                            // matprops.cxx:338-343
        double T = 0;
        for (int i = 0; i < NPE; ++i)
            T += o.temperature[o.conn[i * o.ne + e]];
        T /= NPE;
        return T;
    }

    // matprops.cxx:642-664
    double rho(int e) const {
        const double celsius0 = 273;
        double TinCelsius = elemT(e) - celsius0;
        double result = 0;
        int n = 0;
        const int *mk = markers(e);
        for (int m = 0; m < o.p.nmat; m++) {
            result += o.p.rho0[m] * (1 - o.p.alpha[m] * TinCelsius) * mk[m];
            n += mk[m];
        }
        return result / n;
    }

    // matprops.cxx:333-377
    double visc(int e) const {
        const double min_strain_rate = 1e-30;
        double T = elemT(e);
        double s6[NSTR], e6[NSTR];
        for (int i = 0; i < NSTR; ++i) {
            s6[i] = o.stress[i * o.ne + e];
            e6[i] = o.strain_rate[i * o.ne + e];
        }
        double s0 = trace3(s6) / ND;
        double edot = std::sqrt(second_invariant2(e6));
        edot = std::max(edot, min_strain_rate);
        double result = 0;
        int n = 0;
        const int *mk = markers(e);
        for (int m = 0; m < o.p.nmat; m++) {
            const int marker_count = mk[m];
            if (marker_count == 0) continue;
            double visc0 = 0.25 * m_pow(edot, o.visc_pow_edot[m]) * o.visc_coef_term[m]
                * m_exp((o.p.visc_activation_energy[m] + o.p.visc_activation_volume[m] * s0)
                           / (o.visc_nR[m] * T)) * 1e6;
            result += marker_count / visc0;
            n += marker_count;
        }
        double visc = n / result;
        visc = std::min(std::max(visc, o.p.visc_min), o.p.visc_max);
        return visc;
    }

    // matprops.cxx:380-418
    void plastic_weakening(int e, double pls, double &cohesion, double &friction_angle,
                           double &dilation_angle, double &hardening) const {
        double c, f, d, h;
        c = f = d = h = 0;
        int n = 0;
        const int *mk = markers(e);
        const des_params &p = o.p;
        for (int m = 0; m < p.nmat; m++) {
            int k = mk[m];
            if (k == 0) continue;
            n += k;
            if (pls < p.pls0[m]) {
                c += p.cohesion0[m] * k;
                f += p.friction_angle0[m] * k;
                d += p.dilation_angle0[m] * k;
                h += 0;
            } else if (pls < p.pls1[m]) {
                double q = (pls - p.pls0[m]) / (p.pls1[m] - p.pls0[m]);
                c += (p.cohesion0[m] + q * (p.cohesion1[m] - p.cohesion0[m])) * k;
                f += (p.friction_angle0[m] + q * (p.friction_angle1[m] - p.friction_angle0[m])) * k;
                d += (p.dilation_angle0[m] + q * (p.dilation_angle1[m] - p.dilation_angle0[m])) * k;
                h += (p.cohesion1[m] - p.cohesion0[m]) / (p.pls1[m] - p.pls0[m]) * k;
            } else {
                c += p.cohesion1[m] * k;
                f += p.friction_angle1[m] * k;
                d += p.dilation_angle1[m] * k;
                h += 0;
            }
        }
        cohesion = c / n;
        friction_angle = f / n;
        dilation_angle = d / n;
        hardening = h / n;
    }

    // matprops.cxx:589-606
    void plastic_props(int e, double pls, double &amc, double &anphi, double &anpsi,
                       double &hardn, double &ten_max) const {
        double cohesion, phi_, psi;
        plastic_weakening(e, pls, cohesion, phi_, psi, hardn);
        double sphi = m_sin(phi_ * DEG2RAD);
        double spsi = m_sin(psi * DEG2RAD);
        anphi = (1 + sphi) / (1 - sphi);
        anpsi = (1 + spsi) / (1 - spsi);
        amc = 2 * cohesion * std::sqrt(anphi);
        ten_max = (phi_ == 0) ? o.p.tension_max
                              : std::min(o.p.tension_max, cohesion / m_tan(phi_ * DEG2RAD));
    }
};

// matprops.cxx:116-149 -- marker-count-weighted means; a 1-material run returns s[0]
double arithmetic_mean(const double *s, const int *n, int nmat)
{
    if (nmat == 1) return s[0];
    double result = 0;
    int m = 0;
    for (int i = 0; i < nmat; i++) {
        if (n[i] == 0) continue;
        result += n[i] * s[i];
        m += n[i];
    }
    return result / m;
}

double harmonic_mean(const double *s, const int *n, int nmat)
{
    if (nmat == 1) return s[0];
    double result = 0;
    int m = 0;
    for (int i = 0; i < nmat; i++) {
        if (n[i] == 0) continue;
        result += n[i] / s[i];
        m += n[i];
    }
    return m / result;
}

// matprops.cxx:259-303
void refresh_elem_cache(des_oracle &o)
{
    if (!o.markers_dirty) return;
    const des_params &p = o.p;
    #pragma omp parallel for
    for (int e = 0; e < o.ne; ++e) {
        const int *mk = &o.elemmarkers[(size_t)e * p.nmat];
        o.c_bulkm[e]  = harmonic_mean(p.bulk_modulus, mk, p.nmat);
        o.c_shearm[e] = harmonic_mean(p.shear_modulus, mk, p.nmat);
        o.c_phi[e]    = arithmetic_mean(p.porosity, mk, p.nmat);
        o.c_cp[e]     = arithmetic_mean(p.heat_capacity, mk, p.nmat);
        o.c_k[e]      = arithmetic_mean(p.therm_cond, mk, p.nmat);
    }
    o.markers_dirty = false;
}

// ---------------------------------------------------------------------------------
// geometry helpers
// ---------------------------------------------------------------------------------
inline void node_xyz(const dvec &a, int nn, int n, double x[ND])
{
    for (int d = 0; d < ND; ++d) x[d] = a[(size_t)d * nn + n];
}

// geometry.cxx:36-56
double tetrahedron_volume(const double *d0, const double *d1, const double *d2, const double *d3)
{
    double x01 = d0[0] - d1[0];
    double x12 = d1[0] - d2[0];
    double x23 = d2[0] - d3[0];
    double y01 = d0[1] - d1[1];
    double y12 = d1[1] - d2[1];
    double y23 = d2[1] - d3[1];
    double z01 = d0[2] - d1[2];
    double z12 = d1[2] - d2[2];
    double z23 = d2[2] - d3[2];
    return (x01*(y23*z12 - y12*z23) +
            x12*(y01*z23 - y23*z01) +
            x23*(y12*z01 - y01*z12)) / 6;
}

// geometry.cxx:77-107
double triangle_area(const double *a, const double *b, const double *c)
{
    double ab0 = b[0] - a[0], ab1 = b[1] - a[1];
    double ac0 = c[0] - a[0], ac1 = c[1] - a[1];
#if DES_NDIMS == 2
    return std::fabs(ab0*ac1 - ab1*ac0) / 2;
#else
    double ab2 = b[2] - a[2], ac2 = c[2] - a[2];
    double d0 = ab1*ac2 - ab2*ac1;
    double d1 = ab2*ac0 - ab0*ac2;
    double d2 = ab0*ac1 - ab1*ac0;
    return std::sqrt(d0*d0 + d1*d1 + d2*d2) / 2;
#endif
}

// compute_volume(ConstArrayIndirectAccessor), geometry.cxx:123-135
inline double elem_volume(const double d[NPE][ND])
{
#if DES_NDIMS == 3
    return tetrahedron_volume(d[0], d[1], d[2], d[3]);
#else
    return triangle_area(d[0], d[1], d[2]);
#endif
}

// dist2, geometry.cxx:16-25
inline double dist2(const double *a, const double *b)
{
    double sum = 0;
    for (int i = 0; i < ND; ++i) {
        double d = b[i] - a[i];
        sum += d * d;
    }
    return sum;
}

// geometry.cxx:59-73
double triangle_area2d(const double *a, const double *b, const double *c)
{
    double ab0 = b[0] - a[0], ab1 = b[1] - a[1];
    double ac0 = c[0] - a[0], ac1 = c[1] - a[1];
    return std::fabs(ab0*ac1 - ab1*ac0) / 2;
}

void elem_coords(const des_oracle &o, int e, double d[NPE][ND])
{
    for (int i = 0; i < NPE; ++i)
        node_xyz(o.coord, o.nn, o.conn[i * o.ne + e], d[i]);
}

// fields.cxx:11-38 (THREED), 40-53 (2-D: shpdy is left untouched)
void get_local_shape_fn(const des_oracle &o, int e, double shpdx[NPE], double shpdy[NPE], double shpdz[NPE])
{
    double d[NPE][ND];
    elem_coords(o, e, d);
#if DES_NDIMS == 2
    (void)shpdy;
    double iv = 1.0 / (2.0 * o.volume[e]);

    shpdx[0] = iv * (d[1][1] - d[2][1]);
    shpdx[1] = iv * (d[2][1] - d[0][1]);
    shpdx[2] = iv * (d[0][1] - d[1][1]);

    shpdz[0] = iv * (d[2][0] - d[1][0]);
    shpdz[1] = iv * (d[0][0] - d[2][0]);
    shpdz[2] = iv * (d[1][0] - d[0][0]);
#else
    double iv = 1.0 / (6.0 * o.volume[e]);

    double x01 = d[0][0] - d[1][0]; double x02 = d[0][0] - d[2][0]; double x03 = d[0][0] - d[3][0];
    double x12 = d[1][0] - d[2][0]; double x13 = d[1][0] - d[3][0]; double x23 = d[2][0] - d[3][0];
    double y01 = d[0][1] - d[1][1]; double y02 = d[0][1] - d[2][1]; double y03 = d[0][1] - d[3][1];
    double y12 = d[1][1] - d[2][1]; double y13 = d[1][1] - d[3][1]; double y23 = d[2][1] - d[3][1];
    double z01 = d[0][2] - d[1][2]; double z02 = d[0][2] - d[2][2]; double z03 = d[0][2] - d[3][2];
    double z12 = d[1][2] - d[2][2]; double z13 = d[1][2] - d[3][2]; double z23 = d[2][2] - d[3][2];

    shpdx[0] = iv * (y13*z12 - y12*z13);
    shpdx[1] = iv * (y02*z23 - y23*z02);
    shpdx[2] = iv * (y13*z03 - y03*z13);
    shpdx[3] = iv * (y01*z02 - y02*z01);

    shpdy[0] = iv * (z13*x12 - z12*x13);
    shpdy[1] = iv * (z02*x23 - z23*x02);
    shpdy[2] = iv * (z13*x03 - z03*x13);
    shpdy[3] = iv * (z01*x02 - z02*x01);

    shpdz[0] = iv * (x13*y12 - x12*y13);
    shpdz[1] = iv * (x02*y23 - x23*y02);
    shpdz[2] = iv * (x13*y03 - x03*y13);
    shpdz[3] = iv * (x01*y02 - x02*y01);
#endif
}

// geometry.cxx:170-201
void compute_volume(des_oracle &o, dvec &volume)
{
    #pragma omp parallel for
    for (int e = 0; e < o.ne; ++e) {
        double d[NPE][ND];
        elem_coords(o, e, d);
        volume[e] = elem_volume(d);
    }
}

// ---------------------------------------------------------------------------------
// field kernels
// ---------------------------------------------------------------------------------
// fields.cxx:197-278
void update_temperature(des_oracle &o)
{
    const int ne = o.ne, nn = o.nn;
    (void)nn;
    Mat mat(o);
    #pragma omp parallel for
    for (int e = 0; e < ne; e++) {
        double kv = mat.k(e) * o.volume[e];
        double rh = o.radiogenic[e] * o.volume[e] * mat.rho(e) / NPE;
        double shpdx[NPE], shpdy[NPE], shpdz[NPE];
        get_local_shape_fn(o, e, shpdx, shpdy, shpdz);
        for (int i = 0; i < NPE; ++i) {
            double diffusion = 0.;
            for (int j = 0; j < NPE; ++j)
#if DES_NDIMS == 3
                diffusion += (shpdx[i] * shpdx[j] + shpdy[i] * shpdy[j] + shpdz[i] * shpdz[j])
                             * o.temperature[o.conn[j * ne + e]];
#else
                diffusion += (shpdx[i] * shpdx[j] + shpdz[i] * shpdz[j])
                             * o.temperature[o.conn[j * ne + e]];
#endif
            o.tmp_result[i * ne + e] = diffusion * kv - rh;
        }
    }
    #pragma omp parallel for
    for (int n = o.c0; n < o.c1; n++) {
        if (o.bcflag[n] & BOUNDZ1)
            o.temperature[n] = o.p.surface_temperature;
        else {
            double tdot = 0;
            for (int k = o.sup_idx[n]; k < o.sup_idx[n+1]; ++k)
                tdot += o.tmp_result[o.sup_lidx[k] * ne + o.sup_arr[k]];
            o.temperature[n] -= o.dt * tdot / o.tmass[n];
        }
    }
}

// fields.cxx:405-480
void update_strain_rate(des_oracle &o)
{
    const int ne = o.ne, nn = o.nn;
    #pragma omp parallel for
    for (int e = 0; e < ne; ++e) {
        double shpdx[NPE], shpdy[NPE], shpdz[NPE];
        get_local_shape_fn(o, e, shpdx, shpdy, shpdz);
        double v[NPE][ND];
        for (int i = 0; i < NPE; ++i)
            node_xyz(o.vel, nn, o.conn[i * ne + e], v[i]);
        double s[NSTR];
#if DES_NDIMS == 3
        s[0] = 0; for (int i = 0; i < NPE; ++i) s[0] += v[i][0] * shpdx[i];
        s[1] = 0; for (int i = 0; i < NPE; ++i) s[1] += v[i][1] * shpdy[i];
        s[2] = 0; for (int i = 0; i < NPE; ++i) s[2] += v[i][2] * shpdz[i];
        s[3] = 0; for (int i = 0; i < NPE; ++i) s[3] += 0.5 * (v[i][0] * shpdy[i] + v[i][1] * shpdx[i]);
        s[4] = 0; for (int i = 0; i < NPE; ++i) s[4] += 0.5 * (v[i][0] * shpdz[i] + v[i][2] * shpdx[i]);
        s[5] = 0; for (int i = 0; i < NPE; ++i) s[5] += 0.5 * (v[i][1] * shpdz[i] + v[i][2] * shpdy[i]);
#else
        s[0] = 0; for (int i = 0; i < NPE; ++i) s[0] += v[i][0] * shpdx[i];
        s[1] = 0; for (int i = 0; i < NPE; ++i) s[1] += v[i][1] * shpdz[i];
        s[2] = 0; for (int i = 0; i < NPE; ++i) s[2] += 0.5 * (v[i][0] * shpdz[i] + v[i][1] * shpdx[i]);
#endif
        for (int i = 0; i < NSTR; ++i) o.strain_rate[i * ne + e] = s[i];
    }
}

// geometry.cxx:203-246
void compute_dvoldt(des_oracle &o)
{
    const int ne = o.ne;
    #pragma omp parallel for
    for (int e = 0; e < ne; e++) {
#if DES_NDIMS == 3
        double dj = o.strain_rate[e] + o.strain_rate[ne + e] + o.strain_rate[2 * ne + e];
#else
        double dj = o.strain_rate[e] + o.strain_rate[ne + e];      // trace(strain_rate)
#endif
        o.etmp[e] = dj * o.volume[e];
    }
    #pragma omp parallel for
    for (int n = o.c0; n < o.c1; n++) {
        double acc = 0.;
        for (int k = o.sup_idx[n]; k < o.sup_idx[n+1]; ++k)
            acc += o.etmp[o.sup_arr[k]];
        o.ntmp[n] = acc / o.volume_n[n];
    }
}

// geometry.cxx:249-279
void compute_edvoldt(des_oracle &o)
{
    const int ne = o.ne;
    #pragma omp parallel for
    for (int e = 0; e < ne; ++e) {
        double dj = 0;
        for (int i = 0; i < NPE; ++i)
            dj += o.ntmp[o.conn[i * ne + e]];
        o.edvoldt[e] = dj / NPE;
    }
}

// rheology.cxx:703-1030 (non-RSF, non-hydraulic branches)
void update_stress(des_oracle &o)
{
    const int ne = o.ne;
    const des_params &p = o.p;
    Mat mat(o);
    int n_past = 0;
    #pragma omp parallel for reduction(+:n_past)
    for (int e = 0; e < ne; e++) {
        int past = 0;
        double s[NSTR], es[NSTR], edot[NSTR];
        for (int i = 0; i < NSTR; ++i) {
            s[i] = o.stress[i * ne + e];
            es[i] = o.strain[i * ne + e];
            edot[i] = o.strain_rate[i * ne + e];
        }
        double old_s = trace3(s);

        // anti-mesh-locking correction, rheology.cxx:786-793
        {
            double div = trace3(edot);
            for (int i = 0; i < ND; ++i)
                edot[i] += (o.edvoldt[e] - div) / ND;
        }
        // strain_rate is written back here: visc() below reads the MODIFIED rate
        // through MatProps' reference to var.strain_rate (matprops.hpp:107)
        for (int i = 0; i < NSTR; ++i) o.strain_rate[i * ne + e] = edot[i];

        for (int i = 0; i < NSTR; ++i) es[i] += edot[i] * o.dt;
        double de[NSTR];
        for (int i = 0; i < NSTR; ++i) de[i] = edot[i] * o.dt;

        o.delta_plstrain[e] = 0.;

        switch (p.rheol_type) {
        case DES_RH_ELASTIC:
            elastic(mat.bulkm(e), mat.shearm(e), de, s);
            break;
        case DES_RH_VISCOUS: {
            // visc() reads var.stress[e] (still the old stress) and the modified rate
            o.viscosity[e] = mat.visc(e);
            double total_dv = trace3(es);
            viscous(mat.bulkm(e), o.viscosity[e], total_dv, edot, s);
            break;
        }
        case DES_RH_MAXWELL: {
            o.viscosity[e] = mat.visc(e);
            double dv = o.volume[e] / o.volume_old[e] - 1;
            maxwell(mat.bulkm(e), mat.shearm(e), o.viscosity[e], o.dt, dv, de, s);
            break;
        }
        case DES_RH_EP: {
            double depls = 0;
            double amc, anphi, anpsi, hardn, ten_max;
            mat.plastic_props(e, o.plstrain[e], amc, anphi, anpsi, hardn, ten_max);
            int failure_mode;
#if DES_NDIMS == 2
            if (p.is_plane_strain)
                elasto_plastic2d(mat.bulkm(e), mat.shearm(e), amc, anphi, anpsi, hardn, ten_max,
                                 de, depls, s, o.stressyy[e], failure_mode);
            else
#endif
            elasto_plastic(mat.bulkm(e), mat.shearm(e), amc, anphi, anpsi, hardn, ten_max,
                           de, depls, s, failure_mode, &past);
            n_past += past;
            o.plstrain[e] += depls;
            o.delta_plstrain[e] = depls;
            break;
        }
        case DES_RH_EVP: {
            double depls = 0;
            double bulkm = mat.bulkm(e), shearm = mat.shearm(e);
            o.viscosity[e] = mat.visc(e);
            double dv = o.volume[e] / o.volume_old[e] - 1;
            double sv[NSTR];
            for (int i = 0; i < NSTR; ++i) sv[i] = s[i];
            maxwell(bulkm, shearm, o.viscosity[e], o.dt, dv, de, sv);
            double svII = second_invariant2(sv);

            double amc, anphi, anpsi, hardn, ten_max;
            mat.plastic_props(e, o.plstrain[e], amc, anphi, anpsi, hardn, ten_max);
            double sp[NSTR];
            for (int i = 0; i < NSTR; ++i) sp[i] = s[i];
            int failure_mode;
#if DES_NDIMS == 2
            double spyy = 0;
            if (p.is_plane_strain) {
                spyy = o.stressyy[e];
                elasto_plastic2d(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max,
                                 de, depls, sp, spyy, failure_mode);
            } else
#endif
            elasto_plastic(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max,
                           de, depls, sp, failure_mode, &past);
            n_past += past;
            double spII = second_invariant2(sp);

            if (svII < spII) {
                for (int i = 0; i < NSTR; ++i) s[i] = sv[i];
            } else {
                for (int i = 0; i < NSTR; ++i) s[i] = sp[i];
                o.plstrain[e] += depls;
                o.delta_plstrain[e] = depls;
#if DES_NDIMS == 2
                if (p.is_plane_strain) o.stressyy[e] = spyy;
#endif
            }
            break;
        }
        default:
            break;
        }
        if (p.is_using_mixed_stress)
            o.dpressure[e] = trace3(s) - old_s;

        for (int i = 0; i < NSTR; ++i) {
            o.stress[i * ne + e] = s[i];
            o.strain[i * ne + e] = es[i];
        }
    }
    o.n_return_mapping = n_past;
}

// geometry.cxx:282-336, element part + nodal gather
void NMD_stress_gather(des_oracle &o)
{
    const int ne = o.ne;
    #pragma omp parallel for
    for (int e = 0; e < ne; e++)
        o.etmp[e] = o.dpressure[e] * o.volume[e];
    #pragma omp parallel for
    for (int n = o.c0; n < o.c1; n++) {
        double acc = 0;
        for (int k = o.sup_idx[n]; k < o.sup_idx[n+1]; ++k)
            acc += o.etmp[o.sup_arr[k]];
        o.ntmp[n] = acc / o.volume_n[n];
    }
}

// geometry.cxx:311-331, after the nodal pressure change is known on every local node
void NMD_stress_apply(des_oracle &o)
{
    const int ne = o.ne;
    #pragma omp parallel for
    for (int e = 0; e < ne; ++e) {
        double dp = 0;
        for (int i = 0; i < NPE; ++i)
            dp += o.ntmp[o.conn[i * ne + e]];
        double dp_el = dp / NPE;
        double dp_orig = o.dpressure[e];
        double ddp = (-dp_orig + dp_el) / ND;
        for (int i = 0; i < ND; ++i)
            o.stress[i * ne + e] += ddp;
    }
}

// bc.cxx:24-54
void normal_vector_of_facet(const double fc[NPF][ND], double *normal, double &zcenter)
{
#if DES_NDIMS == 2
    double v01[ND];
    for (int i = 0; i < ND; ++i)
        v01[i] = fc[1][i] - fc[0][i];
    normal[0] = v01[1];
    normal[1] = -v01[0];
    zcenter = (fc[0][1] + fc[1][1]) / NPF;
#else
    double v01[3], v02[3];
    for (int i = 0; i < ND; ++i) {
        v01[i] = fc[1][i] - fc[0][i];
        v02[i] = fc[2][i] - fc[0][i];
    }
    normal[0] = (v01[1] * v02[2] - v01[2] * v02[1]) / 2;
    normal[1] = (v01[2] * v02[0] - v01[0] * v02[2]) / 2;
    normal[2] = (v01[0] * v02[1] - v01[1] * v02[0]) / 2;
    zcenter = (fc[0][2] + fc[1][2] + fc[2][2]) / NPF;
#endif
}

// bc.cxx:661-827
void apply_stress_bcs(des_oracle &o)
{
    const des_params &p = o.p;
    const int ne = o.ne, nn = o.nn;
    if (p.gravity == 0) return;
    Mat mat(o);

    for (int e = 0; e < ne; ++e) o.etmp_int[e] = -1;

    for (int i = 0; i < DES_NBDRY; i++) {
        if (p.vbc_types[i] != 0 && p.vbc_types[i] != 2 && p.vbc_types[i] != 4) continue;
        if (i == iboundz0 && !p.has_winkler_foundation) continue;
        if (i == iboundz1 && !p.has_water_loading) continue;

        const int bound = (int)o.bf_elem[i].size();
        const int nbdry_nodes = (int)o.bnodes[i].size();

        for (int n = 0; n < bound; ++n) {
            int e = o.bf_elem[i][n];
            int f = o.bf_facet[i][n];
            double normal[ND], zcenter;
            double fc[NPF][ND];
            for (int j = 0; j < NPF; ++j)
                node_xyz(o.coord, nn, o.conn[NODE_OF_FACET[f][j] * ne + e], fc[j]);
            normal_vector_of_facet(fc, normal, zcenter);

            double pr;
            if (i == iboundz0 && p.has_winkler_foundation) {
                double rho_effective = mat.rho(e);
                pr = p.compensation_pressure -
                     (rho_effective + p.winkler_delta_rho) * p.gravity * (zcenter + p.zlength);
            } else if (i == iboundz1 && p.has_water_loading) {
                pr = 0;
                if (zcenter < p.surf_base_level)
                    pr = p.sea_water_density * p.gravity * (p.surf_base_level - zcenter);
            } else {
                pr = ref_pressure(p, zcenter);
                if (pr < 0.0) pr = 0.0;
            }

            o.etmp_int[e] = n;
            for (int j = 0; j < NPF; ++j)
                for (int d = 0; d < ND; ++d)
                    o.tmp_result[(j*ND + d) * ne + n] = pr * normal[d] / NPF;
        }

        for (int j = 0; j < nbdry_nodes; ++j) {
            const int n = o.bnodes[i][j];
            if (n < o.c0 || n >= o.c1) continue;
            for (int k = o.sup_idx[n]; k < o.sup_idx[n+1]; ++k) {
                int e = o.sup_arr[k];
                int ibound = o.etmp_int[e];
                if (ibound < 0) continue;
                int f = o.bf_facet[i][ibound];
                for (int l = 0; l < NPF; ++l) {
                    if (n == o.conn[NODE_OF_FACET[f][l] * ne + e]) {
                        for (int d = 0; d < ND; ++d)
                            o.force[d * nn + n] -= o.tmp_result[(l*ND + d) * ne + ibound];
                        break;
                    }
                }
            }
        }

        for (int n = 0; n < bound; ++n)
            o.etmp_int[o.bf_elem[i][n]] = -1;
    }

    if (p.has_elastic_foundation) {
        for (size_t j = 0; j < o.bnodes[iboundz0].size(); ++j) {
            int n = o.bnodes[iboundz0][j];
            o.force[(ND-1) * nn + n] -= p.elastic_foundation_constant
                                   * (o.coord[(ND-1) * nn + n] - o.coord0[(ND-1) * nn + n]);
        }
    }
}

// bc.cxx:829-912
void apply_stress_bcs_neumann(des_oracle &o)
{
    const des_params &p = o.p;
    const int ne = o.ne, nn = o.nn;
    for (int i = 0; i < 6; ++i) {
        if (p.stress_bc_types[i] == 0) continue;
        const int bound = (int)o.bf_elem[i].size();
        for (int n = 0; n < bound; ++n) {
            int e = o.bf_elem[i][n];
            int f = o.bf_facet[i][n];
            double normal[ND] = {0}, zcenter = 0;
            double fc[NPF][ND];
            for (int j = 0; j < NPF; ++j)
                node_xyz(o.coord, nn, o.conn[NODE_OF_FACET[f][j] * ne + e], fc[j]);
            normal_vector_of_facet(fc, normal, zcenter);
            double traction[ND] = {0};
            switch (p.stress_bc_types[i]) {
#if DES_NDIMS == 3
            case 1: traction[0] = p.stress_bc_values[i]; break;
            case 2: traction[1] = p.stress_bc_values[i]; break;
            case 3: traction[2] = p.stress_bc_values[i]; break;
#else
            case 1: traction[0] = p.stress_bc_values[i]; break;
            case 3: traction[1] = p.stress_bc_values[i]; break;
#endif
            default: continue;
            }
            for (int j = 0; j < NPF; ++j) {
                int node = o.conn[NODE_OF_FACET[f][j] * ne + e];
                for (int d = 0; d < ND; ++d)
                    o.force[d * nn + node] += traction[d] * normal[d] / NPF;
            }
        }
    }
}

// fields.cxx:483-579
void apply_damping(des_oracle &o)
{
    const des_params &p = o.p;
    const int nn = o.nn;
    const double small_vel = 1e-13;
    switch (p.damping_option) {
    case 0: break;
    case 1:
        #pragma omp parallel for
        for (int i = o.c0; i < o.c1; ++i)
            for (int j = 0; j < ND; j++)
                if (std::fabs(o.vel[j*nn+i]) > small_vel)
                    o.force[j*nn+i] -= p.damping_factor * std::copysign(o.force[j*nn+i], o.vel[j*nn+i]);
        break;
    case 2:
        #pragma omp parallel for
        for (int i = o.c0; i < o.c1; ++i)
            for (int j = 0; j < ND; j++)
                o.force[j*nn+i] -= p.damping_factor * o.force[j*nn+i];
        break;
    case 3:
        #pragma omp parallel for
        for (int i = o.c0; i < o.c1; ++i)
            for (int j = 0; j < ND; j++) {
                if ((o.force[j*nn+i] < 0) == (o.vel[j*nn+i] < 0)) {
                    // fields.cxx:538 -- comma operator: the trailing vel term has no effect
                    o.force[j*nn+i] -= p.damping_factor * o.force[j*nn+i];
                } else {
                    o.force[j*nn+i] += (1 - p.damping_factor) * o.force[j*nn+i];
                }
            }
        break;
    case 4:
        #pragma omp parallel for
        for (int i = o.c0; i < o.c1; ++i) {
            double critical_coeff = 2.0 * std::sqrt(o.mass[i] * o.ymass[i]);
            for (int j = 0; j < ND; j++)
                if (std::fabs(o.vel[j*nn+i]) > small_vel) {
                    double f_C = p.damping_factor * std::copysign(o.force[j*nn+i], o.vel[j*nn+i]);
                    double f_V = critical_coeff * o.vel[j*nn+i];
                    double f_damping = (std::fabs(f_C) < std::fabs(f_V)) ? f_V : f_C;
                    o.force[j*nn+i] -= f_damping;
                }
        }
        break;
    default: break;
    }
}

// fields.cxx:609-698
void update_force(des_oracle &o)
{
    const des_params &p = o.p;
    const int ne = o.ne, nn = o.nn;
    Mat mat(o);
    #pragma omp parallel for
    for (int e = 0; e < ne; e++) {
        double shpdx[NPE], shpdy[NPE], shpdz[NPE];
        get_local_shape_fn(o, e, shpdx, shpdy, shpdz);
        double s[NSTR];
        for (int i = 0; i < NSTR; ++i) s[i] = o.stress[i * ne + e];
        double vol = o.volume[e];
        double buoy = 0;
        if (p.gravity != 0)
            buoy = (mat.rho(e) * (1 - mat.phi(e)) + 1000.0 * mat.phi(e)) * p.gravity / NPE;
        for (int i = 0; i < NPE; ++i) {
#if DES_NDIMS == 3
            o.tmp_result[i * ne + e] = (s[0]*shpdx[i] + s[3]*shpdy[i] + s[4]*shpdz[i]) * vol;
            o.tmp_result[(i + NPE) * ne + e] = (s[3]*shpdx[i] + s[1]*shpdy[i] + s[5]*shpdz[i]) * vol;
            o.tmp_result[(i + NPE*2) * ne + e] = (s[4]*shpdx[i] + s[5]*shpdy[i] + s[2]*shpdz[i] + buoy) * vol;
#else
            o.tmp_result[i * ne + e] = (s[0]*shpdx[i] + s[2]*shpdz[i]) * vol;
            o.tmp_result[(i + NPE) * ne + e] = (s[2]*shpdx[i] + s[1]*shpdz[i] + buoy) * vol;
#endif
        }
    }
    #pragma omp parallel for
    for (int n = o.c0; n < o.c1; n++) {
        double f[ND] = {0}, fr[ND] = {0};
        for (int k = o.sup_idx[n]; k < o.sup_idx[n+1]; ++k) {
            const int e = o.sup_arr[k], i = o.sup_lidx[k];
            for (int j = 0; j < ND; j++) {
                f[j] -= o.tmp_result[(i + NPE*j) * ne + e];
                fr[j] = o.tmp_result[(i + NPE*j) * ne + e];   // assignment: fields.cxx:673
            }
        }
        for (int j = 0; j < ND; j++) {
            o.force[j*nn+n] = f[j];
            o.force_residual[j*nn+n] = fr[j];
        }
    }
    apply_stress_bcs(o);
    if (!o.body_force_adjustment) apply_stress_bcs_neumann(o);     // fields.cxx:690
    apply_damping(o);
}

// fields.cxx:700-722.  The reference sums with an OpenMP reduction, i.e. in no particular order; where a decision hangs on the
// value (the pseudo-transient loop) the sum is formed in ONE association whatever the partition -- des_params.h, DES_RES_BLOCK:
// per block of B consecutive GLOBAL node ids the nodes' terms one after the other, then the blocks in a fixed shape.
// residual_blocks_local: the partials of the blocks this rank owns, into their places of the global array.
int residual_nblocks(const des_oracle &o) { const int B = des_res_block(o.nn_global); return (o.nn_global + B - 1) / B; }

void residual_blocks_local(des_oracle &o)
{
    const int B = des_res_block(o.nn_global);
    const double num = (double)o.nn_global * ND;
    o.res_blocks.assign((size_t)residual_nblocks(o), 0.0);
    for (int i = o.o0; i < o.o1; ++i) {
        const int g = o.g0 + (i - o.o0);           // local numbering is ascending global order
        double t = std::pow(o.force_residual[i], 2) / num;
        for (int j = 1; j < ND; ++j) t += std::pow(o.force_residual[j*o.nn+i], 2) / num;
        o.res_blocks[g / B] += t;
    }
}

// the fixed shape over the global block array: 256 strided serial sums, then a pairwise tree (what the device's
// k_residual_final does with its 256 lanes)
double residual_final(const double *blocks, int nb)
{
    double s[256];
    for (int j = 0; j < 256; ++j) {
        double t = 0;
        for (int i = j; i < nb; i += 256) t += blocks[i];
        s[j] = t;
    }
    for (int off = 128; off > 0; off >>= 1)
        for (int j = 0; j < off; ++j) s[j] += s[j + off];
    return s[0];
}

double calculate_residual_force(des_oracle &o)
{
    residual_blocks_local(o);
    if (o.o0 == 0 && o.o1 == o.nn && o.nn == o.nn_global) {
        const double l2 = residual_final(o.res_blocks.data(), (int)o.res_blocks.size());
        o.l2_part = l2;
        return std::sqrt(l2);
    }
    // a rank of a decomposed run: its own nodes only (summed over ranks before the root; the pseudo-transient loop's
    // driver takes the block partials instead: des_oracle_residual_blocks / des_oracle_residual_set)
    double l2 = 0.0;
    double num = (double)o.nn_global * ND;
    for (int i = o.o0; i < o.o1; ++i)
        for (int j = 0; j < ND; ++j)
            l2 += std::pow(o.force_residual[j*o.nn+i], 2) / num;
    o.l2_part = l2;
    return std::sqrt(l2);
}

// fields.cxx:725-742
void update_velocity(des_oracle &o)
{
    const int nn = o.nn;
    #pragma omp parallel for
    for (int i = o.c0; i < o.c1; ++i)
        for (int j = 0; j < ND; j++)
            o.vel[j*nn+i] += o.dt * o.force[j*nn+i] / o.mass[i];
}

#if DES_NDIMS == 2
// utils.hpp:259-287: findNearestNeighbourIndex + interp1 over n ascending abscissae
double interp1(const double *x, const double *y, int n, double x_new)
{
    double dist = DBL_MAX;
    int idx = -1;
    for (int i = 0; i < n; ++i) {
        double newDist = x_new - x[i];
        if (newDist >= 0 && newDist <= dist) {
            dist = newDist;
            idx = i;
        }
    }
    double slope = 0;
    if (idx < 0)
        idx = 0;
    else if (idx < n - 1)
        slope = (y[idx+1] - y[idx]) / (x[idx+1] - x[idx]);
    return slope * (x_new - x[idx]) + y[idx];
}

// bc.cxx:227-659, the !THREED branches: time-dependent x boundary values (:247-249), their
// variation with depth along the side walls (:251-300, 427-429), the sheared bottom zone of
// vbc_x0 = 3 (:446-451)
void apply_vbcs(des_oracle &o, bool all_local_nodes = false)
{
    const des_params &p = o.p;
    const int nn = o.nn;
    (void)all_local_nodes;
    double t_now = o.time / YEAR2SEC;
    double vbc_applied_x0 = p.vbc_values[0] * interp1(p.vbc_period_x0_time_in_yr, p.vbc_period_x0_ratio, p.num_vbc_period_x0, t_now);
    double vbc_applied_x1 = p.vbc_values[1] * interp1(p.vbc_period_x1_time_in_yr, p.vbc_period_x1_ratio, p.num_vbc_period_x1, t_now);

    double BOUNDX0_max = 0., BOUNDX0_min = 0., BOUNDX1_max = 0., BOUNDX1_min = 0.;
    bool if_init0 = false, if_init1 = false;
    for (int i = 0; i < nn; ++i) {
        unsigned flag = o.bcflag[i];
        if (!(flag & BOUND_ANY)) continue;
        double z = o.coord[nn + i];
        if (flag & 1u) {
            if (!if_init0) { BOUNDX0_max = z; BOUNDX0_min = z; if_init0 = true; }
            else { if (z > BOUNDX0_max) BOUNDX0_max = z; if (z < BOUNDX0_min) BOUNDX0_min = z; }
        } else if (flag & 2u) {
            if (!if_init1) { BOUNDX1_max = z; BOUNDX1_min = z; if_init1 = true; }
            else { if (z > BOUNDX1_max) BOUNDX1_max = z; if (z < BOUNDX1_min) BOUNDX1_min = z; }
        }
    }
    if (o.wall_given) {                              // a rank of a decomposed mesh: the cross-rank maxima
        const bool any = o.wall[0] != -DBL_MAX;
        BOUNDX0_max = any ? o.wall[0] : 0.;
        BOUNDX0_min = any ? -o.wall[1] : 0.;
    }
    double BOUNDX0_width = BOUNDX0_max - BOUNDX0_min;
    (void)BOUNDX1_max; (void)BOUNDX1_min;
    double div_x0[4], div_x1[4];
    for (int i = 0; i < 4; i++) {
        div_x0[i] = - (BOUNDX0_max - p.vbc_vertical_div_x0[i] * BOUNDX0_width);
        div_x1[i] = - (BOUNDX0_max - p.vbc_vertical_div_x1[i] * BOUNDX0_width);   // x0's extent: bc.cxx:299
    }

    int bc_x0 = p.vbc_types[0], bc_x1 = p.vbc_types[1];
    int bc_z0 = p.vbc_types[4], bc_z1 = p.vbc_types[5];
    double bc_vx0 = p.vbc_values[0], bc_vx1 = p.vbc_values[1];
    double bc_vz0 = p.vbc_values[4], bc_vz1 = p.vbc_values[5];
    const double bc_vx0_l = p.vbc_val_l[0], bc_vx1_l = p.vbc_val_l[1];
    if (o.pt_jump) {
        bc_vx0 = 0.0; bc_vx1 = 0.0; bc_vz0 = 0.0; bc_vz1 = 0.0;
        vbc_applied_x0 = 0.0; vbc_applied_x1 = 0.0;
    }
    if (o.time > p.vbc_val_z1_loading_period) bc_z1 = 0;

    double zmin = 0;
    for (int k = 0; k < nn; ++k)
        if (o.coord[nn + k] < zmin) zmin = o.coord[nn + k];
    if (o.wall_given) zmin = -o.wall[2];

    #pragma omp parallel for
    for (int i = 0; i < nn; ++i) {
        unsigned flag = o.bcflag[i];
        if (!(flag & BOUND_ANY)) continue;
        double v[2] = {o.vel[i], o.vel[nn+i]};
        const double x1 = o.coord[nn + i];
        double vbc_exact_x0 = vbc_applied_x0 * interp1(div_x0, p.vbc_vertical_ratio_x0, 4, -x1);
        double vbc_exact_x1 = vbc_applied_x1 * interp1(div_x1, p.vbc_vertical_ratio_x1, 4, -x1);

        if (flag & 1u) {
            switch (bc_x0) {
            case 0: break;
            case 1: v[0] = vbc_exact_x0; break;
            case 2: v[1] = 0; break;
            case 3:
                v[0] = vbc_exact_x0;
                if (p.bottom_shear_zone_thickness > 0.) {
                    double dz = x1 - zmin;
                    if (dz < p.bottom_shear_zone_thickness)
                        v[0] = v[0] * dz / p.bottom_shear_zone_thickness;
                }
                v[1] = 0;
                break;
            case 4: v[0] = 0; v[1] = bc_vx0; break;
            case 6: v[0] = vbc_exact_x0; v[1] = bc_vx0_l; break;
            }
        }
        if (flag & 2u) {
            switch (bc_x1) {
            case 0: break;
            case 1: v[0] = vbc_exact_x1; break;
            case 2: v[1] = 0; break;
            case 3: v[0] = vbc_exact_x1; v[1] = 0; break;
            case 4: v[0] = 0; v[1] = bc_vx1; break;
            case 6: v[0] = vbc_exact_x1; v[1] = bc_vx1_l; break;
            }
        }

        // slanted boundaries n0..n3, bc.cxx:491-585
        for (int ib = iboundn0; ib <= iboundn3; ib++) {
            if (!(flag & (1u << ib))) continue;
            const double n[2] = {o.bnormals[ib], o.bnormals[DES_NBDRY + ib]};
            double fac = 0;
            switch (p.vbc_types[ib]) {
            case 1:
            case 11: {
                const int nd = (p.vbc_types[ib] == 1) ? ND : ND-1;
                double target = p.vbc_values[ib];
                if (p.vbc_types[ib] == 11) {
                    fac = 1 / std::sqrt(1 - n[ND-1]*n[ND-1]);
                    target = p.vbc_values[ib] * fac;
                }
                if (flag == (1u << ib)) {
                    double vn = 0;
                    for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                    for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                } else {
                    for (int ic = iboundx0; ic < ib; ic++) {
                        if (!(flag & (1u << ic))) continue;
                        if (p.vbc_types[ic] == 0) {
                            double vn = 0;
                            for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                            for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                        } else if (p.vbc_types[ic] == 1) {
                            const int slot = o.edge_slot[ic*DES_NBDRY + ib];
                            if (slot < 0) continue;
                            const double *edge = &o.edge_vec[slot*ND];
                            double ve = 0;
                            for (int d = 0; d < ND; d++) ve += v[d] * edge[d];
                            for (int d = 0; d < ND; d++) v[d] = ve * edge[d];
                        }
                    }
                }
                break;
            }
            case 3:
                for (int d = 0; d < ND; d++) v[d] = p.vbc_values[ib] * n[d];
                break;
            case 13:
                fac = 1 / std::sqrt(1 - n[ND-1]*n[ND-1]);
                for (int d = 0; d < ND-1; d++) v[d] = p.vbc_values[ib] * fac * n[d];
                v[ND-1] = 0;
                break;
            }
        }

        // Z last, bc.cxx:587-650
        if (!(bc_z0 == 0 && bc_z1 == 0)) {
            if (flag & BOUNDZ0) {
                switch (bc_z0) {
                case 0: break;
                case 1: v[1] = bc_vz0; break;
                case 2: v[0] = 0; break;
                case 3: v[0] = 0; v[1] = bc_vz0; break;
                case 4: v[0] = bc_vz0; v[1] = 0; break;
                }
            }
            if (flag & BOUNDZ1) {
                switch (bc_z1) {
                case 0: break;
                case 1: v[1] = bc_vz1; break;
                case 2: v[0] = 0; break;
                case 3: v[0] = 0.0; v[1] = bc_vz1; break;
                case 4: v[0] = bc_vz1; v[1] = 0; break;
                }
            }
        }
        o.vel[i] = v[0]; o.vel[nn+i] = v[1];
    }
}
#else
// bc.cxx:227-659 (THREED branch)
void apply_vbcs(des_oracle &o, bool all_local_nodes = false)
{
    const des_params &p = o.p;
    const int nn = o.nn;
    // init() applies the bcs to every local node (a purely local operation that gives halo
    // nodes the owner's values); inside a step only owned nodes are touched
    const int vb0 = all_local_nodes ? 0 : o.c0, vb1 = all_local_nodes ? nn : o.c1;
    int bc_z0 = p.vbc_types[4], bc_z1 = p.vbc_types[5];
    // PT_jump: the boundaries are held at rest inside the pseudo-transient loop (bc.cxx:330-343)
    const double bc_vz0 = o.pt_jump ? 0.0 : p.vbc_values[4], bc_vz1 = o.pt_jump ? 0.0 : p.vbc_values[5];
    if (o.time > p.vbc_val_z1_loading_period) bc_z1 = 0;

    struct LateralFace { unsigned mask; int ni; int li; int type; double val; double val_l; };
    const LateralFace lateral_faces[] = {
        {1u << 0, 0, 1, p.vbc_types[0], o.pt_jump ? 0.0 : p.vbc_values[0], p.vbc_val_l[0]},
        {1u << 1, 0, 1, p.vbc_types[1], o.pt_jump ? 0.0 : p.vbc_values[1], p.vbc_val_l[1]},
        {1u << 2, 1, 0, p.vbc_types[2], o.pt_jump ? 0.0 : p.vbc_values[2], p.vbc_val_l[2]},
        {1u << 3, 1, 0, p.vbc_types[3], o.pt_jump ? 0.0 : p.vbc_values[3], p.vbc_val_l[3]},
    };

    #pragma omp parallel for
    for (int i = vb0; i < vb1; ++i) {
        unsigned flag = o.bcflag[i];
        if (!(flag & BOUND_ANY)) continue;
        double v[3] = {o.vel[i], o.vel[nn+i], o.vel[2*nn+i]};

        for (int lf = 0; lf < 4; ++lf) {
            const LateralFace &f = lateral_faces[lf];
            if (!(flag & f.mask)) continue;
            switch (f.type) {
            case 0: break;
            case 1: v[f.ni] = f.val; break;
            case 2: v[f.li] = 0; v[2] = 0; break;
            case 3: v[f.ni] = f.val; v[f.li] = 0; v[2] = 0; break;
            case 4: v[f.li] = f.val; v[2] = 0; break;
            case 5: v[f.ni] = 0; v[f.li] = f.val; v[2] = 0; break;
            case 6: v[f.ni] = f.val; v[f.li] = f.val_l; break;
            case 7: v[f.ni] = f.val; v[f.li] = 0; break;
            }
        }

        // slanted boundaries n0..n3, bc.cxx:491-585
        for (int ib = iboundn0; ib <= iboundn3; ib++) {
            if (!(flag & (1u << ib))) continue;
            const double n[3] = {o.bnormals[ib], o.bnormals[DES_NBDRY + ib], o.bnormals[2*DES_NBDRY + ib]};
            double fac = 0;
            switch (p.vbc_types[ib]) {
            case 1:
            case 11: {
                const int nd = (p.vbc_types[ib] == 1) ? ND : ND-1;
                double target = p.vbc_values[ib];
                if (p.vbc_types[ib] == 11) {
                    fac = 1 / std::sqrt(1 - n[ND-1]*n[ND-1]);
                    target = p.vbc_values[ib] * fac;
                }
                if (flag == (1u << ib)) {
                    double vn = 0;
                    for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                    for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                } else {
                    for (int ic = iboundx0; ic < ib; ic++) {
                        if (!(flag & (1u << ic))) continue;
                        if (p.vbc_types[ic] == 0) {
                            double vn = 0;
                            for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                            for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                        } else if (p.vbc_types[ic] == 1) {
                            const int slot = o.edge_slot[ic*DES_NBDRY + ib];
                            if (slot < 0) continue;
                            const double *edge = &o.edge_vec[slot*ND];
                            double ve = 0;
                            for (int d = 0; d < ND; d++) ve += v[d] * edge[d];
                            for (int d = 0; d < ND; d++) v[d] = ve * edge[d];
                        }
                    }
                }
                break;
            }
            case 3:
                for (int d = 0; d < ND; d++) v[d] = p.vbc_values[ib] * n[d];
                break;
            case 13:
                fac = 1 / std::sqrt(1 - n[ND-1]*n[ND-1]);
                for (int d = 0; d < ND-1; d++) v[d] = p.vbc_values[ib] * fac * n[d];
                v[ND-1] = 0;
                break;
            }
        }

        // Z last, bc.cxx:587-650
        if (!(bc_z0 == 0 && bc_z1 == 0)) {
            if (flag & BOUNDZ0) {
                switch (bc_z0) {
                case 0: break;
                case 1: v[2] = bc_vz0; break;
                case 2: v[0] = 0; v[1] = 0; break;
                case 3: v[0] = 0; v[1] = 0; v[2] = bc_vz0; break;
                }
            }
            if (flag & BOUNDZ1) {
                switch (bc_z1) {
                case 0: break;
                case 1: v[2] = bc_vz1; break;
                case 2: v[0] = 0; v[1] = 0; break;
                case 3: v[0] = 0.0; v[1] = 0; v[2] = bc_vz1; break;
                case 4: v[0] = bc_vz1; v[1] = 0; v[2] = 0; break;
                }
            }
        }
        o.vel[i] = v[0]; o.vel[nn+i] = v[1]; o.vel[2*nn+i] = v[2];
    }
}
#endif

// fields.cxx:761-784
void update_coordinate(des_oracle &o)
{
    const int nn = o.nn;
    #pragma omp parallel for
    for (int i = o.c0; i < o.c1; ++i)
        for (int j = 0; j < ND; ++j)
            o.coord[j*nn+i] += o.vel[j*nn+i] * o.dt;
}

#if DES_NDIMS == 2
// bc.cxx:916-1112, the !THREED branches: 1-D diffusion along the sorted top nodes (:1021-1033,
// 1067-1077) with different rates above / below the base level (:1098-1106)
void simple_diffusion(des_oracle &o)
{
    const int ne = o.ne, nn = o.nn;
    for (int i = 0; i < nn; i++) { o.total_dx[i] = 0.; o.total_slope[i] = 0.; }
    for (int i = 0; i < o.etop; ++i) {
        int n0 = o.top_nodes[i];
        int n1 = o.top_nodes[i+1];
        double dx = std::fabs(o.coord[n1] - o.coord[n0]);
        o.etmp[i] = dx;
        o.tmp_result[0 * ne + i] = -(o.coord[nn + n1] - o.coord[nn + n0]) / dx;
        o.tmp_result[1 * ne + i] = (o.coord[nn + n1] - o.coord[nn + n0]) / dx;
    }
    const int ntop = o.ntop;
    for (int i = 0; i < ntop; ++i) {
        int n = o.top_nodes[i];
        if (i == 0) {
            o.total_dx[n] = o.etmp[i];
            o.total_slope[n] = o.tmp_result[0 * ne + i];
        } else if (i == ntop-1) {
            o.total_dx[n] = o.etmp[i-1];
            o.total_slope[n] = o.tmp_result[1 * ne + i-1];
        } else {
            o.total_dx[n] = o.etmp[i-1] + o.etmp[i];
            o.total_slope[n] = o.tmp_result[1 * ne + i-1] + o.tmp_result[0 * ne + i];
        }
    }
    for (int i = 0; i < ntop; ++i) {
        int n = o.top_nodes[i];
        double conv = o.p.surface_diffusivity * o.dt * o.total_slope[n] / o.total_dx[n];
        if (o.coord[nn + n] > o.p.surf_base_level && conv > 0.)
            o.dh[i] -= o.p.surf_diff_ratio_terrig * conv;
        else if (o.coord[nn + n] <= o.p.surf_base_level && conv < 0.)
            o.dh[i] -= o.p.surf_diff_ratio_marine * conv;
        else
            o.dh[i] -= conv;
    }
}
#else
// bc.cxx:916-1112 (THREED)
void simple_diffusion(des_oracle &o)
{
    const int ne = o.ne, nn = o.nn;
    const ivec &top_e = o.bf_elem[iboundz1], &top_f = o.bf_facet[iboundz1];
    for (int i = 0; i < nn; i++) { o.total_dx[i] = 0.; o.total_slope[i] = 0.; }

    for (int i = 0; i < o.etop; ++i) {
        int e = top_e[i], f = top_f[i];
        double cf[3][3];
        for (int j = 0; j < NPF; ++j)
            node_xyz(o.coord, nn, o.conn[NODE_OF_FACET[f][j] * ne + e], cf[j]);
        double x01 = cf[1][0] - cf[0][0], y01 = cf[1][1] - cf[0][1];
        double x02 = cf[2][0] - cf[0][0], y02 = cf[2][1] - cf[0][1];
        double normal2 = x01*y02 - y01*x02;
        double projected_area = 0.5 * normal2;
        o.etmp[i] = projected_area;

        double shp2dx[3], shp2dy[3];
        double iv = 1 / (2 * projected_area);
        shp2dx[0] = iv * (cf[1][1] - cf[2][1]);
        shp2dx[1] = iv * (cf[2][1] - cf[0][1]);
        shp2dx[2] = iv * (cf[0][1] - cf[1][1]);
        shp2dy[0] = iv * (cf[2][0] - cf[1][0]);
        shp2dy[1] = iv * (cf[0][0] - cf[2][0]);
        shp2dy[2] = iv * (cf[1][0] - cf[0][0]);

        double D[3][3];
        for (int j = 0; j < NPF; j++)
            for (int k = 0; k < NPF; k++)
                D[j][k] = (shp2dx[j] * shp2dx[k] + shp2dy[j] * shp2dy[k]);
        for (int j = 0; j < NPF; j++) {
            double slope = 0;
            for (int k = 0; k < NPF; k++)
                slope += D[j][k] * cf[k][2];
            o.tmp_result[j * ne + i] = slope * projected_area;
        }
    }

    for (int i = 0; i < o.ntop; ++i) {
        int n = o.top_nodes[i];
        if (n < o.c0 || n >= o.c1) continue;
        for (int j = o.ssup_idx[i]; j < o.ssup_idx[i+1]; ++j) {
            int k = o.ssup_arr[j];
            o.total_dx[n] += o.etmp[k];
            int e = top_e[k], f = top_f[k];
            for (int m = 0; m < NPF; ++m) {
                if (o.conn[NODE_OF_FACET[f][m] * ne + e] == n) {
                    o.total_slope[n] += o.tmp_result[m * ne + k];
                    break;
                }
            }
        }
    }

    for (int i = 0; i < o.ntop; ++i) {
        int n = o.top_nodes[i];
        if (n < o.c0 || n >= o.c1) continue;
        double conv = o.p.surface_diffusivity * o.dt * o.total_slope[n] / o.total_dx[n];
        o.dh[i] -= conv;
    }
}
#endif

// bc.cxx:1655-1707
void correct_surface_element(des_oracle &o)
{
    const int ne = o.ne;
    for (int i = 0; i < o.ntop_elems; i++) {
        const int e = o.top_elems[i];
        double d[NPE][ND];
        elem_coords(o, e, d);
        double new_volumes = elem_volume(d);
        double rdv = new_volumes / o.volume[e];
        o.volume[e] = new_volumes;
        if (rdv < 1.0) continue;
        o.plstrain[e] /= rdv;
        for (int j = 0; j < NSTR; j++) {
            o.stress[j*ne+e] /= rdv;
            o.strain[j*ne+e] /= rdv;
            o.strain_rate[j*ne+e] /= rdv;
        }
    }
    for (int n = 0; n < o.ntop; n++) {
        int nt = o.top_nodes[n];
        if (nt < o.c0 || nt >= o.c1) continue;
        double acc = 0.;
        for (int k = o.sup_idx[nt]; k < o.sup_idx[nt+1]; ++k)
            acc += o.volume[o.sup_arr[k]];
        o.volume_n[nt] = acc;
    }
}

// bc.cxx:1709-1872 (THREED; marker corrections are host-side and out of scope), first half:
// everything that produces nodal values other ranks need (dh, surface coordinates)
void surface_processes_a(des_oracle &o)
{
    const int nn = o.nn;
    for (int i = 0; i < o.ntop; i++) o.dh[i] = 0.;

    switch (o.p.surface_process_option) {
    case 0: break;
    case 1: simple_diffusion(o); break;
    default: break;
    }

    for (int i = 0; i < o.ntop; i++) {
        int nt = o.top_nodes[i];
        if (nt < o.c0 || nt >= o.c1) continue;
        o.coord[(ND-1)*nn + nt] += o.dh[i];
        o.dhacc[nt] += o.dh[i];
        o.dh_n[nt] = o.dh[i];
    }
}

// second half, after the halo exchange delivered dh and coordinates of halo nodes
void surface_processes_b(des_oracle &o)
{
    const int nn = o.nn;
    for (int i = 0; i < o.ntop; i++) o.dh[i] = o.dh_n[o.top_nodes[i]];
    for (int i = 0; i < o.etop; i++) {
        double dh_e = 0.;
        for (int j = 0; j < ND; j++)
            dh_e += o.dh[o.elem_and_nodes[j * o.etop + i]];
        double c[NPF][ND];
        for (int j = 0; j < NPF; ++j)
            node_xyz(o.coord, nn, o.conn_surf[j * o.etop + i], c[j]);
#if DES_NDIMS == 3
        double base = triangle_area2d(c[0], c[1], c[2]);     // compute_area_facet, geometry.cxx:109-121
#else
        double base = std::fabs(c[0][0] - c[1][0]);
#endif
        o.edvacc_surf[i] += dh_e * base / ND;
    }

    double maxdh = 0.;
    for (int i = 0; i < o.ntop; ++i) {
        double tmp = std::fabs(o.dh[i]);
        if (maxdh < tmp) maxdh = tmp;
    }
    o.max_surf_vel = maxdh / o.dt;

    correct_surface_element(o);

    if (o.steps != 0 && o.steps % o.p.quality_check_step_interval == 0) {
        // correct_surface_marker / set_surface_marker act on host markers (out of scope);
        // the dhacc reset between them is kept (bc.cxx:1837-1838)
        for (int i = 0; i < o.ntop; i++)
            o.dhacc[o.top_nodes[i]] = 0.;
    }
#if DES_NDIMS == 2
    // surface_plstrain_diffusion (bc.cxx:1633-1653) at step 0 and every quality_check_step_interval
    // steps (bc.cxx:1848-1850): plastic strain of the top elements decays with a 100-year half life
    // unless their most abundant material is the oceanic crust
    if (!(o.steps % o.p.quality_check_step_interval && o.steps != 0)) {
        double half_life = 1.e2 * YEAR2SEC;
        double lambha = 0.69314718056 / half_life;
        for (int i = 0; i < o.ntop_elems; i++) {
            const int e = o.top_elems[i];
            const int *a = &o.elemmarkers[(size_t)e * o.p.nmat];
            int mat = (int)(std::max_element(a, a + o.p.nmat) - a);
            if (mat != o.p.mattype_oceanic_crust)
                o.plstrain[e] -= o.plstrain[e] * lambha * o.dt;
        }
    }
#endif
}

// geometry.cxx:1743-1870 (use_global_velocity_scaling == false, no hydraulics)
void compute_mass(des_oracle &o)
{
    const des_params &p = o.p;
    const int ne = o.ne;
    Mat mat(o);
    const double pseudo_speed = p.max_vbc_val * p.inertial_scaling;
    #pragma omp parallel for
    for (int e = 0; e < ne; e++) {
        double rho = p.is_quasi_static ? mat.bulkm(e) / (pseudo_speed * pseudo_speed)
                                       : mat.rho(e);
        double bulk_comp = 1.0 / mat.bulkm(e);
        // alpha_biot and beta_fluid are constant per material set; with hydraulics off
        // hmass is never read, it is not reproduced here
        double m = rho * o.volume[e] / NPE;
        double tm = mat.rho(e) * mat.cp(e) * o.volume[e] / NPE;
        double ym = 9 * mat.bulkm(e) * mat.shearm(e) / (3 * mat.bulkm(e) + mat.shearm(e)) / NPE;
        (void)bulk_comp;
        o.tmp_result[0 * ne + e] = o.volume[e];
        o.tmp_result[1 * ne + e] = m;
        if (p.has_thermal_diffusion)
            o.tmp_result[2 * ne + e] = tm;
        o.tmp_result[4 * ne + e] = ym;
    }
    #pragma omp parallel for
    for (int n = o.c0; n < o.c1; n++) {
        double vn = 0, ms = 0, tms = 0, yms = 0;
        for (int k = o.sup_idx[n]; k < o.sup_idx[n+1]; ++k) {
            const int e = o.sup_arr[k];
            vn += o.tmp_result[0 * ne + e];
            ms += o.tmp_result[1 * ne + e];
            if (p.has_thermal_diffusion)
                tms += o.tmp_result[2 * ne + e];
            yms += o.tmp_result[4 * ne + e];
        }
        o.volume_n[n] = vn; o.mass[n] = ms; o.tmass[n] = tms; o.ymass[n] = yms;
    }
}

// dynearthsol.cxx:448-493

void update_mesh_b(des_oracle &o)
{
    surface_processes_b(o);
    o.volume.swap(o.volume_old);
    compute_volume(o, o.volume);
    refresh_elem_cache(o);
    compute_mass(o);
}

#if DES_NDIMS == 2
// fields.cxx:807-821, 885-900
void jaumann_rate_2d(double *s, double dt, double w2)
{
    double s_inc[3];
    s_inc[0] = -2.0 * s[2] * w2;
    s_inc[1] =  2.0 * s[2] * w2;
    s_inc[2] = s[0] * w2 - s[1] * w2;
    for (int i = 0; i < NSTR; ++i)
        s[i] += dt * s_inc[i];
}

void rotate_stress(des_oracle &o)
{
    const int ne = o.ne, nn = o.nn;
    #pragma omp parallel for
    for (int e = 0; e < ne; ++e) {
        double shpdx[NPE], shpdy[NPE], shpdz[NPE];
        get_local_shape_fn(o, e, shpdx, shpdy, shpdz);
        double v[NPE][ND];
        for (int i = 0; i < NPE; ++i)
            node_xyz(o.vel, nn, o.conn[i * ne + e], v[i]);
        double w2 = 0;
        for (int i = 0; i < NPE; ++i) w2 += 0.5 * (v[i][ND-1] * shpdx[i] - v[i][0] * shpdz[i]);
        double s[NSTR], es[NSTR];
        for (int i = 0; i < NSTR; ++i) { s[i] = o.stress[i*ne+e]; es[i] = o.strain[i*ne+e]; }
        jaumann_rate_2d(s, o.dt, w2);
        jaumann_rate_2d(es, o.dt, w2);
        for (int i = 0; i < NSTR; ++i) { o.stress[i*ne+e] = s[i]; o.strain[i*ne+e] = es[i]; }
    }
}
#else
// fields.cxx:787-902 (THREED)
void jaumann_rate_3d(double *s, double dt, double w3, double w4, double w5)
{
    double s_inc[6];
    s_inc[0] = -2.0 * s[3] * w3 - 2.0 * s[4] * w4;
    s_inc[1] =  2.0 * s[3] * w3 - 2.0 * s[5] * w5;
    s_inc[2] =  2.0 * s[4] * w4 + 2.0 * s[5] * w5;
    s_inc[3] = s[0] * w3 - s[1] * w3 - s[4] * w5 - s[5] * w4;
    s_inc[4] = s[0] * w4 - s[2] * w4 + s[3] * w5 - s[5] * w3;
    s_inc[5] = s[1] * w5 - s[2] * w5 + s[3] * w4 + s[4] * w3;
    for (int i = 0; i < NSTR; ++i)
        s[i] += dt * s_inc[i];
}

void rotate_stress(des_oracle &o)
{
    const int ne = o.ne, nn = o.nn;
    #pragma omp parallel for
    for (int e = 0; e < ne; ++e) {
        double shpdx[4], shpdy[4], shpdz[4];
        get_local_shape_fn(o, e, shpdx, shpdy, shpdz);
        double v[4][3];
        for (int i = 0; i < NPE; ++i)
            node_xyz(o.vel, nn, o.conn[i * ne + e], v[i]);
        double w3 = 0, w4 = 0, w5 = 0;
        for (int i = 0; i < NPE; ++i) w3 += 0.5 * (v[i][0] * shpdy[i] - v[i][1] * shpdx[i]);
        for (int i = 0; i < NPE; ++i) w4 += 0.5 * (v[i][0] * shpdz[i] - v[i][2] * shpdx[i]);
        for (int i = 0; i < NPE; ++i) w5 += 0.5 * (v[i][1] * shpdz[i] - v[i][2] * shpdy[i]);
        double s[6], es[6];
        for (int i = 0; i < NSTR; ++i) { s[i] = o.stress[i*ne+e]; es[i] = o.strain[i*ne+e]; }
        jaumann_rate_3d(s, o.dt, w3, w4, w5);
        jaumann_rate_3d(es, o.dt, w3, w4, w5);
        for (int i = 0; i < NSTR; ++i) { o.stress[i*ne+e] = s[i]; o.strain[i*ne+e] = es[i]; }
    }
}
#endif

// geometry.cxx:1480-1647 (use_global_velocity_scaling == false, no hydraulics).
// The element reduction (1513-1593) gives six partial values; in a decomposed run they are
// min-reduced over the ranks before the tail (1597-1646) turns them into dt.
void compute_dt_partials(des_oracle &o)
{
    const des_params &p = o.p;
    const int ne = o.ne, nn = o.nn;
    Mat mat(o);
    double dt_maxwell = std::numeric_limits<double>::max();
    double dt_diffusion = std::numeric_limits<double>::max();
    double minl = std::numeric_limits<double>::max();
    double global_max_vem = 0.0;
    double global_dt_min = std::numeric_limits<double>::max();

    #pragma omp parallel for reduction(min:minl,dt_maxwell,dt_diffusion,global_dt_min) reduction(max:global_max_vem)
    for (int e = 0; e < ne; ++e) {
        double vx = 0.0, vy = 0.0;
        double weight = 1.0 / NPE;
#if DES_NDIMS == 3
        double vz = 0.0;
        for (int j = 0; j < NPE; ++j) {
            int n = o.conn[j * ne + e];
            vx += o.vel[n] * weight;
            vy += o.vel[nn + n] * weight;
            vz += o.vel[2*nn + n] * weight;
        }
        double max_vem = std::sqrt(vx*vx + vy*vy + vz*vz);
#else
        for (int j = 0; j < NPE; ++j) {
            int n = o.conn[j * ne + e];
            vx += o.vel[n] * weight;
            vy += o.vel[nn + n] * weight;
        }
        double max_vem = std::sqrt(vx*vx + vy*vy);
#endif
        global_max_vem = std::max(global_max_vem, max_vem);

        double d[NPE][ND];
        elem_coords(o, e, d);
#if DES_NDIMS == 3
        const double *a = d[0], *b = d[1], *c = d[2], *dd = d[3];
        double maxa = std::max(std::max(triangle_area(a, b, c), triangle_area(a, b, dd)),
                               std::max(triangle_area(c, dd, a), triangle_area(c, dd, b)));
        double minh = 3 * o.volume[e] / maxa;
#else
        // max edge length of this triangle, geometry.cxx:1569-1575
        double maxl = std::sqrt(std::max(std::max(dist2(d[0], d[1]), dist2(d[1], d[2])), dist2(d[0], d[2])));
        double minh = 2 * o.volume[e] / maxl;
#endif
        dt_maxwell = std::min(dt_maxwell, 0.5 * p.visc_min / (1e-40 + mat.shearm(e)));
        if (p.has_thermal_diffusion)
            dt_diffusion = std::min(dt_diffusion, 0.5 * minh * minh / p.therm_diff_max);
        minl = std::min(minl, minh);
        global_dt_min = std::min(global_dt_min, minh / std::sqrt(mat.shearm(e) / mat.rho(e)) / 5.0);
    }
    o.dt_part[0] = minl; o.dt_part[1] = dt_maxwell; o.dt_part[2] = dt_diffusion;
    o.dt_part[3] = global_dt_min; o.dt_part[4] = -global_max_vem; o.dt_part[5] = -o.max_surf_vel;
}

double compute_dt_finalize(des_oracle &o)
{
    const des_params &p = o.p;
    const double minl = o.dt_part[0], dt_maxwell = o.dt_part[1], dt_diffusion = o.dt_part[2];
    const double dt_hydro_diffusion = std::numeric_limits<double>::max();
    double global_max_vem = -o.dt_part[4];
    o.max_surf_vel = -o.dt_part[5];
    double max_vbc_val;
    if (p.characteristic_speed == 0) {
        max_vbc_val = p.max_vbc_val;
        if (p.surface_process_option > 0)
            max_vbc_val = std::max(max_vbc_val, o.max_surf_vel * 5e-1);
    } else
        max_vbc_val = p.characteristic_speed;

    global_max_vem = std::max(global_max_vem, p.max_vbc_val);
    o.max_global_vel_mag = global_max_vem;
    o.global_dt_min = o.dt_part[3];

    double dt_advection = 0.5 * minl / max_vbc_val;
    double dt_elastic = p.is_quasi_static
        ? 0.5 * minl / (max_vbc_val * p.inertial_scaling)
        : 0.5 * minl / std::sqrt(p.bulk_modulus[p.mattype_ref] / p.rho0[p.mattype_ref]);

    double dt = std::min(std::min(std::min(dt_elastic, dt_maxwell), std::min(dt_advection, dt_diffusion)),
                         dt_hydro_diffusion) * p.dt_fraction;
    if (p.fixed_dt != 0) dt = p.fixed_dt;               // geometry.cxx:1487
    if (dt <= 0) o.status = DES_ERR_RUNTIME_NAN;
    return dt;
}

double compute_dt(des_oracle &o)
{
    compute_dt_partials(o);
    return compute_dt_finalize(o);
}

// One pass of dynearthsol.cxx:768-894 in four phases; a decomposed run exchanges halo values
// between them (des_params.h: DES_X_*).  phase 4 returns 1 when the dt partials are ready.
// Output::average_fields, output.cxx:327-370 (called every step, dynearthsol.cxx:897-898;
// nothing on the step reads its state, so its place after rotate_stress is immaterial)
void average_fields(des_oracle &o)
{
    const int average_interval = o.p.quality_check_step_interval;
    if (o.steps % average_interval == 1) {
        o.avg_time0 = o.time;
        o.coord_avg0 = o.coord;
        o.strain0 = o.strain;
        o.stress_avg = o.stress;
        o.dplstrain_avg = o.delta_plstrain;
    } else {
        for (size_t i = 0; i < o.stress_avg.size(); ++i) o.stress_avg[i] += o.stress[i];
        for (size_t i = 0; i < o.dplstrain_avg.size(); ++i) o.dplstrain_avg[i] += o.delta_plstrain[i];
    }
}

// One step in its two phases (des_params.h, des_halo): everything up to the committed surface
// heights runs on the local mesh alone -- redundantly on the ghost region -- then the ghost
// region is refreshed by the exchange, then the end-of-step geometry pass.
// isostasy_adjustment's loop body (dynearthsol.cxx:506-539): the step without the clock, the
// temperature update, NMD, the velocity bcs, rotate_stress and compute_dt; velocities are made
// vertical (and zero on a bottom without Winkler foundation) before update_mesh.
void isostasy_vel(des_oracle &o)
{
    const int nn = o.nn;
    #pragma omp parallel for default(none) shared(o) firstprivate(nn)
    for (int i = o.c0; i < o.c1; ++i) {
        for (int j = 0; j < ND-1; ++j) o.vel[j*nn + i] = 0;
        if (!o.p.has_winkler_foundation && (o.bcflag[i] & BOUNDZ0))
            o.vel[(ND-1)*nn + i] = 0;
    }
}

int isostasy_phase(des_oracle &o, int phase)
{
    if (phase == 0) {
        refresh_elem_cache(o);
        update_strain_rate(o);
        compute_dvoldt(o);
        compute_edvoldt(o);
        update_stress(o);
        update_force(o);
        update_velocity(o);
        o.l2_residual = calculate_residual_force(o);       // not printed by the reference here; harmless
        isostasy_vel(o);
        update_coordinate(o);
        surface_processes_a(o);
    } else {
        update_mesh_b(o);
    }
    return 0;
}

// The pseudo-transient loop of a step (dynearthsol.cxx:803-864): the quasi-static part of the step
// repeated with the boundaries at rest (PT_jump: bc.cxx:330-343) and without surface processes
// (update_mesh, dynearthsol.cxx:456-461) until the residual stops changing.  Not restated: the
// mesh-quality check inside the loop (:839-859), which can only end in remesh().
// one iteration (on a decomposed mesh: the ghost region refreshed just before)
void pt_iteration(des_oracle &o)
{
    const des_params &p = o.p;
    apply_vbcs(o);
    if (p.has_moving_mesh) {
        update_coordinate(o);                  // update_mesh with PT_jump: no surface_processes
        o.volume.swap(o.volume_old);
        compute_volume(o, o.volume);
        refresh_elem_cache(o);
        compute_mass(o);
    }
    update_strain_rate(o);
    compute_dvoldt(o);
    compute_edvoldt(o);
    update_stress(o);
    update_force(o);
    update_velocity(o);
    o.l2_residual = calculate_residual_force(o);
    ++o.n_pt_iterations;
}

void pt_loop(des_oracle &o)
{
    const des_params &p = o.p;
    double residual_old = o.l2_residual;
    o.pt_jump = true;
    for (int pt_step = 0; pt_step < p.PT_max_iter; ++pt_step) {
        pt_iteration(o);
        double relative_change = std::fabs((o.l2_residual - residual_old) / residual_old);
        if (relative_change < p.PT_relative_tolerance) break;
        residual_old = o.l2_residual;
    }
    o.pt_jump = false;
}

int step_phase(des_oracle &o, int phase)
{
    const des_params &p = o.p;
    if (o.iso) return isostasy_phase(o, phase);
    switch (phase) {
    case 0:
        o.steps++;
        o.time += o.dt;
        refresh_elem_cache(o);
        if (p.has_thermal_diffusion)
            update_temperature(o);
        update_strain_rate(o);
        compute_dvoldt(o);
        compute_edvoldt(o);
        update_stress(o);
        if (p.is_using_mixed_stress) {
            NMD_stress_gather(o);
            NMD_stress_apply(o);
        }
        update_force(o);
        update_velocity(o);
        o.l2_residual = calculate_residual_force(o);
        if (p.has_PT) {
            if (o.o0 > 0 || o.o1 < o.nn || o.nn != o.nn_global) {
                // a rank of a decomposed run: the loop is the caller's (ghost region refreshed and the residual's block
                // partials put together across ranks every iteration: des_params.h) -- phase 2 = one iteration, 3 = the rest
                o.pt_jump = true;
                return 2;
            }
            pt_loop(o);
        }
        // fall through
    case 3:
        o.pt_jump = false;
        apply_vbcs(o);
        if (p.has_moving_mesh) {
            update_coordinate(o);
            surface_processes_a(o);
        }
        return 0;                                   // -> exchange of the ghost region
    case 2:
        pt_iteration(o);
        return 0;
    case 1:
        if (p.has_moving_mesh)
            update_mesh_b(o);
        if (p.rheol_type & DES_RH_ELASTIC)
            rotate_stress(o);
        if (p.is_outputting_averaged_fields)
            average_fields(o);
        if (o.steps % 10 == 0) {
            refresh_elem_cache(o);
            compute_dt_partials(o);
            return 1;                               // -> min-reduce dt_part, then dt finalize
        }
        return 0;
    }
    return 0;
}

void one_step(des_oracle &o)
{
    step_phase(o, 0);
    if (step_phase(o, 1)) o.dt = compute_dt_finalize(o);
}

template <typename T>
void copy_vec(std::vector<T> &dst, const T *src, size_t n) { dst.assign(src, src + n); }

struct FieldRef { void *ptr; long long count; int elsize; };

FieldRef field_ref(des_oracle &o, int field)
{
    const long long nn = o.nn, ne = o.ne;
    switch (field) {
    case DES_F_COORD: return {o.coord.data(), ND*nn, 8};
    case DES_F_VEL: return {o.vel.data(), ND*nn, 8};
    case DES_F_FORCE: return {o.force.data(), ND*nn, 8};
    case DES_F_FORCE_RESIDUAL: return {o.force_residual.data(), ND*nn, 8};
    case DES_F_COORD0: return {o.coord0.data(), ND*nn, 8};
    case DES_F_TEMPERATURE: return {o.temperature.data(), nn, 8};
    case DES_F_VOLUME_N: return {o.volume_n.data(), nn, 8};
    case DES_F_MASS: return {o.mass.data(), nn, 8};
    case DES_F_TMASS: return {o.tmass.data(), nn, 8};
    case DES_F_DHACC: return {o.dhacc.data(), nn, 8};
    case DES_F_NTMP: return {o.ntmp.data(), nn, 8};
    case DES_F_STRESS: return {o.stress.data(), NSTR*ne, 8};
    case DES_F_STRAIN: return {o.strain.data(), NSTR*ne, 8};
    case DES_F_STRAIN_RATE: return {o.strain_rate.data(), NSTR*ne, 8};
    case DES_F_PLSTRAIN: return {o.plstrain.data(), ne, 8};
    case DES_F_DELTA_PLSTRAIN: return {o.delta_plstrain.data(), ne, 8};
    case DES_F_VISCOSITY: return {o.viscosity.data(), ne, 8};
    case DES_F_VOLUME: return {o.volume.data(), ne, 8};
    case DES_F_VOLUME_OLD: return {o.volume_old.data(), ne, 8};
    case DES_F_DPRESSURE: return {o.dpressure.data(), ne, 8};
    case DES_F_EDVOLDT: return {o.edvoldt.data(), ne, 8};
    case DES_F_RADIOGENIC: return {o.radiogenic.data(), ne, 8};
    case DES_F_ELEMMARKERS: return {o.elemmarkers.data(), ne * o.p.nmat, 4};
    case DES_F_EDVACC_SURF: return {o.edvacc_surf.data(), (long long)o.etop, 8};
    case DES_F_DH: return {o.dh.data(), (long long)o.ntop, 8};
    case DES_F_STRESS_AVG: return {o.stress_avg.data(), NSTR*ne, 8};
    case DES_F_DPLSTRAIN_AVG: return {o.dplstrain_avg.data(), ne, 8};
    case DES_F_STRAIN0: return {o.strain0.data(), NSTR*ne, 8};
    case DES_F_COORD_AVG0: return {o.coord_avg0.data(), ND*nn, 8};
    case DES_F_STRESSYY: return {o.stressyy.data(), (long long)o.stressyy.size(), 8};
    default: return {nullptr, 0, 0};
    }
}

} // namespace

extern "C" {

des_oracle *des_oracle_create(const des_params *params, const des_mesh *mesh)
{
    if (!params || !mesh || params->ndims != ND || params->nmat < 1 || params->nmat > DES_MAX_MAT)
        return nullptr;
    if (ND == 2 && (params->num_vbc_period_x0 < 1 || params->num_vbc_period_x0 > DES_MAX_PERIOD ||
                    params->num_vbc_period_x1 < 1 || params->num_vbc_period_x1 > DES_MAX_PERIOD))
        return nullptr;
    des_oracle *h = new des_oracle();
    des_oracle &o = *h;
    o.p = *params;
    const int nn = o.nn = mesh->nnode, ne = o.ne = mesh->nelem;
    copy_vec(o.conn, mesh->connectivity, (size_t)NPE * ne);
    copy_vec(o.sup_idx, mesh->support_idx, (size_t)nn + 1);
    copy_vec(o.sup_arr, mesh->support_arr, (size_t)NPE * ne);
    copy_vec(o.sup_lidx, mesh->support_lidx, (size_t)NPE * ne);
    copy_vec(o.bcflag, mesh->bcflag, (size_t)nn);
    for (int i = 0; i < DES_NBDRY; ++i) {
        copy_vec(o.bf_elem[i], mesh->bfacet_elem[i], (size_t)mesh->nbfacets[i]);
        copy_vec(o.bf_facet[i], mesh->bfacet_facet[i], (size_t)mesh->nbfacets[i]);
        copy_vec(o.bnodes[i], mesh->bnodes[i], (size_t)mesh->nbnodes[i]);
    }
    copy_vec(o.bnormals, mesh->bnormals, (size_t)ND * DES_NBDRY);
    copy_vec(o.edge_vec, mesh->edge_vec, (size_t)ND * mesh->nedge);
    std::memcpy(o.edge_slot, mesh->edge_slot, sizeof(o.edge_slot));
    o.ntop = mesh->ntop; o.etop = mesh->etop; o.ntop_elems = mesh->ntop_elems;
    copy_vec(o.top_nodes, mesh->top_nodes, (size_t)o.ntop);
    copy_vec(o.elem_and_nodes, mesh->elem_and_nodes, (size_t)ND * o.etop);
    copy_vec(o.conn_surf, mesh->connectivity_surface, (size_t)NPE * o.etop);
    copy_vec(o.ssup_idx, mesh->support_surf_idx, (size_t)o.ntop + 1);
    copy_vec(o.ssup_arr, mesh->support_surf_arr, (size_t)o.ssup_idx[o.ntop]);
    copy_vec(o.top_elems, mesh->top_elems, (size_t)o.ntop_elems);

    for (dvec *v : {&o.coord, &o.vel, &o.force, &o.force_residual, &o.coord0})
        v->assign((size_t)ND * nn, 0.0);
    for (dvec *v : {&o.temperature, &o.volume_n, &o.mass, &o.tmass, &o.hmass, &o.ymass,
                    &o.dhacc, &o.ntmp, &o.total_dx, &o.total_slope})
        v->assign((size_t)nn, 0.0);
    for (dvec *v : {&o.stress, &o.strain, &o.strain_rate})
        v->assign((size_t)NSTR * ne, 0.0);
    if (ND == 2) o.stressyy.assign((size_t)ne, 0.0);
    for (dvec *v : {&o.plstrain, &o.delta_plstrain, &o.volume, &o.volume_old, &o.dpressure,
                    &o.edvoldt, &o.radiogenic, &o.etmp, &o.c_bulkm, &o.c_shearm, &o.c_phi,
                    &o.c_cp, &o.c_k})
        v->assign((size_t)ne, 0.0);
    o.viscosity.assign((size_t)ne, params->visc_max);          // fields.cxx:110
    o.stress_avg.assign((size_t)NSTR * ne, 0.0); o.strain0.assign((size_t)NSTR * ne, 0.0);
    o.dplstrain_avg.assign((size_t)ne, 0.0); o.coord_avg0.assign((size_t)ND * nn, 0.0);
    o.avg_time0 = 0;
    o.tmp_result.assign((size_t)12 * ne, 0.0);
    o.elemmarkers.assign((size_t)ne * params->nmat, 0);
    o.etmp_int.assign((size_t)ne, -1);
    o.dh.assign((size_t)o.ntop, 0.0);
    o.dh_n.assign((size_t)nn, 0.0);
    o.o0 = 0; o.o1 = nn; o.c0 = 0; o.c1 = nn; o.nn_global = nn; o.l2_part = 0;
    for (int i = 0; i < 6; ++i) o.dt_part[i] = 0;
    o.edvacc_surf.assign((size_t)o.etop, 0.0);
    o.markers_dirty = true;

    // matprops.cxx:237-250
    const double gas_constant = 8.3144;
    for (int m = 0; m < params->nmat; ++m) {
        o.visc_pow_edot[m] = 1 / params->visc_exponent[m] - 1;
        const double pow1 = -1 / params->visc_exponent[m];
        o.visc_coef_term[m] = std::pow(0.75 * params->visc_coefficient[m], pow1);
        o.visc_nR[m] = params->visc_exponent[m] * gas_constant;
    }
    o.dt = 0; o.time = 0; o.steps = 0; o.l2_residual = 0; o.max_surf_vel = 0;
    o.max_global_vel_mag = 0; o.global_dt_min = 0; o.status = DES_OK;
    return h;
}

void des_oracle_destroy(des_oracle *h) { delete h; }

long long des_oracle_field_count(const des_oracle *h, int field)
{
    return field_ref(*const_cast<des_oracle *>(h), field).count;
}

int des_oracle_upload(des_oracle *h, int field, const void *host, long long count)
{
    FieldRef r = field_ref(*h, field);
    if (!r.ptr && r.count == 0 && field != DES_F_EDVACC_SURF && field != DES_F_DH) return DES_ERR_INTERNAL;
    if (count != r.count) return DES_ERR_INTERNAL;
    std::memcpy(r.ptr, host, (size_t)count * r.elsize);
    if (field == DES_F_ELEMMARKERS) h->markers_dirty = true;
    return DES_OK;
}

int des_oracle_download(des_oracle *h, int field, void *host, long long count)
{
    FieldRef r = field_ref(*h, field);
    if (count != r.count) return DES_ERR_INTERNAL;
    std::memcpy(host, r.ptr, (size_t)count * r.elsize);
    return DES_OK;
}

int des_oracle_set_clock(des_oracle *h, double dt, double time, long long steps)
{
    h->dt = dt; h->time = time; h->steps = steps;
    return DES_OK;
}

// dynearthsol.cxx:184-194 (compute_volume, volume_old = volume, apply_vbcs, compute_mass)
int des_oracle_init_geometry(des_oracle *h)
{
    refresh_elem_cache(*h);
    compute_volume(*h, h->volume);
    h->volume_old = h->volume;
    apply_vbcs(*h, true);
    compute_mass(*h);
    return DES_OK;
}

int des_oracle_compute_dt(des_oracle *h, double *dt)
{
    refresh_elem_cache(*h);
    h->dt = compute_dt(*h);
    if (dt) *dt = h->dt;
    return h->dt > 0 ? DES_OK : DES_ERR_RUNTIME_NAN;
}

int des_oracle_step(des_oracle *h, int nsteps, des_scalars *out)
{
    h->n_pt_iterations = 0;
    for (int i = 0; i < nsteps; ++i)
        one_step(*h);
    if (out) {
        out->dt = h->dt; out->time = h->time; out->l2_residual = h->l2_residual;
        out->max_surf_vel = h->max_surf_vel; out->max_global_vel_mag = h->max_global_vel_mag;
        out->global_dt_min = h->global_dt_min; out->steps = h->steps; out->status = h->status;
        out->n_return_mapping = h->n_return_mapping; out->avg_time0 = h->avg_time0;
        out->n_pt_iterations = h->n_pt_iterations;
    }
    return h->status;
}

// utils.hpp:323-394
int des_oracle_check_nan(des_oracle *h, long long *n_nan)
{
    long long n = 0;
    for (const dvec *v : {&h->volume, &h->dpressure, &h->viscosity, &h->stress,
                          &h->temperature, &h->tmass, &h->force, &h->vel, &h->coord})
        for (double x : *v) if (std::isnan(x)) ++n;
    if (n_nan) *n_nan = n;
    return n ? DES_ERR_RUNTIME_NAN : DES_OK;
}

// bad_mesh_quality's loops (remeshing.cxx:2765-2798, 2841-2844) + elem_quality /
// worst_elem_quality (geometry.cxx:1873-1927)
int des_oracle_mesh_quality(des_oracle *h, double smallest_vol, double bottom, double bottom_dist, des_quality *out)
{
    des_oracle &o = *h;
    out->small_elem = out->bottom_node = -1; out->pad_ = 0;
    for (int e = 0; e < o.ne; e++)
        if (o.volume[e] < smallest_vol) { out->small_elem = e; break; }
    if (bottom_dist >= 0)
        for (int i = 0; i < o.nn; ++i)
            if (o.bcflag[i] & BOUNDZ0) {
                double z = o.coord[(ND-1) * (size_t)o.nn + i];
                if (std::fabs(z - bottom) > bottom_dist) { out->bottom_node = i; break; }
            }
    double q = 1; int worst = 0;
    for (int e = 0; e < o.ne; e++) {
        double d[NPE][ND];
        elem_coords(o, e, d);
        double vol = o.volume[e];
#if DES_NDIMS == 3
        double normalization_factor = 216 * std::sqrt(3);
        double area_sum = (triangle_area(d[0], d[1], d[2]) + triangle_area(d[0], d[1], d[3]) +
                           triangle_area(d[2], d[3], d[0]) + triangle_area(d[2], d[3], d[1]));
        double quality = normalization_factor * vol * vol / (area_sum * area_sum * area_sum);
#else
        double normalization_factor = 4 * std::sqrt(3);
        double dist2_sum = dist2(d[0], d[1]) + dist2(d[1], d[2]) + dist2(d[0], d[2]);
        double quality = normalization_factor * vol / dist2_sum;
#endif
        if (quality < q) { q = quality; worst = e; }
    }
    out->worst_quality = q; out->worst_elem = worst;
    return DES_OK;
}

// ---- domain decomposition hooks (tests drive the exchanges with torch.distributed/gloo) ----
int des_oracle_set_halo(des_oracle *h, int owned_begin, int owned_end, int nnode_global)
{
    if (owned_begin < 0 || owned_end > h->nn || owned_begin > owned_end) return DES_ERR_INTERNAL;
    h->o0 = owned_begin; h->o1 = owned_end; h->nn_global = nnode_global;
    return DES_OK;
}

// des_halo::owned_global_begin: where this rank's owned nodes sit in the global numbering (the residual's blocks)
int des_oracle_set_owned_global(des_oracle *h, int owned_global_begin) { h->g0 = owned_global_begin; return DES_OK; }

// The partition-independent residual across ranks (des_params.h: DES_RES_BLOCK): this rank's block partials -- first = the
// global index of its first block --, and, once the caller has put every rank's together, the fixed-shape sum: sets and
// returns l2_residual.
int des_oracle_residual_blocks(des_oracle *h, double *out, int cap, int *first, int *count)
{
    des_oracle &o = *h;
    const int B = des_res_block(o.nn_global);
    residual_blocks_local(o);
    const int b0 = o.g0 / B, nown = o.o1 - o.o0, nb = (nown + B - 1) / B;
    if (first) *first = b0;
    if (count) *count = nb;
    if (!out || cap < nb) return DES_ERR_INTERNAL;
    for (int k = 0; k < nb; ++k) out[k] = o.res_blocks[(size_t)b0 + k];
    return DES_OK;
}

int des_oracle_residual_set(des_oracle *h, const double *blocks, int nblocks, double *l2)
{
    if (!blocks || nblocks != residual_nblocks(*h)) return DES_ERR_INTERNAL;
    h->l2_part = residual_final(blocks, nblocks);
    h->l2_residual = std::sqrt(h->l2_part);
    if (l2) *l2 = h->l2_residual;
    return DES_OK;
}

int des_oracle_phase(des_oracle *h, int phase) { return step_phase(*h, phase); }

int des_oracle_set_isostasy(des_oracle *h, int on) { h->iso = on != 0; return DES_OK; }

// initial_body_force_adjustment (dynearthsol.cxx:546-591, called once before the time loop when
// ic.has_body_force_adjustment): the pseudo-transient loop on the initial state, Neumann tractions held back
// (fields.cxx:690).  Without control.has_PT it only forms the residual, as the reference does.
int des_oracle_body_force_adjustment(des_oracle *h, des_scalars *out)
{
    des_oracle &o = *h;
    o.n_pt_iterations = 0;
    o.l2_residual = calculate_residual_force(o);
    if (o.p.has_PT) {
        o.body_force_adjustment = true;
        pt_loop(o);
        o.body_force_adjustment = false;
    }
    if (out) {
        const long long it = o.n_pt_iterations;
        des_oracle_step(h, 0, out);
        out->n_pt_iterations = it;
    }
    return DES_OK;
}

// the exchange of a step: what = 0 nodal {coord[ND], vel[ND], T, dh} of the local nodes idx[0..n),
// what = 1 {stress[NSTR], strain[NSTR], plstrain (, stressyy in 2-D)} of the local elements idx[0..n); buf[i*width + c]
// (DES_X_NODE_WIDTH / DES_X_ELEM_WIDTH, and their _2D values)
static const int X_NODE_W = 2 * ND + 2, X_ELEM_W = 2 * NSTR + 1 + (ND == 2 ? 1 : 0);
int des_oracle_halo_pack(des_oracle *h, int what, const int *idx, int n, double *buf)
{
    const int nn = h->nn, ne = h->ne;
    for (int i = 0; i < n; ++i) {
        const int k = idx[i];
        if (what == 0) {
            double *b = buf + (size_t)i * X_NODE_W;
            for (int d = 0; d < ND; ++d) { b[d] = h->coord[d*nn+k]; b[ND+d] = h->vel[d*nn+k]; }
            b[2*ND] = h->temperature[k]; b[2*ND+1] = h->dh_n[k];
        } else {
            double *b = buf + (size_t)i * X_ELEM_W;
            for (int c = 0; c < NSTR; ++c) { b[c] = h->stress[c*ne+k]; b[NSTR+c] = h->strain[c*ne+k]; }
            b[2*NSTR] = h->plstrain[k];
#if DES_NDIMS == 2
            b[2*NSTR+1] = h->stressyy[k];
#endif
        }
    }
    return DES_OK;
}

int des_oracle_halo_unpack(des_oracle *h, int what, const int *idx, int n, const double *buf)
{
    const int nn = h->nn, ne = h->ne;
    for (int i = 0; i < n; ++i) {
        const int k = idx[i];
        if (what == 0) {
            const double *b = buf + (size_t)i * X_NODE_W;
            for (int d = 0; d < ND; ++d) { h->coord[d*nn+k] = b[d]; h->vel[d*nn+k] = b[ND+d]; }
            h->temperature[k] = b[2*ND]; h->dh_n[k] = b[2*ND+1];
        } else {
            const double *b = buf + (size_t)i * X_ELEM_W;
            for (int c = 0; c < NSTR; ++c) { h->stress[c*ne+k] = b[c]; h->strain[c*ne+k] = b[NSTR+c]; }
            h->plstrain[k] = b[2*NSTR];
#if DES_NDIMS == 2
            h->stressyy[k] = b[2*NSTR+1];
#endif
        }
    }
    return DES_OK;
}

// What the 2-D apply_vbcs reads off the whole mesh (bc.cxx:251-300, 350-361), on this rank's mesh: {max z of the x0 wall,
// max -z of it, max(0, max -z) of all nodes}, -DBL_MAX without a wall node; the caller MAX-reduces across ranks and hands
// the result back (des_dev.h: des_dev_wall_get / des_dev_wall_set).  3-D: nothing of the kind.
int des_oracle_wall_get(des_oracle *h, double out[3])
{
    out[0] = out[1] = out[2] = 0.0;
#if DES_NDIMS == 2
    const int nn = h->nn;
    out[0] = out[1] = -DBL_MAX;
    for (int i = 0; i < nn; ++i) {
        const double z = h->coord[nn + i];
        if (h->bcflag[i] & 1u) { out[0] = std::max(out[0], z); out[1] = std::max(out[1], -z); }
        out[2] = std::max(out[2], -z);
    }
#endif
    return DES_OK;
}

int des_oracle_wall_set(des_oracle *h, const double in[3])
{
#if DES_NDIMS == 2
    h->wall_given = true;
    for (int i = 0; i < 3; ++i) h->wall[i] = in[i];
#endif
    return DES_OK;
}

// compute_dt across ranks: partials out (6 doubles, all to be MIN-reduced), reduced values in
int des_oracle_dt_partials(des_oracle *h, double out[6], int recompute)
{
    if (recompute) { refresh_elem_cache(*h); compute_dt_partials(*h); }
    for (int i = 0; i < 6; ++i) out[i] = h->dt_part[i];
    return DES_OK;
}

int des_oracle_dt_finalize(des_oracle *h, const double in[6], double *dt)
{
    for (int i = 0; i < 6; ++i) h->dt_part[i] = in[i];
    h->dt = compute_dt_finalize(*h);
    if (dt) *dt = h->dt;
    return h->dt > 0 ? DES_OK : DES_ERR_RUNTIME_NAN;
}

double des_oracle_l2_partial(des_oracle *h) { return h->l2_part; }

void des_oracle_libm_eval(int fn, long long n, const double *x, const double *y, double *out)
{
#pragma omp parallel for
    for (long long i = 0; i < n; ++i) {
        const double a = x[i], b = y ? y[i] : 0.0;
        switch (fn) {
        case 0:  out[i] = deslibm::pow(a, b); break;
        case 1:  out[i] = deslibm::exp(a); break;
        case 2:  out[i] = deslibm::sin(a); break;
        case 3:  out[i] = deslibm::cos(a); break;
        case 4:  out[i] = deslibm::tan(a); break;
        case 6:  { double c_; deslibm::sincos(a, &out[i], &c_); break; }
        case 7:  { double s_; deslibm::sincos(a, &s_, &out[i]); break; }
        default: out[i] = deslibm::atan2(a, b); break;
        }
    }
}

/* The C library's own functions over an array (what the oracle calls by default): the yardstick
 * of tests/test_libm.py for "deslibm::pow / exp return the C library's bits". */
void des_oracle_clib_eval(int fn, long long n, const double *x, const double *y, double *out)
{
#pragma omp parallel for
    for (long long i = 0; i < n; ++i) {
        const double a = x[i], b = y ? y[i] : 0.0;
        switch (fn) {
        case 0:  out[i] = std::pow(a, b); break;
        case 1:  out[i] = std::exp(a); break;
        case 2:  out[i] = std::sin(a); break;
        case 3:  out[i] = std::cos(a); break;
        case 4:  out[i] = std::tan(a); break;
        case 6:  { double c_; ::sincos(a, &out[i], &c_); break; }
        case 7:  { double s_; ::sincos(a, &s_, &out[i]); break; }
        default: out[i] = std::atan2(a, b); break;
        }
    }
}

int des_oracle_set_libm(int portable)
{
    const int old = g_portable_libm;
    if (portable >= 0) g_portable_libm = portable != 0;
    return old;
}

int des_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

int des_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void des_oracle_principal_values3(const double s[6], double p[3]) { principal_values3(s, p); }

void des_oracle_principal_stresses3(const double s[6], double p[3], double v[9])
{
    double vv[3][3];
    principal_stresses3(s, p, vv);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i*3+j] = vv[i][j];
}

static void to33(const double A[9], double a[3][3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = A[i*3+j];
}

int des_oracle_dsyevc3(const double A[9], double w[3])
{
    double a[3][3]; to33(A, a);
    return dsyevc3(a, w);
}

int des_oracle_dsyevh3(const double A[9], double Q[9], double w[3])
{
    double a[3][3], q[3][3]; to33(A, a);
    int r = dsyevh3(a, q, w);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Q[i*3+j] = q[i][j];
    return r;
}

int des_oracle_dsyevq3(const double A[9], double Q[9], double w[3])
{
    double a[3][3], q[3][3]; to33(A, a);
    int r = dsyevq3(a, q, w);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Q[i*3+j] = q[i][j];
    return r;
}

double des_oracle_elasto_plastic(double bulkm, double shearm, double amc, double anphi,
                                 double anpsi, double hardn, double ten_max,
                                 const double de[6], double s[6], int *failure_mode)
{
    double depls = 0; int fm = 0;
    elasto_plastic(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max, de, depls, s, fm);
    if (failure_mode) *failure_mode = fm;
    return depls;
}

#if DES_NDIMS == 2
// rheology.cxx:486-701 on one stress state {XX, ZZ, XZ} + the out-of-plane stress
double des_oracle_elasto_plastic2d(double bulkm, double shearm, double amc, double anphi,
                                   double anpsi, double hardn, double ten_max,
                                   const double de[3], double s[3], double *syy, int *failure_mode)
{
    double depls = 0; int fm = 0;
    elasto_plastic2d(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max, de, depls, s, *syy, fm);
    if (failure_mode) *failure_mode = fm;
    return depls;
}
#endif

void des_oracle_maxwell(double bulkm, double shearm, double viscosity, double dt, double dv,
                        const double de[6], double s[6])
{
    maxwell(bulkm, shearm, viscosity, dt, dv, de, s);
}

} // extern "C"
