/* des_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of DynEarthSol's explicit time step (reference order, unfused), used
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg to check the
 * HIP path.  Same call shape as include/des_dev.h so a test drives both identically.
 */
#ifndef DES_ORACLE_H
#define DES_ORACLE_H

#include "../include/des_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct des_oracle des_oracle;

des_oracle *des_oracle_create(const des_params *params, const des_mesh *mesh);
void des_oracle_destroy(des_oracle *h);
int des_oracle_upload(des_oracle *h, int field, const void *host, long long count);
int des_oracle_download(des_oracle *h, int field, void *host, long long count);
long long des_oracle_field_count(const des_oracle *h, int field);
int des_oracle_set_clock(des_oracle *h, double dt, double time, long long steps);
int des_oracle_init_geometry(des_oracle *h);
int des_oracle_compute_dt(des_oracle *h, double *dt);
int des_oracle_step(des_oracle *h, int nsteps, des_scalars *out);
int des_oracle_check_nan(des_oracle *h, long long *n_nan);
int des_oracle_mesh_quality(des_oracle *h, double smallest_vol, double bottom, double bottom_dist, des_quality *out);
/* domain decomposition (des_halo in des_params.h): owned node range, the two phases of a step
 * with the ghost-region exchange in between, pack/unpack of nodal (what = 0) and element
 * (what = 1) state by local index list, compute_dt across ranks */
int des_oracle_set_halo(des_oracle *h, int owned_begin, int owned_end, int nnode_global);
int des_oracle_phase(des_oracle *h, int phase);
/* with control.has_PT on a decomposed mesh phase 0 stops in front of the pseudo-transient loop and returns 2; the caller
 * then runs the loop: refresh the ghost region, phase 2 (one iteration), the residual put together across ranks
 * (des_oracle_residual_blocks -> all ranks' partials in global block order -> des_oracle_residual_set), the reference's
 * convergence test; phase 3 = the rest of phase 0.  Same protocol as include/des_dev.h. */
int des_oracle_set_owned_global(des_oracle *h, int owned_global_begin);
int des_oracle_residual_blocks(des_oracle *h, double *out, int cap, int *first, int *count);
int des_oracle_residual_set(des_oracle *h, const double *blocks, int nblocks, double *l2);
/* 1: des_oracle_step / des_oracle_phase run the body of isostasy_adjustment's loop
 * (dynearthsol.cxx:506-539) instead of a time step; 0: back to time steps. */
int des_oracle_set_isostasy(des_oracle *h, int on);
/* initial_body_force_adjustment (dynearthsol.cxx:546-591): the pseudo-transient loop on the initial state without
 * the Neumann tractions (fields.cxx:690); out->n_pt_iterations = its iterations */
int des_oracle_body_force_adjustment(des_oracle *h, des_scalars *out);
int des_oracle_halo_pack(des_oracle *h, int kind, const int *idx, int n, double *buf);
int des_oracle_halo_unpack(des_oracle *h, int kind, const int *idx, int n, const double *buf);
int des_oracle_wall_get(des_oracle *h, double out[3]);
int des_oracle_wall_set(des_oracle *h, const double in[3]);
int des_oracle_dt_partials(des_oracle *h, double out[6], int recompute);
int des_oracle_dt_finalize(des_oracle *h, const double in[6], double *dt);
double des_oracle_l2_partial(des_oracle *h);
/* number of OpenMP threads the oracle loops run on (1 unless built with -fopenmp) */
int des_oracle_threads(void);
/* libgomp may already be initialised by the host process (torch loads it), so OMP_NUM_THREADS
 * set late is ignored: set the team size explicitly; returns the size in effect */
int des_oracle_set_threads(int n);
/* 0: libm calls go to the C library (default, as the reference); 1: to the portable set of
 * dynearthsol_amd/csrc/des_libm.hpp; < 0: query only.  Process-wide; returns the old value. */
int des_oracle_set_libm(int portable);
/* CPU build of the portable libm, one function over an array: fn 0 pow, 1 exp, 2 sin, 3 cos,
 * 4 tan, 5 atan2 (the numbering of des_dev_libm_eval). */
void des_oracle_libm_eval(int fn, long long n, const double *x, const double *y, double *out);
/* the same call with the C library's functions (std::pow ...) */
void des_oracle_clib_eval(int fn, long long n, const double *x, const double *y, double *out);

/* Stand-alone pieces exposed for known-answer tests. */
/* eigenvalues (ascending) of the symmetric tensor s = {XX,YY,ZZ,XY,XZ,YZ};
 * rheology.cxx:63-71 -> 3x3-C/dsyevc3.c:31-80 */
void des_oracle_principal_values3(const double s[6], double p[3]);
/* eigenvalues ascending + eigenvectors as columns; rheology.cxx:76-84 -> dsyevh3.c:112-215 */
void des_oracle_principal_stresses3(const double s[6], double p[3], double v[9]);
/* raw restatements of the vendored solvers, row-major A[9], Q[9] */
int des_oracle_dsyevc3(const double A[9], double w[3]);
int des_oracle_dsyevh3(const double A[9], double Q[9], double w[3]);
int des_oracle_dsyevq3(const double A[9], double Q[9], double w[3]);
/* one constitutive update of a single element, rheology.cxx:312-484 (3D):
 * s is updated in place, returns depls; failure_mode out */
double des_oracle_elasto_plastic(double bulkm, double shearm, double amc, double anphi,
                                 double anpsi, double hardn, double ten_max,
                                 const double de[6], double s[6], int *failure_mode);
void des_oracle_maxwell(double bulkm, double shearm, double viscosity, double dt, double dv,
                        const double de[6], double s[6]);

#ifdef __cplusplus
}
#endif
#endif
