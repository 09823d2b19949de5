/* des_dev.h -- C-ABI of the MI355X-native explicit time-stepper (libdes_hip.so).
 *
 * What it replaces.  The reference calls, on a process-global `Variables var`, the
 * free functions below once per step (dynearthsol.cxx:768-894):
 *     MatProps::refresh_elem_cache   matprops.cxx:259
 *     update_temperature             fields.cxx:197       (fields.hpp:7)
 *     update_strain_rate             fields.cxx:405       (fields.hpp:11)
 *     compute_dvoldt / compute_edvoldt geometry.cxx:203/249 (geometry.hpp:15-19)
 *     update_stress                  rheology.cxx:703     (rheology.hpp:4)
 *     NMD_stress                     geometry.cxx:282     (geometry.hpp:21)
 *     update_force (+apply_stress_bcs, apply_stress_bcs_neumann, apply_damping)
 *                                    fields.cxx:609, bc.cxx:661/829, fields.cxx:483
 *     update_velocity                fields.cxx:725
 *     calculate_residual_force       fields.cxx:700
 *     apply_vbcs                     bc.cxx:227           (bc.hpp:8)
 *     update_mesh = update_coordinate + surface_processes + compute_volume + compute_mass
 *                                    dynearthsol.cxx:448-493
 *     rotate_stress                  fields.cxx:827
 *     compute_dt (every 10 steps)    geometry.cxx:1480
 * des_dev_step() runs that whole sequence on the GPU with all state resident in HBM.
 * Ownership, threading and error behaviour mirror the reference: the host
 * owns the Variables arrays; the caller is single-threaded; errors are returned as the
 * reference's ExitCode numbers instead of std::exit().
 *
 * All arrays crossing this boundary are host pointers in the reference's SoA layout
 * (des_params.h).  No torch / HIP types appear in any signature.
 *
 * Dimension.  The reference is compiled for tets (-DTHREED) or triangles (constants.hpp:12-25); here
 * des_params::ndims says which build a model is, and des_dev_create returns an engine for it behind the
 * same entry points.  ndims = 2: coordinates / velocities / forces [2][nnode] = {x, z}, tensors [3][nelem] =
 * {XX, ZZ, XZ}, connectivity [3][nelem], DES_F_STRESSYY [nelem]; the step runs the !THREED branches --
 * get_local_shape_fn (fields.cxx:40-53), principal_stresses2 / elasto_plastic / elasto_plastic2d
 * (rheology.cxx:86-119, 364-483, 486-701), the 2-D apply_vbcs (bc.cxx:247-300, 425-481), the 1-D surface
 * diffusion (bc.cxx:1021-1033, 1067-1106), jaumann_rate_2d (fields.cxx:807-821), the pseudo-transient loop, and the
 * domain decomposition below (des_dev_wall_get / des_dev_wall_set: what a cut 2-D model shares beside the ghost region).
 * What a 2-D engine does not offer returns DES_ERR_UNSUPPORTED_DIM: the stand-alone des_dev_exchange (a 2-D engine runs
 * its exchange inside des_dev_step).  The overlapped schedule (des_dev_set_overlap) is there for both since round 4.
 */
#ifndef DES_DEV_H
#define DES_DEV_H

#include "des_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct des_dev des_dev;   /* opaque engine handle */

/* Number of visible HIP devices (0 when there is no GPU). Does not initialise a context. */
int des_dev_device_count(void);

/* Create an engine on HIP device `device`, copy params and mesh topology to HBM and
 * allocate every field.  Replaces allocate_variables (fields.cxx:56-122) + the topology
 * uploads an OpenACC build does implicitly.  With mesh->coord (initial coordinates, a layout
 * hint) the engine keeps its arrays in a space-filling-curve (Hilbert) order of its own; everything crossing this
 * interface stays in the caller's numbering and list order (des_params.h, des_mesh).
 * params->ndims = 2 creates the 2-D (triangle) engine (see the header above).
 * Returns NULL on failure and stores a DES_ERR_* code in *err (may be NULL). */
des_dev *des_dev_create(int device, const des_params *params, const des_mesh *mesh, int *err);

void des_dev_destroy(des_dev *h);

/* Copy one field host->device / device->host (blocking). `count` is the number of
 * scalar entries the caller's buffer holds and must match the field's size. */
int des_dev_upload(des_dev *h, int field, const void *host, long long count);
int des_dev_download(des_dev *h, int field, void *host, long long count);
long long des_dev_field_count(const des_dev *h, int field);

/* on = 1: des_dev_step (and des_dev_phase) run the body of isostasy_adjustment's loop
 * (dynearthsol.cxx:506-539) instead of a time step -- strain rate, stress, force and velocity
 * update as usual, then horizontal velocities zeroed (and vz on a bottom without Winkler
 * foundation), update_mesh; no clock advance, temperature update, NMD, velocity bcs,
 * rotate_stress or compute_dt.  on = 0: back to time steps.  The caller reproduces
 * dynearthsol.cxx:503-504, 643: dt = compute_dt; iso_steps = years*YEAR2SEC/dt; set_isostasy(1);
 * step(iso_steps); set_isostasy(0); dt = compute_dt. */
int des_dev_set_isostasy(des_dev *h, int on);

/* initial_body_force_adjustment (dynearthsol.cxx:546-591; main() calls it once before the time loop when
 * ic.has_body_force_adjustment is set, :753-761): the pseudo-transient loop of control.has_PT on the initial state,
 * apply_stress_bcs_neumann held back meanwhile (fields.cxx:690).  out (may be NULL): the scalars afterwards,
 * n_pt_iterations = the loop's iterations.  Single domain. */
int des_dev_body_force_adjustment(des_dev *h, des_scalars *out);

/* Set time-step scalars: dt (Variables::dt), time, steps.  compute_dt semantics: if
 * fixed_dt != 0 it always wins (geometry.cxx:1487). */
int des_dev_set_clock(des_dev *h, double dt, double time, long long steps);

/* (Re)compute the derived start-of-run state on the device exactly as init() does after
 * coord/elemmarkers/vel are set (dynearthsol.cxx:184-194): compute_volume, volume_old =
 * volume, apply_vbcs(vel), compute_mass -- in that order, with whatever temperature is
 * resident (the reference runs this before initial_temperature, i.e. with T = 0). */
int des_dev_init_geometry(des_dev *h);

/* compute_dt (geometry.cxx:1480-1647) on the device; writes the new dt into the engine
 * clock and returns it through *dt (may be NULL).  Returns DES_ERR_RUNTIME_NAN if dt<=0. */
int des_dev_compute_dt(des_dev *h, double *dt);

/* Advance `nsteps` explicit time steps (dynearthsol.cxx:768-894 with PT, RSF, phase
 * changes, hydraulics, monitor and output off).  Asynchronous on the engine's stream;
 * des_dev_sync or any download waits.  `out` (may be NULL) receives the scalars after
 * the last step and forces a sync.  After the call every field holds what the reference holds after that many steps
 * (inside a multi-step call the engine leaves out stores and whole passes nothing reads; the call's last step ends
 * with all of them).  A call that follows another one with nothing but downloads / checks in between starts like an
 * interior step; an upload, a clock change or any other entry point that touches the state brings the full first step
 * back.  The results do not depend on how the steps are grouped into calls. */
int des_dev_step(des_dev *h, int nsteps, des_scalars *out);

int des_dev_sync(des_dev *h);

/* check_nan (utils.hpp:323-394) on the device: number of NaN entries in
 * volume, dpressure, viscosity, stress, temperature, tmass, force, vel, coord. */
int des_dev_check_nan(des_dev *h, long long *n_nan);

/* Diagnostic: evaluates one function of the portable libm (dynearthsol_amd/csrc/des_libm.hpp,
 * what the stress update uses under DES_LIBM=portable) on the device, out[i] = fn(x[i], y[i])
 * (y only for pow / atan2, else NULL).  Host pointers.  A CPU build of the same header must
 * give the same bits; no reference counterpart (the reference calls the C library:
 * rheology.cxx:260-300, matprops.cxx:380-418, 3x3-C/dsyevc3.c:60-70). */
/* (DES_LIBM_SINCOS_S / _C: the sine / the cosine as the C library's sincos(x, &s, &c) returns them -- what a compiler makes
 * of sin(x), cos(x) of one argument, 3x3-C/dsyevc3.c:66-67 -- which are not bit for bit its sin(x) and cos(x)) */
enum { DES_LIBM_POW = 0, DES_LIBM_EXP = 1, DES_LIBM_SIN = 2, DES_LIBM_COS = 3, DES_LIBM_TAN = 4, DES_LIBM_ATAN2 = 5,
       DES_LIBM_SINCOS_S = 6, DES_LIBM_SINCOS_C = 7 };
int des_dev_libm_eval(int device, int fn, long long n, const double *x, const double *y, double *out);

/* Diagnostic: the device build of the reference's 3x3 symmetric eigen-solvers over n tensors, so
 * that the vectors produced by the reference's own compiled 3x3-C (tests/golden/eigen_kat.json)
 * can be put to the HIP code itself.  a[n][6] = {A00, A11, A22, A01, A02, A12}; w[n][3];
 * q[n][9] row-major with the eigenvectors in columns (unused for dsyevc3); branch[n] (may be
 * NULL) = 1 where dsyevh3 handed over to the QL solver (3x3-C/dsyevh3.c:152, 177), for dsyevq3
 * its return value.  libm: 0 ocml, 1 the portable set (des_libm.hpp).  Host pointers.
 * Replaces: dsyevc3 (3x3-C/dsyevc3.c:31-80), dsyevh3 (dsyevh3.c:112-215), dsyevq3 + dsytrd3
 * (dsyevq3.c:245-350, dsytrd3.c:379-455). */
enum { DES_EIG_DSYEVC3 = 0, DES_EIG_DSYEVH3 = 1, DES_EIG_DSYEVQ3 = 2 };
int des_dev_eigen_eval(int device, int fn, int libm, long long n, const double *a, double *w, double *q, int *branch);

/* Diagnostic: n independent calls of the device's elasto_plastic (rheology.cxx:312-484, THREED):
 * props[n][7] = {bulkm, shearm, amc, anphi, anpsi, hardn, ten_max}, de[n][6] strain increment,
 * s[n][6] stress in / out, depls[n] out; mode[n] (may be NULL) = the reference's failure_mode
 * (0 none, 1 tensile, 10 shear) + 100 if dsyevh3 fell back to dsyevq3 + 1000 if the element got
 * past the eigenvalue pre-filter (rheology.cxx:354-361). */
int des_dev_elasto_plastic_eval(int device, int libm, long long n, const double *props, const double *de,
                                double *s, double *depls, int *mode);

/* Measured ceiling for the roofline: a streaming device-to-device copy of `bytes`
 * bytes (16 B per lane), `reps` launches timed with HIP events; *gbs = (read + written) / time. */
int des_dev_copy_ceiling(int device, long long bytes, int reps, double *gbs);

/* The same for the stress update's MEMORY SHAPE (no reference counterpart; measurement only): a kernel that does nothing
 * but read `nr` and write `nw` SoA planes of doubles, one element per lane -- 18 + 15 (the 3-D stress update of an interior
 * step: 141 B read, 120 B written per element, rheology.cxx:703-1030 with the end-of-step pass riding in it) or 12 + 9 (the
 * 2-D one) -- over `nelem` elements, `reps` launches; *gbs = (read + written) / time.  With nelem large enough to leave the
 * 256-MiB Infinity Cache behind (bench.py: 8.8M) this is what the pass could reach from HBM if its arithmetic were free. */
int des_dev_plane_ceiling(int device, int nr, int nw, long long nelem, int reps, double *gbs);

/* Calibration of the rocprofv3 HBM counters on the engine's own access shapes (tools/pmc_calibrate.py):
 * `reps` launches over `items` lanes of pattern 0 (16 B/lane stream copy), 1 (8 B/lane stream copy),
 * 2 (32-B record gather through a permutation + 8 B/lane store), 3 (8-B gather + 8 B/lane store). */
int des_dev_access_bench(int device, int pattern, long long items, int reps, double *ms_per_launch);

/* Timing helpers for bench.py: HIP-event bracket on the engine's own stream. */
int des_dev_timer_start(des_dev *h);
int des_dev_timer_stop(des_dev *h, float *ms);
/* Per-kernel accumulated device time since the last reset (HIP events around every
 * launch when profiling is enabled).  names/ms/calls hold up to `cap` entries. */
int des_dev_profile_enable(des_dev *h, int on);
int des_dev_profile_read(des_dev *h, int cap, char (*names)[64], double *ms, long long *calls);

/* bad_mesh_quality's three reductions (remeshing.cxx:2752-2866) on the device: tiny element,
 * distorted bottom (skipped when bottom_dist < 0), worst elem_quality.  Synchronises. */
int des_dev_mesh_quality(des_dev *h, double smallest_vol, double bottom, double bottom_dist, des_quality *out);

/* Algorithmic HBM bytes of one step for this engine's mesh and options (SURVEY.md 8d:
 * 1420*nelem + 348*nnode with the evp / thermal / NMD variants). */
double des_dev_algorithmic_bytes_per_step(const des_dev *h);

/* ---- multi-GPU (one process per GPU; new: the reference is single-process) -------------
 * A rank's engine is created on its LOCAL mesh (des_host_partition: node slab + four-layer ghost
 * region) and told which nodes it owns and what it exchanges (des_halo, des_params.h).  With a
 * communicator attached des_dev_step does a whole step on the local mesh and then refreshes the
 * ghost region with ONE grouped ncclSend/ncclRecv per neighbour (RCCL) on the engine's own
 * stream; des_dev_compute_dt min-reduces its six partials with one ncclAllReduce.  No host
 * synchronisation is added. */
int des_dev_set_halo(des_dev *h, const des_halo *halo, int nnode_global);
/* rank 0 creates the 128-byte ncclUniqueId; the caller broadcasts it (e.g. torch.distributed) */
int des_dev_comm_unique_id(unsigned char *id128);
int des_dev_comm_init(des_dev *h, int nranks, int rank, const unsigned char *id128);
/* what is attached: nranks = ncclCommCount of the communicator (0: none), this rank in it, and
 * whether des_dev_step runs the overlapped schedule (DES_OVERLAP=1 at create: the exchange on a
 * side stream while the end-of-step pass of the interior elements runs; default: everything
 * in order on the engine's stream) */
int des_dev_comm_info(des_dev *h, int *nranks, int *rank, int *overlapped);
/* Start-up self-check of the attached communicator, collective (every rank calls it once, before the first step):
 * ncclCommCount == expect_world; every neighbour message of the ghost-region exchange -- the buffers, lengths and grouped
 * ncclSend / ncclRecv calls des_dev_step itself issues -- is sent filled with a pattern the receiver can verify double by
 * double (sender, receiver, position), so a truncated, swapped or stale message fails; ncclAllReduce SUM / MIN / MAX give
 * their closed forms.  DES_ERR_RESOURCE + des_dev_last_error() names what failed.  New (the reference is single-process);
 * it exists because no multi-GPU node is available where this library is built: the first run on real xGMI must be able
 * to tell "the wires carry my bytes" from "the step is wrong". */
int des_dev_comm_selfcheck(des_dev *h, int expect_world);
/* in-order (0) or overlapped (1) schedule from the next des_dev_step call on; every rank must choose the same */
int des_dev_set_overlap(des_dev *h, int on);
/* the ghost-region exchange through the attached communicator, asynchronous on the engine's
 * stream: what des_dev_step issues between the two phases of a step */
int des_dev_exchange(des_dev *h);

/* The same step cut into its two phases WITHOUT communication (0: up to the committed surface
 * heights, 1: end-of-step geometry; returns 1 when compute_dt partials are ready), plus raw
 * access to the exchanged state -- what = 0: nodes, DES_X_NODE_WIDTH doubles each; 1: elements,
 * DES_X_ELEM_WIDTH -- and the compute_dt partials: lets a host harness move the ghost region
 * itself (tests with several engines on one GPU; any other transport). */
int des_dev_phase(des_dev *h, int phase);
/* control.has_PT (the pseudo-transient loop of a step, dynearthsol.cxx:803-864) on a decomposed mesh: des_dev_step on RCCL
 * and des_dev_step_group run the loop themselves -- the ghost region refreshed before every iteration, the residual summed
 * in ONE association whatever the partition (des_params.h: DES_RES_BLOCK) so that every rank, and a run on one rank, takes
 * the same decision.  With the two-phase entry points the loop is the caller's: phase 0 then stops in front of it and
 * returns 2; per iteration: refresh the ghost region, des_dev_phase(h, 2), des_dev_residual_blocks on every rank -> the
 * partials in global block order -> des_dev_residual_set on every rank (returns the residual), the reference's test
 * |l2 - l2_old| / l2_old < PT_relative_tolerance (l2_old of iteration 0 = the residual before the loop); des_dev_phase(h, 3)
 * is the rest of phase 0.  (tests/test_decomp_cpu.py, dynearthsol_amd/decomp.py: PhasedStepper.pt_loop) */
int des_dev_residual_blocks(des_dev *h, double *out, int cap, int *first, int *count);
int des_dev_residual_set(des_dev *h, const double *blocks, int nblocks, double *l2);
int des_dev_halo_pack(des_dev *h, int what, const int *idx, int n, double *buf);
int des_dev_halo_unpack(des_dev *h, int what, const int *idx, int n, const double *buf);
int des_dev_dt_partials(des_dev *h, double out[6], int recompute);
int des_dev_dt_finalize(des_dev *h, const double in[6], double *dt);
/* 2-D models (des_params::ndims = 2; widths DES_X_NODE_WIDTH_2D / DES_X_ELEM_WIDTH_2D) cut the same way have one more
 * thing to share: apply_vbcs scales its velocity profiles with the vertical extent of the x0 wall and, for vbc_x0 = 3
 * with a bottom shear zone, the lowest node of the mesh (bc.cxx:251-300, 350-361).  out = {max z of the wall, max -z of
 * it, max(0, max -z) over all nodes} on this rank's mesh (-DBL_MAX without a wall node); the caller reduces with MAX
 * across ranks and hands the result back: after every exchange (between phase 0 and phase 1) and once before
 * des_dev_init_geometry.  des_dev_step_group does it itself.  A 3-D engine: zeros out, nothing in. */
int des_dev_wall_get(des_dev *h, double out[3]);
int des_dev_wall_set(des_dev *h, const double in[3]);

/* Several engines of ONE process as the ranks of one decomposed model -- engines[r] is rank r of the
 * des_halo lists (several engines on one GPU: tests and rehearsals of the multi-GPU step at its real
 * partition; engines on several GPUs of one process where peer access is on).  des_dev_step_group is
 * des_dev_step for all of them in lockstep: the same launches in the same order on every engine's own
 * stream, the ghost region refreshed once per step by device-to-device copies between the engines'
 * message buffers (events order them; no communicator, no host synchronisation but for the six
 * compute_dt partials every 10th step).  `out`: NULL or n entries (l2_residual = over all ranks' owned
 * nodes).  An attached engine refuses des_dev_step; des_dev_group_detach before destroying any member.
 * New: the reference is single-process (dynearthsol.cxx:768-894 is what every rank runs). */
int des_dev_group_attach(des_dev **engines, int n);
int des_dev_group_detach(des_dev **engines, int n);
int des_dev_step_group(des_dev **engines, int n, int nsteps, des_scalars *out);
/* initial_body_force_adjustment (dynearthsol.cxx:546-591) for the engines of a group; a rank on RCCL calls
 * des_dev_body_force_adjustment itself (every rank, collectively) */
int des_dev_body_force_adjustment_group(des_dev **engines, int n, des_scalars *out);

const char *des_dev_last_error(void);
/* The engine's environment switches (DES_PATCH, DES_E2GEO, DES2D_CLUSTER, ... -- each selects between
 * code paths that give the same bits) that are SET in this process, as far as the library has read them: "NAME=value
 * NAME=value" in name order, NUL-terminated, cut to len - 1 characters; returns the full length (0: every switch at its
 * default).  A benchmark line carries it so that the record says which path was measured. */
int des_dev_config_string(char *buf, int len);

#ifdef __cplusplus
}
#endif
#endif
