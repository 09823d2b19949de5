// des_dev2d.hpp -- the 2-D (triangle) engine behind include/des_dev.h.
//
// The reference is built either -DTHREED or 2-D (constants.hpp:12-25); des_dev_create() looks at
// des_params::ndims and hands a 2-D model to this engine (des_dev2d.hip), which runs the
// !THREED branches of the same step on the GPU: get_local_shape_fn (fields.cxx:40-53),
// principal_stresses2 / elasto_plastic / elasto_plastic2d (rheology.cxx:86-119, 312-484, 486-701),
// the 2-D apply_vbcs (bc.cxx:227-400, 425-481), 1-D surface diffusion (bc.cxx:1021-1033,
// 1067-1106), jaumann_rate_2d (fields.cxx:807-821), the 2-D compute_dt / elem_quality
// (geometry.cxx:1566-1576, 1901-1906).  Every des_dev_* entry point dispatches here when the
// handle holds a 2-D engine, the domain decomposition included (node slabs along x, the two-phase
// step with the caller's exchange, des_dev_step_group, des_dev_step on an RCCL communicator);
// the overlapped schedule of des_dev_set_overlap as well (round 4).
//
// Arrays stay in the reference's own SoA layout and the caller's numbering: the 2-D configs of
// BASELINE.json are the CPU-runnable plumbing case (configs[0]), bit-for-bit parity with the
// oracle's -DDES_NDIMS=2 build is the bar here, not bandwidth.
#ifndef DES_DEV2D_HPP
#define DES_DEV2D_HPP

#include <string>

#include "des_params.h"

namespace des2d {

struct Engine;

Engine *create(int device, const des_params *params, const des_mesh *mesh, int *err, std::string &msg);
void destroy(Engine *h);
const std::string &last_error(const Engine *h);

long long field_count(const Engine *h, int field);
int upload(Engine *h, int field, const void *host, long long count);
int download(Engine *h, int field, void *host, long long count);
int set_clock(Engine *h, double dt, double time, long long steps);
int set_isostasy(Engine *h, int on);
int sync(Engine *h);
int init_geometry(Engine *h);
int compute_dt(Engine *h, double *dt);
int step(Engine *h, int nsteps, des_scalars *out);
int check_nan(Engine *h, long long *n_nan);
int mesh_quality(Engine *h, double smallest_vol, double bottom, double bottom_dist, des_quality *out);
int profile_enable(Engine *h, int on);
int profile_read(Engine *h, int cap, char (*names)[64], double *ms, long long *calls);
int timer_start(Engine *h);
int timer_stop(Engine *h, float *ms);
double algorithmic_bytes_per_step(const Engine *h);
int body_force_adjustment(Engine *h, des_scalars *out);
// domain decomposition
int set_halo(Engine *h, const des_halo *halo, int nn_global);
int phase(Engine *h, int ph);
int halo_pack(Engine *h, int what, const int *idx, int n, double *buf);
int halo_unpack(Engine *h, int what, const int *idx, int n, const double *buf);
int wall_get(Engine *h, double out[3]);
int wall_set(Engine *h, const double in[3]);
int dt_partials(Engine *h, double out[6], int recompute);
int dt_finalize(Engine *h, const double in[6], double *dt);
int step_group(Engine **g, int n, int nsteps, des_scalars *out);
int residual_blocks(Engine *h, double *out, int cap, int *first, int *count);
int residual_set(Engine *h, const double *blocks, int nblocks, double *l2);
// the RCCL communicator (an ncclComm_t the caller owns) des_dev_step / init_geometry / compute_dt of a decomposed engine use
int comm_selfcheck(Engine *h, int expect_world, int expect_rank);
int set_comm(Engine *h, void *comm);
// the overlapped schedule (des_dev_set_overlap, des_dev_comm_info)
int set_overlap(Engine *h, int on);
int overlapped(const Engine *h);

} // namespace des2d

#endif
