/* des_host.h -- C entry points of the host side (libdes_host.so): ".cfg" front-end,
 * mesh/topology builders, initial conditions and the driver loop.  This is the part of
 * DynEarthSol that stays on the CPU; it mirrors main()/init() (dynearthsol.cxx:159-228,
 * 593-982) and feeds the device engine declared in des_dev.h.
 */
#ifndef DES_HOST_H
#define DES_HOST_H

#include "des_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct des_host des_host;

/* get_input_parameters + init_var + init() up to the point where fields exist
 * (input.cxx:1503, dynearthsol.cxx:36-228).  `overrides`: extra "section.key = value"
 * lines applied on top of the file (may be NULL).  `mesh_file`: a mesh written by the
 * reference mesher for tetgen-based meshing options (may be NULL / empty).
 * On failure returns NULL and stores the reference ExitCode in *err. */
des_host *des_host_create(const char *cfg_path, const char *overrides, const char *mesh_file, int *err);
/* same, from an in-memory config text */
des_host *des_host_create_from_string(const char *cfg_text, const char *overrides, const char *mesh_file, int *err);
/* The reference is built either -DTHREED (tets) or 2-D (triangles; constants.hpp:12-25, `make
 * ndims=2`); the same host library serves both, the dimension being an argument: ndims = 3 or 2,
 * exactly one of cfg_path / cfg_text given.  The two entry points above are the ndims = 3 case.
 * In a 2-D host every array is the reference's 2-D one: coord [2][nnode] = {x, z}, connectivity
 * [3][nelem], tensors {XX, ZZ, XZ}, "stressyy" [nelem] (plane strain). */
des_host *des_host_create_nd(int ndims, const char *cfg_path, const char *cfg_text, const char *overrides,
                             const char *mesh_file, int *err);
void des_host_destroy(des_host *h);

const des_params *des_host_params(const des_host *h);
const des_mesh *des_host_mesh(const des_host *h);

/* Named host arrays in the reference's SoA layout: "coord", "vel", "temperature",
 * "radiogenic", "stress", "strain", "plstrain", "viscosity" (double); "elemmarkers",
 * "connectivity", "segment", "segflag" (int32); the marker set the host keeps:
 * "markerset.eta" (double, SoA [4][nmarkers]), "markerset.elem" / ".mattype" / ".id" (int32).
 * Returns NULL for an unknown name. */
const void *des_host_array(const des_host *h, const char *name, long long *count);

/* typed access to any .cfg option after defaults/normalisation (as strings are parsed) */
int des_host_cfg_int(const des_host *h, const char *key, int *out);
int des_host_cfg_double(const des_host *h, const char *key, double *out);
int des_host_cfg_string(const des_host *h, const char *key, char *out, int cap);

/* write the mesh in the loader's binary format (tools and tests) */
int des_host_save_mesh(const des_host *h, const char *path);

/* Slab decomposition with a four-layer ghost region for one-process-per-GPU runs (des_params.h:
 * des_halo).  The local mesh (des_part_mesh) feeds des_dev_create; des_part_halo feeds des_dev_set_halo; the l2g maps
 * select this rank's part of every global field. */
typedef struct des_part des_part;
des_part *des_host_partition(const des_host *h, int nranks, int rank, int *err);
void des_part_destroy(des_part *p);
const des_mesh *des_part_mesh(const des_part *p);
const des_halo *des_part_halo(const des_part *p);
const int *des_part_l2g_node(const des_part *p, int *n);
const int *des_part_l2g_elem(const des_part *p, int *n);
/* [local nelem] 1 where this rank owns the element (its lowest-numbered node): each element of
 * the global mesh is owned by exactly one rank -- the one whose copy goes into an output frame */
const int *des_part_elem_owned(const des_part *p, int *n);
const int *des_part_node_ranges(const des_part *p, int *n);

const char *des_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
