/* des_params.h -- plain-data view of the reference's run parameters, mesh topology
 * and field arrays, as they cross the C-ABI of the MI355X explicit time-stepper.
 *
 * The reference (GeoFLAC/DynEarthSol) has no FFI layer: its hot path is a set of free
 * C++ functions reading `const Param&` / `Variables&` (fields.hpp:4-18, rheology.hpp:4-13,
 * geometry.hpp:7-21,96-111, bc.hpp:4-17).  The structs below are a POD copy of exactly the
 * members those functions read, so a maintainer can fill them from `Param`/`Variables`
 * without conversion.
 *
 * Array layout is the reference's own: Array2D<T,N> compiled -DSOA (array2d.hpp:410-425,
 * Makefile:831), i.e. component-major `a[d*n + i]`.  Every `double*`/`int*` below that
 * is documented "SoA [N][n]" uses that layout.
 */
#ifndef DES_PARAMS_H
#define DES_PARAMS_H

#ifdef __cplusplus
extern "C" {
#endif

#define DES_MAX_MAT     16   /* max. number of material types carried in des_params */
#define DES_NBDRY       10   /* constants.hpp:27-38  (x0,x1,y0,y1,z0,z1,n0..n3)      */
#define DES_NBDRY_SIDE   6
#define DES_MAX_PERIOD   8   /* max. entries of bc.vbc_period_x{0,1}_* carried in des_params   */

/* rheology bit flags, matprops.hpp:88-97 */
enum {
    DES_RH_ELASTIC = 1, DES_RH_VISCOUS = 2, DES_RH_PLASTIC = 4,
    DES_RH_MAXWELL = 3, DES_RH_EP = 5, DES_RH_EVP = 7
};

/* Return codes of the C-ABI; the values are the reference's ExitCode categories
 * (utils.hpp:20-55) so a caller can forward them to die(). */
enum {
    DES_OK = 0,
    DES_ERR_CONFIG_VALUE = 11,     /* EXIT_CONFIG_VALUE: unknown option value          */
    DES_ERR_UNSUPPORTED_DIM = 30,  /* EXIT_UNSUPPORTED_DIM                              */
    DES_ERR_UNSUPPORTED = 31,      /* EXIT_UNSUPPORTED_LIB: feature/HIP device missing  */
    DES_ERR_RUNTIME_NAN = 50,      /* EXIT_RUNTIME_NAN: dt<=0 or NaN in state           */
    DES_ERR_RESOURCE = 52,         /* EXIT_RUNTIME_RESOURCE: device allocation failed   */
    DES_ERR_INTERNAL = 60          /* EXIT_INTERNAL_ASSERT: bad handle / argument       */
};

/* POD copy of the Param members the hot path reads (parameters.hpp:206-473) plus the
 * run constants main() derives once (dynearthsol.cxx:55-124, 207). */
typedef struct des_params {
    int ndims;                      /* 3: the THREED build (tets); 2: the 2-D build (triangles; coordinates
                                     * {x, z}, tensors {XX, ZZ, XZ}, constants.hpp:12-25)          */
    int nmat;                       /* mat.nmat                                                     */
    int rheol_type;                 /* mat.rheol_type, DES_RH_*                                     */
    int mattype_ref;                /* mat.mattype_ref                                              */

    /* control.* */
    double gravity;
    double inertial_scaling;
    double damping_factor;
    double dt_fraction;
    double fixed_dt;
    double characteristic_speed;
    double surface_diffusivity;
    double surf_base_level;
    int damping_option;
    int ref_pressure_option;
    int surface_process_option;     /* 0 or 1 (simple_diffusion, bc.cxx:916-1112)                   */
    int is_quasi_static;
    int has_thermal_diffusion;
    int is_using_mixed_stress;
    int has_moving_mesh;
    int quality_check_step_interval; /* mesh.quality_check_step_interval (bc.cxx:1830: dhacc reset) */

    /* bc.* */
    double surface_temperature;
    double winkler_delta_rho;
    double elastic_foundation_constant;
    double sea_water_density;
    double vbc_val_z1_loading_period;
    int has_winkler_foundation;
    int has_elastic_foundation;
    int has_water_loading;
    int is_outputting_averaged_fields; /* sim.*: Output::average_fields runs every step (output.cxx:327-370) */
    int vbc_types[DES_NBDRY];       /* Variables::vbc_types  (dynearthsol.cxx:63-72)               */
    double vbc_values[DES_NBDRY];   /* Variables::vbc_values (dynearthsol.cxx:74-83)               */
    double vbc_val_l[4];            /* bc.vbc_val_{x0,x1,y0,y1}_l (lateral shear, type 6)          */
    int stress_bc_types[DES_NBDRY_SIDE];
    double stress_bc_values[DES_NBDRY_SIDE];

    /* mesh.* */
    double xlength, ylength, zlength;

    /* mat.* scalars */
    double visc_min, visc_max, tension_max, therm_diff_max;

    /* mat.* per-material lists, already broadcast to nmat entries (input.cxx:983-989) */
    double rho0[DES_MAX_MAT];
    double alpha[DES_MAX_MAT];
    double bulk_modulus[DES_MAX_MAT];
    double shear_modulus[DES_MAX_MAT];
    double visc_exponent[DES_MAX_MAT];
    double visc_coefficient[DES_MAX_MAT];
    double visc_activation_energy[DES_MAX_MAT];
    double visc_activation_volume[DES_MAX_MAT];
    double heat_capacity[DES_MAX_MAT];
    double therm_cond[DES_MAX_MAT];
    double pls0[DES_MAX_MAT], pls1[DES_MAX_MAT];
    double cohesion0[DES_MAX_MAT], cohesion1[DES_MAX_MAT];
    double friction_angle0[DES_MAX_MAT], friction_angle1[DES_MAX_MAT];
    double dilation_angle0[DES_MAX_MAT], dilation_angle1[DES_MAX_MAT];
    double porosity[DES_MAX_MAT];

    /* run constants derived once by the driver */
    double max_vbc_val;             /* Variables::max_vbc_val   (dynearthsol.cxx:55-59)            */
    double compensation_pressure;   /* Variables::compensation_pressure (ic.cxx:361)               */

    /* control.has_PT: the pseudo-transient loop inside a step (dynearthsol.cxx:803-864) */
    int has_PT;
    int PT_max_iter;
    double PT_relative_tolerance;

    /* read by the 2-D build only (the reference's !THREED branches) */
    int is_plane_strain;            /* mat.is_plane_strain: elasto_plastic2d (rheology.cxx:486-701) + stressyy */
    int mattype_oceanic_crust;      /* mat.mattype_oceanic_crust (surface_plstrain_diffusion, bc.cxx:1633-1653) */
    int num_vbc_period_x0, num_vbc_period_x1;               /* bc.cxx:247-249 */
    double vbc_period_x0_time_in_yr[DES_MAX_PERIOD], vbc_period_x0_ratio[DES_MAX_PERIOD];
    double vbc_period_x1_time_in_yr[DES_MAX_PERIOD], vbc_period_x1_ratio[DES_MAX_PERIOD];
    double vbc_vertical_div_x0[4], vbc_vertical_div_x1[4];     /* Variables::vbc_vertical_div_x? (dynearthsol.cxx:85-92)   */
    double vbc_vertical_ratio_x0[4], vbc_vertical_ratio_x1[4]; /* Variables::vbc_vertical_ratio_x? (dynearthsol.cxx:94-101) */
    double bottom_shear_zone_thickness;                        /* bc.cxx:446-451 */
    double surf_diff_ratio_terrig, surf_diff_ratio_marine;     /* surfinfo.diff_ratio_* (mesh.cxx:3063-3064, bc.cxx:1098-1104) */
} des_params;

/* Mesh topology as the reference builds it once per (re)mesh (mesh.cxx:2837-3329,
 * bc.cxx:94-224).  All pointers are host pointers, read during create() only. */
typedef struct des_mesh {
    int nnode, nelem;
    const int *connectivity;        /* conn_t, SoA [4][nelem] ([3][nelem] in 2-D: every "4" / "3" below is
                                     * NODES_PER_ELEM / NDIMS of the build, constants.hpp:12-25)    */
    /* Support CSR (parameters.hpp:585-610): node n owns [idx[n], idx[n+1]) */
    const int *support_idx;         /* [nnode+1] */
    const int *support_arr;         /* [4*nelem] element ids, ascending per node                    */
    const int *support_lidx;        /* [4*nelem] local node number inside that element              */
    const unsigned *bcflag;         /* [nnode] boundary bit flags (constants.hpp:41-55)             */
    int nbfacets[DES_NBDRY];        /* Variables::bfacets[i]->size()                                */
    const int *bfacet_elem[DES_NBDRY];   /* .first  of each pair                                    */
    const int *bfacet_facet[DES_NBDRY];  /* .second of each pair                                    */
    int nbnodes[DES_NBDRY];
    const int *bnodes[DES_NBDRY];
    const double *bnormals;         /* array_t(nbdrytypes), SoA [3][10]                             */
    const double *edge_vec;         /* Variables::edge_vec, [nedge*3]                               */
    int nedge;
    int edge_slot[DES_NBDRY * DES_NBDRY];
    /* surface info (parameters.hpp:612-662, mesh.cxx:3031-3103, 2882-2933) */
    int ntop, etop, ntop_elems;
    const int *top_nodes;           /* [ntop]  surfinfo.top_nodes                                   */
    const int *elem_and_nodes;      /* segment_t(etop), SoA [3][etop]: surface-local node numbers   */
    const int *connectivity_surface;/* conn_t(etop), SoA [4][etop] (3 used)                         */
    const int *support_surf_idx;    /* [ntop+1] */
    const int *support_surf_arr;    /* top-facet ids per surface node                               */
    const int *top_elems;           /* [ntop_elems] Variables::top_elems                            */
    /* Layout hints (optional; the reference has no counterpart).  With `coord` -- the initial
     * coordinates, SoA [3][nnode] -- the engine stores its arrays in a space-filling-curve order
     * of its own (nodes within the three id ranges below, elements by centroid) so that a
     * workgroup's nodes / elements are neighbours in space; every array crossing the C-ABI stays
     * in the caller's numbering and every list keeps the caller's order.  NULL: caller's order. */
    const double *coord;
    int owned_begin, owned_end;     /* the des_halo range this mesh will be given (0, nnode if none) */
} des_mesh;

/* Domain decomposition (new: the reference is single-process).  A rank OWNS a
 * contiguous id range of the renumbered (x-sorted, mesh.cxx:2742-2766) global nodes.  Its local
 * mesh is that slab plus a GHOST REGION of `nlayers` element layers: layer 0 = every element
 * touching an owned node, layer k+1 = every further element touching a node of layer k; local
 * numbering in ascending global order, so every complete element patch is summed in the same
 * order as on one GPU (bit-identical results).
 *
 * One step needs the nodal state of a whole patch three times in a row (update_temperature /
 * dvoldt, NMD_stress, update_force) plus once for surface diffusion, so with four layers a rank
 * computes the step of its owned nodes WITHOUT any exchange in between -- redundantly on the
 * ghost region, whose outer layers go stale (about 1 % of the work at 1M tets per rank) -- and
 * the whole ghost region is refreshed ONCE per step: nodal {x,y,z,vx,vy,vz,T,dh} of every ghost
 * node from its owner, {stress, strain, plstrain} of the elements in the two outer layers from
 * the rank that owns their lowest-numbered node.  xGMI is latency-, not bandwidth-bound at
 * these sizes: one 0.5-MB message per neighbour instead of four 40-KB ones. */
typedef struct des_halo {
    int owned_begin, owned_end;      /* owned nodes = local ids [owned_begin, owned_end)             */
    int nlayers;                     /* element layers of the ghost region (4; 3 without diffusion)  */
    int nnbr;                        /* neighbour ranks                                             */
    const int *nbr_rank;             /* [nnbr]                                                      */
    const int *send_ptr;             /* [nnbr+1] offsets into send_idx                              */
    const int *send_idx;             /* local ids of OWNED nodes each neighbour holds as ghosts     */
    const int *recv_ptr;             /* [nnbr+1] offsets into recv_idx                              */
    const int *recv_idx;             /* local ids of GHOST nodes owned by each neighbour, ascending */
    const int *esend_ptr, *esend_idx;/* local ids of elements whose state each neighbour needs      */
    const int *erecv_ptr, *erecv_idx;/* local ids of stale-layer elements, by the rank that sends   */
    int owned_global_begin;          /* global id of the first owned node (a multiple of des_res_block(nnode_global)) */
} des_halo;

/* The residual of calculate_residual_force (fields.cxx:700-722) where a DECISION hangs on it -- the pseudo-transient loop's
 * convergence test (dynearthsol.cxx:803-864) and initial_body_force_adjustment's -- is summed in ONE association whatever
 * the partition (the reference's OpenMP reduction leaves the order open): block b = the global node ids [B b, B b + B),
 * P_b = the nodes' terms ((fr_0^2 / num + fr_1^2 / num) + fr_2^2 / num) added one after the other in ascending id from 0.0;
 * the squared residual = the P_b reduced in a fixed shape over the GLOBAL block array (256 strided serial sums, then a
 * pairwise tree).  B = des_res_block(nnode_global): 64, or 1 on meshes too small to be cut at multiples of 64.  Slabs are
 * cut at multiples of B (des_host_partition), so every block belongs to one rank, and engine and oracle use the same
 * shape: the same bits, the same decision, on 1 or N ranks. */
#define DES_RES_BLOCK 64
static inline int des_res_block(int nnode_global) { return nnode_global >= 64 * DES_RES_BLOCK ? DES_RES_BLOCK : 1; }

/* The exchange of a step (after the surface heights are committed, before the end-of-step
 * geometry pass): widths in doubles per node / per element. */
#define DES_X_NODE_WIDTH 8           /* x, y, z, vx, vy, vz, T, dh                                  */
#define DES_X_ELEM_WIDTH 13          /* stress[6], strain[6], plstrain                              */
#define DES_X_NODE_WIDTH_2D 6        /* 2-D models: x, z, vx, vz, T, dh                             */
#define DES_X_ELEM_WIDTH_2D 8        /* stress[3], strain[3], plstrain, stressyy                    */

/* Field ids for upload/download.  "E" = per element, "N" = per node. */
enum des_field {
    DES_F_COORD = 0,        /* N  array_t   SoA [3][nnode]  */
    DES_F_VEL,              /* N  array_t                   */
    DES_F_FORCE,            /* N  array_t                   */
    DES_F_FORCE_RESIDUAL,   /* N  array_t                   */
    DES_F_COORD0,           /* N  array_t                   */
    DES_F_TEMPERATURE,      /* N  double_vec                */
    DES_F_VOLUME_N,         /* N  double_vec                */
    DES_F_MASS,             /* N  double_vec                */
    DES_F_TMASS,            /* N  double_vec                */
    DES_F_DHACC,            /* N  surfinfo.dhacc            */
    DES_F_STRESS,           /* E  tensor_t  SoA [6][nelem]  */
    DES_F_STRAIN,           /* E  tensor_t                  */
    DES_F_STRAIN_RATE,      /* E  tensor_t                  */
    DES_F_PLSTRAIN,         /* E  double_vec                */
    DES_F_DELTA_PLSTRAIN,   /* E  double_vec                */
    DES_F_VISCOSITY,        /* E  double_vec                */
    DES_F_VOLUME,           /* E  double_vec                */
    DES_F_VOLUME_OLD,       /* E  double_vec                */
    DES_F_DPRESSURE,        /* E  double_vec                */
    DES_F_EDVOLDT,          /* E  double_vec                */
    DES_F_RADIOGENIC,       /* E  double_vec radiogenic_source */
    DES_F_ELEMMARKERS,      /* E  int32 [nelem][nmat] flat copy of int_vec2D elemmarkers */
    DES_F_EDVACC_SURF,      /* surfinfo.edvacc_surf [etop]  */
    DES_F_DH,               /* surfinfo.dh [ntop]           */
    DES_F_NTMP,             /* N  double_vec Variables::ntmp (scratch, exposed for tests) */
    /* state of Output::average_fields (output.hpp:30-36), kept by the engine when
     * is_outputting_averaged_fields: */
    DES_F_STRESS_AVG,       /* E  tensor_t  running sum of stress over the averaging interval   */
    DES_F_DPLSTRAIN_AVG,    /* E  double_vec running sum of delta_plstrain                       */
    DES_F_STRAIN0,          /* E  tensor_t  strain at the first step of the interval             */
    DES_F_COORD_AVG0,       /* N  array_t   coordinates at the first step of the interval        */
    DES_F_STRESSYY,         /* E  double_vec Variables::stressyy (fields.cxx:75); 2-D engines only */
    DES_F_COUNT
};

/* Scalars the driver reads back (dynearthsol.cxx:773-774, 801, 893; bc.cxx:1825;
 * geometry.cxx:1609-1610). */
typedef struct des_scalars {
    double dt;
    double time;
    double l2_residual;
    double max_surf_vel;
    double max_global_vel_mag;
    double global_dt_min;
    long long steps;
    int status;                     /* DES_OK or DES_ERR_RUNTIME_NAN (dt <= 0)                      */
    int n_return_mapping;           /* local elements past the yield pre-filter (rheology.cxx:354-361) in
                                     * the last step, i.e. through dsyevh3 + the return mapping       */
    double avg_time0;               /* Output::time0: time at the first step of the averaging interval */
    long long n_pt_iterations;      /* iterations of the pseudo-transient loop taken by the steps of this call */
} des_scalars;

/* Reductions behind bad_mesh_quality (remeshing.cxx:2752-2866), so the driver can take the
 * remesh decision every mesh.quality_check_step_interval steps without downloading the mesh. */
typedef struct des_quality {
    int small_elem;         /* first element with volume < smallest_vol, -1 if none (code 3)      */
    int bottom_node;        /* first bottom node with |z - bottom| > bottom_dist, -1 if none (2)   */
    int worst_elem;         /* first element attaining the minimum elem_quality                    */
    int pad_;
    double worst_quality;   /* min over elements of elem_quality (geometry.cxx:1873-1909)          */
} des_quality;

#ifdef __cplusplus
}
#endif
#endif
