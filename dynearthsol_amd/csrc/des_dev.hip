// des_dev.hip -- MI355X (gfx950) engine behind include/des_dev.h.
//
// One explicit time step of the reference (dynearthsol.cxx:768-894) is ~24 OpenMP/OpenACC
// loops with five element->node barriers.  Here it is six streaming passes over HBM plus
// O(surface) kernels, all state resident on the device:
//
//   E1  elements  [end of step t]   compute_volume, compute_mass (element part), rotate_stress,
//                                   compute_dt reduction (every 10th step)
//                 [start of t+1]    update_temperature (element part), update_strain_rate,
//                                   compute_dvoldt (element part)
//   N1  nodes     compute_mass gather, update_temperature gather + update, compute_dvoldt gather
//   E2  elements  compute_edvoldt, update_stress, NMD_stress (element part)
//   N2  nodes     NMD_stress gather
//   E3  elements  NMD_stress apply, update_force (element part)
//   N3  nodes     update_force gather, apply_stress_bcs, apply_damping, update_velocity,
//                 calculate_residual_force (partials), apply_vbcs, update_coordinate
//   S*  surface   surface_processes (simple_diffusion, correct_surface_element)
//
// Node assembly is a deterministic gather over the reference's CSR support graph in
// ascending element order (fields.cxx:659-676): no atomics on the data path, and the
// same summation order as the CPU build.
//
// HBM layout (DESIGN.md "Data layout"): element fields are SoA planes a[c*ne + e] so a
// wavefront of 64 consecutive elements reads 512 contiguous bytes per plane; nodal fields
// that elements gather are packed per node into 32-byte records {x,y,z,T} and
// {vx,vy,vz,mass} so that one gather is one aligned 32-byte sector; per-incidence
// temporaries are element-major records so a node reads 24-32 contiguous bytes per
// incident element.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "des_dev.h"
#include "des_dev2d.hpp"
#include "engine/env.hpp"
#include "engine/selfcheck.hpp"
#define DES_LIBM_LDS_TABLES 1     // kernels that call deslibm:: stage its tables in LDS first
#define DES_LIBM_LDS_WAVES 4      // = DES_BLOCK / 64, one private copy per wavefront
#include "des_kernels.hpp"

using desk::d4;

// minimum waves per SIMD the register allocator must leave room for (occupancy knobs,
// chosen by measurement: profiles/README.md)
#ifndef DES_E1_WAVES
#define DES_E1_WAVES 2
#endif
#ifndef DES_E2_WAVES
#define DES_E2_WAVES 2
#endif
#ifndef DES_E2_WAVES_FAST
#define DES_E2_WAVES_FAST 3       // first pass of the stress update without the return mapping: 168 VGPRs
#endif
#ifndef DES_E3_WAVES
#define DES_E3_WAVES 2
#endif

namespace {

thread_local std::string g_last_error;

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    g_last_error = std::string(#call) + ": " + hipGetErrorString(e_); return DES_ERR_RESOURCE; } } while (0)

const int NODE_OF_FACET_H[4][3] = {{1,2,3},{0,3,2},{0,1,3},{0,2,1}};
__device__ const int NODE_OF_FACET_D[4][3] = {{1,2,3},{0,3,2},{0,1,3},{0,2,1}};

// device-resident clock and reduction scratch
struct DevClock {
    double dt, time, l2_residual, max_surf_vel, max_global_vel_mag, global_dt_min;
    // compute_dt reduction slots (geometry.cxx:1490-1503)
    double r_minl, r_dt_maxwell, r_dt_diffusion, r_global_dt_min, r_max_vem;
    double maxdh, l2_sum;
    long long steps;
    int status;
    int iso;                 // inside isostasy_adjustment (des_dev_set_isostasy)
    int pt;                  // inside the pseudo-transient loop of a step (Param::control.PT_jump)
    int no_neumann;          // inside initial_body_force_adjustment: apply_stress_bcs_neumann is held back (fields.cxx:690)
    double avg_time0;        // Output::time0 (output.cxx:332)
    int n_defer;             // elements the first stress pass of this step handed to E2_return_mapping
    int pad;
    double dt_prev;          // the dt k_dt_finalize replaced: what a deferred rotate_stress of the step before ran with
};

// INIT: C part without rotate_stress; AVG: Output::average_fields on the final stress of the step
// DEFER (with C | A): rotate_stress of this step (and the pending NMD increment) is left to the next
// step's stress update, which reads stress / strain anyway -- E1 stores the three spin components only
// VOLX (with DT alone): the element's volume is formed from the coordinates instead of read -- the compute_dt
// reduction of a step whose end-of-step pass the next stress update does (E2<GEO>)
enum { MODE_A = 1, MODE_C = 2, MODE_DT = 4, MODE_INIT = 8, MODE_NOREC = 16, MODE_DEFER = 32, MODE_VOLX = 64 };

enum KernelId { K_E1, K_N1, K_E2, K_E2R, K_N2, K_E3, K_N3, K_S2, K_S3,
                K_DTFIN, K_MISC, K_EXCH, K_EN3, K_EN1, K_EN2, K_E2G, K_COUNT };
const char *kKernelNames[K_COUNT] = {
    "E1_geom_rotate_strainrate", "N1_mass_temperature_dvoldt", "E2_update_stress", "E2_return_mapping", "N2_nmd_gather",
    "E3_nmd_force", "N3_force_velocity_coord", "S2_surface_diffusion",
    "S3_edvacc_step_finalize", "dt_finalize", "misc", "ghost_exchange", "EN3_force_nodes", "EN1_mass_temperature_dvoldt", "EN2_nmd_gather",
    "E2G_geom_rotate_update_stress" };

struct ProfRec { int k; hipEvent_t a, b; };

// device copies of host arrays for the diagnostic entries; freed by the destructor
struct DiagBuf {
    std::vector<void *> ptrs;
    int rc = DES_OK;
    template <typename T> T *in(const T *host, size_t count) {
        T *d = out<T>(count);
        if (d && host && hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) fail("hipMemcpy");
        return d;
    }
    template <typename T> T *out(size_t count) {
        void *d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) { fail("hipMalloc"); return nullptr; }
        ptrs.push_back(d);
        return (T *)d;
    }
    template <typename T> void back(T *host, const T *dev, size_t count) {
        if (rc == DES_OK && host && hipMemcpy(host, dev, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) fail("hipMemcpy");
    }
    void fail(const char *what) { if (rc == DES_OK) { rc = DES_ERR_RESOURCE; g_last_error = what; } }
    ~DiagBuf() { for (void *p : ptrs) hipFree(p); }
};

} // namespace

struct des_dev {
    des2d::Engine *d2;       // a 2-D (triangle) model lives in the engine of des_dev2d.hip; everything below is then unused
    int device;
    int n_cu;                // compute units of the device (launch shapes by size: engine/launch.hpp, e2_three_waves)
    int portable_libm;       // DES_LIBM=portable: des_libm.hpp instead of ocml in the stress update
    des_params p;
    int nn, ne, nmat;
    hipStream_t stream;
    hipEvent_t ev0, ev1;

    des_params *d_p;
    desk::ViscTerms *d_vt;
    DevClock *d_clk;
    DevClock *h_clk;         // pinned mirror

    // topology
    int4 *conn;
    int *sup_idx, *sup_pack;             // pack = elem*4 + local node
    unsigned *bcflag;
    // nodal
    d4 *xt, *vm;                          // {x,y,z,T}, {vx,vy,vz,mass}
    double *ntmp, *volume_n, *tmass, *ymass, *force, *fres, *coord0, *dhacc;
    // Output::average_fields state (only allocated when is_outputting_averaged_fields)
    double *stress_avg, *dplstrain_avg, *strain0, *coord_avg0;
    // element
    double *stress, *strain, *strain_rate, *plstrain, *delta_plstrain, *viscosity, *volume,
           *volume_old, *dpressure, *radiogenic;
    int *markers;
    int *defer_list;                      // [ne] elements set aside by the first stress pass of the step
    bool use_graph;                       // DES_GRAPH=1
    hipGraphExec_t graph_exec[2];
    bool graph_two_pass[2];
    hipGraphExec_t pgraph_exec[6];        // the fused (patch) step: plain, with compute_dt, after compute_dt;
    bool pgraph_two_pass[6];              // x which of the two coordinate buffers is the current one
    int e2_defer;                         // DES_E2_DEFER: 0 one pass, 1 two passes, 2 (default) chosen per call
    bool verbose = false, said_pipe = false;  // DES_PATCH_VERBOSE at create: the engine says on stderr which shapes it picked
    int e2_pipe_mode = -1, e2_w3_mode = -1;   // DES_E2_PIPE / DES_E2_W3 at create: 0 / 1 pinned, -1 chosen by size (engine/launch.hpp)
    bool e2_two_pass;                     // the current choice
    int *mono;                            // [ne] (material << 16) | count of single-material elements, else -1
    double *ptab;                         // [nmat][DES_PTAB_CNT][5] property means of single-material elements
    double *pptab;                        // [nmat][DES_PPTAB_CNT][3][5] plastic_props of single-material elements by weakening regime
    unsigned char *topflag;               // element touches the top surface (Variables::top_elems)
    double *props;                        // [5][ne] bulkm, shearm, phi, cp, k  (nmat > 1 only)
    // temporaries
    d4 *mrec, *ttmp;                      // {vol, m, tm, dvol}, thermal tr[4]
    double *etmp2, *ftmp;                 // dp*vol ; force tr [ne][4][3]
    double *res_part;                     // per-block partial sums of the residual
    int n3_blocks;
    int res_nb;                           // residual partials the last N3 / EN3 launch wrote
    int npb;                              // nodes per node-kernel workgroup (choose_npb)
    // node-block patches: EN3 replaces E3 + N3 (passes/en3.hpp, engine/patch.hpp)
    bool patch;
    int patch_npb, patch_nb, patch_max_inc, patch_max_pn, patch_max_pe, patch_threads;
    bool patch_n1;                        // EN1 replaces N1 inside multi-step calls (passes/en1.hpp)
    int *pe_ptr, *pn_ptr, *pn_id;
    ulonglong2 *pe_pack;                  // (elem, local ids, slots) in one 16-byte record, engine/patch.hpp
    double *ddp;                          // [ne] NMD increment of the stress diagonal, applied by the next E1
    double *spin;                         // [3][ne] w3, w4, w5 of a deferred rotate_stress (E1<DEFER> -> next E2)
    bool defer_rot;                       // DES_DEFER_ROT != 0 (default on): fused end-of-step passes defer the rotation
    bool rot_pending, rot_prev_dt;        // the next E2 applies it; with the dt of before the last k_dt_finalize
    double *dt_part;                      // compute_dt partials, [5][dt_part_cap]: one slot per E1<MODE_DT> workgroup
    int dt_part_cap, dt_parts_used;       // ... slots filled since the last reduction (k_dt_finalize / k_dt_pack)
    bool elide_ok;                        // DES_E2_ELIDE != 0
    bool e2_not_last;                     // this step is not the last of its call (its end-of-step pass rides in the next stress update)
    bool e2_elide;                        // this step is not the last of its call: E2<GEO> skips the output-only stores
    bool e2geo_next;                      // the next E2 does what the skipped end-of-step pass would have done (E2<GEO>)
    bool e2_fresh;                        // ... or: the next E2 is E2<GEO> on a finished state (engine/launch.hpp: fresh_ok)
    bool finished;                        // the state is what the last des_dev_step call left: nothing has touched it since
    bool fresh_on;                        // DES_FRESH != 0 (read at create)
    d4 *xt_alt;                           // the other buffer of the {x,y,z,T} pair (EN3 writes it, then they swap)
    // stress-bc lists
    int nbcf;                             // facets with a stress bc (incl. neumann)
    int *bcf_elem, *bcf_facet, *bcf_kind; // kind: 0 winkler, 1 water, 2 side wall, 3+d neumann dir d
    double *bcf_val;                      // neumann value per facet
    double *bcf_tmp;                      // [nbcf][9]
    int *bcn_idx, *bcn_ent;               // per-node CSR of entries (facet*3+l)*2 + is_neumann
    unsigned bc_mask;                     // bcflag bits that have any entry
    // surface
    int ntop, etop, ntop_elems;
    int *top_nodes, *ean, *conn_surf, *ssup_idx, *ssup_arr, *ssup_nodes;
    double *dh, *edvacc, *znew;           // znew: surface heights between k_s2 and their commit
    // bnormals / edges for slanted boundaries
    double *bnormals, *edge_vec; int *edge_slot;

    // domain decomposition (des_halo): owned nodes [o0, o1), halo lists, exchange buffers
    int o0, o1, nn_global;
    // the partition-independent residual of the pseudo-transient loop (des_params.h: DES_RES_BLOCK): global id of the first
    // owned node, this rank's first block / block count, the global block count; res_gidx[i] = engine index of the i-th
    // owned node in ascending global id; res_blocks = the GLOBAL block array (own part written here, the rest received)
    int g0, res_b0, res_nb_own, res_nb_global;
    int *res_gidx;
    double *res_blocks;
    int nnbr;
    std::vector<int> nbr_rank, send_ptr, recv_ptr;         // host copies of the list offsets
    int *d_send_idx, *d_recv_idx;
    double *d_sendbuf, *d_recvbuf;                         // one message per neighbour: node records, then element records
    double *d_red;                                         // 8 doubles: dt partials / scalar reductions
    double *dh_n;                                          // nodal copy of surfinfo.dh
    ncclComm_t comm;
    int comm_rank, comm_size;
    // overlapped schedule (DES_OVERLAP=1): side stream + fork / join events; interior elements
    // (every node owned) are [e_int0, e_int1) in the engine's order
    bool overlap;
    hipStream_t comm_stream;
    hipEvent_t ev_fork, ev_join;
    int e_int0, e_int1;
    // ... and the part of them (and of the node blocks) that lies deep inside the slab: what the passes compute there reads
    // nothing the exchange writes (engine/order.hpp: DES_DEEP_DIST).  Empty ranges: the slab is too thin.
    int e_deep0, e_deep1, n_deep0, n_deep1;
    bool join_pending;         // the exchange of the last step is still running on the side stream (step_front joins)
    // element part of the exchange lists and the record offsets inside the message buffers
    std::vector<int> esend_ptr, erecv_ptr;
    std::vector<long long> send_off, recv_off;             // [nnbr+1] message offsets (doubles) per neighbour
    int *d_esend_idx, *d_erecv_idx, *d_send_noff, *d_send_eoff, *d_recv_noff, *d_recv_eoff;
    // internal data order (des_mesh::coord hint): device index <-> caller's index; empty = identity
    std::vector<int> n_new2old, n_old2new, e_new2old, e_old2new;
    int *d_n_new2old, *d_e_new2old;
    bool markers_dirty;
    bool radiogenic_zero;                 // every heat source is +0.0 (the default): EN1 does not fetch them
    bool const_mass;                      // quasi-static, one material: nodal mass from volumes alone
    bool pending_c;                       // C part of the last step has been run (always true outside step())
    long long steps_host;
    bool iso;                 // des_dev_set_isostasy
    long long n_pt_iterations; // pseudo-transient iterations of the current des_dev_step call
    bool in_pt;                // host side of DevClock::pt
    // engines of ONE process as each other's neighbours (des_dev_step_group): the ghost state moves by
    // device-to-device copies between the message buffers instead of RCCL messages
    des_dev **group;           // [group_n], indexed by rank (des_halo::nbr_rank); null: no group
    int group_n, group_rank;
    hipEvent_t ev_packed, ev_taken;   // messages packed (my stream) / the neighbours' messages copied out and unpacked
    int *bperm;                // [patch_nb] launch order of the patch blocks (nullptr: as numbered)
    int *pb_top;               // [patch_nb] the block's patch holds a surface node (an int: EN1 reads it with a scalar load)
    int2 *tfan;                // [nn] {position in top_nodes | fan size << 27, start of the fan in ssup_nodes}, {-1, 0} below the surface (passes/en1.hpp)
    bool s2_defer;             // DES_S2_DEFER != 0 (read at create)
    bool s2_pending;           // the surface step of the last step has not run: the next EN1 does it (s2_defer_ok)
    bool edv_pending;
    bool s2_skipped;           // this step's S2 / S3 launches were left out (step_front -> step_back)          // ... and the next stress update its edvacc_surf part
    bool ddp_live;             // EN3 has left NMD increments in ddp[] that no pass has folded into the stress yet
    // profiling
    bool prof;
    std::vector<ProfRec> prof_recs;
    double prof_ms[K_COUNT]; long long prof_calls[K_COUNT];
};

// engine/order.hpp: how far from every ghost node a "deep" node lies (its patch = ring 1 and the surface fans of its
// patch nodes = ring 2 are then owned), and the id margin that keeps every node block (<= 128 nodes, DES_PATCH) holding
// a node of a deep element inside the deep range
#define DES_DEEP_DIST 3
#define DES_DEEP_MARGIN 128

namespace des_hip {

#include "passes/common.hpp"
#include "passes/surface.hpp"
#include "passes/e1.hpp"
#include "passes/node_gather.hpp"
#include "passes/n1.hpp"
#include "passes/e2.hpp"
#include "passes/n2.hpp"
#include "passes/e3.hpp"
#include "passes/n3.hpp"
#include "passes/en3.hpp"
#include "passes/en1.hpp"
#include "passes/en2.hpp"
#include "passes/small_kernels.hpp"
#include "engine/patch.hpp"
#include "engine/launch.hpp"
#include "engine/residual.hpp"
#include "engine/exchange.hpp"
#include "engine/order.hpp"

} // namespace des_hip

using namespace des_hip;

// a handle that holds a 2-D engine forwards the call (and keeps the engine's error text)
#define D2_FORWARD(h, call) do { if ((h) && (h)->d2) { int rc2_ = des2d::call; if (rc2_) g_last_error = des2d::last_error((h)->d2); return rc2_; } } while (0)
#define D2_REFUSE(h, what) do { if ((h) && (h)->d2) { g_last_error = what " is offered for 3-D models only"; return DES_ERR_UNSUPPORTED_DIM; } } while (0)

// =====================================================================================
// C-ABI
// =====================================================================================
extern "C" {

const char *des_dev_last_error(void) { return g_last_error.c_str(); }

int des_dev_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void des_dev_destroy(des_dev *h)
{
    if (!h) return;
    if (h->d2) { des2d::destroy(h->d2); if (h->comm) ncclCommDestroy(h->comm); delete h; return; }
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm_stream) { hipStreamSynchronize(h->comm_stream); hipStreamDestroy(h->comm_stream); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ev_packed) hipEventDestroy(h->ev_packed);
    if (h->ev_taken) hipEventDestroy(h->ev_taken);
    delete[] h->group;
    if (h->comm) ncclCommDestroy(h->comm);
    for (ProfRec &r : h->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (hipGraphExec_t g : h->graph_exec) if (g) hipGraphExecDestroy(g);
    for (hipGraphExec_t g : h->pgraph_exec) if (g) hipGraphExecDestroy(g);
    void *ptrs[] = { h->pe_ptr, h->pn_ptr, h->pn_id, h->pe_pack, h->ddp, h->xt_alt, h->spin,
        h->d_p, h->d_vt, h->d_clk, h->conn, h->sup_idx, h->sup_pack, h->bcflag, h->xt, h->vm,
        h->ntmp, h->volume_n, h->tmass, h->ymass, h->force, h->fres, h->coord0, h->dhacc, h->dh_n, h->d_red, h->dt_part, h->d_n_new2old, h->d_e_new2old,
        h->d_send_idx, h->d_recv_idx, h->d_sendbuf, h->d_recvbuf, h->d_esend_idx, h->d_erecv_idx, h->d_send_noff,
        h->d_send_eoff, h->d_recv_noff, h->d_recv_eoff, h->stress, h->strain,
        h->strain_rate, h->plstrain, h->delta_plstrain, h->viscosity, h->volume, h->volume_old, h->dpressure,
        h->stress_avg, h->dplstrain_avg, h->strain0, h->coord_avg0,
        h->radiogenic, h->markers, h->props, h->mono, h->defer_list, h->ptab, h->pptab, h->mrec, h->ttmp, h->etmp2, h->ftmp, h->res_part, h->bcf_elem,
        h->bcf_facet, h->bcf_kind, h->bcf_val, h->bcf_tmp, h->bcn_idx, h->bcn_ent, h->top_nodes, h->ean,
        h->conn_surf, h->ssup_idx, h->ssup_arr, h->ssup_nodes, h->topflag, h->tfan, h->pb_top, h->bperm, h->dh, h->edvacc,
        h->znew, h->bnormals, h->edge_vec, h->edge_slot };
    for (void *q : ptrs) if (q) hipFree(q);
    if (h->h_clk) hipHostFree(h->h_clk);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

des_dev *des_dev_create(int device, const des_params *params, const des_mesh *mesh, int *err)
{
    int dummy; if (!err) err = &dummy;
    *err = DES_OK;
    if (!params || !mesh) { *err = DES_ERR_INTERNAL; g_last_error = "null argument"; return nullptr; }
    if (params->ndims == 2) {
        // the reference's 2-D build: its own engine behind the same entry points (des_dev2d.hpp)
        des2d::Engine *e2 = des2d::create(device, params, mesh, err, g_last_error);
        if (!e2) return nullptr;
        des_dev *h2 = new des_dev();
        h2->d2 = e2; h2->device = device; h2->p = *params; h2->nn = mesh->nnode; h2->ne = mesh->nelem; h2->nmat = params->nmat;
        return h2;
    }
    if (params->ndims != 3) { *err = DES_ERR_UNSUPPORTED_DIM; g_last_error = "ndims must be 2 or 3"; return nullptr; }
    if (params->nmat < 1 || params->nmat > DES_MAX_MAT) { *err = DES_ERR_CONFIG_VALUE; g_last_error = "bad nmat"; return nullptr; }
    switch (params->rheol_type) {
    case DES_RH_ELASTIC: case DES_RH_VISCOUS: case DES_RH_MAXWELL: case DES_RH_EP: case DES_RH_EVP: break;
    default: *err = DES_ERR_UNSUPPORTED; g_last_error = "rheology not offloaded"; return nullptr;
    }
    if (des_dev_device_count() <= device) { *err = DES_ERR_UNSUPPORTED; g_last_error = "no such HIP device"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *err = DES_ERR_UNSUPPORTED; g_last_error = "hipSetDevice failed"; return nullptr; }

    if (mesh->nelem < 1 || mesh->nnode < 4 || mesh->nelem >= (1 << 29) || mesh->nnode >= (1 << 27)) {
        // incidences are packed as elem*4 + local node in 32 bits (and 4*nelem indexes the CSR arrays); the stress
        // update addresses planes and 32-byte nodal records by 32-bit byte offsets (passes/common.hpp: pl_ld, rec_ld)
        *err = DES_ERR_RESOURCE; g_last_error = "mesh size outside 1 <= nelem < 2^29, nnode < 2^27"; return nullptr;
    }
    des_dev *h = new des_dev();       // value-initialised: every pointer/scalar member starts at 0
    h->device = device;
    {
        int ncu = 0;
        h->n_cu = (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) ? ncu : 256;
    }
    h->p = *params;
    { const char *fe = des_env::get("DES_FRESH"); h->fresh_on = !(fe && fe[0] == '0'); }
    {
        const char *gr = des_env::get("DES_GRAPH");
        h->use_graph = gr && gr[0] == '1';
        const char *ov = des_env::get("DES_OVERLAP");
        h->overlap = ov && ov[0] == '1';
        h->verbose = des_env::get("DES_PATCH_VERBOSE") != nullptr;
        { const char *e = des_env::get("DES_E2_PIPE"); h->e2_pipe_mode = (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1; }
        { const char *e = des_env::get("DES_E2_W3"); h->e2_w3_mode = (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1; }
        const char *e2d = des_env::get("DES_E2_DEFER");
        h->e2_defer = (e2d && (e2d[0] == '0' || e2d[0] == '1')) ? e2d[0] - '0' : 2;
        h->e2_two_pass = h->e2_defer != 0;
        // Default: des_libm.hpp, whose pow / exp return the bits of the C library the CPU reference
        // runs on (glibc, x86-64 with FMA) -- a model that does not yield then equals the CPU run
        // bit for bit (tests/test_gpu_headline.py).  DES_LIBM=ocml: ROCm's device libm, 1-2 ulp
        // away per call and ~2 % faster per step (E2 74 vs 81 us at 1M tets).
        const char *env = des_env::get("DES_LIBM");
        h->portable_libm = !env || std::strcmp(env, "portable") == 0;
        if (env && !h->portable_libm && std::strcmp(env, "ocml") != 0) {
            *err = DES_ERR_CONFIG_VALUE; g_last_error = "DES_LIBM must be 'ocml' or 'portable'"; delete h; return nullptr;
        }
    }
    PermMesh pm;
    {
        const char *env = des_env::get("DES_REORDER");
        if (mesh->coord && mesh->nnode > 0 && mesh->nelem > 0 && !(env && env[0] == '0')) {
            build_perm_mesh(mesh, pm);
            h->n_new2old.swap(pm.n_new2old); h->n_old2new.swap(pm.n_old2new);
            h->e_new2old.swap(pm.e_new2old); h->e_old2new.swap(pm.e_old2new);
            h->e_int0 = pm.e_int0; h->e_int1 = pm.e_int1;
            h->e_deep0 = pm.e_deep0; h->e_deep1 = pm.e_deep1; h->n_deep0 = pm.n_deep0; h->n_deep1 = pm.n_deep1;
            mesh = &pm.view;             // everything below builds the device state in the internal order
        } else {
            h->e_int0 = 0; h->e_int1 = 0;      // caller's order kept: no interior range known, no overlap
            h->e_deep0 = h->e_deep1 = h->n_deep0 = h->n_deep1 = 0;
        }
    }
    const int nn = h->nn = mesh->nnode, ne = h->ne = mesh->nelem, nmat = h->nmat = params->nmat;
    h->markers_dirty = true;
    h->radiogenic_zero = true;            // (the array starts zeroed; des_dev_upload looks at what it is given)
    h->pending_c = true;
    h->const_mass = params->is_quasi_static && params->nmat == 1;
    h->o0 = 0; h->o1 = nn; h->nn_global = nn; h->nnbr = 0; h->comm = nullptr; h->comm_rank = 0; h->comm_size = 1;
    h->g0 = 0; h->res_gidx = nullptr; h->res_blocks = nullptr;

#define CK(x) do { int rc_ = (x); if (rc_ != DES_OK) { *err = rc_; des_dev_destroy(h); return nullptr; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_last_error = std::string(#x) + ": " + hipGetErrorString(e_); \
                   *err = DES_ERR_RESOURCE; des_dev_destroy(h); return nullptr; } } while (0)
    HK(hipStreamCreate(&h->stream));
    HK(hipEventCreate(&h->ev0)); HK(hipEventCreate(&h->ev1));
    // (side stream + fork / join events of the overlapped schedule: always there, so that des_dev_set_overlap can switch
    //  schedules on a live engine)
    HK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    HK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HK(hipHostMalloc((void **)&h->h_clk, sizeof(DevClock)));

    CK(dev_alloc(h->d_p, 1)); CK(dev_alloc(h->d_vt, 1)); CK(dev_alloc(h->d_clk, 1));
    CK(dev_upload(h->d_p, params, 1, h->stream));
    {
        desk::ViscTerms vt;
        std::memset(&vt, 0, sizeof(vt));
        const double gas_constant = 8.3144;                       // matprops.cxx:237-250
        for (int m = 0; m < nmat; ++m) {
            vt.pow_edot[m] = 1 / params->visc_exponent[m] - 1;
            const double pow1 = -1 / params->visc_exponent[m];
            vt.coef_term[m] = std::pow(0.75 * params->visc_coefficient[m], pow1);
            vt.nR[m] = params->visc_exponent[m] * gas_constant;
        }
        CK(dev_upload(h->d_vt, &vt, 1, h->stream));
    }
    {
        DevClock c;
        std::memset(&c, 0, sizeof(c));
        c.r_minl = c.r_dt_maxwell = c.r_dt_diffusion = c.r_global_dt_min = DBL_MAX;
        *h->h_clk = c;
        CK(dev_upload(h->d_clk, &c, 1, h->stream));
    }

    // topology
    {
        std::vector<int4> c4((size_t)ne);
        for (int e = 0; e < ne; ++e)
            c4[e] = make_int4(mesh->connectivity[e], mesh->connectivity[(size_t)ne + e],
                              mesh->connectivity[(size_t)2*ne + e], mesh->connectivity[(size_t)3*ne + e]);
        CK(dev_alloc(h->conn, (size_t)ne)); CK(dev_upload(h->conn, c4.data(), (size_t)ne, h->stream));
        std::vector<int> pack((size_t)4*ne);
        for (size_t k = 0; k < pack.size(); ++k) pack[k] = mesh->support_arr[k] * 4 + mesh->support_lidx[k];
        CK(dev_alloc(h->sup_idx, (size_t)nn + 1)); CK(dev_upload(h->sup_idx, mesh->support_idx, (size_t)nn + 1, h->stream));
        CK(dev_alloc(h->sup_pack, pack.size())); CK(dev_upload(h->sup_pack, pack.data(), pack.size(), h->stream));
        CK(dev_alloc(h->bcflag, (size_t)nn)); CK(dev_upload(h->bcflag, mesh->bcflag, (size_t)nn, h->stream));
        if (!h->n_new2old.empty()) {
            CK(dev_alloc(h->d_n_new2old, (size_t)nn)); CK(dev_upload(h->d_n_new2old, h->n_new2old.data(), (size_t)nn, h->stream));
            CK(dev_alloc(h->d_e_new2old, (size_t)ne)); CK(dev_upload(h->d_e_new2old, h->e_new2old.data(), (size_t)ne, h->stream));
        }
    }
    // node-block patches for EN3 (DES_PATCH=0: the classic pair E3 + N3); blocks of 64 nodes, or 32
    // where a block of 64 would not fit the kernel's LDS slots
    {
        const char *env = des_env::get("DES_PATCH");
        h->patch = false;
        if (!(env && env[0] == '0')) {
            PatchLists P;
            // DES_PATCH=<n>: n nodes per block; default: the largest of 64, 56, 48, 40, 32 whose biggest
            // block fits the three-workgroups-per-CU LDS shapes of both patch kernels (valence-32 nodes of
            // the regular 5-tet mesh make smaller blocks than a TetGen mesh of the same size), else 64 /
            // 32 / 16 with the full-size shapes
            int want = env ? std::atoi(env) : 0;
            if (want >= 16 && want <= 128 && want % 8 == 0)
                h->patch = build_patches(mesh, want, DES_PATCH_INC, DES_PATCH_PN, P);
            else {
                // (three workgroups per CU in both patch kernels: their dynamic LDS for this mesh's largest block <= 160 KiB / 3)
                const int fits[5] = {64, 56, 48, 40, 32};
                const size_t lds3 = 160 * 1024 / 3;
                for (int t = 0; t < 5 && !h->patch; ++t)
                    h->patch = build_patches(mesh, fits[t], DES_PATCH_INC, DES_PATCH_PN, P) && P.max_pe <= DES_PATCH_PE
                               && en1_lds_bytes(patch_cap(P.max_inc), patch_cap(P.max_pn), patch_cap(P.max_pe), h->const_mass) <= lds3
                               && en3_lds_bytes(patch_cap(P.max_inc), patch_cap(P.max_pn)) <= lds3;
                const int tries[3] = {64, 32, 16};
                for (int t = 0; t < 3 && !h->patch; ++t)
                    h->patch = build_patches(mesh, tries[t], DES_PATCH_INC, DES_PATCH_PN, P);
            }
            if (h->patch) {
                h->patch_npb = P.npb; h->patch_nb = P.nb; h->patch_max_inc = P.max_inc; h->patch_max_pn = P.max_pn; h->patch_max_pe = P.max_pe;
                if (des_env::get("DES_PATCH_VERBOSE"))
                    std::fprintf(stderr, "patches: %d nodes per block, %d blocks, max incidences %d, patch nodes %d, patch elements %d, "
                                 "elements listed %zu (%.2f x nelem)\n", P.npb, P.nb, P.max_inc, P.max_pn, P.max_pe, P.pe_elem.size(),
                                 (double)P.pe_elem.size() / ne);
                const char *pn1 = des_env::get("DES_PATCH_N1");
                h->patch_n1 = !(pn1 && pn1[0] == '0') && P.max_pe <= DES_PATCH_PE;
                const char *pt = des_env::get("DES_PATCH_THREADS");
                h->patch_threads = (pt && std::atoi(pt) == 256) ? 256 : 512;
                CK(dev_alloc(h->pe_ptr, P.pe_ptr.size())); CK(dev_upload(h->pe_ptr, P.pe_ptr.data(), P.pe_ptr.size(), h->stream));
                CK(dev_alloc(h->pe_pack, P.pe_pack.size())); CK(dev_upload(h->pe_pack, P.pe_pack.data(), P.pe_pack.size(), h->stream));
                CK(dev_alloc(h->pn_ptr, P.pn_ptr.size())); CK(dev_upload(h->pn_ptr, P.pn_ptr.data(), P.pn_ptr.size(), h->stream));
                {
                    // surface nodes among a patch's foreign nodes: bit 31 of their pn_id entry; blocks whose patch has any
                    // (EN1's deferred surface step, passes/en1.hpp)
                    std::vector<unsigned char> is_top((size_t)nn, 0); std::vector<int> pbt((size_t)P.nb, 0);
                    for (int i = 0; i < mesh->ntop; ++i) is_top[mesh->top_nodes[i]] = 1;
                    for (int b = 0; b < P.nb; ++b) {
                        for (int n = b * P.npb; n < std::min(nn, (b + 1) * P.npb); ++n) pbt[b] |= is_top[n];
                        for (int k = P.pn_ptr[b]; k < P.pn_ptr[b + 1]; ++k)
                            if (is_top[P.pn_id[k]]) { P.pn_id[k] |= (int)0x80000000u; pbt[b] = 1; }
                    }
                    CK(dev_alloc(h->pb_top, pbt.size())); CK(dev_upload(h->pb_top, pbt.data(), pbt.size(), h->stream));
                    // The surface blocks carry extra work (the deferred surface step in EN1: +5 us of the pass at 1M tets) and the
                    // Morton order packs them into half of the eight contiguous runs desk::logical_block hands the XCDs.  Launch
                    // order: run k = an eighth of the surface blocks, first, then its share of the others, both as numbered.
                    // (Which block does what does not change: the same bits.  DES_TOP_BALANCE=0: off.)
                    const char *be = des_env::get("DES_TOP_BALANCE");
                    if (P.nb >= 64 && !(be && be[0] == '0')) {
                        std::vector<int> T, O, perm;
                        for (int b = 0; b < P.nb; ++b) (pbt[b] ? T : O).push_back(b);
                        const int per = (P.nb + 7) / 8;
                        size_t ot = 0;
                        for (int k = 0; k < 8; ++k) {
                            const int want = std::max(0, std::min(per, P.nb - k * per));
                            const size_t t0 = T.size() * k / 8, t1 = T.size() * (k + 1) / 8;
                            int got = 0;
                            for (size_t t = t0; t < t1 && got < want; ++t, ++got) perm.push_back(T[t]);
                            for (; got < want && ot < O.size(); ++got) perm.push_back(O[ot++]);
                        }
                        std::vector<char> seen((size_t)P.nb, 0);
                        bool ok = (int)perm.size() == P.nb;
                        for (int b : perm) { if (seen[b]) ok = false; seen[b] = 1; }
                        if (ok) { CK(dev_alloc(h->bperm, perm.size())); CK(dev_upload(h->bperm, perm.data(), perm.size(), h->stream)); }
                    }
                }
                P.pn_id.push_back(0);                       // (a spare entry: EN1 / EN3 read pn_id without a branch, passes/en3.hpp)
                CK(dev_alloc(h->pn_id, P.pn_id.size())); CK(dev_upload(h->pn_id, P.pn_id.data(), P.pn_id.size(), h->stream));
                CK(dev_alloc(h->ddp, (size_t)ne)); CK(dev_alloc(h->xt_alt, (size_t)nn));
                HK(hipMemsetAsync(h->ddp, 0, 8*(size_t)ne, h->stream));
                HK(hipMemsetAsync(h->xt_alt, 0, sizeof(d4)*(size_t)nn, h->stream));
            }
        }
    }
    // fields
    CK(dev_alloc(h->xt, (size_t)nn)); CK(dev_alloc(h->vm, (size_t)nn));
    CK(dev_alloc(h->ntmp, (size_t)nn)); CK(dev_alloc(h->volume_n, (size_t)nn)); CK(dev_alloc(h->tmass, (size_t)nn));
    CK(dev_alloc(h->ymass, (size_t)nn)); CK(dev_alloc(h->force, (size_t)3*nn)); CK(dev_alloc(h->fres, (size_t)3*nn));
    CK(dev_alloc(h->coord0, (size_t)3*nn)); CK(dev_alloc(h->dhacc, (size_t)nn)); CK(dev_alloc(h->dh_n, (size_t)nn));
    CK(dev_alloc(h->d_red, 8));
    h->dt_part_cap = 2 * (((ne + DES_BLOCK - 1) / DES_BLOCK + 7) / 8 * 8) + 64;
    CK(dev_alloc(h->dt_part, (size_t)5 * h->dt_part_cap));
    CK(dev_alloc(h->stress, (size_t)6*ne)); CK(dev_alloc(h->strain, (size_t)6*ne)); CK(dev_alloc(h->strain_rate, (size_t)6*ne));
    CK(dev_alloc(h->plstrain, (size_t)ne)); CK(dev_alloc(h->delta_plstrain, (size_t)ne)); CK(dev_alloc(h->viscosity, (size_t)ne));
    CK(dev_alloc(h->volume, (size_t)ne)); CK(dev_alloc(h->volume_old, (size_t)ne)); CK(dev_alloc(h->dpressure, (size_t)ne));
    CK(dev_alloc(h->radiogenic, (size_t)ne)); CK(dev_alloc(h->markers, (size_t)ne * nmat));
    CK(dev_alloc(h->mono, (size_t)ne));
    CK(dev_alloc(h->defer_list, (size_t)ne));
    {
        // deferred rotate_stress (MODE_DEFER, passes/e1.hpp): on by default, DES_DEFER_ROT=0 keeps rotate_stress in E1
        const char *el = des_env::get("DES_E2_ELIDE");      // =0: every step stores every field
        h->elide_ok = !(el && el[0] == '0');
        const char *sd = des_env::get("DES_S2_DEFER");      // =0: every step launches its own S2 / S3
        h->s2_defer = !(sd && sd[0] == '0');
        const char *dr = des_env::get("DES_DEFER_ROT");
        h->defer_rot = !(dr && dr[0] == '0');
        if (h->defer_rot) { CK(dev_alloc(h->spin, (size_t)3*ne)); HK(hipMemsetAsync(h->spin, 0, 24*(size_t)ne, h->stream)); }
    }
    if (nmat > 1) {
        CK(dev_alloc(h->props, (size_t)5*ne));
        CK(dev_alloc(h->ptab, (size_t)nmat * DES_PTAB_CNT * 5));
        hipLaunchKernelGGL(k_ptab, dim3((nmat * DES_PTAB_CNT + 63) / 64), dim3(64), 0, h->stream, h->d_p, h->ptab);
    }
    if (params->rheol_type == DES_RH_EP || params->rheol_type == DES_RH_EVP) {
        // plastic_props outside the linear weakening range, by (material, marker count, regime): DES_PPTAB=0 switches it off
        const char *pp = des_env::get("DES_PPTAB");
        if (!(pp && pp[0] == '0')) {
            const int n = nmat * DES_PPTAB_CNT * 3;
            CK(dev_alloc(h->pptab, (size_t)n * 5));
            if (h->portable_libm) hipLaunchKernelGGL(k_pptab<desk::MathPortable>, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d_p, h->pptab);
            else                  hipLaunchKernelGGL(k_pptab<desk::MathOcml>, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d_p, h->pptab);
        }
    }
    CK(dev_alloc(h->mrec, (size_t)ne)); CK(dev_alloc(h->ttmp, (size_t)ne)); CK(dev_alloc(h->etmp2, (size_t)ne));
    CK(dev_alloc(h->ftmp, (size_t)12*ne));
    if (h->p.is_outputting_averaged_fields) {
        CK(dev_alloc(h->stress_avg, (size_t)6*ne)); CK(dev_alloc(h->strain0, (size_t)6*ne));
        CK(dev_alloc(h->dplstrain_avg, (size_t)ne)); CK(dev_alloc(h->coord_avg0, (size_t)3*nn));
        HK(hipMemsetAsync(h->stress_avg, 0, 48*(size_t)ne, h->stream)); HK(hipMemsetAsync(h->strain0, 0, 48*(size_t)ne, h->stream));
        HK(hipMemsetAsync(h->dplstrain_avg, 0, 8*(size_t)ne, h->stream)); HK(hipMemsetAsync(h->coord_avg0, 0, 24*(size_t)nn, h->stream));
    }
    choose_npb(h);
    h->n3_blocks = res_part_size(h);
    h->res_nb = 0;
    CK(dev_alloc(h->res_part, (size_t)h->n3_blocks));
    CK(build_residual_blocks(h));
    {
        struct { void *p; size_t bytes; } zero[] = {
            {h->xt, sizeof(d4)*(size_t)nn}, {h->vm, sizeof(d4)*(size_t)nn}, {h->ntmp, 8*(size_t)nn},
            {h->volume_n, 8*(size_t)nn}, {h->tmass, 8*(size_t)nn}, {h->ymass, 8*(size_t)nn},
            {h->force, 24*(size_t)nn}, {h->fres, 24*(size_t)nn}, {h->coord0, 24*(size_t)nn}, {h->dhacc, 8*(size_t)nn},
            {h->dh_n, 8*(size_t)nn}, {h->d_red, 64},
            {h->stress, 48*(size_t)ne}, {h->strain, 48*(size_t)ne}, {h->strain_rate, 48*(size_t)ne},
            {h->plstrain, 8*(size_t)ne}, {h->delta_plstrain, 8*(size_t)ne}, {h->volume, 8*(size_t)ne},
            {h->volume_old, 8*(size_t)ne}, {h->dpressure, 8*(size_t)ne}, {h->radiogenic, 8*(size_t)ne},
            {h->markers, 4*(size_t)ne*nmat}, {h->mrec, 32*(size_t)ne}, {h->ttmp, 32*(size_t)ne},
            {h->etmp2, 8*(size_t)ne}, {h->ftmp, 96*(size_t)ne}, {h->res_part, 8*(size_t)h->n3_blocks} };
        for (auto &z : zero) HK(hipMemsetAsync(z.p, 0, z.bytes, h->stream));
        std::vector<double> vmax((size_t)ne, params->visc_max);          // fields.cxx:110
        CK(dev_upload(h->viscosity, vmax.data(), (size_t)ne, h->stream));
    }

    // stress-bc facets and per-node entry lists, in the order apply_stress_bcs (bc.cxx:681-813)
    // and apply_stress_bcs_neumann (bc.cxx:838-907) visit them
    {
        std::vector<int> f_elem, f_facet, f_kind; std::vector<double> f_val;
        std::vector<std::vector<int> > node_ent((size_t)nn);
        unsigned mask = 0;
        std::vector<int> etmp_int((size_t)ne, -1);
        if (params->gravity != 0) {
            for (int i = 0; i < DES_NBDRY; i++) {
                const int t = params->vbc_types[i];
                if (t != 0 && t != 2 && t != 4) continue;
                if (i == 4 && !params->has_winkler_foundation) continue;
                if (i == 5 && !params->has_water_loading) continue;
                const int bound = mesh->nbfacets[i];
                const int offset = (int)f_elem.size();
                const int kind = (i == 4 && params->has_winkler_foundation) ? 0
                               : (i == 5 && params->has_water_loading) ? 1 : 2;
                for (int n = 0; n < bound; ++n) {
                    f_elem.push_back(mesh->bfacet_elem[i][n]); f_facet.push_back(mesh->bfacet_facet[i][n]);
                    f_kind.push_back(kind); f_val.push_back(0);
                    etmp_int[mesh->bfacet_elem[i][n]] = n;
                }
                for (int j = 0; j < mesh->nbnodes[i]; ++j) {
                    const int n = mesh->bnodes[i][j];
                    for (int k = mesh->support_idx[n]; k < mesh->support_idx[n+1]; ++k) {
                        const int e = mesh->support_arr[k];
                        const int ibound = etmp_int[e];
                        if (ibound < 0) continue;
                        const int f = mesh->bfacet_facet[i][ibound];
                        for (int l = 0; l < 3; ++l) {
                            if (n == mesh->connectivity[(size_t)NODE_OF_FACET_H[f][l]*ne + e]) {
                                node_ent[n].push_back((((offset + ibound) * 3 + l) << 1) | 0);
                                mask |= (1u << i);
                                break;
                            }
                        }
                    }
                }
                for (int n = 0; n < bound; ++n) etmp_int[mesh->bfacet_elem[i][n]] = -1;
            }
            if (params->has_elastic_foundation) mask |= (1u << 4);
        }
        for (int i = 0; i < 6; ++i) {
            const int t = params->stress_bc_types[i];
            if (t == 0) continue;
            if (t < 1 || t > 3) continue;
            for (int n = 0; n < mesh->nbfacets[i]; ++n) {
                const int e = mesh->bfacet_elem[i][n], f = mesh->bfacet_facet[i][n];
                const int g = (int)f_elem.size();
                f_elem.push_back(e); f_facet.push_back(f); f_kind.push_back(3 + (t - 1));
                f_val.push_back(params->stress_bc_values[i]);
                for (int j = 0; j < 3; ++j) {
                    const int node = mesh->connectivity[(size_t)NODE_OF_FACET_H[f][j]*ne + e];
                    node_ent[node].push_back(((g * 3 + j) << 1) | 1);
                }
                mask |= (1u << i);
            }
        }
        // every flagged node must be able to index bcn_idx; interior nodes never read it
        h->bc_mask = mask;
        h->nbcf = (int)f_elem.size();
        std::vector<int> idx((size_t)nn + 1, 0), ent;
        for (int n = 0; n < nn; ++n) {
            idx[n] = (int)ent.size();
            ent.insert(ent.end(), node_ent[n].begin(), node_ent[n].end());
        }
        idx[nn] = (int)ent.size();
        CK(dev_alloc(h->bcf_elem, f_elem.size())); CK(dev_upload(h->bcf_elem, f_elem.data(), f_elem.size(), h->stream));
        CK(dev_alloc(h->bcf_facet, f_facet.size())); CK(dev_upload(h->bcf_facet, f_facet.data(), f_facet.size(), h->stream));
        CK(dev_alloc(h->bcf_kind, f_kind.size())); CK(dev_upload(h->bcf_kind, f_kind.data(), f_kind.size(), h->stream));
        CK(dev_alloc(h->bcf_val, f_val.size())); CK(dev_upload(h->bcf_val, f_val.data(), f_val.size(), h->stream));
        CK(dev_alloc(h->bcf_tmp, f_elem.size() * 9));
        CK(dev_alloc(h->bcn_idx, idx.size())); CK(dev_upload(h->bcn_idx, idx.data(), idx.size(), h->stream));
        CK(dev_alloc(h->bcn_ent, ent.size())); CK(dev_upload(h->bcn_ent, ent.data(), ent.size(), h->stream));
    }
    // surface
    {
        h->ntop = mesh->ntop; h->etop = mesh->etop; h->ntop_elems = mesh->ntop_elems;
        const size_t ntop = (size_t)h->ntop, etop = (size_t)h->etop;
        CK(dev_alloc(h->top_nodes, ntop)); CK(dev_upload(h->top_nodes, mesh->top_nodes, ntop, h->stream));
        CK(dev_alloc(h->ean, 3*etop)); CK(dev_upload(h->ean, mesh->elem_and_nodes, 3*etop, h->stream));
        CK(dev_alloc(h->conn_surf, 4*etop)); CK(dev_upload(h->conn_surf, mesh->connectivity_surface, 4*etop, h->stream));
        CK(dev_alloc(h->ssup_idx, ntop + 1));
        if (ntop) CK(dev_upload(h->ssup_idx, mesh->support_surf_idx, ntop + 1, h->stream));
        const size_t nss = ntop ? (size_t)mesh->support_surf_idx[ntop] : 0;
        CK(dev_alloc(h->ssup_arr, nss)); CK(dev_upload(h->ssup_arr, mesh->support_surf_arr, nss, h->stream));
        {
            std::vector<int> sn(3 * nss);
            for (size_t k = 0; k < nss; ++k)
                for (int m = 0; m < 3; ++m) sn[3*k + m] = mesh->connectivity_surface[(size_t)m*etop + mesh->support_surf_arr[k]];
            CK(dev_alloc(h->ssup_nodes, 3 * nss)); CK(dev_upload(h->ssup_nodes, sn.data(), 3 * nss, h->stream));
        }
        {
            std::vector<unsigned char> flag((size_t)ne, 0);
            for (int i = 0; i < h->ntop_elems; ++i) flag[mesh->top_elems[i]] = 1;
            CK(dev_alloc(h->topflag, (size_t)ne)); CK(dev_upload(h->topflag, flag.data(), (size_t)ne, h->stream));
        }
        if (h->patch && ntop) {
            std::vector<int2> tf((size_t)nn, make_int2(-1, 0));
            bool fits = true;                      // 4 bits for the fan size (a surface node of a tet mesh has ~6 facets)
            for (int i = 0; i < h->ntop; ++i) {
                const int jb = mesh->support_surf_idx[i], nf = mesh->support_surf_idx[i + 1] - jb;
                fits = fits && nf >= 0 && nf <= 15;
                tf[mesh->top_nodes[i]] = make_int2(i | (nf << 27), jb);
            }
            if (fits) { CK(dev_alloc(h->tfan, (size_t)nn)); CK(dev_upload(h->tfan, tf.data(), (size_t)nn, h->stream)); }
        }
        CK(dev_alloc(h->dh, ntop)); CK(dev_alloc(h->edvacc, etop)); CK(dev_alloc(h->znew, ntop));
        HK(hipMemsetAsync(h->dh, 0, 8*std::max<size_t>(ntop, 1), h->stream));
        HK(hipMemsetAsync(h->edvacc, 0, 8*std::max<size_t>(etop, 1), h->stream));
    }
    CK(dev_alloc(h->bnormals, (size_t)3*DES_NBDRY)); CK(dev_upload(h->bnormals, mesh->bnormals, (size_t)3*DES_NBDRY, h->stream));
    CK(dev_alloc(h->edge_vec, (size_t)3*mesh->nedge)); CK(dev_upload(h->edge_vec, mesh->edge_vec, (size_t)3*mesh->nedge, h->stream));
    CK(dev_alloc(h->edge_slot, (size_t)DES_NBDRY*DES_NBDRY));
    CK(dev_upload(h->edge_slot, mesh->edge_slot, (size_t)DES_NBDRY*DES_NBDRY, h->stream));
    HK(hipStreamSynchronize(h->stream));
#undef CK
#undef HK
    return h;
}

long long des_dev_field_count(const des_dev *h, int field)
{
    if (h && h->d2) return des2d::field_count(h->d2, field);
    const long long nn = h->nn, ne = h->ne;
    switch (field) {
    case DES_F_COORD: case DES_F_VEL: case DES_F_FORCE: case DES_F_FORCE_RESIDUAL: case DES_F_COORD0: return 3*nn;
    case DES_F_COORD_AVG0: return h->coord_avg0 ? 3*nn : -1;
    case DES_F_STRESS_AVG: case DES_F_STRAIN0: return h->stress_avg ? 6*ne : -1;
    case DES_F_DPLSTRAIN_AVG: return h->dplstrain_avg ? ne : -1;
    case DES_F_TEMPERATURE: case DES_F_VOLUME_N: case DES_F_MASS: case DES_F_TMASS: case DES_F_DHACC: case DES_F_NTMP: return nn;
    case DES_F_STRESS: case DES_F_STRAIN: case DES_F_STRAIN_RATE: return 6*ne;
    case DES_F_PLSTRAIN: case DES_F_DELTA_PLSTRAIN: case DES_F_VISCOSITY: case DES_F_VOLUME: case DES_F_VOLUME_OLD:
    case DES_F_DPRESSURE: case DES_F_RADIOGENIC: return ne;
    case DES_F_ELEMMARKERS: return ne * h->nmat;
    case DES_F_EDVACC_SURF: return h->etop;
    case DES_F_DH: return h->ntop;
    default: return -1;
    }
}

static double *plain_field(des_dev *h, int field)
{
    switch (field) {
    case DES_F_FORCE: return h->force;
    case DES_F_FORCE_RESIDUAL: return h->fres;
    case DES_F_COORD0: return h->coord0;
    case DES_F_VOLUME_N: return h->volume_n;
    case DES_F_TMASS: return h->tmass;
    case DES_F_DHACC: return h->dhacc;
    case DES_F_NTMP: return h->ntmp;
    case DES_F_STRESS: return h->stress;
    case DES_F_STRAIN: return h->strain;
    case DES_F_STRAIN_RATE: return h->strain_rate;
    case DES_F_PLSTRAIN: return h->plstrain;
    case DES_F_DELTA_PLSTRAIN: return h->delta_plstrain;
    case DES_F_VISCOSITY: return h->viscosity;
    case DES_F_VOLUME: return h->volume;
    case DES_F_VOLUME_OLD: return h->volume_old;
    case DES_F_DPRESSURE: return h->dpressure;
    case DES_F_RADIOGENIC: return h->radiogenic;
    case DES_F_EDVACC_SURF: return h->edvacc;
    case DES_F_DH: return h->dh;
    case DES_F_STRESS_AVG: return h->stress_avg;
    case DES_F_DPLSTRAIN_AVG: return h->dplstrain_avg;
    case DES_F_STRAIN0: return h->strain0;
    case DES_F_COORD_AVG0: return h->coord_avg0;
    default: return nullptr;
    }
}

// packed nodal records <-> the reference's SoA arrays
static int packed_io(des_dev *h, int field, void *host, bool upload)
{
    const size_t nn = (size_t)h->nn;
    d4 *dev = (field == DES_F_COORD || field == DES_F_TEMPERATURE) ? h->xt : h->vm;
    std::vector<d4> tmp(nn);
    HIP_OK(hipMemcpyAsync(tmp.data(), dev, nn * sizeof(d4), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    double *a = (double *)host;
    const bool vec = (field == DES_F_COORD || field == DES_F_VEL);
    const int *o2n = h->n_old2new.empty() ? nullptr : h->n_old2new.data();     // caller's id -> device id
    if (!upload) {
        for (size_t n = 0; n < nn; ++n) {
            const d4 &t = tmp[o2n ? (size_t)o2n[n] : n];
            if (vec) { a[n] = t.x; a[nn + n] = t.y; a[2*nn + n] = t.z; }
            else a[n] = t.w;
        }
        return DES_OK;
    }
    for (size_t n = 0; n < nn; ++n) {
        d4 &t = tmp[o2n ? (size_t)o2n[n] : n];
        if (vec) { t.x = a[n]; t.y = a[nn + n]; t.z = a[2*nn + n]; }
        else t.w = a[n];
    }
    HIP_OK(hipMemcpyAsync(dev, tmp.data(), nn * sizeof(d4), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int des_dev_upload(des_dev *h, int field, const void *host, long long count)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    D2_FORWARD(h, upload(h->d2, field, host, count));
    if (!h || !host) return DES_ERR_INTERNAL;
    if (count != des_dev_field_count(h, field)) { g_last_error = "field size mismatch"; return DES_ERR_INTERNAL; }
    hipSetDevice(h->device);
    if (field == DES_F_COORD || field == DES_F_VEL || field == DES_F_TEMPERATURE || field == DES_F_MASS)
        return packed_io(h, field, const_cast<void *>(host), true);
    if (field == DES_F_ELEMMARKERS) {
        h->markers_dirty = true;
        if (h->e_new2old.empty()) return dev_upload(h->markers, (const int *)host, (size_t)count, h->stream);
        std::vector<int> tmp((size_t)count);
        permute_planes((const int *)host, tmp.data(), (size_t)h->ne, 0, (size_t)h->nmat, h->e_new2old, true);
        return dev_upload(h->markers, tmp.data(), (size_t)count, h->stream);
    }
    double *dst = plain_field(h, field);
    if (!dst) return DES_ERR_INTERNAL;
    if (field == DES_F_RADIOGENIC) {
        // +0.0 everywhere (bit pattern 0): 0.0 * vol * rho / 4 is formed without the 8 B per listed patch element
        const unsigned long long *u = (const unsigned long long *)host;
        bool z = true;
        for (long long i = 0; i < count && z; ++i) z = u[i] == 0ull;
        h->radiogenic_zero = z;
    }
    const int space = field_space(field);
    if (space == 0 || h->n_new2old.empty()) return dev_upload(dst, (const double *)host, (size_t)count, h->stream);
    const size_t n = space == 1 ? (size_t)h->nn : (size_t)h->ne;
    std::vector<double> tmp((size_t)count);
    permute_planes((const double *)host, tmp.data(), n, (size_t)count / n, 0, space == 1 ? h->n_new2old : h->e_new2old, true);
    return dev_upload(dst, tmp.data(), (size_t)count, h->stream);
}

int des_dev_download(des_dev *h, int field, void *host, long long count)
{
    D2_FORWARD(h, download(h->d2, field, host, count));
    if (!h || !host) return DES_ERR_INTERNAL;
    if (count != des_dev_field_count(h, field)) { g_last_error = "field size mismatch"; return DES_ERR_INTERNAL; }
    hipSetDevice(h->device);
    if (field == DES_F_COORD || field == DES_F_VEL || field == DES_F_TEMPERATURE || field == DES_F_MASS)
        return packed_io(h, field, host, false);
    const void *src = (field == DES_F_ELEMMARKERS) ? (const void *)h->markers : (const void *)plain_field(h, field);
    if (!src) return DES_ERR_INTERNAL;
    if (count == 0) return DES_OK;
    const size_t bytes = (size_t)count * (field == DES_F_ELEMMARKERS ? 4 : 8);
    const int space = field == DES_F_ELEMMARKERS ? 2 : field_space(field);
    if (space == 0 || h->n_new2old.empty()) {
        HIP_OK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        return DES_OK;
    }
    std::vector<char> tmp(bytes);
    HIP_OK(hipMemcpyAsync(tmp.data(), src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    if (field == DES_F_ELEMMARKERS)
        permute_planes((const int *)tmp.data(), (int *)host, (size_t)h->ne, 0, (size_t)h->nmat, h->e_new2old, false);
    else {
        const size_t n = space == 1 ? (size_t)h->nn : (size_t)h->ne;
        permute_planes((const double *)tmp.data(), (double *)host, n, (size_t)count / n, 0,
                       space == 1 ? h->n_new2old : h->e_new2old, false);
    }
    return DES_OK;
}

int des_dev_set_clock(des_dev *h, double dt, double time, long long steps)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    D2_FORWARD(h, set_clock(h->d2, dt, time, steps));
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    int rc = sync_clock(h);
    if (rc) return rc;
    h->h_clk->dt = dt; h->h_clk->time = time; h->h_clk->steps = steps;
    h->steps_host = steps;
    HIP_OK(hipMemcpyAsync(h->d_clk, h->h_clk, sizeof(DevClock), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

// isostasy_adjustment (dynearthsol.cxx:496-544) is the time step minus the clock, the
// temperature update, NMD_stress, the velocity bcs, rotate_stress and compute_dt: DevClock::iso
// tells the kernels, the host side skips the N2 launch and passes nmd = 0 to E3.  (update_stress
// still stores dpressure and compute_mass the thermal mass, as in the reference.)
int des_dev_set_isostasy(des_dev *h, int on)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    D2_FORWARD(h, set_isostasy(h->d2, on));
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    int rc = sync_clock(h);
    if (rc) return rc;
    h->iso = on != 0;
    h->h_clk->iso = h->iso;
    HIP_OK(hipMemcpyAsync(h->d_clk, h->h_clk, sizeof(DevClock), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int des_dev_sync(des_dev *h)
{
    D2_FORWARD(h, sync(h->d2));
    if (!h) return DES_ERR_INTERNAL;
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int des_dev_init_geometry(des_dev *h)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    D2_FORWARD(h, init_geometry(h->d2));
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    refresh_props(h);
    // compute_volume; volume_old = volume (dynearthsol.cxx:184-188): run the C part twice so
    // both arrays hold the new volume (the first run also copies the stale volume into
    // volume_old, the second overwrites it)
    launch_e1<MODE_C | MODE_INIT>(h);
    launch_e1<MODE_C | MODE_INIT>(h);
    // apply_vbcs (dynearthsol.cxx:192) on every local node: a purely nodal operation that
    // gives halo nodes the same values their owners compute
    hipLaunchKernelGGL(k_apply_vbcs, dim3(nblk(h->nn)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->nn,
                       h->bcflag, h->bnormals, h->edge_vec, h->edge_slot, h->vm, (d4 *)nullptr);
    // compute_mass (dynearthsol.cxx:194)
    launch_mass_gather(h);
    HIP_OK(hipStreamSynchronize(h->stream));
    HIP_OK(hipGetLastError());
    return DES_OK;
}

int des_dev_compute_dt(des_dev *h, double *dt)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    D2_FORWARD(h, compute_dt(h->d2, dt));
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    refresh_props(h);
    launch_e1<MODE_DT>(h);
    int rc = reduce_dt(h);
    if (rc) return rc;
    rc = sync_clock(h);
    if (rc) return rc;
    if (dt) *dt = h->h_clk->dt;
    return h->h_clk->status;
}

// One step.  On a decomposed mesh everything up to the committed surface heights runs on the
// local mesh alone (redundantly on the ghost region), then ONE exchange refreshes the ghost
// region (des_halo, des_params.h), then the end-of-step geometry pass; the end of step t (E1's C
// part) stays fused with the start of step t+1 (A part) whenever another step follows.
//
// The step is issued in three pieces -- step_front (everything up to the committed surface heights),
// the exchange of the ghost region, step_back (the rest of the surface bookkeeping and the end-of-step
// pass) -- so that the same launches serve des_dev_step (RCCL between the pieces) and
// des_dev_step_group (several engines of one process, device-to-device copies between them).
namespace {
struct StepPlan {
    bool multi, iso, nmd, graphs, pgraphs;
    long long qcsi;
    int nsteps;
};

void step_plan(des_dev *h, int nsteps, StepPlan &c)
{
    c.nsteps = nsteps;
    c.multi = h->nnbr > 0;
    c.iso = h->iso;
    c.nmd = h->p.is_using_mixed_stress && !c.iso;
    // DES_GRAPH=1: the launches of an interior step of a single-GPU call are replayed from a
    // hipGraph captured once (two graphs: with and without the compute_dt variant of E1)
    c.graphs = h->use_graph && !c.multi && !c.iso && !h->prof && !h->p.is_outputting_averaged_fields
               && !h->patch           // EN3 swaps the two coordinate buffers every step: nothing to replay
               && !h->p.has_PT;
    // ... and of the fused step (EN1 .. S3 with E2<GEO>): EN1 and EN3 each swap the two coordinate
    // buffers, so the steps inside a call all start on the same one (the first step of a call swaps
    // once, N1 + EN3); three graphs per buffer, because the E2 after a compute_dt step rotates with
    // the dt of before it and a compute_dt step ends with the reduction
    c.pgraphs = h->use_graph && h->patch && !c.multi && !h->prof && e2geo_ok(h)
                && !h->p.is_outputting_averaged_fields;       // (k_avg_coord0 rides on some steps)
    c.qcsi = h->p.quality_check_step_interval;
    h->n_pt_iterations = 0;
    if (h->e2_defer == 2 && e2geo_ok(h)) h->e2_two_pass = false;      // (choose_e2_mode: the fused step runs one pass)
}

inline bool step_overlapped(const des_dev *h, const StepPlan &c)
{
    return c.multi && h->overlap && !c.iso && h->e_int1 > h->e_int0;
}

// Step i of the call up to the committed surface heights.  *whole = true: the step was replayed from a
// hipGraph, end-of-step pass and compute_dt included (single GPU only) -- nothing is left to do for it.
// the rest of step_front behind the force pass (and the pseudo-transient loop, if any)
int step_front_tail(des_dev *h, const StepPlan &c, int i, long long step_no)
{
    const int nsteps = c.nsteps;
    if (s2_defer_ok(h, i < nsteps - 1, step_no)) {
        // no S2 / S3 launch: the next step's EN1 and E2 do the surface step of this one; with diffusion switched off
        // there is nothing to do at all (dh = 0: heights, dhacc and edvacc_surf keep their values)
        h->s2_pending = surface_diffusion_on(h);
        h->s2_skipped = true;
        return DES_OK;
    }
    h->s2_skipped = false;
    launch_s2(h, step_no);
    // decomposed: only the commit of the surface heights -- the rest of S3 reads the ghost nodes' dh
    if (c.multi) launch_s3(h, true, false, false);
    else         launch_s3(h, true, true, true);
    return DES_OK;
}

// pt_pending != nullptr (des_dev_step_group): with control.has_PT the front stops behind the force pass -- the loop runs for
// all engines of the group in lockstep (pt_loop_group), step_front_tail follows
int step_front(des_dev *h, const StepPlan &c, int i, long long *step_no_out, bool *whole, bool *pt_pending = nullptr)
{
    int rc;
    const int nsteps = c.nsteps;
    const long long step_no = c.iso ? h->steps_host : ++h->steps_host;
    *step_no_out = step_no;
    *whole = false;
    const bool fresh = i == 0 && !c.iso && fresh_ok(h);
    h->finished = false;
    if (i == 0 && !fresh) launch_e1<MODE_A>(h);
    h->e2_fresh = fresh;
    if (c.graphs && i < nsteps - 1 && step_no % c.qcsi != 0) {
        const int which = (step_no % 10 == 0) ? 1 : 0;
        if (!h->graph_exec[which] || h->graph_two_pass[which] != h->e2_two_pass) {
            if (h->graph_exec[which]) { hipGraphExecDestroy(h->graph_exec[which]); h->graph_exec[which] = nullptr; }
            hipGraph_t g = nullptr;
            HIP_OK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            launch_n1(h); launch_e2(h); if (c.nmd) launch_n2(h); launch_force_pass(h);
            launch_s2(h, 1);                                   // (no dhacc reset: those steps are not replayed)
            launch_s3(h, true, true, true);
            launch_e1_end(h, which ? 10 : 1, true);
            if (which) launch_dt_finalize(h, nullptr);
            HIP_OK(hipStreamEndCapture(h->stream, &g));
            HIP_OK(hipGraphInstantiate(&h->graph_exec[which], g, nullptr, nullptr, 0));
            hipGraphDestroy(g);
            h->graph_two_pass[which] = h->e2_two_pass;
        }
        HIP_OK(hipGraphLaunch(h->graph_exec[which], h->stream));
        *whole = true;
        return DES_OK;
    }
    h->e2_elide = h->elide_ok && i < nsteps - 1;
    h->e2_not_last = i < nsteps - 1;
    if (c.pgraphs && i > 0 && i < nsteps - 1 && step_no % c.qcsi != 0 && h->e2geo_next) {
        const bool do_dt = (step_no % 10 == 0);
        const int which = 2 * (do_dt ? 1 : (h->rot_prev_dt ? 2 : 0)) + (h->xt < h->xt_alt ? 0 : 1);
        if (!h->pgraph_exec[which] || h->pgraph_two_pass[which] != h->e2_two_pass) {
            if (h->pgraph_exec[which]) { hipGraphExecDestroy(h->pgraph_exec[which]); h->pgraph_exec[which] = nullptr; }
            hipGraph_t g = nullptr;
            d4 *const xt0 = h->xt;
            HIP_OK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            launch_en1(h); launch_e2(h); if (c.nmd) launch_n2(h); launch_force_pass(h);
            launch_s2(h, 1);
            launch_s3(h, true, true, true);
            launch_e1_end(h, do_dt ? 10 : 1, true);
            if (do_dt) launch_dt_finalize(h, nullptr);
            HIP_OK(hipStreamEndCapture(h->stream, &g));
            if (h->xt != xt0) { hipGraphDestroy(g); g_last_error = "graph capture: the coordinate buffers did not swap back"; return DES_ERR_INTERNAL; }
            HIP_OK(hipGraphInstantiate(&h->pgraph_exec[which], g, nullptr, nullptr, 0));
            hipGraphDestroy(g);
            h->pgraph_two_pass[which] = h->e2_two_pass;
        }
        HIP_OK(hipGraphLaunch(h->pgraph_exec[which], h->stream));
        h->rot_pending = false; h->e2geo_next = true; h->rot_prev_dt = do_dt;     // as the launches leave them
        *whole = true;
        return DES_OK;
    }
    // inside a multi-step call the step before ended with the fused E1<C | A | NOREC>, and EN1
    // forms the element terms itself; the first step of a call gathers what E1 stored (the
    // caller may have uploaded fields in between)
    if (h->join_pending) {
        // the exchange of the step before is still on the side stream (step_back): EN1 and E2<GEO> of the deep part of
        // the slab run beside it, the rest behind the join
        const bool split = i > 0 && en1_ok(h) && h->e2geo_next && deep_split_ok(h);
        if (split) { launch_en1(h, PART_DEEP); launch_e2(h, PART_DEEP); }
        if ((rc = exchange_join(h))) return rc;
        h->join_pending = false;
        if (split) { launch_en1(h, PART_REST); launch_e2(h, PART_REST); }
        else { if (i > 0 && en1_ok(h)) launch_en1(h); else launch_n1(h); launch_e2(h); }
    } else {
        if ((i > 0 || fresh) && en1_ok(h)) launch_en1(h); else launch_n1(h);
        launch_e2(h);
    }
    if (c.nmd) launch_n2(h);
    launch_force_pass(h);
    if (pt_pending) *pt_pending = false;
    if (h->p.has_PT && !c.iso) {
        if (pt_pending) { *pt_pending = true; return DES_OK; }
        if ((rc = pt_loop(h))) return rc;
    }
    return step_front_tail(h, c, i, step_no);
}

// The rest of step i, once the exchange has been issued (decomposed meshes).  Returns through *do_dt
// whether the compute_dt partials of this step are waiting for their reduction.
int step_back(des_dev *h, const StepPlan &c, int i, long long step_no, bool *do_dt)
{
    int rc;
    const bool last = (i == c.nsteps - 1);
    *do_dt = false;
    if (step_overlapped(h, c) && e2geo_ok(h)) {
        if (h->s2_skipped && !last && deep_split_ok(h)) {
            // the fused step: nothing left of this step reads the ghost region -- the join waits for the next step's
            // front, which has work for the meantime (deep_split_ok)
            launch_e1_end(h, step_no, true);           // (a plain step: no launch, the next stress update does it)
            h->join_pending = true;
            return DES_OK;
        }
        if ((rc = exchange_join(h))) return rc;        // a step that needs the ghost region now: in order from here
    } else if (step_overlapped(h, c)) {
        // (classic passes, DES_E2GEO=0) the exchange runs on the side stream || end-of-step pass of the interior
        // elements; then the rest of the surface bookkeeping (it reads the ghost nodes' dh) and the two element
        // groups that touch the ghost region
        launch_e1_end(h, step_no, !last, E1_INTERIOR);
        if ((rc = exchange_join(h))) return rc;
        launch_avg_coord0(h, step_no);                 // owned and ghost coordinates alike: after the join
        if (!h->s2_skipped) launch_s3(h, false, true, true);
        launch_e1_end(h, step_no, !last, E1_GHOST_SIDE);
        *do_dt = (step_no % 10 == 0);
        return DES_OK;
    }
    if (c.multi && !h->s2_skipped) launch_s3(h, false, true, true);
    if (c.iso) {                                       // no averaging, no compute_dt in that loop
        if (last) launch_e1<MODE_C>(h); else launch_e1<MODE_C | MODE_A>(h);
        return DES_OK;
    }
    launch_avg_coord0(h, step_no);
    launch_e1_end(h, step_no, !last);
    *do_dt = (step_no % 10 == 0);
    return DES_OK;
}

// after the last step of a call: the scalars of des_scalars from the (synchronised) clock
void fill_scalars(const des_dev *h, des_scalars *out)
{
    const DevClock &c = *h->h_clk;
    out->dt = c.dt; out->time = c.time; out->l2_residual = c.l2_residual; out->max_surf_vel = c.max_surf_vel;
    out->max_global_vel_mag = c.max_global_vel_mag; out->global_dt_min = c.global_dt_min;
    out->steps = c.steps; out->status = c.status; out->n_return_mapping = c.n_defer; out->avg_time0 = c.avg_time0;
    out->n_pt_iterations = h->n_pt_iterations;
}
// A step that fails half-way (an RCCL or launch error in the exchange, the PT loop, a hipGraph capture) must not leave
// the engine between two pieces of the overlapped schedule: work may still be on the side stream, and the flags that
// tell the next step's front what the last one left to it would otherwise outlive the call -- des_dev_set_overlap
// would refuse for ever ("inside a step"), the next des_dev_step would wait on a stale join event.  After this the
// engine can be stepped again from uploads (the device state itself is that of an interrupted step: not a result).
int step_abort(des_dev *h, int rc)
{
    hipSetDevice(h->device);
    if (h->comm_stream) hipStreamSynchronize(h->comm_stream);
    hipStreamSynchronize(h->stream);
    h->join_pending = false; h->s2_pending = false; h->edv_pending = false; h->s2_skipped = false;
    h->e2geo_next = false; h->rot_pending = false; h->e2_fresh = false;
    h->finished = false;
    // the two device flags a loop may have been inside of when the error came (DevClock::pt: boundaries at rest, no clock;
    // no_neumann: initial_body_force_adjustment) -- a later step must not inherit them
    static const int zero = 0;
    hipMemcpy(&h->d_clk->pt, &zero, sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(&h->d_clk->no_neumann, &zero, sizeof(int), hipMemcpyHostToDevice);
    h->in_pt = false;
    return rc;
}
} // namespace

int des_dev_step(des_dev *h, int nsteps, des_scalars *out)
{
    D2_FORWARD(h, step(h->d2, nsteps, out));
    if (!h) return DES_ERR_INTERNAL;
    if (h->group) { g_last_error = "this engine belongs to a group: step it with des_dev_step_group"; return DES_ERR_INTERNAL; }
    hipSetDevice(h->device);
    refresh_props(h);
    StepPlan c;
    step_plan(h, nsteps, c);
    int rc;
    for (int i = 0; i < nsteps; ++i) {
        long long step_no; bool whole, do_dt;
        if ((rc = step_front(h, c, i, &step_no, &whole))) return step_abort(h, rc);
        if (whole) continue;
        if (c.multi && (rc = step_overlapped(h, c) ? exchange_begin(h) : exchange(h))) return step_abort(h, rc);
        if ((rc = step_back(h, c, i, step_no, &do_dt))) return step_abort(h, rc);
        if (do_dt && (rc = reduce_dt(h))) return step_abort(h, rc);
    }
    // compute_mass gather of the last update_mesh, so that volume_n / mass / tmass hold the
    // reference's end-of-step values (inside a multi-step call it is fused into the next N1)
    if (nsteps > 0) launch_mass_gather(h);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { g_last_error = std::string("kernel launch: ") + hipGetErrorString(le); return step_abort(h, DES_ERR_RESOURCE); }
    // the state is now the one the reference holds after `steps` steps; the next call may start from it without the
    // classic first step (engine/launch.hpp: fresh_ok) unless another entry point touches it first
    if (nsteps > 0) h->finished = !c.iso;
    if (out) {
        if (h->comm_size > 1) {
            // l2_residual is a sum over all ranks' owned nodes
            HIP_OK(hipMemcpyAsync(h->d_red + 6, &h->d_clk->l2_sum, 8, hipMemcpyDeviceToDevice, h->stream));
            ncclAllReduce(h->d_red + 6, h->d_red + 6, 1, ncclDouble, ncclSum, h->comm, h->stream);
        }
        rc = sync_clock(h);
        if (rc) return rc;
        choose_e2_mode(h);
        fill_scalars(h, out);
        if (h->comm_size > 1) {
            double l2sum = 0;
            HIP_OK(hipMemcpy(&l2sum, h->d_red + 6, 8, hipMemcpyDeviceToHost));
            out->l2_residual = std::sqrt(l2sum);
        }
        return h->h_clk->status;
    }
    return DES_OK;
}

// ---- several engines of one process as one decomposed model -------------------------------
int des_dev_group_attach(des_dev **engines, int n)
{
    if (!engines || n < 1) return DES_ERR_INTERNAL;
    {
        int n2 = 0;
        for (int k = 0; k < n; ++k) if (engines[k] && engines[k]->d2) ++n2;
        if (n2 == n) return DES_OK;                 // 2-D engines: des2d::step_group checks the lists each call
        if (n2) { g_last_error = "a group mixes 2-D and 3-D engines"; return DES_ERR_UNSUPPORTED_DIM; }
    }
    for (int k = 0; k < n; ++k) {
        des_dev *h = engines[k];
        if (!h) return DES_ERR_INTERNAL;
        if (h->comm) { g_last_error = "an engine with an RCCL communicator cannot join a group"; return DES_ERR_INTERNAL; }
        for (int q = 0; q < h->nnbr; ++q)
            if (h->nbr_rank[q] < 0 || h->nbr_rank[q] >= n || h->nbr_rank[q] == k) {
                g_last_error = "des_halo::nbr_rank outside the group"; return DES_ERR_INTERNAL;
            }
    }
    // every message must have its counterpart of the same length on the other side
    for (int k = 0; k < n; ++k)
        for (int q = 0; q < engines[k]->nnbr; ++q) {
            const des_dev *h = engines[k], *o = engines[h->nbr_rank[q]];
            int qo = -1;
            for (int j = 0; j < o->nnbr; ++j) if (o->nbr_rank[j] == k) qo = j;
            if (qo < 0 || o->send_off[qo + 1] - o->send_off[qo] != h->recv_off[q + 1] - h->recv_off[q]) {
                g_last_error = "group: the exchange lists of two neighbours do not match"; return DES_ERR_INTERNAL;
            }
        }
    for (int k = 0; k < n; ++k) {
        des_dev *h = engines[k];
        hipSetDevice(h->device);
        delete[] h->group;
        h->group = new des_dev *[n];
        for (int j = 0; j < n; ++j) h->group[j] = engines[j];
        h->group_n = n; h->group_rank = k;
        if (!h->ev_packed) HIP_OK(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
        if (!h->ev_taken) HIP_OK(hipEventCreateWithFlags(&h->ev_taken, hipEventDisableTiming));
    }
    return DES_OK;
}

int des_dev_group_detach(des_dev **engines, int n)
{
    if (!engines || n < 0) return DES_ERR_INTERNAL;
    for (int k = 0; k < n; ++k) {
        des_dev *h = engines[k];
        if (!h || h->d2) continue;
        hipSetDevice(h->device);
        hipStreamSynchronize(h->stream);
        if (h->comm_stream) hipStreamSynchronize(h->comm_stream);
        delete[] h->group;
        h->group = nullptr; h->group_n = 0; h->group_rank = 0;
    }
    return DES_OK;
}

int des_dev_step_group(des_dev **engines, int n, int nsteps, des_scalars *out)
{
    if (!engines || n < 1 || nsteps < 0) return DES_ERR_INTERNAL;
    if (engines[0] && engines[0]->d2) {
        std::vector<des2d::Engine *> g2((size_t)n);
        for (int k = 0; k < n; ++k) {
            if (!engines[k] || !engines[k]->d2) { g_last_error = "a group mixes 2-D and 3-D engines"; return DES_ERR_UNSUPPORTED_DIM; }
            g2[k] = engines[k]->d2;
        }
        const int rc2 = des2d::step_group(g2.data(), n, nsteps, out);
        if (rc2) for (int k = 0; k < n; ++k) if (!des2d::last_error(g2[k]).empty()) { g_last_error = des2d::last_error(g2[k]); break; }
        return rc2;
    }
    for (int k = 0; k < n; ++k)
        if (!engines[k] || engines[k]->d2 || engines[k]->group_n != n || engines[k]->group_rank != k || engines[k]->group[k] != engines[k]) {
            g_last_error = "des_dev_step_group: not the group des_dev_group_attach was given"; return DES_ERR_INTERNAL;
        }
    std::vector<StepPlan> plan((size_t)n);
    for (int k = 0; k < n; ++k) {
        des_dev *h = engines[k];
        hipSetDevice(h->device);
        refresh_props(h);
        step_plan(h, nsteps, plan[k]);
        if (plan[k].graphs || plan[k].pgraphs) { g_last_error = "des_dev_step_group: decomposed engines only (no hipGraph replay)"; return DES_ERR_UNSUPPORTED; }
    }
    int rc;
    auto abort_all = [&](int code) { for (int k = 0; k < n; ++k) step_abort(engines[k], code); return code; };
    std::vector<long long> step_no((size_t)n);
    for (int i = 0; i < nsteps; ++i) {
        bool any_dt = false;
        bool any_pt = false;
        for (int k = 0; k < n; ++k) {                  // every engine's step up to its packed messages
            des_dev *h = engines[k];
            bool whole, pt = false;
            hipSetDevice(h->device);
            if ((rc = step_front(h, plan[k], i, &step_no[k], &whole, &pt))) return abort_all(rc);
            any_pt = any_pt || pt;
            if (!pt && plan[k].multi && (rc = exchange_local_pack(h))) return abort_all(rc);
        }
        if (any_pt) {
            // control.has_PT: the fronts have stopped behind the force pass; the loop for all engines in lockstep (a ghost
            // refresh and one residual per iteration), then the rest of every front and its messages
            if ((rc = pt_loop_group(engines, n, true))) return abort_all(rc);
            for (int k = 0; k < n; ++k) {
                des_dev *h = engines[k];
                hipSetDevice(h->device);
                if ((rc = step_front_tail(h, plan[k], i, step_no[k]))) return abort_all(rc);
                if (plan[k].multi && (rc = exchange_local_pack(h))) return abort_all(rc);
            }
        }
        for (int k = 0; k < n; ++k) {                  // the messages change hands; the rest of the step
            des_dev *h = engines[k];
            bool do_dt;
            hipSetDevice(h->device);
            if (plan[k].multi && (rc = exchange_local_take(h, step_overlapped(h, plan[k]) ? h->comm_stream : h->stream))) return abort_all(rc);
            if ((rc = step_back(h, plan[k], i, step_no[k], &do_dt))) return abort_all(rc);
            any_dt = any_dt || do_dt;
        }
        if (any_dt && (rc = reduce_dt_group(engines, n))) return abort_all(rc);
    }
    double l2sum = 0;
    int status = DES_OK;
    for (int k = 0; k < n; ++k) {
        des_dev *h = engines[k];
        hipSetDevice(h->device);
        if (nsteps > 0) launch_mass_gather(h);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { g_last_error = std::string("kernel launch: ") + hipGetErrorString(le); return DES_ERR_RESOURCE; }
        if (nsteps > 0) h->finished = !plan[k].iso;     // (as des_dev_step leaves it: engine/launch.hpp, fresh_ok)
        if (!out) continue;
        if ((rc = sync_clock(h))) return rc;
        choose_e2_mode(h);
        fill_scalars(h, &out[k]);
        l2sum += h->h_clk->l2_sum;                     // owned nodes only: every node counts once
        if (h->h_clk->status) status = h->h_clk->status;
    }
    if (out) for (int k = 0; k < n; ++k) out[k].l2_residual = std::sqrt(l2sum);
    return status;
}

// initial_body_force_adjustment (dynearthsol.cxx:546-591; main() calls it once before the time loop when
// ic.has_body_force_adjustment): the pseudo-transient loop on the initial state with the Neumann tractions held back
// (fields.cxx:690: apply_stress_bcs_neumann is skipped while Param::ic.has_body_force_adjustment is set).  Without
// control.has_PT the reference only forms the residual of the force_residual it holds; so does this.
int des_dev_body_force_adjustment(des_dev *h, des_scalars *out)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    if (!h) return DES_ERR_INTERNAL;
    D2_FORWARD(h, body_force_adjustment(h->d2, out));
    if (h->group) { g_last_error = "this engine belongs to a group: des_dev_body_force_adjustment_group"; return DES_ERR_INTERNAL; }
    if (h->nnbr > 0 && !(h->comm && h->comm_size > 1)) {
        g_last_error = "the initial body-force adjustment on a decomposed mesh: through RCCL (des_dev_comm_init) or des_dev_body_force_adjustment_group";
        return DES_ERR_UNSUPPORTED;
    }
    hipSetDevice(h->device);
    refresh_props(h);
    h->n_pt_iterations = 0;
    int rc;
    if (h->p.has_PT) {
        static const int on = 1, off = 0;
        HIP_OK(hipMemcpyAsync(&h->d_clk->no_neumann, &on, sizeof(int), hipMemcpyHostToDevice, h->stream));
        rc = pt_loop(h, false);
        if (rc) return step_abort(h, rc);                       // (clears DevClock::pt / no_neumann too)
        HIP_OK(hipMemcpyAsync(&h->d_clk->no_neumann, &off, sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (h->nnbr > 0 && (rc = exchange(h))) return step_abort(h, rc);      // the ghost region as the last iteration left the owners
        // (the loop leaves the masses of its last update_mesh in the element records: gathered as after a step)
        launch_mass_gather(h);
    } else if ((rc = residual_global(h))) return rc;
    if ((rc = sync_clock(h))) return rc;
    if (out) { fill_scalars(h, out); }
    return h->h_clk->status;
}

// ... for the engines of a group (des_dev_group_attach), in lockstep
int des_dev_body_force_adjustment_group(des_dev **engines, int n, des_scalars *out)
{
    if (!engines || n < 1) return DES_ERR_INTERNAL;
    for (int k = 0; k < n; ++k)
        if (!engines[k] || engines[k]->d2 || engines[k]->group_n != n || engines[k]->group_rank != k) {
            g_last_error = "des_dev_body_force_adjustment_group: not the group des_dev_group_attach was given (3-D engines)"; return DES_ERR_INTERNAL;
        }
    int rc;
    static const int on = 1, off = 0;
    for (int k = 0; k < n; ++k) {
        des_dev *h = engines[k];
        hipSetDevice(h->device);
        h->finished = false;
        refresh_props(h);
        h->n_pt_iterations = 0;
        if (h->p.has_PT) HIP_OK(hipMemcpyAsync(&h->d_clk->no_neumann, &on, sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    auto abort_all = [&](int code) { for (int k = 0; k < n; ++k) step_abort(engines[k], code); return code; };
    if (engines[0]->p.has_PT) {
        if ((rc = pt_loop_group(engines, n, false))) return abort_all(rc);
        for (int k = 0; k < n; ++k) { hipSetDevice(engines[k]->device); if (engines[k]->nnbr > 0 && (rc = exchange_local_pack(engines[k]))) return abort_all(rc); }
        for (int k = 0; k < n; ++k) {
            des_dev *h = engines[k];
            hipSetDevice(h->device);
            if (h->nnbr > 0 && (rc = exchange_local_take(h, h->stream))) return abort_all(rc);
            HIP_OK(hipMemcpyAsync(&h->d_clk->no_neumann, &off, sizeof(int), hipMemcpyHostToDevice, h->stream));
            launch_mass_gather(h);
        }
    } else if ((rc = residual_global_group(engines, n))) return rc;
    int status = DES_OK;
    for (int k = 0; k < n; ++k) {
        des_dev *h = engines[k];
        hipSetDevice(h->device);
        if ((rc = sync_clock(h))) return rc;
        if (out) fill_scalars(h, &out[k]);
        if (h->h_clk->status) status = h->h_clk->status;
    }
    return status;
}

// ---- domain decomposition ---------------------------------------------------------
int des_dev_set_halo(des_dev *h, const des_halo *halo, int nnode_global)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    if (!h || !halo) return DES_ERR_INTERNAL;
    D2_FORWARD(h, set_halo(h->d2, halo, nnode_global));
    if (halo->owned_begin < 0 || halo->owned_end > h->nn || halo->owned_begin >= halo->owned_end) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    h->o0 = halo->owned_begin; h->o1 = halo->owned_end; h->nn_global = nnode_global;
    h->g0 = (halo->nnbr > 0 || halo->owned_begin > 0 || halo->owned_end < h->nn) ? halo->owned_global_begin : 0;
    h->nnbr = halo->nnbr;
    h->nbr_rank.assign(halo->nbr_rank, halo->nbr_rank + halo->nnbr);
    const int nq = halo->nnbr;
    h->send_ptr.assign(halo->send_ptr, halo->send_ptr + nq + 1);
    h->recv_ptr.assign(halo->recv_ptr, halo->recv_ptr + nq + 1);
    h->esend_ptr.assign(halo->esend_ptr, halo->esend_ptr + nq + 1);
    h->erecv_ptr.assign(halo->erecv_ptr, halo->erecv_ptr + nq + 1);
    const size_t ns = (size_t)h->send_ptr[nq], nr = (size_t)h->recv_ptr[nq];
    const size_t nes = (size_t)h->esend_ptr[nq], ner = (size_t)h->erecv_ptr[nq];
    for (void *q : {(void *)h->d_send_idx, (void *)h->d_recv_idx, (void *)h->d_sendbuf, (void *)h->d_recvbuf,
                    (void *)h->d_esend_idx, (void *)h->d_erecv_idx, (void *)h->d_send_noff, (void *)h->d_send_eoff,
                    (void *)h->d_recv_noff, (void *)h->d_recv_eoff}) if (q) hipFree(q);
    if (!h->n_old2new.empty()) {
        // the internal order was laid out around des_mesh::owned_begin/end: it must be this range
        bool range_ok = true;
        for (int n = 0; n < h->nn && range_ok; ++n)
            range_ok = (n >= h->o0 && n < h->o1) == (h->n_old2new[n] >= h->o0 && h->n_old2new[n] < h->o1);
        if (!range_ok) { g_last_error = "des_halo owned range differs from des_mesh::owned_begin/owned_end"; return DES_ERR_INTERNAL; }
    }
    // a message = the node records of that neighbour, then its element records
    auto layout = [&](const std::vector<int> &np, const std::vector<int> &ep, std::vector<long long> &off,
                      std::vector<int> &noff, std::vector<int> &eoff) {
        off.assign((size_t)nq + 1, 0);
        noff.resize((size_t)np[nq]); eoff.resize((size_t)ep[nq]);
        for (int q = 0; q < nq; ++q) {
            long long base = off[q];
            for (int k = np[q]; k < np[q+1]; ++k) noff[k] = (int)(base + (long long)(k - np[q]) * DES_X_NODE_WIDTH);
            base += (long long)(np[q+1] - np[q]) * DES_X_NODE_WIDTH;
            for (int k = ep[q]; k < ep[q+1]; ++k) eoff[k] = (int)(base + (long long)(k - ep[q]) * DES_X_ELEM_WIDTH);
            off[q+1] = base + (long long)(ep[q+1] - ep[q]) * DES_X_ELEM_WIDTH;
        }
    };
    std::vector<int> snoff, seoff, rnoff, reoff;
    layout(h->send_ptr, h->esend_ptr, h->send_off, snoff, seoff);
    layout(h->recv_ptr, h->erecv_ptr, h->recv_off, rnoff, reoff);
    auto mapped = [&](const int *idx, size_t n, const std::vector<int> &map) {
        std::vector<int> v(idx, idx + n);
        if (!map.empty()) for (size_t k = 0; k < n; ++k) v[k] = map[v[k]];
        return v;
    };
    const std::vector<int> sidx = mapped(halo->send_idx, ns, h->n_old2new), ridx = mapped(halo->recv_idx, nr, h->n_old2new);
    const std::vector<int> seidx = mapped(halo->esend_idx, nes, h->e_old2new), reidx = mapped(halo->erecv_idx, ner, h->e_old2new);
    int rc;
    struct { int *&dst; const std::vector<int> &src; } ups[] = {
        {h->d_send_idx, sidx}, {h->d_recv_idx, ridx}, {h->d_esend_idx, seidx}, {h->d_erecv_idx, reidx},
        {h->d_send_noff, snoff}, {h->d_send_eoff, seoff}, {h->d_recv_noff, rnoff}, {h->d_recv_eoff, reoff} };
    for (auto &u : ups) {
        if ((rc = dev_alloc(u.dst, u.src.size()))) return rc;
        if ((rc = dev_upload(u.dst, u.src.data(), u.src.size(), h->stream))) return rc;
    }
    if ((rc = dev_alloc(h->d_sendbuf, (size_t)h->send_off[nq]))) return rc;
    if ((rc = dev_alloc(h->d_recvbuf, (size_t)h->recv_off[nq]))) return rc;
    // the residual partials are indexed by owned-node block
    if (h->res_part) hipFree(h->res_part);
    choose_npb(h);
    h->n3_blocks = res_part_size(h);
    { const int rcp = dev_alloc(h->res_part, (size_t)h->n3_blocks); if (rcp) return rcp; }
    return build_residual_blocks(h);               // the owned range (and its place in the global numbering) has changed
}

// The partition-independent residual across ranks for a caller that moves the data itself (des_dev_phase; des_params.h:
// DES_RES_BLOCK): this rank's block partials (first = global index of its first block), and -- every rank's put together in
// global block order -- the fixed-shape sum, which sets and returns l2_residual.
int des_dev_residual_blocks(des_dev *h, double *out, int cap, int *first, int *count)
{
    if (!h) return DES_ERR_INTERNAL;
    D2_FORWARD(h, residual_blocks(h->d2, out, cap, first, count));
    hipSetDevice(h->device);
    if (first) *first = h->res_b0;
    if (count) *count = h->res_nb_own;
    if (!out) return DES_OK;
    if (cap < h->res_nb_own) return DES_ERR_INTERNAL;
    launch_residual_blocks(h);
    HIP_OK(hipMemcpyAsync(out, h->res_blocks + h->res_b0, (size_t)h->res_nb_own * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int des_dev_residual_set(des_dev *h, const double *blocks, int nblocks, double *l2)
{
    if (!h || !blocks) return DES_ERR_INTERNAL;
    D2_FORWARD(h, residual_set(h->d2, blocks, nblocks, l2));
    if (nblocks != h->res_nb_global) { g_last_error = "des_dev_residual_set: not the global block count"; return DES_ERR_INTERNAL; }
    hipSetDevice(h->device);
    HIP_OK(hipMemcpyAsync(h->res_blocks, blocks, (size_t)nblocks * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    launch_residual_final(h);
    const int rc = sync_clock(h);
    if (rc) return rc;
    if (l2) *l2 = h->h_clk->l2_residual;
    return DES_OK;
}

int des_dev_comm_unique_id(unsigned char *id128)
{
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) { g_last_error = "ncclGetUniqueId failed"; return DES_ERR_RESOURCE; }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, 128);
    return DES_OK;
}

int des_dev_comm_init(des_dev *h, int nranks, int rank, const unsigned char *id128)
{
    if (!h || !id128) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclResult_t r = ncclCommInitRank(&h->comm, nranks, id, rank);
    if (r != ncclSuccess) { g_last_error = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); h->comm = nullptr; return DES_ERR_RESOURCE; }
    h->comm_rank = rank; h->comm_size = nranks;
    D2_FORWARD(h, set_comm(h->d2, (void *)h->comm));         // a 2-D engine: the same communicator inside its own step
    return DES_OK;
}

// Start-up self-check of the attached communicator (engine/selfcheck.hpp): rank count, the exchange's own messages
// with a pattern the receiver can verify, the three reductions.  Collective: every rank calls it, before the first step.
int des_dev_comm_selfcheck(des_dev *h, int expect_world)
{
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    if (h->d2) {
        const int rc2 = des2d::comm_selfcheck(h->d2, expect_world, h->comm_rank);
        if (rc2) g_last_error = des2d::last_error(h->d2);
        return rc2;
    }
    HIP_OK(hipStreamSynchronize(h->stream));
    const std::string bad = des_selfcheck::run(h->comm, h->stream, expect_world, h->comm_rank, h->nnbr, h->nbr_rank.data(),
                                               h->send_off.data(), h->recv_off.data(), h->d_sendbuf, h->d_recvbuf, h->d_red);
    if (!bad.empty()) { g_last_error = "RCCL self-check, rank " + std::to_string(h->comm_rank) + ": " + bad; return DES_ERR_RESOURCE; }
    return DES_OK;
}

// The engine's environment switches that are SET in this process, as far as the library has read them (engine/env.hpp):
// "NAME=value NAME=value", NUL-terminated, cut to len - 1 characters; returns the full length.
int des_dev_config_string(char *buf, int len)
{
    const std::string s = des_env::summary();
    if (buf && len > 0) {
        const size_t n = std::min(s.size(), (size_t)len - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size();
}

int des_dev_comm_info(des_dev *h, int *nranks, int *rank, int *overlapped)
{
    if (!h) return DES_ERR_INTERNAL;
    int n = 1, r = 0;
    if (h->comm) {
        if (ncclCommCount(h->comm, &n) != ncclSuccess || ncclCommUserRank(h->comm, &r) != ncclSuccess) {
            g_last_error = "ncclCommCount failed"; return DES_ERR_RESOURCE;
        }
    }
    if (nranks) *nranks = h->comm ? n : 0;
    if (rank) *rank = r;
    if (overlapped) *overlapped = h->d2 ? des2d::overlapped(h->d2) : (h->overlap && h->nnbr > 0 && h->e_int1 > h->e_int0);
    return DES_OK;
}

// Switches between the in-order schedule (everything on the engine's stream) and the overlapped one (DES_OVERLAP=1 at
// create selects it from the start) between two des_dev_step calls.
int des_dev_set_overlap(des_dev *h, int on)
{
    D2_FORWARD(h, set_overlap(h->d2, on));
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    HIP_OK(hipStreamSynchronize(h->stream));
    HIP_OK(hipStreamSynchronize(h->comm_stream));
    if (h->join_pending) { g_last_error = "des_dev_set_overlap inside a step"; return DES_ERR_INTERNAL; }
    h->overlap = on != 0;
    return DES_OK;
}

// The exchange of the ghost region through the attached communicator (what des_dev_step issues
// between the two phases of a step); asynchronous on the engine's stream.
int des_dev_exchange(des_dev *h)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    D2_REFUSE(h, "the RCCL communicator inside des_dev_step");
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    return exchange(h);
}

// One phase of a step without any communication: the caller moves the ghost-region state
// (des_dev_halo_pack / des_dev_halo_unpack) -- used to test the decomposition with several
// engines on one GPU.  Returns 1 after phase 1 when the compute_dt partials are ready.
int des_dev_phase(des_dev *h, int phase)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    if (h && h->d2) {
        const int r = des2d::phase(h->d2, phase);
        if (r < 0) g_last_error = des2d::last_error(h->d2);
        return r;
    }
    if (!h) return -DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    switch (phase) {
    case 0:
        refresh_props(h);
        if (!h->iso) ++h->steps_host;
        launch_e1<MODE_A>(h);
        launch_n1(h);
        launch_e2(h);
        if (h->p.is_using_mixed_stress && !h->iso) launch_n2(h);
        launch_force_pass(h);
        if (h->p.has_PT && !h->iso) {
            // the pseudo-transient loop is the caller's (des_dev.h): ghost refresh, phase 2, the residual across ranks, the
            // reference's test -- per iteration; phase 3 = the rest of this phase
            h->n_pt_iterations = 0;
            if (set_pt(h, 1)) return -DES_ERR_RESOURCE;
            return 2;
        }
        launch_s2(h, h->steps_host);
        launch_s3(h, true, false, false);
        return 0;
    case 2:
        pt_iteration(h);
        ++h->n_pt_iterations;
        return 0;
    case 3:
        if (set_pt(h, 0)) return -DES_ERR_RESOURCE;
        launch_vbcs_coord(h);                      // apply_vbcs + update_coordinate of the step itself
        launch_s2(h, h->steps_host);
        launch_s3(h, true, false, false);
        return 0;
    case 1: {
        launch_s3(h, false, true, true);
        if (h->iso) {
            launch_e1<MODE_C>(h);
            launch_mass_gather(h);
            return 0;
        }
        const bool do_dt = (h->steps_host % 10 == 0);
        launch_avg_coord0(h, h->steps_host);
        launch_e1_end(h, h->steps_host, false);
        launch_mass_gather(h);
        return do_dt ? 1 : 0;
    }
    }
    return -DES_ERR_INTERNAL;
}

// state records (what = 0: nodes, DES_X_NODE_WIDTH doubles each; 1: elements, DES_X_ELEM_WIDTH)
// of the listed local ids to / from a host buffer
static int state_io(des_dev *h, int what, const int *idx, int n, double *buf, bool pack)
{
    if (!h || what < 0 || what > 1) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    if (n == 0) return DES_OK;
    const size_t w = what == 0 ? DES_X_NODE_WIDTH : DES_X_ELEM_WIDTH;
    const std::vector<int> &map = what == 0 ? h->n_old2new : h->e_old2new;
    std::vector<int> midx(idx, idx + n), off((size_t)n);
    for (int k = 0; k < n; ++k) { if (!map.empty()) midx[k] = map[midx[k]]; off[k] = (int)(k * w); }
    int *d_idx = nullptr, *d_off = nullptr; double *d_buf = nullptr;
    int rc;
    if ((rc = dev_alloc(d_idx, (size_t)n)) || (rc = dev_alloc(d_off, (size_t)n)) || (rc = dev_alloc(d_buf, (size_t)n * w))) return rc;
    if ((rc = dev_upload(d_idx, midx.data(), (size_t)n, h->stream)) || (rc = dev_upload(d_off, off.data(), (size_t)n, h->stream))) return rc;
    const int nn_items = what == 0 ? n : 0, ne_items = what == 0 ? 0 : n;
    if (pack) {
        hipLaunchKernelGGL(k_state_pack, dim3(nblk(n)), dim3(DES_BLOCK), 0, h->stream, nn_items, d_idx, d_off, ne_items, d_idx, d_off,
                           h->xt, h->vm, h->dh_n, h->stress, pending_ddp(h), h->strain, h->plstrain, h->ne, d_buf);
        HIP_OK(hipMemcpyAsync(buf, d_buf, (size_t)n * w * 8, hipMemcpyDeviceToHost, h->stream));
    } else {
        if ((rc = dev_upload(d_buf, buf, (size_t)n * w, h->stream))) return rc;
        hipLaunchKernelGGL(k_state_unpack, dim3(nblk(n)), dim3(DES_BLOCK), 0, h->stream, nn_items, d_idx, d_off, ne_items, d_idx, d_off,
                           h->xt, h->vm, h->dh_n, h->stress, pending_ddp(h), h->strain, h->plstrain, h->ne, d_buf);
    }
    HIP_OK(hipStreamSynchronize(h->stream));
    hipFree(d_idx); hipFree(d_off); hipFree(d_buf);
    return DES_OK;
}

int des_dev_halo_pack(des_dev *h, int what, const int *idx, int n, double *buf)
{
    D2_FORWARD(h, halo_pack(h->d2, what, idx, n, buf));
    return state_io(h, what, idx, n, buf, true);
}

int des_dev_halo_unpack(des_dev *h, int what, const int *idx, int n, const double *buf)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    D2_FORWARD(h, halo_unpack(h->d2, what, idx, n, buf));
    return state_io(h, what, idx, n, const_cast<double *>(buf), false);
}

// What apply_vbcs reads off the WHOLE mesh in a 2-D model (bc.cxx:251-290, 350-361: the x0 wall's vertical extent, the
// lowest node), as this rank's mesh has it / as the cross-rank MAX gave it -- once after every exchange and before
// des_dev_init_geometry.  A 3-D model has nothing of the kind: zeros out, nothing in.
int des_dev_wall_get(des_dev *h, double out[3])
{
    if (!h || !out) return DES_ERR_INTERNAL;
    D2_FORWARD(h, wall_get(h->d2, out));
    out[0] = out[1] = out[2] = 0.0;
    return DES_OK;
}

int des_dev_wall_set(des_dev *h, const double in[3])
{
    if (!h || !in) return DES_ERR_INTERNAL;
    D2_FORWARD(h, wall_set(h->d2, in));
    return DES_OK;
}

int des_dev_dt_partials(des_dev *h, double out[6], int recompute)
{
    if (!h) return DES_ERR_INTERNAL;
    D2_FORWARD(h, dt_partials(h->d2, out, recompute));
    hipSetDevice(h->device);
    if (recompute) { refresh_props(h); launch_e1<MODE_DT>(h); }
    hipLaunchKernelGGL(k_dt_pack, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->d_red, h->dt_part, h->dt_part_cap,
                       h->dt_parts_used);
    h->dt_parts_used = 0;
    HIP_OK(hipMemcpyAsync(out, h->d_red, 48, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int des_dev_dt_finalize(des_dev *h, const double in[6], double *dt)
{
    if (h) h->finished = false;            // (fresh_ok: the next des_dev_step call starts with the classic first step)
    if (!h) return DES_ERR_INTERNAL;
    D2_FORWARD(h, dt_finalize(h->d2, in, dt));
    hipSetDevice(h->device);
    HIP_OK(hipMemcpyAsync(h->d_red, in, 48, hipMemcpyHostToDevice, h->stream));
    launch_dt_finalize(h, h->d_red);
    int rc = sync_clock(h);
    if (rc) return rc;
    if (dt) *dt = h->h_clk->dt;
    return h->h_clk->status;
}

int des_dev_libm_eval(int device, int fn, long long n, const double *x, const double *y, double *out)
{
    if (fn < DES_LIBM_POW || fn > DES_LIBM_SINCOS_C || n < 0 || !x || !out) return DES_ERR_INTERNAL;
    if ((fn == DES_LIBM_POW || fn == DES_LIBM_ATAN2) && !y) return DES_ERR_INTERNAL;
    if (des_dev_device_count() <= device) { g_last_error = "no such HIP device"; return DES_ERR_UNSUPPORTED; }
    if (n == 0) return DES_OK;
    HIP_OK(hipSetDevice(device));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    int rc = DES_OK;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == DES_OK) { rc = DES_ERR_RESOURCE; g_last_error = hipGetErrorString(e); } return e == hipSuccess; };
    if (ok(hipMalloc((void **)&dx, n * 8)) && ok(hipMalloc((void **)&dout, n * 8)) && (!y || ok(hipMalloc((void **)&dy, n * 8)))
        && ok(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice)) && (!y || ok(hipMemcpy(dy, y, n * 8, hipMemcpyHostToDevice)))) {
        hipLaunchKernelGGL(k_libm_eval, dim3((unsigned)((n + DES_BLOCK - 1) / DES_BLOCK)), dim3(DES_BLOCK), 0, 0, fn, n, dx, dy, dout);
        ok(hipGetLastError());
        ok(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
    }
    hipFree(dx); hipFree(dy); hipFree(dout);
    return rc;
}

int des_dev_eigen_eval(int device, int fn, int libm, long long n, const double *a, double *w, double *q, int *branch)
{
    if (fn < DES_EIG_DSYEVC3 || fn > DES_EIG_DSYEVQ3 || libm < 0 || libm > 1 || n < 0 || !a || !w || (fn != DES_EIG_DSYEVC3 && !q))
        return DES_ERR_INTERNAL;
    if (des_dev_device_count() <= device) { g_last_error = "no such HIP device"; return DES_ERR_UNSUPPORTED; }
    if (n == 0) return DES_OK;
    HIP_OK(hipSetDevice(device));
    DiagBuf b;
    const double *da = b.in(a, (size_t)n * 6);
    double *dw = b.out<double>((size_t)n * 3), *dq = b.out<double>((size_t)n * 9);
    int *dbr = b.out<int>((size_t)n);
    if (b.rc) return b.rc;
    const dim3 grid((unsigned)((n + DES_BLOCK - 1) / DES_BLOCK));
    if (libm) hipLaunchKernelGGL(k_eigen_eval<desk::MathPortable>, grid, dim3(DES_BLOCK), 0, 0, fn, n, da, dw, dq, dbr);
    else      hipLaunchKernelGGL(k_eigen_eval<desk::MathOcml>, grid, dim3(DES_BLOCK), 0, 0, fn, n, da, dw, dq, dbr);
    if (hipGetLastError() != hipSuccess) b.fail("kernel launch");
    b.back(w, dw, (size_t)n * 3);
    if (fn != DES_EIG_DSYEVC3) b.back(q, dq, (size_t)n * 9);
    b.back(branch, dbr, (size_t)n);
    return b.rc;
}

int des_dev_elasto_plastic_eval(int device, int libm, long long n, const double *props, const double *de,
                                double *s, double *depls, int *mode)
{
    if (libm < 0 || libm > 1 || n < 0 || !props || !de || !s || !depls) return DES_ERR_INTERNAL;
    if (des_dev_device_count() <= device) { g_last_error = "no such HIP device"; return DES_ERR_UNSUPPORTED; }
    if (n == 0) return DES_OK;
    HIP_OK(hipSetDevice(device));
    DiagBuf b;
    const double *dp = b.in(props, (size_t)n * 7), *dde = b.in(de, (size_t)n * 6);
    double *ds = b.in(s, (size_t)n * 6), *ddp = b.out<double>((size_t)n);
    int *dm = b.out<int>((size_t)n);
    if (b.rc) return b.rc;
    const dim3 grid((unsigned)((n + DES_BLOCK - 1) / DES_BLOCK));
    if (libm) hipLaunchKernelGGL(k_elasto_plastic_eval<desk::MathPortable>, grid, dim3(DES_BLOCK), 0, 0, n, dp, dde, ds, ddp, dm);
    else      hipLaunchKernelGGL(k_elasto_plastic_eval<desk::MathOcml>, grid, dim3(DES_BLOCK), 0, 0, n, dp, dde, ds, ddp, dm);
    if (hipGetLastError() != hipSuccess) b.fail("kernel launch");
    b.back(s, ds, (size_t)n * 6);
    b.back(depls, ddp, (size_t)n);
    b.back(mode, dm, (size_t)n);
    return b.rc;
}

int des_dev_check_nan(des_dev *h, long long *n_nan)
{
    D2_FORWARD(h, check_nan(h->d2, n_nan));
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    unsigned long long *d_count = nullptr;
    HIP_OK(hipMalloc((void **)&d_count, 8));
    HIP_OK(hipMemsetAsync(d_count, 0, 8, h->stream));
    const long long nn = h->nn, ne = h->ne;
    struct { const double *p; long long n; } arrs[] = {
        {h->volume, ne}, {h->dpressure, ne}, {h->viscosity, ne}, {h->stress, 6*ne}, {h->tmass, nn},
        {h->force, 3*nn}, {(const double *)h->xt, 4*nn}, {(const double *)h->vm, 4*nn} };
    for (auto &a : arrs)
        hipLaunchKernelGGL(k_count_nan, dim3(std::min(nblk(a.n), 2048)), dim3(DES_BLOCK), 0, h->stream, a.p, a.n, d_count);
    unsigned long long c = 0;
    HIP_OK(hipMemcpyAsync(&c, d_count, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    hipFree(d_count);
    if (n_nan) *n_nan = (long long)c;
    return c ? DES_ERR_RUNTIME_NAN : DES_OK;
}

int des_dev_mesh_quality(des_dev *h, double smallest_vol, double bottom, double bottom_dist, des_quality *out)
{
    D2_FORWARD(h, mesh_quality(h->d2, smallest_vol, bottom, bottom_dist, out));
    if (!h || !out) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    struct Slots { double q; int i[4]; } init = { 1.0, { INT_MAX, INT_MAX, INT_MAX, 0 } }, res;
    Slots *d = nullptr;
    HIP_OK(hipMalloc((void **)&d, sizeof(Slots)));
    HIP_OK(hipMemcpyAsync(d, &init, sizeof(Slots), hipMemcpyHostToDevice, h->stream));
    const int n = std::max(h->ne, h->nn);
    hipLaunchKernelGGL(k_quality_a, dim3(nblk(n)), dim3(DES_BLOCK), 0, h->stream, h->ne, h->nn, h->conn, h->xt, h->volume,
                       h->bcflag, smallest_vol, bottom, bottom_dist, &d->q, d->i, h->d_n_new2old, h->d_e_new2old);
    hipLaunchKernelGGL(k_quality_b, dim3(nblk(h->ne)), dim3(DES_BLOCK), 0, h->stream, h->ne, h->conn, h->xt, h->volume,
                       &d->q, d->i, h->d_e_new2old);
    HIP_OK(hipMemcpyAsync(&res, d, sizeof(Slots), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    hipFree(d);
    out->small_elem = res.i[0] == INT_MAX ? -1 : res.i[0];
    out->bottom_node = res.i[1] == INT_MAX ? -1 : res.i[1];
    out->worst_elem = res.i[2] == INT_MAX ? 0 : res.i[2];      // no element below quality 1: the reference keeps 0
    out->worst_quality = res.q; out->pad_ = 0;
    return DES_OK;
}

// *gbs = the best of a few launch shapes of that copy, (bytes read + bytes written) / time
int des_dev_copy_ceiling(int device, long long bytes, int reps, double *gbs)
{
    if (!gbs || bytes < 16 || reps < 1) return DES_ERR_INTERNAL;
    if (des_dev_device_count() <= device) { g_last_error = "no such HIP device"; return DES_ERR_UNSUPPORTED; }
    HIP_OK(hipSetDevice(device));
    const size_t n = (size_t)bytes / 16;
    double2 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DES_OK;
    *gbs = 0;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == DES_OK) { rc = DES_ERR_RESOURCE; g_last_error = hipGetErrorString(e); } return e == hipSuccess; };
    if (ok(hipMalloc((void **)&a, n * 16)) && ok(hipMalloc((void **)&b, n * 16)) && ok(hipMemset(a, 1, n * 16)) && ok(hipMemset(b, 0, n * 16))
        && ok(hipEventCreate(&e0)) && ok(hipEventCreate(&e1))) {
        for (int variant = 0; variant < 7 && rc == DES_OK; ++variant) {
            const unsigned per_cu[3] = {4, 8, 16};
            const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * per_cu[variant % 3]);
            if (variant == 6 && (n + 255) / 256 > 0x7fffffffull) break;
            auto launch = [&]() {
                if (variant < 3)      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy16<4, false>), dim3(grid), dim3(256), 0, 0, a, b, n);
                else if (variant < 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy16<4, true>), dim3(grid), dim3(256), 0, 0, a, b, n);
                else                  hipLaunchKernelGGL(k_copy16_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n);   // one item per lane
            };
            launch();                                                                     // warm-up
            ok(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) launch();
            ok(hipEventRecord(e1, 0));
            ok(hipEventSynchronize(e1));
            float ms = 0;
            if (ok(hipEventElapsedTime(&ms, e0, e1)) && ms > 0) *gbs = std::max(*gbs, 2.0 * n * 16 * reps / (ms * 1e-3) / 1e9);
        }
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    return rc;
}

int des_dev_plane_ceiling(int device, int nr, int nw, long long nelem, int reps, double *gbs)
{
    if (!gbs || nelem < 256 || nelem > (1ll << 31) * 255 || reps < 1) return DES_ERR_INTERNAL;
    if (!((nr == 18 && nw == 15) || (nr == 12 && nw == 9))) { g_last_error = "plane ceiling: shapes 18 + 15 and 12 + 9 only"; return DES_ERR_UNSUPPORTED; }
    if (des_dev_device_count() <= device) { g_last_error = "no such HIP device"; return DES_ERR_UNSUPPORTED; }
    HIP_OK(hipSetDevice(device));
    const size_t n = (size_t)nelem;
    double *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DES_OK;
    *gbs = 0;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == DES_OK) { rc = DES_ERR_RESOURCE; g_last_error = hipGetErrorString(e); } return e == hipSuccess; };
    if (ok(hipMalloc((void **)&a, n * 8 * nr)) && ok(hipMalloc((void **)&b, n * 8 * nw)) && ok(hipMemset(a, 0, n * 8 * nr))
        && ok(hipMemset(b, 0, n * 8 * nw)) && ok(hipEventCreate(&e0)) && ok(hipEventCreate(&e1))) {
        const dim3 grid((unsigned)((n + 255) / 256));
        auto launch = [&]() {
            if (nr == 18) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_plane_stream<18, 15>), grid, dim3(256), 0, 0, a, b, nelem);
            else          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_plane_stream<12, 9>), grid, dim3(256), 0, 0, a, b, nelem);
        };
        launch();                                                                         // warm-up
        ok(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch();
        ok(hipEventRecord(e1, 0));
        ok(hipEventSynchronize(e1));
        float ms = 0;
        if (ok(hipEventElapsedTime(&ms, e0, e1)) && ms > 0) *gbs = (double)n * 8 * (nr + nw) * reps / (ms * 1e-3) / 1e9;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    return rc;
}

// items: lanes of the launch.  Bytes per launch: pattern 0 reads 16 and writes 16 per item; 1 reads 8,
// writes 8; 2 reads 4 (index) + 32 (record), writes 8; 3 reads 4 + 8, writes 8 (gathers touch every
// record exactly once, in a random order that defeats coalescing but not the caches' line reuse).
int des_dev_access_bench(int device, int pattern, long long items, int reps, double *ms_per_launch)
{
    if (pattern < 0 || pattern > 3 || items < 1 || items > (1ll << 31) - 1 || reps < 1) return DES_ERR_INTERNAL;
    if (des_dev_device_count() <= device) { g_last_error = "no such HIP device"; return DES_ERR_UNSUPPORTED; }
    HIP_OK(hipSetDevice(device));
    const size_t n = (size_t)items;
    const size_t src_doubles = pattern == 0 ? 2 * n : pattern == 2 ? 4 * n : n, dst_doubles = pattern == 0 ? 2 * n : n;
    DiagBuf b;
    double *src = b.out<double>(src_doubles), *dst = b.out<double>(dst_doubles);
    int *perm = nullptr;
    if (b.rc) return b.rc;
    if (pattern >= 2) {
        std::vector<int> p(n);
        for (size_t i = 0; i < n; ++i) p[i] = (int)i;
        unsigned long long st = 88172645463325252ull;                      // xorshift: a fixed shuffle
        for (size_t i = n - 1; i > 0; --i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            std::swap(p[i], p[st % (i + 1)]);
        }
        perm = b.in(p.data(), n);
        if (b.rc) return b.rc;
    }
    HIP_OK(hipMemset(src, 0, src_doubles * 8));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((n + 255) / 256));
    auto launch = [&]() {
        switch (pattern) {
        case 0: hipLaunchKernelGGL(k_access<0>, grid, dim3(256), 0, 0, src, dst, perm, n); break;
        case 1: hipLaunchKernelGGL(k_access<1>, grid, dim3(256), 0, 0, src, dst, perm, n); break;
        case 2: hipLaunchKernelGGL(k_access<2>, grid, dim3(256), 0, 0, src, dst, perm, n); break;
        default: hipLaunchKernelGGL(k_access<3>, grid, dim3(256), 0, 0, src, dst, perm, n); break;
        }
    };
    launch();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms_per_launch) *ms_per_launch = ms / reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? DES_OK : DES_ERR_RESOURCE;
}

int des_dev_timer_start(des_dev *h)
{
    D2_FORWARD(h, timer_start(h->d2));
    if (!h) return DES_ERR_INTERNAL;
    HIP_OK(hipEventRecord(h->ev0, h->stream));
    return DES_OK;
}

int des_dev_timer_stop(des_dev *h, float *ms)
{
    D2_FORWARD(h, timer_stop(h->d2, ms));
    if (!h) return DES_ERR_INTERNAL;
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    HIP_OK(hipEventSynchronize(h->ev1));
    float t = 0;
    HIP_OK(hipEventElapsedTime(&t, h->ev0, h->ev1));
    if (ms) *ms = t;
    return DES_OK;
}

int des_dev_profile_enable(des_dev *h, int on)
{
    D2_FORWARD(h, profile_enable(h->d2, on));
    if (!h) return DES_ERR_INTERNAL;
    hipStreamSynchronize(h->stream);
    for (ProfRec &r : h->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    h->prof_recs.clear();
    for (int k = 0; k < K_COUNT; ++k) { h->prof_ms[k] = 0; h->prof_calls[k] = 0; }
    h->prof = on != 0;
    return DES_OK;
}

int des_dev_profile_read(des_dev *h, int cap, char (*names)[64], double *ms, long long *calls)
{
    if (h && h->d2) return des2d::profile_read(h->d2, cap, names, ms, calls);
    if (!h) return 0;
    hipStreamSynchronize(h->stream);
    for (ProfRec &r : h->prof_recs) {
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { h->prof_ms[r.k] += t; h->prof_calls[r.k] += 1; }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    h->prof_recs.clear();
    int n = 0;
    for (int k = 0; k < K_COUNT && n < cap; ++k) {
        if (h->prof_calls[k] == 0) continue;
        std::strncpy(names[n], kKernelNames[k], 63); names[n][63] = 0;
        ms[n] = h->prof_ms[k]; calls[n] = h->prof_calls[k];
        ++n;
    }
    return n;
}

// B_alg = 1420*ne + 348*nn; evp +24*ne +8*nn; thermal off -88*ne -24*nn;
// NMD off -96*ne -28*nn
double des_dev_algorithmic_bytes_per_step(const des_dev *h)
{
    if (h && h->d2) return des2d::algorithmic_bytes_per_step(h->d2);
    double be = 1420, bn = 348;
    if (h->p.rheol_type == DES_RH_EVP) { be += 24; bn += 8; }
    if (!h->p.has_thermal_diffusion) { be -= 88; bn -= 24; }
    if (!h->p.is_using_mixed_stress) { be -= 96; bn -= 28; }
    // average_fields: stress_avg read+write, delta_plstrain read, dplstrain_avg read+write
    if (h->p.is_outputting_averaged_fields) be += 120;
    return be * h->ne + bn * h->nn;
}

#ifdef DES_STAMPS
// instrumented builds only (passes/common.hpp): the stamps of the last EN1 (pass 0) / EN3 (pass 1) launch
int des_dev_debug_stamps(int pass, unsigned long long *out, int cap)
{
    const size_t n = (size_t)DES_STAMP_SLOTS * DES_STAMP_WG;
    if (!out || pass < 0 || pass > 3 || (size_t)cap < n) return (int)n;
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(des_hip::g_stamps), n * sizeof(unsigned long long), (size_t)pass * n * sizeof(unsigned long long),
                            hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)n;
}
#endif
} // extern "C"
