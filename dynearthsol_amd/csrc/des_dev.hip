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
    double avg_time0;        // Output::time0 (output.cxx:332)
    int n_defer;             // elements the first stress pass of this step handed to E2_return_mapping
    int pad;
};

// INIT: C part without rotate_stress; AVG: Output::average_fields on the final stress of the step
enum { MODE_A = 1, MODE_C = 2, MODE_DT = 4, MODE_INIT = 8 };

enum KernelId { K_E1, K_N1, K_E2, K_E2R, K_N2, K_E3, K_N3, K_S2, K_S3,
                K_DTFIN, K_MISC, K_COUNT };
const char *kKernelNames[K_COUNT] = {
    "E1_geom_rotate_strainrate", "N1_mass_temperature_dvoldt", "E2_update_stress", "E2_return_mapping", "N2_nmd_gather",
    "E3_nmd_force", "N3_force_velocity_coord", "S2_surface_diffusion",
    "S3_edvacc_step_finalize", "dt_finalize", "misc" };

struct ProfRec { int k; hipEvent_t a, b; };

} // namespace

struct des_dev {
    int device;
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
    int e2_defer;                         // DES_E2_DEFER: 0 one pass, 1 two passes, 2 (default) chosen per call
    bool e2_two_pass;                     // the current choice
    int *mono;                            // [ne] (material << 16) | count of single-material elements, else -1
    double *ptab;                         // [nmat][DES_PTAB_CNT][5] property means of single-material elements
    unsigned char *topflag;               // element touches the top surface (Variables::top_elems)
    double *props;                        // [5][ne] bulkm, shearm, phi, cp, k  (nmat > 1 only)
    // temporaries
    d4 *mrec, *ttmp;                      // {vol, m, tm, dvol}, thermal tr[4]
    double *etmp2, *ftmp;                 // dp*vol ; force tr [ne][4][3]
    double *res_part;                     // per-block partial sums of the residual
    int n3_blocks;
    int npb;                              // nodes per node-kernel workgroup (choose_npb)
    // stress-bc lists
    int nbcf;                             // facets with a stress bc (incl. neumann)
    int *bcf_elem, *bcf_facet, *bcf_kind; // kind: 0 winkler, 1 water, 2 side wall, 3+d neumann dir d
    double *bcf_val;                      // neumann value per facet
    double *bcf_tmp;                      // [nbcf][9]
    int *bcn_idx, *bcn_ent;               // per-node CSR of entries (facet*3+l)*2 + is_neumann
    unsigned bc_mask;                     // bcflag bits that have any entry
    // surface
    int ntop, etop, ntop_elems;
    int *top_nodes, *ean, *conn_surf, *ssup_idx, *ssup_arr;
    double *dh, *edvacc, *znew;           // znew: surface heights between k_s2 and their commit
    // bnormals / edges for slanted boundaries
    double *bnormals, *edge_vec; int *edge_slot;

    // domain decomposition (des_halo): owned nodes [o0, o1), halo lists, exchange buffers
    int o0, o1, nn_global;
    int nnbr;
    std::vector<int> nbr_rank, send_ptr, recv_ptr;         // host copies of the list offsets
    int *d_send_idx, *d_recv_idx;
    double *d_sendbuf, *d_recvbuf;                         // one message per neighbour: node records, then element records
    double *d_red;                                         // 8 doubles: dt partials / scalar reductions
    double *dh_n;                                          // nodal copy of surfinfo.dh
    ncclComm_t comm;
    int comm_rank, comm_size;
    // element part of the exchange lists and the record offsets inside the message buffers
    std::vector<int> esend_ptr, erecv_ptr;
    std::vector<long long> send_off, recv_off;             // [nnbr+1] message offsets (doubles) per neighbour
    int *d_esend_idx, *d_erecv_idx, *d_send_noff, *d_send_eoff, *d_recv_noff, *d_recv_eoff;
    // internal data order (des_mesh::coord hint): device index <-> caller's index; empty = identity
    std::vector<int> n_new2old, n_old2new, e_new2old, e_old2new;
    int *d_n_new2old, *d_e_new2old;
    bool markers_dirty;
    bool const_mass;                      // quasi-static, one material: nodal mass from volumes alone
    bool pending_c;                       // C part of the last step has been run (always true outside step())
    long long steps_host;
    bool iso;                 // des_dev_set_isostasy
    // profiling
    bool prof;
    std::vector<ProfRec> prof_recs;
    double prof_ms[K_COUNT]; long long prof_calls[K_COUNT];
};

namespace des_hip {

// =====================================================================================
// kernels
// =====================================================================================
struct ElemProps { double bulkm, shearm, phi, cp, k; };

// What the kernels know about the materials of an element (refresh_elem_cache,
// matprops.cxx:259-303, redone whenever the marker counts change):
//   markers [ne][nmat]  the counts themselves
//   mono    [ne]        (material << 16) | count where one material holds every marker, else -1:
//                       4 bytes per element and pass instead of 4*nmat + 40
//   props   [5][ne]     bulkm, shearm, phi, cp, k of every element (nmat > 1 only)
//   ptab    [nmat][DES_PTAB_CNT][5]  the same five means for a single-material element with
//                       `count` markers (the means are count-dependent in the last bit:
//                       count / (count / s)), so those elements read a cached table row
#define DES_PTAB_CNT 64
struct MatData { const int *markers; const int *mono; const double *props; const double *ptab; };

__device__ __forceinline__ desk::Mix mix_of(const MatData &md, int nmat, int e)
{
    const int mo = md.mono[e];
    desk::Mix mx;
    if (mo >= 0) { mx.mk = nullptr; mx.mat = mo >> 16; mx.cnt = mo & 0xffff; }
    else         { mx.mk = md.markers + (size_t)e * nmat; mx.mat = -1; mx.cnt = 0; }
    return mx;
}

__device__ __forceinline__ ElemProps load_props(const des_params *p, const MatData &md, const desk::Mix &mx, int ne, int e)
{
    ElemProps r;
    if (!md.props) {                                       // nmat == 1: the means are the values (matprops.cxx:118, 136)
        r.bulkm = p->bulk_modulus[0]; r.shearm = p->shear_modulus[0]; r.phi = p->porosity[0];
        r.cp = p->heat_capacity[0]; r.k = p->therm_cond[0];
    } else if (!mx.mk && mx.cnt < DES_PTAB_CNT) {
        const double *t = md.ptab + ((size_t)mx.mat * DES_PTAB_CNT + mx.cnt) * 5;
        r.bulkm = t[0]; r.shearm = t[1]; r.phi = t[2]; r.cp = t[3]; r.k = t[4];
    } else {
        r.bulkm = md.props[e]; r.shearm = md.props[(size_t)ne + e]; r.phi = md.props[(size_t)2*ne + e];
        r.cp = md.props[(size_t)3*ne + e]; r.k = md.props[(size_t)4*ne + e];
    }
    return r;
}

// Young's-modulus "mass" of an element, only read by damping option 4 (geometry.cxx:1832)
__device__ __forceinline__ double elem_ym(const des_params *p, const MatData &md, int ne, int e)
{
    const ElemProps pr = load_props(p, md, mix_of(md, p->nmat, e), ne, e);
    return 9 * pr.bulkm * pr.shearm / (3 * pr.bulkm + pr.shearm) / 4;
}

// refresh_elem_cache (matprops.cxx:259-303): mono[] for every element, props[] for nmat > 1
__global__ void __launch_bounds__(DES_BLOCK)
k_props(const des_params *p, const int *markers, double *props, int *mono, int ne)
{
    int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    const int nmat = p->nmat;
    const int *mk = markers + (size_t)e * nmat;
    int used = 0, mat = 0, cnt = 0;
    for (int m = 0; m < nmat; ++m) if (mk[m] != 0) { ++used; mat = m; cnt = mk[m]; }
    mono[e] = (used == 1 && cnt > 0 && cnt < 65536) ? ((mat << 16) | cnt) : -1;
    if (!props) return;
    props[e]                = desk::harmonic_mean(p->bulk_modulus, mk, nmat);
    props[(size_t)ne + e]   = desk::harmonic_mean(p->shear_modulus, mk, nmat);
    props[(size_t)2*ne + e] = desk::arithmetic_mean(p->porosity, mk, nmat);
    props[(size_t)3*ne + e] = desk::arithmetic_mean(p->heat_capacity, mk, nmat);
    props[(size_t)4*ne + e] = desk::arithmetic_mean(p->therm_cond, mk, nmat);
}

// the five means of a single-material element, by (material, marker count)
__global__ void k_ptab(const des_params *p, double *ptab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nmat = p->nmat;
    if (i >= nmat * DES_PTAB_CNT) return;
    const int mat = i / DES_PTAB_CNT, cnt = i % DES_PTAB_CNT;
    int mk[DES_MAX_MAT];
    for (int m = 0; m < DES_MAX_MAT; ++m) mk[m] = 0;
    mk[mat] = cnt > 0 ? cnt : 1;                           // row 0 is never read
    double *t = ptab + (size_t)i * 5;
    t[0] = desk::harmonic_mean(p->bulk_modulus, mk, nmat);
    t[1] = desk::harmonic_mean(p->shear_modulus, mk, nmat);
    t[2] = desk::arithmetic_mean(p->porosity, mk, nmat);
    t[3] = desk::arithmetic_mean(p->heat_capacity, mk, nmat);
    t[4] = desk::arithmetic_mean(p->therm_cond, mk, nmat);
}

// ---- E1 --------------------------------------------------------------------------
// MODE_C: compute_volume (geometry.cxx:170-201) after the volume swap (dynearthsol.cxx:466-470),
//         compute_mass element part (geometry.cxx:1795-1840), rotate_stress (fields.cxx:827-902);
//         MODE_DT adds the compute_dt reduction (geometry.cxx:1513-1593).
// MODE_A: update_temperature element part (fields.cxx:211-239), update_strain_rate
//         (fields.cxx:415-476), compute_dvoldt element part (geometry.cxx:218-224).
template <int MODE>
__global__ void __launch_bounds__(DES_BLOCK, DES_E1_WAVES)
E1_geom_rotate_strainrate(const des_params *__restrict__ p, DevClock *__restrict__ clk, int ne, int nblocks,
     const int4 *__restrict__ conn, const d4 *__restrict__ xt, const d4 *__restrict__ vm,
     const MatData md,
     const double *__restrict__ radiogenic, const unsigned char *__restrict__ topflag,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ plstrain,
     double *__restrict__ volume, double *__restrict__ volume_old,
     double *__restrict__ strain_rate, d4 *__restrict__ mrec, d4 *__restrict__ ttmp)
{
    const int e = desk::logical_block(nblocks) * DES_BLOCK + threadIdx.x;
    const bool active = e < ne;

    double r_minl = DBL_MAX, r_maxw = DBL_MAX, r_diff = DBL_MAX, r_gdt = DBL_MAX, r_vem = 0.0;

    if (active) {
        const int4 cn = conn[e];
        d4 c[4], v[4];
        c[0] = xt[cn.x]; c[1] = xt[cn.y]; c[2] = xt[cn.z]; c[3] = xt[cn.w];
        v[0] = vm[cn.x]; v[1] = vm[cn.y]; v[2] = vm[cn.z]; v[3] = vm[cn.w];
        const desk::Mix mx = mix_of(md, p->nmat, e);
        const ElemProps pr = load_props(p, md, mx, ne, e);

        // mean nodal temperature, matprops.cxx:338-343
        double T = 0;
        T += c[0].w; T += c[1].w; T += c[2].w; T += c[3].w;
        T /= 4;
        const double rho = desk::mat_rho(p, mx, T);

        double vol;
        d4 rec;
        double rdv = 0.0;                        // >= 1: correct_surface_element rescales this element
        // update_mesh only runs on a moving mesh (dynearthsol.cxx:870-873; always in the isostasy
        // loop): without it volumes and masses keep their values, rotate_stress below still runs
        const bool remesh_geom = (MODE & MODE_INIT) || p->has_moving_mesh || clk->iso;
        if ((MODE & MODE_C) && !remesh_geom) {
            vol = volume[e];
            const d4 old = mrec[e];
            rec.x = old.x; rec.y = old.y; rec.z = old.z;
        } else if (MODE & MODE_C) {
            const double vol_prev = volume[e];
            vol = desk::tet_volume(c);
            if (!(MODE & MODE_INIT) && topflag[e]) {
                // correct_surface_element (bc.cxx:1670-1687) runs before the swap: it already
                // stored the new volume, so the swap moves the NEW volume into volume_old
                rdv = vol / vol_prev;
                volume_old[e] = vol;
            } else {
                volume_old[e] = vol_prev;        // pointer swap of dynearthsol.cxx:466-470
            }
            volume[e] = vol;
            // compute_mass, element part
            const double pseudo_speed = p->max_vbc_val * p->inertial_scaling;
            double rho_m = p->is_quasi_static ? pr.bulkm / (pseudo_speed * pseudo_speed) : rho;
            rec.x = vol;
            rec.y = rho_m * vol / 4;
            rec.z = rho * pr.cp * vol / 4;
        } else {
            vol = volume[e];
            if (MODE & MODE_A) {
                const d4 old = mrec[e];
                rec.x = old.x; rec.y = old.y; rec.z = old.z;
            }
        }

        double sx[4], sy[4], sz[4];
        desk::shape_fn(c, vol, sx, sy, sz);

        if ((MODE & MODE_C) && !(MODE & MODE_INIT)) {
            const bool rescale = rdv >= 1.0;                         // bc.cxx:1677
            const bool rotate = (p->rheol_type & DES_RH_ELASTIC) != 0 && !clk->iso;   // not in the isostasy loop
            if (rescale || rotate) {
                double s[6], es[6];
                for (int i = 0; i < 6; ++i) { s[i] = stress[(size_t)i*ne + e]; es[i] = strain[(size_t)i*ne + e]; }
                if (rescale) {
                    plstrain[e] /= rdv;
                    for (int i = 0; i < 6; ++i) { s[i] /= rdv; es[i] /= rdv; }
                    if (!(MODE & MODE_A))            // otherwise update_strain_rate overwrites it below
                        for (int i = 0; i < 6; ++i) strain_rate[(size_t)i*ne + e] /= rdv;
                }
                if (rotate) {
                    const double dt = clk->dt;
                    double w3 = 0, w4 = 0, w5 = 0;
                    for (int i = 0; i < 4; ++i) w3 += 0.5 * (v[i].x * sy[i] - v[i].y * sx[i]);
                    for (int i = 0; i < 4; ++i) w4 += 0.5 * (v[i].x * sz[i] - v[i].z * sx[i]);
                    for (int i = 0; i < 4; ++i) w5 += 0.5 * (v[i].y * sz[i] - v[i].z * sy[i]);
                    desk::jaumann_rate_3d(s, dt, w3, w4, w5);
                    desk::jaumann_rate_3d(es, dt, w3, w4, w5);
                }
                if (rescale || rotate)
                    for (int i = 0; i < 6; ++i) { stress[(size_t)i*ne + e] = s[i]; strain[(size_t)i*ne + e] = es[i]; }
            }
        }

        if (MODE & MODE_DT) {
            double vx = 0.0, vy = 0.0, vz = 0.0;
            const double weight = 1.0 / 4;
            for (int j = 0; j < 4; ++j) { vx += v[j].x * weight; vy += v[j].y * weight; vz += v[j].z * weight; }
            r_vem = sqrt(vx*vx + vy*vy + vz*vz);
            double maxa = fmax(fmax(desk::tri_area(c[0], c[1], c[2]), desk::tri_area(c[0], c[1], c[3])),
                               fmax(desk::tri_area(c[2], c[3], c[0]), desk::tri_area(c[2], c[3], c[1])));
            double minh = 3 * vol / maxa;
            r_maxw = 0.5 * p->visc_min / (1e-40 + pr.shearm);
            if (p->has_thermal_diffusion) r_diff = 0.5 * minh * minh / p->therm_diff_max;
            r_minl = minh;
            r_gdt = minh / sqrt(pr.shearm / rho) / 5.0;
        }

        if (MODE & MODE_A) {
            if (p->has_thermal_diffusion) {
                double kv = pr.k * vol;
                double rh = radiogenic[e] * vol * rho / 4;
                d4 tr;
                double *trp = &tr.x;
                for (int i = 0; i < 4; ++i) {
                    double diffusion = 0.;
                    for (int j = 0; j < 4; ++j)
                        diffusion += (sx[i] * sx[j] + sy[i] * sy[j] + sz[i] * sz[j]) * c[j].w;
                    trp[i] = diffusion * kv - rh;
                }
                ttmp[e] = tr;
            }
            double s[6];
            s[0] = 0; for (int i = 0; i < 4; ++i) s[0] += v[i].x * sx[i];
            s[1] = 0; for (int i = 0; i < 4; ++i) s[1] += v[i].y * sy[i];
            s[2] = 0; for (int i = 0; i < 4; ++i) s[2] += v[i].z * sz[i];
            s[3] = 0; for (int i = 0; i < 4; ++i) s[3] += 0.5 * (v[i].x * sy[i] + v[i].y * sx[i]);
            s[4] = 0; for (int i = 0; i < 4; ++i) s[4] += 0.5 * (v[i].x * sz[i] + v[i].z * sx[i]);
            s[5] = 0; for (int i = 0; i < 4; ++i) s[5] += 0.5 * (v[i].y * sz[i] + v[i].z * sy[i]);
            for (int i = 0; i < 6; ++i) strain_rate[(size_t)i*ne + e] = s[i];
            double dj = s[0] + s[1] + s[2];
            rec.w = dj * vol;
        } else {
            rec.w = 0;
        }
        if (MODE & (MODE_C | MODE_A)) mrec[e] = rec;
    }

    if (MODE & MODE_DT) {
        __shared__ double red[5][DES_BLOCK / 64];
        r_minl = desk::wave_min(r_minl); r_maxw = desk::wave_min(r_maxw); r_diff = desk::wave_min(r_diff);
        r_gdt = desk::wave_min(r_gdt);   r_vem = desk::wave_max(r_vem);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) { red[0][w] = r_minl; red[1][w] = r_maxw; red[2][w] = r_diff; red[3][w] = r_gdt; red[4][w] = r_vem; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < DES_BLOCK / 64; ++i) {
                red[0][0] = fmin(red[0][0], red[0][i]); red[1][0] = fmin(red[1][0], red[1][i]);
                red[2][0] = fmin(red[2][0], red[2][i]); red[3][0] = fmin(red[3][0], red[3][i]);
                red[4][0] = fmax(red[4][0], red[4][i]);
            }
            desk::atomic_min_double(&clk->r_minl, red[0][0]);
            desk::atomic_min_double(&clk->r_dt_maxwell, red[1][0]);
            desk::atomic_min_double(&clk->r_dt_diffusion, red[2][0]);
            desk::atomic_min_double(&clk->r_global_dt_min, red[3][0]);
            desk::atomic_max_double(&clk->r_max_vem, red[4][0]);
        }
    }
}

// compute_dt tail (geometry.cxx:1597-1646); one thread
__global__ void k_dt_finalize(const des_params *p, DevClock *clk, const double *red)
{
    if (red) {          // partials min-reduced over the ranks (k_dt_pack layout)
        clk->r_minl = red[0]; clk->r_dt_maxwell = red[1]; clk->r_dt_diffusion = red[2];
        clk->r_global_dt_min = red[3]; clk->r_max_vem = -red[4]; clk->max_surf_vel = -red[5];
    }
    double dt_maxwell = clk->r_dt_maxwell, dt_diffusion = clk->r_dt_diffusion, minl = clk->r_minl;
    const double dt_hydro_diffusion = DBL_MAX;
    double global_max_vem = clk->r_max_vem;
    double max_vbc_val;
    if (p->characteristic_speed == 0) {
        max_vbc_val = p->max_vbc_val;
        if (p->surface_process_option > 0)
            max_vbc_val = fmax(max_vbc_val, clk->max_surf_vel * 5e-1);
    } else
        max_vbc_val = p->characteristic_speed;
    global_max_vem = fmax(global_max_vem, p->max_vbc_val);
    clk->max_global_vel_mag = global_max_vem;
    clk->global_dt_min = clk->r_global_dt_min;
    double dt_advection = 0.5 * minl / max_vbc_val;
    double dt_elastic = p->is_quasi_static
        ? 0.5 * minl / (max_vbc_val * p->inertial_scaling)
        : 0.5 * minl / sqrt(p->bulk_modulus[p->mattype_ref] / p->rho0[p->mattype_ref]);
    double dt = fmin(fmin(fmin(dt_elastic, dt_maxwell), fmin(dt_advection, dt_diffusion)), dt_hydro_diffusion)
                * p->dt_fraction;
    if (p->fixed_dt != 0) dt = p->fixed_dt;
    if (!(dt > 0)) clk->status = DES_ERR_RUNTIME_NAN;
    clk->dt = dt;
    clk->r_minl = DBL_MAX; clk->r_dt_maxwell = DBL_MAX; clk->r_dt_diffusion = DBL_MAX;
    clk->r_global_dt_min = DBL_MAX; clk->r_max_vem = 0.0;
}

// ---- node gathers: LDS-staged segmented reduction ---------------------------------
// A workgroup owns DES_BLOCK consecutive nodes, i.e. one contiguous range [kb, ke) of the
// CSR incidence list.  Phase 1: all lanes walk that range with stride DES_BLOCK -- the index
// loads are fully coalesced and every lane has the same number of independent record
// gathers in flight, whatever the valence of "its" node (8 or 32 on the regular mesh, 8-50
// on TetGen meshes) -- and park the gathered values in LDS.  Phase 2: each lane sums the
// slice of LDS that belongs to its node, sequentially and in ascending element order, which
// is the reference's summation order (fields.cxx:667-675) -> bit-identical sums.
// LDS slot of incidence j is skewed by j/8 so that row starts that are multiples of 8
// doubles apart (regular mesh) do not land on the same banks.
#ifndef DES_TILE_N1
#define DES_TILE_N1 768
#endif
#ifndef DES_TILE_N1C
#define DES_TILE_N1C 1024
#endif
#ifndef DES_TILE_N3
#define DES_TILE_N3 1024
#endif
#ifndef DES_TILE_N2
#define DES_TILE_N2 2048
#endif
// DES_PIPE: the record gathers of incidence tile t+1 are issued before the sums of tile t are
// formed from LDS (registers hold them across the sum), so HBM/L2 latency overlaps the LDS phase
// instead of alternating with it.
#ifndef DES_PIPE
#define DES_PIPE 1
#endif
__device__ __forceinline__ int lds_slot(int j) { return j + (j >> 3); }
#define DES_TILE_LDS(T) ((T) + (T) / 8 + 1)

// ---- N1 --------------------------------------------------------------------------
// compute_mass gather (geometry.cxx:1846-1864), update_temperature node loop
// (fields.cxx:245-262), compute_dvoldt gather (geometry.cxx:231-238).
// Also advances the clock: steps++, time += dt (dynearthsol.cxx:773-774).
// FULL = 0: compute_mass only (init_geometry and the end of a des_dev_step call).
// CONSTM = 1: quasi-static run with one material -- the inertial mass of an element is
// (K / pseudo_speed^2) * V / 4 with a run constant factor (geometry.cxx:1814-1816, 1829), so it
// is formed from the gathered volume instead of being gathered itself (one LDS plane less).
template <int FULL, int CONSTM>
__global__ void __launch_bounds__(DES_BLOCK)
N1_mass_temperature_dvoldt(const des_params *__restrict__ p, DevClock *__restrict__ clk, int o0, int nn, int nblocks, int npb,
     const int *__restrict__ sup_idx, const int *__restrict__ sup_pack, const unsigned *__restrict__ bcflag,
     const d4 *__restrict__ mrec, const d4 *__restrict__ ttmp, const MatData md, int ne,
     d4 *__restrict__ xt, d4 *__restrict__ vm, double *__restrict__ volume_n, double *__restrict__ tmass,
     double *__restrict__ ymass, double *__restrict__ ntmp)
{
    constexpr int TILE = CONSTM ? DES_TILE_N1C : DES_TILE_N1;
    constexpr int NPL = CONSTM ? 4 : 5;
    __shared__ double lds[NPL][DES_TILE_LDS(TILE)];
    // nodes [o0, nn) are this rank's owned nodes (the whole mesh on one GPU)
    // a workgroup owns `npb` consecutive nodes (256, or 64 on small meshes so that there are
    // enough workgroups: all 256 lanes still share the gather phase, the first npb do the sums)
    const int lb = desk::logical_block(nblocks);
    const int n0 = o0 + lb * npb;
    const int n = (threadIdx.x < npb) ? n0 + threadIdx.x : nn;
    const double dt = clk->dt;
    if (FULL && blockIdx.x == 0 && threadIdx.x == 0) {
        if (!clk->iso) {                                   // the isostasy loop does not advance the clock
            clk->steps += 1;
            clk->time += dt;
        }
        clk->maxdh = 0.0;
        clk->n_defer = 0;
    }
    if (n0 >= nn) return;                                   // whole block idle (grid padding)
    const bool thermal = p->has_thermal_diffusion;
    const bool need_ym = p->damping_option == 4;
    const double pseudo_speed = p->max_vbc_val * p->inertial_scaling;
    const double rho_m = p->bulk_modulus[0] / (pseudo_speed * pseudo_speed);
    const int nlast = min(n0 + npb, nn);
    const int kb = sup_idx[n0], ke = sup_idx[nlast];
    int r0 = ke, r1 = ke;
    if (n < nn) { r0 = sup_idx[n]; r1 = sup_idx[n+1]; }
    double vn = 0, ms = 0, tms = 0, acc = 0, tdot = 0, yms = 0;
#if DES_PIPE
    constexpr int PER = TILE / DES_BLOCK;
    static_assert(TILE % DES_BLOCK == 0, "tile must be a multiple of the block");
    d4 rr[PER]; double r3[PER];
    auto fetch = [&](int t0, int tn) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) {
                const int pk = sup_pack[t0 + j];
                const int e = pk >> 2;
                rr[u] = mrec[e];
                if (FULL && thermal) r3[u] = (&ttmp[e].x)[pk & 3];
                else if (need_ym)    r3[u] = elem_ym(p, md, ne, e);
            }
        }
    };
    if (kb < ke) fetch(kb, min(TILE, ke - kb));
#endif
    for (int t0 = kb; t0 < ke; t0 += TILE) {
        const int tn = min(TILE, ke - t0);
#if DES_PIPE
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) {
                const int sl = lds_slot(j);
                lds[0][sl] = rr[u].x; lds[1][sl] = rr[u].z; lds[2][sl] = rr[u].w;
                if ((FULL && thermal) || need_ym) lds[3][sl] = r3[u];
                if (!CONSTM) lds[NPL - 1][sl] = rr[u].y;
            }
        }
        __syncthreads();
        if (t0 + TILE < ke) fetch(t0 + TILE, min(TILE, ke - t0 - TILE));
#else
        for (int j = threadIdx.x; j < tn; j += DES_BLOCK) {
            const int pk = sup_pack[t0 + j];
            const int e = pk >> 2;
            const d4 r = mrec[e];
            const int sl = lds_slot(j);
            lds[0][sl] = r.x; lds[1][sl] = r.z; lds[2][sl] = r.w;
            if (FULL && thermal) lds[3][sl] = (&ttmp[e].x)[pk & 3];
            else if (need_ym)    lds[3][sl] = elem_ym(p, md, ne, e);
            if (!CONSTM) lds[NPL - 1][sl] = r.y;
        }
        __syncthreads();
#endif
        const int a = max(r0, t0) - t0, b = min(r1, t0 + tn) - t0;
        for (int j = a; j < b; ++j) {
            const int sl = lds_slot(j);
            const double vol = lds[0][sl];
            vn += vol;
            if (CONSTM) ms += rho_m * vol / 4;
            else        ms += lds[NPL - 1][sl];
            if (thermal) tms += lds[1][sl];
            if (FULL) {
                if (thermal) tdot += lds[3][sl];
                acc += lds[2][sl];
            }
            if (need_ym && !(FULL && thermal)) yms += lds[3][sl];
        }
        __syncthreads();
    }
    if (n >= nn) return;
    if (need_ym && FULL && thermal) {
        // damping option 4 together with thermal diffusion: the spare LDS plane is taken by
        // the conduction term, so the Young's-modulus mass is summed straight from memory
        for (int k = r0; k < r1; ++k) yms += elem_ym(p, md, ne, sup_pack[k] >> 2);
    }
    volume_n[n] = vn;
    tmass[n] = tms;
    if (need_ym) ymass[n] = yms;
    d4 m4 = vm[n];
    m4.w = ms;
    vm[n] = m4;
    if (FULL) {
        if (thermal && !clk->iso) {                        // the isostasy loop has no update_temperature
            d4 x4 = xt[n];
            if (bcflag[n] & (1u << 5))
                x4.w = p->surface_temperature;
            else
                x4.w -= dt * tdot / tms;
            xt[n] = x4;
        }
        ntmp[n] = acc / vn;
    }
}

// ---- E2 --------------------------------------------------------------------------
// compute_edvoldt (geometry.cxx:264-272), update_stress (rheology.cxx:728-1026),
// NMD_stress element part (geometry.cxx:294-296)
// The stress update of element e.  DEFER = 1 (first pass): returns true WITHOUT having stored
// anything but viscosity[e] when the element needs the Mohr-Coulomb return mapping; the second
// pass then runs the same code with DEFER = 0 for exactly those elements.
template <class M, int DEFER>
__device__ __forceinline__ bool e2_element(const int e, const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt,
     const DevClock *__restrict__ clk, int ne, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
     const double *__restrict__ ntmp, const MatData &md,
     const double *__restrict__ volume, const double *__restrict__ volume_old,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ strain_rate,
     double *__restrict__ plstrain, double *__restrict__ delta_plstrain, double *__restrict__ viscosity,
     double *__restrict__ dpressure, double *__restrict__ etmp2)
{
    const double dt = clk->dt;
    const int4 cn = conn[e];
    const int rheol = p->rheol_type;
    const desk::Mix mx = mix_of(md, p->nmat, e);
    const ElemProps pr = load_props(p, md, mx, ne, e);

    double dj = 0;
    dj += ntmp[cn.x]; dj += ntmp[cn.y]; dj += ntmp[cn.z]; dj += ntmp[cn.w];
    const double edvoldt = dj / 4;

    double s[6], es[6], edot[6];
    for (int i = 0; i < 6; ++i) {
        s[i] = stress[(size_t)i*ne + e];
        es[i] = strain[(size_t)i*ne + e];
        edot[i] = strain_rate[(size_t)i*ne + e];
    }
    const double old_s = desk::trace3(s);
    {
        double div = desk::trace3(edot);
        for (int i = 0; i < 3; ++i) edot[i] += (edvoldt - div) / 3;
    }
    for (int i = 0; i < 6; ++i) es[i] += edot[i] * dt;
    double de[6];
    for (int i = 0; i < 6; ++i) de[i] = edot[i] * dt;
    double dpl = 0.;
    bool defer = false;
    const double vol = volume[e];

    M::stage_end();
    double visc = 0;
    if (rheol & DES_RH_VISCOUS) {
        double T = 0;
        T += xt[cn.x].w; T += xt[cn.y].w; T += xt[cn.z].w; T += xt[cn.w].w;
        T /= 4;
        visc = desk::mat_visc<M>(p, vt, mx, T, s, edot);
        viscosity[e] = visc;
    }

    switch (rheol) {
    case DES_RH_ELASTIC:
        desk::elastic(pr.bulkm, pr.shearm, de, s);
        break;
    case DES_RH_VISCOUS:
        desk::viscous(pr.bulkm, visc, desk::trace3(es), edot, s);
        break;
    case DES_RH_MAXWELL: {
        double dv = vol / volume_old[e] - 1;
        desk::maxwell(pr.bulkm, pr.shearm, visc, dt, dv, de, s);
        break;
    }
    case DES_RH_EP: {
        double amc, anphi, anpsi, hardn, ten_max;
        double pls = plstrain[e];
        desk::plastic_props<M>(p, mx, pls, amc, anphi, anpsi, hardn, ten_max);
        double depls = desk::elasto_plastic<M, DEFER>(pr.bulkm, pr.shearm, amc, anphi, anpsi, hardn, ten_max, de, s, &defer);
        if (DEFER && defer) return true;
        if (depls != 0) plstrain[e] = pls + depls;       // plstrain += 0 is the identity
        dpl = depls;
        break;
    }
    case DES_RH_EVP: {
        double dv = vol / volume_old[e] - 1;
        double sv[6];
        for (int i = 0; i < 6; ++i) sv[i] = s[i];
        desk::maxwell(pr.bulkm, pr.shearm, visc, dt, dv, de, sv);
        double svII = desk::second_invariant2(sv);
        double amc, anphi, anpsi, hardn, ten_max;
        double pls = plstrain[e];
        desk::plastic_props<M>(p, mx, pls, amc, anphi, anpsi, hardn, ten_max);
        double sp[6];
        for (int i = 0; i < 6; ++i) sp[i] = s[i];
        double depls = desk::elasto_plastic<M, DEFER>(pr.bulkm, pr.shearm, amc, anphi, anpsi, hardn, ten_max, de, sp, &defer);
        if (DEFER && defer) return true;
        double spII = desk::second_invariant2(sp);
        if (svII < spII) {
            for (int i = 0; i < 6; ++i) s[i] = sv[i];
        } else {
            for (int i = 0; i < 6; ++i) s[i] = sp[i];
            plstrain[e] = pls + depls;
            dpl = depls;
        }
        break;
    }
    default: break;
    }
    delta_plstrain[e] = dpl;
    for (int i = 0; i < 6; ++i) {
        stress[(size_t)i*ne + e] = s[i];
        strain[(size_t)i*ne + e] = es[i];
    }
    for (int i = 0; i < 3; ++i) strain_rate[(size_t)i*ne + e] = edot[i];   // only the diagonal changed
    if (p->is_using_mixed_stress) {
        double dp = desk::trace3(s) - old_s;
        dpressure[e] = dp;
        etmp2[e] = dp * vol;
    }
    return defer;                  // went past the yield pre-filter
}

// First pass: every element [e_begin, e_begin + e_count) (the whole local mesh, or a sub-range:
// ne stays the SoA plane stride).  DEFER = 1: elements that need the return mapping are appended
// to `list` (wave-aggregated: one atomic per wavefront) for E2_return_mapping.
template <class M, int DEFER>
__global__ void __launch_bounds__(DES_BLOCK, DEFER ? DES_E2_WAVES_FAST : DES_E2_WAVES)
E2_update_stress(const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt, const DevClock *__restrict__ clk,
     int ne, int e_begin, int e_count, int nblocks, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
     const double *__restrict__ ntmp, const MatData md,
     const double *__restrict__ volume, const double *__restrict__ volume_old,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ strain_rate,
     double *__restrict__ plstrain, double *__restrict__ delta_plstrain, double *__restrict__ viscosity,
     double *__restrict__ dpressure, double *__restrict__ etmp2, int *__restrict__ list, int *__restrict__ count)
{
    M::stage_begin();
    const int el = desk::logical_block(nblocks) * DES_BLOCK + threadIdx.x;
    if (el >= e_count) return;
    const int e = e_begin + el;
    const bool defer = e2_element<M, DEFER>(e, p, vt, clk, ne, conn, xt, ntmp, md, volume, volume_old, stress, strain, strain_rate,
                                            plstrain, delta_plstrain, viscosity, dpressure, etmp2);
    // one atomic per wavefront that has such elements; without DEFER only the count is kept
    // (des_scalars::n_return_mapping, and what the host picks the next call's mode from)
    const unsigned long long mask = __ballot(defer);
    if (defer) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mask) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(count, __popcll(mask));
        if (DEFER) {
            base = __shfl(base, leader);
            list[base + __popcll(mask & ((1ull << lane) - 1))] = e;
        }
    }
}

// Second pass: the elements the first pass set aside, full stress update with the return mapping
// (same code, same arithmetic; the order of the list does not matter, every element is its own).
template <class M>
__global__ void __launch_bounds__(DES_BLOCK, DES_E2_WAVES)
E2_return_mapping(const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt, const DevClock *__restrict__ clk,
     int ne, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
     const double *__restrict__ ntmp, const MatData md,
     const double *__restrict__ volume, const double *__restrict__ volume_old,
     double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ strain_rate,
     double *__restrict__ plstrain, double *__restrict__ delta_plstrain, double *__restrict__ viscosity,
     double *__restrict__ dpressure, double *__restrict__ etmp2, const int *__restrict__ list, const int *__restrict__ count)
{
    M::stage_begin();
    M::stage_end();
    const int n = *count;
    for (int i = blockIdx.x * DES_BLOCK + threadIdx.x; i < n; i += gridDim.x * DES_BLOCK)
        e2_element<M, 0>(list[i], p, vt, clk, ne, conn, xt, ntmp, md, volume, volume_old, stress, strain, strain_rate,
                         plstrain, delta_plstrain, viscosity, dpressure, etmp2);
}


// ---- N2 --------------------------------------------------------------------------
// NMD_stress gather (geometry.cxx:302-309)
__global__ void __launch_bounds__(DES_BLOCK)
N2_nmd_gather(int o0, int nn, int nblocks, int npb, const int *__restrict__ sup_idx, const int *__restrict__ sup_pack,
     const double *__restrict__ etmp2, const double *__restrict__ volume_n, double *__restrict__ ntmp)
{
    __shared__ double lds[DES_TILE_LDS(DES_TILE_N2)];
    const int TILE = DES_TILE_N2;
    const int n0 = o0 + desk::logical_block(nblocks) * npb;
    const int n = (threadIdx.x < npb) ? n0 + threadIdx.x : nn;
    if (n0 >= nn) return;
    const int nlast = min(n0 + npb, nn);
    const int kb = sup_idx[n0], ke = sup_idx[nlast];
    int r0 = ke, r1 = ke;
    if (n < nn) { r0 = sup_idx[n]; r1 = sup_idx[n+1]; }
    double acc = 0;
#if DES_PIPE
    constexpr int PER = DES_TILE_N2 / DES_BLOCK;
    double rv[PER];
    auto fetch = [&](int t0, int tn) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) rv[u] = etmp2[sup_pack[t0 + j] >> 2];
        }
    };
    if (kb < ke) fetch(kb, min(TILE, ke - kb));
#endif
    for (int t0 = kb; t0 < ke; t0 += TILE) {
        const int tn = min(TILE, ke - t0);
#if DES_PIPE
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) lds[lds_slot(j)] = rv[u];
        }
        __syncthreads();
        if (t0 + TILE < ke) fetch(t0 + TILE, min(TILE, ke - t0 - TILE));
#else
        for (int j = threadIdx.x; j < tn; j += DES_BLOCK)
            lds[lds_slot(j)] = etmp2[sup_pack[t0 + j] >> 2];
        __syncthreads();
#endif
        const int a = max(r0, t0) - t0, b = min(r1, t0 + tn) - t0;
        for (int j = a; j < b; ++j) acc += lds[lds_slot(j)];
        __syncthreads();
    }
    if (n < nn) ntmp[n] = acc / volume_n[n];
}

// ---- E3 --------------------------------------------------------------------------
// NMD_stress apply (geometry.cxx:316-331), update_force element part (fields.cxx:623-653).
// The 12 force terms of an element are one 96-byte record; a wavefront's records are staged
// in LDS and written back as contiguous 16-byte pieces instead of 64 strided 8-byte stores.
// Workgroups past the element range compute the stress-bc facet terms (bc_facet_work):
// both only read the nodal records, and N3 consumes both.
__device__ void bc_facet_work(const des_params *__restrict__ p, int g, const int4 *__restrict__ conn,
                              const d4 *__restrict__ xt, const MatData &md,
                              const int *__restrict__ f_elem, const int *__restrict__ f_facet,
                              const int *__restrict__ f_kind, const double *__restrict__ f_val,
                              double *__restrict__ f_tmp);

__global__ void __launch_bounds__(DES_BLOCK, DES_E3_WAVES)
E3_nmd_force(const des_params *__restrict__ p, int nmd, int ne, int e_begin, int e_count, int nblocks, int nblocks8,
     const int4 *__restrict__ conn,
     const d4 *__restrict__ xt, const double *__restrict__ ntmp, const MatData md,
     const double *__restrict__ volume,
     const double *__restrict__ dpressure, double *__restrict__ stress, double *__restrict__ ftmp,
     int nbcf, const int *__restrict__ f_elem, const int *__restrict__ f_facet,
     const int *__restrict__ f_kind, const double *__restrict__ f_val, double *__restrict__ f_tmp)
{
    if ((int)blockIdx.x >= nblocks8) {                    // facet blocks (uniform per workgroup)
        const int g = ((int)blockIdx.x - nblocks8) * DES_BLOCK + threadIdx.x;
        if (g < nbcf) bc_facet_work(p, g, conn, xt, md, f_elem, f_facet, f_kind, f_val, f_tmp);
        return;
    }
    __shared__ double stage[DES_BLOCK * 13];              // 12 doubles per element, row stride 13
    const int e_end = e_begin + e_count;                  // this launch's element range
    const int e0 = e_begin + desk::logical_block(nblocks) * DES_BLOCK;
    const int e = e0 + threadIdx.x;
    if (e0 >= e_end) return;
    if (e < e_end) {
        const int4 cn = conn[e];
        d4 c[4];
        c[0] = xt[cn.x]; c[1] = xt[cn.y]; c[2] = xt[cn.z]; c[3] = xt[cn.w];
        double s[6];
        for (int i = 0; i < 6; ++i) s[i] = stress[(size_t)i*ne + e];
        if (nmd) {                                          // is_using_mixed_stress, outside the isostasy loop
            double dp = 0;
            dp += ntmp[cn.x]; dp += ntmp[cn.y]; dp += ntmp[cn.z]; dp += ntmp[cn.w];
            double dp_el = dp / 4;
            double dp_orig = dpressure[e];
            double ddp = (-dp_orig + dp_el) / 3;
            for (int i = 0; i < 3; ++i) { s[i] += ddp; stress[(size_t)i*ne + e] = s[i]; }
        }
        const double vol = volume[e];
        double sx[4], sy[4], sz[4];
        desk::shape_fn(c, vol, sx, sy, sz);
        double buoy = 0;
        if (p->gravity != 0) {
            double T = 0;
            T += c[0].w; T += c[1].w; T += c[2].w; T += c[3].w;
            T /= 4;
            const desk::Mix mx = mix_of(md, p->nmat, e);
            const double rho = desk::mat_rho(p, mx, T);
            const double phi = load_props(p, md, mx, ne, e).phi;
            buoy = (rho * (1 - phi) + 1000.0 * phi) * p->gravity / 4;
        }
        double *out = stage + threadIdx.x * 13;
        for (int i = 0; i < 4; ++i) {
            out[i*3 + 0] = (s[0]*sx[i] + s[3]*sy[i] + s[4]*sz[i]) * vol;
            out[i*3 + 1] = (s[3]*sx[i] + s[1]*sy[i] + s[5]*sz[i]) * vol;
            out[i*3 + 2] = (s[4]*sx[i] + s[5]*sy[i] + s[2]*sz[i] + buoy) * vol;
        }
    }
    __syncthreads();
    const int nvalid = min(DES_BLOCK, e_end - e0) * 12;
    double *dst = ftmp + (size_t)e0 * 12;
    for (int idx = threadIdx.x; idx < nvalid; idx += DES_BLOCK) {
        const int t = idx / 12, k = idx - t * 12;
        dst[idx] = stage[t * 13 + k];
    }
}

// ---- stress-bc facets ------------------------------------------------------------
// apply_stress_bcs facet loop (bc.cxx:707-777) and apply_stress_bcs_neumann (bc.cxx:846-905)
__device__ void bc_facet_work(const des_params *__restrict__ p, int g, const int4 *__restrict__ conn,
                              const d4 *__restrict__ xt, const MatData &md,
                              const int *__restrict__ f_elem, const int *__restrict__ f_facet,
                              const int *__restrict__ f_kind, const double *__restrict__ f_val,
                              double *__restrict__ f_tmp)
{
    const int e = f_elem[g], f = f_facet[g], kind = f_kind[g];
    const int4 cn = conn[e];
    const int cna[4] = {cn.x, cn.y, cn.z, cn.w};
    d4 fc[3];
    for (int j = 0; j < 3; ++j) fc[j] = xt[cna[NODE_OF_FACET_D[f][j]]];
    // normal_vector_of_facet, bc.cxx:24-54
    double v01[3] = {fc[1].x - fc[0].x, fc[1].y - fc[0].y, fc[1].z - fc[0].z};
    double v02[3] = {fc[2].x - fc[0].x, fc[2].y - fc[0].y, fc[2].z - fc[0].z};
    double normal[3];
    normal[0] = (v01[1] * v02[2] - v01[2] * v02[1]) / 2;
    normal[1] = (v01[2] * v02[0] - v01[0] * v02[2]) / 2;
    normal[2] = (v01[0] * v02[1] - v01[1] * v02[0]) / 2;
    double zcenter = (fc[0].z + fc[1].z + fc[2].z) / 3;
    double *out = f_tmp + (size_t)g * 9;
    if (kind >= 3) {
        double traction[3] = {0, 0, 0};
        traction[kind - 3] = f_val[g];
        for (int j = 0; j < 3; ++j)
            for (int d = 0; d < 3; ++d) out[j*3 + d] = traction[d] * normal[d] / 3;
        return;
    }
    double pr;
    if (kind == 0) {
        double T = 0;
        T += xt[cn.x].w; T += xt[cn.y].w; T += xt[cn.z].w; T += xt[cn.w].w;
        T /= 4;
        double rho_effective = desk::mat_rho(p, mix_of(md, p->nmat, e), T);
        pr = p->compensation_pressure -
             (rho_effective + p->winkler_delta_rho) * p->gravity * (zcenter + p->zlength);
    } else if (kind == 1) {
        pr = 0;
        if (zcenter < p->surf_base_level)
            pr = p->sea_water_density * p->gravity * (p->surf_base_level - zcenter);
    } else {
        pr = desk::ref_pressure(p, zcenter);
        if (pr < 0.0) pr = 0.0;
    }
    for (int j = 0; j < 3; ++j)
        for (int d = 0; d < 3; ++d) out[j*3 + d] = pr * normal[d] / 3;
}

// ---- N3 --------------------------------------------------------------------------
// apply_vbcs for one node (bc.cxx:400-651, THREED)
__device__ __forceinline__ void apply_vbcs_node(const des_params *p, unsigned flag, double time,
                                                const double *bnormals, const double *edge_vec,
                                                const int *edge_slot, double v[3])
{
    for (int lf = 0; lf < 4; ++lf) {
        if (!(flag & (1u << lf))) continue;
        const int ni = (lf < 2) ? 0 : 1, li = (lf < 2) ? 1 : 0;
        const double val = p->vbc_values[lf], val_l = p->vbc_val_l[lf];
        switch (p->vbc_types[lf]) {
        case 0: break;
        case 1: v[ni] = val; break;
        case 2: v[li] = 0; v[2] = 0; break;
        case 3: v[ni] = val; v[li] = 0; v[2] = 0; break;
        case 4: v[li] = val; v[2] = 0; break;
        case 5: v[ni] = 0; v[li] = val; v[2] = 0; break;
        case 6: v[ni] = val; v[li] = val_l; break;
        case 7: v[ni] = val; v[li] = 0; break;
        }
    }
    if (flag & 0x3c0u) {
        for (int ib = 6; ib <= 9; ib++) {
            if (!(flag & (1u << ib))) continue;
            const double n[3] = {bnormals[ib], bnormals[DES_NBDRY + ib], bnormals[2*DES_NBDRY + ib]};
            const int type = p->vbc_types[ib];
            double fac = 0;
            if (type == 1 || type == 11) {
                const int nd = (type == 1) ? 3 : 2;
                double target = p->vbc_values[ib];
                if (type == 11) {
                    fac = 1 / sqrt(1 - n[2]*n[2]);
                    target = p->vbc_values[ib] * fac;
                }
                if (flag == (1u << ib)) {
                    double vn = 0;
                    for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                    for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                } else {
                    for (int ic = 0; ic < ib; ic++) {
                        if (!(flag & (1u << ic))) continue;
                        if (p->vbc_types[ic] == 0) {
                            double vn = 0;
                            for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                            for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                        } else if (p->vbc_types[ic] == 1) {
                            const int slot = edge_slot[ic*DES_NBDRY + ib];
                            if (slot < 0) continue;
                            const double *edge = &edge_vec[slot*3];
                            double ve = 0;
                            for (int d = 0; d < 3; d++) ve += v[d] * edge[d];
                            for (int d = 0; d < 3; d++) v[d] = ve * edge[d];
                        }
                    }
                }
            } else if (type == 3) {
                for (int d = 0; d < 3; d++) v[d] = p->vbc_values[ib] * n[d];
            } else if (type == 13) {
                fac = 1 / sqrt(1 - n[2]*n[2]);
                for (int d = 0; d < 2; d++) v[d] = p->vbc_values[ib] * fac * n[d];
                v[2] = 0;
            }
        }
    }
    int bc_z0 = p->vbc_types[4], bc_z1 = p->vbc_types[5];
    if (time > p->vbc_val_z1_loading_period) bc_z1 = 0;
    if (bc_z0 == 0 && bc_z1 == 0) return;
    const double bc_vz0 = p->vbc_values[4], bc_vz1 = p->vbc_values[5];
    if (flag & (1u << 4)) {
        switch (bc_z0) {
        case 1: v[2] = bc_vz0; break;
        case 2: v[0] = 0; v[1] = 0; break;
        case 3: v[0] = 0; v[1] = 0; v[2] = bc_vz0; break;
        }
    }
    if (flag & (1u << 5)) {
        switch (bc_z1) {
        case 1: v[2] = bc_vz1; break;
        case 2: v[0] = 0; v[1] = 0; break;
        case 3: v[0] = 0.0; v[1] = 0; v[2] = bc_vz1; break;
        case 4: v[0] = bc_vz1; v[1] = 0; v[2] = 0; break;
        }
    }
}

__global__ void __launch_bounds__(DES_BLOCK)
k_apply_vbcs(const des_params *__restrict__ p, const DevClock *__restrict__ clk, int nn,
             const unsigned *__restrict__ bcflag, const double *__restrict__ bnormals,
             const double *__restrict__ edge_vec, const int *__restrict__ edge_slot, d4 *__restrict__ vm)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n >= nn) return;
    const unsigned flag = bcflag[n];
    if (!(flag & 0x3ffu)) return;
    d4 m4 = vm[n];
    double v[3] = {m4.x, m4.y, m4.z};
    apply_vbcs_node(p, flag, clk->time, bnormals, edge_vec, edge_slot, v);
    m4.x = v[0]; m4.y = v[1]; m4.z = v[2];
    vm[n] = m4;
}

// update_force node loop (fields.cxx:659-676), apply_stress_bcs node loop (bc.cxx:783-802,
// 817-823), apply_stress_bcs_neumann, apply_damping (fields.cxx:483-579), update_velocity
// (fields.cxx:725-742), residual partial sums (fields.cxx:700-722), apply_vbcs,
// update_coordinate (fields.cxx:761-784)
__global__ void __launch_bounds__(DES_BLOCK)
N3_force_velocity_coord(const des_params *__restrict__ p, const DevClock *__restrict__ clk, int o0, int nn_own_end,
     int nn, int nn_global, int nblocks, int npb,
     const int *__restrict__ sup_idx, const int *__restrict__ sup_pack, const unsigned *__restrict__ bcflag,
     const double *__restrict__ ftmp, unsigned bc_mask, const int *__restrict__ bcn_idx,
     const int *__restrict__ bcn_ent, const double *__restrict__ bcf_tmp,
     const double *__restrict__ coord0, const double *__restrict__ ymass,
     const double *__restrict__ bnormals, const double *__restrict__ edge_vec, const int *__restrict__ edge_slot,
     d4 *__restrict__ xt, d4 *__restrict__ vm, double *__restrict__ force, double *__restrict__ fres,
     double *__restrict__ res_part)
{
    __shared__ double lds[3][DES_TILE_LDS(DES_TILE_N3)];
    __shared__ double red[DES_BLOCK / 64];
    // every local node is updated (nn = local node count = stride of the SoA planes); the owned
    // nodes [o0, nn_own_end) alone enter the residual
    const int lb = desk::logical_block(nblocks);
    const int n0 = lb * npb;
    const int n = (threadIdx.x < npb) ? n0 + threadIdx.x : nn;
    if (n0 >= nn) return;
    const int nlast = min(n0 + npb, nn);
    const int kb = sup_idx[n0], ke = sup_idx[nlast];
    int r0 = ke, r1 = ke;
    if (n < nn) { r0 = sup_idx[n]; r1 = sup_idx[n+1]; }
    double f[3] = {0, 0, 0}, fr[3] = {0, 0, 0};
#if DES_PIPE
    constexpr int PER = DES_TILE_N3 / DES_BLOCK;
    static_assert(DES_TILE_N3 % DES_BLOCK == 0, "tile must be a multiple of the block");
    double q0[PER], q1[PER], q2[PER];
    auto fetch = [&](int t0, int tn) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) {
                const int pk = sup_pack[t0 + j];
                const double *tr = ftmp + (size_t)(pk >> 2) * 12 + (pk & 3) * 3;
                q0[u] = tr[0]; q1[u] = tr[1]; q2[u] = tr[2];
            }
        }
    };
    if (kb < ke) fetch(kb, min(DES_TILE_N3, ke - kb));
#endif
    for (int t0 = kb; t0 < ke; t0 += DES_TILE_N3) {
        const int tn = min(DES_TILE_N3, ke - t0);
#if DES_PIPE
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = threadIdx.x + u * DES_BLOCK;
            if (j < tn) {
                const int sl = lds_slot(j);
                lds[0][sl] = q0[u]; lds[1][sl] = q1[u]; lds[2][sl] = q2[u];
            }
        }
        __syncthreads();
        if (t0 + DES_TILE_N3 < ke) fetch(t0 + DES_TILE_N3, min(DES_TILE_N3, ke - t0 - DES_TILE_N3));
#else
        for (int j = threadIdx.x; j < tn; j += DES_BLOCK) {
            const int pk = sup_pack[t0 + j];
            const double *tr = ftmp + (size_t)(pk >> 2) * 12 + (pk & 3) * 3;
            const int sl = lds_slot(j);
            lds[0][sl] = tr[0]; lds[1][sl] = tr[1]; lds[2][sl] = tr[2];
        }
        __syncthreads();
#endif
        const int a = max(r0, t0) - t0, b = min(r1, t0 + tn) - t0;
        for (int j = a; j < b; ++j) {
            const int sl = lds_slot(j);
            const double t0v = lds[0][sl], t1v = lds[1][sl], t2v = lds[2][sl];
            f[0] -= t0v; f[1] -= t1v; f[2] -= t2v;
            fr[0] = t0v; fr[1] = t1v; fr[2] = t2v;          // assignment: fields.cxx:673
        }
        __syncthreads();
    }
    double l2 = 0.0;
    if (n < nn) {
        const double dt = clk->dt;
        const unsigned flag = bcflag[n];
        d4 x4 = xt[n];
        if (flag & bc_mask) {
            const int b0 = bcn_idx[n], b1 = bcn_idx[n+1];
            int b = b0;
            for (; b < b1; ++b) {
                const int ent = bcn_ent[b];
                if (ent & 1) break;
                const double *t = bcf_tmp + (size_t)(ent >> 1) * 3;
                f[0] -= t[0]; f[1] -= t[1]; f[2] -= t[2];
            }
            if (p->has_elastic_foundation && (flag & (1u << 4)))
                f[2] -= p->elastic_foundation_constant * (x4.z - coord0[(size_t)2*nn + n]);
            for (; b < b1; ++b) {
                const double *t = bcf_tmp + (size_t)(bcn_ent[b] >> 1) * 3;
                f[0] += t[0]; f[1] += t[1]; f[2] += t[2];
            }
        }
        d4 m4 = vm[n];
        double v[3] = {m4.x, m4.y, m4.z};
        const double small_vel = 1e-13;
        const double dfac = p->damping_factor;
        switch (p->damping_option) {
        case 1:
            for (int j = 0; j < 3; j++)
                if (fabs(v[j]) > small_vel) f[j] -= dfac * copysign(f[j], v[j]);
            break;
        case 2:
            for (int j = 0; j < 3; j++) f[j] -= dfac * f[j];
            break;
        case 3:
            for (int j = 0; j < 3; j++) {
                if ((f[j] < 0) == (v[j] < 0)) f[j] -= dfac * f[j];   // fields.cxx:538 (comma operator)
                else                          f[j] += (1 - dfac) * f[j];
            }
            break;
        case 4: {
            double critical_coeff = 2.0 * sqrt(m4.w * ymass[n]);
            for (int j = 0; j < 3; j++)
                if (fabs(v[j]) > small_vel) {
                    double f_C = dfac * copysign(f[j], v[j]);
                    double f_V = critical_coeff * v[j];
                    double f_damping = (fabs(f_C) < fabs(f_V)) ? f_V : f_C;
                    f[j] -= f_damping;
                }
            break;
        }
        default: break;
        }
        for (int j = 0; j < 3; j++) {
            force[(size_t)j*nn + n] = f[j];
            fres[(size_t)j*nn + n] = fr[j];
            v[j] += dt * f[j] / m4.w;
        }
        if (n >= o0 && n < nn_own_end) {
            const double num = (double)nn_global * 3;
            l2 = fr[0]*fr[0] / num;
            l2 += fr[1]*fr[1] / num;
            l2 += fr[2]*fr[2] / num;
        }
        if (clk->iso) {
            // isostasy_adjustment (dynearthsol.cxx:521-535): no velocity bcs, vertical motion only,
            // a bottom without Winkler foundation is held
            v[0] = 0; v[1] = 0;
            if (!p->has_winkler_foundation && (flag & (1u << 4))) v[2] = 0;       // BOUNDZ0
        } else if (flag & 0x3ffu)
            apply_vbcs_node(p, flag, clk->time, bnormals, edge_vec, edge_slot, v);
        m4.x = v[0]; m4.y = v[1]; m4.z = v[2];
        vm[n] = m4;
        if (p->has_moving_mesh || clk->iso) {
            x4.x += v[0] * dt; x4.y += v[1] * dt; x4.z += v[2] * dt;
            xt[n] = x4;
        }
    }
    // per-block partial of the residual; the partials are added in block order afterwards
    l2 = desk::wave_sum(l2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0];
        for (int i = 1; i < DES_BLOCK / 64; ++i) t += red[i];
        res_part[lb] = t;
    }
}

// ---- surface processes -----------------------------------------------------------
// simple_diffusion (bc.cxx:954-1107) + coordinate/dhacc update (bc.cxx:1770-1777), one thread
// per surface node.  The facet quantities of bc.cxx:954-1039 (projected area, slope term of the
// facet's local node) are recomputed by every node that touches the facet -- ~6x redundant
// work on O(surface) data, in exchange for one launch and no facet temporaries; the values are
// the same deterministic expressions, so the sums are bit-identical.
__global__ void __launch_bounds__(DES_BLOCK)
k_s2(const des_params *__restrict__ p, DevClock *__restrict__ clk, int ntop, int diffuse,
     const int *__restrict__ top_nodes, const int *__restrict__ ssup_idx, const int *__restrict__ ssup_arr,
     const int *__restrict__ conn_surf, int etop, const d4 *__restrict__ xt_in, int o0, int o1,
     double *__restrict__ dh, double *__restrict__ dhacc, double *__restrict__ znew, double *__restrict__ dh_n)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    double d = 0.;
    const int n = (i < ntop) ? top_nodes[i] : -1;
    if (n >= 0) {                               // every local surface node; [o0, o1) = the owned ones
        if (diffuse) {
            double total_dx = 0., total_slope = 0.;
            // facets in batches of four: all facet ids, then all node ids, then all node records are
            // requested before the first is used, so a batch costs three memory latencies instead
            // of three per facet; the sums below still run in list order
            const int jb = ssup_idx[i], je = ssup_idx[i+1];
            for (int j0 = jb; j0 < je; j0 += 4) {
                int kf[4], nd[4][3];
                d4 cf[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) kf[u] = (j0 + u < je) ? ssup_arr[j0 + u] : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    for (int m = 0; m < 3; ++m) nd[u][m] = (kf[u] >= 0) ? conn_surf[(size_t)m*etop + kf[u]] : n;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    for (int m = 0; m < 3; ++m) cf[u][m] = xt_in[nd[u][m]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (kf[u] < 0) continue;
                    const d4 *c = cf[u];
                    double x01 = c[1].x - c[0].x, y01 = c[1].y - c[0].y;
                    double x02 = c[2].x - c[0].x, y02 = c[2].y - c[0].y;
                    double projected_area = 0.5 * (x01*y02 - y01*x02);
                    total_dx += projected_area;
                    double shp2dx[3], shp2dy[3];
                    double iv = 1 / (2 * projected_area);
                    shp2dx[0] = iv * (c[1].y - c[2].y);
                    shp2dx[1] = iv * (c[2].y - c[0].y);
                    shp2dx[2] = iv * (c[0].y - c[1].y);
                    shp2dy[0] = iv * (c[2].x - c[1].x);
                    shp2dy[1] = iv * (c[0].x - c[2].x);
                    shp2dy[2] = iv * (c[1].x - c[0].x);
                    const double zz[3] = {c[0].z, c[1].z, c[2].z};
                    for (int m = 0; m < 3; ++m) {
                        if (nd[u][m] == n) {
                            double slope = 0;
                            for (int q = 0; q < 3; q++)
                                slope += (shp2dx[m] * shp2dx[q] + shp2dy[m] * shp2dy[q]) * zz[q];
                            total_slope += slope * projected_area;
                            break;
                        }
                    }
                }
            }
            double conv = p->surface_diffusivity * clk->dt * total_slope / total_dx;
            d -= conv;
        }
        dh[i] = d;
        // neighbours still need this node's OLD height: the new one goes to a side buffer and
        // is committed by the next launch (k_s3_finalize)
        znew[i] = xt_in[n].z + d;
        dhacc[n] += d;
        dh_n[n] = d;
    }
    // max |dh| (bc.cxx:1811-1821); max is order-independent
    __shared__ double red[DES_BLOCK / 64];
    double m = desk::wave_max((n >= o0 && n < o1) ? fabs(d) : 0.0);      // owned nodes only: ghosts may be stale
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < DES_BLOCK / 64; ++k) m = fmax(m, red[k]);
        if (m > 0) desk::atomic_max_double(&clk->maxdh, m);
    }
}

// edvacc_surf update (bc.cxx:1784-1794), commit of the surface heights k_s2 computed
// (bc.cxx:1775), and -- in the last workgroup -- the end-of-step scalars: l2_residual
// (fields.cxx:721) and max_surf_vel (bc.cxx:1825).  The three kinds of workgroup do not depend on
// each other (the facet-area term only reads x and y); a decomposed run launches the commit
// before the surface halo exchange and the rest after it.
__global__ void __launch_bounds__(DES_BLOCK)
k_s3_finalize(DevClock *__restrict__ clk, int etop, int nsurf_blocks, const int *__restrict__ conn_surf,
              d4 *__restrict__ xt, const double *__restrict__ dh_n, double *__restrict__ edvacc,
              const double *__restrict__ res_part, int nres, int ntop, int nz_blocks,
              const int *__restrict__ top_nodes, const double *__restrict__ znew, int o0, int o1, int do_finalize)
{
    if ((int)blockIdx.x >= nsurf_blocks && (int)blockIdx.x < nsurf_blocks + nz_blocks) {
        const int i = ((int)blockIdx.x - nsurf_blocks) * DES_BLOCK + threadIdx.x;
        if (i < ntop) {
            xt[top_nodes[i]].z = znew[i];
        }
        return;
    }
    if ((int)blockIdx.x < nsurf_blocks) {
        const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
        if (i >= etop) return;
        const int na = conn_surf[i], nb = conn_surf[(size_t)etop + i], nc = conn_surf[(size_t)2*etop + i];
        double dh_e = 0.;
        dh_e += dh_n[na]; dh_e += dh_n[nb]; dh_e += dh_n[nc];
        const d4 a = xt[na], b = xt[nb], c = xt[nc];
        double ab0 = b.x - a.x, ab1 = b.y - a.y, ac0 = c.x - a.x, ac1 = c.y - a.y;
        double base = fabs(ab0*ac1 - ab1*ac0) / 2;           // triangle_area2d, geometry.cxx:59-73
        edvacc[i] += dh_e * base / 3;
        return;
    }
    if (!do_finalize) return;
    __shared__ double red[DES_BLOCK];
    double t = 0;
    for (int i = threadIdx.x; i < nres; i += DES_BLOCK) t += res_part[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int off = DES_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        clk->l2_sum = red[0];
        clk->l2_residual = sqrt(red[0]);
        if (do_finalize == 1) clk->max_surf_vel = clk->maxdh / clk->dt;     // part of surface_processes: moving mesh only
    }
}

// ---- ghost-region exchange (des_halo, des_params.h) ------------------------------------
// State of the listed nodes {x,y,z,vx,vy,vz,T,dh} and elements {stress, strain, plstrain} to /
// from a message buffer; off[i] = position (in doubles) of item i's record in the buffer, so one
// launch fills the messages of all neighbours (a message = node records, then element records).
__global__ void __launch_bounds__(DES_BLOCK)
k_state_pack(int nnodes, const int *__restrict__ nidx, const int *__restrict__ noff,
             int nelems, const int *__restrict__ eidx, const int *__restrict__ eoff,
             const d4 *__restrict__ xt, const d4 *__restrict__ vm, const double *__restrict__ dh_n,
             const double *__restrict__ stress, const double *__restrict__ strain,
             const double *__restrict__ plstrain, int ne, double *__restrict__ buf)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < nnodes) {
        const int k = nidx[i];
        const d4 x = xt[k], v = vm[k];
        double *b = buf + noff[i];
        b[0] = x.x; b[1] = x.y; b[2] = x.z; b[3] = v.x; b[4] = v.y; b[5] = v.z; b[6] = x.w; b[7] = dh_n[k];
    } else if (i < nnodes + nelems) {
        const int j = i - nnodes, e = eidx[j];
        double *b = buf + eoff[j];
        for (int c = 0; c < 6; ++c) { b[c] = stress[(size_t)c*ne + e]; b[6 + c] = strain[(size_t)c*ne + e]; }
        b[12] = plstrain[e];
    }
}

__global__ void __launch_bounds__(DES_BLOCK)
k_state_unpack(int nnodes, const int *__restrict__ nidx, const int *__restrict__ noff,
               int nelems, const int *__restrict__ eidx, const int *__restrict__ eoff,
               d4 *__restrict__ xt, d4 *__restrict__ vm, double *__restrict__ dh_n,
               double *__restrict__ stress, double *__restrict__ strain, double *__restrict__ plstrain,
               int ne, const double *__restrict__ buf)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < nnodes) {
        const int k = nidx[i];
        const double *b = buf + noff[i];
        d4 x, v = vm[k];                                   // the nodal mass stays this rank's own
        x.x = b[0]; x.y = b[1]; x.z = b[2]; x.w = b[6];
        v.x = b[3]; v.y = b[4]; v.z = b[5];
        xt[k] = x; vm[k] = v; dh_n[k] = b[7];
    } else if (i < nnodes + nelems) {
        const int j = i - nnodes, e = eidx[j];
        const double *b = buf + eoff[j];
        for (int c = 0; c < 6; ++c) { stress[(size_t)c*ne + e] = b[c]; strain[(size_t)c*ne + e] = b[6 + c]; }
        plstrain[e] = b[12];
    }
}

// compute_dt partials of this rank, all arranged for a MIN reduction across ranks
__global__ void k_dt_pack(const DevClock *clk, double *red)
{
    red[0] = clk->r_minl; red[1] = clk->r_dt_maxwell; red[2] = clk->r_dt_diffusion;
    red[3] = clk->r_global_dt_min; red[4] = -clk->r_max_vem; red[5] = -clk->max_surf_vel;
}

__global__ void k_dhacc_reset(int ntop, const int *__restrict__ top_nodes, double *__restrict__ dhacc)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < ntop) dhacc[top_nodes[i]] = 0.;
}

// check_nan (utils.hpp:323-394)
// des_dev_libm_eval: one portable-libm function over an array (diagnostic entry)
__global__ void k_libm_eval(int fn, long long n, const double *__restrict__ x, const double *__restrict__ y,
                            double *__restrict__ out)
{
    deslibm::lds_stage();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y ? y[i] : 0.0;
    double r;
    switch (fn) {
    case DES_LIBM_POW:   r = deslibm::pow(a, b); break;
    case DES_LIBM_EXP:   r = deslibm::exp(a); break;
    case DES_LIBM_SIN:   r = deslibm::sin(a); break;
    case DES_LIBM_COS:   r = deslibm::cos(a); break;
    case DES_LIBM_TAN:   r = deslibm::tan(a); break;
    default:             r = deslibm::atan2(a, b); break;
    }
    out[i] = r;
}

__global__ void k_count_nan(const double *a, long long n, unsigned long long *count)
{
    long long i = (long long)blockIdx.x * DES_BLOCK + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (long long)gridDim.x * DES_BLOCK) c += isnan(a[i]) ? 1 : 0;
    if (c) atomicAdd(count, c);
}

// bad_mesh_quality reductions (remeshing.cxx:2752-2866).  slots: [0] min quality (double bits),
// then ints: [2] first tiny element, [3] first distorted bottom node, [4] first worst element
__device__ __forceinline__ double elem_quality3(const int4 cn, const d4 *__restrict__ xt, double vol)
{
    const d4 a = xt[cn.x], b = xt[cn.y], c = xt[cn.z], d = xt[cn.w];
    const double normalization_factor = 216 * sqrt(3.0);
    const double area_sum = (desk::tri_area(a, b, c) + desk::tri_area(a, b, d) +
                             desk::tri_area(c, d, a) + desk::tri_area(c, d, b));
    return normalization_factor * vol * vol / (area_sum * area_sum * area_sum);
}

__global__ void k_quality_a(int ne, int nn, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
                            const double *__restrict__ volume, const unsigned *__restrict__ bcflag,
                            double smallest_vol, double bottom, double bottom_dist, double *qmin, int *islot,
                            const int *__restrict__ n_id, const int *__restrict__ e_id)
{
    // n_id / e_id: the caller's index of a device index ("first" means first in the caller's order)
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    double q = 1.0;
    if (i < ne) {
        const double vol = volume[i];
        if (vol < smallest_vol) atomicMin(&islot[0], e_id ? e_id[i] : i);
        q = fmin(q, elem_quality3(conn[i], xt, vol));
    }
    if (i < nn && bottom_dist >= 0 && (bcflag[i] & (1u << 4)))            // is_bottom: BOUNDZ0
        if (fabs(xt[i].z - bottom) > bottom_dist) atomicMin(&islot[1], n_id ? n_id[i] : i);
    q = desk::wave_min(q);
    if ((threadIdx.x & 63) == 0 && q < 1.0) desk::atomic_min_double(qmin, q);
}

__global__ void k_quality_b(int ne, const int4 *__restrict__ conn, const d4 *__restrict__ xt,
                            const double *__restrict__ volume, const double *qmin, int *islot,
                            const int *__restrict__ e_id)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    const double q = elem_quality3(conn[e], xt, volume[e]);
    if (q < 1.0 && q == *qmin) atomicMin(&islot[2], e_id ? e_id[e] : e);
}

// =====================================================================================
// host side of the engine
// =====================================================================================
template <typename T>
int dev_alloc(T *&ptr, size_t count)
{
    ptr = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)&ptr, count * sizeof(T));
    if (e != hipSuccess) { g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e); return DES_ERR_RESOURCE; }
    return DES_OK;
}

template <typename T>
int dev_upload(T *dst, const T *src, size_t count, hipStream_t s)
{
    if (count == 0) return DES_OK;
    HIP_OK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));
    return DES_OK;
}

inline int nblk(long long n) { return (int)((n + DES_BLOCK - 1) / DES_BLOCK); }
// grids are rounded up to a multiple of 8 so the XCD-aware block map covers every chunk
inline int nblk8(long long n) { int b = nblk(n); return (b + 7) / 8 * 8; }

struct Launch {
    des_dev *h; int k; ProfRec rec; bool on;
    Launch(des_dev *h_, int k_) : h(h_), k(k_), on(h_->prof) {
        if (on) { hipEventCreate(&rec.a); hipEventCreate(&rec.b); rec.k = k; hipEventRecord(rec.a, h->stream); }
    }
    ~Launch() { if (on) { hipEventRecord(rec.b, h->stream); h->prof_recs.push_back(rec); } }
};

inline MatData mat_data(const des_dev *h) { return MatData{ h->markers, h->mono, h->props, h->ptab }; }

void refresh_props(des_dev *h)
{
    if (!h->markers_dirty) return;
    Launch l(h, K_MISC);
    hipLaunchKernelGGL(k_props, dim3(nblk(h->ne)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->markers, h->props, h->mono, h->ne);
    h->markers_dirty = false;
}

template <int MODE>
void launch_e1(des_dev *h)
{
    Launch l(h, K_E1);
    const int nb = nblk(h->ne);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(E1_geom_rotate_strainrate<MODE>), dim3(nblk8(h->ne)), dim3(DES_BLOCK), 0, h->stream,
                       h->d_p, h->d_clk, h->ne, nb, h->conn, h->xt, h->vm, mat_data(h), h->radiogenic,
                       h->topflag, h->stress, h->strain, h->plstrain, h->volume, h->volume_old, h->strain_rate,
                       h->mrec, h->ttmp);
}

// Output::average_fields (output.cxx:327-370) on the end-of-step fields, i.e. after the C part
// of E1.  A kernel of its own: fused into E1 it cost that kernel a wave of occupancy (186 VGPRs;
// E1 117 us instead of 74 + 30 for this pure stream of 168 B per element).
__global__ void __launch_bounds__(DES_BLOCK)
k_average_fields(const des_params *__restrict__ p, DevClock *__restrict__ clk, int ne,
                 const double *__restrict__ stress, const double *__restrict__ strain,
                 const double *__restrict__ delta_plstrain, double *__restrict__ stress_avg,
                 double *__restrict__ dplstrain_avg, double *__restrict__ strain0)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    if (clk->steps % p->quality_check_step_interval == 1) {
        if (e == 0) clk->avg_time0 = clk->time;
        for (int i = 0; i < 6; ++i) {
            stress_avg[(size_t)i*ne + e] = stress[(size_t)i*ne + e];
            strain0[(size_t)i*ne + e] = strain[(size_t)i*ne + e];
        }
        dplstrain_avg[e] = delta_plstrain[e];
    } else {
        for (int i = 0; i < 6; ++i) stress_avg[(size_t)i*ne + e] += stress[(size_t)i*ne + e];
        dplstrain_avg[e] += delta_plstrain[e];
    }
}

// end-of-step E1 (C part) of step `step_no`, optionally fused with the A part of the next step
void launch_e1_end(des_dev *h, long long step_no, bool with_next)
{
    const bool do_dt = (step_no % 10 == 0);
    const int sel = (with_next ? 1 : 0) | (do_dt ? 2 : 0);
    switch (sel) {
    case 0: launch_e1<MODE_C>(h); break;
    case 1: launch_e1<MODE_C | MODE_A>(h); break;
    case 2: launch_e1<MODE_C | MODE_DT>(h); break;
    case 3: launch_e1<MODE_C | MODE_A | MODE_DT>(h); break;
    }
    if (h->p.is_outputting_averaged_fields) {
        Launch l(h, K_MISC);
        hipLaunchKernelGGL(k_average_fields, dim3(nblk(h->ne)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->ne,
                           h->stress, h->strain, h->delta_plstrain, h->stress_avg, h->dplstrain_avg, h->strain0);
    }
}

// coordinates at the first step of an averaging interval (output.cxx:334-338)
__global__ void k_avg_coord0(int nn, const d4 *__restrict__ xt, double *__restrict__ coord_avg0)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    const d4 c = xt[n];
    coord_avg0[n] = c.x; coord_avg0[(size_t)nn + n] = c.y; coord_avg0[(size_t)2*nn + n] = c.z;
}

void launch_avg_coord0(des_dev *h, long long step_no)
{
    if (h->p.is_outputting_averaged_fields && step_no % h->p.quality_check_step_interval == 1)
        hipLaunchKernelGGL(k_avg_coord0, dim3(nblk(h->nn)), dim3(DES_BLOCK), 0, h->stream, h->nn, h->xt, h->coord_avg0);
}

// node workgroups: ceil(owned nodes / nodes per workgroup), and the grid rounded up to the 8 XCDs
// The node kernels run over EVERY local node: on a decomposed mesh the ghost region is computed
// redundantly (des_halo); only the reductions are restricted to the owned range [o0, o1).
inline int node_blocks(const des_dev *h) { return (h->nn + h->npb - 1) / h->npb; }
inline int node_grid(const des_dev *h) { return (node_blocks(h) + 7) / 8 * 8; }

// nodes per node-kernel workgroup: 256, or 64 while that leaves fewer than two workgroups per CU
// (a 137k-tet mesh has 31k nodes = 120 workgroups of 256 on 256 CUs, each walking 4-5 incidence
// tiles one after the other)
void choose_npb(des_dev *h)
{
    const char *env = std::getenv("DES_NPB");
    const int nown = h->nn;
    h->npb = (nown < 512 * DES_BLOCK) ? 64 : DES_BLOCK;
    if (env && (std::atoi(env) == 64 || std::atoi(env) == 128 || std::atoi(env) == 256)) h->npb = std::atoi(env);
}

// compute_mass gather alone (N1 without the temperature / dvoldt parts)
void launch_mass_gather(des_dev *h)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(N1_mass_temperature_dvoldt<0, 0>), dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream,
                       h->d_p, h->d_clk, 0, h->nn, node_blocks(h), h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->mrec, h->ttmp,
                       mat_data(h), h->ne, h->xt, h->vm, h->volume_n, h->tmass, h->ymass, h->ntmp);
}

void launch_dt_finalize(des_dev *h, const double *red)
{
    Launch l(h, K_DTFIN);
    hipLaunchKernelGGL(k_dt_finalize, dim3(1), dim3(1), 0, h->stream, h->d_p, h->d_clk, red);
}

inline bool surface_diffusion_on(const des_dev *h)
{
    return (h->p.has_moving_mesh || h->iso) && h->p.surface_process_option == 1 && h->ntop > 0;
}

// ---- passes of one step, in launch order -----------------------------------------
void launch_n1(des_dev *h)
{
    Launch l(h, K_N1);
    const int nbn = node_blocks(h);
    if (h->const_mass)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(N1_mass_temperature_dvoldt<1, 1>), dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream,
                           h->d_p, h->d_clk, 0, h->nn, nbn, h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->mrec, h->ttmp, mat_data(h),
                           h->ne, h->xt, h->vm, h->volume_n, h->tmass, h->ymass, h->ntmp);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(N1_mass_temperature_dvoldt<1, 0>), dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream,
                           h->d_p, h->d_clk, 0, h->nn, nbn, h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->mrec, h->ttmp, mat_data(h),
                           h->ne, h->xt, h->vm, h->volume_n, h->tmass, h->ymass, h->ntmp);
}

// update_stress in two passes when the rheology has a yield surface: the first pass (3 waves
// per SIMD) sets the few elements that need the return mapping aside, the second one (the same
// code with the return mapping, 2 waves per SIMD) works that list off.  The list is sparse, so
// the second pass pays ~8x per element; above DES_E2_DEFER_MAX of the mesh one pass is cheaper.
// Both give the same bits.  DES_E2_DEFER=0 / 1 pins the mode; default: choose_e2_mode().
#ifndef DES_E2R_GRID
#define DES_E2R_GRID 512          // workgroups of the second pass (grid-stride loop): two per CU, all resident
#endif
#ifndef DES_E2_DEFER_MAX
#define DES_E2_DEFER_MAX 0.02
#endif
// called whenever the host copy of the clock is fresh (end of des_dev_step / des_dev_phase calls)
void choose_e2_mode(des_dev *h)
{
    if (h->e2_defer == 2) h->e2_two_pass = h->h_clk->n_defer <= DES_E2_DEFER_MAX * h->ne;
}

void launch_e2(des_dev *h, int e_begin = 0, int e_count = -1)
{
    if (e_count < 0) e_count = h->ne;
    if (e_count == 0) return;
    const bool defer = h->e2_two_pass && (h->p.rheol_type == DES_RH_EP || h->p.rheol_type == DES_RH_EVP);
    int *count = &h->d_clk->n_defer;
    {
        Launch l(h, K_E2);
        auto k = h->portable_libm ? (defer ? E2_update_stress<desk::MathPortable, 1> : E2_update_stress<desk::MathPortable, 0>)
                                  : (defer ? E2_update_stress<desk::MathOcml, 1> : E2_update_stress<desk::MathOcml, 0>);
        hipLaunchKernelGGL(k, dim3(nblk8(e_count)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_vt, h->d_clk, h->ne,
                           e_begin, e_count, nblk(e_count), h->conn, h->xt, h->ntmp, mat_data(h), h->volume, h->volume_old, h->stress,
                           h->strain, h->strain_rate, h->plstrain, h->delta_plstrain, h->viscosity, h->dpressure,
                           h->etmp2, h->defer_list, count);
    }
    if (defer) {
        Launch l(h, K_E2R);
        auto k = h->portable_libm ? E2_return_mapping<desk::MathPortable> : E2_return_mapping<desk::MathOcml>;
        hipLaunchKernelGGL(k, dim3(std::min(nblk(e_count), DES_E2R_GRID)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_vt, h->d_clk, h->ne,
                           h->conn, h->xt, h->ntmp, mat_data(h), h->volume, h->volume_old, h->stress,
                           h->strain, h->strain_rate, h->plstrain, h->delta_plstrain, h->viscosity, h->dpressure,
                           h->etmp2, h->defer_list, count);
    }
}

void launch_n2(des_dev *h)
{
    Launch l(h, K_N2);
    // one double per incidence: the lightest gather, best with at most 128 nodes per workgroup
    // even on large meshes (1.1M tets: 22.7 us at 256, 17.6 at 128)
    const int npb2 = std::min(h->npb, 128), nb2 = (h->nn + npb2 - 1) / npb2;
    hipLaunchKernelGGL(N2_nmd_gather, dim3((nb2 + 7) / 8 * 8), dim3(DES_BLOCK), 0, h->stream, 0, h->nn, nb2, npb2, h->sup_idx,
                       h->sup_pack, h->etmp2, h->volume_n, h->ntmp);
}

// `facets`: this launch also carries the stress-bc facet workgroups (once per step)
void launch_e3(des_dev *h, int e_begin = 0, int e_count = -1, bool facets = true)
{
    if (e_count < 0) e_count = h->ne;
    const int nbe8 = nblk8(e_count), nbf = facets ? nblk(h->nbcf) : 0;
    if (nbe8 + nbf == 0) return;
    Launch l(h, K_E3);
    hipLaunchKernelGGL(E3_nmd_force, dim3(nbe8 + nbf), dim3(DES_BLOCK), 0, h->stream, h->d_p,
                       (int)(h->p.is_using_mixed_stress && !h->iso), h->ne, e_begin, e_count,
                       nblk(e_count), nbe8,
                       h->conn, h->xt, h->ntmp, mat_data(h), h->volume, h->dpressure, h->stress, h->ftmp,
                       facets ? h->nbcf : 0, h->bcf_elem, h->bcf_facet, h->bcf_kind, h->bcf_val, h->bcf_tmp);
}

void launch_n3(des_dev *h)
{
    Launch l(h, K_N3);
    const int nown = h->o1 - h->o0;
    hipLaunchKernelGGL(N3_force_velocity_coord, dim3(node_grid(h)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->o0, h->o1,
                       h->nn, h->nn_global, node_blocks(h), h->npb, h->sup_idx, h->sup_pack, h->bcflag, h->ftmp, h->bc_mask, h->bcn_idx,
                       h->bcn_ent, h->bcf_tmp, h->coord0, h->ymass, h->bnormals, h->edge_vec, h->edge_slot, h->xt, h->vm,
                       h->force, h->fres, h->res_part);
}

// surface_processes (bc.cxx:1709-1872) as far as the device state is concerned, first part:
// diffusion of the owned surface nodes
void launch_s2(des_dev *h, long long step_no)
{
    if (!(h->p.has_moving_mesh || h->iso)) return;          // surface_processes is part of update_mesh
    if (h->ntop > 0) {
        Launch l(h, K_S2);
        hipLaunchKernelGGL(k_s2, dim3(nblk(h->ntop)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->ntop,
                           (int)(h->p.surface_process_option == 1), h->top_nodes, h->ssup_idx, h->ssup_arr, h->conn_surf,
                           h->etop, h->xt, h->o0, h->o1, h->dh, h->dhacc, h->znew, h->dh_n);
    }
    if (h->ntop > 0 && step_no != 0 && step_no % h->p.quality_check_step_interval == 0)
        hipLaunchKernelGGL(k_dhacc_reset, dim3(nblk(h->ntop)), dim3(DES_BLOCK), 0, h->stream, h->ntop,
                           h->top_nodes, h->dhacc);
}

// commit of the new surface heights / edvacc_surf / end-of-step scalars (k_s3_finalize)
void launch_s3(des_dev *h, bool commit, bool edvacc, bool finalize)
{
    Launch l(h, K_S3);
    const bool surf = (h->p.has_moving_mesh || h->iso) && h->ntop > 0;
    const int nsb = (edvacc && surface_diffusion_on(h)) ? nblk(h->etop) : 0;
    const int nzb = (commit && surf) ? nblk(h->ntop) : 0;
    const int nown = h->o1 - h->o0;
    hipLaunchKernelGGL(k_s3_finalize, dim3(nsb + nzb + 1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->etop, nsb,
                       h->conn_surf, h->xt, h->dh_n, h->edvacc, h->res_part, node_blocks(h), h->ntop, nzb, h->top_nodes,
                       h->znew, h->o0, h->o1, finalize ? ((h->p.has_moving_mesh || h->iso) ? 1 : 2) : 0);
}

// ---- halo exchange through RCCL on the engine's stream ---------------------------

// The exchange of a step: one grouped send/recv per neighbour carrying the state of the whole
// ghost region, between a pack and an unpack launch, all on the engine's stream.
int exchange(des_dev *h)
{
    if (h->nnbr == 0) return DES_OK;
    if (!h->comm) { g_last_error = "decomposed engine without a communicator: call des_dev_comm_init"; return DES_ERR_INTERNAL; }
    const int ns = h->send_ptr[h->nnbr], nes = h->esend_ptr[h->nnbr];
    const int nr = h->recv_ptr[h->nnbr], ner = h->erecv_ptr[h->nnbr];
    hipLaunchKernelGGL(k_state_pack, dim3(nblk(ns + nes)), dim3(DES_BLOCK), 0, h->stream, ns, h->d_send_idx, h->d_send_noff,
                       nes, h->d_esend_idx, h->d_send_eoff, h->xt, h->vm, h->dh_n, h->stress, h->strain, h->plstrain,
                       h->ne, h->d_sendbuf);
    ncclGroupStart();
    for (int q = 0; q < h->nnbr; ++q) {
        ncclSend(h->d_sendbuf + h->send_off[q], (size_t)(h->send_off[q+1] - h->send_off[q]), ncclDouble,
                 h->nbr_rank[q], h->comm, h->stream);
        ncclRecv(h->d_recvbuf + h->recv_off[q], (size_t)(h->recv_off[q+1] - h->recv_off[q]), ncclDouble,
                 h->nbr_rank[q], h->comm, h->stream);
    }
    ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) { g_last_error = std::string("RCCL: ") + ncclGetErrorString(r); return DES_ERR_RESOURCE; }
    hipLaunchKernelGGL(k_state_unpack, dim3(nblk(nr + ner)), dim3(DES_BLOCK), 0, h->stream, nr, h->d_recv_idx, h->d_recv_noff,
                       ner, h->d_erecv_idx, h->d_recv_eoff, h->xt, h->vm, h->dh_n, h->stress, h->strain, h->plstrain,
                       h->ne, h->d_recvbuf);
    return DES_OK;
}

// compute_dt across ranks: pack the six partials, MIN-allreduce, finalize
int reduce_dt(des_dev *h)
{
    if (h->comm_size <= 1) { launch_dt_finalize(h, nullptr); return DES_OK; }
    hipLaunchKernelGGL(k_dt_pack, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red);
    ncclResult_t r = ncclAllReduce(h->d_red, h->d_red, 6, ncclDouble, ncclMin, h->comm, h->stream);
    if (r != ncclSuccess) { g_last_error = std::string("RCCL: ") + ncclGetErrorString(r); return DES_ERR_RESOURCE; }
    launch_dt_finalize(h, h->d_red);
    return DES_OK;
}

int sync_clock(des_dev *h)
{
    HIP_OK(hipMemcpyAsync(h->h_clk, h->d_clk, sizeof(DevClock), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

// ---- internal data order ------------------------------------------------------------
// The reference numbers nodes and elements along x only (mesh.cxx:2742-2792): 256 consecutive
// nodes of a TetGen mesh are a thin slice scattered over the whole y-z section, so a workgroup's
// gathers hardly share anything (test-3d-big at 460 m: every element record is fetched by 2.8
// node workgroups, every node record by 12 element workgroups; on a Morton order 1.6 and 2.6).
// With the coordinates at hand (des_mesh::coord) the engine therefore keeps its arrays in Morton
// order -- nodes within [0, owned_begin), [owned_begin, owned_end), [owned_end, nnode) so that the
// owned range stays a range; elements by centroid, those touching the low / high halo first /
// last.  Only names change: every list keeps the caller's ORDER (the support lists stay in
// ascending caller element id = the reference's summation order), and upload / download /
// halo lists / reported indices translate at the boundary.
struct PermMesh {
    std::vector<int> n_new2old, n_old2new, e_new2old, e_old2new;
    std::vector<int> conn, sup_idx, sup_arr, sup_lidx, top_nodes, conn_surf, top_elems;
    std::vector<unsigned> bcflag;
    std::vector<int> bf_elem[DES_NBDRY], bnodes[DES_NBDRY];
    des_mesh view;
};

inline unsigned long long morton3(unsigned x, unsigned y, unsigned z)
{
    auto spread = [](unsigned long long v) {                 // 21 bits -> every third bit
        v &= 0x1fffffULL;
        v = (v | v << 32) & 0x1f00000000ffffULL;
        v = (v | v << 16) & 0x1f0000ff0000ffULL;
        v = (v | v << 8) & 0x100f00f00f00f00fULL;
        v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
        v = (v | v << 2) & 0x1249249249249249ULL;
        return v;
    };
    return spread(x) | spread(y) << 1 | spread(z) << 2;
}

void build_perm_mesh(const des_mesh *in, PermMesh &pm)
{
    const int nn = in->nnode, ne = in->nelem;
    const double *X = in->coord;
    double lo[3], hi[3], ext = 0;
    for (int d = 0; d < 3; ++d) {
        lo[d] = hi[d] = X[(size_t)d * nn];
        for (int n = 0; n < nn; ++n) { lo[d] = std::min(lo[d], X[(size_t)d*nn + n]); hi[d] = std::max(hi[d], X[(size_t)d*nn + n]); }
        ext = std::max(ext, hi[d] - lo[d]);
    }
    const double scale = ext > 0 ? 2097151.0 / ext : 0.0;    // cubic cells: one scale for all axes
    auto code = [&](double x, double y, double z) {
        return morton3((unsigned)((x - lo[0]) * scale), (unsigned)((y - lo[1]) * scale), (unsigned)((z - lo[2]) * scale));
    };
    const int ob = in->owned_begin, oe = in->owned_end > 0 ? in->owned_end : nn;
    {
        std::vector<std::pair<unsigned long long, int> > key((size_t)nn);
        for (int n = 0; n < nn; ++n) key[n] = std::make_pair(code(X[n], X[(size_t)nn + n], X[(size_t)2*nn + n]), n);
        // Morton order inside each of the three id ranges (the pairs break ties by caller id)
        std::sort(key.begin(), key.begin() + ob);
        std::sort(key.begin() + ob, key.begin() + oe);
        std::sort(key.begin() + oe, key.end());
        pm.n_new2old.resize((size_t)nn); pm.n_old2new.resize((size_t)nn);
        for (int i = 0; i < nn; ++i) { pm.n_new2old[i] = key[i].second; pm.n_old2new[key[i].second] = i; }
    }
    {
        std::vector<std::pair<unsigned long long, int> > key((size_t)ne);
        for (int e = 0; e < ne; ++e) {
            double c[3] = {0, 0, 0};
            unsigned long long grp = 1;
            bool touches_lo = false, touches_hi = false;
            for (int i = 0; i < 4; ++i) {
                const int n = in->connectivity[(size_t)i*ne + e];
                for (int d = 0; d < 3; ++d) c[d] += X[(size_t)d*nn + n] / 4;
                touches_lo |= n < ob; touches_hi |= n >= oe;
            }
            if (touches_lo) grp = 0; else if (touches_hi) grp = 2;
            key[e] = std::make_pair(grp << 62 | code(c[0], c[1], c[2]) >> 2, e);      // group, then Morton
        }
        std::sort(key.begin(), key.end());
        pm.e_new2old.resize((size_t)ne); pm.e_old2new.resize((size_t)ne);
        for (int i = 0; i < ne; ++i) { pm.e_new2old[i] = key[i].second; pm.e_old2new[key[i].second] = i; }
    }
    const std::vector<int> &nmap = pm.n_old2new, &emap = pm.e_old2new;
    pm.conn.resize((size_t)4*ne);
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < ne; ++e) pm.conn[(size_t)i*ne + emap[e]] = nmap[in->connectivity[(size_t)i*ne + e]];
    pm.sup_idx.assign((size_t)nn + 1, 0);
    for (int i = 0; i < nn; ++i) {
        const int n = pm.n_new2old[i];
        pm.sup_idx[i + 1] = pm.sup_idx[i] + (in->support_idx[n + 1] - in->support_idx[n]);
    }
    pm.sup_arr.resize((size_t)pm.sup_idx[nn]); pm.sup_lidx.resize((size_t)pm.sup_idx[nn]);
    for (int i = 0; i < nn; ++i) {
        const int n = pm.n_new2old[i];
        int k2 = pm.sup_idx[i];
        for (int k = in->support_idx[n]; k < in->support_idx[n + 1]; ++k, ++k2) {   // caller's order kept
            pm.sup_arr[k2] = emap[in->support_arr[k]];
            pm.sup_lidx[k2] = in->support_lidx[k];
        }
    }
    pm.bcflag.resize((size_t)nn);
    for (int i = 0; i < nn; ++i) pm.bcflag[i] = in->bcflag[pm.n_new2old[i]];
    pm.view = *in;
    for (int b = 0; b < DES_NBDRY; ++b) {
        pm.bf_elem[b].resize((size_t)in->nbfacets[b]);
        for (int q = 0; q < in->nbfacets[b]; ++q) pm.bf_elem[b][q] = emap[in->bfacet_elem[b][q]];
        pm.bnodes[b].resize((size_t)in->nbnodes[b]);
        for (int q = 0; q < in->nbnodes[b]; ++q) pm.bnodes[b][q] = nmap[in->bnodes[b][q]];
        pm.view.bfacet_elem[b] = pm.bf_elem[b].data();
        pm.view.bnodes[b] = pm.bnodes[b].data();
    }
    pm.top_nodes.resize((size_t)in->ntop);
    for (int i = 0; i < in->ntop; ++i) pm.top_nodes[i] = nmap[in->top_nodes[i]];
    pm.conn_surf.assign(in->connectivity_surface, in->connectivity_surface + (size_t)4 * in->etop);
    for (int m = 0; m < 3; ++m)
        for (int k = 0; k < in->etop; ++k) pm.conn_surf[(size_t)m * in->etop + k] = nmap[in->connectivity_surface[(size_t)m * in->etop + k]];
    pm.top_elems.resize((size_t)in->ntop_elems);
    for (int i = 0; i < in->ntop_elems; ++i) pm.top_elems[i] = emap[in->top_elems[i]];
    pm.view.connectivity = pm.conn.data();
    pm.view.support_idx = pm.sup_idx.data(); pm.view.support_arr = pm.sup_arr.data(); pm.view.support_lidx = pm.sup_lidx.data();
    pm.view.bcflag = pm.bcflag.data();
    pm.view.top_nodes = pm.top_nodes.data();
    pm.view.connectivity_surface = pm.conn_surf.data();
    pm.view.top_elems = pm.top_elems.data();
    pm.view.coord = nullptr;
}

// which index space a plain field lives in: 1 nodal, 2 elemental, 0 neither (surface lists)
int field_space(int field)
{
    switch (field) {
    case DES_F_FORCE: case DES_F_FORCE_RESIDUAL: case DES_F_COORD0: case DES_F_VOLUME_N: case DES_F_TMASS:
    case DES_F_DHACC: case DES_F_NTMP: case DES_F_COORD_AVG0: return 1;
    case DES_F_STRESS: case DES_F_STRAIN: case DES_F_STRAIN_RATE: case DES_F_PLSTRAIN: case DES_F_DELTA_PLSTRAIN:
    case DES_F_VISCOSITY: case DES_F_VOLUME: case DES_F_VOLUME_OLD: case DES_F_DPRESSURE: case DES_F_RADIOGENIC:
    case DES_F_STRESS_AVG: case DES_F_DPLSTRAIN_AVG: case DES_F_STRAIN0: return 2;
    default: return 0;
    }
}

// SoA planes [ncomp][n] (or rows of `row` items when ncomp == 0) between the caller's numbering
// and the engine's; `to_dev`: out[new] = in[new2old[new]], else out[new2old[new]] = in[new]
template <typename T>
void permute_planes(const T *in, T *out, size_t n, size_t ncomp, size_t row, const std::vector<int> &new2old, bool to_dev)
{
    if (ncomp == 0) {                                   // AoS rows (elemmarkers)
        for (size_t i = 0; i < n; ++i) {
            const size_t o = (size_t)new2old[i];
            const T *src = in + (to_dev ? o : i) * row;
            T *dst = out + (to_dev ? i : o) * row;
            for (size_t k = 0; k < row; ++k) dst[k] = src[k];
        }
        return;
    }
    for (size_t c = 0; c < ncomp; ++c)
        for (size_t i = 0; i < n; ++i) {
            const size_t o = (size_t)new2old[i];
            if (to_dev) out[c*n + i] = in[c*n + o]; else out[c*n + o] = in[c*n + i];
        }
}

struct FieldInfo { int kind; long long count; };   // kind: 0 none, 1 elem plane array, 2 nodal plane array, ...

} // namespace des_hip

using namespace des_hip;

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
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm) ncclCommDestroy(h->comm);
    for (ProfRec &r : h->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    void *ptrs[] = { h->d_p, h->d_vt, h->d_clk, h->conn, h->sup_idx, h->sup_pack, h->bcflag, h->xt, h->vm,
        h->ntmp, h->volume_n, h->tmass, h->ymass, h->force, h->fres, h->coord0, h->dhacc, h->dh_n, h->d_red, h->d_n_new2old, h->d_e_new2old,
        h->d_send_idx, h->d_recv_idx, h->d_sendbuf, h->d_recvbuf, h->d_esend_idx, h->d_erecv_idx, h->d_send_noff,
        h->d_send_eoff, h->d_recv_noff, h->d_recv_eoff, h->stress, h->strain,
        h->strain_rate, h->plstrain, h->delta_plstrain, h->viscosity, h->volume, h->volume_old, h->dpressure,
        h->stress_avg, h->dplstrain_avg, h->strain0, h->coord_avg0,
        h->radiogenic, h->markers, h->props, h->mono, h->defer_list, h->ptab, h->mrec, h->ttmp, h->etmp2, h->ftmp, h->res_part, h->bcf_elem,
        h->bcf_facet, h->bcf_kind, h->bcf_val, h->bcf_tmp, h->bcn_idx, h->bcn_ent, h->top_nodes, h->ean,
        h->conn_surf, h->ssup_idx, h->ssup_arr, h->topflag, h->dh, h->edvacc,
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
    if (params->ndims != 3) { *err = DES_ERR_UNSUPPORTED_DIM; g_last_error = "only the 3D (THREED) path is offloaded"; return nullptr; }
    if (params->nmat < 1 || params->nmat > DES_MAX_MAT) { *err = DES_ERR_CONFIG_VALUE; g_last_error = "bad nmat"; return nullptr; }
    switch (params->rheol_type) {
    case DES_RH_ELASTIC: case DES_RH_VISCOUS: case DES_RH_MAXWELL: case DES_RH_EP: case DES_RH_EVP: break;
    default: *err = DES_ERR_UNSUPPORTED; g_last_error = "rheology not offloaded"; return nullptr;
    }
    if (des_dev_device_count() <= device) { *err = DES_ERR_UNSUPPORTED; g_last_error = "no such HIP device"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *err = DES_ERR_UNSUPPORTED; g_last_error = "hipSetDevice failed"; return nullptr; }

    des_dev *h = new des_dev();       // value-initialised: every pointer/scalar member starts at 0
    h->device = device;
    h->p = *params;
    {
        const char *e2d = std::getenv("DES_E2_DEFER");
        h->e2_defer = (e2d && (e2d[0] == '0' || e2d[0] == '1')) ? e2d[0] - '0' : 2;
        h->e2_two_pass = h->e2_defer != 0;
        const char *env = std::getenv("DES_LIBM");
        h->portable_libm = env && std::strcmp(env, "portable") == 0;
        if (env && !h->portable_libm && std::strcmp(env, "ocml") != 0) {
            *err = DES_ERR_CONFIG_VALUE; g_last_error = "DES_LIBM must be 'ocml' or 'portable'"; delete h; return nullptr;
        }
    }
    PermMesh pm;
    {
        const char *env = std::getenv("DES_REORDER");
        if (mesh->coord && mesh->nnode > 0 && mesh->nelem > 0 && !(env && env[0] == '0')) {
            build_perm_mesh(mesh, pm);
            h->n_new2old.swap(pm.n_new2old); h->n_old2new.swap(pm.n_old2new);
            h->e_new2old.swap(pm.e_new2old); h->e_old2new.swap(pm.e_old2new);
            mesh = &pm.view;             // everything below builds the device state in the internal order
        }
    }
    const int nn = h->nn = mesh->nnode, ne = h->ne = mesh->nelem, nmat = h->nmat = params->nmat;
    h->markers_dirty = true;
    h->pending_c = true;
    h->const_mass = params->is_quasi_static && params->nmat == 1;
    h->o0 = 0; h->o1 = nn; h->nn_global = nn; h->nnbr = 0; h->comm = nullptr; h->comm_rank = 0; h->comm_size = 1;

#define CK(x) do { int rc_ = (x); if (rc_ != DES_OK) { *err = rc_; des_dev_destroy(h); return nullptr; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_last_error = std::string(#x) + ": " + hipGetErrorString(e_); \
                   *err = DES_ERR_RESOURCE; des_dev_destroy(h); return nullptr; } } while (0)
    HK(hipStreamCreate(&h->stream));
    HK(hipEventCreate(&h->ev0)); HK(hipEventCreate(&h->ev1));
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
    // fields
    CK(dev_alloc(h->xt, (size_t)nn)); CK(dev_alloc(h->vm, (size_t)nn));
    CK(dev_alloc(h->ntmp, (size_t)nn)); CK(dev_alloc(h->volume_n, (size_t)nn)); CK(dev_alloc(h->tmass, (size_t)nn));
    CK(dev_alloc(h->ymass, (size_t)nn)); CK(dev_alloc(h->force, (size_t)3*nn)); CK(dev_alloc(h->fres, (size_t)3*nn));
    CK(dev_alloc(h->coord0, (size_t)3*nn)); CK(dev_alloc(h->dhacc, (size_t)nn)); CK(dev_alloc(h->dh_n, (size_t)nn));
    CK(dev_alloc(h->d_red, 8));
    CK(dev_alloc(h->stress, (size_t)6*ne)); CK(dev_alloc(h->strain, (size_t)6*ne)); CK(dev_alloc(h->strain_rate, (size_t)6*ne));
    CK(dev_alloc(h->plstrain, (size_t)ne)); CK(dev_alloc(h->delta_plstrain, (size_t)ne)); CK(dev_alloc(h->viscosity, (size_t)ne));
    CK(dev_alloc(h->volume, (size_t)ne)); CK(dev_alloc(h->volume_old, (size_t)ne)); CK(dev_alloc(h->dpressure, (size_t)ne));
    CK(dev_alloc(h->radiogenic, (size_t)ne)); CK(dev_alloc(h->markers, (size_t)ne * nmat));
    CK(dev_alloc(h->mono, (size_t)ne));
    CK(dev_alloc(h->defer_list, (size_t)ne));
    if (nmat > 1) {
        CK(dev_alloc(h->props, (size_t)5*ne));
        CK(dev_alloc(h->ptab, (size_t)nmat * DES_PTAB_CNT * 5));
        hipLaunchKernelGGL(k_ptab, dim3((nmat * DES_PTAB_CNT + 63) / 64), dim3(64), 0, h->stream, h->d_p, h->ptab);
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
    h->n3_blocks = node_grid(h);
    CK(dev_alloc(h->res_part, (size_t)h->n3_blocks));
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
            std::vector<unsigned char> flag((size_t)ne, 0);
            for (int i = 0; i < h->ntop_elems; ++i) flag[mesh->top_elems[i]] = 1;
            CK(dev_alloc(h->topflag, (size_t)ne)); CK(dev_upload(h->topflag, flag.data(), (size_t)ne, h->stream));
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
    const int space = field_space(field);
    if (space == 0 || h->n_new2old.empty()) return dev_upload(dst, (const double *)host, (size_t)count, h->stream);
    const size_t n = space == 1 ? (size_t)h->nn : (size_t)h->ne;
    std::vector<double> tmp((size_t)count);
    permute_planes((const double *)host, tmp.data(), n, (size_t)count / n, 0, space == 1 ? h->n_new2old : h->e_new2old, true);
    return dev_upload(dst, tmp.data(), (size_t)count, h->stream);
}

int des_dev_download(des_dev *h, int field, void *host, long long count)
{
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
    if (!h) return DES_ERR_INTERNAL;
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int des_dev_init_geometry(des_dev *h)
{
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
                       h->bcflag, h->bnormals, h->edge_vec, h->edge_slot, h->vm);
    // compute_mass (dynearthsol.cxx:194)
    launch_mass_gather(h);
    HIP_OK(hipStreamSynchronize(h->stream));
    HIP_OK(hipGetLastError());
    return DES_OK;
}

int des_dev_compute_dt(des_dev *h, double *dt)
{
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
int des_dev_step(des_dev *h, int nsteps, des_scalars *out)
{
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    refresh_props(h);
    const bool multi = h->nnbr > 0;
    const bool iso = h->iso;
    const bool nmd = h->p.is_using_mixed_stress && !iso;
    int rc;
    for (int i = 0; i < nsteps; ++i) {
        const long long step_no = iso ? h->steps_host : ++h->steps_host;
        if (i == 0) launch_e1<MODE_A>(h);
        launch_n1(h);
        launch_e2(h);
        if (nmd) launch_n2(h);
        launch_e3(h);
        launch_n3(h);
        launch_s2(h, step_no);
        if (multi) {
            launch_s3(h, true, false, false);                      // commit the surface heights
            if ((rc = exchange(h))) return rc;
            launch_s3(h, false, true, true);
        } else {
            launch_s3(h, true, true, true);
        }
        const bool last = (i == nsteps - 1);
        if (iso) {                                         // no averaging, no compute_dt in that loop
            if (last) launch_e1<MODE_C>(h); else launch_e1<MODE_C | MODE_A>(h);
            continue;
        }
        const bool do_dt = (step_no % 10 == 0);
        launch_avg_coord0(h, step_no);
        launch_e1_end(h, step_no, !last);
        if (do_dt && (rc = reduce_dt(h))) return rc;
    }
    // compute_mass gather of the last update_mesh, so that volume_n / mass / tmass hold the
    // reference's end-of-step values (inside a multi-step call it is fused into the next N1)
    if (nsteps > 0) launch_mass_gather(h);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { g_last_error = std::string("kernel launch: ") + hipGetErrorString(le); return DES_ERR_RESOURCE; }
    if (out) {
        if (h->comm_size > 1) {
            // l2_residual is a sum over all ranks' owned nodes
            HIP_OK(hipMemcpyAsync(h->d_red + 6, &h->d_clk->l2_sum, 8, hipMemcpyDeviceToDevice, h->stream));
            ncclAllReduce(h->d_red + 6, h->d_red + 6, 1, ncclDouble, ncclSum, h->comm, h->stream);
        }
        rc = sync_clock(h);
        if (rc) return rc;
        choose_e2_mode(h);
        const DevClock &c = *h->h_clk;
        out->dt = c.dt; out->time = c.time; out->l2_residual = c.l2_residual; out->max_surf_vel = c.max_surf_vel;
        if (h->comm_size > 1) {
            double l2sum = 0;
            HIP_OK(hipMemcpy(&l2sum, h->d_red + 6, 8, hipMemcpyDeviceToHost));
            out->l2_residual = std::sqrt(l2sum);
        }
        out->max_global_vel_mag = c.max_global_vel_mag; out->global_dt_min = c.global_dt_min;
        out->steps = c.steps; out->status = c.status; out->n_return_mapping = c.n_defer; out->avg_time0 = c.avg_time0;
        return c.status;
    }
    return DES_OK;
}

// ---- domain decomposition ---------------------------------------------------------
int des_dev_set_halo(des_dev *h, const des_halo *halo, int nnode_global)
{
    if (!h || !halo) return DES_ERR_INTERNAL;
    if (halo->owned_begin < 0 || halo->owned_end > h->nn || halo->owned_begin >= halo->owned_end) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    h->o0 = halo->owned_begin; h->o1 = halo->owned_end; h->nn_global = nnode_global;
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
    h->n3_blocks = node_grid(h);
    return dev_alloc(h->res_part, (size_t)h->n3_blocks);
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
    return DES_OK;
}

// The exchange of the ghost region through the attached communicator (what des_dev_step issues
// between the two phases of a step); asynchronous on the engine's stream.
int des_dev_exchange(des_dev *h)
{
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    return exchange(h);
}

// One phase of a step without any communication: the caller moves the ghost-region state
// (des_dev_halo_pack / des_dev_halo_unpack) -- used to test the decomposition with several
// engines on one GPU.  Returns 1 after phase 1 when the compute_dt partials are ready.
int des_dev_phase(des_dev *h, int phase)
{
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
        launch_e3(h);
        launch_n3(h);
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
                           h->xt, h->vm, h->dh_n, h->stress, h->strain, h->plstrain, h->ne, d_buf);
        HIP_OK(hipMemcpyAsync(buf, d_buf, (size_t)n * w * 8, hipMemcpyDeviceToHost, h->stream));
    } else {
        if ((rc = dev_upload(d_buf, buf, (size_t)n * w, h->stream))) return rc;
        hipLaunchKernelGGL(k_state_unpack, dim3(nblk(n)), dim3(DES_BLOCK), 0, h->stream, nn_items, d_idx, d_off, ne_items, d_idx, d_off,
                           h->xt, h->vm, h->dh_n, h->stress, h->strain, h->plstrain, h->ne, d_buf);
    }
    HIP_OK(hipStreamSynchronize(h->stream));
    hipFree(d_idx); hipFree(d_off); hipFree(d_buf);
    return DES_OK;
}

int des_dev_halo_pack(des_dev *h, int what, const int *idx, int n, double *buf)
{
    return state_io(h, what, idx, n, buf, true);
}

int des_dev_halo_unpack(des_dev *h, int what, const int *idx, int n, const double *buf)
{
    return state_io(h, what, idx, n, const_cast<double *>(buf), false);
}

int des_dev_dt_partials(des_dev *h, double out[6], int recompute)
{
    if (!h) return DES_ERR_INTERNAL;
    hipSetDevice(h->device);
    if (recompute) { refresh_props(h); launch_e1<MODE_DT>(h); }
    hipLaunchKernelGGL(k_dt_pack, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red);
    HIP_OK(hipMemcpyAsync(out, h->d_red, 48, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int des_dev_dt_finalize(des_dev *h, const double in[6], double *dt)
{
    if (!h) return DES_ERR_INTERNAL;
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
    if (fn < DES_LIBM_POW || fn > DES_LIBM_ATAN2 || n < 0 || !x || !out) return DES_ERR_INTERNAL;
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

int des_dev_check_nan(des_dev *h, long long *n_nan)
{
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

int des_dev_timer_start(des_dev *h)
{
    if (!h) return DES_ERR_INTERNAL;
    HIP_OK(hipEventRecord(h->ev0, h->stream));
    return DES_OK;
}

int des_dev_timer_stop(des_dev *h, float *ms)
{
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
    double be = 1420, bn = 348;
    if (h->p.rheol_type == DES_RH_EVP) { be += 24; bn += 8; }
    if (!h->p.has_thermal_diffusion) { be -= 88; bn -= 24; }
    if (!h->p.is_using_mixed_stress) { be -= 96; bn -= 28; }
    // average_fields: stress_avg read+write, delta_plstrain read, dplstrain_avg read+write
    if (h->p.is_outputting_averaged_fields) be += 120;
    return be * h->ne + bn * h->nn;
}

} // extern "C"
