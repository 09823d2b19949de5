// des_dev2d.hip -- the 2-D (triangle) build of the explicit time step on gfx950 (des_dev2d.hpp).
//
// One kernel per loop of the reference's 2-D build, in the reference's order; node loops sum their
// element patch through the same Support CSR in the same ascending order (parameters.hpp:585-610),
// so with -ffp-contract=off every field equals the CPU build's bit for bit, as the 3-D engine's do.
// Arrays are the reference's own (SoA, caller's numbering): coord / vel / force [2][nnode] =
// {x, z}; stress / strain / strain_rate [3][nelem] = {XX, ZZ, XZ}; connectivity [3][nelem].
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "engine/env.hpp"
#include "engine/selfcheck.hpp"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#define DES_LIBM_LDS_TABLES 1     // the stress kernel stages the libm tables in LDS first
#define DES_LIBM_LDS_WAVES 4      // = DES_BLOCK / 64
#include "des_kernels.hpp"
#include "des_dev2d.hpp"

namespace des2d {

namespace {

const int ND = 2, NPE = 3, NSTR = 3, NPF = 2;
__device__ const int NODE_OF_FACET_D[3][2] = {{1,2},{2,0},{0,1}};     // constants.hpp:71-75
const unsigned BOUNDX0 = 1u, BOUNDX1 = 2u, BOUNDZ0 = 1u << 4, BOUNDZ1 = 1u << 5, BOUND_ANY = 0x3ffu;
const int iboundx0 = 0, iboundz0 = 4, iboundz1 = 5, iboundn0 = 6, iboundn3 = 9;
#define DES2_YEAR2SEC (365.2422 * 86400)                               /* constants.hpp:76 */

// device-resident clock, reduction slots and the per-step scalars of apply_vbcs
struct Clock {
    double dt, time, l2_residual, max_surf_vel, max_global_vel_mag, global_dt_min;
    double r_minl, r_dt_maxwell, r_dt_diffusion, r_global_dt_min, r_max_vem;   // compute_dt (geometry.cxx:1490-1503)
    double maxdh, l2_sum;
    double x0_max, x0_min, zmin;       // bc.cxx:251-290, 350-361
    double avg_time0;
    long long steps;
    int status, iso, n_past, x0_init;
    int pt;                            // inside the pseudo-transient loop of a step (Param::control.PT_jump)
    int l2_global;                     // l2_residual / l2_sum hold the GLOBAL block sum (k2_residual_final), not this rank's owned share
    double dt_prev;                    // the dt a compute_dt step ran with (its end-of-step rotation may ride in the next stress update)
};

inline int nblk(long long n) { return (int)((n + DES_BLOCK - 1) / DES_BLOCK); }

} // namespace

struct Engine {
    int device = 0;
    int portable_libm = 1;
    des_params p;
    int nn = 0, ne = 0, nmat = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    des_params *d_p = nullptr;
    desk::ViscTerms *d_vt = nullptr;
    Clock *d_clk = nullptr, *h_clk = nullptr;
    // topology
    int *conn = nullptr, *sup_idx = nullptr, *sup_arr = nullptr, *sup_lidx = nullptr;
    unsigned *bcflag = nullptr;
    int nbf[DES_NBDRY] = {0}, nbn[DES_NBDRY] = {0};
    int *bf_elem[DES_NBDRY] = {nullptr}, *bf_facet[DES_NBDRY] = {nullptr}, *bnodes[DES_NBDRY] = {nullptr};
    int4 *binc[DES_NBDRY] = {nullptr};         // [2 * nbn] per boundary with facets: the boundary-facet incidences of its nodes (k2_sbc_direct)
    std::vector<int4> h_binc[DES_NBDRY];       // (host copies, create only)
    double *bnormals = nullptr, *edge_vec = nullptr; int *edge_slot = nullptr;
    int ntop = 0, etop = 0, ntop_elems = 0;
    int *top_nodes = nullptr, *ean = nullptr, *conn_surf = nullptr, *top_elems = nullptr;
    // nodal
    double *coord = nullptr, *vel = nullptr, *force = nullptr, *fres = nullptr, *coord0 = nullptr, *temperature = nullptr,
           *volume_n = nullptr, *mass = nullptr, *tmass = nullptr, *ymass = nullptr, *dhacc = nullptr, *ntmp = nullptr,
           *total_dx = nullptr, *total_slope = nullptr;
    // element
    double *stress = nullptr, *strain = nullptr, *strain_rate = nullptr, *stressyy = nullptr, *plstrain = nullptr,
           *delta_plstrain = nullptr, *viscosity = nullptr, *volume = nullptr, *volume_old = nullptr, *dpressure = nullptr,
           *edvoldt = nullptr, *radiogenic = nullptr, *etmp = nullptr, *tmp_result = nullptr /* [6][ne] */, *props = nullptr /* [5][ne] */;
    int *markers = nullptr, *etmp_int = nullptr;
    int *mono = nullptr;               // per element: material << 16 | count of a single-material element, -1 mixed (k2_props)
    double *pptab = nullptr;           // plastic_props of single-material elements outside the weakening range (k2_pptab)
    double *dh = nullptr, *edvacc = nullptr;
    double *stress_avg = nullptr, *dplstrain_avg = nullptr, *strain0 = nullptr, *coord_avg0 = nullptr;
    double *res_part = nullptr; int res_nb = 0;
    double *neg_zmin = nullptr;        // scratch of k2_vbc_zmin
    // node-block patch passes (des_dev2d_patch.hpp); DES2D_PATCH=0 or a mesh outside their LDS caps: the plain kernels
    bool patch = false, res_fin_pending = false, tick_pending = false;
    // DES2D_FOLD (round 5): a plain step's stress bcs, damping / velocity / vbcs / coordinates and residual partials inside
    // k2p_force's node phase (k2p_force<1>); fold_ok: this model's boundary loads can all be formed per node (create)
    bool fold_on = true, fold_ok = false;
    double *coord_alt = nullptr;                // the other buffer of the coordinate pair (k2p_force<1> writes the moved nodes there)
    int *sbcn_idx = nullptr; int4 *sbcn_ent = nullptr;      // per node: its boundary-facet incidences {element, facet, which node, boundary}
    // (round 5) a plain step inside a multi-step call leaves its surface step to the next step's first two passes as well
    // (one_step: surf_late -> surf_pending; DES2D_SURF_DEFER=0: off); topflag: the top elements
    bool surf_defer_on = true, surf_late = false, surf_pending = false;
    // (round 5) ... and a compute_dt step inside a call leaves its end-of-step element pass and compute_mass behind like any other
    // (the single engine; DES2D_DT_DEFER=0: off): compute_dt forms the volumes it wants from the coordinates, the rotation keeps its dt
    bool dt_defer_on = true, rot_prev_dt = false;
    // (round 5) the strain rate of a step whose stress update also finishes the step before (k2_stress<M, 2>) is formed THERE, from
    // the coordinates and velocities that pass gathers anyway -- k2p_temp_dvoldt<1> does not store it (the single engine;
    // DES2D_SR_FUSE=0: off).  sr_fused: decided for this step's two launches together.
    bool sr_fuse_on = true, sr_fused = false;
    unsigned char *topflag = nullptr;
    int *pt_ptr = nullptr, *pt_zero = nullptr; int4 *pt_ent = nullptr; bool surf_defer_fits = true;
    int *d_bperm = nullptr;                      // the blocks in launch order: every XCD its share of the surface blocks, first
    unsigned char *d_surfpre = nullptr;          // (a SurfPre, des_dev2d_patch.hpp)
    double2 *xz_pre = nullptr; int *fold_top_pos = nullptr; bool xz_pre_valid = false;   // k2p_force<1>: the moved top nodes, in top_nodes order (k2_surf_commit)
    bool radiogenic_zero = true;                // every heat source is +0.0 (the array starts zeroed; upload looks at what it is given)
    double *dt_part = nullptr;                  // [5][nblk(ne)] compute_dt partials, one slot per workgroup of k2_dt_partials
    int res_count = 0;                          // partials in res_part[] (blocks of 256 owned nodes, or the patch blocks)
    bool geo_on = true, elide_on = true;       // DES2D_GEO / DES2D_ELIDE != 0 (read at create)
    bool elide = false;                        // this step's output-only element stores can go (a later step of the same call rewrites them)
    bool geo_pending = false;                  // compute_volume + rotate_stress of the last step left to the next k2_stress<M, 2>
    bool mass_fuse_on = true, mass_pending = false;   // DES2D_MASS_FUSE != 0; compute_mass of the last step left to the next k2p_temp_dvoldt<1>
    bool no_neumann = false;                   // initial_body_force_adjustment: Neumann tractions held back (fields.cxx:690)
    int p_npb = 0, p_nb = 0, p_pn_cap = 0, p_inc_cap = 0, p_pe_cap = 0;
    bool it3_forced = false;                   // DES2D_PATCH_IT=3: the three-elements-per-lane forms of the patch passes whatever the mesh (tests)
    int *po_ptr = nullptr, *po_id = nullptr, *po_slot = nullptr, *pe_ptr = nullptr, *pn_ptr = nullptr, *pn_id = nullptr;
    ulonglong2 *pe_pack = nullptr;
    double *temperature_alt = nullptr;         // the other buffer of the temperature pair (k2p_temp_dvoldt)
    double *stress_pre = nullptr;              // the stress between update_stress and NMD_stress (k2p_force)
    // domain decomposition (set_halo): this engine holds one node slab + its ghost region
    bool halo = false, halo_set = false;       // halo: the mesh has neighbours (set_halo with nnbr > 0)
    int o0 = 0, o1 = 0, nn_global = 0;         // owned nodes [o0, o1) of nn; the residual's divisor is global
    // the partition-independent residual of the pseudo-transient loop (des_params.h: DES_RES_BLOCK; the 3-D engine's
    // engine/residual.hpp): global id of the first owned node, the GLOBAL block array (own part computed, the rest received)
    int g0 = 0;
    double *res_blocks = nullptr; int res_nb_global = 0;
    int *top_pos = nullptr;                    // [nn] position of a node in top_nodes, -1 below the surface
    int nnbr = 0;
    std::vector<int> nbr_rank, send_ptr, recv_ptr, esend_ptr, erecv_ptr;
    std::vector<long long> send_off, recv_off; // [nnbr + 1] doubles: a message = node records, then element records
    int *d_send_idx = nullptr, *d_send_noff = nullptr, *d_esend_idx = nullptr, *d_send_eoff = nullptr;
    int *d_recv_idx = nullptr, *d_recv_noff = nullptr, *d_erecv_idx = nullptr, *d_recv_eoff = nullptr;
    double *d_sendbuf = nullptr, *d_recvbuf = nullptr;
    double *d_red = nullptr;                   // [8] scratch of the cross-rank reductions
    double *h_red = nullptr;                   // pinned copy
    ncclComm_t comm = nullptr;                 // set_comm: the exchange and the two reductions run inside step() on RCCL (not owned)
    bool markers_dirty = true, iso = false;
    bool count_past = false;           // this step feeds des_scalars::n_return_mapping (the last one of a call)
    int *past_part = nullptr; int past_cap = 0, past_base = 0, n_past_host = 0; bool past_used = false;   // its per-wavefront counts (k2_stress)
    std::vector<int> past_host;
    long long steps_host = 0;
    long long n_pt_iterations = 0;   // pseudo-transient iterations of the current step() call
    bool pt_defer = false;           // step_front stops in front of the pseudo-transient loop (step_group, des_dev_phase)
    // Overlapped schedule of a decomposed mesh (round 4; DES_OVERLAP=1 or set_overlap): transfer, unpack and the wall's
    // reduction of step t on a side stream beside compute_mass, update_temperature + compute_dvoldt and update_stress of
    // step t + 1 on the blocks / elements far from the cut (set_halo: d_blist [0, nb_deep), d_elist [0, ne_deep)).
    bool overlap = false, join_pending = false, wall_pending = false, far_issued = false;
    const double *mT_in = nullptr; double *mT_out = nullptr;     // the two temperature buffers of the step whose far part is out
    hipStream_t xstream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_wall = nullptr;
    int *d_blist = nullptr, *d_elist = nullptr;
    int nb_deep = 0, ne_deep = 0;
    long long back_steps = 0;        // the step whose surface bookkeeping waits behind the join
    // host copies for set_halo's classification: the patch lists (own nodes, other nodes, elements of every block), the
    // connectivity, the surface lists
    std::vector<int> hp_po_ptr, hp_po_id, hp_pn_ptr, hp_pn_id, hp_pe_ptr, hp_pe_elem;
    std::vector<int> h_conn, h_top_nodes, h_top_elems;
    std::vector<void *> allocs;
    std::string err;
    // per-kernel HIP-event accounting of the main launches (des_dev_profile_enable / _read; off: no events at all)
    bool prof = false;
    struct ProfRec { hipEvent_t a, b; int k; };
    std::vector<ProfRec> prof_recs;
    double prof_ms[8] = {0};
    long long prof_calls[8] = {0};
};

static int exchange_rccl(Engine *h);       // (defined behind the anonymous namespace: the pseudo-transient loop on RCCL calls it)

namespace {

#define HIP2(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    h->err = std::string(#call) + ": " + hipGetErrorString(e_); return DES_ERR_RESOURCE; } } while (0)

template <typename T>
int dalloc(Engine *h, T *&ptr, size_t count)
{
    void *d = nullptr;
    HIP2(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)));
    HIP2(hipMemset(d, 0, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocs.push_back(d);
    ptr = (T *)d;
    return DES_OK;
}

template <typename T>
int dcopy(Engine *h, T *&ptr, const T *host, size_t count)
{
    int rc = dalloc(h, ptr, count);
    if (rc) return rc;
    if (count && host) HIP2(hipMemcpy(ptr, host, count * sizeof(T), hipMemcpyHostToDevice));
    return DES_OK;
}

// ---------------------------------------------------------------------------------
// device functions: the !THREED branches
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double trace2(const double *s) { return s[0] + s[1]; }                  // utils.hpp:211-219
__device__ __forceinline__ double second_invariant2_2d(const double *t)                             // utils.hpp:222-231
{
    return 0.25*(t[0]-t[1])*(t[0]-t[1]) + t[2]*t[2];
}

// geometry.cxx:77-93 (!THREED)
__device__ __forceinline__ double triangle_area(const double *a, const double *b, const double *c)
{
    double ab0 = b[0] - a[0], ab1 = b[1] - a[1];
    double ac0 = c[0] - a[0], ac1 = c[1] - a[1];
    return fabs(ab0*ac1 - ab1*ac0) / 2;
}

__device__ __forceinline__ double dist2(const double *a, const double *b)                             // geometry.cxx:16-25
{
    double sum = 0;
    for (int i = 0; i < 2; ++i) { double d = b[i] - a[i]; sum += d * d; }
    return sum;
}

__device__ __forceinline__ void elem_coords(const double *coord, const int *conn, int nn, int ne, int e, double d[3][2])
{
    for (int i = 0; i < 3; ++i) {
        const int n = conn[i * ne + e];
        d[i][0] = coord[n]; d[i][1] = coord[nn + n];
    }
}

// fields.cxx:40-53
__device__ __forceinline__ void shape_fn2(const double d[3][2], double vol, double shpdx[3], double shpdz[3])
{
    double iv = 1.0 / (2.0 * vol);
    shpdx[0] = iv * (d[1][1] - d[2][1]);
    shpdx[1] = iv * (d[2][1] - d[0][1]);
    shpdx[2] = iv * (d[0][1] - d[1][1]);
    shpdz[0] = iv * (d[2][0] - d[1][0]);
    shpdz[1] = iv * (d[0][0] - d[2][0]);
    shpdz[2] = iv * (d[1][0] - d[0][0]);
}

// The five property means of an element (refresh_elem_cache, matprops.cxx:259-303: bulk modulus 0, shear modulus 1, porosity 2,
// heat capacity 3, conductivity 4).  With ONE material the means ARE its values (matprops.cxx:118, 136: k2_props stores exactly
// these) -- nothing to fetch: a wave-uniform branch instead of an 8-byte gather per element, pass and property (round 5).
__device__ __forceinline__ double prop2(const des_params *__restrict__ p, const double *props, int ne, int e, int w)
{
    if (p->nmat == 1)
        return w == 0 ? p->bulk_modulus[0] : (w == 1 ? p->shear_modulus[0] : (w == 2 ? p->porosity[0] : (w == 3 ? p->heat_capacity[0] : p->therm_cond[0])));
    return props[(size_t)w * ne + e];
}

__device__ __forceinline__ double elemT(const double *temperature, const int *conn, int ne, int e)    // matprops.cxx:338-343
{
    double T = 0;
    for (int i = 0; i < 3; ++i) T += temperature[conn[i * ne + e]];
    T /= 3;
    return T;
}

// matprops.cxx:333-377 with the 2-D trace / second invariant
template <class M>
__device__ __forceinline__ double mat_visc2(const des_params *__restrict__ p, const desk::ViscTerms *vt, const desk::Mix &mx,
                                            double T, const double *s, const double *edot3)
{
    const double min_strain_rate = 1e-30;
    double s0 = trace2(s) / 2;
    double edot = sqrt(second_invariant2_2d(edot3));
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

// rheology.cxx:248-260
__device__ __forceinline__ void elastic2(double bulkm, double shearm, const double *de, double *s)
{
    double lambda = bulkm - 2. / 3 * shearm;
    double dev = trace2(de);
    for (int i = 0; i < 2; ++i) s[i] += 2 * shearm * de[i] + lambda * dev;
    s[2] += 2 * shearm * de[2];
}

// rheology.cxx:277-295
__device__ __forceinline__ void maxwell2(double bulkm, double shearm, double viscosity, double dt, double dv,
                                         const double *de, double *s)
{
    double tmp = 0.5 * dt * shearm / viscosity;
    double f1 = 1 - tmp;
    double f2 = 1 / (1 + tmp);
    double dev = trace2(de) / 2;
    double s0 = trace2(s) / 2;
    for (int i = 0; i < 2; ++i)
        s[i] = ((s[i] - s0) * f1 + 2 * shearm * (de[i] - dev)) * f2 + s0 + bulkm * dv;
    s[2] = (s[2] * f1 + 2 * shearm * de[2]) * f2;
}

// rheology.cxx:298-310
__device__ __forceinline__ void viscous2(double bulkm, double viscosity, double total_dv, const double *edot, double *s)
{
    double dev = trace2(edot) / 2;
    for (int i = 0; i < 2; ++i) s[i] = 2 * viscosity * (edot[i] - dev) + bulkm * total_dv;
    s[2] = 2 * viscosity * edot[2];
}

// rheology.cxx:86-119
__device__ __forceinline__ void principal_stresses2(const double *s, double p[2], double &cos2t, double &sin2t)
{
    double s0 = 0.5 * (s[0] + s[1]);
    double rad = sqrt(second_invariant2_2d(s));
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

// rheology.cxx:312-484, the !THREED branches (pure 2-D Mohr-Coulomb)
__device__ __forceinline__ void elasto_plastic(double bulkm, double shearm, double amc, double anphi, double anpsi,
                                               double hardn, double ten_max, const double *de, double &depls, double *s)
{
    elastic2(bulkm, shearm, de, s);
    depls = 0;
    double p[2];
    double cos2t, sin2t;
    principal_stresses2(s, p, cos2t, sin2t);

    double fs = p[0] - p[1] * anphi + amc;
    double ft = p[1] - ten_max;
    if (fs > 0 && ft < 0) return;

    double pa = sqrt(1 + anphi*anphi) + anphi;
    double ps = ten_max * anphi - amc;
    double hh = p[1] - ten_max + pa * (p[0] - ps);
    double a1 = bulkm + 4. / 3 * shearm;
    double a2 = bulkm - 2. / 3 * shearm;

    double alam;
    if (hh < 0) {
        alam = fs / (a1 - a2*anpsi + a1*anphi*anpsi - a2*anphi + 2*sqrt(anphi)*hardn);
        p[0] -= alam * (a1 - a2 * anpsi);
        p[1] -= alam * (a2 - a1 * anpsi);
        depls = fabs(alam) * sqrt((3 + 2*anpsi + 3*anpsi*anpsi) / 8);
    } else {
        alam = ft / a1;
        p[0] -= alam * a2;
        p[1] -= alam * a1;
        depls = fabs(alam) * sqrt(3. / 8);
    }
    double dc2 = (p[0] - p[1]) * cos2t;
    double dss = p[0] + p[1];
    s[0] = 0.5 * (dss + dc2);
    s[1] = 0.5 * (dss - dc2);
    s[2] = 0.5 * (p[0] - p[1]) * sin2t;
}

// rheology.cxx:486-701 (plane strain)
__device__ __forceinline__ void elasto_plastic2d(double bulkm, double shearm, double amc, double anphi, double anpsi,
                                                 double hardn, double ten_max, const double *de, double &depls,
                                                 double *s, double &syy)
{
    depls = 0;
    double a1 = bulkm + 4. / 3 * shearm;
    double a2 = bulkm - 2. / 3 * shearm;
    double sxx = s[0] + de[1]*a2 + de[0]*a1;
    double szz = s[1] + de[0]*a2 + de[1]*a1;
    double sxz = s[2] + de[2]*2*shearm;
    syy += (de[0] + de[1]) * a2;

    // p[n1], p[n2], p[n3] as scalars: which principal stress is the out-of-plane one
    double p0, p1, p2;
    double cos2t, sin2t;
    int order;                       // 0: syy largest, 1: syy smallest, 2: syy intermediate
    {
        double s0 = 0.5 * (sxx + szz);
        double rad = 0.5 * sqrt((sxx-szz)*(sxx-szz) + 4*sxz*sxz);
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
        if (syy > sii)      { order = 0; p0 = si;  p1 = sii; p2 = syy; }
        else if (syy < si)  { order = 1; p0 = syy; p1 = si;  p2 = sii; }
        else                { order = 2; p0 = si;  p1 = syy; p2 = sii; }
    }

    if (p0 >= ten_max) {
        s[0] = s[1] = syy = ten_max;
        s[2] = 0.0;
        return;
    }
    if (p1 >= ten_max) { p1 = p2 = ten_max; }
    else if (p2 >= ten_max) { p2 = ten_max; }

    double fs = p0 - p2 * anphi + amc;
    if (fs >= 0.0) {
        s[0] = sxx;
        s[1] = szz;
        s[2] = sxz;
        return;
    }

    const double alams = fs / (a1 - a2*anpsi + a1*anphi*anpsi - a2*anphi + hardn);
    p0 -= alams * (a1 - a2 * anpsi);
    p1 -= alams * (a2 - a2 * anpsi);
    p2 -= alams * (a2 - a1 * anpsi);

    depls = 0.5 * fabs(alams + alams * anpsi);

    if (p0 >= ten_max) {
        s[0] = s[1] = syy = ten_max;
        s[2] = 0.0;
        return;
    }
    if (p1 >= ten_max) { p1 = p2 = ten_max; }
    else if (p2 >= ten_max) { p2 = ten_max; }

    {
        // (n1, n2, n3) = (0,1,2) / (1,2,0) / (0,2,1)
        const double pn1 = (order == 1) ? p1 : p0;
        const double pn2 = (order == 0) ? p1 : p2;
        const double pn3 = (order == 0) ? p2 : (order == 1) ? p0 : p1;
        double dc2 = (pn1 - pn2) * cos2t;
        double dss = pn1 + pn2;
        s[0] = 0.5 * (dss + dc2);
        s[1] = 0.5 * (dss - dc2);
        s[2] = 0.5 * (pn1 - pn2) * sin2t;
        syy = pn3;
    }
}

// utils.hpp:259-287
__device__ __forceinline__ double interp1(const double *x, const double *y, int n, double x_new)
{
    double dist = DBL_MAX;
    int idx = -1;
    for (int i = 0; i < n; ++i) {
        double newDist = x_new - x[i];
        if (newDist >= 0 && newDist <= dist) { dist = newDist; idx = i; }
    }
    double slope = 0;
    if (idx < 0) idx = 0;
    else if (idx < n - 1) slope = (y[idx+1] - y[idx]) / (x[idx+1] - x[idx]);
    return slope * (x_new - x[idx]) + y[idx];
}

// bc.cxx:42-50
__device__ __forceinline__ void normal_vector_of_facet(const double fc[2][2], double *normal, double &zcenter)
{
    double v01[2];
    for (int i = 0; i < 2; ++i) v01[i] = fc[1][i] - fc[0][i];
    normal[0] = v01[1];
    normal[1] = -v01[0];
    zcenter = (fc[0][1] + fc[1][1]) / 2;
}

// ---------------------------------------------------------------------------------
// kernels, in the order of a step
// ---------------------------------------------------------------------------------
__global__ void k2_clock(Clock *clk)                                  // dynearthsol.cxx:769-770
{
    clk->steps++;
    clk->time += clk->dt;
}

// The marker mix of element e from its mono[] word (k2_props: material << 16 | count of a single-material element, -1 a mixed
// one): single-material elements answer desk::Mix::count from registers, the others read their nmat counts (des_kernels.hpp)
__device__ __forceinline__ desk::Mix mix2(int mo, const int *markers, int nmat, int e)
{
    desk::Mix mx;
    if (mo >= 0) { mx.mk = nullptr; mx.mat = mo >> 16; mx.cnt = mo & 0xffff; }
    else         { mx.mk = markers + (size_t)e * nmat; mx.mat = -1; mx.cnt = 0; }
    return mx;
}

// MatProps::refresh_elem_cache (matprops.cxx:259-303) + the mono[] word
__global__ void k2_props(const des_params *__restrict__ p, int ne, const int *markers, double *props, int *mono)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    const int *mk = &markers[(size_t)e * p->nmat];
    {
        int used = 0, mat = 0, cnt = 0;
        for (int m = 0; m < p->nmat; ++m) if (mk[m] != 0) { ++used; mat = m; cnt = mk[m]; }
        mono[e] = (used == 1 && cnt > 0 && cnt < 65536) ? ((mat << 16) | cnt) : -1;
    }
    props[e]          = desk::harmonic_mean(p->bulk_modulus, mk, p->nmat);
    props[ne + e]     = desk::harmonic_mean(p->shear_modulus, mk, p->nmat);
    props[2 * ne + e] = desk::arithmetic_mean(p->porosity, mk, p->nmat);
    props[3 * ne + e] = desk::arithmetic_mean(p->heat_capacity, mk, p->nmat);
    props[4 * ne + e] = desk::arithmetic_mean(p->therm_cond, mk, p->nmat);
}

// plastic_props of a single-material element by (material, marker count, weakening regime): desk::plastic_props itself,
// run once per entry with a plastic strain of that regime (des_kernels.hpp; as the 3-D engine's k_pptab)
template <class M>
__global__ void k2_pptab(const des_params *__restrict__ p, double *pptab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nmat = p->nmat;
    if (i >= nmat * DES_PPTAB_CNT * 3) return;
    const int regime = i % 3, cnt = (i / 3) % DES_PPTAB_CNT, mat = i / (3 * DES_PPTAB_CNT);
    desk::Mix mx;
    mx.mk = nullptr; mx.mat = mat; mx.cnt = cnt > 0 ? cnt : 1;           // entry 0 is never read
    const double pls = regime == 0 ? p->pls0[mat] - 1.0 : (regime == 1 ? p->pls0[mat] : p->pls1[mat]);
    double *t = pptab + (size_t)i * 5;
    desk::plastic_props<M>(p, mx, pls, t[0], t[1], t[2], t[3], t[4]);
}

// update_temperature (fields.cxx:197-278), element part
__global__ void k2_temp_elem(const des_params *__restrict__ p, int nn, int ne, const int *conn, const double *coord,
                             const double *temperature, const double *volume, const double *radiogenic,
                             const double *props, const int *markers, double *tmp_result)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    desk::Mix mx = {&markers[(size_t)e * p->nmat], 0, 0};
    double kv = props[4 * ne + e] * volume[e];
    double rh = radiogenic[e] * volume[e] * desk::mat_rho(p, mx, elemT(temperature, conn, ne, e)) / 3;
    double d[3][2], shpdx[3], shpdz[3];
    elem_coords(coord, conn, nn, ne, e, d);
    shape_fn2(d, volume[e], shpdx, shpdz);
    for (int i = 0; i < 3; ++i) {
        double diffusion = 0.;
        for (int j = 0; j < 3; ++j)
            diffusion += (shpdx[i] * shpdx[j] + shpdz[i] * shpdz[j]) * temperature[conn[j * ne + e]];
        tmp_result[i * ne + e] = diffusion * kv - rh;
    }
}

__global__ void k2_temp_node(const des_params *__restrict__ p, const Clock *clk, int nn, int ne, const int *sup_idx, const int *sup_arr,
                             const int *sup_lidx, const unsigned *bcflag, const double *tmp_result, const double *tmass,
                             double *temperature)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n >= nn) return;
    if (bcflag[n] & BOUNDZ1)
        temperature[n] = p->surface_temperature;
    else {
        double tdot = 0;
        for (int k = sup_idx[n]; k < sup_idx[n+1]; ++k)
            tdot += tmp_result[sup_lidx[k] * ne + sup_arr[k]];
        temperature[n] -= clk->dt * tdot / tmass[n];
    }
}

// update_strain_rate (fields.cxx:405-480) + the element part of compute_dvoldt (geometry.cxx:214-224)
__global__ void k2_strain_rate(int nn, int ne, const int *conn, const double *coord, const double *vel,
                               const double *volume, double *strain_rate, double *etmp)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    double d[3][2], shpdx[3], shpdz[3], v[3][2];
    elem_coords(coord, conn, nn, ne, e, d);
    shape_fn2(d, volume[e], shpdx, shpdz);
    elem_coords(vel, conn, nn, ne, e, v);
    double s[3];
    s[0] = 0; for (int i = 0; i < 3; ++i) s[0] += v[i][0] * shpdx[i];
    s[1] = 0; for (int i = 0; i < 3; ++i) s[1] += v[i][1] * shpdz[i];
    s[2] = 0; for (int i = 0; i < 3; ++i) s[2] += 0.5 * (v[i][0] * shpdz[i] + v[i][1] * shpdx[i]);
    for (int i = 0; i < 3; ++i) strain_rate[i * ne + e] = s[i];
    double dj = s[0] + s[1];
    etmp[e] = dj * volume[e];
}

// nodal gather of compute_dvoldt (geometry.cxx:226-237) and of NMD_stress (geometry.cxx:298-308)
__global__ void k2_node_avg(int nn, const int *sup_idx, const int *sup_arr, const double *etmp, const double *volume_n, double *ntmp)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n >= nn) return;
    double acc = 0.;
    for (int k = sup_idx[n]; k < sup_idx[n+1]; ++k) acc += etmp[sup_arr[k]];
    ntmp[n] = acc / volume_n[n];
}

// compute_edvoldt (geometry.cxx:249-279)
__global__ void k2_edvoldt(int ne, const int *conn, const double *ntmp, double *edvoldt)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    double dj = 0;
    for (int i = 0; i < 3; ++i) dj += ntmp[conn[i * ne + e]];
    edvoldt[e] = dj / 3;
}

// update_stress (rheology.cxx:703-1030, non-RSF, non-hydraulic) + the element part of NMD_stress
// FUSED = 1 (the patch path, des_dev2d_patch.hpp): edvoldt (compute_edvoldt, geometry.cxx:249-279) is formed here from the
// nodal values with k2_edvoldt's statements instead of being read back, and the new stress goes to stress_out (another
// buffer when NMD_stress follows).
// FUSED = 2: this pass also finishes the step BEFORE, whose end-of-step element pass was left out (step_back: geo_pending):
// compute_volume after the volume swap (geometry.cxx:170-201) and rotate_stress (fields.cxx:807-900) with k2_rotate_vol's
// statements, from the coordinates and velocities that pass would have seen (nothing but the temperature has moved since)
// -- on the stress it is about to update anyway: one read and one write of stress and strain per step instead of two.
// volume[] still holds the volumes of the step before (the top elements' already corrected, bc.cxx:1670-1687): they are
// this step's volume_old.  The dt is the one that step ran with (a compute_dt step never leaves its rotation behind).
// outs = 0 (a step of a multi-step call that is not its last one): what no pass reads before the next update overwrites it --
// the corrected strain rate (the next k2p_temp_dvoldt stores a new one), edvoldt, viscosity, delta_plstrain, volume_old --
// is not stored: 56 of the pass's 128 B of stores per element.  The last step of every call stores them all.
__device__ __forceinline__ void jaumann_rate_2d(double *s, double dt, double w2);

// RH != 0: the rheology is known at compile time (the kernel holds that law only).
// (p, vt: __restrict__ -- the parameters are never written on the device, and without the promise every store of the pass turns
//  the later reads of them from scalar loads into per-lane ones: 24 in this kernel, 71 -> 65 us at 1.28M triangles; round 5.
//  The same promise for the read-only arrays, tried with it: nothing.)
template <class M, int FUSED = 0, int RH = 0>
__global__ void __launch_bounds__(DES_BLOCK)
k2_stress(const des_params *__restrict__ p, const desk::ViscTerms *__restrict__ vt, Clock *clk, int ne, const int *conn, const double *temperature,
          const double *props, const int *markers, double *edvoldt, double *volume, double *volume_old,
          const double *stress, double *strain, double *strain_rate, double *stressyy, double *plstrain, double *delta_plstrain,
          double *viscosity, double *dpressure, double *etmp, int count_past, const double *ntmp, double *stress_out,
          int nn, const double *coord, const double *vel, int rotate_arg, int outs, const int *elist, int nlist,
          const int *mono, const double *pptab, int *past_part, int past_base, double *stress_out_shear)
{
    M::stage_begin();
    M::stage_end();
    // elist: the nlist elements of this launch (overlapped schedule of a decomposed mesh: the elements far from the cut, then
    // the others); nullptr: elements 0 .. nlist - 1
    const int t = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (t >= nlist) return;
    const int e = elist ? elist[t] : t;
    const double dt = clk->dt;
    // (the marker word is asked for here and USED below, behind the other requests: mix2 branches on it, and in front of
    //  them that was a trip to memory of its own)
    const int mono_e = mono[e];
    const double bulkm = prop2(p, props, ne, e, 0), shearm = prop2(p, props, ne, e, 1);
    // (rotate_arg: bit 0-1 rotate_stress of the step before / with the dt before, bit 2 (round 5): the strain rate is formed HERE
    //  from the coordinates and velocities this pass gathers -- k2_strain_rate's statements; k2p_temp_dvoldt<1> has not stored it)
    const int rotate = rotate_arg & 3;
    const bool sr_here = FUSED == 2 && (rotate_arg & 4);
    double s[3], es[3], edot[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        s[i] = stress[i * ne + e];
        es[i] = strain[i * ne + e];
        if (!sr_here) edot[i] = strain_rate[i * ne + e];
    }
    // (round 5: the mean temperature and the plastic strain the law wants are asked for HERE, beside the other fields -- inside the
    //  law's case they were two more dependent trips to memory in the middle of the pass)
    const int rh_ = RH ? RH : p->rheol_type;
    const bool law_T = rh_ == DES_RH_VISCOUS || rh_ == DES_RH_MAXWELL || rh_ == DES_RH_EVP;
    const bool law_pls = rh_ == DES_RH_EP || rh_ == DES_RH_EVP;
    double Tm = 0, pls0 = 0;
    if (law_pls) pls0 = plstrain[e];
    // ... and every nodal value the pass gathers through the connectivity in ONE round behind it: the coordinates and velocities
    // of the step before's end-of-step part (FUSED = 2), the nodal dvoldt (FUSED), the temperatures
    int cn[3];
    for (int i = 0; i < 3; ++i) cn[i] = conn[i * ne + e];
    double d[3][2], vv[3][2], nt3[3] = {0, 0, 0};
    if (FUSED == 2) for (int i = 0; i < 3; ++i) { d[i][0] = coord[cn[i]]; d[i][1] = coord[nn + cn[i]]; }
    if (FUSED == 2 && (rotate || sr_here)) for (int i = 0; i < 3; ++i) { vv[i][0] = vel[cn[i]]; vv[i][1] = vel[nn + cn[i]]; }
    if (FUSED) for (int i = 0; i < 3; ++i) nt3[i] = ntmp[cn[i]];
    if (law_T) {                                           // elemT's statements (matprops.cxx:338-343)
        double T = 0;
        for (int i = 0; i < 3; ++i) T += temperature[cn[i]];
        T /= 3;
        Tm = T;
    }
    const desk::Mix mx = mix2(mono_e, markers, p->nmat, e);
    double vol, vol_old;
    if (FUSED == 2) {
        vol_old = volume[e];
        vol = triangle_area(d[0], d[1], d[2]);
        volume[e] = vol;
        if (outs) volume_old[e] = vol_old;
        double shpdx[3], shpdz[3];
        if (rotate || sr_here) shape_fn2(d, vol, shpdx, shpdz);
        if (sr_here) {
            // k2_strain_rate's statements (the patch pass forms the same sums from the same staged values for its dvoldt)
            double s0 = 0, s1 = 0, s2 = 0;
            for (int i = 0; i < 3; ++i) s0 += vv[i][0] * shpdx[i];
            for (int i = 0; i < 3; ++i) s1 += vv[i][1] * shpdz[i];
            for (int i = 0; i < 3; ++i) s2 += 0.5 * (vv[i][0] * shpdz[i] + vv[i][1] * shpdx[i]);
            edot[0] = s0; edot[1] = s1; edot[2] = s2;
        }
        if (rotate) {
            double w2 = 0;
            for (int i = 0; i < 3; ++i) w2 += 0.5 * (vv[i][1] * shpdx[i] - vv[i][0] * shpdz[i]);
            // (rotate = 2: the step being finished was a compute_dt step -- it ran with the dt before the new one)
            const double dt_rot = rotate == 2 ? clk->dt_prev : dt;
            jaumann_rate_2d(s, dt_rot, w2);
            jaumann_rate_2d(es, dt_rot, w2);
        }
    } else {
        vol = volume[e];
        vol_old = volume_old[e];
    }
    double edv;
    if (FUSED) {
        double dj = 0;
        for (int i = 0; i < 3; ++i) dj += nt3[i];
        edv = dj / 3;
        if (outs) edvoldt[e] = edv;
    } else
        edv = edvoldt[e];
    double old_s = trace2(s);
    {
        double div = trace2(edot);
        for (int i = 0; i < 2; ++i) edot[i] += (edv - div) / 2;
    }
    if (outs) for (int i = 0; i < 3; ++i) strain_rate[i * ne + e] = edot[i];
    for (int i = 0; i < 3; ++i) es[i] += edot[i] * dt;
    double de[3];
    for (int i = 0; i < 3; ++i) de[i] = edot[i] * dt;

    double dpls = 0.;
    int past = 0;                  // statistics only (des_scalars::n_return_mapping): reached the yield test
    switch (RH ? RH : p->rheol_type) {
    case DES_RH_ELASTIC:
        elastic2(bulkm, shearm, de, s);
        break;
    case DES_RH_VISCOUS: {
        double visc = mat_visc2<M>(p, vt, mx, Tm, s, edot);
        if (outs) viscosity[e] = visc;
        double total_dv = trace2(es);
        viscous2(bulkm, visc, total_dv, edot, s);
        break;
    }
    case DES_RH_MAXWELL: {
        double visc = mat_visc2<M>(p, vt, mx, Tm, s, edot);
        if (outs) viscosity[e] = visc;
        double dv = vol / vol_old - 1;
        maxwell2(bulkm, shearm, visc, dt, dv, de, s);
        break;
    }
    case DES_RH_EP: {
        double depls = 0;
        double amc, anphi, anpsi, hardn, ten_max;
        desk::plastic_props<M>(p, mx, pls0, amc, anphi, anpsi, hardn, ten_max, pptab);
        if (p->is_plane_strain) {
            double syy = stressyy[e];
            elasto_plastic2d(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max, de, depls, s, syy);
            stressyy[e] = syy;
        } else {
            elasto_plastic(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max, de, depls, s);
            past = 1;
        }
        plstrain[e] = pls0 + depls;
        dpls = depls;
        break;
    }
    case DES_RH_EVP: {
        double depls = 0;
        double visc = mat_visc2<M>(p, vt, mx, Tm, s, edot);
        if (outs) viscosity[e] = visc;
        double dv = vol / vol_old - 1;
        double sv[3];
        for (int i = 0; i < 3; ++i) sv[i] = s[i];
        maxwell2(bulkm, shearm, visc, dt, dv, de, sv);
        double svII = second_invariant2_2d(sv);

        double amc, anphi, anpsi, hardn, ten_max;
        desk::plastic_props<M>(p, mx, pls0, amc, anphi, anpsi, hardn, ten_max, pptab);
        double sp[3], spyy = 0;
        for (int i = 0; i < 3; ++i) sp[i] = s[i];
        if (p->is_plane_strain) {
            spyy = stressyy[e];
            elasto_plastic2d(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max, de, depls, sp, spyy);
        } else {
            elasto_plastic(bulkm, shearm, amc, anphi, anpsi, hardn, ten_max, de, depls, sp);
            past = 1;
        }
        double spII = second_invariant2_2d(sp);
        if (svII < spII) {
            for (int i = 0; i < 3; ++i) s[i] = sv[i];
        } else {
            for (int i = 0; i < 3; ++i) s[i] = sp[i];
            plstrain[e] = pls0 + depls;
            dpls = depls;
            if (p->is_plane_strain) stressyy[e] = spyy;
        }
        break;
    }
    default: break;
    }
    if (outs) delta_plstrain[e] = dpls;
    if (count_past) {
        // (the last step of a call only.  Every wavefront leaves its count in a slot of its own -- past_part[base + wavefront of
        //  the launch] -- and fill_scalars adds them up: with the pure-2-D Mohr-Coulomb law EVERY element reaches the yield test,
        //  and twenty thousand wavefronts adding to one word took 200 us, 10 us per step of a 20-step call; round 5)
        const unsigned long long b = __ballot(past);
        if ((threadIdx.x & 63) == 0) past_part[past_base + (int)(blockIdx.x * (DES_BLOCK / 64) + (threadIdx.x >> 6))] = (int)__popcll(b);
    }
    if (p->is_using_mixed_stress) {
        const double dp = trace2(s) - old_s;
        dpressure[e] = dp;
        etmp[e] = dp * vol;                      // NMD_stress, geometry.cxx:292-296
    }
    // (stress_out_shear: where the shear component goes -- with NMD_stress behind this pass the diagonal goes to another buffer,
    //  from which the force pass writes the corrected diagonal; the shear component, which NMD_stress does not touch, goes straight
    //  to the stress array: one plane less for the force pass to copy)
    stress_out[e] = s[0]; stress_out[ne + e] = s[1]; stress_out_shear[e] = s[2];
    for (int i = 0; i < 3; ++i) strain[i * ne + e] = es[i];
}

// NMD_stress (geometry.cxx:311-331), element update
__global__ void k2_nmd_apply(int ne, const int *conn, const double *ntmp, const double *dpressure, double *stress)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    double dp = 0;
    for (int i = 0; i < 3; ++i) dp += ntmp[conn[i * ne + e]];
    double dp_el = dp / 3;
    double dp_orig = dpressure[e];
    double ddp = (-dp_orig + dp_el) / 2;
    for (int i = 0; i < 2; ++i) stress[i * ne + e] += ddp;
}

// update_force (fields.cxx:609-698): element part
__global__ void k2_force_elem(const des_params *__restrict__ p, int nn, int ne, const int *conn, const double *coord, const double *temperature,
                              const double *volume, const double *stress, const double *props, const int *markers, double *tmp_result)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    double d[3][2], shpdx[3], shpdz[3];
    elem_coords(coord, conn, nn, ne, e, d);
    double vol = volume[e];
    shape_fn2(d, vol, shpdx, shpdz);
    double s[3];
    for (int i = 0; i < 3; ++i) s[i] = stress[i * ne + e];
    double buoy = 0;
    if (p->gravity != 0) {
        desk::Mix mx = {&markers[(size_t)e * p->nmat], 0, 0};
        const double phi = props[2 * ne + e];
        buoy = (desk::mat_rho(p, mx, elemT(temperature, conn, ne, e)) * (1 - phi) + 1000.0 * phi) * p->gravity / 3;
    }
    for (int i = 0; i < 3; ++i) {
        tmp_result[i * ne + e] = (s[0]*shpdx[i] + s[2]*shpdz[i]) * vol;
        tmp_result[(i + 3) * ne + e] = (s[2]*shpdx[i] + s[1]*shpdz[i] + buoy) * vol;
    }
}

// node part
__global__ void k2_force_node(int nn, int ne, const int *sup_idx, const int *sup_arr, const int *sup_lidx,
                              const double *tmp_result, double *force, double *fres)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n >= nn) return;
    double f[2] = {0, 0}, fr[2] = {0, 0};
    for (int k = sup_idx[n]; k < sup_idx[n+1]; ++k) {
        const int e = sup_arr[k], i = sup_lidx[k];
        for (int j = 0; j < 2; j++) {
            f[j] -= tmp_result[(i + 3*j) * ne + e];
            fr[j] = tmp_result[(i + 3*j) * ne + e];            // assignment: fields.cxx:673
        }
    }
    for (int j = 0; j < 2; j++) { force[j*nn + n] = f[j]; fres[j*nn + n] = fr[j]; }
}

// apply_stress_bcs (bc.cxx:661-827) of boundary `ib`: facet part ...
// the pressure on a boundary facet and its (unnormalised) outward normal
__device__ __forceinline__ double sbc_facet_pressure(const des_params *__restrict__ p, int ib, int e, int f, int nn, int ne, const int *conn,
                                                     const double *coord, const double *temperature, const int *markers, double normal[2])
{
    double zcenter, fc[2][2];
    for (int j = 0; j < 2; ++j) {
        const int nd = conn[NODE_OF_FACET_D[f][j] * ne + e];
        fc[j][0] = coord[nd]; fc[j][1] = coord[nn + nd];
    }
    normal_vector_of_facet(fc, normal, zcenter);
    double pr;
    if (ib == iboundz0 && p->has_winkler_foundation) {
        desk::Mix mx = {&markers[(size_t)e * p->nmat], 0, 0};
        double rho_effective = desk::mat_rho(p, mx, elemT(temperature, conn, ne, e));
        pr = p->compensation_pressure - (rho_effective + p->winkler_delta_rho) * p->gravity * (zcenter + p->zlength);
    } else if (ib == iboundz1 && p->has_water_loading) {
        pr = 0;
        if (zcenter < p->surf_base_level)
            pr = p->sea_water_density * p->gravity * (p->surf_base_level - zcenter);
    } else {
        pr = desk::ref_pressure(p, zcenter);
        if (pr < 0.0) pr = 0.0;
    }
    return pr;
}

__global__ void k2_sbc_facet(const des_params *__restrict__ p, int ib, int bound, const int *bf_elem, const int *bf_facet, int nn, int ne,
                             const int *conn, const double *coord, const double *temperature, const int *markers,
                             double *tmp_result, int *etmp_int)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n >= bound) return;
    const int e = bf_elem[n], f = bf_facet[n];
    double normal[2];
    const double pr = sbc_facet_pressure(p, ib, e, f, nn, ne, conn, coord, temperature, markers, normal);
    etmp_int[e] = n;
    for (int j = 0; j < 2; ++j)
        for (int d = 0; d < 2; ++d)
            tmp_result[(j*2 + d) * ne + n] = pr * normal[d] / 2;
}

// The same in ONE launch (the patch path): every boundary node forms the terms of its boundary facets itself -- the same
// expressions, subtracted in the order of its support list as k2_sbc_node does -- from a list built once per mesh:
// binc[(2*j + q)] = {element, facet, which of the facet's two nodes} of the q-th such incidence of boundary node j, element -1: none.
#define DES2_SBC_INC 2
__global__ void k2_sbc_direct(const des_params *__restrict__ p, int ib, int nbdry_nodes, const int *bnodes, const int4 *binc,
                              int nn, int ne, const int *conn, const double *coord, const double *temperature, const int *markers,
                              double *force)
{
    const int j = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (j >= nbdry_nodes) return;
    const int n = bnodes[j];
    int4 inc[DES2_SBC_INC];
    for (int q = 0; q < DES2_SBC_INC; ++q) inc[q] = binc[DES2_SBC_INC * j + q];
    double f0 = force[n], f1 = force[nn + n];
    for (int q = 0; q < DES2_SBC_INC; ++q) {
        if (inc[q].x < 0) break;
        double normal[2];
        const double pr = sbc_facet_pressure(p, ib, inc[q].x, inc[q].y, nn, ne, conn, coord, temperature, markers, normal);
        f0 -= pr * normal[0] / 2;
        f1 -= pr * normal[1] / 2;
    }
    force[n] = f0; force[nn + n] = f1;
}

// ... node part ...
__global__ void k2_sbc_node(int nbdry_nodes, const int *bnodes, const int *bf_facet, int nn, int ne, const int *conn,
                            const int *sup_idx, const int *sup_arr, const double *tmp_result, const int *etmp_int, double *force)
{
    const int j = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (j >= nbdry_nodes) return;
    const int n = bnodes[j];
    for (int k = sup_idx[n]; k < sup_idx[n+1]; ++k) {
        int e = sup_arr[k];
        int ibound = etmp_int[e];
        if (ibound < 0) continue;
        int f = bf_facet[ibound];
        for (int l = 0; l < 2; ++l) {
            if (n == conn[NODE_OF_FACET_D[f][l] * ne + e]) {
                for (int d = 0; d < 2; ++d)
                    force[d * nn + n] -= tmp_result[(l*2 + d) * ne + ibound];
                break;
            }
        }
    }
}

// ... and the reset of the facet marks
__global__ void k2_sbc_reset(int bound, const int *bf_elem, int *etmp_int)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n < bound) etmp_int[bf_elem[n]] = -1;
}

__global__ void k2_fill_int(int n, int *a, int v)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < n) a[i] = v;
}

// elastic foundation (bc.cxx:819-825)
__global__ void k2_elastic_foundation(const des_params *__restrict__ p, int nb, const int *bnodes, int nn, const double *coord,
                                      const double *coord0, double *force)
{
    const int j = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (j >= nb) return;
    const int n = bnodes[j];
    force[nn + n] -= p->elastic_foundation_constant * (coord[nn + n] - coord0[nn + n]);
}

// apply_stress_bcs_neumann (bc.cxx:829-912) of one boundary: a serial loop in the reference, and
// facets of one boundary share nodes, so one lane walks them in the reference's order
__global__ void k2_neumann(const des_params *__restrict__ p, int ib, int bound, const int *bf_elem, const int *bf_facet, int nn, int ne,
                           const int *conn, const double *coord, double *force)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int n = 0; n < bound; ++n) {
        const int e = bf_elem[n], f = bf_facet[n];
        double normal[2] = {0, 0}, zcenter = 0, fc[2][2];
        for (int j = 0; j < 2; ++j) {
            const int nd = conn[NODE_OF_FACET_D[f][j] * ne + e];
            fc[j][0] = coord[nd]; fc[j][1] = coord[nn + nd];
        }
        normal_vector_of_facet(fc, normal, zcenter);
        double traction[2] = {0, 0};
        switch (p->stress_bc_types[ib]) {
        case 1: traction[0] = p->stress_bc_values[ib]; break;
        case 3: traction[1] = p->stress_bc_values[ib]; break;
        default: continue;
        }
        for (int j = 0; j < 2; ++j) {
            const int node = conn[NODE_OF_FACET_D[f][j] * ne + e];
            for (int d = 0; d < 2; ++d)
                force[d * nn + node] += traction[d] * normal[d] / 2;
        }
    }
}

// apply_damping (fields.cxx:483-579) + update_velocity (fields.cxx:725-742) of a node, on values the caller holds:
// f[] in: the force sums, out: the damped force; v[] in / out: the velocity
// (dopt, dfac = Param::control.damping_option / damping_factor: read by the caller, which may do so long before the values
//  are used -- a scalar load behind a barrier is a trip to memory the compiler may not hoist)
__device__ __forceinline__ void damp_vel_regs(const int dopt, const double dfac, double dt, double mass_i, double ymass_i, double f_io[2], double v_io[2])
{
    const double small_vel = 1e-13;
    for (int j = 0; j < 2; j++) {
        double f = f_io[j];
        const double v = v_io[j];
        switch (dopt) {
        case 1:
            if (fabs(v) > small_vel) f -= dfac * copysign(f, v);
            break;
        case 2:
            f -= dfac * f;
            break;
        case 3:
            if ((f < 0) == (v < 0)) f -= dfac * f;          // fields.cxx:538: comma operator
            else f += (1 - dfac) * f;
            break;
        case 4:
            if (fabs(v) > small_vel) {
                double critical_coeff = 2.0 * sqrt(mass_i * ymass_i);
                double f_C = dfac * copysign(f, v);
                double f_V = critical_coeff * v;
                double f_damping = (fabs(f_C) < fabs(f_V)) ? f_V : f_C;
                f -= f_damping;
            }
            break;
        default: break;
        }
        f_io[j] = f;
        v_io[j] = v + dt * f / mass_i;
    }
}

// ... and on the node's entries of the global arrays (ymass is only read by damping option 4)
__device__ __forceinline__ void damp_vel_node(const des_params *__restrict__ p, const Clock *clk, int i, int nn, const double *mass,
                                              const double *ymass, double *force, double *vel)
{
    double f[2] = {force[i], force[nn + i]}, v[2] = {vel[i], vel[nn + i]};
    damp_vel_regs(p->damping_option, p->damping_factor, clk->dt, mass[i], p->damping_option == 4 ? ymass[i] : 0.0, f, v);
    for (int j = 0; j < 2; j++) { force[j*nn + i] = f[j]; vel[j*nn + i] = v[j]; }
}

__global__ void k2_damp_vel(const des_params *__restrict__ p, const Clock *clk, int nn, const double *mass, const double *ymass,
                            double *force, double *vel)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < nn) damp_vel_node(p, clk, i, nn, mass, ymass, force, vel);
}

// calculate_residual_force (fields.cxx:700-722): per-block partial sums, then one block adds them
// (a decomposed mesh: over the rank's owned nodes [o0, o1), divisor = the global node count)
__global__ void k2_residual_part(int nn, int o0, int o1, int nn_global, const double *fres, double *part)
{
    __shared__ double sm[DES_BLOCK / 64];
    const int i = o0 + blockIdx.x * DES_BLOCK + threadIdx.x;
    double v = 0;
    if (i < o1) {
        const double num = (double)nn_global * 2;
        for (int j = 0; j < 2; ++j) { const double f = fres[j*nn + i]; v += f * f / num; }
    }
    v = desk::wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__device__ __forceinline__ void residual_fin_block(int nb, const double *part, Clock *clk)
{
    __shared__ double sm[DES_BLOCK / 64];
    double v = 0;
    for (int i = threadIdx.x; i < nb; i += DES_BLOCK) v += part[i];
    v = desk::wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    // (l2_sum: the sum under the root -- on a decomposed mesh the owned nodes' share, added across ranks at the end of a call)
    if (threadIdx.x == 0) { const double t = (sm[0] + sm[1]) + (sm[2] + sm[3]); clk->l2_sum = t; clk->l2_residual = sqrt(t); clk->l2_global = 0; }
}

__global__ void k2_residual_fin(int nb, const double *part, Clock *clk) { residual_fin_block(nb, part, clk); }

// The residual where a decision hangs on it (the pseudo-transient loop): ONE association whatever the partition -- per
// block of B consecutive GLOBAL node ids the nodes' terms one after the other (one wavefront per block, lane 0's running sum
// walks the lanes' terms in order), the blocks in a fixed shape over the global array (des_params.h: DES_RES_BLOCK).
__global__ void __launch_bounds__(DES_BLOCK)
k2_residual_blocks(int nown, int B, int nblocks, int nn, int o0, int nn_global, const double *fres, double *out)
{
    const int b = (int)(blockIdx.x * (DES_BLOCK / 64) + (threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    if (b >= nblocks) return;
    const double num = (double)nn_global * 2;
    const int i = b * B + lane;
    double t = 0.0;
    if (lane < B && i < nown) {
        const double f0 = fres[o0 + i], f1 = fres[(size_t)nn + o0 + i];
        t = f0 * f0 / num;
        t += f1 * f1 / num;
    }
    double s = 0.0;
    for (int k = 0; k < B; ++k) s += __shfl(t, k);
    if (lane == 0) out[b] = s;
}

__global__ void __launch_bounds__(DES_BLOCK)
k2_residual_final(Clock *clk, const double *blocks, int nb)
{
    __shared__ double red[DES_BLOCK];
    double t = 0;
    for (int i = threadIdx.x; i < nb; i += DES_BLOCK) t += blocks[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int off = DES_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    // (the sum over EVERY rank's blocks: the end of a call must not add it across ranks again -- l2_global says so)
    if (threadIdx.x == 0) { clk->l2_sum = red[0]; clk->l2_residual = sqrt(red[0]); clk->l2_global = 1; }
}

// apply_vbcs, 2-D: vertical extent of the x0 wall (bc.cxx:251-290; only x0's is used, :292-300) over the list of
// its nodes.  min / max: exact whatever the order.  One workgroup.
// (tick: the step counter and the model time move on here, k2_clock's two statements -- the patch path's plain step, where
//  nothing before this launch reads them)
__device__ __forceinline__ void vbc_extent_block(int nb, const int *bnodes_x0, int nn, const double *coord, Clock *clk, int tick)
{
    if (tick == 2) {                     // a rank of a decomposed mesh: the extent in the clock is the cross-rank one (k2_wall_set); count the step
        if (threadIdx.x == 0) { clk->steps++; clk->time += clk->dt; }
        return;
    }
    __shared__ double s_max[DES_BLOCK / 64], s_min[DES_BLOCK / 64];
    double mx = -DBL_MAX, mn = DBL_MAX;
    for (int j = threadIdx.x; j < nb; j += DES_BLOCK) {
        const double z = coord[nn + bnodes_x0[j]];
        mx = fmax(mx, z); mn = fmin(mn, z);
    }
    mx = desk::wave_max(mx); mn = desk::wave_min(mn);
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; s_max[w] = mx; s_min[w] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < DES_BLOCK / 64; ++w) { s_max[0] = fmax(s_max[0], s_max[w]); s_min[0] = fmin(s_min[0], s_min[w]); }
        clk->x0_init = nb > 0;
        clk->x0_max = nb > 0 ? s_max[0] : 0.;
        clk->x0_min = nb > 0 ? s_min[0] : 0.;
        clk->zmin = 0;                       // k2_vbc_zmin lowers it when the sheared bottom zone needs it
        if (tick) { clk->steps++; clk->time += clk->dt; }
    }
}

__global__ void k2_vbc_extent(int nb, const int *bnodes_x0, int nn, const double *coord, Clock *clk, int tick = 0)
{
    vbc_extent_block(nb, bnodes_x0, nn, coord, clk, tick);
}

// NMD_stress' nodal average with the wall's extent as one extra workgroup (round 5: the plain step of the patch path, where the
// extent -- and with it the step's count -- only has to be in the clock before the force pass's node phase)
// The gather itself as an LDS-staged segmented sum: a workgroup owns 256 consecutive nodes, i.e. one contiguous slice of the
// support list; all lanes first fetch the slice's element values (coalesced index reads, one gather each) into LDS, then
// every node's lane adds its own segment in list order -- the thread-per-node walk's order, the same bits, without a chain
// of three dependent trips to memory per incidence.
#define DES2_AVG_TILE 2048
__global__ void __launch_bounds__(DES_BLOCK)
k2_node_avg_extent(int nn, int nb_avg, const int *sup_idx, const int *sup_arr, const double *etmp, const double *volume_n, double *ntmp,
                   int nb, const int *bnodes_x0, const double *coord, Clock *clk, int tick)
{
    if ((int)blockIdx.x >= nb_avg) { vbc_extent_block(nb, bnodes_x0, nn, coord, clk, tick); return; }
    __shared__ double lv[DES2_AVG_TILE];
    const int n0 = blockIdx.x * DES_BLOCK, n1 = min(nn, n0 + DES_BLOCK);
    const int n = n0 + threadIdx.x;
    const int kb = sup_idx[n0], ke = sup_idx[n1];
    int r0 = ke, r1 = ke;
    double vn = 1.0;
    if (n < nn) { r0 = sup_idx[n]; r1 = sup_idx[n + 1]; vn = volume_n[n]; }
    double acc = 0.;
    for (int t0 = kb; t0 < ke; t0 += DES2_AVG_TILE) {
        const int tn = min(DES2_AVG_TILE, ke - t0);
        for (int j = threadIdx.x; j < tn; j += DES_BLOCK) lv[j] = etmp[sup_arr[t0 + j]];
        __syncthreads();
        const int a = max(r0, t0) - t0, b = min(r1, t0 + tn) - t0;
        for (int j = a; j < b; ++j) acc += lv[j];
        __syncthreads();
    }
    if (n < nn) ntmp[n] = acc / vn;
}

// ... and the lowest node of the mesh, zmin = min(0, min z) (bc.cxx:350-361): read only by vbc_x0 = 3 with a
// bottom shear zone (bc.cxx:446-451), so only launched then.  -zmin = max(0, max(-z)): a maximum of
// non-negative doubles, one native 64-bit integer atomic per workgroup.
__global__ void k2_vbc_zmin(int nn, const double *coord, double *neg_zmin)
{
    __shared__ double sm[DES_BLOCK / 64];
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    double m = (i < nn) ? fmax(0.0, -coord[nn + i]) : 0.0;
    m = desk::wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < DES_BLOCK / 64; ++w) sm[0] = fmax(sm[0], sm[w]);
        desk::atomic_max_double(neg_zmin, sm[0]);
    }
}
__global__ void k2_vbc_zmin_fin(Clock *clk, double *neg_zmin) { clk->zmin = -(*neg_zmin); *neg_zmin = 0.0; }

// apply_vbcs (bc.cxx:227-659, !THREED) of a node; Clock::pt = PT_jump: boundaries at rest (bc.cxx:330-343)
// (register form: flag = bcflag of the node, x1 = its z BEFORE it moves, v[] = its velocity in / out)
__device__ __forceinline__ void vbcs_regs(const des_params *__restrict__ p, const Clock *clk, const unsigned flag, const double x1,
                                          const double *bnormals, const double *edge_vec, const int *edge_slot, double v[2])
{
    if (!(flag & BOUND_ANY)) return;

    double t_now = clk->time / DES2_YEAR2SEC;
    double vbc_applied_x0 = p->vbc_values[0] * interp1(p->vbc_period_x0_time_in_yr, p->vbc_period_x0_ratio, p->num_vbc_period_x0, t_now);
    double vbc_applied_x1 = p->vbc_values[1] * interp1(p->vbc_period_x1_time_in_yr, p->vbc_period_x1_ratio, p->num_vbc_period_x1, t_now);
    const double BOUNDX0_max = clk->x0_max, BOUNDX0_width = clk->x0_max - clk->x0_min;
    double div_x0[4], div_x1[4];
    for (int k = 0; k < 4; k++) {
        div_x0[k] = - (BOUNDX0_max - p->vbc_vertical_div_x0[k] * BOUNDX0_width);
        div_x1[k] = - (BOUNDX0_max - p->vbc_vertical_div_x1[k] * BOUNDX0_width);      // x0's extent: bc.cxx:299
    }
    const int bc_x0 = p->vbc_types[0], bc_x1 = p->vbc_types[1];
    const int bc_z0 = p->vbc_types[4];
    int bc_z1 = p->vbc_types[5];
    double bc_vx0 = p->vbc_values[0], bc_vx1 = p->vbc_values[1];
    double bc_vz0 = p->vbc_values[4], bc_vz1 = p->vbc_values[5];
    const double bc_vx0_l = p->vbc_val_l[0], bc_vx1_l = p->vbc_val_l[1];
    if (clk->pt) {
        bc_vx0 = 0.0; bc_vx1 = 0.0; bc_vz0 = 0.0; bc_vz1 = 0.0;
        vbc_applied_x0 = 0.0; vbc_applied_x1 = 0.0;
    }
    if (clk->time > p->vbc_val_z1_loading_period) bc_z1 = 0;
    const double zmin = clk->zmin;

    double vbc_exact_x0 = vbc_applied_x0 * interp1(div_x0, p->vbc_vertical_ratio_x0, 4, -x1);
    double vbc_exact_x1 = vbc_applied_x1 * interp1(div_x1, p->vbc_vertical_ratio_x1, 4, -x1);

    if (flag & BOUNDX0) {
        switch (bc_x0) {
        case 0: break;
        case 1: v[0] = vbc_exact_x0; break;
        case 2: v[1] = 0; break;
        case 3:
            v[0] = vbc_exact_x0;
            if (p->bottom_shear_zone_thickness > 0.) {
                double dz = x1 - zmin;
                if (dz < p->bottom_shear_zone_thickness)
                    v[0] = v[0] * dz / p->bottom_shear_zone_thickness;
            }
            v[1] = 0;
            break;
        case 4: v[0] = 0; v[1] = bc_vx0; break;
        case 6: v[0] = vbc_exact_x0; v[1] = bc_vx0_l; break;
        }
    }
    if (flag & BOUNDX1) {
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
        const double n[2] = {bnormals[ib], bnormals[DES_NBDRY + ib]};
        double fac = 0;
        switch (p->vbc_types[ib]) {
        case 1:
        case 11: {
            const int nd = (p->vbc_types[ib] == 1) ? 2 : 1;
            double target = p->vbc_values[ib];
            if (p->vbc_types[ib] == 11) {
                fac = 1 / sqrt(1 - n[1]*n[1]);
                target = p->vbc_values[ib] * fac;
            }
            if (flag == (1u << ib)) {
                double vn = 0;
                for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
            } else {
                for (int ic = iboundx0; ic < ib; ic++) {
                    if (!(flag & (1u << ic))) continue;
                    if (p->vbc_types[ic] == 0) {
                        double vn = 0;
                        for (int d = 0; d < nd; d++) vn += v[d] * n[d];
                        for (int d = 0; d < nd; d++) v[d] += (target - vn) * n[d];
                    } else if (p->vbc_types[ic] == 1) {
                        const int slot = edge_slot[ic*DES_NBDRY + ib];
                        if (slot < 0) continue;
                        const double *edge = &edge_vec[slot*2];
                        double ve = 0;
                        for (int d = 0; d < 2; d++) ve += v[d] * edge[d];
                        for (int d = 0; d < 2; d++) v[d] = ve * edge[d];
                    }
                }
            }
            break;
        }
        case 3:
            for (int d = 0; d < 2; d++) v[d] = p->vbc_values[ib] * n[d];
            break;
        case 13:
            fac = 1 / sqrt(1 - n[1]*n[1]);
            v[0] = p->vbc_values[ib] * fac * n[0];
            v[1] = 0;
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
}

__device__ __forceinline__ void vbcs_node(const des_params *__restrict__ p, const Clock *clk, int i, int nn, const unsigned *bcflag,
                                          const double *bnormals, const double *edge_vec, const int *edge_slot,
                                          const double *coord, double *vel)
{
    const unsigned flag = bcflag[i];
    if (!(flag & BOUND_ANY)) return;
    double v[2] = {vel[i], vel[nn + i]};
    vbcs_regs(p, clk, flag, coord[nn + i], bnormals, edge_vec, edge_slot, v);
    vel[i] = v[0]; vel[nn + i] = v[1];
}

__global__ void k2_apply_vbcs(const des_params *__restrict__ p, const Clock *clk, int nn, const unsigned *bcflag, const double *bnormals,
                              const double *edge_vec, const int *edge_slot, const double *coord, double *vel)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < nn) vbcs_node(p, clk, i, nn, bcflag, bnormals, edge_vec, edge_slot, coord, vel);
}

// apply_damping + update_velocity, apply_vbcs and update_coordinate of a node in one launch: each touches the node's own
// entries only (apply_vbcs reads the wall extent the clock already holds and the node's own z before it moves)
__global__ void k2_node_final(const des_params *__restrict__ p, const Clock *clk, int nn, const double *mass, const double *ymass,
                              const unsigned *bcflag, const double *bnormals, const double *edge_vec, const int *edge_slot,
                              double *force, double *vel, double *coord)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i >= nn) return;
    damp_vel_node(p, clk, i, nn, mass, ymass, force, vel);
    vbcs_node(p, clk, i, nn, bcflag, bnormals, edge_vec, edge_slot, coord, vel);
    coord[i] += vel[i] * clk->dt;
    coord[nn + i] += vel[nn + i] * clk->dt;
}

// isostasy_adjustment's velocity filter (dynearthsol.cxx:524-533)
__global__ void k2_iso_vel(const des_params *__restrict__ p, int nn, const unsigned *bcflag, double *vel)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i >= nn) return;
    vel[i] = 0;
    if (!p->has_winkler_foundation && (bcflag[i] & BOUNDZ0)) vel[nn + i] = 0;
}

// update_coordinate (fields.cxx:761-784)
__global__ void k2_update_coord(const Clock *clk, int n2, const double *vel, double *coord)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < n2) coord[i] += vel[i] * clk->dt;
}

// simple_diffusion (bc.cxx:916-1112, !THREED): segments of the sorted top nodes ...
__device__ __forceinline__ void surf_seg_at(int i, int etop, int nn, int ne, const int *top_nodes, const double *coord, double *etmp, double *tmp_result)
{
    const int n0 = top_nodes[i], n1 = top_nodes[i+1];
    double dx = fabs(coord[n1] - coord[n0]);
    etmp[i] = dx;
    tmp_result[0 * ne + i] = -(coord[nn + n1] - coord[nn + n0]) / dx;
    tmp_result[1 * ne + i] = (coord[nn + n1] - coord[nn + n0]) / dx;
}

__global__ void k2_surf_seg(int etop, int nn, int ne, const int *top_nodes, const double *coord, double *etmp, double *tmp_result)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < etop) surf_seg_at(i, etop, nn, ne, top_nodes, coord, etmp, tmp_result);
}

// ... then the height change of every top node; surface_processes moves the node and books dhacc
// (bc.cxx:1773-1786).  dh[] starts from 0 (bc.cxx:1718-1724).
__device__ __forceinline__ void surf_node_at(int i, const des_params *__restrict__ p, const Clock *clk, int ntop, int nn, int ne, const int *top_nodes,
                             const double *etmp, const double *tmp_result, double *total_dx, double *total_slope,
                             double *coord, double *dhacc, double *dh)
{
    const int n = top_nodes[i];
    double d = 0.;
    if (p->surface_process_option == 1) {
        double tdx, tsl;
        if (i == 0) { tdx = etmp[i]; tsl = tmp_result[0 * ne + i]; }
        else if (i == ntop-1) { tdx = etmp[i-1]; tsl = tmp_result[1 * ne + i-1]; }
        else { tdx = etmp[i-1] + etmp[i]; tsl = tmp_result[1 * ne + i-1] + tmp_result[0 * ne + i]; }
        total_dx[n] = tdx; total_slope[n] = tsl;
        double conv = p->surface_diffusivity * clk->dt * tsl / tdx;
        const double z = coord[nn + n];
        if (z > p->surf_base_level && conv > 0.) d -= p->surf_diff_ratio_terrig * conv;
        else if (z <= p->surf_base_level && conv < 0.) d -= p->surf_diff_ratio_marine * conv;
        else d -= conv;
    }
    dh[i] = d;
    coord[nn + n] += d;
    dhacc[n] += d;
}

__global__ void k2_surf_node(const des_params *__restrict__ p, const Clock *clk, int ntop, int nn, int ne, const int *top_nodes,
                             const double *etmp, const double *tmp_result, double *total_dx, double *total_slope,
                             double *coord, double *dhacc, double *dh)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < ntop) surf_node_at(i, p, clk, ntop, nn, ne, top_nodes, etmp, tmp_result, total_dx, total_slope, coord, dhacc, dh);
}

// a top node's step of simple_diffusion (bc.cxx:1709-1787) from the moved {x, z} of the top nodes in top_nodes order
// (a, c, b: the entries of node i - 1, i, i + 1; a / b unused at the ends of the line)
__device__ __forceinline__ double surf_commit_dh_regs(const des_params *__restrict__ p, double dt, int ntop, int i, const double2 a, const double2 c, const double2 b,
                                                      double &tdx, double &tsl)
{
    // the segment to the left (i - 1 .. i) and to the right (i .. i + 1): dx, and the two slope terms surf_seg_at stores
    double dxl = 0, sl1 = 0, dxr = 0, sr0 = 0;
    if (i > 0) { dxl = fabs(c.x - a.x); sl1 = (c.y - a.y) / dxl; }
    if (i < ntop - 1) { dxr = fabs(b.x - c.x); sr0 = -(b.y - c.y) / dxr; }
    if (i == 0) { tdx = dxr; tsl = sr0; }
    else if (i == ntop-1) { tdx = dxl; tsl = sl1; }
    else { tdx = dxl + dxr; tsl = sl1 + sr0; }
    double d = 0.;
    double conv = p->surface_diffusivity * dt * tsl / tdx;
    const double z = c.y;
    if (z > p->surf_base_level && conv > 0.) d -= p->surf_diff_ratio_terrig * conv;
    else if (z <= p->surf_base_level && conv < 0.) d -= p->surf_diff_ratio_marine * conv;
    else d -= conv;
    return d;
}

__device__ __forceinline__ double surf_commit_dh(const des_params *__restrict__ p, double dt, int ntop, int i, const double2 *xz_pre, double &tdx, double &tsl)
{
    const double2 c = xz_pre[i];
    const double2 a = i > 0 ? xz_pre[i - 1] : c, b = i < ntop - 1 ? xz_pre[i + 1] : c;
    return surf_commit_dh_regs(p, dt, ntop, i, a, c, b, tdx, tsl);
}

// Round 5: segments + nodes of simple_diffusion in ONE launch (the plain step of the patch path with the nodal tail folded
// into the force pass).  A node needs the two segments it sits between; each is a function of its two end nodes alone, so the
// node's lane forms both itself -- surf_seg_at's expressions -- instead of a launch of its own writing them first.  What
// stood in the way was the update in place: a lane moves its node's z while its neighbours' lanes still want the old one.
// k2p_force<1> therefore leaves the moved {x, z} of every top node in a compact array in top_nodes order as well (xz_pre:
// the coordinates update_coordinate left, before the surface step), which is all this kernel reads of other nodes.
// The last workgroup forms calculate_residual_force's final sum (as in k2_surf_seg_resfin).
__global__ void k2_surf_commit(const des_params *__restrict__ p, Clock *clk, int ntop, int nn, const int *top_nodes, const double2 *xz_pre,
                               double *total_dx, double *total_slope, double *coord, double *dhacc, double *dh,
                               int nb_node, int res_nb, const double *res_part)
{
    if ((int)blockIdx.x >= nb_node) { residual_fin_block(res_nb, res_part, clk); return; }
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i >= ntop) return;
    const int n = top_nodes[i];
    double tdx, tsl;
    const double d = surf_commit_dh(p, clk->dt, ntop, i, xz_pre, tdx, tsl);
    const double z = xz_pre[i].y;
    total_dx[n] = tdx; total_slope[n] = tsl;
    dh[i] = d;
    coord[nn + n] = z + d;
    dhacc[n] += d;
}

// edvacc_surf (bc.cxx:1788-1805)
__device__ __forceinline__ void surf_edv_at(int i, int etop, int nn, const int *ean, const int *conn_surf, const double *coord, const double *dh, double *edvacc)
{
    double dh_e = 0.;
    for (int j = 0; j < 2; j++) dh_e += dh[ean[j * etop + i]];
    const double base = fabs(coord[conn_surf[i]] - coord[conn_surf[etop + i]]);      // compute_area_facet, geometry.cxx:109-121
    edvacc[i] += dh_e * base / 2;
}

__global__ void k2_surf_edv(int etop, int nn, const int *ean, const int *conn_surf, const double *coord, const double *dh, double *edvacc)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < etop) surf_edv_at(i, etop, nn, ean, conn_surf, coord, dh, edvacc);
}

// max |dh| -> max_surf_vel (bc.cxx:1820-1836).  One workgroup.
__device__ __forceinline__ void surf_maxdh_block(int ntop, const int *top_nodes, int o0, int o1, const double *dh, Clock *clk)
{
    __shared__ double sm[DES_BLOCK / 64];
    double m = 0.;
    for (int i = threadIdx.x; i < ntop; i += DES_BLOCK) {
        const int n = top_nodes[i];
        if (n >= o0 && n < o1) m = fmax(m, fabs(dh[i]));
    }
    m = desk::wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < DES_BLOCK / 64; ++w) sm[0] = fmax(sm[0], sm[w]);
        clk->maxdh = sm[0];
        clk->max_surf_vel = sm[0] / clk->dt;
    }
}

__global__ void k2_surf_maxdh(int ntop, const int *top_nodes, int o0, int o1, const double *dh, Clock *clk)
{
    surf_maxdh_block(ntop, top_nodes, o0, o1, dh, clk);
}

// correct_surface_element (bc.cxx:1655-1707), element part; surface_plstrain_diffusion (bc.cxx:1633-1653)
// rides along when `decay` (same elements, applied after the correction as in surface_processes)
__device__ __forceinline__ void cse_elem_at(int i, const des_params *__restrict__ p, const Clock *clk, int ntop_elems, const int *top_elems, int nn, int ne,
                            const int *conn, const double *coord, const int *markers, int decay, double *volume,
                            double *plstrain, double *stress, double *strain, double *strain_rate)
{
    const int e = top_elems[i];
    double d[3][2];
    elem_coords(coord, conn, nn, ne, e, d);
    double new_volumes = triangle_area(d[0], d[1], d[2]);
    double rdv = new_volumes / volume[e];
    volume[e] = new_volumes;
    double pls = plstrain[e];
    if (!(rdv < 1.0)) {
        pls /= rdv;
        for (int j = 0; j < 3; j++) {
            stress[j*ne+e] /= rdv;
            strain[j*ne+e] /= rdv;
            strain_rate[j*ne+e] /= rdv;
        }
    }
    if (decay) {
        const int *a = &markers[(size_t)e * p->nmat];
        int mat = 0;
        for (int m = 1; m < p->nmat; ++m) if (a[m] > a[mat]) mat = m;        // std::max_element: first maximum
        if (mat != p->mattype_oceanic_crust) {
            double half_life = 1.e2 * DES2_YEAR2SEC;
            double lambha = 0.69314718056 / half_life;
            pls -= pls * lambha * clk->dt;
        }
    }
    plstrain[e] = pls;
}

__global__ void k2_cse_elem(const des_params *__restrict__ p, const Clock *clk, int ntop_elems, const int *top_elems, int nn, int ne,
                            const int *conn, const double *coord, const int *markers, int decay, double *volume,
                            double *plstrain, double *stress, double *strain, double *strain_rate)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < ntop_elems) cse_elem_at(i, p, clk, ntop_elems, top_elems, nn, ne, conn, coord, markers, decay, volume, plstrain, stress, strain, strain_rate);
}

__device__ __forceinline__ void cse_node_at(int i, int ntop, const int *top_nodes, const int *sup_idx, const int *sup_arr, const double *volume,
                            double *volume_n, int reset_dhacc, double *dhacc)
{
    const int nt = top_nodes[i];
    double acc = 0.;
    for (int k = sup_idx[nt]; k < sup_idx[nt+1]; ++k) acc += volume[sup_arr[k]];
    volume_n[nt] = acc;
    if (reset_dhacc) dhacc[nt] = 0.;                      // bc.cxx:1837-1838
}

__global__ void k2_cse_node(int ntop, const int *top_nodes, const int *sup_idx, const int *sup_arr, const double *volume,
                            double *volume_n, int reset_dhacc, double *dhacc)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < ntop) cse_node_at(i, ntop, top_nodes, sup_idx, sup_arr, volume, volume_n, reset_dhacc, dhacc);
}

// The independent loops of surface_processes share launches on the patch path: the segments of simple_diffusion with the
// final sum of calculate_residual_force (one extra workgroup); edvacc_surf with correct_surface_element's element part;
// the nodal part with the max |dh| reduction (one extra workgroup).
__global__ void k2_surf_seg_resfin(int etop, int nn, int ne, const int *top_nodes, const double *coord, double *etmp, double *tmp_result,
                                   int nb_seg, int res_nb, const double *res_part, Clock *clk)
{
    if ((int)blockIdx.x >= nb_seg) { residual_fin_block(res_nb, res_part, clk); return; }
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < etop) surf_seg_at(i, etop, nn, ne, top_nodes, coord, etmp, tmp_result);
}

__global__ void k2_surf_edv_cse_elem(const des_params *__restrict__ p, const Clock *clk, int etop, int ntop_elems, int nb_edv, int nn, int ne,
                                     const int *ean, const int *conn_surf, const int *top_elems, const int *conn, const int *markers,
                                     int decay, const double *coord, const double *dh, double *edvacc, double *volume, double *plstrain,
                                     double *stress, double *strain, double *strain_rate)
{
    if ((int)blockIdx.x < nb_edv) {
        const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
        if (i < etop) surf_edv_at(i, etop, nn, ean, conn_surf, coord, dh, edvacc);
        return;
    }
    const int i = ((int)blockIdx.x - nb_edv) * DES_BLOCK + threadIdx.x;
    if (i < ntop_elems) cse_elem_at(i, p, clk, ntop_elems, top_elems, nn, ne, conn, coord, markers, decay, volume, plstrain, stress, strain, strain_rate);
}

// correct_surface_element's nodal part WITHOUT waiting for its element part: every element of a top node's support list is a
// top element (create_top_elems), whose volume the element part has just set to triangle_area of its moved nodes -- the same
// expression on the same coordinates gives the same bits, so the node's lane forms the areas itself
__device__ __forceinline__ void cse_node_areas_at(int i, int nn, int ne, const int *top_nodes, const int *sup_idx, const int *sup_arr,
                                                  const int *conn, const double *coord, double *volume_n, int reset_dhacc, double *dhacc)
{
    const int nt = top_nodes[i];
    double acc = 0.;
    for (int k = sup_idx[nt]; k < sup_idx[nt+1]; ++k) {
        double d[3][2];
        elem_coords(coord, conn, nn, ne, sup_arr[k], d);
        acc += triangle_area(d[0], d[1], d[2]);
    }
    volume_n[nt] = acc;
    if (reset_dhacc) dhacc[nt] = 0.;                      // bc.cxx:1837-1838
}

// ... and with that edvacc_surf, both parts of correct_surface_element and the max |dh| reduction are ONE launch (round 5)
__global__ void k2_surf_edv_cse_all(const des_params *__restrict__ p, Clock *clk, int etop, int ntop_elems, int ntop, int nb_edv, int nb_elem, int nb_node,
                                    int nn, int ne, const int *ean, const int *conn_surf, const int *top_elems, const int *top_nodes,
                                    const int *conn, const int *markers, const int *sup_idx, const int *sup_arr, int decay, int reset_dhacc,
                                    int o0, int o1, const double *coord, const double *dh, double *edvacc, double *volume, double *volume_n,
                                    double *dhacc, double *plstrain, double *stress, double *strain, double *strain_rate)
{
    int b = (int)blockIdx.x;
    if (b < nb_edv) {
        const int i = b * DES_BLOCK + threadIdx.x;
        if (i < etop) surf_edv_at(i, etop, nn, ean, conn_surf, coord, dh, edvacc);
        return;
    }
    b -= nb_edv;
    if (b < nb_elem) {
        const int i = b * DES_BLOCK + threadIdx.x;
        if (i < ntop_elems) cse_elem_at(i, p, clk, ntop_elems, top_elems, nn, ne, conn, coord, markers, decay, volume, plstrain, stress, strain, strain_rate);
        return;
    }
    b -= nb_elem;
    if (b < nb_node) {
        const int i = b * DES_BLOCK + threadIdx.x;
        if (i < ntop) cse_node_areas_at(i, nn, ne, top_nodes, sup_idx, sup_arr, conn, coord, volume_n, reset_dhacc, dhacc);
        return;
    }
    surf_maxdh_block(ntop, top_nodes, o0, o1, dh, clk);
}

__global__ void k2_cse_node_maxdh(int ntop, int nb_node, const int *top_nodes, const int *sup_idx, const int *sup_arr, const double *volume,
                                  double *volume_n, int reset_dhacc, double *dhacc, int o0, int o1, const double *dh, Clock *clk)
{
    if ((int)blockIdx.x >= nb_node) { surf_maxdh_block(ntop, top_nodes, o0, o1, dh, clk); return; }
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < ntop) cse_node_at(i, ntop, top_nodes, sup_idx, sup_arr, volume, volume_n, reset_dhacc, dhacc);
}

// (Round 5, measured and dropped: the four dependent loops of the surface step -- segments, nodes, edvacc_surf /
//  correct_surface_element, nodal volumes / max |dh| -- as ONE launch of ONE workgroup walking them with barriers in between: a 2-D
//  surface is a line of a few thousand nodes.  130-140 us against 28 for the four launches (profiles/r05_f_*, r05_g_*): one CU
//  has too little memory-level parallelism for chains of dependent loads, each thread walking its items one after the other.)
// compute_volume (geometry.cxx:170-201) + the element part of compute_mass (geometry.cxx:1743-1870)
__global__ void k2_volume_mass_elem(const des_params *__restrict__ p, int nn, int ne, const int *conn, const double *coord,
                                    const double *temperature, const double *props, const int *markers, int with_mass,
                                    double *volume, double *tmp_result)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    double d[3][2];
    elem_coords(coord, conn, nn, ne, e, d);
    const double vol = triangle_area(d[0], d[1], d[2]);
    volume[e] = vol;
    if (!with_mass) return;
    desk::Mix mx = {&markers[(size_t)e * p->nmat], 0, 0};
    const double bulkm = props[e], shearm = props[ne + e];
    const double mrho = desk::mat_rho(p, mx, elemT(temperature, conn, ne, e));
    const double pseudo_speed = p->max_vbc_val * p->inertial_scaling;
    double rho = p->is_quasi_static ? bulkm / (pseudo_speed * pseudo_speed) : mrho;
    double m = rho * vol / 3;
    double tm = mrho * props[3 * ne + e] * vol / 3;
    double ym = 9 * bulkm * shearm / (3 * bulkm + shearm) / 3;
    tmp_result[0 * ne + e] = m;
    tmp_result[1 * ne + e] = tm;
    tmp_result[2 * ne + e] = ym;
}

__global__ void k2_mass_node(const des_params *__restrict__ p, int nn, int ne, const int *sup_idx, const int *sup_arr, const double *volume,
                             const double *tmp_result, double *volume_n, double *mass, double *tmass, double *ymass)
{
    const int n = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (n >= nn) return;
    double vn = 0, ms = 0, tms = 0, yms = 0;
    for (int k = sup_idx[n]; k < sup_idx[n+1]; ++k) {
        const int e = sup_arr[k];
        vn += volume[e];
        ms += tmp_result[0 * ne + e];
        if (p->has_thermal_diffusion) tms += tmp_result[1 * ne + e];
        yms += tmp_result[2 * ne + e];
    }
    volume_n[n] = vn; mass[n] = ms; tmass[n] = tms; ymass[n] = yms;
}

// rotate_stress (fields.cxx:807-821, 885-900)
__device__ __forceinline__ void jaumann_rate_2d(double *s, double dt, double w2)
{
    double s_inc[3];
    s_inc[0] = -2.0 * s[2] * w2;
    s_inc[1] =  2.0 * s[2] * w2;
    s_inc[2] = s[0] * w2 - s[1] * w2;
    for (int i = 0; i < 3; ++i) s[i] += dt * s_inc[i];
}

__global__ void k2_rotate(const Clock *clk, int nn, int ne, const int *conn, const double *coord, const double *vel,
                          const double *volume, double *stress, double *strain)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    double d[3][2], shpdx[3], shpdz[3], v[3][2];
    elem_coords(coord, conn, nn, ne, e, d);
    shape_fn2(d, volume[e], shpdx, shpdz);
    elem_coords(vel, conn, nn, ne, e, v);
    double w2 = 0;
    for (int i = 0; i < 3; ++i) w2 += 0.5 * (v[i][1] * shpdx[i] - v[i][0] * shpdz[i]);
    double s[3], es[3];
    for (int i = 0; i < 3; ++i) { s[i] = stress[i*ne+e]; es[i] = strain[i*ne+e]; }
    jaumann_rate_2d(s, clk->dt, w2);
    jaumann_rate_2d(es, clk->dt, w2);
    for (int i = 0; i < 3; ++i) { stress[i*ne+e] = s[i]; strain[i*ne+e] = es[i]; }
}

// Output::average_fields (output.cxx:327-370)
__global__ void k2_average(Clock *clk, int first, int nn2, int ne, const double *coord, const double *strain, const double *stress,
                           const double *delta_plstrain, double *coord_avg0, double *strain0, double *stress_avg, double *dplstrain_avg)
{
    const int i = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i == 0 && first) clk->avg_time0 = clk->time;
    if (first) {
        if (i < nn2) coord_avg0[i] = coord[i];
        if (i < 3 * ne) { strain0[i] = strain[i]; stress_avg[i] = stress[i]; }
        if (i < ne) dplstrain_avg[i] = delta_plstrain[i];
    } else {
        if (i < 3 * ne) stress_avg[i] += stress[i];
        if (i < ne) dplstrain_avg[i] += delta_plstrain[i];
    }
}

// compute_dt (geometry.cxx:1480-1647): element reduction ...
__global__ void k2_dt_init(Clock *clk)
{
    clk->r_minl = DBL_MAX; clk->r_dt_maxwell = DBL_MAX; clk->r_dt_diffusion = DBL_MAX;
    clk->r_global_dt_min = DBL_MAX; clk->r_max_vem = 0.0;
}

// (Round 5: every workgroup stores its five partials to a slot of its own -- dt_part[q * nb + block] -- and k2_dt_reduce, one
//  workgroup, reduces them: five thousand workgroups doing five 64-bit atomic min / max on the five words of ONE cache line
//  serialised in the L2, 80-93 us for 1.28M triangles against 23 us for the 3-D engine's pass, which has stored partials since
//  round 2.  min and max are exact whatever the order: the same bits.)
// volume == nullptr (round 5): compute_volume of this step has been left to the next stress update -- its own expression on the
// moved coordinates gives the value it will store
__global__ void k2_dt_partials(const des_params *__restrict__ p, double *dt_part, int nn, int ne, const int *conn, const double *coord,
                               const double *vel, const double *temperature, const double *volume, const double *props,
                               const int *markers)
{
    __shared__ double sm[5][DES_BLOCK / 64];
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    double minl = DBL_MAX, dt_maxwell = DBL_MAX, dt_diffusion = DBL_MAX, gdt = DBL_MAX, vem = 0.0;
    if (e < ne) {
        double vx = 0.0, vy = 0.0;
        double weight = 1.0 / 3;
        for (int j = 0; j < 3; ++j) {
            int n = conn[j * ne + e];
            vx += vel[n] * weight;
            vy += vel[nn + n] * weight;
        }
        vem = sqrt(vx*vx + vy*vy);
        double d[3][2];
        elem_coords(coord, conn, nn, ne, e, d);
        double maxl = sqrt(fmax(fmax(dist2(d[0], d[1]), dist2(d[1], d[2])), dist2(d[0], d[2])));
        double minh = 2 * (volume ? volume[e] : triangle_area(d[0], d[1], d[2])) / maxl;
        const double shearm = prop2(p, props, ne, e, 1);
        dt_maxwell = 0.5 * p->visc_min / (1e-40 + shearm);
        if (p->has_thermal_diffusion) dt_diffusion = 0.5 * minh * minh / p->therm_diff_max;
        minl = minh;
        desk::Mix mx = {&markers[(size_t)e * p->nmat], 0, 0};
        gdt = minh / sqrt(shearm / desk::mat_rho(p, mx, elemT(temperature, conn, ne, e))) / 5.0;
    }
    minl = desk::wave_min(minl); dt_maxwell = desk::wave_min(dt_maxwell); dt_diffusion = desk::wave_min(dt_diffusion);
    gdt = desk::wave_min(gdt); vem = desk::wave_max(vem);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        sm[0][w] = minl; sm[1][w] = dt_maxwell; sm[2][w] = dt_diffusion; sm[3][w] = gdt; sm[4][w] = vem;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < DES_BLOCK / 64; ++w) {
            for (int q = 0; q < 4; ++q) sm[q][0] = fmin(sm[q][0], sm[q][w]);
            sm[4][0] = fmax(sm[4][0], sm[4][w]);
        }
        const size_t nb = gridDim.x;
        for (int q = 0; q < 5; ++q) dt_part[q * nb + blockIdx.x] = sm[q][0];
    }
}

// ... the partials reduced into the clock's five slots by one workgroup (what k2_dt_init + the atomics left there);
// finalize != 0: k2_dt_finalize's statements follow at once (the single engine: one launch less)
__device__ __forceinline__ void dt_finalize_body(const des_params *__restrict__ p, Clock *clk);
// (one workgroup of 1024 lanes: five rounds of requests for the 1.28M-triangle mesh's five thousand partials instead of twenty)
#define DES2_DT_REDUCE_THREADS 1024
__global__ void __launch_bounds__(DES2_DT_REDUCE_THREADS)
k2_dt_reduce(const des_params *__restrict__ p, Clock *clk, const double *dt_part, int nb, int finalize)
{
    __shared__ double sm[5][DES2_DT_REDUCE_THREADS / 64];
    double r[5] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX, 0.0};
    for (int i = threadIdx.x; i < nb; i += DES2_DT_REDUCE_THREADS) {
        for (int q = 0; q < 4; ++q) r[q] = fmin(r[q], dt_part[(size_t)q * nb + i]);
        r[4] = fmax(r[4], dt_part[(size_t)4 * nb + i]);
    }
    for (int q = 0; q < 4; ++q) r[q] = desk::wave_min(r[q]);
    r[4] = desk::wave_max(r[4]);
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 5; ++q) sm[q][threadIdx.x >> 6] = r[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < DES2_DT_REDUCE_THREADS / 64; ++w) {
            for (int q = 0; q < 4; ++q) sm[q][0] = fmin(sm[q][0], sm[q][w]);
            sm[4][0] = fmax(sm[4][0], sm[4][w]);
        }
        clk->r_minl = sm[0][0]; clk->r_dt_maxwell = sm[1][0]; clk->r_dt_diffusion = sm[2][0];
        clk->r_global_dt_min = sm[3][0]; clk->r_max_vem = sm[4][0];
        if (finalize) dt_finalize_body(p, clk);
    }
}

// ... and its tail (geometry.cxx:1597-1646)
__device__ __forceinline__ void dt_finalize_body(const des_params *__restrict__ p, Clock *clk)
{
    const double minl = clk->r_minl, dt_maxwell = clk->r_dt_maxwell, dt_diffusion = clk->r_dt_diffusion;
    const double dt_hydro_diffusion = DBL_MAX;
    clk->dt_prev = clk->dt;
    double global_max_vem = clk->r_max_vem;
    double max_vbc_val;
    if (p->characteristic_speed == 0) {
        max_vbc_val = p->max_vbc_val;
        if (p->surface_process_option > 0) max_vbc_val = fmax(max_vbc_val, clk->max_surf_vel * 5e-1);
    } else
        max_vbc_val = p->characteristic_speed;
    global_max_vem = fmax(global_max_vem, p->max_vbc_val);
    clk->max_global_vel_mag = global_max_vem;
    clk->global_dt_min = clk->r_global_dt_min;
    double dt_advection = 0.5 * minl / max_vbc_val;
    double dt_elastic = p->is_quasi_static
        ? 0.5 * minl / (max_vbc_val * p->inertial_scaling)
        : 0.5 * minl / sqrt(p->bulk_modulus[p->mattype_ref] / p->rho0[p->mattype_ref]);
    double dt = fmin(fmin(fmin(dt_elastic, dt_maxwell), fmin(dt_advection, dt_diffusion)), dt_hydro_diffusion) * p->dt_fraction;
    if (p->fixed_dt != 0) dt = p->fixed_dt;
    if (dt <= 0) clk->status = DES_ERR_RUNTIME_NAN;
    clk->dt = dt;
}

__global__ void k2_dt_finalize(const des_params *__restrict__ p, Clock *clk) { dt_finalize_body(p, clk); }

__global__ void k2_count_nan(long long n, const double *a, unsigned long long *count)
{
    const long long i = (long long)blockIdx.x * DES_BLOCK + threadIdx.x;
    if (i < n && a[i] != a[i]) atomicAdd(count, 1ull);
}

// bad_mesh_quality's reductions (remeshing.cxx:2765-2798, 2841-2844) + elem_quality (geometry.cxx:1873-1909, !THREED).
// One workgroup; "first" = lowest index, as the reference's serial loops find it.
__global__ void k2_quality(int nn, int ne, const int *conn, const double *coord, const double *volume, const unsigned *bcflag,
                           double smallest_vol, double bottom, double bottom_dist, des_quality *out)
{
    __shared__ int s_small[DES_BLOCK], s_bot[DES_BLOCK], s_worst[DES_BLOCK];
    __shared__ double s_q[DES_BLOCK];
    int small = INT_MAX, bot = INT_MAX, worst = 0;
    double q = 1;
    for (int e = threadIdx.x; e < ne; e += DES_BLOCK) {
        if (volume[e] < smallest_vol && e < small) small = e;
        double d[3][2];
        elem_coords(coord, conn, nn, ne, e, d);
        double normalization_factor = 4 * sqrt(3.0);
        double dist2_sum = dist2(d[0], d[1]) + dist2(d[1], d[2]) + dist2(d[0], d[2]);
        double quality = normalization_factor * volume[e] / dist2_sum;
        if (quality < q) { q = quality; worst = e; }
    }
    if (bottom_dist >= 0)
        for (int i = threadIdx.x; i < nn; i += DES_BLOCK)
            if ((bcflag[i] & BOUNDZ0) && fabs(coord[nn + i] - bottom) > bottom_dist && i < bot) bot = i;
    s_small[threadIdx.x] = small; s_bot[threadIdx.x] = bot; s_worst[threadIdx.x] = worst; s_q[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        // the serial loop keeps the first element attaining the strict minimum
        for (int t = 1; t < DES_BLOCK; ++t) {
            small = min(small, s_small[t]); bot = min(bot, s_bot[t]);
            if (s_q[t] < q || (s_q[t] == q && s_q[t] < 1 && s_worst[t] < worst)) { q = s_q[t]; worst = s_worst[t]; }
        }
        out->small_elem = small == INT_MAX ? -1 : small;
        out->bottom_node = bot == INT_MAX ? -1 : bot;
        out->worst_elem = worst; out->worst_quality = q; out->pad_ = 0;
    }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------

// ---- domain decomposition: the ghost-region records, the wall extent, the split compute_dt ----------
// A node record = x, z, vx, vz, T, dh (DES_X_NODE_WIDTH_2D); an element record = stress[3], strain[3],
// plstrain, stressyy (DES_X_ELEM_WIDTH_2D).  dh[] is indexed by position in top_nodes: top_pos maps a node there.
__global__ void k2_state_pack(int n_nodes, const int *nidx, const int *noff, int n_elems, const int *eidx, const int *eoff,
                              int nn, int ne, const double *coord, const double *vel, const double *temperature,
                              const int *top_pos, const double *dh, const double *stress, const double *strain,
                              const double *plstrain, const double *stressyy, double *buf)
{
    const int k = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (k < n_nodes) {
        const int n = nidx[k];
        double *r = buf + noff[k];
        r[0] = coord[n]; r[1] = coord[nn + n]; r[2] = vel[n]; r[3] = vel[nn + n]; r[4] = temperature[n];
        const int t = top_pos[n];
        r[5] = t >= 0 ? dh[t] : 0.0;
    } else if (k - n_nodes < n_elems) {
        const int e = eidx[k - n_nodes];
        double *r = buf + eoff[k - n_nodes];
        for (int j = 0; j < 3; ++j) { r[j] = stress[j*ne + e]; r[3 + j] = strain[j*ne + e]; }
        r[6] = plstrain[e]; r[7] = stressyy[e];
    }
}

__global__ void k2_state_unpack(int n_nodes, const int *nidx, const int *noff, int n_elems, const int *eidx, const int *eoff,
                                int nn, int ne, double *coord, double *vel, double *temperature, const int *top_pos, double *dh,
                                double *stress, double *strain, double *plstrain, double *stressyy, const double *buf)
{
    const int k = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (k < n_nodes) {
        const int n = nidx[k];
        const double *r = buf + noff[k];
        coord[n] = r[0]; coord[nn + n] = r[1]; vel[n] = r[2]; vel[nn + n] = r[3]; temperature[n] = r[4];
        const int t = top_pos[n];
        if (t >= 0) dh[t] = r[5];
    } else if (k - n_nodes < n_elems) {
        const int e = eidx[k - n_nodes];
        const double *r = buf + eoff[k - n_nodes];
        for (int j = 0; j < 3; ++j) { stress[j*ne + e] = r[j]; strain[j*ne + e] = r[3 + j]; }
        plstrain[e] = r[6]; stressyy[e] = r[7];
    }
}

// What apply_vbcs reads off the whole mesh (bc.cxx:251-290, 350-361), as this rank sees it: out = {max z of the x0
// wall, max -z of it, max(0, max -z) of every node}, -DBL_MAX where the rank holds no wall node -- three maxima, so the
// cross-rank reduction (MAX) is exact whatever the order.  One workgroup.
__global__ void k2_wall_local(int nb, const int *bnodes_x0, int nn, const double *coord, int with_zmin, double *out)
{
    __shared__ double sm[3][DES_BLOCK / 64];
    double mx = -DBL_MAX, mn = -DBL_MAX, nz = 0.0;
    for (int j = threadIdx.x; j < nb; j += DES_BLOCK) {
        const double z = coord[nn + bnodes_x0[j]];
        mx = fmax(mx, z); mn = fmax(mn, -z);
    }
    if (with_zmin) for (int i = threadIdx.x; i < nn; i += DES_BLOCK) nz = fmax(nz, -coord[nn + i]);
    mx = desk::wave_max(mx); mn = desk::wave_max(mn); nz = desk::wave_max(nz);
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sm[0][w] = mx; sm[1][w] = mn; sm[2][w] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < DES_BLOCK / 64; ++w) for (int q = 0; q < 3; ++q) sm[q][0] = fmax(sm[q][0], sm[q][w]);
        out[0] = sm[0][0]; out[1] = sm[1][0]; out[2] = sm[2][0];
    }
}

// ... and the reduced values into the clock, where k2_vbc_extent / k2_vbc_zmin put them on an undivided mesh
__global__ void k2_wall_set(Clock *clk, double x0_max, double neg_x0_min, double neg_zmin, int with_zmin)
{
    const bool any = x0_max != -DBL_MAX;
    clk->x0_init = any;
    clk->x0_max = any ? x0_max : 0.;
    clk->x0_min = any ? -neg_x0_min : 0.;
    clk->zmin = with_zmin ? -neg_zmin : 0.;
}

__global__ void k2_wall_set_from(Clock *clk, const double *red, int with_zmin)
{
    const bool any = red[0] != -DBL_MAX;
    clk->x0_init = any;
    clk->x0_max = any ? red[0] : 0.;
    clk->x0_min = any ? -red[1] : 0.;
    clk->zmin = with_zmin ? -red[2] : 0.;
}

// compute_dt across ranks: the element reduction's result out (six minima: the two maxima negated) ...
__global__ void k2_dt_pack(const Clock *clk, double *red)
{
    red[0] = clk->r_minl; red[1] = clk->r_dt_maxwell; red[2] = clk->r_dt_diffusion;
    red[3] = clk->r_global_dt_min; red[4] = -clk->r_max_vem; red[5] = -clk->max_surf_vel;
}

// ... and the reduced six back in, ahead of k2_dt_finalize
__global__ void k2_dt_unpack(Clock *clk, const double *red)
{
    clk->r_minl = red[0]; clk->r_dt_maxwell = red[1]; clk->r_dt_diffusion = red[2];
    clk->r_global_dt_min = red[3]; clk->r_max_vem = -red[4]; clk->max_surf_vel = -red[5];
}

#include "des_dev2d_patch.hpp"

struct FieldRef { void *ptr; long long count; int elsize; };

FieldRef field_ref(const Engine *h, int field)
{
    const long long nn = h->nn, ne = h->ne;
    switch (field) {
    case DES_F_COORD: return {h->coord, 2*nn, 8};
    case DES_F_VEL: return {h->vel, 2*nn, 8};
    case DES_F_FORCE: return {h->force, 2*nn, 8};
    case DES_F_FORCE_RESIDUAL: return {h->fres, 2*nn, 8};
    case DES_F_COORD0: return {h->coord0, 2*nn, 8};
    case DES_F_TEMPERATURE: return {h->temperature, nn, 8};
    case DES_F_VOLUME_N: return {h->volume_n, nn, 8};
    case DES_F_MASS: return {h->mass, nn, 8};
    case DES_F_TMASS: return {h->tmass, nn, 8};
    case DES_F_DHACC: return {h->dhacc, nn, 8};
    case DES_F_NTMP: return {h->ntmp, nn, 8};
    case DES_F_STRESS: return {h->stress, 3*ne, 8};
    case DES_F_STRAIN: return {h->strain, 3*ne, 8};
    case DES_F_STRAIN_RATE: return {h->strain_rate, 3*ne, 8};
    case DES_F_PLSTRAIN: return {h->plstrain, ne, 8};
    case DES_F_DELTA_PLSTRAIN: return {h->delta_plstrain, ne, 8};
    case DES_F_VISCOSITY: return {h->viscosity, ne, 8};
    case DES_F_VOLUME: return {h->volume, ne, 8};
    case DES_F_VOLUME_OLD: return {h->volume_old, ne, 8};
    case DES_F_DPRESSURE: return {h->dpressure, ne, 8};
    case DES_F_EDVOLDT: return {h->edvoldt, ne, 8};
    case DES_F_RADIOGENIC: return {h->radiogenic, ne, 8};
    case DES_F_ELEMMARKERS: return {h->markers, ne * h->nmat, 4};
    case DES_F_EDVACC_SURF: return {h->edvacc, (long long)h->etop, 8};
    case DES_F_DH: return {h->dh, (long long)h->ntop, 8};
    case DES_F_STRESSYY: return {h->stressyy, ne, 8};
    case DES_F_STRESS_AVG: return {h->stress_avg, h->stress_avg ? 3*ne : 0, 8};
    case DES_F_DPLSTRAIN_AVG: return {h->dplstrain_avg, h->dplstrain_avg ? ne : 0, 8};
    case DES_F_STRAIN0: return {h->strain0, h->strain0 ? 3*ne : 0, 8};
    case DES_F_COORD_AVG0: return {h->coord_avg0, h->coord_avg0 ? 2*nn : 0, 8};
    default: return {nullptr, -1, 0};
    }
}

#define L2(kernel, n, ...) hipLaunchKernelGGL(kernel, dim3(nblk(n)), dim3(DES_BLOCK), 0, h->stream, __VA_ARGS__)

// the launches bench.py prices (HIP events on the engine's stream around each, only while profiling is on)
enum { P2_TEMP = 0, P2_STRESS, P2_NODEAVG, P2_FORCE, P2_MASS, P2_ROTVOL, P2_EXCH, P2_COUNT };
static const char *const p2_names[P2_COUNT] = {"K2P_temp_dvoldt", "K2_stress", "K2_node_avg", "K2P_force", "K2P_mass", "K2_rotate_vol",
                                               "ghost_exchange"};
struct Prof2 {
    Engine *h; Engine::ProfRec rec; bool on; hipStream_t s;
    Prof2(Engine *h_, int k, hipStream_t s_ = nullptr) : h(h_), on(h_->prof), s(s_ ? s_ : h_->stream)
    { if (on) { hipEventCreateWithFlags(&rec.a, hipEventDisableSystemFence); hipEventCreateWithFlags(&rec.b, hipEventDisableSystemFence);       // (timing-only events: engine/launch.hpp)
                rec.k = k; hipEventRecord(rec.a, s); } }
    ~Prof2() { if (on) { hipEventRecord(rec.b, s); h->prof_recs.push_back(rec); } }
};

void refresh_props(Engine *h)
{
    if (!h->markers_dirty) return;
    L2(k2_props, h->ne, h->d_p, h->ne, h->markers, h->props, h->mono);
    h->markers_dirty = false;
}

void launch_volume_mass(Engine *h, bool with_mass)
{
    L2(k2_volume_mass_elem, h->ne, h->d_p, h->nn, h->ne, h->conn, h->coord, h->temperature, h->props, h->markers,
       with_mass ? 1 : 0, h->volume, h->tmp_result);
    if (with_mass)
        L2(k2_mass_node, h->nn, h->d_p, h->nn, h->ne, h->sup_idx, h->sup_arr, h->volume, h->tmp_result, h->volume_n,
           h->mass, h->tmass, h->ymass);
}

inline bool wall_needs_zmin(const Engine *h) { return h->p.vbc_types[0] == 3 && h->p.bottom_shear_zone_thickness > 0.; }

inline PatchArgs patch_args(const Engine *h)
{
    PatchArgs a = {h->nn, h->ne, h->p_npb, h->p_nb, h->p_pn_cap, h->p_inc_cap, h->po_ptr, h->po_id, h->po_slot, h->pe_ptr, h->pe_pack, h->pn_ptr,
                   h->pn_id, h->sup_idx, h->d_bperm};
    return a;
}

// overlapped schedule: the wall's extent of the step before may still be on its way through the side stream
inline void join_wall(Engine *h)
{
    if (!h->wall_pending) return;
    hipStreamWaitEvent(h->stream, h->ev_wall, 0);
    h->wall_pending = false;
}

void launch_vbcs(Engine *h, bool apply = true, bool tick = false)
{
    join_wall(h);
    // (a decomposed mesh: the wall's extent is the cross-rank maximum wall_set left in the clock -- the coordinates
    // have not moved since it was taken, update_coordinate comes after apply_vbcs)
    if (!h->halo) {
        hipLaunchKernelGGL(k2_vbc_extent, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->nbn[iboundx0], h->bnodes[iboundx0], h->nn, h->coord, h->d_clk,
                           tick ? 1 : 0);
        if (wall_needs_zmin(h)) {
            L2(k2_vbc_zmin, h->nn, h->nn, h->coord, h->neg_zmin);
            hipLaunchKernelGGL(k2_vbc_zmin_fin, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->neg_zmin);
        }
    }
    if (apply) L2(k2_apply_vbcs, h->nn, h->d_p, h->d_clk, h->nn, h->bcflag, h->bnormals, h->edge_vec, h->edge_slot, h->coord, h->vel);
}

// elist / nlist: the elements of this launch (nullptr: all ne); `first`: the first launch of this update_stress (a split one
// has two); the caller clears geo_pending once the whole update is issued
template <class M>
void launch_stress(Engine *h, bool fused = false, double *stress_out = nullptr, const int *elist = nullptr, int nlist = -1, bool first = true)
{
    if (nlist < 0) nlist = h->ne;
    if (nlist == 0) return;
    // the count of THIS update_stress: one slot per wavefront, zeroed once per counted step (a split update -- the overlapped
    // schedule's far and near elements -- writes the second launch's slots behind the first's)
    if (h->count_past && first) { hipMemsetAsync(h->past_part, 0, (size_t)h->past_cap * sizeof(int), h->stream); h->past_base = 0; h->past_used = true; }
    const int past_base = h->past_base;
    if (h->count_past) h->past_base += nblk(nlist) * (DES_BLOCK / 64);
    Prof2 pr(h, P2_STRESS);
    const int rot = ((h->p.rheol_type & DES_RH_ELASTIC) ? (h->rot_prev_dt ? 2 : 1) : 0) | (h->sr_fused ? 4 : 0);
#define K2S_ARGS(out, outs) h->d_p, h->d_vt, h->d_clk, h->ne, h->conn, h->temperature, h->props, h->markers, h->edvoldt, \
        h->volume, h->volume_old, h->stress, h->strain, h->strain_rate, h->stressyy, h->plstrain, h->delta_plstrain, \
        h->viscosity, h->dpressure, h->etmp, h->count_past ? 1 : 0, h->ntmp, out, h->nn, h->coord, h->vel, rot, outs, elist, nlist, \
        h->mono, h->pptab, h->past_part, past_base, h->stress + 2 * (size_t)h->ne
    if (fused && h->p.rheol_type == DES_RH_EVP) {          // the common rheology has instantiations of its own
        if (h->geo_pending) L2((k2_stress<M, 2, DES_RH_EVP>), nlist, K2S_ARGS(stress_out, h->elide ? 0 : 1));
        else                L2((k2_stress<M, 1, DES_RH_EVP>), nlist, K2S_ARGS(stress_out, h->elide ? 0 : 1));
    } else if (fused && h->geo_pending) L2((k2_stress<M, 2>), nlist, K2S_ARGS(stress_out, h->elide ? 0 : 1));
    else if (fused)                     L2((k2_stress<M, 1>), nlist, K2S_ARGS(stress_out, h->elide ? 0 : 1));
    else                                L2((k2_stress<M, 0>), nlist, K2S_ARGS(h->stress, 1));
#undef K2S_ARGS
}

// update_force's boundary terms in the reference's order (fields.cxx:682-691)
void launch_stress_bcs(Engine *h)
{
    const des_params &p = h->p;
    if (p.gravity != 0) {
        for (int i = 0; i < DES_NBDRY; i++) {
            if (p.vbc_types[i] != 0 && p.vbc_types[i] != 2 && p.vbc_types[i] != 4) continue;
            if (i == iboundz0 && !p.has_winkler_foundation) continue;
            if (i == iboundz1 && !p.has_water_loading) continue;
            if (h->nbf[i] == 0) continue;
            if (h->patch && h->binc[i]) {
                if (h->nbn[i])
                    L2(k2_sbc_direct, h->nbn[i], h->d_p, i, h->nbn[i], h->bnodes[i], h->binc[i], h->nn, h->ne, h->conn, h->coord,
                       h->temperature, h->markers, h->force);
                continue;
            }
            L2(k2_sbc_facet, h->nbf[i], h->d_p, i, h->nbf[i], h->bf_elem[i], h->bf_facet[i], h->nn, h->ne, h->conn, h->coord,
               h->temperature, h->markers, h->tmp_result, h->etmp_int);
            if (h->nbn[i])
                L2(k2_sbc_node, h->nbn[i], h->nbn[i], h->bnodes[i], h->bf_facet[i], h->nn, h->ne, h->conn, h->sup_idx, h->sup_arr,
                   h->tmp_result, h->etmp_int, h->force);
            L2(k2_sbc_reset, h->nbf[i], h->nbf[i], h->bf_elem[i], h->etmp_int);
        }
        if (p.has_elastic_foundation && h->nbn[iboundz0])
            L2(k2_elastic_foundation, h->nbn[iboundz0], h->d_p, h->nbn[iboundz0], h->bnodes[iboundz0], h->nn, h->coord, h->coord0, h->force);
    }
    for (int i = 0; i < 6 && !h->no_neumann; ++i) {
        if (p.stress_bc_types[i] == 0 || h->nbf[i] == 0) continue;
        hipLaunchKernelGGL(k2_neumann, dim3(1), dim3(64), 0, h->stream, h->d_p, i, h->nbf[i], h->bf_elem[i], h->bf_facet[i],
                           h->nn, h->ne, h->conn, h->coord, h->force);
    }
}

// update_mesh (dynearthsol.cxx:448-493) after update_coordinate, in two parts: up to the committed surface heights
// (where a decomposed mesh refreshes its ghost region) ...
void launch_surface_commit(Engine *h)
{
    const des_params &p = h->p;
    if (h->xz_pre_valid && p.surface_process_option == 1 && h->etop > 0 && h->ntop > 0) {
        // (k2p_force<1> of this step left the moved top nodes in xz_pre: segments and nodes in one launch)
        const int nbn = nblk(h->ntop);
        hipLaunchKernelGGL(k2_surf_commit, dim3(nbn + (h->res_fin_pending ? 1 : 0)), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->ntop, h->nn,
                           h->top_nodes, h->xz_pre, h->total_dx, h->total_slope, h->coord, h->dhacc, h->dh, nbn, h->res_count, h->res_part);
        h->res_fin_pending = false;
        h->xz_pre_valid = false;
        return;
    }
    h->xz_pre_valid = false;
    if (h->res_fin_pending && p.surface_process_option == 1 && h->etop > 0) {
        const int nbs = nblk(h->etop);
        hipLaunchKernelGGL(k2_surf_seg_resfin, dim3(nbs + 1), dim3(DES_BLOCK), 0, h->stream, h->etop, h->nn, h->ne, h->top_nodes, h->coord,
                           h->etmp, h->tmp_result, nbs, h->res_count, h->res_part, h->d_clk);
        h->res_fin_pending = false;
    } else
    // surface_processes (bc.cxx:1709-1872)
    if (p.surface_process_option == 1 && h->etop > 0)
        L2(k2_surf_seg, h->etop, h->etop, h->nn, h->ne, h->top_nodes, h->coord, h->etmp, h->tmp_result);
    if (h->ntop > 0)
        L2(k2_surf_node, h->ntop, h->d_p, h->d_clk, h->ntop, h->nn, h->ne, h->top_nodes, h->etmp, h->tmp_result, h->total_dx,
           h->total_slope, h->coord, h->dhacc, h->dh);
}

// ... and the rest, in two parts: the surface bookkeeping (it reads the ghost nodes' heights) ...
void launch_update_mesh_surface(Engine *h, long long steps)
{
    const des_params &p = h->p;
    const bool at_interval = steps % p.quality_check_step_interval == 0;
    const int decay = !(steps % p.quality_check_step_interval && steps != 0) ? 1 : 0;     // bc.cxx:1848
    if (h->patch && h->ntop > 0 && h->fold_on) {
        // one launch: the nodal part forms the areas it sums itself (cse_node_areas_at)
        const int nbe = nblk(h->etop), nbc = nblk(h->ntop_elems), nbn = nblk(h->ntop);
        hipLaunchKernelGGL(k2_surf_edv_cse_all, dim3(nbe + nbc + nbn + 1), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->etop, h->ntop_elems,
                           h->ntop, nbe, nbc, nbn, h->nn, h->ne, h->ean, h->conn_surf, h->top_elems, h->top_nodes, h->conn, h->markers,
                           h->sup_idx, h->sup_arr, decay, (steps != 0 && at_interval) ? 1 : 0, h->o0, h->o1, h->coord, h->dh, h->edvacc,
                           h->volume, h->volume_n, h->dhacc, h->plstrain, h->stress, h->strain, h->strain_rate);
        return;
    }
    if (h->patch && h->ntop > 0) {
        const int nbe = nblk(h->etop), nbc = nblk(h->ntop_elems), nbn = nblk(h->ntop);
        if (nbe + nbc > 0)
            hipLaunchKernelGGL(k2_surf_edv_cse_elem, dim3(nbe + nbc), dim3(DES_BLOCK), 0, h->stream, h->d_p, h->d_clk, h->etop, h->ntop_elems,
                               nbe, h->nn, h->ne, h->ean, h->conn_surf, h->top_elems, h->conn, h->markers, decay, h->coord, h->dh,
                               h->edvacc, h->volume, h->plstrain, h->stress, h->strain, h->strain_rate);
        hipLaunchKernelGGL(k2_cse_node_maxdh, dim3(nbn + 1), dim3(DES_BLOCK), 0, h->stream, h->ntop, nbn, h->top_nodes, h->sup_idx, h->sup_arr,
                           h->volume, h->volume_n, (steps != 0 && at_interval) ? 1 : 0, h->dhacc, h->o0, h->o1, h->dh, h->d_clk);
        return;
    }
    if (h->etop > 0)
        L2(k2_surf_edv, h->etop, h->etop, h->nn, h->ean, h->conn_surf, h->coord, h->dh, h->edvacc);
    hipLaunchKernelGGL(k2_surf_maxdh, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->ntop, h->top_nodes, h->o0, h->o1, h->dh, h->d_clk);
    if (h->ntop_elems > 0)
        L2(k2_cse_elem, h->ntop_elems, h->d_p, h->d_clk, h->ntop_elems, h->top_elems, h->nn, h->ne, h->conn, h->coord, h->markers,
           decay, h->volume, h->plstrain, h->stress, h->strain, h->strain_rate);
    if (h->ntop > 0)
        L2(k2_cse_node, h->ntop, h->ntop, h->top_nodes, h->sup_idx, h->sup_arr, h->volume, h->volume_n,
           (steps != 0 && at_interval) ? 1 : 0, h->dhacc);
}

// compute_mass over the node-block patches (it forms the volumes it sums from the coordinates); blist / nb: the blocks of this launch
void launch_patch_mass(Engine *h, const int *blist = nullptr, int nb = -1, const double *T = nullptr)
{
    PatchArgs a = patch_args(h);
    if (!T) T = h->temperature;
    if (nb >= 0) { a.blist = blist; a.nb = nb; }
    if (a.nb == 0) return;
    Prof2 pr(h, P2_MASS);
    hipLaunchKernelGGL(k2p_mass, dim3((a.nb + 7) / 8 * 8), dim3(DES2_PATCH_THREADS), 8 * (3 * (size_t)h->p_pn_cap + 4 * (size_t)h->p_inc_cap), h->stream, h->d_p, a, h->coord,
                       T, h->props, h->markers, h->mono, h->volume_n, h->mass, h->tmass, h->ymass);
}

// ... and compute_volume, rotate_stress, compute_mass
void launch_update_mesh_rest(Engine *h, long long steps, bool rotate, bool defer = false)
{
    if (!h->surf_pending) launch_update_mesh_surface(h, steps);
    if (!defer) std::swap(h->volume, h->volume_old);
    refresh_props(h);
    if (h->patch) {
        // compute_volume + rotate_stress in one element pass -- or, inside a multi-step call, left to the next step's stress
        // update (k2_stress<M, 2>), which reads and writes the same stress and strain anyway
        if (defer) h->geo_pending = true;
        else { Prof2 pr(h, P2_ROTVOL); L2(k2_rotate_vol, h->ne, h->d_clk, rotate ? 1 : 0, h->nn, h->ne, h->conn, h->coord, h->vel, h->volume, h->stress, h->strain); }
        // compute_mass: now -- or, when the next step follows at once and forms its volumes from the coordinates anyway
        // (defer), inside that step's update_temperature + compute_dvoldt pass (k2p_temp_dvoldt<1>)
        if (defer && h->mass_fuse_on) h->mass_pending = true;
        else launch_patch_mass(h);
        return;
    }
    launch_volume_mass(h, true);
}

void launch_dt(Engine *h)
{
    refresh_props(h);
    // (geo_pending: this step's compute_volume waits in the next stress update -- and its rotation must keep this step's dt)
    if (h->geo_pending) h->rot_prev_dt = true;
    L2(k2_dt_partials, h->ne, h->d_p, h->dt_part, h->nn, h->ne, h->conn, h->coord, h->vel, h->temperature,
       h->geo_pending ? (const double *)nullptr : h->volume, h->props, h->markers);
    hipLaunchKernelGGL(k2_dt_reduce, dim3(1), dim3(DES2_DT_REDUCE_THREADS), 0, h->stream, h->d_p, h->d_clk, h->dt_part, nblk(h->ne), 1);
}

int sync_clock(Engine *h)
{
    HIP2(hipMemcpyAsync(h->h_clk, h->d_clk, sizeof(Clock), hipMemcpyDeviceToHost, h->stream));
    HIP2(hipStreamSynchronize(h->stream));
    return DES_OK;
}

// strain rate -> stress -> force -> velocity -> residual: the part of a step the pseudo-transient loop repeats
// Patch path: `thermal` -- update_temperature rides in the first patch pass; `tail` -- apply_vbcs and update_coordinate
// follow in the velocity kernel (a plain step with a moving mesh), the residual's final sum is left to the surface kernel.
// update_temperature + compute_dvoldt over the listed node blocks (nullptr / -1: all of them); T_in -> T_out
// (mass_pending: compute_mass of the step before rides in this pass -- step_back left it out)
void launch_temp_dvoldt(Engine *h, bool thermal, const double *T_in, double *T_out, const int *blist = nullptr, int nb = -1)
{
    PatchArgs a = patch_args(h);
    if (nb >= 0) { a.blist = blist; a.nb = nb; }
    if (a.nb == 0) return;
    Prof2 pr(h, P2_TEMP);
#define K2T_ARGS h->d_p, h->d_clk, thermal ? 1 : 0, h->geo_pending ? 1 : 0, a, \
                       h->bcflag, h->coord, h->vel, T_in, T_out, h->volume, h->radiogenic_zero ? (const double *)nullptr : h->radiogenic, h->props, \
                       h->markers, h->mono, h->tmass, h->volume_n, h->ntmp, h->strain_rate, h->volume_n, h->mass, h->tmass, h->ymass, pre, nb_tail, pre ? h->pt_ptr : h->pt_zero, h->p_pe_cap, (h->elide ? 0 : 1) | (h->sr_fused ? 2 : 0)
    const SurfPre *pre = nullptr;
    int nb_tail = 0;
    if (h->surf_pending && h->mass_pending) {
        // (the surface step the step before left behind: one_step)
        nb_tail = ((h->etop + DES2_PATCH_THREADS - 1) / DES2_PATCH_THREADS + 7) / 8 * 8;     // (in front of the blocks: a multiple of 8)
        pre = reinterpret_cast<const SurfPre *>(h->d_surfpre);
    }
    if (h->mass_pending) {
        // (compute_mass's four sums and this pass's two in one node phase: five values per patch ELEMENT, one per incidence and
        //  the incidence's element as 16 bits -- des_dev2d_patch.hpp)
        const size_t lds = 8 * (5 * (size_t)a.pn_cap + 5 * (size_t)h->p_pe_cap + (size_t)a.inc_cap) + 2 * (size_t)a.inc_cap;
        // (two patch elements per lane at most when the mesh's largest patch allows: 17 registers less)
        if (h->p_pe_cap <= 2 * DES2_PATCH_THREADS && !h->it3_forced) {
            if (lds > 65536) hipFuncSetAttribute((const void *)k2p_temp_dvoldt<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k2p_temp_dvoldt<1, 2>), dim3((a.nb + 7) / 8 * 8 + nb_tail), dim3(DES2_PATCH_THREADS), lds, h->stream, K2T_ARGS);
        } else {
            if (lds > 65536) hipFuncSetAttribute((const void *)k2p_temp_dvoldt<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k2p_temp_dvoldt<1, 3>), dim3((a.nb + 7) / 8 * 8 + nb_tail), dim3(DES2_PATCH_THREADS), lds, h->stream, K2T_ARGS);
        }
    }
    else
        hipLaunchKernelGGL(k2p_temp_dvoldt<0>, dim3((a.nb + 7) / 8 * 8), dim3(DES2_PATCH_THREADS), 8 * (5 * (size_t)a.pn_cap + 2 * (size_t)a.inc_cap), h->stream, K2T_ARGS);
#undef K2T_ARGS
}

// Overlapped schedule: the first two passes of a step on the blocks and elements far from the cut (set_halo: nothing they
// read is written by the unpack or by the surface bookkeeping left over from the step before) -- issued while that step's
// exchange is still on the side stream, by launch_mechanics or, earlier, by front_begin
template <class M>
void launch_mechanics_far(Engine *h, bool nmd, bool thermal)
{
    // (the new temperatures go to the other buffer: a block must not move a temperature another block may still read)
    h->mT_in = h->temperature; h->mT_out = h->temperature_alt;
    if (thermal) std::swap(h->temperature, h->temperature_alt);         // k2_stress reads the new ones
    launch_temp_dvoldt(h, thermal, h->mT_in, h->mT_out, h->d_blist, h->nb_deep);
    launch_stress<M>(h, true, nmd ? h->stress_pre : h->stress, h->d_elist, h->ne_deep, true);
    h->far_issued = true;
}

// k2p_force<1> (the nodal tail of a plain step inside the force pass): boundary loads that can all be formed per node, no Neumann
// tractions / elastic foundation behind them (create: fold_ok).  A rank of a decomposed mesh takes it too: every local node is
// some block's own (the ghost region goes stale as before), the wall's extent is the cross-rank one the clock already holds,
// the residual partial only counts owned nodes.
inline bool fold_now(const Engine *h) { return h->fold_on && h->fold_ok && h->patch && !h->no_neumann; }

template <class M>
void launch_mechanics(Engine *h, bool nmd, bool thermal = false, bool tail = false)
{
    const int nn = h->nn, ne = h->ne;
    if (h->patch) {
        const PatchArgs a = patch_args(h);
        double *const s_law = nmd ? h->stress_pre : h->stress;
        if (h->join_pending) {
            // overlapped schedule: behind the join with the exchange of the step before, that step's surface bookkeeping,
            // compute_mass of the blocks near the cut (on the temperatures of that step), and the rest of the two passes
            if (!h->far_issued) launch_mechanics_far<M>(h, nmd, thermal);
            hipStreamWaitEvent(h->stream, h->ev_join, 0);
            h->join_pending = false; h->far_issued = false;
            launch_update_mesh_surface(h, h->back_steps);
            if (!h->mass_pending) launch_patch_mass(h, h->d_blist + h->nb_deep, h->p_nb - h->nb_deep, h->mT_in);
            launch_temp_dvoldt(h, thermal, h->mT_in, h->mT_out, h->d_blist + h->nb_deep, h->p_nb - h->nb_deep);
            launch_stress<M>(h, true, s_law, h->d_elist + h->ne_deep, ne - h->ne_deep, false);
        } else {
            const double *const T_in = h->temperature;
            double *const T_out = h->temperature_alt;
            if (thermal) std::swap(h->temperature, h->temperature_alt);
            h->sr_fused = h->sr_fuse_on && !h->halo && h->geo_pending && h->mass_pending;
            launch_temp_dvoldt(h, thermal, T_in, T_out);
            launch_stress<M>(h, true, s_law);
            h->sr_fused = false;
        }
        h->geo_pending = false; h->mass_pending = false; h->surf_pending = false; h->rot_prev_dt = false;
        const bool fold = tail && fold_now(h);
        // (the folded tail wants the wall's extent in the clock before the force pass: it rides in the nodal average's launch)
        const bool avg_extent = fold && nmd && (h->halo || !wall_needs_zmin(h));
        if (avg_extent) {
            Prof2 pr(h, P2_NODEAVG);
            // (decomposed: no local extent -- the clock holds the cross-rank one --, the extra workgroup only counts the step)
            const int tick = h->tick_pending ? (h->halo ? 2 : 1) : 0;
            const int extra = (h->halo && !tick) ? 0 : 1;
            hipLaunchKernelGGL(k2_node_avg_extent, dim3(nblk(nn) + extra), dim3(DES_BLOCK), 0, h->stream, nn, nblk(nn), h->sup_idx, h->sup_arr, h->etmp,
                               h->volume_n, h->ntmp, h->nbn[iboundx0], h->bnodes[iboundx0], h->coord, h->d_clk, tick);
            h->tick_pending = false;
        } else if (nmd) { Prof2 pr(h, P2_NODEAVG); L2(k2_node_avg, nn, nn, h->sup_idx, h->sup_arr, h->etmp, h->volume_n, h->ntmp); }
        ForceTail ft = {h->d_clk, h->mass, h->ymass, h->bcflag, h->bnormals, h->edge_vec, h->edge_slot, h->vel, h->coord_alt, h->conn,
                        h->sbcn_idx, h->sbcn_ent, h->res_part, h->o0, h->o1, h->nn_global, h->fold_top_pos, h->xz_pre};
        if (fold) {
            // Everything nodal behind the force sums rides in k2p_force<1>.  The wall's extent first (the coordinates have not
            // moved since the step began; with it the step is counted: nothing between here and apply_vbcs reads the clock's time)
            if (!avg_extent) {
                if (h->halo) { join_wall(h); if (h->tick_pending) hipLaunchKernelGGL(k2_clock, dim3(1), dim3(1), 0, h->stream, h->d_clk); }
                else launch_vbcs(h, false, h->tick_pending);
                h->tick_pending = false;
            } else join_wall(h);
            { Prof2 pr(h, P2_FORCE);
            if (h->p_pe_cap <= 2 * DES2_PATCH_THREADS && !h->it3_forced)
            hipLaunchKernelGGL((k2p_force<1, 2>), dim3((h->p_nb + 7) / 8 * 8), dim3(DES2_PATCH_THREADS), 8 * (4 * (size_t)a.pn_cap + 2 * (size_t)a.inc_cap), h->stream, h->d_p, nmd ? 1 : 0, h->elide ? 0 : 1, a, h->coord,
                               h->temperature, h->ntmp, h->volume, h->dpressure, s_law, h->stress, h->props, h->markers, h->mono, h->force, h->fres, ft, h->stress + 2 * (size_t)h->ne);
            else
            hipLaunchKernelGGL((k2p_force<1, 3>), dim3((h->p_nb + 7) / 8 * 8), dim3(DES2_PATCH_THREADS), 8 * (4 * (size_t)a.pn_cap + 2 * (size_t)a.inc_cap), h->stream, h->d_p, nmd ? 1 : 0, h->elide ? 0 : 1, a, h->coord,
                               h->temperature, h->ntmp, h->volume, h->dpressure, s_law, h->stress, h->props, h->markers, h->mono, h->force, h->fres, ft, h->stress + 2 * (size_t)h->ne);
            }
            std::swap(h->coord, h->coord_alt);             // the moved coordinates are the current ones from here on
            h->xz_pre_valid = true;
            h->res_count = h->p_nb;
            // (surf_late: never the last step of a call -- nothing reads this step's residual sum)
            if (h->surf_late) ;
            else if (h->p.surface_process_option == 1 && h->etop > 0) h->res_fin_pending = true;
            else hipLaunchKernelGGL(k2_residual_fin, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->res_count, h->res_part, h->d_clk);
            return;
        }
        { Prof2 pr(h, P2_FORCE);
        hipLaunchKernelGGL(k2p_force<0>, dim3((h->p_nb + 7) / 8 * 8), dim3(DES2_PATCH_THREADS), 8 * (4 * (size_t)a.pn_cap + 2 * (size_t)a.inc_cap), h->stream, h->d_p, nmd ? 1 : 0, h->elide ? 0 : 1, a, h->coord,
                           h->temperature, h->ntmp, h->volume, h->dpressure, s_law, h->stress, h->props, h->markers, h->mono, h->force, h->fres, ft, h->stress + 2 * (size_t)h->ne);
        }
        launch_stress_bcs(h);
        join_wall(h);
        if (tail) {
            launch_vbcs(h, false, h->tick_pending);     // the wall's extent into the clock (the coordinates have not moved yet)
            h->tick_pending = false;
            L2(k2_node_final, nn, h->d_p, h->d_clk, nn, h->mass, h->ymass, h->bcflag, h->bnormals, h->edge_vec, h->edge_slot,
               h->force, h->vel, h->coord);
        } else
            L2(k2_damp_vel, nn, h->d_p, h->d_clk, nn, h->mass, h->ymass, h->force, h->vel);
        L2(k2_residual_part, h->o1 - h->o0, nn, h->o0, h->o1, h->nn_global, h->fres, h->res_part);
        h->res_count = nblk(h->o1 - h->o0);
        if (tail && h->p.surface_process_option == 1 && h->etop > 0) h->res_fin_pending = true;
        else hipLaunchKernelGGL(k2_residual_fin, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->res_count, h->res_part, h->d_clk);
        return;
    }
    L2(k2_strain_rate, ne, nn, ne, h->conn, h->coord, h->vel, h->volume, h->strain_rate, h->etmp);
    L2(k2_node_avg, nn, nn, h->sup_idx, h->sup_arr, h->etmp, h->volume_n, h->ntmp);
    L2(k2_edvoldt, ne, ne, h->conn, h->ntmp, h->edvoldt);
    launch_stress<M>(h);
    if (nmd) {
        L2(k2_node_avg, nn, nn, h->sup_idx, h->sup_arr, h->etmp, h->volume_n, h->ntmp);
        L2(k2_nmd_apply, ne, ne, h->conn, h->ntmp, h->dpressure, h->stress);
    }
    L2(k2_force_elem, ne, h->d_p, nn, ne, h->conn, h->coord, h->temperature, h->volume, h->stress, h->props, h->markers, h->tmp_result);
    L2(k2_force_node, nn, nn, ne, h->sup_idx, h->sup_arr, h->sup_lidx, h->tmp_result, h->force, h->fres);
    launch_stress_bcs(h);
    L2(k2_damp_vel, nn, h->d_p, h->d_clk, nn, h->mass, h->ymass, h->force, h->vel);
    L2(k2_residual_part, h->o1 - h->o0, nn, h->o0, h->o1, h->nn_global, h->fres, h->res_part);
    hipLaunchKernelGGL(k2_residual_fin, dim3(1), dim3(DES_BLOCK), 0, h->stream, nblk(h->o1 - h->o0), h->res_part, h->d_clk);
}

int set_pt(Engine *h, int on)
{
    static const int vals[2] = {0, 1};
    HIP2(hipMemcpyAsync(&h->d_clk->pt, &vals[on ? 1 : 0], sizeof(int), hipMemcpyHostToDevice, h->stream));
    return DES_OK;
}

// (re)built whenever the owned range is set: create (the whole mesh) and set_halo
int build_residual_blocks(Engine *h)
{
    const int B = des_res_block(h->nn_global);
    h->res_nb_global = (h->nn_global + B - 1) / B;
    if (h->g0 % B != 0 || h->g0 / B + (h->o1 - h->o0 + B - 1) / B > h->res_nb_global) {
        h->err = "des_halo::owned_global_begin is not a multiple of the residual's block size (des_params.h: des_res_block)";
        return DES_ERR_INTERNAL;
    }
    int rc = dalloc(h, h->res_blocks, (size_t)h->res_nb_global);        // (an earlier array stays in h->allocs until destroy)
    if (rc) return rc;
    HIP2(hipMemsetAsync(h->res_blocks, 0, (size_t)h->res_nb_global * sizeof(double), h->stream));
    return DES_OK;
}

// this rank's block partials into their places of the (zero-filled) global array
void launch_residual_blocks(Engine *h)
{
    const int B = des_res_block(h->nn_global), nown = h->o1 - h->o0, nb_own = (nown + B - 1) / B;
    if (nb_own < h->res_nb_global) hipMemsetAsync(h->res_blocks, 0, (size_t)h->res_nb_global * sizeof(double), h->stream);
    hipLaunchKernelGGL(k2_residual_blocks, dim3((nb_own + DES_BLOCK / 64 - 1) / (DES_BLOCK / 64)), dim3(DES_BLOCK), 0, h->stream,
                       nown, B, nb_own, h->nn, h->o0, h->nn_global, h->fres, h->res_blocks + h->g0 / B);
}

// the partition-independent residual on the engine's stream: one engine, or a rank on RCCL
int residual_global(Engine *h)
{
    const int B = des_res_block(h->nn_global), nb_own = (h->o1 - h->o0 + B - 1) / B;
    launch_residual_blocks(h);
    if (nb_own < h->res_nb_global) {
        if (!h->comm) { h->err = "the pseudo-transient loop on a decomposed mesh needs the ranks' residual partials: des_dev_step on RCCL, des_dev_step_group, or des_dev_phase + des_dev_residual_blocks / _set"; return DES_ERR_UNSUPPORTED; }
        const ncclResult_t r = ncclAllReduce(h->res_blocks, h->res_blocks, (size_t)h->res_nb_global, ncclDouble, ncclSum, h->comm, h->stream);
        if (r != ncclSuccess) { h->err = std::string("RCCL: ") + ncclGetErrorString(r); return DES_ERR_RESOURCE; }
    }
    hipLaunchKernelGGL(k2_residual_final, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->res_blocks, h->res_nb_global);
    return DES_OK;
}

// one iteration of the pseudo-transient loop (on a decomposed mesh: the ghost region refreshed just before)
template <class M>
void pt_iteration(Engine *h)
{
    const des_params &p = h->p;
    launch_vbcs(h);
    if (p.has_moving_mesh) {
        L2(k2_update_coord, 2 * h->nn, h->d_clk, 2 * h->nn, h->vel, h->coord);
        std::swap(h->volume, h->volume_old);
        refresh_props(h);
        launch_volume_mass(h, true);
    }
    launch_mechanics<M>(h, false);
}

// The pseudo-transient loop of a step (dynearthsol.cxx:803-864): the quasi-static part of the step repeated
// with the boundaries at rest (bc.cxx:330-343) and update_mesh without surface processes
// (dynearthsol.cxx:456-461) until the relative change of the residual drops below the tolerance.  The host
// joins the stream once per iteration for that decision, as the reference's loop does.  A rank of a decomposed mesh on RCCL
// (round 4): the ghost region (and the wall's extent) refreshed before every iteration, the residual the partition-
// independent one, its block partials all-reduced -- every rank takes the decision a single engine takes.
template <class M>
int pt_loop(Engine *h)
{
    const des_params &p = h->p;
    int rc;
    const bool multi = h->halo && h->halo_set;
    if ((rc = residual_global(h)) || (rc = sync_clock(h))) return rc;
    double residual_old = h->h_clk->l2_residual;
    if ((rc = set_pt(h, 1))) return rc;
    rc = [&]() -> int {
        int r;
        for (int pt_step = 0; pt_step < p.PT_max_iter; ++pt_step) {
            if (multi && (r = exchange_rccl(h))) return r;
            pt_iteration<M>(h);
            if ((r = residual_global(h)) || (r = sync_clock(h))) return r;
            ++h->n_pt_iterations;
            const double l2 = h->h_clk->l2_residual;
            const double relative_change = std::fabs((l2 - residual_old) / residual_old);
            if (relative_change < p.PT_relative_tolerance) break;
            residual_old = l2;
        }
        return DES_OK;
    }();
    const int rc_off = set_pt(h, 0);       // on EVERY exit: an error inside the loop must not leave Clock::pt set for later steps
    return rc ? rc : rc_off;
}

int step_front_rest(Engine *h);

// the step counted, the material means up to date
void front_clock(Engine *h)
{
    const des_params &p = h->p;
    const bool tail = h->patch && !h->iso && !p.has_PT && p.has_moving_mesh;
    if (!h->iso) {
        if (tail && (!h->halo || fold_now(h))) h->tick_pending = true;      // (k2_vbc_extent / the nodal average's extra workgroup counts the step)
        else hipLaunchKernelGGL(k2_clock, dim3(1), dim3(1), 0, h->stream, h->d_clk);
        ++h->steps_host;
    }
    refresh_props(h);
}

// Overlapped schedule: the beginning of the NEXT step, issued before the host turns to the exchange of this one (whose
// RCCL calls cost the host tens of microseconds: the device has this much to do meanwhile).  step_front picks up behind it.
template <class M>
void front_begin(Engine *h)
{
    front_clock(h);
    launch_mechanics_far<M>(h, !h->iso && h->p.is_using_mixed_stress, !h->iso && h->p.has_thermal_diffusion);
}

// A step up to the committed surface heights ...
template <class M>
int step_front(Engine *h)
{
    const des_params &p = h->p;
    const int nn = h->nn, ne = h->ne;
    const bool tail = h->patch && !h->iso && !p.has_PT && p.has_moving_mesh;
    if (!h->far_issued) front_clock(h);                    // (else front_begin has done this much of the step already)
    const bool thermal = !h->iso && p.has_thermal_diffusion;
    if (thermal && !h->patch) {
        L2(k2_temp_elem, ne, h->d_p, nn, ne, h->conn, h->coord, h->temperature, h->volume, h->radiogenic, h->props, h->markers, h->tmp_result);
        L2(k2_temp_node, nn, h->d_p, h->d_clk, nn, ne, h->sup_idx, h->sup_arr, h->sup_lidx, h->bcflag, h->tmp_result, h->tmass, h->temperature);
    }
    launch_mechanics<M>(h, !h->iso && p.is_using_mixed_stress, thermal, tail);
    if (!h->iso && p.has_PT) {
        // (pt_defer: the loop is driven from outside -- the engines of a group in lockstep, or the caller of des_dev_phase --
        //  and step_front_rest follows)
        if (h->pt_defer) return DES_OK;
        int rc = pt_loop<M>(h); if (rc) return rc;
    }
    return step_front_rest(h);
}

// ... the rest of the front, behind the pseudo-transient loop if there is one
int step_front_rest(Engine *h)
{
    const des_params &p = h->p;
    const int nn = h->nn;
    const bool tail = h->patch && !h->iso && !p.has_PT && p.has_moving_mesh;
    if (h->iso) L2(k2_iso_vel, nn, h->d_p, nn, h->bcflag, h->vel);
    else if (!tail) launch_vbcs(h);
    if (p.has_moving_mesh || h->iso) {
        if (!tail) L2(k2_update_coord, 2 * nn, h->d_clk, 2 * nn, h->vel, h->coord);
        // (surf_late: xz_pre keeps the moved top nodes for the next step's k2p_temp_dvoldt<1>)
        if (h->surf_late && h->xz_pre_valid) { h->xz_pre_valid = false; h->surf_pending = true; }
        else launch_surface_commit(h);
    }
    if (h->res_fin_pending) {
        hipLaunchKernelGGL(k2_residual_fin, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->res_count, h->res_part, h->d_clk);
        h->res_fin_pending = false;
    }
    return DES_OK;
}

// the end-of-step element pass rides in the next stress update: a plain step (no compute_dt, which wants this step's
// volumes; no averaging, which wants its rotated stress; no PT loop, which re-enters the passes)
inline bool defer_ok(const Engine *h, bool more)
{
    const des_params &p = h->p;
    const bool rotate = !h->iso && (p.rheol_type & DES_RH_ELASTIC);
    const bool moved = p.has_moving_mesh || h->iso;
    return more && h->patch && moved && rotate && !p.is_outputting_averaged_fields && !p.has_PT
           && (h->steps_host % 10 != 0 || (h->dt_defer_on && !h->halo && h->mass_fuse_on)) && h->geo_on;
}

// Overlapped schedule: this step's exchange may run beside the next step's first passes when nothing else of this step
// is left to do (a deferring step) and the slab has blocks and elements far enough from the cut
inline bool overlap_ok(const Engine *h, bool more)
{
    return h->overlap && h->halo && h->halo_set && h->nb_deep > 0 && h->ne_deep > 0 && defer_ok(h, more);
}

// step_back of such a step: what launch_update_mesh_rest(defer) would do, less everything that reads the ghost region --
// the surface bookkeeping and compute_mass of the blocks near the cut wait behind the join in the next step's front
// (launch_mechanics); compute_mass of the far blocks goes now
void step_back_overlapped(Engine *h)
{
    h->back_steps = h->steps_host;
    refresh_props(h);
    h->geo_pending = true;
    if (h->mass_fuse_on) h->mass_pending = true;           // (compute_mass rides in both parts of the next pass)
    else launch_patch_mass(h, h->d_blist, h->nb_deep);
    h->join_pending = true;
}

// ... and from there on (a decomposed mesh has refreshed its ghost region in between); compute_dt excluded
// `more`: another step of the same call follows at once (nothing reads the element fields in between)
void step_back(Engine *h, bool more = false)
{
    const des_params &p = h->p;
    const int nn = h->nn, ne = h->ne;
    const bool rotate = !h->iso && (p.rheol_type & DES_RH_ELASTIC);
    const bool moved = p.has_moving_mesh || h->iso;
    const bool defer = defer_ok(h, more);
    if (moved) launch_update_mesh_rest(h, h->steps_host, rotate, defer);
    if (h->iso) return;
    if (rotate && !(h->patch && moved))
        L2(k2_rotate, ne, h->d_clk, nn, ne, h->conn, h->coord, h->vel, h->volume, h->stress, h->strain);
    if (p.is_outputting_averaged_fields) {
        const int first = (h->steps_host % p.quality_check_step_interval == 1) ? 1 : 0;
        L2(k2_average, std::max(3 * ne, 2 * nn), h->d_clk, first, 2 * nn, ne, h->coord, h->strain, h->stress, h->delta_plstrain,
           h->coord_avg0, h->strain0, h->stress_avg, h->dplstrain_avg);
    }
}

// ... and so does the surface step (simple_diffusion, edvacc_surf, correct_surface_element): a single engine on the folded
// patch path, on a step that is neither a compute_dt step (it wants max_surf_vel) nor one of the quality-check interval
// (plastic-strain decay, dhacc reset).  `steps`: the number this step will carry.
inline bool surf_late_ok(const Engine *h, bool more, long long steps)
{
    const des_params &p = h->p;
    return more && h->surf_defer_on && h->surf_defer_fits && h->topflag && !h->halo && !h->iso && h->patch && p.has_moving_mesh && (p.rheol_type & DES_RH_ELASTIC)
           && !p.is_outputting_averaged_fields && !p.has_PT && steps % 10 != 0 && steps % p.quality_check_step_interval != 0
           && h->geo_on && h->mass_fuse_on && fold_now(h) && p.surface_process_option == 1 && h->etop > 0 && h->ntop > 0;
}

inline bool elide_ok(const Engine *h, bool more)
{
    return more && h->patch && !h->iso && !h->p.has_PT && !h->p.is_outputting_averaged_fields && h->elide_on;
}

template <class M>
int one_step(Engine *h, bool more = false)
{
    h->elide = elide_ok(h, more);
    h->surf_late = surf_late_ok(h, more, h->steps_host + 1);
    int rc = step_front<M>(h);
    h->surf_late = false;
    if (rc) return rc;
    step_back(h, more);
    if (!h->iso && h->steps_host % 10 == 0) launch_dt(h);
    return DES_OK;
}

void launch_dt_partials(Engine *h)
{
    refresh_props(h);
    if (h->geo_pending) h->rot_prev_dt = true;             // (as launch_dt)
    L2(k2_dt_partials, h->ne, h->d_p, h->dt_part, h->nn, h->ne, h->conn, h->coord, h->vel, h->temperature,
       h->geo_pending ? (const double *)nullptr : h->volume, h->props, h->markers);
    hipLaunchKernelGGL(k2_dt_reduce, dim3(1), dim3(DES2_DT_REDUCE_THREADS), 0, h->stream, h->d_p, h->d_clk, h->dt_part, nblk(h->ne), 0);
}

void launch_pack(Engine *h)
{
    const int nq = h->nnbr, n = h->send_ptr[nq], m = h->esend_ptr[nq];
    if (n + m == 0) return;
    L2(k2_state_pack, n + m, n, h->d_send_idx, h->d_send_noff, m, h->d_esend_idx, h->d_send_eoff, h->nn, h->ne, h->coord, h->vel,
       h->temperature, h->top_pos, h->dh, h->stress, h->strain, h->plstrain, h->stressyy, h->d_sendbuf);
}

// (xs: the engine's stream, or the side stream of the overlapped schedule)
void launch_unpack(Engine *h, hipStream_t xs = nullptr)
{
    const int nq = h->nnbr, n = h->recv_ptr[nq], m = h->erecv_ptr[nq];
    if (n + m == 0) return;
    // (overlapped schedule: the far part of the next step is out already and has swapped the temperature buffers -- the
    //  ghost nodes' temperatures belong into the one that step reads)
    double *const T = h->far_issued ? const_cast<double *>(h->mT_in) : h->temperature;
    hipLaunchKernelGGL(k2_state_unpack, dim3(nblk(n + m)), dim3(DES_BLOCK), 0, xs ? xs : h->stream, n, h->d_recv_idx, h->d_recv_noff, m,
                       h->d_erecv_idx, h->d_recv_eoff, h->nn, h->ne, h->coord, h->vel,
                       T, h->top_pos, h->dh, h->stress, h->strain, h->plstrain, h->stressyy, h->d_recvbuf);
}

void launch_wall_local(Engine *h, hipStream_t xs = nullptr)
{
    hipLaunchKernelGGL(k2_wall_local, dim3(1), dim3(DES_BLOCK), 0, xs ? xs : h->stream, h->nbn[iboundx0], h->bnodes[iboundx0], h->nn, h->coord,
                       wall_needs_zmin(h) ? 1 : 0, h->d_red);
}

void launch_wall_set(Engine *h, const double in[3])
{
    hipLaunchKernelGGL(k2_wall_set, dim3(1), dim3(1), 0, h->stream, h->d_clk, in[0], in[1], in[2], wall_needs_zmin(h) ? 1 : 0);
}

} // namespace

const std::string &last_error(const Engine *h) { return h->err; }

void destroy(Engine *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->xstream) hipStreamSynchronize(h->xstream);
    for (void *q : h->allocs) hipFree(q);
    if (h->h_clk) hipHostFree(h->h_clk);
    if (h->h_red) hipHostFree(h->h_red);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ev_wall) hipEventDestroy(h->ev_wall);
    if (h->xstream) hipStreamDestroy(h->xstream);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

static int create_impl(Engine *h, const des_params *params, const des_mesh *mesh)
{
    const int nn = h->nn = mesh->nnode, ne = h->ne = mesh->nelem, nmat = h->nmat = params->nmat;
    h->o0 = 0; h->o1 = nn; h->nn_global = nn;
    HIP2(hipStreamCreate(&h->stream));
    HIP2(hipStreamCreate(&h->xstream));
    HIP2(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIP2(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIP2(hipEventCreateWithFlags(&h->ev_wall, hipEventDisableTiming));
    { const char *ov = des_env::get("DES_OVERLAP"); h->overlap = ov && ov[0] == '1'; }
    HIP2(hipEventCreate(&h->ev0));
    HIP2(hipEventCreate(&h->ev1));
    HIP2(hipHostMalloc((void **)&h->h_clk, sizeof(Clock)));
    int rc;
#define A2(x) do { rc = (x); if (rc) return rc; } while (0)
    A2(dcopy(h, h->d_p, params, 1));
    {
        // matprops.cxx:237-250
        desk::ViscTerms vt;
        const double gas_constant = 8.3144;
        for (int m = 0; m < DES_MAX_MAT; ++m) vt.pow_edot[m] = vt.coef_term[m] = vt.nR[m] = 0;
        for (int m = 0; m < nmat; ++m) {
            vt.pow_edot[m] = 1 / params->visc_exponent[m] - 1;
            const double pow1 = -1 / params->visc_exponent[m];
            vt.coef_term[m] = std::pow(0.75 * params->visc_coefficient[m], pow1);
            vt.nR[m] = params->visc_exponent[m] * gas_constant;
        }
        A2(dcopy(h, h->d_vt, &vt, 1));
    }
    A2(dalloc(h, h->d_clk, 1));
    A2(dcopy(h, h->conn, mesh->connectivity, (size_t)3 * ne));
    A2(dcopy(h, h->sup_idx, mesh->support_idx, (size_t)nn + 1));
    A2(dcopy(h, h->sup_arr, mesh->support_arr, (size_t)3 * ne));
    A2(dcopy(h, h->sup_lidx, mesh->support_lidx, (size_t)3 * ne));
    A2(dcopy(h, h->bcflag, mesh->bcflag, (size_t)nn));
    for (int i = 0; i < DES_NBDRY; ++i) {
        h->nbf[i] = mesh->nbfacets[i]; h->nbn[i] = mesh->nbnodes[i];
        A2(dcopy(h, h->bf_elem[i], mesh->bfacet_elem[i], (size_t)h->nbf[i]));
        A2(dcopy(h, h->bf_facet[i], mesh->bfacet_facet[i], (size_t)h->nbf[i]));
        A2(dcopy(h, h->bnodes[i], mesh->bnodes[i], (size_t)h->nbn[i]));
    }
    A2(dcopy(h, h->bnormals, mesh->bnormals, (size_t)2 * DES_NBDRY));
    A2(dcopy(h, h->edge_vec, mesh->edge_vec, (size_t)2 * mesh->nedge));
    A2(dcopy(h, h->edge_slot, mesh->edge_slot, (size_t)DES_NBDRY * DES_NBDRY));
    h->ntop = mesh->ntop; h->etop = mesh->etop; h->ntop_elems = mesh->ntop_elems;
    A2(dcopy(h, h->top_nodes, mesh->top_nodes, (size_t)h->ntop));
    A2(dcopy(h, h->ean, mesh->elem_and_nodes, (size_t)2 * h->etop));
    A2(dcopy(h, h->conn_surf, mesh->connectivity_surface, (size_t)3 * h->etop));
    A2(dcopy(h, h->top_elems, mesh->top_elems, (size_t)h->ntop_elems));

    for (double **v : {&h->coord, &h->vel, &h->force, &h->fres, &h->coord0}) A2(dalloc(h, *v, (size_t)2 * nn));
    for (double **v : {&h->temperature, &h->volume_n, &h->mass, &h->tmass, &h->ymass, &h->dhacc, &h->ntmp, &h->total_dx, &h->total_slope})
        A2(dalloc(h, *v, (size_t)nn));
    for (double **v : {&h->stress, &h->strain, &h->strain_rate}) A2(dalloc(h, *v, (size_t)3 * ne));
    for (double **v : {&h->stressyy, &h->plstrain, &h->delta_plstrain, &h->viscosity, &h->volume, &h->volume_old, &h->dpressure,
                       &h->edvoldt, &h->radiogenic, &h->etmp})
        A2(dalloc(h, *v, (size_t)ne));
    A2(dalloc(h, h->tmp_result, (size_t)6 * ne));
    A2(dalloc(h, h->props, (size_t)5 * ne));
    A2(dalloc(h, h->markers, (size_t)ne * nmat));
    A2(dalloc(h, h->mono, (size_t)ne));
    {
        const int n = nmat * DES_PPTAB_CNT * 3;
        A2(dalloc(h, h->pptab, (size_t)n * 5));
        if (h->portable_libm) hipLaunchKernelGGL(k2_pptab<desk::MathPortable>, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d_p, h->pptab);
        else                  hipLaunchKernelGGL(k2_pptab<desk::MathOcml>, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d_p, h->pptab);
    }
    A2(dalloc(h, h->etmp_int, (size_t)ne));
    A2(dalloc(h, h->dh, (size_t)h->ntop));
    A2(dalloc(h, h->edvacc, (size_t)h->etop));
    A2(dalloc(h, h->dt_part, (size_t)5 * std::max(nblk(ne), 1)));
    h->past_cap = (nblk(ne) + 2) * (DES_BLOCK / 64);        // (+ the round-up of a split update's two launches)
    A2(dalloc(h, h->past_part, (size_t)h->past_cap));
    A2(dalloc(h, h->res_part, (size_t)std::max(nblk(nn), nn / 16 + 2)));      // (blocks of 256 owned nodes, or the patch blocks: DES2D_PATCH >= 16)
    A2(dalloc(h, h->neg_zmin, 1));
    A2(dalloc(h, h->d_red, 8));
    A2(build_residual_blocks(h));
    {
        // node-block patches (des_dev2d_patch.hpp): clusters of 128 nodes along a Morton curve (of 64 where 128 do not fit the
        // LDS caps; runs of consecutive ids without coordinates or with DES2D_CLUSTER=0); DES2D_PATCH=<n>: n nodes per block
        const char *env = des_env::get("DES2D_PATCH");
        const char *cl = des_env::get("DES2D_CLUSTER");
        const char *ge = des_env::get("DES2D_GEO"), *ee = des_env::get("DES2D_ELIDE");
        h->geo_on = !(ge && ge[0] == '0'); h->elide_on = !(ee && ee[0] == '0');
        { const char *me = des_env::get("DES2D_MASS_FUSE"); h->mass_fuse_on = !(me && me[0] == '0'); }
        const bool cluster = !(cl && cl[0] == '0');
        Patch2 P;
        bool ok = false;
        if (!(env && env[0] == '0')) {
            const int want = env ? std::atoi(env) : 0;
            if (want >= 16 && want <= DES2_PATCH_THREADS) ok = build_patches2(mesh, want, cluster, P);
            else ok = build_patches2(mesh, 128, cluster, P) || build_patches2(mesh, 64, cluster, P)
                      || (cluster && (build_patches2(mesh, 128, false, P) || build_patches2(mesh, 64, false, P)));
        }
        if (ok) {
            h->patch = true; h->p_npb = P.npb; h->p_nb = P.nb;
            h->p_pn_cap = (P.max_pn + 7) / 8 * 8; h->p_inc_cap = (P.max_inc + 7) / 8 * 8; h->p_pe_cap = (P.max_pe + 7) / 8 * 8;
            if (des_env::get("DES_PATCH_VERBOSE"))
                std::fprintf(stderr, "2-D patches: %d nodes per block, %d blocks, max incidences %d, patch nodes %d, patch elements %d, "
                             "elements listed %zu (%.2f x nelem)\n", P.npb, P.nb, P.max_inc, P.max_pn, P.max_pe, P.pe_pack.size(),
                             (double)P.pe_pack.size() / ne);
            A2(dcopy(h, h->po_ptr, P.po_ptr.data(), P.po_ptr.size()));
            A2(dcopy(h, h->po_id, P.po_id.data(), P.po_id.size()));
            { const std::vector<int> z((size_t)P.nb + 1, 0); A2(dcopy(h, h->pt_zero, z.data(), z.size())); }
            A2(dcopy(h, h->po_slot, P.po_slot.data(), P.po_slot.size()));
            A2(dcopy(h, h->pe_ptr, P.pe_ptr.data(), P.pe_ptr.size()));
            A2(dcopy(h, h->pe_pack, P.pe_pack.data(), P.pe_pack.size()));
            A2(dcopy(h, h->pn_ptr, P.pn_ptr.data(), P.pn_ptr.size()));
            A2(dcopy(h, h->pn_id, P.pn_id.data(), P.pn_id.size()));
            A2(dalloc(h, h->temperature_alt, (size_t)nn));
            h->hp_po_ptr = P.po_ptr; h->hp_po_id = P.po_id; h->hp_pn_ptr = P.pn_ptr; h->hp_pn_id = P.pn_id; h->hp_pe_ptr = P.pe_ptr;
            h->hp_pe_elem.resize(P.pe_pack.size());
            for (size_t q = 0; q < P.pe_pack.size(); ++q) h->hp_pe_elem[q] = (int)(P.pe_pack[q].x & 0x3fffffffull);
            h->h_conn.assign(mesh->connectivity, mesh->connectivity + (size_t)3 * ne);
            h->h_top_nodes.assign(mesh->top_nodes, mesh->top_nodes + h->ntop);
            h->h_top_elems.assign(mesh->top_elems, mesh->top_elems + h->ntop_elems);
            for (int i = 0; i < DES_NBDRY; ++i) {
                if (h->nbf[i] == 0 || h->nbn[i] == 0) continue;
                // k2_sbc_node's walk, once: the marks k2_sbc_facet leaves (the last facet of an element wins), then for every
                // boundary node the elements of its support list that carry a marked facet holding the node
                static const int nof[3][2] = {{1,2},{2,0},{0,1}};
                std::vector<int> bm((size_t)ne, -1);
                for (int f = 0; f < h->nbf[i]; ++f) bm[mesh->bfacet_elem[i][f]] = f;
                std::vector<int4> inc((size_t)DES2_SBC_INC * h->nbn[i], make_int4(-1, 0, 0, 0));
                bool fits = true;
                for (int j = 0; j < h->nbn[i] && fits; ++j) {
                    const int n = mesh->bnodes[i][j];
                    int q = 0;
                    for (int k = mesh->support_idx[n]; k < mesh->support_idx[n + 1]; ++k) {
                        const int e = mesh->support_arr[k], ib = bm[e];
                        if (ib < 0) continue;
                        const int f = mesh->bfacet_facet[i][ib];
                        for (int l = 0; l < 2; ++l)
                            if (n == mesh->connectivity[(size_t)nof[f][l] * ne + e]) {
                                if (q == DES2_SBC_INC) { fits = false; break; }
                                inc[(size_t)DES2_SBC_INC * j + q++] = make_int4(e, f, l, 0);
                                break;
                            }
                    }
                }
                if (fits) A2(dcopy(h, h->binc[i], inc.data(), inc.size()));      // (else: the three-launch form for this boundary)
                if (fits) h->h_binc[i] = inc;
            }
            // k2p_force<1>: the boundary loads of launch_stress_bcs as ONE list per node, in that function's order (boundary by
            // boundary, then the node's incidences); possible when every loaded boundary has its per-node incidences and nothing
            // else follows the loads (no elastic foundation, no Neumann tractions)
            {
                const des_params &p = *params;
                bool ok = true;
                std::vector<std::vector<int4>> per_node((size_t)nn);
                if (p.gravity != 0) {
                    for (int i = 0; i < DES_NBDRY && ok; i++) {
                        if (p.vbc_types[i] != 0 && p.vbc_types[i] != 2 && p.vbc_types[i] != 4) continue;
                        if (i == iboundz0 && !p.has_winkler_foundation) continue;
                        if (i == iboundz1 && !p.has_water_loading) continue;
                        if (h->nbf[i] == 0) continue;
                        if (h->nbn[i] == 0) continue;                         // (launch_stress_bcs: nothing to launch either)
                        if (h->h_binc[i].empty()) { ok = false; break; }
                        for (int j = 0; j < h->nbn[i]; ++j)
                            for (int q = 0; q < DES2_SBC_INC; ++q) {
                                const int4 e = h->h_binc[i][(size_t)DES2_SBC_INC * j + q];
                                if (e.x < 0) break;
                                per_node[mesh->bnodes[i][j]].push_back(make_int4(e.x, e.y, e.z, i));
                            }
                    }
                    if (p.has_elastic_foundation && h->nbn[iboundz0]) ok = false;
                }
                for (int i = 0; i < 6; ++i) if (p.stress_bc_types[i] != 0 && h->nbf[i] != 0) ok = false;
                if (ok) {
                    std::vector<int> idx((size_t)nn + 1, 0);
                    std::vector<int4> ent;
                    for (int n = 0; n < nn; ++n) { idx[n] = (int)ent.size(); ent.insert(ent.end(), per_node[n].begin(), per_node[n].end()); }
                    idx[nn] = (int)ent.size();
                    if (ent.empty()) ent.push_back(make_int4(-1, 0, 0, 0));
                    A2(dcopy(h, h->sbcn_idx, idx.data(), idx.size()));
                    A2(dcopy(h, h->sbcn_ent, ent.data(), ent.size()));
                    A2(dalloc(h, h->coord_alt, (size_t)2 * nn));
                    {
                        std::vector<int> tpos((size_t)nn, -1);
                        for (int i = 0; i < h->ntop; ++i) tpos[mesh->top_nodes[i]] = i;
                        A2(dcopy(h, h->fold_top_pos, tpos.data(), tpos.size()));
                        A2(dalloc(h, h->xz_pre, (size_t)std::max(h->ntop, 1)));
                        std::vector<unsigned char> tf((size_t)ne, 0), bt((size_t)std::max(h->p_nb, 1), 0);
                        for (int i = 0; i < h->ntop_elems; ++i) tf[mesh->top_elems[i]] = 1;
                        for (int b = 0; b < h->p_nb; ++b) {
                            for (int k = h->hp_po_ptr[b]; k < h->hp_po_ptr[b + 1] && !bt[b]; ++k) if (tpos[h->hp_po_id[k]] >= 0) bt[b] = 1;
                            for (int k = h->hp_pn_ptr[b]; k < h->hp_pn_ptr[b + 1] && !bt[b]; ++k) if (tpos[h->hp_pn_id[k]] >= 0) bt[b] = 1;
                        }
                        A2(dcopy(h, h->topflag, tf.data(), tf.size()));
                        // The surface blocks carry extra work (the boundary loads and the top nodes' bookkeeping in k2p_force<1>, the
                        // late surface step in k2p_temp_dvoldt<1>) and the Morton order packs them into two of the eight
                        // contiguous runs desk::logical_block hands the XCDs: those two then finish late (k2p_temp_dvoldt<1>: 70 us
                        // against 59).  Launch order: run k = an eighth of the surface blocks, first, then its share of the others,
                        // both in Morton order.  (Which block does what does not change: results are the same bits.)
                        const char *be = des_env::get("DES2D_TOP_BALANCE");
                        if (!h->halo && h->p_nb >= 64 && !(be && be[0] == '0')) {
                            std::vector<int> T, O, perm;
                            for (int b = 0; b < h->p_nb; ++b) (bt[b] ? T : O).push_back(b);
                            const int per = (h->p_nb + 7) / 8;
                            size_t ot = 0;
                            for (int k = 0; k < 8; ++k) {
                                const int want = std::max(0, std::min(per, h->p_nb - k * per));
                                const size_t t0 = T.size() * k / 8, t1 = T.size() * (k + 1) / 8;
                                int got = 0;
                                for (size_t t = t0; t < t1 && got < want; ++t, ++got) perm.push_back(T[t]);
                                for (; got < want && ot < O.size(); ++got) perm.push_back(O[ot++]);
                            }
                            // (a run too short for its share of the surface blocks: the rest of either list goes to the end -- never in practice)
                            if ((int)perm.size() == h->p_nb) {
                                std::vector<char> seen((size_t)h->p_nb, 0);
                                bool ok = true;
                                for (int b : perm) { if (seen[b]) ok = false; seen[b] = 1; }
                                if (ok) A2(dcopy(h, h->d_bperm, perm.data(), perm.size()));
                            }
                        }
                        // per block: the top nodes of its patch, {staged slot, position in top_nodes, node}
                        {
                            std::vector<int> pp(1, 0);
                            std::vector<int4> pe;
                            bool fits = true;
                            for (int b = 0; b < h->p_nb; ++b) {
                                const int nown = h->hp_po_ptr[b + 1] - h->hp_po_ptr[b];
                                for (int k = h->hp_po_ptr[b]; k < h->hp_po_ptr[b + 1]; ++k)
                                    if (tpos[h->hp_po_id[k]] >= 0) pe.push_back(make_int4(k - h->hp_po_ptr[b], tpos[h->hp_po_id[k]], h->hp_po_id[k], 0));
                                for (int k = h->hp_pn_ptr[b]; k < h->hp_pn_ptr[b + 1]; ++k)
                                    if (tpos[h->hp_pn_id[k]] >= 0) pe.push_back(make_int4(nown + k - h->hp_pn_ptr[b], tpos[h->hp_pn_id[k]], h->hp_pn_id[k], 0));
                                if ((int)pe.size() - pp.back() > DES2_PATCH_THREADS) fits = false;      // (one lane per entry)
                                pp.push_back((int)pe.size());
                            }
                            if (pe.empty()) pe.push_back(make_int4(0, 0, 0, 0));
                            A2(dcopy(h, h->pt_ptr, pp.data(), pp.size()));
                            A2(dcopy(h, h->pt_ent, pe.data(), pe.size()));
                            if (!fits) h->surf_defer_fits = false;
                        }
                        const SurfPre sp = {h->xz_pre, h->pt_ptr, h->pt_ent, h->topflag, h->ntop, h->etop, h->ean, h->conn_surf, h->total_dx,
                                            h->total_slope, h->dhacc, h->dh, h->edvacc, h->plstrain, h->stress, h->strain};
                        A2(dcopy(h, h->d_surfpre, reinterpret_cast<const unsigned char *>(&sp), sizeof(sp)));
                    }
                    h->fold_ok = true;
                }
                const char *fe = des_env::get("DES2D_FOLD"); h->fold_on = !(fe && fe[0] == '0');
                const char *se = des_env::get("DES2D_SURF_DEFER"); h->surf_defer_on = !(se && se[0] == '0');
                const char *de = des_env::get("DES2D_DT_DEFER"); h->dt_defer_on = !(de && de[0] == '0');
                const char *ie = des_env::get("DES2D_PATCH_IT"); h->it3_forced = ie && ie[0] == '3';
                const char *re = des_env::get("DES2D_SR_FUSE"); h->sr_fuse_on = !(re && re[0] == '0');
            }
            A2(dalloc(h, h->stress_pre, (size_t)3 * ne));
        }
    }
    HIP2(hipHostMalloc((void **)&h->h_red, 8 * sizeof(double)));
    if (params->is_outputting_averaged_fields) {
        A2(dalloc(h, h->stress_avg, (size_t)3 * ne)); A2(dalloc(h, h->strain0, (size_t)3 * ne));
        A2(dalloc(h, h->dplstrain_avg, (size_t)ne)); A2(dalloc(h, h->coord_avg0, (size_t)2 * nn));
    }
    {
        std::vector<double> v((size_t)ne, params->visc_max);               // fields.cxx:110
        HIP2(hipMemcpy(h->viscosity, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    L2(k2_fill_int, ne, ne, h->etmp_int, -1);
    HIP2(hipStreamSynchronize(h->stream));
#undef A2
    return DES_OK;
}

Engine *create(int device, const des_params *params, const des_mesh *mesh, int *err, std::string &msg)
{
    *err = DES_OK;
    if (params->ndims != 2) { *err = DES_ERR_UNSUPPORTED_DIM; msg = "not a 2-D model"; return nullptr; }
    if (params->nmat < 1 || params->nmat > DES_MAX_MAT) { *err = DES_ERR_CONFIG_VALUE; msg = "bad nmat"; return nullptr; }
    switch (params->rheol_type) {
    case DES_RH_ELASTIC: case DES_RH_VISCOUS: case DES_RH_MAXWELL: case DES_RH_EP: case DES_RH_EVP: break;
    default: *err = DES_ERR_UNSUPPORTED; msg = "rheology not offloaded"; return nullptr;
    }
    if (params->num_vbc_period_x0 < 1 || params->num_vbc_period_x0 > DES_MAX_PERIOD ||
        params->num_vbc_period_x1 < 1 || params->num_vbc_period_x1 > DES_MAX_PERIOD) {
        *err = DES_ERR_CONFIG_VALUE; msg = "bad num_vbc_period_x?"; return nullptr;
    }
    if (mesh->nelem < 1 || mesh->nnode < 3) { *err = DES_ERR_RESOURCE; msg = "empty mesh"; return nullptr; }
    if (mesh->etop > 0 && mesh->etop != mesh->ntop - 1) {
        // simple_diffusion walks consecutive pairs of the sorted top nodes (bc.cxx:1021-1033)
        *err = DES_ERR_CONFIG_VALUE; msg = "2-D surface: etop must be ntop - 1"; return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) { *err = DES_ERR_UNSUPPORTED; msg = "no such HIP device"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *err = DES_ERR_UNSUPPORTED; msg = "hipSetDevice failed"; return nullptr; }
    Engine *h = new Engine();
    h->device = device;
    h->p = *params;
    const char *env = des_env::get("DES_LIBM");
    h->portable_libm = !env || std::strcmp(env, "portable") == 0;
    if (env && !h->portable_libm && std::strcmp(env, "ocml") != 0) {
        *err = DES_ERR_CONFIG_VALUE; msg = "DES_LIBM must be 'ocml' or 'portable'"; delete h; return nullptr;
    }
    int rc = create_impl(h, params, mesh);
    if (rc) { *err = rc; msg = h->err; destroy(h); return nullptr; }
    return h;
}

long long field_count(const Engine *h, int field)
{
    FieldRef r = field_ref(h, field);
    return r.count;
}

int upload(Engine *h, int field, const void *host, long long count)
{
    FieldRef r = field_ref(h, field);
    if (r.count < 0 || count != r.count) { h->err = "upload: field size mismatch"; return DES_ERR_INTERNAL; }
    HIP2(hipSetDevice(h->device));
    HIP2(hipStreamSynchronize(h->stream));
    if (count) HIP2(hipMemcpy(r.ptr, host, (size_t)count * r.elsize, hipMemcpyHostToDevice));
    if (field == DES_F_ELEMMARKERS) h->markers_dirty = true;
    if (field == DES_F_RADIOGENIC) {
        // every heat source +0.0 (ic.cxx's default)?  then the temperature pass does not fetch them: the same arithmetic on a literal 0.0
        const unsigned long long *b = (const unsigned long long *)host;
        bool z = true;
        for (long long i = 0; i < count && z; ++i) z = b[i] == 0ull;
        h->radiogenic_zero = z;
    }
    return DES_OK;
}

int download(Engine *h, int field, void *host, long long count)
{
    FieldRef r = field_ref(h, field);
    if (r.count < 0 || count != r.count) { h->err = "download: field size mismatch"; return DES_ERR_INTERNAL; }
    HIP2(hipSetDevice(h->device));
    HIP2(hipStreamSynchronize(h->stream));
    if (count) HIP2(hipMemcpy(host, r.ptr, (size_t)count * r.elsize, hipMemcpyDeviceToHost));
    return DES_OK;
}

int set_clock(Engine *h, double dt, double time, long long steps)
{
    HIP2(hipSetDevice(h->device));
    int rc = sync_clock(h);
    if (rc) return rc;
    h->h_clk->dt = dt; h->h_clk->time = time; h->h_clk->steps = steps;
    h->steps_host = steps;
    HIP2(hipMemcpy(h->d_clk, h->h_clk, sizeof(Clock), hipMemcpyHostToDevice));
    return DES_OK;
}

int set_isostasy(Engine *h, int on) { h->iso = on != 0; return DES_OK; }

int sync(Engine *h)
{
    HIP2(hipSetDevice(h->device));
    HIP2(hipStreamSynchronize(h->stream));
    return DES_OK;
}

static int wall_allreduce(Engine *h);
static int dt_allreduce(Engine *h, bool recompute);

// dynearthsol.cxx:184-194 (compute_volume, volume_old = volume, apply_vbcs, compute_mass)
int init_geometry(Engine *h)
{
    HIP2(hipSetDevice(h->device));
    refresh_props(h);
    if (h->comm && h->halo_set) { int rc = wall_allreduce(h); if (rc) return rc; }     // apply_vbcs below reads the whole mesh's wall
    launch_volume_mass(h, false);
    HIP2(hipMemcpyAsync(h->volume_old, h->volume, (size_t)h->ne * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    launch_vbcs(h);
    launch_volume_mass(h, true);
    HIP2(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int compute_dt(Engine *h, double *dt)
{
    HIP2(hipSetDevice(h->device));
    if (h->comm && h->halo_set) { int rcd = dt_allreduce(h, true); if (rcd) return rcd; }
    else launch_dt(h);
    int rc = sync_clock(h);
    if (rc) return rc;
    if (dt) *dt = h->h_clk->dt;
    return h->h_clk->dt > 0 ? DES_OK : DES_ERR_RUNTIME_NAN;
}

static int fill_scalars(Engine *h, des_scalars *out)
{
    int rc = sync_clock(h);
    if (rc) return rc;
    const Clock &c = *h->h_clk;
    out->dt = c.dt; out->time = c.time; out->l2_residual = c.l2_residual; out->max_surf_vel = c.max_surf_vel;
    out->max_global_vel_mag = c.max_global_vel_mag; out->global_dt_min = c.global_dt_min; out->steps = c.steps;
    if (h->past_used) {
        // des_scalars::n_return_mapping of the call's last update_stress: the wavefronts' counts (k2_stress), added here
        h->past_host.resize((size_t)h->past_cap);
        if (hipMemcpy(h->past_host.data(), h->past_part, (size_t)h->past_cap * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return DES_ERR_RESOURCE;
        long long n = 0;
        for (int v : h->past_host) n += v;
        h->n_past_host = (int)n;
        h->past_used = false;
    }
    out->status = c.status; out->n_return_mapping = h->n_past_host; out->avg_time0 = c.avg_time0; out->n_pt_iterations = h->n_pt_iterations;
    return c.status;
}

// ---- the decomposed step on RCCL (set_comm): everything on the engine's stream, no host synchronisation ----------------
#define NCCL2(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    h->err = std::string("RCCL: ") + ncclGetErrorString(r_); return DES_ERR_RESOURCE; } } while (0)

// the x0 wall's extent / the lowest node across ranks into the clock (bc.cxx:251-300, 350-361)
static int wall_allreduce(Engine *h)
{
    launch_wall_local(h);
    NCCL2(ncclAllReduce(h->d_red, h->d_red, 3, ncclDouble, ncclMax, h->comm, h->stream));
    hipLaunchKernelGGL(k2_wall_set_from, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red, wall_needs_zmin(h) ? 1 : 0);
    return DES_OK;
}

// one grouped send / recv per neighbour between the pack and the unpack launch, then the wall reduction
static int exchange_rccl(Engine *h)
{
    Prof2 pr(h, P2_EXCH);
    launch_pack(h);
    NCCL2(ncclGroupStart());
    for (int q = 0; q < h->nnbr; ++q) {
        NCCL2(ncclSend(h->d_sendbuf + h->send_off[q], (size_t)(h->send_off[q+1] - h->send_off[q]), ncclDouble, h->nbr_rank[q], h->comm, h->stream));
        NCCL2(ncclRecv(h->d_recvbuf + h->recv_off[q], (size_t)(h->recv_off[q+1] - h->recv_off[q]), ncclDouble, h->nbr_rank[q], h->comm, h->stream));
    }
    NCCL2(ncclGroupEnd());
    launch_unpack(h);
    return wall_allreduce(h);
}

// Overlapped schedule: the pack on the engine's stream (it reads what the next step's stress update is about to change),
// then transfer, unpack and the wall's reduction on the side stream; ev_join: the ghost region is refreshed, ev_wall: the
// wall's extent is in the clock.  launch_mechanics / join_wall make the engine's stream wait for them.
static int exchange_pack_fork(Engine *h)
{
    launch_pack(h);
    HIP2(hipEventRecord(h->ev_fork, h->stream));
    return DES_OK;
}
static int exchange_side(Engine *h)
{
    HIP2(hipStreamWaitEvent(h->xstream, h->ev_fork, 0));
    const hipStream_t xs = h->xstream;
    {
        Prof2 pr(h, P2_EXCH, xs);
        NCCL2(ncclGroupStart());
        for (int q = 0; q < h->nnbr; ++q) {
            NCCL2(ncclSend(h->d_sendbuf + h->send_off[q], (size_t)(h->send_off[q+1] - h->send_off[q]), ncclDouble, h->nbr_rank[q], h->comm, xs));
            NCCL2(ncclRecv(h->d_recvbuf + h->recv_off[q], (size_t)(h->recv_off[q+1] - h->recv_off[q]), ncclDouble, h->nbr_rank[q], h->comm, xs));
        }
        NCCL2(ncclGroupEnd());
        launch_unpack(h, xs);
        HIP2(hipEventRecord(h->ev_join, xs));
        launch_wall_local(h, xs);
        NCCL2(ncclAllReduce(h->d_red, h->d_red, 3, ncclDouble, ncclMax, h->comm, xs));
        hipLaunchKernelGGL(k2_wall_set_from, dim3(1), dim3(1), 0, xs, h->d_clk, h->d_red, wall_needs_zmin(h) ? 1 : 0);
    }
    HIP2(hipEventRecord(h->ev_wall, xs));
    h->wall_pending = true;
    return DES_OK;
}

// a step that fails half-way must not leave the engine between two pieces of the overlapped schedule
static int step_abort(Engine *h, int rc)
{
    hipStreamSynchronize(h->xstream);
    hipStreamSynchronize(h->stream);
    h->join_pending = false; h->wall_pending = false; h->far_issued = false; h->mass_pending = false; h->surf_pending = false; h->rot_prev_dt = false;
    static const int zero = 0;
    hipMemcpy(&h->d_clk->pt, &zero, sizeof(int), hipMemcpyHostToDevice);     // a loop the error may have come from
    h->no_neumann = false;
    return rc;
}

static int dt_allreduce(Engine *h, bool recompute)
{
    if (recompute) launch_dt_partials(h);
    hipLaunchKernelGGL(k2_dt_pack, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red);
    NCCL2(ncclAllReduce(h->d_red, h->d_red, 6, ncclDouble, ncclMin, h->comm, h->stream));
    hipLaunchKernelGGL(k2_dt_unpack, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red);
    hipLaunchKernelGGL(k2_dt_finalize, dim3(1), dim3(1), 0, h->stream, h->d_p, h->d_clk);
    return DES_OK;
}

int set_comm(Engine *h, void *comm)
{
    h->comm = (ncclComm_t)comm;
    return DES_OK;
}

// des_dev_comm_selfcheck for a 2-D engine (engine/selfcheck.hpp)
int comm_selfcheck(Engine *h, int expect_world, int expect_rank)
{
    HIP2(hipSetDevice(h->device));
    HIP2(hipStreamSynchronize(h->stream));
    const std::string bad = des_selfcheck::run(h->comm, h->stream, expect_world, expect_rank, h->halo_set ? h->nnbr : 0, h->nbr_rank.data(),
                                               h->send_off.data(), h->recv_off.data(), h->d_sendbuf, h->d_recvbuf, h->d_red);
    if (!bad.empty()) { h->err = "RCCL self-check, rank " + std::to_string(expect_rank) + ": " + bad; return DES_ERR_RESOURCE; }
    return DES_OK;
}

int step(Engine *h, int nsteps, des_scalars *out)
{
    HIP2(hipSetDevice(h->device));
    if (h->comm && h->halo_set) {
        h->n_pt_iterations = 0;
        for (int i = 0; i < nsteps; ++i) {
            h->count_past = (i == nsteps - 1);
            h->elide = elide_ok(h, i < nsteps - 1);
            int rc = h->portable_libm ? step_front<desk::MathPortable>(h) : step_front<desk::MathOcml>(h);
            if (rc) return step_abort(h, rc);
            if (overlap_ok(h, i < nsteps - 1)) {
                // the messages packed; then, BEFORE the host spends its time in RCCL, everything the device can do meanwhile:
                // compute_mass of the far blocks and the far part of the next step's first two passes
                if ((rc = exchange_pack_fork(h))) return step_abort(h, rc);
                step_back_overlapped(h);
                h->count_past = (i + 1 == nsteps - 1);
                h->elide = elide_ok(h, i + 1 < nsteps - 1);
                if (h->portable_libm) front_begin<desk::MathPortable>(h); else front_begin<desk::MathOcml>(h);
                if ((rc = exchange_side(h))) return step_abort(h, rc);
                continue;
            }
            if ((rc = exchange_rccl(h))) return step_abort(h, rc);
            step_back(h, i < nsteps - 1);
            if (!h->iso && h->steps_host % 10 == 0 && (rc = dt_allreduce(h, true))) return step_abort(h, rc);
        }
        HIP2(hipGetLastError());
        if (!out) return DES_OK;
        // l2_residual is a sum over all ranks' owned nodes (des_dev.h): the partial sums are added before the root
        HIP2(hipMemcpyAsync(h->d_red + 6, &h->d_clk->l2_sum, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        NCCL2(ncclAllReduce(h->d_red + 6, h->d_red + 6, 1, ncclDouble, ncclSum, h->comm, h->stream));
        const int rcs = fill_scalars(h, out);
        double l2sum = 0;
        HIP2(hipMemcpy(&l2sum, h->d_red + 6, sizeof(double), hipMemcpyDeviceToHost));
        // (a step that ended in the pseudo-transient loop left the global block sum on every rank: nothing to add)
        if (!h->h_clk->l2_global) out->l2_residual = std::sqrt(l2sum);
        return rcs;
    }
    if (h->halo && nsteps > 0) {
        h->err = "a decomposed 2-D engine steps through des_dev_step (after des_dev_comm_init), des_dev_step_group, or des_dev_phase + the exchange entry points";
        return DES_ERR_UNSUPPORTED;
    }
    h->n_pt_iterations = 0;
    for (int i = 0; i < nsteps; ++i) {
        h->count_past = (i == nsteps - 1);
        const int rc = h->portable_libm ? one_step<desk::MathPortable>(h, i < nsteps - 1) : one_step<desk::MathOcml>(h, i < nsteps - 1);
        if (rc) return rc;
    }
    HIP2(hipGetLastError());
    if (out) return fill_scalars(h, out);
    return DES_OK;
}

// initial_body_force_adjustment (dynearthsol.cxx:546-591; main() calls it once before the time loop when
// ic.has_body_force_adjustment): the residual of the force_residual the engine holds, then -- with control.has_PT -- the
// pseudo-transient loop on the initial state with the Neumann tractions held back (fields.cxx:690).
int body_force_adjustment(Engine *h, des_scalars *out)
{
    HIP2(hipSetDevice(h->device));
    if (h->halo && !h->comm) { h->err = "the initial body-force adjustment on a decomposed 2-D mesh: through RCCL (des_dev_comm_init)"; return DES_ERR_UNSUPPORTED; }
    refresh_props(h);
    h->n_pt_iterations = 0;
    if (h->p.has_PT) {
        h->no_neumann = true;
        int rc = h->portable_libm ? pt_loop<desk::MathPortable>(h) : pt_loop<desk::MathOcml>(h);
        h->no_neumann = false;
        if (!rc && h->halo) rc = exchange_rccl(h);             // the ghost region as the last iteration left the owners
        if (rc) return step_abort(h, rc);
    } else { const int rc = residual_global(h); if (rc) return rc; }
    HIP2(hipGetLastError());
    if (out) return fill_scalars(h, out);
    return sync_clock(h);
}

// ---- domain decomposition (des_dev.h: des_dev_set_halo ... des_dev_step_group) ------------------------------------
// The 2-D model is cut like the 3-D one (host/partition.cpp: node slabs along x in the reference's renumbered
// order + four element layers of ghost region); one exchange per step after the surface heights are committed.
// Beside the ghost records two small reductions cross the ranks: the x0 wall's vertical extent apply_vbcs scales its
// velocity profiles with (every step, MAX of three), and compute_dt's minima (every 10th step).
// The split of the overlapped schedule.  A block is FAR from the cut when no node of its patch (its own and the others) is
// written by the unpack -- a node of a receive list or any node outside the owned range -- or lies on the surface, and no
// element of its patch is received or belongs to the surface elements (correct_surface_element rewrites their volume,
// stress and strain in the bookkeeping that waits behind the join); an element when its three nodes belong to far blocks
// and it is neither received nor a surface element.  What k2p_mass, k2p_temp_dvoldt and k2_stress read and write for
// those is then disjoint from everything the side stream and the postponed bookkeeping touch.  d_blist / d_elist: the far
// ones first (ascending), then the others.
static int build_deep_lists(Engine *h, const des_halo *halo)
{
    h->nb_deep = 0; h->ne_deep = 0;
    if (!h->patch || halo->nnbr == 0) return DES_OK;
    const int nn = h->nn, ne = h->ne, nb = h->p_nb, nq = halo->nnbr;
    std::vector<char> wn((size_t)nn, 0), we((size_t)ne, 0);
    for (int n = 0; n < nn; ++n) if (n < h->o0 || n >= h->o1) wn[n] = 1;
    for (int k = 0; k < halo->recv_ptr[nq]; ++k) wn[halo->recv_idx[k]] = 1;
    for (int k = 0; k < halo->erecv_ptr[nq]; ++k) we[halo->erecv_idx[k]] = 1;
    for (int n : h->h_top_nodes) wn[n] = 1;
    for (int e : h->h_top_elems) we[e] = 1;
    std::vector<char> far_b((size_t)nb, 1);
    std::vector<int> blk_of((size_t)nn, 0);
    for (int b = 0; b < nb; ++b) {
        for (int k = h->hp_po_ptr[b]; k < h->hp_po_ptr[b + 1]; ++k) { blk_of[h->hp_po_id[k]] = b; if (wn[h->hp_po_id[k]]) far_b[b] = 0; }
        for (int k = h->hp_pn_ptr[b]; k < h->hp_pn_ptr[b + 1] && far_b[b]; ++k) if (wn[h->hp_pn_id[k]]) far_b[b] = 0;
        for (int k = h->hp_pe_ptr[b]; k < h->hp_pe_ptr[b + 1] && far_b[b]; ++k) if (we[h->hp_pe_elem[k]]) far_b[b] = 0;
    }
    std::vector<int> blist, elist;
    blist.reserve((size_t)nb); elist.reserve((size_t)ne);
    for (int b = 0; b < nb; ++b) if (far_b[b]) blist.push_back(b);
    const int nb_deep = (int)blist.size();
    for (int b = 0; b < nb; ++b) if (!far_b[b]) blist.push_back(b);
    std::vector<char> far_e((size_t)ne, 0);
    for (int e = 0; e < ne; ++e) {
        if (we[e]) continue;
        bool far = true;
        for (int i = 0; i < 3; ++i) far = far && far_b[blk_of[h->h_conn[(size_t)i * ne + e]]];
        far_e[e] = far;
    }
    for (int e = 0; e < ne; ++e) if (far_e[e]) elist.push_back(e);
    const int ne_deep = (int)elist.size();
    for (int e = 0; e < ne; ++e) if (!far_e[e]) elist.push_back(e);
    int rc;
    if ((rc = dcopy(h, h->d_blist, blist.data(), blist.size())) || (rc = dcopy(h, h->d_elist, elist.data(), elist.size()))) return rc;
    h->nb_deep = nb_deep; h->ne_deep = ne_deep;
    if (des_env::get("DES_PATCH_VERBOSE"))
        std::fprintf(stderr, "2-D overlapped schedule: %d of %d blocks and %d of %d elements far from the cut\n", nb_deep, nb, ne_deep, ne);
    return DES_OK;
}

// des_dev_set_overlap / des_dev_comm_info for a 2-D engine
int set_overlap(Engine *h, int on)
{
    if (h->join_pending || h->wall_pending) { h->err = "des_dev_set_overlap inside a step"; return DES_ERR_INTERNAL; }
    h->overlap = on != 0;
    return DES_OK;
}
int overlapped(const Engine *h) { return h->overlap && h->halo && h->nb_deep > 0 && h->ne_deep > 0; }

int set_halo(Engine *h, const des_halo *halo, int nn_global)
{
    HIP2(hipSetDevice(h->device));
    if (halo->owned_begin < 0 || halo->owned_end > h->nn || halo->owned_begin >= halo->owned_end) return DES_ERR_INTERNAL;
    h->o0 = halo->owned_begin; h->o1 = halo->owned_end; h->nn_global = nn_global;
    h->g0 = (halo->nnbr > 0 || halo->owned_begin > 0 || halo->owned_end < h->nn) ? halo->owned_global_begin : 0;
    { const int rcb = build_residual_blocks(h); if (rcb) return rcb; }
    const int nq = h->nnbr = halo->nnbr;
    h->halo_set = true; h->halo = nq > 0;
    h->nbr_rank.assign(halo->nbr_rank, halo->nbr_rank + nq);
    h->send_ptr.assign(halo->send_ptr, halo->send_ptr + nq + 1);
    h->recv_ptr.assign(halo->recv_ptr, halo->recv_ptr + nq + 1);
    h->esend_ptr.assign(halo->esend_ptr, halo->esend_ptr + nq + 1);
    h->erecv_ptr.assign(halo->erecv_ptr, halo->erecv_ptr + nq + 1);
    auto layout = [&](const std::vector<int> &np, const std::vector<int> &ep, std::vector<long long> &off,
                      std::vector<int> &noff, std::vector<int> &eoff) {
        off.assign((size_t)nq + 1, 0);
        noff.resize((size_t)np[nq]); eoff.resize((size_t)ep[nq]);
        for (int q = 0; q < nq; ++q) {
            long long base = off[q];
            for (int k = np[q]; k < np[q+1]; ++k) noff[k] = (int)(base + (long long)(k - np[q]) * DES_X_NODE_WIDTH_2D);
            base += (long long)(np[q+1] - np[q]) * DES_X_NODE_WIDTH_2D;
            for (int k = ep[q]; k < ep[q+1]; ++k) eoff[k] = (int)(base + (long long)(k - ep[q]) * DES_X_ELEM_WIDTH_2D);
            off[q+1] = base + (long long)(ep[q+1] - ep[q]) * DES_X_ELEM_WIDTH_2D;
        }
    };
    std::vector<int> snoff, seoff, rnoff, reoff;
    layout(h->send_ptr, h->esend_ptr, h->send_off, snoff, seoff);
    layout(h->recv_ptr, h->erecv_ptr, h->recv_off, rnoff, reoff);
    int rc;
#define A2(x) do { rc = (x); if (rc) return rc; } while (0)
    A2(dcopy(h, h->d_send_idx, halo->send_idx, (size_t)h->send_ptr[nq]));
    A2(dcopy(h, h->d_recv_idx, halo->recv_idx, (size_t)h->recv_ptr[nq]));
    A2(dcopy(h, h->d_esend_idx, halo->esend_idx, (size_t)h->esend_ptr[nq]));
    A2(dcopy(h, h->d_erecv_idx, halo->erecv_idx, (size_t)h->erecv_ptr[nq]));
    A2(dcopy(h, h->d_send_noff, snoff.data(), snoff.size()));
    A2(dcopy(h, h->d_send_eoff, seoff.data(), seoff.size()));
    A2(dcopy(h, h->d_recv_noff, rnoff.data(), rnoff.size()));
    A2(dcopy(h, h->d_recv_eoff, reoff.data(), reoff.size()));
    A2(dalloc(h, h->d_sendbuf, (size_t)h->send_off[nq]));
    A2(dalloc(h, h->d_recvbuf, (size_t)h->recv_off[nq]));
    {
        std::vector<int> top((size_t)h->ntop), pos((size_t)h->nn, -1);
        if (h->ntop) HIP2(hipMemcpy(top.data(), h->top_nodes, top.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < h->ntop; ++i) pos[top[i]] = i;
        A2(dcopy(h, h->top_pos, pos.data(), pos.size()));
    }
    A2(build_deep_lists(h, halo));
#undef A2
    return DES_OK;
}

// One phase of a step without any communication (the caller moves the ghost records and reduces the wall extent in
// between).  Returns 1 after phase 1 when this is a compute_dt step (the partials are in the clock), < 0: -error.
// the partition-independent residual for a caller that moves the data itself (des_dev.h: des_dev_residual_blocks / _set)
int residual_blocks(Engine *h, double *out, int cap, int *first, int *count)
{
    HIP2(hipSetDevice(h->device));
    const int B = des_res_block(h->nn_global), nb_own = (h->o1 - h->o0 + B - 1) / B;
    if (first) *first = h->g0 / B;
    if (count) *count = nb_own;
    if (!out) return DES_OK;
    if (cap < nb_own) return DES_ERR_INTERNAL;
    launch_residual_blocks(h);
    HIP2(hipMemcpyAsync(out, h->res_blocks + h->g0 / B, (size_t)nb_own * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP2(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int residual_set(Engine *h, const double *blocks, int nblocks, double *l2)
{
    HIP2(hipSetDevice(h->device));
    if (nblocks != h->res_nb_global) { h->err = "des_dev_residual_set: not the global block count"; return DES_ERR_INTERNAL; }
    HIP2(hipMemcpyAsync(h->res_blocks, blocks, (size_t)nblocks * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP2(hipStreamSynchronize(h->stream));
    hipLaunchKernelGGL(k2_residual_final, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->d_clk, h->res_blocks, h->res_nb_global);
    const int rc = sync_clock(h);
    if (rc) return rc;
    if (l2) *l2 = h->h_clk->l2_residual;
    return DES_OK;
}

int phase(Engine *h, int ph)
{
    if (hipSetDevice(h->device) != hipSuccess) return -DES_ERR_RESOURCE;
    if (ph == 0) {
        h->count_past = true;
        h->elide = false;
        // control.has_PT: the loop is the caller's (des_dev.h) -- the front stops in front of it and phase 0 returns 2
        const bool pt = h->p.has_PT && !h->iso;
        h->pt_defer = pt;
        const int rc = h->portable_libm ? step_front<desk::MathPortable>(h) : step_front<desk::MathOcml>(h);
        h->pt_defer = false;
        if (rc) return -rc;
        if (!pt) return 0;
        h->n_pt_iterations = 0;
        return set_pt(h, 1) ? -DES_ERR_RESOURCE : 2;
    }
    if (ph == 2) {
        if (h->portable_libm) pt_iteration<desk::MathPortable>(h); else pt_iteration<desk::MathOcml>(h);
        ++h->n_pt_iterations;
        return 0;
    }
    if (ph == 3) {
        if (set_pt(h, 0)) return -DES_ERR_RESOURCE;
        const int rc = step_front_rest(h);
        return rc ? -rc : 0;
    }
    if (ph != 1) return -DES_ERR_INTERNAL;
    step_back(h);
    if (h->iso || h->steps_host % 10 != 0) return 0;
    launch_dt_partials(h);
    return 1;
}

// state records of the listed local ids to / from a host buffer (what = 0: nodes, 1: elements)
static int state_io(Engine *h, int what, const int *idx, int n, double *buf, bool pack)
{
    if (what < 0 || what > 1) return DES_ERR_INTERNAL;
    HIP2(hipSetDevice(h->device));
    if (n == 0) return DES_OK;
    if (!h->top_pos) { h->err = "des_dev_halo_pack / unpack before des_dev_set_halo"; return DES_ERR_INTERNAL; }
    const size_t w = what == 0 ? DES_X_NODE_WIDTH_2D : DES_X_ELEM_WIDTH_2D;
    std::vector<int> off((size_t)n);
    for (int k = 0; k < n; ++k) off[k] = (int)(k * w);
    int *d_idx = nullptr, *d_off = nullptr; double *d_buf = nullptr;
    HIP2(hipMalloc((void **)&d_idx, (size_t)n * sizeof(int)));
    HIP2(hipMalloc((void **)&d_off, (size_t)n * sizeof(int)));
    HIP2(hipMalloc((void **)&d_buf, (size_t)n * w * sizeof(double)));
    HIP2(hipMemcpyAsync(d_idx, idx, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP2(hipMemcpyAsync(d_off, off.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    const int n_nodes = what == 0 ? n : 0, n_elems = what == 0 ? 0 : n;
    if (pack) {
        L2(k2_state_pack, n, n_nodes, d_idx, d_off, n_elems, d_idx, d_off, h->nn, h->ne, h->coord, h->vel, h->temperature,
           h->top_pos, h->dh, h->stress, h->strain, h->plstrain, h->stressyy, d_buf);
        HIP2(hipMemcpyAsync(buf, d_buf, (size_t)n * w * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else {
        HIP2(hipMemcpyAsync(d_buf, buf, (size_t)n * w * sizeof(double), hipMemcpyHostToDevice, h->stream));
        L2(k2_state_unpack, n, n_nodes, d_idx, d_off, n_elems, d_idx, d_off, h->nn, h->ne, h->coord, h->vel, h->temperature,
           h->top_pos, h->dh, h->stress, h->strain, h->plstrain, h->stressyy, d_buf);
    }
    HIP2(hipStreamSynchronize(h->stream));
    hipFree(d_idx); hipFree(d_off); hipFree(d_buf);
    return DES_OK;
}

int halo_pack(Engine *h, int what, const int *idx, int n, double *buf) { return state_io(h, what, idx, n, buf, true); }
int halo_unpack(Engine *h, int what, const int *idx, int n, const double *buf) { return state_io(h, what, idx, n, const_cast<double *>(buf), false); }

int wall_get(Engine *h, double out[3])
{
    HIP2(hipSetDevice(h->device));
    launch_wall_local(h);
    HIP2(hipMemcpyAsync(out, h->d_red, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP2(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int wall_set(Engine *h, const double in[3])
{
    HIP2(hipSetDevice(h->device));
    launch_wall_set(h, in);
    return DES_OK;
}

int dt_partials(Engine *h, double out[6], int recompute)
{
    HIP2(hipSetDevice(h->device));
    if (recompute) launch_dt_partials(h);
    hipLaunchKernelGGL(k2_dt_pack, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red);
    HIP2(hipMemcpyAsync(out, h->d_red, 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP2(hipStreamSynchronize(h->stream));
    return DES_OK;
}

int dt_finalize(Engine *h, const double in[6], double *dt)
{
    HIP2(hipSetDevice(h->device));
    HIP2(hipMemcpyAsync(h->d_red, in, 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k2_dt_unpack, dim3(1), dim3(1), 0, h->stream, h->d_clk, h->d_red);
    hipLaunchKernelGGL(k2_dt_finalize, dim3(1), dim3(1), 0, h->stream, h->d_p, h->d_clk);
    int rc = sync_clock(h);
    if (rc) return rc;
    if (dt) *dt = h->h_clk->dt;
    return h->h_clk->status;
}

// Several engines of ONE process as the ranks of one decomposed model (des_dev_step_group): ghost records by
// device-to-device copies between the message buffers; the two reductions through the host, which joins every
// engine's stream for them (the 2-D models are the small ones: simplicity over overlap here).
static int step_group_impl(Engine **g, int n, int nsteps, des_scalars *out);

// (a call that fails half-way must not leave an engine between two pieces of the overlapped schedule: step_abort)
int step_group(Engine **g, int n, int nsteps, des_scalars *out)
{
    const int rc = step_group_impl(g, n, nsteps, out);
    if (rc) for (int k = 0; k < n; ++k) if (g[k]) { hipSetDevice(g[k]->device); step_abort(g[k], rc); }
    return rc;
}

static int step_group_impl(Engine **g, int n, int nsteps, des_scalars *out)
{
    for (int k = 0; k < n; ++k) {
        Engine *h = g[k];
        if (!h->halo_set) { h->err = "des_dev_step_group: des_dev_set_halo first"; return DES_ERR_INTERNAL; }
        for (int q = 0; q < h->nnbr; ++q) {
            const int r = h->nbr_rank[q];
            int qo = -1;
            if (r >= 0 && r < n && r != k) for (int j = 0; j < g[r]->nnbr; ++j) if (g[r]->nbr_rank[j] == k) qo = j;
            if (qo < 0 || g[r]->send_off[qo + 1] - g[r]->send_off[qo] != h->recv_off[q + 1] - h->recv_off[q]) {
                h->err = "group: the exchange lists of two neighbours do not match"; return DES_ERR_INTERNAL;
            }
        }
        h->n_pt_iterations = 0;
    }
    Engine *h = g[0];                               // (HIP2 books errors on rank 0)
    HIP2(hipSetDevice(h->device));
    for (int i = 0; i < nsteps; ++i) {
        // the ghost region (and the wall's extent) of every engine refreshed from its neighbours' packed messages
        // engine k takes its neighbours' packed messages (device-to-device copies), unpacks them and forms its local wall
        // extent, all on stream xs (its own, or its side stream on the overlapped schedule)
        auto take_messages = [&](int k, hipStream_t xs) -> int {
            Engine *e = g[k];
            for (int q = 0; q < e->nnbr; ++q) {
                Engine *o = g[e->nbr_rank[q]];
                int qo = 0;
                for (int j = 0; j < o->nnbr; ++j) if (o->nbr_rank[j] == k) qo = j;
                const long long len = e->recv_off[q + 1] - e->recv_off[q];
                if (len) HIP2(hipMemcpyAsync(e->d_recvbuf + e->recv_off[q], o->d_sendbuf + o->send_off[qo], (size_t)len * sizeof(double),
                                             hipMemcpyDeviceToDevice, xs));
            }
            launch_unpack(e, xs);
            if (xs != e->stream) HIP2(hipEventRecord(e->ev_join, xs));
            launch_wall_local(e, xs);
            HIP2(hipMemcpyAsync(e->h_red, e->d_red, 3 * sizeof(double), hipMemcpyDeviceToHost, xs));
            return DES_OK;
        };
        auto exchange_all = [&](bool set_wall) -> int {
            for (int k = 0; k < n; ++k) launch_pack(g[k]);
            for (int k = 0; k < n; ++k) HIP2(hipStreamSynchronize(g[k]->stream));
            for (int k = 0; k < n; ++k) { const int rc = take_messages(k, g[k]->stream); if (rc) return rc; }
            double wall[3] = {-DBL_MAX, -DBL_MAX, 0.0};
            for (int k = 0; k < n; ++k) {
                HIP2(hipStreamSynchronize(g[k]->stream));
                for (int q = 0; q < 3; ++q) wall[q] = std::max(wall[q], g[k]->h_red[q]);
            }
            if (set_wall) for (int k = 0; k < n; ++k) launch_wall_set(g[k], wall);
            return DES_OK;
        };
        // the partition-independent residual of all engines: the parts through the host, the fixed-shape sum on each
        auto residual_all = [&](double *l2) -> int {
            std::vector<double> all((size_t)g[0]->res_nb_global, 0.0);
            for (int k = 0; k < n; ++k) {
                Engine *e = g[k];
                const int B = des_res_block(e->nn_global), nb_own = (e->o1 - e->o0 + B - 1) / B;
                launch_residual_blocks(e);
                HIP2(hipMemcpyAsync(all.data() + e->g0 / B, e->res_blocks + e->g0 / B, (size_t)nb_own * sizeof(double), hipMemcpyDeviceToHost, e->stream));
                HIP2(hipStreamSynchronize(e->stream));
            }
            for (int k = 0; k < n; ++k) {
                Engine *e = g[k];
                HIP2(hipMemcpyAsync(e->res_blocks, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
                HIP2(hipStreamSynchronize(e->stream));
                hipLaunchKernelGGL(k2_residual_final, dim3(1), dim3(DES_BLOCK), 0, e->stream, e->d_clk, e->res_blocks, e->res_nb_global);
                const int rc = sync_clock(e);
                if (rc) return rc;
            }
            *l2 = g[0]->h_clk->l2_residual;
            return DES_OK;
        };
        const bool pt = g[0]->p.has_PT && !g[0]->iso;
        for (int k = 0; k < n; ++k) {
            g[k]->count_past = (i == nsteps - 1);
            g[k]->elide = elide_ok(g[k], i < nsteps - 1);
            g[k]->pt_defer = pt;
            const int rc = g[k]->portable_libm ? step_front<desk::MathPortable>(g[k]) : step_front<desk::MathOcml>(g[k]);
            g[k]->pt_defer = false;
            if (rc) return rc;
        }
        if (pt) {
            // control.has_PT: the fronts have stopped in front of the loop; all engines in lockstep -- a ghost refresh and one
            // residual per iteration -- then the rest of every front
            int rc;
            double residual_old = 0, l2 = 0;
            if ((rc = residual_all(&residual_old))) return rc;
            auto pt_all = [&](int on) -> int { int first = DES_OK; for (int k = 0; k < n; ++k) { const int r = set_pt(g[k], on); if (r && !first) first = r; } return first; };
            if ((rc = pt_all(1))) { pt_all(0); return rc; }
            rc = [&]() -> int {
                int r;
                for (int pt_step = 0; pt_step < g[0]->p.PT_max_iter; ++pt_step) {
                    if ((r = exchange_all(true))) return r;
                    for (int k = 0; k < n; ++k) { if (g[k]->portable_libm) pt_iteration<desk::MathPortable>(g[k]); else pt_iteration<desk::MathOcml>(g[k]); }
                    if ((r = residual_all(&l2))) return r;
                    for (int k = 0; k < n; ++k) ++g[k]->n_pt_iterations;
                    if (std::fabs((l2 - residual_old) / residual_old) < g[0]->p.PT_relative_tolerance) break;
                    residual_old = l2;
                }
                return DES_OK;
            }();
            const int rc_off = pt_all(0);                     // on every exit of the loop
            if (rc || (rc = rc_off)) return rc;
            for (int k = 0; k < n; ++k) if ((rc = step_front_rest(g[k]))) return rc;
        }
        // Overlapped schedule (every engine takes the same decision): messages, unpack and the wall's local extent on the side
        // streams; compute_mass of the far blocks on the engines' streams at once, the rest of the step behind the join in the
        // next step's front -- the launches of a rank on RCCL in the same order, the transfer a device-to-device copy
        bool ov = true;
        for (int k = 0; k < n; ++k) ov = ov && overlap_ok(g[k], i < nsteps - 1);
        if (ov) {
            for (int k = 0; k < n; ++k) launch_pack(g[k]);
            for (int k = 0; k < n; ++k) HIP2(hipStreamSynchronize(g[k]->stream));
            for (int k = 0; k < n; ++k) {
                Engine *e = g[k];
                { const int rc = take_messages(k, e->xstream); if (rc) return rc; }
                step_back_overlapped(e);
                e->count_past = (i + 1 == nsteps - 1);
                e->elide = elide_ok(e, i + 1 < nsteps - 1);
                if (e->portable_libm) front_begin<desk::MathPortable>(e); else front_begin<desk::MathOcml>(e);
            }
            double wall[3] = {-DBL_MAX, -DBL_MAX, 0.0};
            for (int k = 0; k < n; ++k) {
                HIP2(hipStreamSynchronize(g[k]->xstream));
                for (int q = 0; q < 3; ++q) wall[q] = std::max(wall[q], g[k]->h_red[q]);
            }
            for (int k = 0; k < n; ++k) launch_wall_set(g[k], wall);
            continue;
        }
        { const int rcx = exchange_all(true); if (rcx) return rcx; }
        bool do_dt = false;
        for (int k = 0; k < n; ++k) {
            Engine *e = g[k];
            step_back(e, i < nsteps - 1);
            if (!e->iso && e->steps_host % 10 == 0) {
                do_dt = true;
                launch_dt_partials(e);
                hipLaunchKernelGGL(k2_dt_pack, dim3(1), dim3(1), 0, e->stream, e->d_clk, e->d_red);
                HIP2(hipMemcpyAsync(e->h_red, e->d_red, 6 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
            }
        }
        if (do_dt) {
            double red[6];
            for (int q = 0; q < 6; ++q) red[q] = DBL_MAX;
            for (int k = 0; k < n; ++k) {
                HIP2(hipStreamSynchronize(g[k]->stream));
                for (int q = 0; q < 6; ++q) red[q] = std::min(red[q], g[k]->h_red[q]);
            }
            for (int k = 0; k < n; ++k) {
                Engine *e = g[k];
                for (int q = 0; q < 6; ++q) e->h_red[q] = red[q];
                HIP2(hipMemcpyAsync(e->d_red, e->h_red, 6 * sizeof(double), hipMemcpyHostToDevice, e->stream));
                hipLaunchKernelGGL(k2_dt_unpack, dim3(1), dim3(1), 0, e->stream, e->d_clk, e->d_red);
                hipLaunchKernelGGL(k2_dt_finalize, dim3(1), dim3(1), 0, e->stream, e->d_p, e->d_clk);
            }
        }
    }
    HIP2(hipGetLastError());
    int worst = DES_OK;
    double l2sum = 0;
    for (int k = 0; k < n; ++k) {
        if (out) { const int rc = fill_scalars(g[k], &out[k]); if (rc && !worst) worst = rc; l2sum += g[k]->h_clk->l2_sum; }
        else HIP2(hipStreamSynchronize(g[k]->stream));
    }
    // l2_residual over all ranks' owned nodes (every node counts once), as des_dev.h promises and the 3-D group does
    // (unless the last residual of the call was the pseudo-transient loop's global block sum, which every engine holds whole)
    if (out && n > 1 && !g[0]->h_clk->l2_global) for (int k = 0; k < n; ++k) out[k].l2_residual = std::sqrt(l2sum);
    return worst;
}

int check_nan(Engine *h, long long *n_nan)
{
    HIP2(hipSetDevice(h->device));
    unsigned long long *d_count = nullptr;
    HIP2(hipMalloc((void **)&d_count, sizeof(unsigned long long)));
    hipMemsetAsync(d_count, 0, sizeof(unsigned long long), h->stream);
    const long long nn = h->nn, ne = h->ne;
    struct { const double *a; long long n; } arr[] = {
        {h->volume, ne}, {h->dpressure, ne}, {h->viscosity, ne}, {h->stress, 3*ne}, {h->temperature, nn},
        {h->tmass, nn}, {h->force, 2*nn}, {h->vel, 2*nn}, {h->coord, 2*nn} };
    for (auto &a : arr) L2(k2_count_nan, a.n, a.n, a.a, d_count);
    unsigned long long c = 0;
    hipError_t e = hipMemcpyAsync(&c, d_count, sizeof(c), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d_count);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return DES_ERR_RESOURCE; }
    if (n_nan) *n_nan = (long long)c;
    return c ? DES_ERR_RUNTIME_NAN : DES_OK;
}

int mesh_quality(Engine *h, double smallest_vol, double bottom, double bottom_dist, des_quality *out)
{
    HIP2(hipSetDevice(h->device));
    des_quality *d_out = nullptr;
    HIP2(hipMalloc((void **)&d_out, sizeof(des_quality)));
    hipLaunchKernelGGL(k2_quality, dim3(1), dim3(DES_BLOCK), 0, h->stream, h->nn, h->ne, h->conn, h->coord, h->volume, h->bcflag,
                       smallest_vol, bottom, bottom_dist, d_out);
    hipError_t e = hipMemcpyAsync(out, d_out, sizeof(des_quality), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d_out);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return DES_ERR_RESOURCE; }
    return DES_OK;
}

int profile_enable(Engine *h, int on)
{
    HIP2(hipSetDevice(h->device));
    HIP2(hipStreamSynchronize(h->stream));
    for (Engine::ProfRec &r : h->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    h->prof_recs.clear();
    for (int k = 0; k < P2_COUNT; ++k) { h->prof_ms[k] = 0; h->prof_calls[k] = 0; }
    h->prof = on != 0;
    return DES_OK;
}

int profile_read(Engine *h, int cap, char (*names)[64], double *ms, long long *calls)
{
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    hipStreamSynchronize(h->xstream);
    for (Engine::ProfRec &r : h->prof_recs) {
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { h->prof_ms[r.k] += t; h->prof_calls[r.k] += 1; }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    h->prof_recs.clear();
    int n = 0;
    for (int k = 0; k < P2_COUNT && n < cap; ++k) {
        if (!h->prof_calls[k]) continue;
        std::strncpy(names[n], p2_names[k], 63); names[n][63] = 0;
        ms[n] = h->prof_ms[k]; calls[n] = h->prof_calls[k];
        ++n;
    }
    return n;
}

int timer_start(Engine *h) { HIP2(hipSetDevice(h->device)); HIP2(hipEventRecord(h->ev0, h->stream)); return DES_OK; }
int timer_stop(Engine *h, float *ms)
{
    HIP2(hipEventRecord(h->ev1, h->stream));
    HIP2(hipEventSynchronize(h->ev1));
    HIP2(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return DES_OK;
}

// reads + writes of the kernels above per step (8-byte words; gathers counted once per use)
double algorithmic_bytes_per_step(const Engine *h)
{
    return 8.0 * (118.0 * h->ne + 60.0 * h->nn);
}

} // namespace des2d
