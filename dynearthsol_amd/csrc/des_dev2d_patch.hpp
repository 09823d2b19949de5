// des_dev2d_patch.hpp -- the node-block patch passes of the 2-D (triangle) engine.
// Part of the translation unit des_dev2d.hip (included inside its anonymous namespace, after the
// device functions and the one-kernel-per-loop kernels); not a stand-alone header.
//
// The plain 2-D step runs every element -> node sum of the reference as a pair of launches: an element
// kernel stores its terms (three to six doubles per element), a node kernel gathers them per incidence
// through the support list -- scattered 8-byte reads, the slowest traffic the device has.  Here, as in
// the 3-D engine (passes/en1.hpp, passes/en3.hpp), a workgroup takes a block of `npb` nodes (a compact
// cluster of the mesh, below) and RECOMPUTES the terms of the block's patch (every element touching one of its nodes, listed once
// per mesh): the nodal records of the patch are staged in LDS (one gather per node instead of one per
// incidence), each lane takes patch elements and drops the terms of the element's nodes that belong
// to the block into the LDS slot of that incidence = its position in the block's slice of the CSR
// support list; after a barrier each node's lane adds its slots in CSR order, i.e. in the reference's
// ascending element order: same association, same bits as the gather.  Three such passes:
//   k2p_temp_dvoldt   update_temperature (fields.cxx:197-278) + compute_dvoldt (geometry.cxx:214-237)
//   k2p_force         NMD_stress' element update (geometry.cxx:311-331) + update_force (fields.cxx:609-698)
//   k2p_mass          compute_mass (geometry.cxx:1743-1870) with its volumes
// and the element kernels that remain do more per visit (k2_stress forms edvoldt from the nodal values;
// k2_rotate_vol = compute_volume + rotate_stress).

// A block is a list of up to `npb` nodes (po_id [po_ptr[b] .. po_ptr[b+1])): with the mesh's coordinates at hand
// (des_mesh::coord, the layout hint) the nodes are sorted along a Morton curve and cut into blocks, i.e. compact clusters
// -- an 11 x 11 cluster of a regular mesh touches 1.2 x its share of the elements, a strip of consecutive node ids (the
// renumbered mesh is sorted along x) 2.0 x; without coordinates the blocks are runs of consecutive ids.  The arrays stay
// in the caller's numbering: only the grouping of the work changes.  For every block:
//   po_slot [po_ptr[b] .. po_ptr[b+1])   where each of its nodes' incidences start in the block's LDS slots
//   pn_id   [pn_ptr[b] .. pn_ptr[b+1])   the nodes of its patch that are NOT its own, ascending
//   pe_pack [pe_ptr[b] .. pe_ptr[b+1])   its patch elements, ascending, one 16-byte record each:
//           .x = element (bits 0..29) | 1 << 30 if this block owns it (holds its lowest node) | ln0 << 31 | ln1 << 40 | ln2 << 49
//           .y = slot0 | slot1 << 12 | slot2 << 24
//           ln  = local id of the element's nodes in connectivity order (own: position in the block, others: nown + position in pn_id)
//           slot = po_slot of that node + the position of the incidence in the node's support list, 0xfff: not this block's node
struct Patch2 {
    int npb = 0, nb = 0, max_inc = 0, max_pn = 0, max_pe = 0;
    std::vector<int> po_ptr, po_id, po_slot, pe_ptr, pn_ptr, pn_id;
    std::vector<ulonglong2> pe_pack;
};

#define DES2_PATCH_INC 1024         // LDS slots of a block (incidences of its nodes)
#define DES2_PATCH_PN 448           // ... and nodes of its patch
#define DES2_PATCH_THREADS 256
#define DES2_PATCH_IT 3             // patch elements per lane at most: their list entries and element data are loaded up front

// false: a block exceeds the caps (the engine then tries smaller blocks, then keeps the plain kernels)
static bool build_patches2(const des_mesh *m, int npb, bool cluster, Patch2 &P)
{
    const int nn = m->nnode, ne = m->nelem;
    const int *conn = m->connectivity, *sidx = m->support_idx, *sarr = m->support_arr, *slid = m->support_lidx;
    P = Patch2();
    P.npb = npb; P.nb = (nn + npb - 1) / npb;
    std::vector<int> order((size_t)nn);
    for (int n = 0; n < nn; ++n) order[n] = n;
    if (cluster && m->coord) {
        // Morton order of the nodes on an isotropic 16-bit grid over the bounding box
        const double *X = m->coord, *Z = m->coord + nn;
        double x0 = X[0], x1 = X[0], z0 = Z[0], z1 = Z[0];
        for (int n = 1; n < nn; ++n) { x0 = std::min(x0, X[n]); x1 = std::max(x1, X[n]); z0 = std::min(z0, Z[n]); z1 = std::max(z1, Z[n]); }
        // clusters `aspect` times as tall (z) as wide (x): the renumbered mesh is sorted along x, so the ids -- of nodes and of
        // elements -- inside a thin x-strip are contiguous, and a cluster that is a few columns wide reads longer runs of
        // the element planes than a square one (DES2D_CLUSTER_ASPECT, default 4)
        const char *ae = des_env::get("DES2D_CLUSTER_ASPECT");
        const double aspect = ae && std::atof(ae) > 0 ? std::atof(ae) : 4.0;
        const double ext = std::max(x1 - x0, (z1 - z0) / aspect);
        const double sc = ext > 0 ? 65535.0 / ext : 0.0, scz = sc / aspect;
        auto spread = [](unsigned v) { unsigned long long r = v; r = (r | (r << 8)) & 0x00ff00ffull; r = (r | (r << 4)) & 0x0f0f0f0full;
                                       r = (r | (r << 2)) & 0x33333333ull; r = (r | (r << 1)) & 0x55555555ull; return r; };
        std::vector<unsigned long long> key((size_t)nn);
        for (int n = 0; n < nn; ++n)
            key[n] = ((spread((unsigned)((X[n] - x0) * sc)) | (spread((unsigned)((Z[n] - z0) * scz)) << 1)) << 32) | (unsigned)n;
        std::sort(key.begin(), key.end());
        for (int n = 0; n < nn; ++n) order[n] = (int)(key[n] & 0xffffffffull);
    }
    std::vector<int> blk_of((size_t)nn);
    for (int q = 0; q < nn; ++q) blk_of[order[q]] = q / npb;
    P.po_ptr.assign(1, 0); P.pe_ptr.assign(1, 0); P.pn_ptr.assign(1, 0);
    std::vector<int> emark((size_t)ne, -1), nmark((size_t)nn, -1), elems, others, own, pos((size_t)nn, -1), base;
    std::vector<int> slots;
    for (int b = 0; b < P.nb; ++b) {
        own.assign(order.begin() + (size_t)b * npb, order.begin() + std::min((size_t)nn, (size_t)(b + 1) * npb));
        std::sort(own.begin(), own.end());
        const int nown = (int)own.size();
        base.assign((size_t)nown, 0);
        int ninc = 0;
        elems.clear(); others.clear();
        for (int t = 0; t < nown; ++t) {
            const int n = own[t];
            pos[n] = t; base[t] = ninc; ninc += sidx[n + 1] - sidx[n];
            for (int k = sidx[n]; k < sidx[n + 1]; ++k) if (emark[sarr[k]] != b) { emark[sarr[k]] = b; elems.push_back(sarr[k]); }
        }
        if (ninc > DES2_PATCH_INC) return false;
        std::sort(elems.begin(), elems.end());
        for (int e : elems)
            for (int i = 0; i < 3; ++i) {
                const int n = conn[(size_t)i * ne + e];
                if (blk_of[n] != b && nmark[n] != b) { nmark[n] = b; others.push_back(n); }
            }
        std::sort(others.begin(), others.end());
        if (nown + (int)others.size() > DES2_PATCH_PN) return false;
        slots.assign(3 * elems.size(), 0xfff);
        for (int t = 0; t < nown; ++t) {
            const int n = own[t];
            for (int k = sidx[n]; k < sidx[n + 1]; ++k) {
                const size_t q = std::lower_bound(elems.begin(), elems.end(), sarr[k]) - elems.begin();
                slots[3 * q + slid[k]] = base[t] + (k - sidx[n]);
            }
        }
        for (size_t q = 0; q < elems.size(); ++q) {
            const int e = elems[q];
            int nmin = nn;
            unsigned long long ln[3];
            for (int i = 0; i < 3; ++i) {
                const int n = conn[(size_t)i * ne + e];
                nmin = std::min(nmin, n);
                ln[i] = blk_of[n] == b ? (unsigned long long)pos[n]
                                       : (unsigned long long)(nown + (std::lower_bound(others.begin(), others.end(), n) - others.begin()));
            }
            ulonglong2 r;
            r.x = (unsigned long long)(unsigned)e | (blk_of[nmin] == b ? 0x40000000ull : 0ull) | (ln[0] << 31) | (ln[1] << 40) | (ln[2] << 49);
            r.y = (unsigned long long)slots[3*q] | ((unsigned long long)slots[3*q + 1] << 12) | ((unsigned long long)slots[3*q + 2] << 24);
            P.pe_pack.push_back(r);
        }
        P.po_id.insert(P.po_id.end(), own.begin(), own.end());
        P.po_slot.insert(P.po_slot.end(), base.begin(), base.end());
        P.pn_id.insert(P.pn_id.end(), others.begin(), others.end());
        P.po_ptr.push_back((int)P.po_id.size());
        P.pe_ptr.push_back((int)P.pe_pack.size());
        P.pn_ptr.push_back((int)P.pn_id.size());
        P.max_inc = std::max(P.max_inc, ninc);
        P.max_pn = std::max(P.max_pn, nown + (int)others.size());
        P.max_pe = std::max(P.max_pe, (int)elems.size());
        if (P.max_pe > DES2_PATCH_IT * DES2_PATCH_THREADS) return false;
    }
    return true;
}

struct PatchElem2 { int e; bool owner; int ln[3]; int sl[3]; };
__device__ __forceinline__ PatchElem2 patch_elem2(const ulonglong2 r)
{
    PatchElem2 E;
    E.e = (int)(r.x & 0x3fffffffull);
    E.owner = (r.x >> 30) & 1ull;
    E.ln[0] = (int)((r.x >> 31) & 0x1ffull); E.ln[1] = (int)((r.x >> 40) & 0x1ffull); E.ln[2] = (int)((r.x >> 49) & 0x1ffull);
    E.sl[0] = (int)(r.y & 0xfffull); E.sl[1] = (int)((r.y >> 12) & 0xfffull); E.sl[2] = (int)((r.y >> 24) & 0xfffull);
    return E;
}

// pn_cap / inc_cap: LDS entries per nodal array / per slot array = the mesh's largest block, rounded up (dynamic LDS: a
// workgroup takes what the mesh needs, not the caps, so that more of them fit a CU)
// nb / blist: the blocks of THIS launch -- all of them (blist = nullptr), or the nb listed ones (overlapped schedule of a
// decomposed mesh: the blocks far from the cut first, the others behind the join with the exchange)
struct PatchArgs { int nn, ne, npb, nb, pn_cap, inc_cap; const int *po_ptr, *po_id, *po_slot, *pe_ptr; const ulonglong2 *pe_pack; const int *pn_ptr, *pn_id, *sup_idx;
                   const int *blist; };
__device__ __forceinline__ int patch_block(const PatchArgs &a, int front = 0)
{
    // (front: workgroups in front of the blocks' -- a multiple of 8, so that the blocks keep their XCDs)
    const int id = (int)blockIdx.x - front, per = (a.nb + 7) >> 3;
    const int b = (id & 7) * per + (id >> 3);              // desk::logical_block
    if (b >= a.nb) return -1;
    return a.blist ? a.blist[b] : b;
}

// ---- update_temperature + compute_dvoldt ------------------------------------------------------------
// thermal = 0: the temperature stands (isostasy loop, pseudo-transient iterations, has_thermal_diffusion = no).
// T_in / T_out: a block must not move a temperature another block may still be reading -- the host swaps the two.
// MASS = 1 (round 4): compute_mass of the step BEFORE rides in this pass (k2p_mass's statements on the same staged patch,
// the volumes from the coordinates as there; only together with vol_from_coords, i.e. on the steps whose end-of-step element
// pass was left to the coming stress update): the four sums of a node go first -- their LDS slots are then reused for this
// pass's two -- and the node's lane keeps volume_n and tmass for its own update.  One patch pass per step instead of two.
// pre.xz_pre (round 5, MASS = 1 only): the surface step of the step BEFORE rides here as well (one_step: surf_late).
//  - simple_diffusion (bc.cxx:1709-1787): that step's k2p_force<1> left the moved top nodes in xz_pre; a top node's committed
//    height is a function of its own and its two surface neighbours' entries alone (surf_commit_dh: k2_surf_commit's
//    statements), so every block that stages a top node forms the same height for it -- never using the z in memory, which
//    the node's own block is replacing meanwhile -- and the node's own lane stores what k2_surf_commit stores;
//  - correct_surface_element's element part (bc.cxx:1655-1707): the block that owns a top element rescales it at the end, on
//    the area of the staged (committed) coordinates -- cse_elem_at's statements; the coming k2_stress<M, 2> then finds in
//    volume[] the value compute_volume would find again (its nodal part is this pass's compute_mass anyway);
//  - edvacc_surf (bc.cxx:1788-1805): `nb_front` workgroups in front of the blocks', each segment from its two end
//    nodes' surf_commit_dh.
// pt_ptr / pt_ent: per block, the top nodes its patch holds as {staged slot, position in top_nodes, node, -}.
// (the struct lives in device memory -- `pre` is nullptr on every other step --: as kernel arguments its nineteen words cost the
//  pass 48 spilled SGPRs and 16 us.  coord and volume, which the host swaps, are the pass's own arguments.)
struct SurfPre { const double2 *xz_pre; const int *pt_ptr; const int4 *pt_ent; const unsigned char *topflag; int ntop, etop; const int *ean, *conn_surf;
                 double *total_dx, *total_slope, *dhacc, *dh, *edvacc, *plstrain, *stress, *strain; };

// IT: patch elements per lane at most (2 when the mesh's largest patch has at most 512: 17 registers less than with 3)
template <int MASS, int IT = DES2_PATCH_IT>
__global__ void __launch_bounds__(DES2_PATCH_THREADS, 5)
k2p_temp_dvoldt(const des_params *__restrict__ p, const Clock *__restrict__ clk, int thermal, int vol_from_coords, const PatchArgs a, const unsigned *bcflag,
                const double *coord, const double *vel, const double *T_in, double *T_out, const double *volume,
                const double *radiogenic, const double *props, const int *markers, const int *mono, const double *tmass_in, const double *volume_n_in,
                double *ntmp, double *strain_rate, double *volume_n_out, double *mass_out, double *tmass_out, double *ymass_out,
                const SurfPre *pre, int nb_front, const int *pt_ptr, int pe_cap, int outs_arg)
{
    extern __shared__ double lds[];
    double *const lx = lds, *const lz = lx + a.pn_cap, *const lvx = lz + a.pn_cap, *const lvz = lvx + a.pn_cap, *const lT = lvz + a.pn_cap;
    double *const lf0 = lT + a.pn_cap, *const lf1 = lf0 + a.inc_cap;
    // (blocks next to each other share half their patch: desk::logical_block keeps them on one XCD, i.e. one L2)
    const int nn = a.nn, ne = a.ne;
    if (MASS && (int)blockIdx.x < nb_front) {
        // (in front: at the end of the grid they would start when the last blocks do and finish after them)
        const int i = (int)blockIdx.x * DES2_PATCH_THREADS + threadIdx.x;
        if (i < pre->etop) {
            // surf_edv_at's statements, the two heights from the nodes' own step
            double dh_e = 0., tdx, tsl;
            for (int j = 0; j < 2; j++) dh_e += surf_commit_dh(p, clk->dt, pre->ntop, pre->ean[j * pre->etop + i], pre->xz_pre, tdx, tsl);
            const double base = fabs(coord[pre->conn_surf[i]] - coord[pre->conn_surf[pre->etop + i]]);
            pre->edvacc[i] += dh_e * base / 2;
        }
        return;
    }
    const int b = patch_block(a, MASS ? nb_front : 0);
    if (b < 0) return;
    const int o0 = a.po_ptr[b], nown = a.po_ptr[b + 1] - o0;
    const int h0 = a.pn_ptr[b], nh = a.pn_ptr[b + 1] - h0;
    // (pt_ptr: SurfPre's, or all zeros on the other steps -- an argument of its own and read unconditionally, so that the
    //  request goes out beside the block's other list bounds: behind `pre->` and a test it was two more trips for EVERY block)
    const int pt0 = MASS ? pt_ptr[b] : 0, npt = MASS ? pt_ptr[b + 1] - pt0 : 0;
    const bool topb = npt > 0;
    // (read here, in front of everything the pass may store: two scalar loads -- behind the surface blocks' part the compiler
    //  makes them per-lane loads inside the compute_mass loop)
    const double pseudo_speed = MASS ? p->max_vbc_val * p->inertial_scaling : 0.0;
    // everything that does not depend on the staged records is loaded first, so that a workgroup's trips to memory overlap:
    // this lane's list entries, then the element data they name, beside the nodal records
    ulonglong2 rec[IT];
    double g_vol[IT], g_kc[IT], g_rad[IT];
    double g_bulk[IT], g_shear[IT], g_cp[IT];
    int g_mono[IT];
    const int q0 = a.pe_ptr[b] + threadIdx.x, qe = a.pe_ptr[b + 1];
#pragma unroll
    for (int k = 0; k < IT; ++k)
        if (q0 + k * DES2_PATCH_THREADS < qe) rec[k] = a.pe_pack[q0 + k * DES2_PATCH_THREADS];
#pragma unroll
    for (int k = 0; k < IT; ++k)
        if (q0 + k * DES2_PATCH_THREADS < qe) {
            const int e = (int)(rec[k].x & 0x3fffffffull);
            g_vol[k] = vol_from_coords ? 0.0 : volume[e];
            // (radiogenic == nullptr: every heat source is +0.0 -- the same arithmetic on a literal; prop2: nothing to fetch with one material)
            if (thermal) { g_kc[k] = prop2(p, props, ne, e, 4); g_rad[k] = radiogenic ? radiogenic[e] : 0.0; }
            if (thermal || MASS) g_mono[k] = mono[e];
            if (MASS) { g_bulk[k] = prop2(p, props, ne, e, 0); g_shear[k] = prop2(p, props, ne, e, 1); g_cp[k] = prop2(p, props, ne, e, 3); }
        }
    for (int j = threadIdx.x; j < nown + nh; j += DES2_PATCH_THREADS) {
        const int id = j < nown ? a.po_id[o0 + j] : a.pn_id[h0 + j - nown];
        lx[j] = coord[id]; lz[j] = coord[nn + id]; lvx[j] = vel[id]; lvz[j] = vel[nn + id]; lT[j] = T_in[id];
    }
    __syncthreads();
    if (topb) {
        // A surface block (one in twenty).  The staged z of a top node is stale or half-way replaced: the committed one goes
        // over it.  Nothing is stored to memory here -- see the end of the pass.
        if ((int)threadIdx.x < npt) {
            const int4 en = pre->pt_ent[pt0 + threadIdx.x];
            double t_tdx, t_tsl;
            const double t_d = surf_commit_dh(p, clk->dt, pre->ntop, en.y, pre->xz_pre, t_tdx, t_tsl);
            lz[en.x] = pre->xz_pre[en.y].y + t_d;
        }
        __syncthreads();
    }
    if (MASS) {
        // Round 5: ONE element loop and ONE node phase for compute_mass of the step before and this step's two sums (six LDS slot
        // arrays instead of four reused: three workgroups per CU instead of four, but two barriers instead of four, and the area,
        // the mean temperature and mat_rho of an element formed once instead of twice -- the same expressions on the same
        // values, so the same bits).
        // LDS (the launch sizes it): behind the five nodal arrays, five values per patch ELEMENT (volume, the three mass terms, the
        // dvoldt term: the same for each of the element's nodes), the conduction term per INCIDENCE (it differs per node) and the
        // incidence's patch element as 16 bits -- 31 KB for the 1.28M-triangle mesh's blocks instead of 46 with six values per
        // incidence: five workgroups per CU instead of three, and 11 LDS stores per element instead of 18
        double *const lvol = lf0, *const lm = lvol + pe_cap, *const ltm = lm + pe_cap, *const lym = ltm + pe_cap, *const let = lym + pe_cap;
        double *const ltd = let + pe_cap;
        unsigned short *const lidx = (unsigned short *)(ltd + a.inc_cap);
        const int mass_thermal = p->has_thermal_diffusion;
        // (outs_arg: bit 0 the output-only stores, bit 1: the coming k2_stress<M, 2> forms the strain rate itself -- none stored here)
        const int outs = outs_arg & 1;
        const bool sr_store = !(outs_arg & 2);
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            if (q0 + k * DES2_PATCH_THREADS >= qe) break;
            const PatchElem2 E = patch_elem2(rec[k]);
            const int e = E.e;
            double d[3][2], v[3][2], shpdx[3], shpdz[3], T[3];
            for (int i = 0; i < 3; ++i) {
                d[i][0] = lx[E.ln[i]]; d[i][1] = lz[E.ln[i]]; v[i][0] = lvx[E.ln[i]]; v[i][1] = lvz[E.ln[i]]; T[i] = lT[E.ln[i]];
            }
            const double vol_m = triangle_area(d[0], d[1], d[2]);
            // k2_volume_mass_elem's statements
            const desk::Mix mx = mix2(g_mono[k], markers, p->nmat, e);
            const double bulkm = g_bulk[k], shearm = g_shear[k];
            double Te = 0;
            for (int i = 0; i < 3; ++i) Te += T[i];
            Te /= 3;
            const double mrho = desk::mat_rho(p, mx, Te);
            double rho = p->is_quasi_static ? bulkm / (pseudo_speed * pseudo_speed) : mrho;
            double m = rho * vol_m / 3;
            double tm = mrho * g_cp[k] * vol_m / 3;
            double ym = 9 * bulkm * shearm / (3 * bulkm + shearm) / 3;
            // (vol_from_coords: see below; MASS is only launched with it)
            const double vol = vol_from_coords ? vol_m : g_vol[k];
            shape_fn2(d, vol, shpdx, shpdz);
            double kv = 0, rh = 0;
            if (thermal) { kv = g_kc[k] * vol; rh = g_rad[k] * vol * mrho / 3; }       // k2_temp_elem's statements
            // k2_strain_rate's statements; the block that owns the element stores the strain rate
            double s0 = 0, s1 = 0;
            for (int i = 0; i < 3; ++i) s0 += v[i][0] * shpdx[i];
            for (int i = 0; i < 3; ++i) s1 += v[i][1] * shpdz[i];
            if (E.owner && sr_store) {
                double s2 = 0;
                for (int i = 0; i < 3; ++i) s2 += 0.5 * (v[i][0] * shpdz[i] + v[i][1] * shpdx[i]);
                strain_rate[e] = s0; strain_rate[ne + e] = s1; strain_rate[2 * ne + e] = s2;
            }
            double dj = s0 + s1;
            const double et = dj * vol;
            const int q = (int)threadIdx.x + k * DES2_PATCH_THREADS;         // the element's position in the patch
            lvol[q] = vol_m; lm[q] = m; ltm[q] = tm; lym[q] = ym; let[q] = et;
            for (int i = 0; i < 3; ++i) {
                if (E.sl[i] == 0xfff) continue;
                lidx[E.sl[i]] = (unsigned short)q;
                if (thermal) {
                    double diffusion = 0.;
                    for (int j = 0; j < 3; ++j)
                        diffusion += (shpdx[i] * shpdx[j] + shpdz[i] * shpdz[j]) * T[j];
                    ltd[E.sl[i]] = diffusion * kv - rh;
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nown) {
            const int n = a.po_id[o0 + threadIdx.x];
            const int r0 = a.po_slot[o0 + threadIdx.x], r1 = r0 + (a.sup_idx[n + 1] - a.sup_idx[n]);
            const bool t_node = thermal && !(bcflag[n] & BOUNDZ1);
            double vn = 0, ms = 0, tms = 0, yms = 0, acc = 0., tdot = 0;
            for (int k = r0; k < r1; ++k) {
                const int q = lidx[k];
                vn += lvol[q];
                ms += lm[q];
                if (mass_thermal) tms += ltm[q];
                yms += lym[q];
                acc += let[q];
                if (t_node) tdot += ltd[k];
            }
            // (outs = 0: a step of a multi-step call that is not its last -- the thermal mass is used right here and formed anew by
            //  the next pass; ymass only enters damping option 4)
            volume_n_out[n] = vn; mass_out[n] = ms;
            if (outs) tmass_out[n] = tms;
            if (outs || p->damping_option == 4) ymass_out[n] = yms;
            ntmp[n] = acc / vn;
            if (thermal) T_out[n] = t_node ? lT[threadIdx.x] - clk->dt * tdot / tms : p->surface_temperature;
        }
    } else {
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        if (q0 + k * DES2_PATCH_THREADS >= qe) break;
        const PatchElem2 E = patch_elem2(rec[k]);
        const int e = E.e;
        double d[3][2], v[3][2], shpdx[3], shpdz[3], T[3];
        for (int i = 0; i < 3; ++i) {
            d[i][0] = lx[E.ln[i]]; d[i][1] = lz[E.ln[i]]; v[i][0] = lvx[E.ln[i]]; v[i][1] = lvz[E.ln[i]]; T[i] = lT[E.ln[i]];
        }
        // (vol_from_coords: the end-of-step element pass of the step before was left to the coming stress update, volume[]
        //  is a step old -- compute_volume's own expression on the staged coordinates gives the value it will store)
        const double vol = vol_from_coords ? triangle_area(d[0], d[1], d[2]) : g_vol[k];
        shape_fn2(d, vol, shpdx, shpdz);
        if (thermal) {
            // k2_temp_elem's statements
            const desk::Mix mx = mix2(g_mono[k], markers, p->nmat, e);
            double kv = g_kc[k] * vol;
            double Te = 0;
            for (int i = 0; i < 3; ++i) Te += T[i];
            Te /= 3;
            double rh = g_rad[k] * vol * desk::mat_rho(p, mx, Te) / 3;
            for (int i = 0; i < 3; ++i) {
                if (E.sl[i] == 0xfff) continue;
                double diffusion = 0.;
                for (int j = 0; j < 3; ++j)
                    diffusion += (shpdx[i] * shpdx[j] + shpdz[i] * shpdz[j]) * T[j];
                lf0[E.sl[i]] = diffusion * kv - rh;
            }
        }
        // k2_strain_rate's statements; the block that owns the element stores the strain rate
        double s0 = 0, s1 = 0;
        for (int i = 0; i < 3; ++i) s0 += v[i][0] * shpdx[i];
        for (int i = 0; i < 3; ++i) s1 += v[i][1] * shpdz[i];
        if (E.owner) {
            double s2 = 0;
            for (int i = 0; i < 3; ++i) s2 += 0.5 * (v[i][0] * shpdz[i] + v[i][1] * shpdx[i]);
            strain_rate[e] = s0; strain_rate[ne + e] = s1; strain_rate[2 * ne + e] = s2;
        }
        double dj = s0 + s1;
        const double et = dj * vol;
        for (int i = 0; i < 3; ++i) if (E.sl[i] != 0xfff) lf1[E.sl[i]] = et;
    }
    __syncthreads();
    if ((int)threadIdx.x < nown) {
        const int n = a.po_id[o0 + threadIdx.x];
        const int r0 = a.po_slot[o0 + threadIdx.x], r1 = r0 + (a.sup_idx[n + 1] - a.sup_idx[n]);
        double acc = 0.;
        for (int k = r0; k < r1; ++k) acc += lf1[k];
        ntmp[n] = acc / volume_n_in[n];
        if (thermal) {
            if (bcflag[n] & BOUNDZ1)
                T_out[n] = p->surface_temperature;
            else {
                double tdot = 0;
                for (int k = r0; k < r1; ++k) tdot += lf0[k];
                T_out[n] = lT[threadIdx.x] - clk->dt * tdot / tmass_in[n];
            }
        }
    }
    }
    if (topb) {
        // The late surface step's stores, all at the very end: a store through a pointer the compiler cannot tell from `p`
        // turns every later read of the parameters from a scalar load into a per-lane one (three in the compute_mass loop alone;
        // with the stores up front the pass was 6 us longer for EVERY block).  The heights once more, then
        // correct_surface_element's element part by the block that owns the element (lx / lz still hold the staged patch).
        if ((int)threadIdx.x < npt) {
            const int4 en = pre->pt_ent[pt0 + threadIdx.x];
            const int t_slot = en.x, t_pos = en.y, t_node = en.z;
            if (t_slot < nown) {
                double t_tdx, t_tsl;
                const double t_d = surf_commit_dh(p, clk->dt, pre->ntop, t_pos, pre->xz_pre, t_tdx, t_tsl);
                pre->total_dx[t_node] = t_tdx; pre->total_slope[t_node] = t_tsl;
                pre->dh[t_pos] = t_d;
                const_cast<double *>(coord)[nn + t_node] = pre->xz_pre[t_pos].y + t_d;
                pre->dhacc[t_node] += t_d;
            }
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            if (q0 + k * DES2_PATCH_THREADS >= qe) break;
            const PatchElem2 E = patch_elem2(rec[k]);
            const int e = E.e;
            if (!E.owner || !pre->topflag[e]) continue;
            double d[3][2];
            for (int i = 0; i < 3; ++i) { d[i][0] = lx[E.ln[i]]; d[i][1] = lz[E.ln[i]]; }
            const double new_volumes = triangle_area(d[0], d[1], d[2]);
            const double rdv = new_volumes / volume[e];
            const_cast<double *>(volume)[e] = new_volumes;
            if (!(rdv < 1.0)) {
                pre->plstrain[e] = pre->plstrain[e] / rdv;
                for (int j = 0; j < 3; j++) {
                    pre->stress[j*ne+e] /= rdv;
                    pre->strain[j*ne+e] /= rdv;
                }
            }
        }
    }
}

// ---- NMD_stress (element update) + update_force ---------------------------------------------------------
// nmd: the stress k2_stress left in stress_in gets its diagonal corrected here; the block that owns the element stores the
// result to stress_out (another buffer: a neighbouring block may still be reading stress_in).  nmd = 0: stress_in is read only.
// TAIL = 1 (round 5, the launch diet the 3-D engine's EN3 has had since round 2): everything nodal that follows the force sums
// of a plain step rides in the node phase -- apply_stress_bcs for the node's own boundary facets (k2_sbc_direct's statements,
// from one list per node in that kernel's order: boundary by boundary, incidence by incidence), apply_damping +
// update_velocity, apply_vbcs, update_coordinate (k2_node_final's statements: the very device functions) and the node's share
// of calculate_residual_force, summed per patch block -- instead of k2_sbc_direct, k2_node_final and k2_residual_part as
// launches of their own.  The moved coordinates go to the OTHER buffer of a pair (another block may still be staging this
// block's nodes); the host swaps.  The wall's extent (apply_vbcs' depth profiles) is in the clock before this launch.
struct ForceTail {
    const Clock *clk; const double *mass, *ymass; const unsigned *bcflag; const double *bnormals, *edge_vec; const int *edge_slot;
    double *vel, *coord_out; const int *conn; const int *sbcn_idx; const int4 *sbcn_ent; double *res_part; int o0, o1, nn_global;
    const int *top_pos; double2 *xz_pre;        // a top node's position in top_nodes (-1: below the surface); its moved {x, z} there (k2_surf_commit)
};
template <int TAIL, int IT = DES2_PATCH_IT>
__global__ void __launch_bounds__(DES2_PATCH_THREADS)
k2p_force(const des_params *__restrict__ p, int nmd, int outs, const PatchArgs a, const double *coord, const double *temperature, const double *ntmp,
          const double *volume, const double *dpressure, const double *stress_in, double *stress_out, const double *props,
          const int *markers, const int *mono, double *force, double *fres, const ForceTail ft, const double *stress_shear)
{
    extern __shared__ double lds[];
    double *const lx = lds, *const lz = lx + a.pn_cap, *const lT = lz + a.pn_cap, *const lnt = lT + a.pn_cap;
    double *const lf0 = lnt + a.pn_cap, *const lf1 = lf0 + a.inc_cap;
    // (blocks next to each other share half their patch: desk::logical_block keeps them on one XCD, i.e. one L2)
    const int b = patch_block(a), nn = a.nn, ne = a.ne;
    if (b < 0) return;
    const int o0 = a.po_ptr[b], nown = a.po_ptr[b + 1] - o0;
    const int h0 = a.pn_ptr[b], nh = a.pn_ptr[b + 1] - h0;
    const double gravity = p->gravity;
    ulonglong2 rec[IT];
    double g_vol[IT], g_s[IT][3], g_dp[IT], g_phi[IT];
    int g_mono[IT];
    const int q0 = a.pe_ptr[b] + threadIdx.x, qe = a.pe_ptr[b + 1];
#pragma unroll
    for (int k = 0; k < IT; ++k)
        if (q0 + k * DES2_PATCH_THREADS < qe) rec[k] = a.pe_pack[q0 + k * DES2_PATCH_THREADS];
#pragma unroll
    for (int k = 0; k < IT; ++k)
        if (q0 + k * DES2_PATCH_THREADS < qe) {
            const int e = (int)(rec[k].x & 0x3fffffffull);
            // (TAIL: the plain step -- volume[] holds compute_volume's expression on the very coordinates staged here, so the pass
            //  forms it instead of reading 8 B per patch element)
            g_vol[k] = TAIL ? 0.0 : volume[e];
            // (the shear component comes from the stress array itself: k2_stress put it there, NMD_stress does not touch it)
            g_s[k][0] = stress_in[e]; g_s[k][1] = stress_in[ne + e]; g_s[k][2] = stress_shear[e];
            g_dp[k] = nmd ? dpressure[e] : 0.0;
            g_phi[k] = gravity != 0 ? prop2(p, props, ne, e, 2) : 0.0;
            g_mono[k] = gravity != 0 ? mono[e] : 0;
        }
    // TAIL: what the node phase needs of its own node is requested HERE, with everything else (one trip to memory for the
    // whole workgroup, not another chain of them behind the force sums)
    const int t_dopt = TAIL ? p->damping_option : 0;
    const double t_dfac = TAIL ? p->damping_factor : 0.0, t_dt = TAIL ? ft.clk->dt : 0.0;
    int t_n = 0, t_b0 = 0, t_b1 = 0, t_top = -1;
    unsigned t_flag = 0;
    double t_mass = 1.0, t_ymass = 0.0, t_v[2] = {0, 0};
    if (TAIL && (int)threadIdx.x < nown) {
        t_n = a.po_id[o0 + threadIdx.x];
        t_flag = ft.bcflag[t_n];
        t_mass = ft.mass[t_n];
        if (t_dopt == 4) t_ymass = ft.ymass[t_n];
        t_v[0] = ft.vel[t_n]; t_v[1] = ft.vel[nn + t_n];
        if (t_flag & BOUND_ANY) { t_b0 = ft.sbcn_idx[t_n]; t_b1 = ft.sbcn_idx[t_n + 1]; }
        if (t_flag & BOUNDZ1) t_top = ft.top_pos[t_n];
    }
    for (int j = threadIdx.x; j < nown + nh; j += DES2_PATCH_THREADS) {
        const int id = j < nown ? a.po_id[o0 + j] : a.pn_id[h0 + j - nown];
        lx[j] = coord[id]; lz[j] = coord[nn + id];
        if (gravity != 0) lT[j] = temperature[id];
        if (nmd) lnt[j] = ntmp[id];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        if (q0 + k * DES2_PATCH_THREADS >= qe) break;
        const PatchElem2 E = patch_elem2(rec[k]);
        const int e = E.e;
        double d[3][2], shpdx[3], shpdz[3];
        for (int i = 0; i < 3; ++i) { d[i][0] = lx[E.ln[i]]; d[i][1] = lz[E.ln[i]]; }
        double vol = TAIL ? triangle_area(d[0], d[1], d[2]) : g_vol[k];
        shape_fn2(d, vol, shpdx, shpdz);
        double s[3];
        for (int i = 0; i < 3; ++i) s[i] = g_s[k][i];
        if (nmd) {
            // k2_nmd_apply's statements
            double dp = 0;
            for (int i = 0; i < 3; ++i) dp += lnt[E.ln[i]];
            double dp_el = dp / 3;
            double dp_orig = g_dp[k];
            double ddp = (-dp_orig + dp_el) / 2;
            for (int i = 0; i < 2; ++i) s[i] += ddp;
            if (E.owner) for (int i = 0; i < 2; ++i) stress_out[i * ne + e] = s[i];
        }
        double buoy = 0;
        if (gravity != 0) {
            const desk::Mix mx = mix2(g_mono[k], markers, p->nmat, e);
            const double phi = g_phi[k];
            double Te = 0;
            for (int i = 0; i < 3; ++i) Te += lT[E.ln[i]];
            Te /= 3;
            buoy = (desk::mat_rho(p, mx, Te) * (1 - phi) + 1000.0 * phi) * gravity / 3;
        }
        for (int i = 0; i < 3; ++i) {
            if (E.sl[i] == 0xfff) continue;
            lf0[E.sl[i]] = (s[0]*shpdx[i] + s[2]*shpdz[i]) * vol;
            lf1[E.sl[i]] = (s[2]*shpdx[i] + s[1]*shpdz[i] + buoy) * vol;
        }
    }
    __syncthreads();
    double l2 = 0.0;
    if ((int)threadIdx.x < nown) {
        const int n = a.po_id[o0 + threadIdx.x];
        const int r0 = a.po_slot[o0 + threadIdx.x], r1 = r0 + (a.sup_idx[n + 1] - a.sup_idx[n]);
        double f[2] = {0, 0}, fr[2] = {0, 0};
        for (int k = r0; k < r1; ++k) {
            f[0] -= lf0[k]; fr[0] = lf0[k];            // assignment: fields.cxx:673
            f[1] -= lf1[k]; fr[1] = lf1[k];
        }
        // (outs = 0: a step of a multi-step call that is not its last -- force and force_residual are formed anew by the next
        //  step's pass before anything reads them, and what the step needs of them it has here in registers)
        if (!TAIL || outs) for (int j = 0; j < 2; j++) fres[j*nn + n] = fr[j];
        if (!TAIL) { for (int j = 0; j < 2; j++) force[j*nn + n] = f[j]; }
        else {
            // apply_stress_bcs (bc.cxx:661-827): k2_sbc_direct's walk for this node, one loaded boundary after the other
            for (int k = t_b0; k < t_b1; ++k) {
                const int4 ent = ft.sbcn_ent[k];
                double normal[2];
                const double pr = sbc_facet_pressure(p, ent.w, ent.x, ent.y, nn, ne, ft.conn, coord, temperature, markers, normal);
                f[0] -= pr * normal[0] / 2;
                f[1] -= pr * normal[1] / 2;
            }
            // k2_node_final's statements on the values held here: apply_damping + update_velocity, apply_vbcs (the node's z
            // before it moves: the staged one), update_coordinate into the other buffer of the pair
            const double dt = t_dt;
            damp_vel_regs(t_dopt, t_dfac, dt, t_mass, t_ymass, f, t_v);
            if (outs) for (int j = 0; j < 2; j++) force[j*nn + n] = f[j];
            const double x0 = lx[threadIdx.x], z0 = lz[threadIdx.x];
            vbcs_regs(p, ft.clk, t_flag, z0, ft.bnormals, ft.edge_vec, ft.edge_slot, t_v);
            ft.vel[n] = t_v[0]; ft.vel[nn + n] = t_v[1];
            const double x_new = x0 + t_v[0] * dt, z_new = z0 + t_v[1] * dt;
            ft.coord_out[n] = x_new;
            ft.coord_out[nn + n] = z_new;
            if (t_top >= 0) ft.xz_pre[t_top] = make_double2(x_new, z_new);
            // calculate_residual_force (fields.cxx:700-722): this node's terms (k2_residual_part's expression)
            if (n >= ft.o0 && n < ft.o1) {
                const double num = (double)ft.nn_global * 2;
                for (int j = 0; j < 2; ++j) l2 += fr[j] * fr[j] / num;
            }
        }
    }
    if (!TAIL) return;
    // per-block partial; the partials are added in block order afterwards (k2_residual_fin / k2_surf_seg_resfin)
    __shared__ double red[DES2_PATCH_THREADS / 64];
    l2 = desk::wave_sum(l2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l2;
    __syncthreads();
    if (threadIdx.x == 0) ft.res_part[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- compute_mass (with the volumes it sums) ----------------------------------------------------------------
// The element volumes are recomputed from the staged coordinates with compute_volume's own expression (k2_rotate_vol
// stores the same value to volume[]).
__global__ void __launch_bounds__(DES2_PATCH_THREADS)
k2p_mass(const des_params *__restrict__ p, const PatchArgs a, const double *coord, const double *temperature, const double *props,
         const int *markers, const int *mono, double *volume_n, double *mass, double *tmass, double *ymass)
{
    extern __shared__ double lds[];
    double *const lx = lds, *const lz = lx + a.pn_cap, *const lT = lz + a.pn_cap;
    double *const lf0 = lT + a.pn_cap, *const lf1 = lf0 + a.inc_cap, *const lf2 = lf1 + a.inc_cap, *const lf3 = lf2 + a.inc_cap;
    // (blocks next to each other share half their patch: desk::logical_block keeps them on one XCD, i.e. one L2)
    const int b = patch_block(a), nn = a.nn, ne = a.ne;
    if (b < 0) return;
    const int o0 = a.po_ptr[b], nown = a.po_ptr[b + 1] - o0;
    const int h0 = a.pn_ptr[b], nh = a.pn_ptr[b + 1] - h0;
    ulonglong2 rec[DES2_PATCH_IT];
    double g_bulk[DES2_PATCH_IT], g_shear[DES2_PATCH_IT], g_cp[DES2_PATCH_IT];
    int g_mono[DES2_PATCH_IT];
    const int q0 = a.pe_ptr[b] + threadIdx.x, qe = a.pe_ptr[b + 1];
#pragma unroll
    for (int k = 0; k < DES2_PATCH_IT; ++k)
        if (q0 + k * DES2_PATCH_THREADS < qe) rec[k] = a.pe_pack[q0 + k * DES2_PATCH_THREADS];
#pragma unroll
    for (int k = 0; k < DES2_PATCH_IT; ++k)
        if (q0 + k * DES2_PATCH_THREADS < qe) {
            const int e = (int)(rec[k].x & 0x3fffffffull);
            g_bulk[k] = prop2(p, props, ne, e, 0); g_shear[k] = prop2(p, props, ne, e, 1); g_cp[k] = prop2(p, props, ne, e, 3); g_mono[k] = mono[e];
        }
    for (int j = threadIdx.x; j < nown + nh; j += DES2_PATCH_THREADS) {
        const int id = j < nown ? a.po_id[o0 + j] : a.pn_id[h0 + j - nown];
        lx[j] = coord[id]; lz[j] = coord[nn + id]; lT[j] = temperature[id];
    }
    __syncthreads();
    const int thermal = p->has_thermal_diffusion;
#pragma unroll
    for (int k = 0; k < DES2_PATCH_IT; ++k) {
        if (q0 + k * DES2_PATCH_THREADS >= qe) break;
        const PatchElem2 E = patch_elem2(rec[k]);
        const int e = E.e;
        double d[3][2];
        for (int i = 0; i < 3; ++i) { d[i][0] = lx[E.ln[i]]; d[i][1] = lz[E.ln[i]]; }
        const double vol = triangle_area(d[0], d[1], d[2]);
        // k2_volume_mass_elem's statements
        const desk::Mix mx = mix2(g_mono[k], markers, p->nmat, e);
        const double bulkm = g_bulk[k], shearm = g_shear[k];
        double Te = 0;
        for (int i = 0; i < 3; ++i) Te += lT[E.ln[i]];
        Te /= 3;
        const double mrho = desk::mat_rho(p, mx, Te);
        const double pseudo_speed = p->max_vbc_val * p->inertial_scaling;
        double rho = p->is_quasi_static ? bulkm / (pseudo_speed * pseudo_speed) : mrho;
        double m = rho * vol / 3;
        double tm = mrho * g_cp[k] * vol / 3;
        double ym = 9 * bulkm * shearm / (3 * bulkm + shearm) / 3;
        for (int i = 0; i < 3; ++i) {
            if (E.sl[i] == 0xfff) continue;
            lf0[E.sl[i]] = vol; lf1[E.sl[i]] = m; lf2[E.sl[i]] = tm; lf3[E.sl[i]] = ym;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nown) {
        const int n = a.po_id[o0 + threadIdx.x];
        const int r0 = a.po_slot[o0 + threadIdx.x], r1 = r0 + (a.sup_idx[n + 1] - a.sup_idx[n]);
        double vn = 0, ms = 0, tms = 0, yms = 0;
        for (int k = r0; k < r1; ++k) {
            vn += lf0[k];
            ms += lf1[k];
            if (thermal) tms += lf2[k];
            yms += lf3[k];
        }
        volume_n[n] = vn; mass[n] = ms; tmass[n] = tms; ymass[n] = yms;
    }
}

// compute_volume (geometry.cxx:170-201) after the volume swap + rotate_stress (fields.cxx:807-821, 885-900) of an element
__global__ void k2_rotate_vol(const Clock *clk, int rotate, int nn, int ne, const int *conn, const double *coord, const double *vel,
                              double *volume, double *stress, double *strain)
{
    const int e = blockIdx.x * DES_BLOCK + threadIdx.x;
    if (e >= ne) return;
    double d[3][2];
    elem_coords(coord, conn, nn, ne, e, d);
    const double vol = triangle_area(d[0], d[1], d[2]);
    volume[e] = vol;
    if (!rotate) return;
    double shpdx[3], shpdz[3], v[3][2];
    shape_fn2(d, vol, shpdx, shpdz);
    elem_coords(vel, conn, nn, ne, e, v);
    double w2 = 0;
    for (int i = 0; i < 3; ++i) w2 += 0.5 * (v[i][1] * shpdx[i] - v[i][0] * shpdz[i]);
    double s[3], es[3];
    for (int i = 0; i < 3; ++i) { s[i] = stress[i*ne+e]; es[i] = strain[i*ne+e]; }
    jaumann_rate_2d(s, clk->dt, w2);
    jaumann_rate_2d(es, clk->dt, w2);
    for (int i = 0; i < 3; ++i) { stress[i*ne+e] = s[i]; strain[i*ne+e] = es[i]; }
}
