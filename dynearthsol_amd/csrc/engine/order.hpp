// engine/order.hpp -- Engine-internal (Morton) order and the translation tables.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

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
    int e_int0, e_int1;        // [e_int0, e_int1): elements whose four nodes are all owned
    // Decomposed meshes: the owned nodes are ordered [near the low cut | deep | near the high cut], "deep" = at least
    // DES_DEEP_DIST element layers from every ghost node; [n_deep0, n_deep1) is that middle range (engine numbering)
    // and [e_deep0, e_deep1) the elements whose four nodes lie DES_DEEP_MARGIN ids inside it -- so that any node block
    // of up to that many nodes which holds one of their nodes is made of deep nodes only.  What the passes of a step
    // compute on deep blocks / elements reads nothing the ghost-region exchange writes (engine/launch.hpp: overlap).
    int n_deep0, n_deep1, e_deep0, e_deep1;
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

// the same 3 x 21 bits along a Hilbert curve (Skilling's transpose): consecutive keys are always face neighbours, where the
// Morton curve jumps at every power-of-two boundary
inline unsigned long long hilbert3(unsigned x, unsigned y, unsigned z)
{
    unsigned X[3] = {x & 0x1fffffu, y & 0x1fffffu, z & 0x1fffffu};
    const unsigned M = 1u << 20;
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    return morton3(X[2], X[1], X[0]);                        // (X[0] carries the most significant bit of every level)
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
    // Round 5 (late): the HILBERT curve is the default -- consecutive keys are always face neighbours, so a block of 64 consecutive
    // nodes is more compact than on the Morton curve, which jumps at every power-of-two boundary: the patch lists of the 1M-tet
    // mesh hold 1.89 x the elements instead of 1.96 x, the largest patch 272 nodes / 843 elements instead of 290 / 870, and
    // neighbouring blocks share more of an L2: EN1 50.9 -> 49.3, EN3 51.3 -> 50.2, EN2 11.6 -> 11.0 us, the step 0.1866 -> 0.1828 ms
    // (profiles/r05_final_ab_hilbert_order.txt).  Names only: the same bits.  DES_CURVE=morton: the order of rounds 1-5.
    const char *curve = des_env::get("DES_CURVE");
    const bool hilbert = !(curve && curve[0] == 'm');
    auto code = [&](double x, double y, double z) {
        const unsigned a = (unsigned)((x - lo[0]) * scale), b = (unsigned)((y - lo[1]) * scale), c = (unsigned)((z - lo[2]) * scale);
        return hilbert ? hilbert3(a, b, c) : morton3(a, b, c);
    };
    const int ob = in->owned_begin, oe = in->owned_end > 0 ? in->owned_end : nn;
    // element layers between a node and the nearest ghost node of the low / the high range (capped at DES_DEEP_DIST)
    std::vector<unsigned char> dlo((size_t)nn, DES_DEEP_DIST), dhi((size_t)nn, DES_DEEP_DIST);
    if (ob > 0 || oe < nn) {
        auto spread = [&](std::vector<unsigned char> &d, int g0, int g1) {
            std::vector<int> front, next;
            for (int n = g0; n < g1; ++n) { d[n] = 0; front.push_back(n); }
            for (int layer = 1; layer < DES_DEEP_DIST && !front.empty(); ++layer) {
                next.clear();
                for (int n : front)
                    for (int k = in->support_idx[n]; k < in->support_idx[n + 1]; ++k) {
                        const int e = in->support_arr[k];
                        for (int i = 0; i < 4; ++i) {
                            const int m = in->connectivity[(size_t)i*ne + e];
                            if (d[m] > layer) { d[m] = (unsigned char)layer; next.push_back(m); }
                        }
                    }
                front.swap(next);
            }
        };
        spread(dlo, 0, ob);
        spread(dhi, oe, nn);
    }
    pm.n_deep0 = ob; pm.n_deep1 = oe;
    // (one domain: no groups -- the keys are the full 63-bit Morton codes; the group prefix, which costs the two or three
    //  finest bits, is only there when the mesh is a slab of a decomposed one)
    const bool cut = ob > 0 || oe < nn;
    {
        std::vector<std::pair<unsigned long long, int> > key((size_t)nn);
        for (int n = 0; n < nn; ++n) {
            // owned nodes: near the low cut, deep, near the high cut -- Morton order inside each group
            unsigned long long grp = 0;
            if (n >= ob && n < oe) grp = dlo[n] < DES_DEEP_DIST ? 0 : (dhi[n] < DES_DEEP_DIST ? 2 : 1);
            const unsigned long long mc = code(X[n], X[(size_t)nn + n], X[(size_t)2*nn + n]);
            key[n] = std::make_pair(cut ? (grp << 62 | mc >> 2) : mc, n);
        }
        // ... inside each of the three id ranges (the pairs break ties by caller id)
        std::sort(key.begin(), key.begin() + ob);
        std::sort(key.begin() + ob, key.begin() + oe);
        std::sort(key.begin() + oe, key.end());
        pm.n_new2old.resize((size_t)nn); pm.n_old2new.resize((size_t)nn);
        for (int i = 0; i < nn; ++i) { pm.n_new2old[i] = key[i].second; pm.n_old2new[key[i].second] = i; }
        for (int i = ob; cut && i < oe; ++i) {
            const unsigned long long grp = key[i].first >> 62;
            if (grp == 0) pm.n_deep0 = i + 1;
            if (grp == 2) { pm.n_deep1 = i; break; }
        }
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
            // groups: 0 touches the low ghost range, 1 owned but not deep (low side), 2 deep, 3 owned but not deep (high
            // side), 4 touches the high ghost range
            // (one domain: every element is deep, the order is the plain Morton one)
            const int d0 = cut ? pm.n_deep0 + DES_DEEP_MARGIN : 0, d1 = cut ? pm.n_deep1 - DES_DEEP_MARGIN : nn;
            bool deep = true, lowside = false;
            for (int i = 0; i < 4; ++i) {
                const int m = pm.n_old2new[in->connectivity[(size_t)i*ne + e]];
                deep = deep && m >= d0 && m < d1;
                lowside = lowside || m < d0;
            }
            grp = deep ? 2 : (lowside ? 1 : 3);
            if (touches_lo) grp = 0; else if (touches_hi) grp = 4;
            const unsigned long long mc = code(c[0], c[1], c[2]);
            key[e] = std::make_pair(cut ? (grp << 61 | mc >> 3) : mc, e);             // group, then Morton
        }
        std::sort(key.begin(), key.end());
        pm.e_int0 = 0; pm.e_int1 = ne; pm.e_deep0 = 0; pm.e_deep1 = ne;
        bool seen_deep = false;
        for (int i = 0; cut && i < ne; ++i) {
            const unsigned long long grp = key[i].first >> 61;
            if (grp == 0) pm.e_int0 = i + 1;
            if (grp <= 1) pm.e_deep0 = i + 1;
            if (grp == 2) seen_deep = true;
            if (grp >= 3 && pm.e_deep1 == ne) pm.e_deep1 = i;
            if (grp == 4) { pm.e_int1 = i; break; }
        }
        if (cut && !seen_deep) pm.e_deep1 = pm.e_deep0;
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
