// partition.cpp -- slab decomposition of the renumbered mesh for one-process-per-GPU runs.
//
// New work: the reference is single-process.  The renumbered mesh is sorted
// along its longest axis (mesh.cxx:2742-2766), so a contiguous range of node ids is a slab.
// Rank r OWNS such a range; its local mesh is the slab plus a ghost region of DES_GHOST_LAYERS
// element layers (des_halo, des_params.h), local numbering in ascending global order.  That keeps
// (a) every element patch that is complete locally (b) in the same ascending element order as
// on one GPU, so the nodal sums are bit-identical to an undecomposed run, and lets a rank do a
// whole step between two exchanges.
#include "des_host.h"
#include "des_host.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <set>
#include <unordered_map>

namespace {
const int NODE_OF_FACET3[4][3] = {{1,2,3},{0,3,2},{0,1,3},{0,2,1}};   // constants.hpp:64-69
const int NODE_OF_FACET2[3][2] = {{1,2},{2,0},{0,1}};                  // constants.hpp:71-75
inline int node_of_facet(int nd, int f, int j) { return nd == 3 ? NODE_OF_FACET3[f][j] : NODE_OF_FACET2[f % 3][j & 1]; }
}

#define DES_GHOST_LAYERS 4

struct des_part {
    des::HostMesh local;
    des_mesh view;
    des_halo halo;
    std::vector<int> l2g_node, l2g_elem;
    std::vector<int> nbr_rank, send_ptr, send_idx, recv_ptr, recv_idx;
    std::vector<int> esend_ptr, esend_idx, erecv_ptr, erecv_idx;
    std::vector<int> elem_owned;        // [local nelem] 1 where this rank owns the element's lowest-numbered node
    std::vector<int> node_start;        // [nranks+1] global ownership ranges
};

namespace des {

// split nodes into `nranks` contiguous ranges of about equal support size (element work)
static std::vector<int> split_nodes(const HostMesh &g, int nranks)
{
    std::vector<int> start(nranks + 1, 0);
    const long long total = g.sup_idx[g.nnode];
    int n = 0;
    for (int r = 1; r < nranks; ++r) {
        const long long target = total * r / nranks;
        while (n < g.nnode && g.sup_idx[n] < target) ++n;
        // cuts at multiples of the residual's block size: a block of the partition-independent residual (des_params.h) has one owner
        const int rb = des_res_block(g.nnode);
        n = std::min(g.nnode, (n + rb - 1) / rb * rb);
        start[r] = n;
    }
    start[nranks] = g.nnode;
    return start;
}

static inline int owner_of(const std::vector<int> &start, int n)
{
    return (int)(std::upper_bound(start.begin(), start.end(), n) - start.begin()) - 1;
}

// The part of the global mesh one rank holds: element layer (0 = touches an owned node, k+1 =
// touches a node first reached by layer k; -1 = not held) and node depth (0 = owned, k+1 =
// first reached by element layer k; -1 = not held).
struct Reach {
    std::vector<signed char> elem_layer, node_depth;
};

static void grow(const HostMesh &g, int a, int b, int nlayers, Reach &R)
{
    const int ne = g.nelem, nn = g.nnode, npe = g.nd + 1;
    R.elem_layer.assign((size_t)ne, -1);
    R.node_depth.assign((size_t)nn, -1);
    std::vector<int> frontier;
    for (int n = a; n < b; ++n) { R.node_depth[n] = 0; frontier.push_back(n); }
    for (int k = 0; k < nlayers; ++k) {
        std::vector<int> next;
        for (int n : frontier)
            for (int q = g.sup_idx[n]; q < g.sup_idx[n + 1]; ++q) {
                const int e = g.sup_arr[q];
                if (R.elem_layer[e] >= 0) continue;
                R.elem_layer[e] = (signed char)k;
                for (int i = 0; i < npe; ++i) {
                    const int m = g.conn[(size_t)i*ne + e];
                    if (R.node_depth[m] < 0) { R.node_depth[m] = (signed char)(k + 1); next.push_back(m); }
                }
            }
        frontier.swap(next);
    }
}

void build_partition(const HostMesh &g, int nranks, int rank, des_part &P)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) throw Error(60, "bad rank / nranks");
    const int ne = g.nelem, nn = g.nnode;
    const int nd = g.nd, npe = nd + 1, npf = nd;           // tets / triangles (the 2-D build: constants.hpp:12-25)
    const int nlayers = DES_GHOST_LAYERS;
    P.node_start = split_nodes(g, nranks);
    const int a = P.node_start[rank], b = P.node_start[rank + 1];
    if (b <= a) throw Error(52, "a rank owns no node: too many ranks for this mesh");

    // local elements and nodes: the slab and its ghost region
    Reach mine;
    grow(g, a, b, nlayers, mine);
    P.l2g_elem.clear();
    for (int e = 0; e < ne; ++e) if (mine.elem_layer[e] >= 0) P.l2g_elem.push_back(e);
    P.l2g_node.clear();
    std::vector<int> g2l((size_t)nn, -1);
    for (int n = 0; n < nn; ++n) if (mine.node_depth[n] >= 0) { g2l[n] = (int)P.l2g_node.size(); P.l2g_node.push_back(n); }
    const int lnn = (int)P.l2g_node.size(), lne = (int)P.l2g_elem.size();
    std::vector<int> g2l_elem((size_t)ne, -1);
    for (int e = 0; e < lne; ++e) g2l_elem[P.l2g_elem[e]] = e;

    HostMesh &m = P.local;
    m = HostMesh();
    m.nd = nd;
    m.nnode = lnn; m.nelem = lne; m.nseg = 0;
    m.coord.resize((size_t)nd*lnn);
    for (int n = 0; n < lnn; ++n)
        for (int d = 0; d < nd; ++d) m.coord[(size_t)d*lnn + n] = g.coord[(size_t)d*nn + P.l2g_node[n]];
    m.conn.resize((size_t)npe*lne);
    for (int e = 0; e < lne; ++e)
        for (int i = 0; i < npe; ++i) m.conn[(size_t)i*lne + e] = g2l[g.conn[(size_t)i*ne + P.l2g_elem[e]]];
    m.regattr.resize((size_t)lne);
    for (int e = 0; e < lne; ++e) m.regattr[e] = g.regattr[P.l2g_elem[e]];
    m.bcflag.resize((size_t)lnn);
    for (int n = 0; n < lnn; ++n) m.bcflag[n] = g.bcflag[P.l2g_node[n]];
    for (int i = 0; i < DES_NBDRY; ++i) {
        m.bnodes[i].clear();
        for (int n = 0; n < lnn; ++n) if (m.bcflag[n] & (1u << i)) m.bnodes[i].push_back(n);
        m.bfacet_elem[i].clear(); m.bfacet_facet[i].clear();
        for (size_t q = 0; q < g.bfacet_elem[i].size(); ++q) {
            int le = g2l_elem[g.bfacet_elem[i][q]];
            if (le < 0) continue;
            m.bfacet_elem[i].push_back(le);
            m.bfacet_facet[i].push_back(g.bfacet_facet[i][q]);
        }
    }
    // support of the local mesh (complete up to ghost depth nlayers-1, partial for the outermost nodes)
    m.sup_idx.assign((size_t)lnn + 1, 0);
    for (int e = 0; e < lne; ++e)
        for (int i = 0; i < npe; ++i) m.sup_idx[m.conn[(size_t)i*lne + e] + 1]++;
    for (int n = 1; n <= lnn; ++n) m.sup_idx[n] += m.sup_idx[n-1];
    m.sup_arr.resize((size_t)m.sup_idx[lnn]); m.sup_lidx.resize((size_t)m.sup_idx[lnn]);
    {
        std::vector<int> cursor(m.sup_idx.begin(), m.sup_idx.end() - 1);
        for (int e = 0; e < lne; ++e)
            for (int i = 0; i < npe; ++i) {
                int slot = cursor[m.conn[(size_t)i*lne + e]]++;
                m.sup_arr[slot] = e; m.sup_lidx[slot] = i;
            }
    }
    // surface lists, in the global x-sorted order of surfinfo.top_nodes
    m.top_nodes.clear();
    for (size_t i = 0; i < g.top_nodes.size(); ++i) { int l = g2l[g.top_nodes[i]]; if (l >= 0) m.top_nodes.push_back(l); }
    const int etop = (int)m.bfacet_elem[5].size(), ntop = (int)m.top_nodes.size();
    m.conn_surf.assign((size_t)npe*etop, 0);
    for (int i = 0; i < etop; ++i)
        for (int j = 0; j < npf; ++j)
            m.conn_surf[(size_t)j*etop + i] = m.conn[(size_t)node_of_facet(nd, m.bfacet_facet[5][i], j)*lne + m.bfacet_elem[5][i]];
    std::unordered_map<int,int> arctop;
    for (int i = 0; i < ntop; ++i) arctop[m.top_nodes[i]] = i;
    m.elem_and_nodes.assign((size_t)npf*etop, 0);
    for (int i = 0; i < etop; ++i)
        for (int k = 0; k < npf; ++k) m.elem_and_nodes[(size_t)k*etop + i] = arctop[m.conn_surf[(size_t)k*etop + i]];
    m.ssup_idx.assign((size_t)ntop + 1, 0);
    for (int i = 0; i < etop; ++i)
        for (int k = 0; k < npf; ++k) m.ssup_idx[m.elem_and_nodes[(size_t)k*etop + i] + 1]++;
    for (int n = 1; n <= ntop; ++n) m.ssup_idx[n] += m.ssup_idx[n-1];
    m.ssup_arr.resize((size_t)m.ssup_idx[ntop]);
    {
        std::vector<int> cursor(m.ssup_idx.begin(), m.ssup_idx.end() - 1);
        for (int i = 0; i < etop; ++i)
            for (int k = 0; k < npf; ++k) m.ssup_arr[cursor[m.elem_and_nodes[(size_t)k*etop + i]]++] = i;
    }
    m.top_elems.clear();
    for (size_t i = 0; i < g.top_elems.size(); ++i) { int le = g2l_elem[g.top_elems[i]]; if (le >= 0) m.top_elems.push_back(le); }
    m.bnormals = g.bnormals;
    m.edge_vec = g.edge_vec;
    std::memcpy(m.edge_slot, g.edge_slot, sizeof(m.edge_slot));

    // exchange lists.  Nodes: every ghost node comes from its owner.  Elements: the state of the
    // two outer layers (stale after a step: NMD_stress is only right up to layer nlayers-3) comes
    // from the rank owning the element's lowest-numbered node, for which it is a layer-0 element.
    // Both sides derive the same ascending lists from the same rule (the sender replays the
    // receiver's growth).
    auto elem_owner = [&](int e) {
        int lo = g.conn[e];
        for (int i = 1; i < npe; ++i) lo = std::min(lo, g.conn[(size_t)i*ne + e]);
        return owner_of(P.node_start, lo);
    };
    P.elem_owned.assign((size_t)lne, 0);
    for (int e = 0; e < lne; ++e) P.elem_owned[e] = elem_owner(P.l2g_elem[e]) == rank;
    std::set<int> nbrs;
    for (int n = 0; n < nn; ++n) if (mine.node_depth[n] >= 1) nbrs.insert(owner_of(P.node_start, n));
    P.nbr_rank.assign(nbrs.begin(), nbrs.end());
    P.send_ptr.assign(1, 0); P.recv_ptr.assign(1, 0); P.esend_ptr.assign(1, 0); P.erecv_ptr.assign(1, 0);
    P.send_idx.clear(); P.recv_idx.clear(); P.esend_idx.clear(); P.erecv_idx.clear();
    const int stale = nlayers - 2;
    for (int q : P.nbr_rank) {
        Reach theirs;
        grow(g, P.node_start[q], P.node_start[q + 1], nlayers, theirs);
        for (int n = a; n < b; ++n) if (theirs.node_depth[n] >= 1) P.send_idx.push_back(g2l[n]);
        for (int n = P.node_start[q]; n < P.node_start[q + 1]; ++n) if (mine.node_depth[n] >= 1) P.recv_idx.push_back(g2l[n]);
        for (int e = 0; e < ne; ++e) {
            if (theirs.elem_layer[e] >= stale && elem_owner(e) == rank) {
                if (mine.elem_layer[e] != 0) throw Error(60, "element to send is not a layer-0 element of its owner");
                P.esend_idx.push_back(g2l_elem[e]);
            }
            if (mine.elem_layer[e] >= stale && elem_owner(e) == q) P.erecv_idx.push_back(g2l_elem[e]);
        }
        P.send_ptr.push_back((int)P.send_idx.size()); P.recv_ptr.push_back((int)P.recv_idx.size());
        P.esend_ptr.push_back((int)P.esend_idx.size()); P.erecv_ptr.push_back((int)P.erecv_idx.size());
    }
    // every ghost node / stale element must be received exactly once
    {
        std::vector<char> got((size_t)lnn, 0);
        for (int l : P.recv_idx) { if (got[l]) throw Error(60, "ghost node received twice"); got[l] = 1; }
        for (int n = 0; n < lnn; ++n) {
            bool owned = P.l2g_node[n] >= a && P.l2g_node[n] < b;
            if (owned == (bool)got[n]) throw Error(60, "exchange lists do not cover the ghost nodes exactly");
        }
        std::vector<char> egot((size_t)lne, 0);
        for (int l : P.erecv_idx) { if (egot[l]) throw Error(60, "ghost element received twice"); egot[l] = 1; }
        for (int e = 0; e < lne; ++e)
            if ((mine.elem_layer[P.l2g_elem[e]] >= stale) != (bool)egot[e])
                throw Error(60, "exchange lists do not cover the stale element layers exactly");
    }
    P.view = m.view();
    P.view.owned_begin = (int)(std::lower_bound(P.l2g_node.begin(), P.l2g_node.end(), a) - P.l2g_node.begin());
    P.view.owned_end = (int)(std::lower_bound(P.l2g_node.begin(), P.l2g_node.end(), b) - P.l2g_node.begin());
    P.halo.owned_begin = (int)(std::lower_bound(P.l2g_node.begin(), P.l2g_node.end(), a) - P.l2g_node.begin());
    P.halo.owned_end = (int)(std::lower_bound(P.l2g_node.begin(), P.l2g_node.end(), b) - P.l2g_node.begin());
    P.halo.nlayers = nlayers;
    P.halo.owned_global_begin = a;
    P.halo.nnbr = (int)P.nbr_rank.size();
    P.halo.nbr_rank = P.nbr_rank.data();
    P.halo.send_ptr = P.send_ptr.data(); P.halo.send_idx = P.send_idx.data();
    P.halo.recv_ptr = P.recv_ptr.data(); P.halo.recv_idx = P.recv_idx.data();
    P.halo.esend_ptr = P.esend_ptr.data(); P.halo.esend_idx = P.esend_idx.data();
    P.halo.erecv_ptr = P.erecv_ptr.data(); P.halo.erecv_idx = P.erecv_idx.data();
}

} // namespace des

extern "C" {

// defined in capi.cpp
const des::HostMesh *des_host_mesh_internal(const des_host *h);
void des_host_set_error(const char *msg);

des_part *des_host_partition(const des_host *h, int nranks, int rank, int *err)
{
    des_part *P = new des_part();
    try {
        des::build_partition(*des_host_mesh_internal(h), nranks, rank, *P);
        if (err) *err = DES_OK;
        return P;
    } catch (const des::Error &e) {
        des_host_set_error(e.what());
        if (err) *err = e.code;
    } catch (const std::exception &e) {
        des_host_set_error(e.what());
        if (err) *err = DES_ERR_INTERNAL;
    }
    delete P;
    return nullptr;
}

void des_part_destroy(des_part *p) { delete p; }
const des_mesh *des_part_mesh(const des_part *p) { return &p->view; }
const des_halo *des_part_halo(const des_part *p) { return &p->halo; }
const int *des_part_l2g_node(const des_part *p, int *n) { if (n) *n = (int)p->l2g_node.size(); return p->l2g_node.data(); }
const int *des_part_l2g_elem(const des_part *p, int *n) { if (n) *n = (int)p->l2g_elem.size(); return p->l2g_elem.data(); }
const int *des_part_elem_owned(const des_part *p, int *n) { if (n) *n = (int)p->elem_owned.size(); return p->elem_owned.data(); }
const int *des_part_node_ranges(const des_part *p, int *n) { if (n) *n = (int)p->node_start.size(); return p->node_start.data(); }

} // extern "C"
