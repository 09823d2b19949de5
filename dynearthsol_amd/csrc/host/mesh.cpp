// mesh.cpp -- host mesh: regular-grid mesher, renumbering and the topology lists the
// time-stepper consumes.  Built once per (re)mesh on the CPU, as in the reference.
#include "des_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <set>
#include <unordered_map>

namespace des {

namespace {

const int NODE_OF_FACET3[4][3] = {{1,2,3},{0,3,2},{0,1,3},{0,2,1}};   // constants.hpp:64-69
const int NODE_OF_FACET2[3][2] = {{1,2},{2,0},{0,1}};                  // constants.hpp:71-75
// local node j of facet f in an nd-dimensional build
inline int node_of_facet(int nd, int f, int j) { return nd == 3 ? NODE_OF_FACET3[f][j] : NODE_OF_FACET2[f % 3][j & 1]; }
const unsigned BOUNDX0 = 1, BOUNDX1 = 2, BOUNDY0 = 4, BOUNDY1 = 8, BOUNDZ0 = 16, BOUNDZ1 = 32;
const unsigned BOUND_ANY = 0x3ff;
const int iboundz1 = 5, iboundn0 = 6;

// Tet split of one hexahedral cell; two mirror-image patterns alternate with the parity
// of (i+j+k) so that neighbouring cells share face diagonals (mesh.cxx:191-269).
const int TET_OF_CELL[2][5][4] = {
    {{0,1,2,5}, {0,2,3,7}, {0,4,5,7}, {2,5,6,7}, {0,5,2,7}},
    {{1,2,3,6}, {0,1,3,4}, {1,4,5,6}, {3,4,6,7}, {1,3,4,6}},
};

struct RegularGrid {
    int nx, ny, nz;
    std::vector<int> cell;      // [ncell][8]
};

// mesh.cxx:147-189 (THREED)
void cells_of_grid(RegularGrid &g)
{
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    g.cell.resize((size_t)(nx-1)*(ny-1)*(nz-1)*8);
    size_t c = 0;
    for (int i = 0; i < nx-1; ++i)
        for (int j = 0; j < ny-1; ++j)
            for (int k = 0; k < nz-1; ++k) {
                int idx0 = i*ny*nz + j*nz + k;
                int idx1 = idx0 + nz;
                int idx2 = idx1 + ny*nz;
                int idx3 = idx2 - nz;
                const int v[8] = {idx0, idx1, idx2, idx3, idx0+1, idx1+1, idx2+1, idx3+1};
                for (int q = 0; q < 8; ++q) g.cell[c++] = v[q];
            }
}

// The 2-D build of the same mesher (mesh.cxx:146-166, 271-297, 320-336, 350-390): quadrilateral
// cells cut into two counter-clockwise triangles, the diagonal alternating with the parity of i+j.
void new_mesh_regular2d(const Config &cfg, HostMesh &m)
{
    const double Lx = cfg.d("mesh.xlength"), Lz = cfg.d("mesh.zlength");
    const double res = cfg.d("mesh.resolution");
    const int nx = (int)std::round(Lx / res) + 1, nz = (int)std::round(Lz / res) + 1;   // mesh.cxx:1453-1458
    if (nx < 2 || nz < 2) throw Error(11, "regular mesh needs at least one cell per direction");
    const int ncell = (nx-1)*(nz-1), nnode = nx*nz, nelem = 2*ncell, nseg = 2*(nx + nz - 2);
    m.nd = 2;
    m.nnode = nnode; m.nelem = nelem; m.nseg = nseg;
    m.coord.resize((size_t)2*nnode);
    const double dx = Lx / (nx-1), dz = -Lz / (nz-1);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nz; ++j) {
            m.coord[(size_t)j + (size_t)i*nz] = i * dx;
            m.coord[(size_t)nnode + j + (size_t)i*nz] = j * dz;
        }
    m.conn.resize((size_t)3*nelem);
    for (int i = 0; i < nx-1; ++i)
        for (int j = 0; j < nz-1; ++j) {
            const int idx = i*(nz-1) + j;
            const int idx0 = i*nz + j, idx1 = idx0 + nz;
            const int cell[4] = {idx0, idx1, idx1 + 1, idx0 + 1};
            int t[6];
            if ((i+j) % 2 == 0) { t[0] = cell[0]; t[1] = cell[2]; t[2] = cell[1]; t[3] = cell[0]; t[4] = cell[3]; t[5] = cell[2]; }
            else                { t[0] = cell[0]; t[1] = cell[3]; t[2] = cell[1]; t[3] = cell[1]; t[4] = cell[3]; t[5] = cell[2]; }
            for (int q = 0; q < 2; ++q)
                for (int k = 0; k < 3; ++k) m.conn[(size_t)k*nelem + 2*idx + q] = t[3*q + k];
        }
    std::vector<int> seg, flag;
    auto add = [&](int a, int b, int f) { seg.push_back(a); seg.push_back(b); flag.push_back(f); };
    for (int i = 0; i < nx-1; ++i)
        for (int j = 0; j < nz-1; ++j) {
            if (i == 0) add(j + i*nz, j + i*nz + 1, (int)BOUNDX0);
            if (j == 0) add(j + i*nz, j + i*nz + nz, (int)BOUNDZ1);
        }
    for (int i = 1; i < nx; ++i) add(i*nz - 1, i*nz + nz - 1, (int)BOUNDZ0);
    for (int j = 0; j < nz-1; ++j) add(nz*(nx-1) + j, nz*(nx-1) + j + 1, (int)BOUNDX1);
    if ((int)flag.size() != nseg) throw Error(60, "regular mesh: segment count mismatch");
    m.segment.resize((size_t)2*nseg);
    for (int q = 0; q < nseg; ++q)
        for (int d = 0; d < 2; ++d) m.segment[(size_t)d*nseg + q] = seg[(size_t)q*2 + d];
    m.segflag = flag;
    m.regattr.assign((size_t)nelem, 0.0);     // mesh.cxx:570-575
}

// meshing_elem_shape = 2 (2-D only): rows of near-equilateral triangles (new_mesh_regular_equilateral,
// mesh.cxx:578-684, 2513-2596).  Rows alternate between nx nodes ("even" rows, the top one first) and
// nx + 1 nodes shifted by half a spacing ("odd" rows); the side walls stay at x = 0 and xlength, the last
// row sits at -zlength.  Node numbering: all even rows first, then all odd rows; each strip between two
// rows carries nx - 1 triangles with their base on the even row and nx with their base on the odd row,
// the family whose base is on the strip's upper row first; connectivity counter-clockwise.
void new_mesh_equilateral2d(const Config &cfg, HostMesh &m)
{
    const double Lx = cfg.d("mesh.xlength"), Lz = cfg.d("mesh.zlength"), res = cfg.d("mesh.resolution");
    const double sqrt3_to_2 = 2. / std::sqrt(3.0);
    const double x_mid = Lx / 2;
    const int nx = int((x_mid - 0.5*res) / res) * 2 + 2;
    const int nz = int(Lz * sqrt3_to_2 / res) + 1;
    if (nx < 2 || nz < 2) throw Error(11, "regular mesh needs at least one cell per direction");
    const int n_even_rows = (nz + 1) / 2, n_odd_rows = nz / 2;
    const int nnode = nx * n_even_rows + (nx + 1) * n_odd_rows;
    const int nelem = (2*nx - 1) * (nz - 1);
    const int nseg = (nx - 1) + (nx - nz % 2) + 2 * (nz - 1);
    const int odd_base = nx * n_even_rows;
    // first node of row j and its node count
    auto row_start = [&](int j) { return (j % 2 == 0) ? nx * (j / 2) : odd_base + (nx + 1) * (j / 2); };
    auto row_count = [&](int j) { return (j % 2 == 0) ? nx : nx + 1; };

    m.nd = 2;
    m.nnode = nnode; m.nelem = nelem; m.nseg = nseg;
    m.coord.assign((size_t)2*nnode, 0.0);
    const double dx = res, dz = -res * std::sqrt(3.0) / 2.;
    const double bdy_dx = (Lx - (nx-1)*dx) / 2.;
    for (int j = 0; j < nz; ++j) {
        const int s0 = row_start(j), cnt = row_count(j);
        const double z = (j == nz-1) ? -Lz : j * dz;
        for (int i = 0; i < cnt; ++i) {
            double x;
            if (i == 0) x = 0.;
            else if (i == cnt-1) x = Lx;
            else x = (j % 2 == 0) ? i * dx + bdy_dx : ((i-1) + 0.5) * dx + bdy_dx;
            m.coord[(size_t)s0 + i] = x;
            m.coord[(size_t)nnode + s0 + i] = z;
        }
    }

    m.conn.assign((size_t)3*nelem, 0);
    int e = 0;
    auto tri = [&](int a, int b, int c) {
        m.conn[e] = a; m.conn[(size_t)nelem + e] = b; m.conn[(size_t)2*nelem + e] = c; ++e;
    };
    for (int j = 0; j < nz-1; ++j) {
        const bool even_on_top = (j % 2 == 0);
        const int ev = row_start(even_on_top ? j : j+1), od = row_start(even_on_top ? j+1 : j);
        // base on the even row: (even i, even i+1) + the odd node between them
        auto even_based = [&]() {
            for (int i = 0; i < nx-1; ++i) {
                if (even_on_top) tri(ev + i, od + i + 1, ev + i + 1);
                else             tri(ev + i, ev + i + 1, od + i + 1);
            }
        };
        // base on the odd row: (odd i, odd i+1) + the even node between them
        auto odd_based = [&]() {
            for (int i = 0; i < nx; ++i) {
                if (even_on_top) tri(od + i, od + i + 1, ev + i);
                else             tri(od + i, ev + i, od + i + 1);
            }
        };
        if (even_on_top) { even_based(); odd_based(); }
        else             { odd_based(); even_based(); }
    }
    if (e != nelem) throw Error(60, "equilateral mesh: element count mismatch");

    std::vector<int> seg, flag;
    auto add = [&](int a, int b, unsigned f) { seg.push_back(a); seg.push_back(b); flag.push_back((int)f); };
    for (int i = 0; i < nx-1; ++i) add(i, i + 1, BOUNDZ1);
    {
        const int b0 = row_start(nz-1), nb = row_count(nz-1) - 1;
        for (int i = 0; i < nb; ++i) add(b0 + i, b0 + i + 1, BOUNDZ0);
    }
    for (int j = 0; j < nz-1; ++j) {
        // side walls, upper node first
        add(row_start(j), row_start(j+1), BOUNDX0);
        add(row_start(j) + row_count(j) - 1, row_start(j+1) + row_count(j+1) - 1, BOUNDX1);
    }
    if ((int)flag.size() != nseg) throw Error(60, "equilateral mesh: segment count mismatch");
    m.segment.resize((size_t)2*nseg);
    for (int q = 0; q < nseg; ++q)
        for (int d = 0; d < 2; ++d) m.segment[(size_t)d*nseg + q] = seg[(size_t)q*2 + d];
    m.segflag = flag;
    m.regattr.assign((size_t)nelem, 0.0);
}

// The regular mesher writes AoS scratch arrays first (as the reference's
// create_rect_node / create_elem_from_cell / create_regular_segments do) and converts to SoA.
void new_mesh_regular(const Config &cfg, HostMesh &m)
{
    if (m.nd == 2) { new_mesh_regular2d(cfg, m); return; }
    const double Lx = cfg.d("mesh.xlength"), Ly = cfg.d("mesh.ylength"), Lz = cfg.d("mesh.zlength");
    const double res = cfg.d("mesh.resolution");
    RegularGrid g;
    // dynearthsol.cxx:127-141
    g.nx = (int)std::round(Lx / res) + 1;
    g.nz = (int)std::round(Lz / res) + 1;
    g.ny = (int)std::round(Ly / res) + 1;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    if (nx < 2 || ny < 2 || nz < 2) throw Error(11, "regular mesh needs at least one cell per direction");
    const int ncell = (nx-1)*(ny-1)*(nz-1);
    const int nnode = nx*ny*nz;
    const int nelem = 5*ncell;
    const int nseg = 4*((nx-1)*(ny-1) + (ny-1)*(nz-1) + (nz-1)*(nx-1));
    cells_of_grid(g);

    // nodes, mesh.cxx:320-347: z runs downward from 0
    std::vector<double> pts((size_t)nnode*3);
    const double dx = Lx / (nx-1), dz = -Lz / (nz-1), dy = Ly / (ny-1);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                size_t idx = (size_t)k + (size_t)j*nz + (size_t)i*ny*nz;
                pts[idx*3] = i*dx; pts[idx*3+1] = j*dy; pts[idx*3+2] = k*dz;
            }

    // elements, mesh.cxx:271-317
    std::vector<int> conn((size_t)nelem*4);
    for (int i = 0; i < nx-1; ++i)
        for (int j = 0; j < ny-1; ++j)
            for (int k = 0; k < nz-1; ++k) {
                size_t idx = (size_t)i*(ny-1)*(nz-1) + (size_t)j*(nz-1) + k;
                int order = (i+j+k) % 2;
                for (int n = 0; n < 5; ++n)
                    for (int q = 0; q < 4; ++q)
                        conn[(idx*5+n)*4 + q] = g.cell[idx*8 + TET_OF_CELL[order][n][q]];
            }

    // boundary segments, mesh.cxx:350-568: each boundary cell face is covered by the outer
    // faces of two of its five tets; (tet, three local nodes) per face and parity.
    std::vector<int> seg((size_t)nseg*3), flag((size_t)nseg);
    int s = 0;
    auto add = [&](size_t cell, int tet, int a, int b, int c, unsigned f) {
        const int *cn = &conn[(cell*5 + tet)*4];
        seg[(size_t)s*3] = cn[a]; seg[(size_t)s*3+1] = cn[b]; seg[(size_t)s*3+2] = cn[c];
        flag[s] = (int)f; ++s;
    };
    auto cellid = [&](int i, int j, int k) { return (size_t)i*(ny-1)*(nz-1) + (size_t)j*(nz-1) + k; };
    for (int i = 0; i < nx-1; ++i)
        for (int j = 0; j < ny-1; ++j) {
            size_t ct = cellid(i, j, 0);            // k = 0 is the top layer (z = 0)
            add(ct, 0, 0, 2, 1, BOUNDZ1);
            add(ct, 1, 0, 2, 1, BOUNDZ1);
            size_t cb = cellid(i, j, nz-2);
            add(cb, 2, 1, 2, 3, BOUNDZ0);
            add(cb, 3, 1, 2, 3, BOUNDZ0);
        }
    for (int j = 0; j < ny-1; ++j)
        for (int k = 0; k < nz-1; ++k) {
            int i = 0;
            int order = (i+j+k) % 2;
            size_t c0 = cellid(i, j, k);
            add(c0, order ? 1 : 0, 0, 1, 3, BOUNDX0);
            add(c0, 2, 0, 2, 1, BOUNDX0);
            i = nx-2;
            order = (i+j+k) % 2;
            size_t c1 = cellid(i, j, k);
            add(c1, order ? 0 : 1, 1, 2, 3, BOUNDX1);
            add(c1, 3, 0, 3, 2, BOUNDX1);
        }
    for (int i = 0; i < nx-1; ++i)
        for (int k = 0; k < nz-1; ++k) {
            int j = 0;
            int order = (i+j+k) % 2;
            size_t c0 = cellid(i, j, k);
            add(c0, 1, 0, 3, 2, BOUNDY0);
            add(c0, order ? 3 : 2, 0, 1, 3, BOUNDY0);
            j = ny-2;
            order = (i+j+k) % 2;
            size_t c1 = cellid(i, j, k);
            if (order) add(c1, 0, 0, 1, 3, BOUNDY1);
            else       add(c1, 0, 1, 2, 3, BOUNDY1);
            if (order) add(c1, 2, 0, 3, 2, BOUNDY1);
            else       add(c1, 3, 0, 2, 1, BOUNDY1);
        }
    if (s != nseg) throw Error(60, "regular mesh: segment count mismatch");

    m.nnode = nnode; m.nelem = nelem; m.nseg = nseg;
    m.coord.resize((size_t)3*nnode);
    for (int n = 0; n < nnode; ++n)
        for (int d = 0; d < 3; ++d) m.coord[(size_t)d*nnode + n] = pts[(size_t)n*3 + d];
    m.conn.resize((size_t)4*nelem);
    for (int e = 0; e < nelem; ++e)
        for (int q = 0; q < 4; ++q) m.conn[(size_t)q*nelem + e] = conn[(size_t)e*4 + q];
    m.segment.resize((size_t)3*nseg);
    for (int q = 0; q < nseg; ++q)
        for (int d = 0; d < 3; ++d) m.segment[(size_t)d*nseg + q] = seg[(size_t)q*3 + d];
    m.segflag = flag;
    m.regattr.assign((size_t)nelem, 0.0);     // mesh.cxx:570-575
}

// discard_internal_segments (mesh.cxx:2672-2693)
void discard_internal_segments(HostMesh &m)
{
    int nseg = m.nseg;
    const int npf = m.nd;
    std::vector<int> seg((size_t)nseg*npf), flag(m.segflag);
    for (int q = 0; q < nseg; ++q)
        for (int d = 0; d < npf; ++d) seg[(size_t)q*npf+d] = m.segment[(size_t)d*m.nseg + q];
    int n = 0;
    while (n < nseg) {
        if ((unsigned)flag[n] & BOUND_ANY) { n++; }
        else {
            nseg--;
            flag[n] = flag[nseg];
            for (int d = 0; d < npf; ++d) seg[(size_t)n*npf+d] = seg[(size_t)nseg*npf+d];
        }
    }
    m.nseg = nseg;
    m.segflag.assign(flag.begin(), flag.begin() + nseg);
    m.segment.resize((size_t)npf*nseg);
    for (int q = 0; q < nseg; ++q)
        for (int d = 0; d < npf; ++d) m.segment[(size_t)d*nseg + q] = seg[(size_t)q*npf+d];
}

struct IdxLess {
    const double *x;
    bool operator()(int l, int r) const { return x[l] < x[r]; }
};

// sortindex (sortindex.hpp:24-32): std::sort of an iota by value
template <typename I>
void sortindex(const std::vector<double> &x, std::vector<I> &idx)
{
    std::iota(idx.begin(), idx.end(), 0);
    const double *px = x.data();
    std::sort(idx.begin(), idx.end(), [px](I l, I r) { return px[l] < px[r]; });
}

} // namespace

// mesh.cxx:2696-2821
void renumbering_mesh(const Config &cfg, HostMesh &m)
{
    const int nnode = m.nnode, nelem = m.nelem, nseg = m.nseg;
    const int nd = m.nd, npe = nd + 1;
    std::vector<double> wn(nnode), we(nelem);
    const double f = 1e-3;
    if (nd == 3) {
        std::vector<double> lengths = {cfg.d("mesh.xlength"), cfg.d("mesh.ylength"), cfg.d("mesh.zlength")};
        std::vector<size_t> idx(3);
        sortindex(lengths, idx);
        int dmin, dmid, dmax;
        if (cfg.i("mesh.meshing_elem_shape") == 0) {
            dmin = (int)idx[0]; dmid = (int)idx[1]; dmax = (int)idx[2];
        } else {
            dmax = 0; dmid = 1; dmin = 2;
        }
        for (int i = 0; i < nnode; i++)
            wn[i] = m.coord[(size_t)dmax*nnode + i] + f * m.coord[(size_t)dmid*nnode + i]
                    + f * f * m.coord[(size_t)dmin*nnode + i];
        for (int i = 0; i < nelem; i++)
            we[i] = wn[m.conn[i]] + wn[m.conn[(size_t)nelem + i]] + wn[m.conn[(size_t)2*nelem + i]]
                    + wn[m.conn[(size_t)3*nelem + i]];
    } else {
        // the !THREED branches of mesh.cxx:2710-2760: no middle axis
        std::vector<double> lengths = {cfg.d("mesh.xlength"), cfg.d("mesh.zlength")};
        std::vector<size_t> idx(2);
        sortindex(lengths, idx);
        int dmin, dmax;
        if (cfg.i("mesh.meshing_elem_shape") == 0) { dmin = (int)idx[0]; dmax = (int)idx[1]; }
        else { dmax = 0; dmin = 1; }
        for (int i = 0; i < nnode; i++)
            wn[i] = m.coord[(size_t)dmax*nnode + i] + f * f * m.coord[(size_t)dmin*nnode + i];
        for (int i = 0; i < nelem; i++)
            we[i] = wn[m.conn[i]] + wn[m.conn[(size_t)nelem + i]] + wn[m.conn[(size_t)2*nelem + i]];
    }

    std::vector<int> nd_idx(nnode), el_idx(nelem);
    sortindex(wn, nd_idx);
    sortindex(we, el_idx);
    std::vector<int> nd_inv(nnode);
    for (int i = 0; i < nnode; i++) nd_inv[nd_idx[i]] = i;

    std::vector<double> coord2(m.coord.size());
    for (int i = 0; i < nnode; i++)
        for (int d = 0; d < nd; ++d)
            coord2[(size_t)d*nnode + i] = m.coord[(size_t)d*nnode + nd_idx[i]];
    m.coord.swap(coord2);

    std::vector<int> conn2(m.conn.size());
    for (int i = 0; i < nelem; i++)
        for (int j = 0; j < npe; ++j)
            conn2[(size_t)j*nelem + i] = nd_inv[m.conn[(size_t)j*nelem + el_idx[i]]];
    m.conn.swap(conn2);

    for (int i = 0; i < nseg; i++)
        for (int j = 0; j < nd; ++j)
            m.segment[(size_t)j*nseg + i] = nd_inv[m.segment[(size_t)j*nseg + i]];

    std::vector<double> reg2(nelem);
    for (int i = 0; i < nelem; i++) reg2[i] = m.regattr[el_idx[i]];
    m.regattr.swap(reg2);
}

// "DESMESH1" / "DESMESH0": 3-D, finished / raw mesher output; "DESMSH21" / "DESMSH20": the same in 2-D
static const char kMeshMagic[8] = {'D','E','S','M','E','S','H','1'};
static const char kMeshMagic2d[8] = {'D','E','S','M','S','H','2','1'};

void save_mesh_file(const std::string &path, const HostMesh &m)
{
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw Error(20, "cannot open mesh file for writing: " + path);
    int hdr[3] = {m.nnode, m.nelem, m.nseg};
    bool ok = std::fwrite(m.nd == 2 ? kMeshMagic2d : kMeshMagic, 1, 8, fp) == 8 && std::fwrite(hdr, sizeof(int), 3, fp) == 3
        && std::fwrite(m.coord.data(), sizeof(double), m.coord.size(), fp) == m.coord.size()
        && std::fwrite(m.conn.data(), sizeof(int), m.conn.size(), fp) == m.conn.size()
        && std::fwrite(m.segment.data(), sizeof(int), m.segment.size(), fp) == m.segment.size()
        && std::fwrite(m.segflag.data(), sizeof(int), m.segflag.size(), fp) == m.segflag.size();
    std::fclose(fp);
    if (!ok) throw Error(21, "write failed: " + path);
}

// DESMESH1: a finished mesh (after create_new_mesh).  DESMESH0: the raw output of the
// reference mesher (oracle/ref_tetmesh) incl. region attributes; returns true for those so
// the caller finishes the job (discard internal segments, renumber).
bool load_mesh_file_raw(const std::string &path, HostMesh &m)
{
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw Error(20, "cannot open mesh file: " + path);
    char magic[8]; int hdr[3];
    bool ok = std::fread(magic, 1, 8, fp) == 8
              && (std::memcmp(magic, kMeshMagic, 7) == 0 || std::memcmp(magic, kMeshMagic2d, 7) == 0)
              && (magic[7] == '0' || magic[7] == '1') && std::fread(hdr, sizeof(int), 3, fp) == 3;
    const bool raw = ok && magic[7] == '0';
    if (ok && (std::memcmp(magic, kMeshMagic2d, 7) == 0 ? 2 : 3) != m.nd) {
        std::fclose(fp);
        throw Error(30, "mesh file is for the other dimension: " + path);
    }
    if (ok) {
        const int nd = m.nd;
        m.nnode = hdr[0]; m.nelem = hdr[1]; m.nseg = hdr[2];
        m.coord.resize((size_t)nd*m.nnode); m.conn.resize((size_t)(nd+1)*m.nelem);
        m.segment.resize((size_t)nd*m.nseg); m.segflag.resize((size_t)m.nseg);
        ok = std::fread(m.coord.data(), sizeof(double), m.coord.size(), fp) == m.coord.size()
          && std::fread(m.conn.data(), sizeof(int), m.conn.size(), fp) == m.conn.size()
          && std::fread(m.segment.data(), sizeof(int), m.segment.size(), fp) == m.segment.size()
          && std::fread(m.segflag.data(), sizeof(int), m.segflag.size(), fp) == m.segflag.size();
    }
    m.regattr.assign((size_t)m.nelem, 0.0);
    if (ok && raw)
        ok = std::fread(m.regattr.data(), sizeof(double), m.regattr.size(), fp) == m.regattr.size();
    std::fclose(fp);
    if (!ok) throw Error(12, "malformed mesh file: " + path);
    return raw;
}

void load_mesh_file(const std::string &path, HostMesh &m)
{
    if (load_mesh_file_raw(path, m))
        throw Error(12, "raw mesher output needs a .cfg to be finished: " + path);
}

// mesh.cxx:3460-3506
void create_new_mesh(const Config &cfg, HostMesh &m, const std::string &mesh_file)
{
    if (!mesh_file.empty()) {
        // a mesh written after create_new_mesh() by the reference mesher (tools/), i.e.
        // already renumbered and with internal segments discarded
        if (load_mesh_file_raw(mesh_file, m)) {
            // raw mesher output: finish as create_new_mesh does (mesh.cxx:3499-3502)
            if (cfg.b("mesh.is_discarding_internal_segments"))
                discard_internal_segments(m);
            renumbering_mesh(cfg, m);
        }
        return;
    }
    const int opt = cfg.i("mesh.meshing_option");
    const int shape = cfg.i("mesh.meshing_elem_shape");
    if (shape >= 1 && opt != 1)
        throw Error(30, "mesh.meshing_elem_shape >= 1 is only for mesh.meshing_option == 1.");
    if (shape == 2 && m.nd == 3)
        throw Error(30, "mesh.meshing_elem_shape == 2 is not available in 3D.");
    if (opt == 1 && shape == 2) {
        new_mesh_equilateral2d(cfg, m);
    } else if (opt == 1 && shape == 1) {
        new_mesh_regular(cfg, m);
    } else if (opt == 1 || opt == 2 || opt == 90 || opt == 91) {
        throw Error(31, "this meshing_option needs TetGen / Triangle, host-side libraries of the "
                        "reference; pass a mesh file generated with them (see DESIGN.md)");
    } else if (opt == 95) {
        // mesh.cxx:3482-3489 without USEEXODUS
        throw Error(31, "Error: Install Exodus library and rebuild with 'useexo' turned on in Makefile.");
    } else {
        throw Error(11, "Error: unknown meshing option");
    }
    if (cfg.b("mesh.is_discarding_internal_segments"))
        discard_internal_segments(m);
    renumbering_mesh(cfg, m);
}

namespace {

// a facet's node set: three nodes in 3-D, two (+ -1) in 2-D (OrderedInt, mesh.cxx)
struct Tri {
    int a, b, c;
    Tri(int x, int y, int z) {
        int v[3] = {x, y, z};
        std::sort(v, v + 3);
        a = v[0]; b = v[1]; c = v[2];
    }
    bool operator==(const Tri &o) const { return a == o.a && b == o.b && c == o.c; }
};
struct TriHash {
    size_t operator()(const Tri &t) const {
        size_t h = (size_t)t.a * 0x9E3779B97F4A7C15ull;
        h ^= (size_t)t.b + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h ^= (size_t)t.c + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        return h;
    }
};

// bc.cxx:24-54
void facet_normal(const HostMesh &m, int e, int f, double normal[3])
{
    double fc[3][3];
    for (int j = 0; j < m.nd; ++j) {
        int n = m.conn[(size_t)node_of_facet(m.nd, f, j)*m.nelem + e];
        for (int d = 0; d < m.nd; ++d) fc[j][d] = m.coord[(size_t)d*m.nnode + n];
    }
    if (m.nd == 2) {
        // the normal vector to the edge, pointing outward (bc.cxx:42-50)
        double v01[2];
        for (int i = 0; i < 2; ++i) v01[i] = fc[1][i] - fc[0][i];
        normal[0] = v01[1];
        normal[1] = -v01[0];
        return;
    }
    double v01[3], v02[3];
    for (int i = 0; i < 3; ++i) { v01[i] = fc[1][i] - fc[0][i]; v02[i] = fc[2][i] - fc[0][i]; }
    normal[0] = (v01[1]*v02[2] - v01[2]*v02[1]) / 2;
    normal[1] = (v01[2]*v02[0] - v01[0]*v02[2]) / 2;
    normal[2] = (v01[0]*v02[1] - v01[1]*v02[0]) / 2;
}

} // namespace

void build_topology(HostMesh &m, const int vbc_types[DES_NBDRY])
{
    const int nnode = m.nnode, nelem = m.nelem, nseg = m.nseg;
    const int nd = m.nd, npe = nd + 1, npf = nd;
    // node k of facet f of element e, or -1 past the facet's last node
    auto fnode = [&](int e, int f, int k) { return k < npf ? m.conn[(size_t)node_of_facet(nd, f, k)*nelem + e] : -1; };

    // create_boundary_flags (mesh.cxx:2824-2851)
    m.bcflag.assign((size_t)nnode, 0u);
    for (int i = 0; i < nseg; ++i)
        for (int j = 0; j < npf; ++j)
            m.bcflag[m.segment[(size_t)j*nseg + i]] |= (unsigned)m.segflag[i];

    // create_boundary_nodes (mesh.cxx:2854-2880)
    for (int j = 0; j < DES_NBDRY; ++j) m.bnodes[j].clear();
    for (int i = 0; i < nnode; ++i)
        for (int j = 0; j < DES_NBDRY; ++j)
            if (m.bcflag[i] & (1u << j)) m.bnodes[j].push_back(i);

    // create_boundary_facets (mesh.cxx:3161-3283).  The reference searches every element
    // for every segment; a facet lookup table gives the same (element, facet) pairs.
    std::unordered_map<Tri, std::pair<int,int>, TriHash> facet_of;
    for (int e = 0; e < nelem; ++e)
        for (int f = 0; f < npe; ++f) {
            int n0 = fnode(e, f, 0), n1 = fnode(e, f, 1), n2 = fnode(e, f, 2);
            if ((m.bcflag[n0] & m.bcflag[n1] & (n2 >= 0 ? m.bcflag[n2] : ~0u)) == 0u) continue;
            Tri t(n0, n1, n2);
            if (!facet_of.count(t)) facet_of[t] = std::make_pair(e, f);   // first (e, f) wins
        }
    std::vector<std::pair<int,int> > bf[DES_NBDRY];
    for (int i = 0; i < nseg; ++i) {
        unsigned flag = (unsigned)m.segflag[i];
        if ((flag & BOUND_ANY) == 0) continue;
        Tri t(m.segment[i], m.segment[(size_t)nseg + i], nd == 3 ? m.segment[(size_t)2*nseg + i] : -1);
        std::unordered_map<Tri, std::pair<int,int>, TriHash>::const_iterator it = facet_of.find(t);
        bool found = false;
        if (it != facet_of.end()) {
            int e = it->second.first, f = it->second.second;
            unsigned facet_flag = m.bcflag[fnode(e, f, 0)] & m.bcflag[fnode(e, f, 1)]
                                & (nd == 3 ? m.bcflag[fnode(e, f, 2)] : ~0u);
            if (flag & facet_flag)
                for (int k = 0; k < DES_NBDRY; ++k)
                    if (flag == (1u << k)) { bf[k].push_back(it->second); found = true; break; }
        }
        if (!found)
            throw Error(61, "Error: " + std::to_string(i) + "-th segment is not on any element");
    }
    for (int k = 0; k < DES_NBDRY; ++k) {
        std::sort(bf[k].begin(), bf[k].end(),
                  [](const std::pair<int,int> &a, const std::pair<int,int> &b) { return a.first < b.first; });
        m.bfacet_elem[k].resize(bf[k].size());
        m.bfacet_facet[k].resize(bf[k].size());
        for (size_t q = 0; q < bf[k].size(); ++q) {
            m.bfacet_elem[k][q] = bf[k][q].first;
            m.bfacet_facet[k][q] = bf[k][q].second;
        }
    }
    const int etop = (int)bf[iboundz1].size();
    m.conn_surf.assign((size_t)npe*etop, 0);
    for (int i = 0; i < etop; ++i)
        for (int j = 0; j < npf; ++j)
            m.conn_surf[(size_t)j*etop + i] = fnode(bf[iboundz1][i].first, bf[iboundz1][i].second, j);

    // create_support (mesh.cxx:3287-3329)
    m.sup_idx.assign((size_t)nnode + 1, 0);
    for (int e = 0; e < nelem; ++e)
        for (int i = 0; i < npe; ++i) m.sup_idx[m.conn[(size_t)i*nelem + e] + 1]++;
    for (int n = 1; n <= nnode; ++n) m.sup_idx[n] += m.sup_idx[n-1];
    m.sup_arr.resize((size_t)m.sup_idx[nnode]);
    m.sup_lidx.resize((size_t)m.sup_idx[nnode]);
    {
        std::vector<int> cursor(m.sup_idx.begin(), m.sup_idx.end() - 1);
        for (int e = 0; e < nelem; ++e)
            for (int i = 0; i < npe; ++i) {
                int slot = cursor[m.conn[(size_t)i*nelem + e]]++;
                m.sup_arr[slot] = e;
                m.sup_lidx[slot] = i;
            }
    }

    // create_top_elems (mesh.cxx:2882-2933) and create_surface_info (mesh.cxx:3031-3103):
    // surface nodes ordered by x with std::sort, as the reference does
    const std::vector<int> &top_tmp = m.bnodes[iboundz1];
    const int ntop = (int)top_tmp.size();
    std::vector<int> top_ind(ntop);
    std::vector<double> top_x(ntop);
    for (int i = 0; i < ntop; ++i) { top_ind[i] = i; top_x[i] = m.coord[top_tmp[i]]; }
    std::sort(top_ind.begin(), top_ind.end(), [&](const int &a, const int &b) { return top_x[a] < top_x[b]; });
    m.top_nodes.resize(ntop);
    for (int i = 0; i < ntop; ++i) m.top_nodes[i] = top_tmp[top_ind[i]];

    std::set<int> elem_set;
    for (int i = 0; i < ntop; ++i) {
        int n = m.top_nodes[i];
        for (int k = m.sup_idx[n]; k < m.sup_idx[n+1]; ++k) elem_set.insert(m.sup_arr[k]);
    }
    m.top_elems.assign(elem_set.begin(), elem_set.end());

    std::unordered_map<int,int> arctop;
    for (int i = 0; i < ntop; ++i) arctop[m.top_nodes[i]] = i;
    m.elem_and_nodes.assign((size_t)npf*etop, 0);
    for (int i = 0; i < etop; ++i)
        for (int k = 0; k < npf; ++k)
            m.elem_and_nodes[(size_t)k*etop + i] = arctop[m.conn_surf[(size_t)k*etop + i]];
    // create_support_surf (mesh.cxx:2937-2962)
    m.ssup_idx.assign((size_t)ntop + 1, 0);
    for (int i = 0; i < etop; ++i)
        for (int k = 0; k < npf; ++k) m.ssup_idx[m.elem_and_nodes[(size_t)k*etop + i] + 1]++;
    for (int n = 1; n <= ntop; ++n) m.ssup_idx[n] += m.ssup_idx[n-1];
    m.ssup_arr.resize((size_t)m.ssup_idx[ntop]);
    {
        std::vector<int> cursor(m.ssup_idx.begin(), m.ssup_idx.end() - 1);
        for (int i = 0; i < etop; ++i)
            for (int k = 0; k < npf; ++k)
                m.ssup_arr[cursor[m.elem_and_nodes[(size_t)k*etop + i]]++] = i;
    }

    // create_boundary_normals (bc.cxx:94-224)
    m.bnormals.assign((size_t)nd*DES_NBDRY, 0.0);
    for (int i = 0; i < DES_NBDRY; i++) {
        const size_t nf = m.bfacet_elem[i].size();
        if (nf == 0) continue;
        for (size_t j = 0; j < nf; ++j) {
            double normal[3];
            facet_normal(m, m.bfacet_elem[i][j], m.bfacet_facet[i][j], normal);
            double len = 0;
            for (int d = 0; d < nd; d++) len += normal[d]*normal[d];
            len = std::sqrt(len);
            for (int d = 0; d < nd; d++) normal[d] = normal[d] / len;
            if (j == 0) {
                for (int d = 0; d < nd; d++) m.bnormals[(size_t)d*DES_NBDRY + i] = normal[d];
                if (i < iboundn0) break;
            } else {
                const double eps2 = 1e-12;
                double diff2 = 0;
                for (int d = 0; d < nd; d++) {
                    double t = m.bnormals[(size_t)d*DES_NBDRY + i] - normal[d];
                    diff2 += t * t;
                }
                if (diff2 > eps2)
                    throw Error(42, "Error: boundary " + std::to_string(i) + " is curved.");
            }
        }
    }
    std::fill_n(m.edge_slot, DES_NBDRY*DES_NBDRY, -1);
    m.edge_vec.clear();
    for (int i = 0; i < DES_NBDRY; i++) {
        if (m.bfacet_elem[i].empty()) continue;
        const double eps = 1e-15;
        for (int j = i+1; j < DES_NBDRY; j++) {
            if (m.bfacet_elem[j].empty()) continue;
            if (nd == 2) {
                // bc.cxx:181-184: the "edge" two boundaries share in 2-D is the vertical
                m.edge_slot[i*DES_NBDRY + j] = (int)(m.edge_vec.size() / 2);
                m.edge_vec.push_back(0); m.edge_vec.push_back(1);
                continue;
            }
            const double ni[3] = {m.bnormals[i], m.bnormals[DES_NBDRY+i], m.bnormals[2*DES_NBDRY+i]};
            const double nj[3] = {m.bnormals[j], m.bnormals[DES_NBDRY+j], m.bnormals[2*DES_NBDRY+j]};
            double s[3];
            if (std::abs(ni[2]) < eps && std::abs(nj[2]) < eps) {
                s[0] = s[1] = 0; s[2] = 1;
            } else {
                s[0] = ni[1]*nj[2] - ni[2]*nj[1];
                s[1] = ni[2]*nj[0] - ni[0]*nj[2];
                s[2] = ni[0]*nj[1] - ni[1]*nj[0];
            }
            m.edge_slot[i*DES_NBDRY + j] = (int)(m.edge_vec.size() / 3);
            m.edge_vec.push_back(s[0]); m.edge_vec.push_back(s[1]); m.edge_vec.push_back(s[2]);
        }
    }
    (void)vbc_types;
}

des_mesh HostMesh::view() const
{
    des_mesh v;
    std::memset(&v, 0, sizeof(v));
    v.nnode = nnode; v.nelem = nelem;
    v.connectivity = conn.data();
    v.support_idx = sup_idx.data(); v.support_arr = sup_arr.data(); v.support_lidx = sup_lidx.data();
    v.bcflag = bcflag.data();
    for (int i = 0; i < DES_NBDRY; ++i) {
        v.nbfacets[i] = (int)bfacet_elem[i].size();
        v.bfacet_elem[i] = bfacet_elem[i].data();
        v.bfacet_facet[i] = bfacet_facet[i].data();
        v.nbnodes[i] = (int)bnodes[i].size();
        v.bnodes[i] = bnodes[i].data();
    }
    v.bnormals = bnormals.data();
    v.edge_vec = edge_vec.data();
    v.nedge = (int)(edge_vec.size() / nd);
    std::memcpy(v.edge_slot, edge_slot, sizeof(edge_slot));
    v.ntop = (int)top_nodes.size();
    v.etop = (int)bfacet_elem[iboundz1].size();
    v.ntop_elems = (int)top_elems.size();
    v.top_nodes = top_nodes.data();
    v.elem_and_nodes = elem_and_nodes.data();
    v.connectivity_surface = conn_surf.data();
    v.support_surf_idx = ssup_idx.data();
    v.support_surf_arr = ssup_arr.data();
    v.top_elems = top_elems.data();
    v.coord = coord.empty() ? nullptr : coord.data();         // layout hint for the engine
    v.owned_begin = 0; v.owned_end = nnode;
    return v;
}

} // namespace des
