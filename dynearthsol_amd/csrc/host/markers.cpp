// markers.cpp -- the marker work that stays on the host between remeshings:
//   * markers.init_marker_option = 2: MarkerSet::regularly_spaced_markers (markerset.cxx:556-663),
//   * mat.phase_change_option = 1 / 101: phase_changes (phasechanges.cxx:109-152) every 10 steps
//     (dynearthsol.cxx:881-894), on the marker set the host keeps, with the nodal coordinates and
//     temperatures the loop downloads; the device only ever sees the per-element counts
//     (DES_F_ELEMMARKERS, dirty flag -> k_props).
// Not here: hydration (control.has_hydration_processes: the second, hydrous marker set), refused by
// the .cfg front-end; without it var.hydrous_elemmarkers does not exist in the reference
// (mesh.cxx:3439-3442), so the mantle -> serpentinite branch of phasechanges.cxx:80-87 has no
// hydrous markers to find and is never taken.
#include "des_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace des {

namespace {

// Barycentric_transformation (barycentric-fn.cxx:224-268 coefficients, 135-147 transform):
// eta[0..2] of point x in tetrahedron (a, b, c, d) of volume `volume`; eta[3] = 1 - sum.
void bary3(const double a[3], const double b[3], const double c[3], const double d[3], double volume,
           const double x[3], double eta[4])
{
    const double det = 6 * volume;
    double cf[4][3];
    cf[0][0] = (b[0] * (c[1]*d[2] - d[1]*c[2]) + c[0] * (d[1]*b[2] - b[1]*d[2]) + d[0] * (b[1]*c[2] - c[1]*b[2])) / det;
    cf[0][1] = (a[0] * (d[1]*c[2] - c[1]*d[2]) + c[0] * (a[1]*d[2] - d[1]*a[2]) + d[0] * (c[1]*a[2] - a[1]*c[2])) / det;
    cf[0][2] = (a[0] * (b[1]*d[2] - d[1]*b[2]) + b[0] * (d[1]*a[2] - a[1]*d[2]) + d[0] * (a[1]*b[2] - b[1]*a[2])) / det;
    cf[1][0] = ((d[1] - b[1]) * (c[2] - b[2]) - (c[1] - b[1]) * (d[2] - b[2])) / det;
    cf[1][1] = ((c[1] - a[1]) * (d[2] - c[2]) - (d[1] - c[1]) * (c[2] - a[2])) / det;
    cf[1][2] = ((b[1] - d[1]) * (a[2] - d[2]) - (a[1] - d[1]) * (b[2] - d[2])) / det;
    cf[2][0] = ((d[2] - b[2]) * (c[0] - b[0]) - (c[2] - b[2]) * (d[0] - b[0])) / det;
    cf[2][1] = ((c[2] - a[2]) * (d[0] - c[0]) - (d[2] - c[2]) * (c[0] - a[0])) / det;
    cf[2][2] = ((b[2] - d[2]) * (a[0] - d[0]) - (a[2] - d[2]) * (b[0] - d[0])) / det;
    cf[3][0] = ((d[0] - b[0]) * (c[1] - b[1]) - (c[0] - b[0]) * (d[1] - b[1])) / det;
    cf[3][1] = ((c[0] - a[0]) * (d[1] - c[1]) - (d[0] - c[0]) * (c[1] - a[1])) / det;
    cf[3][2] = ((b[0] - d[0]) * (a[1] - d[1]) - (a[0] - d[0]) * (b[1] - d[1])) / det;
    for (int k = 0; k < 3; ++k) {
        eta[k] = cf[0][k];
        for (int i = 0; i < 3; ++i) eta[k] += cf[i + 1][k] * x[i];
    }
    double tmp = 1;
    for (int k = 0; k < 3; ++k) tmp -= eta[k];
    eta[3] = tmp;
}

// Barycentric_transformation::is_inside, 3-D tolerance (barycentric-fn.cxx:182-197)
bool inside3(const double r[4])
{
    const double tolerance = 5e-11;
    return r[0] >= -tolerance && r[1] >= -tolerance && r[2] >= -tolerance && (r[0] + r[1] + r[2]) <= 1 + tolerance;
}

// the same for a triangle (barycentric-fn.cxx:230-244 coefficients, 149-155 transform; is_inside :198-204)
void bary2(const double a[2], const double b[2], const double c[2], double area, const double x[2], double eta[3])
{
    const double det = 2 * area;
    double cf[3][2];
    cf[0][0] = (b[0]*c[1] - b[1]*c[0]) / det;
    cf[0][1] = (c[0]*a[1] - c[1]*a[0]) / det;
    cf[1][0] = (b[1] - c[1]) / det;
    cf[1][1] = (c[1] - a[1]) / det;
    cf[2][0] = (c[0] - b[0]) / det;
    cf[2][1] = (a[0] - c[0]) / det;
    for (int k = 0; k < 2; ++k) {
        eta[k] = cf[0][k];
        for (int i = 0; i < 2; ++i) eta[k] += cf[i + 1][k] * x[i];
    }
    double tmp = 1;
    for (int k = 0; k < 2; ++k) tmp -= eta[k];
    eta[2] = tmp;
}

bool inside2(const double r[3])
{
    const double tolerance = 1e-12;
    return r[0] >= -tolerance && r[1] >= -tolerance && (r[0] + r[1]) <= 1 + tolerance;
}

// triangle_area (geometry.cxx:77-93, !THREED)
double tri_area(const double *a, const double *b, const double *c)
{
    double ab0 = b[0] - a[0], ab1 = b[1] - a[1];
    double ac0 = c[0] - a[0], ac1 = c[1] - a[1];
    return std::fabs(ab0*ac1 - ab1*ac0) / 2;
}

// tetrahedron_volume (geometry.cxx:36-56)
double tet_volume(const double *a, const double *b, const double *c, const double *d)
{
    double x01 = a[0] - b[0], x12 = b[0] - c[0], x23 = c[0] - d[0];
    double y01 = a[1] - b[1], y12 = b[1] - c[1], y23 = c[1] - d[1];
    double z01 = a[2] - b[2], z12 = b[2] - c[2], z23 = c[2] - d[2];
    return (x01*(y23*z12 - y12*z23) + x12*(y01*z23 - y23*z01) + x23*(y12*z01 - y01*z12)) / 6;
}

// k nearest element centroids of a point, nearest first (what the reference asks of nanoflann's
// KD-tree, markerset.cxx:601-634: an exact L2 k-NN query; ties between equidistant centroids are
// broken here by element index -- nanoflann leaves them to its traversal order).  Uniform grid of
// centroid buckets, rings searched outwards until the k-th distance is covered.
struct CentroidGrid {
    int n[3];
    double lo[3], h;
    std::vector<int> start, items;
    std::vector<double> cen;                      // [3][ne]
    int ne;

    void build(const HostMesh &m, double cell)
    {
        ne = m.nelem;
        const int nn = m.nnode, nd = m.nd, npe = nd + 1;
        cen.assign((size_t)3 * ne, 0.0);
        double hi[3];
        for (int d = 0; d < 3; ++d) { lo[d] = 1e300; hi[d] = -1e300; }
        // (a 2-D mesh: {x, z} in the grid's first and last dimension, one layer of cells in between)
        for (int e = 0; e < ne; ++e)
            for (int d = 0; d < 3; ++d) {
                double s = 0;
                if (nd == 3 || d != 1) {
                    const int dm = nd == 3 ? d : (d == 0 ? 0 : 1);
                    // average_nodal_to_elem (utils): sum of the element's nodes / their number
                    for (int i = 0; i < npe; ++i) s += m.coord[(size_t)dm*nn + m.conn[(size_t)i*ne + e]];
                    s /= npe;
                }
                cen[(size_t)d*ne + e] = s;
                lo[d] = std::min(lo[d], s); hi[d] = std::max(hi[d], s);
            }
        h = cell;
        for (int d = 0; d < 3; ++d) n[d] = std::max(1, (int)((hi[d] - lo[d]) / h) + 1);
        while ((double)n[0] * n[1] * n[2] > 4.0 * ne + 64) { h *= 1.5; for (int d = 0; d < 3; ++d) n[d] = std::max(1, (int)((hi[d] - lo[d]) / h) + 1); }
        start.assign((size_t)n[0] * n[1] * n[2] + 1, 0);
        std::vector<int> cellof((size_t)ne);
        for (int e = 0; e < ne; ++e) { cellof[e] = cell_of(&cen[e], ne); ++start[cellof[e] + 1]; }
        for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
        items.resize((size_t)ne);
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int e = 0; e < ne; ++e) items[fill[cellof[e]]++] = e;
    }
    int clampi(double x, int d) const { int i = (int)std::floor((x - lo[d]) / h); return i < 0 ? 0 : (i >= n[d] ? n[d] - 1 : i); }
    int cell_of(const double *c, int stride) const { return (clampi(c[2*stride], 2) * n[1] + clampi(c[stride], 1)) * n[0] + clampi(c[0], 0); }

    void knn(const double x[3], int k, std::vector<std::pair<double, int> > &out) const
    {
        out.clear();
        const int c[3] = {clampi(x[0], 0), clampi(x[1], 1), clampi(x[2], 2)};
        const int rmax = std::max(n[0], std::max(n[1], n[2]));
        for (int r = 0; r <= rmax; ++r) {
            for (int kz = std::max(0, c[2] - r); kz <= std::min(n[2] - 1, c[2] + r); ++kz)
                for (int ky = std::max(0, c[1] - r); ky <= std::min(n[1] - 1, c[1] + r); ++ky)
                    for (int kx = std::max(0, c[0] - r); kx <= std::min(n[0] - 1, c[0] + r); ++kx) {
                        if (std::max(std::abs(kx - c[0]), std::max(std::abs(ky - c[1]), std::abs(kz - c[2]))) != r) continue;   // the shell only
                        const size_t cell = ((size_t)kz * n[1] + ky) * n[0] + kx;
                        for (int q = start[cell]; q < start[cell + 1]; ++q) {
                            const int e = items[q];
                            double d2 = 0;
                            for (int d = 0; d < 3; ++d) { const double t = cen[(size_t)d*ne + e] - x[d]; d2 += t * t; }
                            out.push_back(std::make_pair(d2, e));
                        }
                    }
            if ((int)out.size() >= k) {
                // everything outside the searched cube is at least r*h away (x may lie outside the grid: then more)
                std::partial_sort(out.begin(), out.begin() + k, out.end());
                const double reach = (double)r * h;
                if (out[k - 1].first <= reach * reach || r == rmax) break;
            }
        }
        if ((int)out.size() > k) { std::partial_sort(out.begin(), out.begin() + k, out.end()); out.resize(k); }
        else std::sort(out.begin(), out.end());
    }
};

// MarkerSet::initial_mattype with the marker's position known (markerset.cxx:666-716)
int initial_mattype_at(const des_params &p, const HostMesh &m, int mattype_option, const std::vector<double> &layer_mt,
                       const std::vector<double> &depths, int e, double z)
{
    if (mattype_option == 0) {
        const int mt = (int)m.regattr[e];
        if (mt < 0 || mt >= p.nmat) throw Error(11, "region attribute is not a valid material");
        return mt;
    }
    int mt = (int)layer_mt[layer_mt.size() - 1];
    for (size_t i = 0; i < depths.size(); ++i)
        if (z >= -p.zlength * depths[i]) { mt = (int)layer_mt[i]; break; }
    return mt;
}

} // namespace

// MarkerSet::regularly_spaced_markers (markerset.cxx:556-663) + the check of the constructor
// (:56-66) that no element is left without a marker
void regularly_spaced_markers(const Config &cfg, const des_params &p, const HostMesh &m, HostFields &f)
{
    const int nn = m.nnode, ne = m.nelem, nmat = p.nmat, nd = m.nd, npe = nd + 1;
    const int d = (int)(cfg.d("markers.init_marker_spacing") * cfg.d("mesh.resolution"));    // `const int d`, as in the reference
    if (d <= 0) throw Error(11, "markers.init_marker_spacing * mesh.resolution must be at least 1 m");
    const int mattype_option = cfg.i("ic.mattype_option");
    if (mattype_option != 0 && mattype_option != 1) throw Error(11, "Error: unknown ic.mattype_option");
    std::vector<double> layer_mt, depths;
    if (mattype_option == 1) {
        const int nlayers = cfg.i("ic.num_mattype_layers");
        layer_mt = cfg.list("ic.layer_mattypes", nlayers);
        depths = cfg.list("ic.mattype_layer_depths", nlayers - 1);
        if (!std::is_sorted(depths.begin(), depths.end()))
            throw Error(11, "Error: the content of ic.mattype_layer_depths is not ordered from small to big values.");
    }
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int k = 0; k < nd; ++k) {
        lo[k] = hi[k] = m.coord[(size_t)k*nn];
        for (int i = 1; i < nn; ++i) { lo[k] = std::min(lo[k], m.coord[(size_t)k*nn + i]); hi[k] = std::max(hi[k], m.coord[(size_t)k*nn + i]); }
    }
    // (!THREED: ny = 1, the last coordinate is z -- markerset.cxx:577-590)
    const double xlength = hi[0] - lo[0], ylength = nd == 3 ? hi[1] - lo[1] : 0.0, zlength = hi[nd-1] - lo[nd-1];
    const int nx = (int)(xlength / d + 1), ny = nd == 3 ? (int)(ylength / d + 1) : 1, nz = (int)(zlength / d + 1);
    const double x0 = lo[0] + 0.5 * (xlength - (nx - 1) * d), y0 = nd == 3 ? lo[1] + 0.5 * (ylength - (ny - 1) * d) : 0.0,
                 z0 = lo[nd-1] + 0.5 * (zlength - (nz - 1) * d);
    const long long num_markers = (long long)nx * ny * nz;
    if (num_markers > 2000000000LL) throw Error(52, "too many markers");

    CentroidGrid grid;
    grid.build(m, std::max((double)d, cfg.d("mesh.resolution")));
    const int k = std::min(20, ne);
    std::vector<double> vol((size_t)ne);
    auto node = [&](int e, int i, double out[3]) {
        const int n = m.conn[(size_t)i*ne + e];
        for (int k = 0; k < nd; ++k) out[k] = m.coord[(size_t)k*nn + n];
    };
    for (int e = 0; e < ne; ++e) {
        double a[3], b[3], c[3], dd[3];
        node(e, 0, a); node(e, 1, b); node(e, 2, c);
        if (nd == 3) { node(e, 3, dd); vol[e] = tet_volume(a, b, c, dd); }
        else vol[e] = tri_area(a, b, c);
    }

    HostMarkers &mk = f.markers;
    mk = HostMarkers();
    f.elemmarkers.assign((size_t)ne * nmat, 0);
    std::vector<double> eta_aos;                  // [marker][npe] while the count is unknown
    std::vector<std::pair<double, int> > near;
    for (long long n = 0; n < num_markers; ++n) {
        const int ix = (int)(n % nx), iy = (int)((n / nx) % ny), iz = (int)(n / ((long long)nx * ny));
        const double x[3] = {x0 + ix * d, y0 + iy * d, z0 + iz * d};         // (grid space: {x, y or 0, z})
        grid.knn(x, k, near);
        for (size_t j = 0; j < near.size(); ++j) {
            const int e = near[j].second;
            double a[3], b[3], c[3], dd[3], eta[4];
            node(e, 0, a); node(e, 1, b); node(e, 2, c);
            if (nd == 3) {
                node(e, 3, dd);
                bary3(a, b, c, dd, vol[e], x, eta);
                if (!inside3(eta)) continue;
            } else {
                const double x2[2] = {x[0], x[2]};
                bary2(a, b, c, vol[e], x2, eta);
                if (!inside2(eta)) continue;
            }
            const int mt = initial_mattype_at(p, m, mattype_option, layer_mt, depths, e, x[2]);
            eta_aos.insert(eta_aos.end(), eta, eta + npe);
            mk.elem.push_back(e); mk.mattype.push_back(mt); mk.id.push_back(mk.nmarkers);
            ++mk.nmarkers;
            ++f.elemmarkers[(size_t)e*nmat + mt];
            break;
        }
        // not found: x is outside the domain (the domain is not rectangular) -- no marker
    }
    const size_t nm = (size_t)mk.nmarkers;
    mk.last_id = mk.nmarkers;
    mk.reserved_space = (int)(num_markers * 2.0);                // over_alloc_ratio, markerset.cxx:25, 590
    mk.eta.assign((size_t)npe * nm, 0.0);
    for (size_t i = 0; i < nm; ++i) for (int j = 0; j < npe; ++j) mk.eta[(size_t)j*nm + i] = eta_aos[(size_t)npe*i + j];
    mk.genesis.assign(nm, 0); mk.time.assign(nm, 0.0); mk.z.assign(nm, 0.0); mk.distance.assign(nm, 0.0); mk.slope.assign(nm, 0.0);
    for (int e = 0; e < ne; ++e) {
        int cnt = 0;
        for (int i = 0; i < nmat; ++i) cnt += f.elemmarkers[(size_t)e*nmat + i];
        if (cnt <= 0)
            throw Error(52, "Error: no marker in element #" + std::to_string(e) + ". Please increase the number of markers.");
    }
}

namespace {

// simple_subduction (phasechanges.cxx:10-90): metamorphic transitions of the eight-material set
int simple_subduction(int current_mt, double Z, double P, double T)
{
    const int mt_mantle = 0, mt_serpentinized_mantle = 1, mt_oceanic_crust = 2, mt_eclogite = 3, mt_sediment = 4, mt_schist = 5;
    int new_mt = current_mt;
    switch (current_mt) {
    case mt_oceanic_crust: {                       // basalt -> eclogite (Hacker, 1996)
        const double min_eclogite_T = 500 + 273;
        const double transition_pressure = -0.3e9 + 2.2e6 * T;
        if (T > min_eclogite_T && P > transition_pressure) new_mt = mt_eclogite;
        break;
    }
    case mt_sediment: {                            // sediment -> schist / gneiss (Nichols et al., 1994)
        const double min_schist_T = 650 + 273;
        const double min_schist_Z = -20e3;
        if (T > min_schist_T && Z < min_schist_Z) new_mt = mt_schist;
        break;
    }
    case mt_serpentinized_mantle: {                // serpentinite -> normal mantle (Ulmer and Trommsdorff, 1995)
        const double transition_pressure = 2.1e9 + (7.5e9 - 2.1e9) * (T - (730 + 273)) / (500 - 730);
        const double min_serpentine_T = 550 + 273;
        if (T > min_serpentine_T && P > transition_pressure) new_mt = mt_mantle;
        break;
    }
    case mt_mantle:
        // -> serpentinite needs a hydrous marker in the element: none without hydration processes
        break;
    }
    return new_mt;
}

} // namespace

// phase_changes (phasechanges.cxx:109-152) with the current nodal coordinates / temperatures
// (SoA [nd][nnode], [nnode]).  Moves the markers' material and the per-element counts; returns the
// number of markers that changed.
int phase_changes(const Config &cfg, const des_params &p, const HostMesh &m, HostFields &f,
                  const double *coord, const double *temperature)
{
    const int option = cfg.i("mat.phase_change_option");
    if (p.nmat == 1 || option == 0) return 0;
    if (option != 1 && option != 101)
        throw Error(11, "Error: unknown phase_change_option: " + std::to_string(option));
    HostMarkers &mk = f.markers;
    const size_t nm = (size_t)mk.nmarkers;
    const int nn = m.nnode, ne = m.nelem, nmat = p.nmat;
    int changed = 0;
    for (size_t i = 0; i < nm; ++i) {
        const int e = mk.elem[i];
        const int current_mt = mk.mattype[i];
        int new_mt = current_mt;
        if (option == 1) {
            // MarkerSet::get_ZPT (markerset.cxx:973-986)
            double Z = 0, T = 0;
            for (int j = 0; j < m.nd + 1; ++j) {
                const int n = m.conn[(size_t)j*ne + e];
                Z += coord[(size_t)(m.nd - 1)*nn + n] * mk.eta[(size_t)j*nm + i];
                T += temperature[n] * mk.eta[(size_t)j*nm + i];
            }
            const double P = ref_pressure(p, Z);
            new_mt = simple_subduction(current_mt, Z, P, T);
        }
        // option 101: custom_phase_change is a template that keeps the material (phasechanges.cxx:92-106)
        if (new_mt != current_mt) {
            if (new_mt < 0 || new_mt >= nmat) throw Error(11, "phase change to a material the model does not have");
            mk.mattype[i] = new_mt;
            --f.elemmarkers[(size_t)e*nmat + current_mt];
            ++f.elemmarkers[(size_t)e*nmat + new_mt];
            ++changed;
        }
    }
    return changed;
}

} // namespace des
