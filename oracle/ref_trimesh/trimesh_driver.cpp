// trimesh_driver.cpp -- TEST INFRASTRUCTURE (dev-time): produce the mesh the reference's 2-D build
// makes for `mesh.meshing_option = 90 / 91` (.poly file) or `= 1` (uniform rectangle) by calling
// the reference's vendored Triangle (compiled from /root/reference/triangle where it lies,
// oracle/Makefile target `ref`, with the reference's own flags: Makefile:1181-1183) with the same
// points, segments, regions and switches as the reference builds them:
//   .poly reader     : new_mesh_from_polyfile        mesh.cxx:1872-2251 (!THREED branches)
//   uniform rectangle: new_mesh_uniform_resolution   mesh.cxx:1461-1523
//   switches         : triangulate_polygon           mesh.cxx:688-770, set_*_str 71-118
// Output: raw Triangle result in the host library's mesh-file format (magic DESMSH20); the host
// library then applies discard_internal_segments + renumbering_mesh exactly as create_new_mesh does
// (mesh.cxx:3499-3502).
//
// usage: trimesh --poly file.poly meshing_option resolution min_angle nmat out.desmesh
//        trimesh --uniform xlength zlength resolution min_angle nregions out.desmesh
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define REAL double
#define VOID void
#define ANSI_DECLARATORS
#include "triangle.h"

static const int BOUNDX0 = 1, BOUNDX1 = 2, BOUNDZ0 = 16, BOUNDZ1 = 32;

// my_fgets (mesh.cxx): next line that is neither blank nor a comment
static bool next_line(char *buf, int len, std::FILE *fp)
{
    while (std::fgets(buf, len, fp)) {
        const char *p = buf;
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == '#' || *p == '\n' || *p == '\r' || *p == 0) continue;
        return true;
    }
    return false;
}

static int run(double min_angle, double max_area, int npoints, int nsegments, const double *points,
               const int *segments, const int *segflags, int nregions, const double *regattr, const char *out_path)
{
    // set_verbosity_str(-1) + set_2d_quality_str + "pjz" + set_volume_str + 'A' (mesh.cxx:701-709);
    // no steiner cap while an area target is active (mesh.cxx:118-143)
    std::string options = "Q";
    if (min_angle > 0) { options += 'q'; options += std::to_string((long double)min_angle); }
    options += "pjz";
    if (max_area > 0) { options += 'a'; options += std::to_string((long double)max_area); }
    else if (max_area == 0) options += 'a';
    if (nregions > 0) options += 'A';
    std::fprintf(stderr, "triangle switches: %s\n", options.c_str());

    triangulateio in, out;
    std::memset(&in, 0, sizeof(in));
    std::memset(&out, 0, sizeof(out));
    in.pointlist = const_cast<double *>(points);
    in.numberofpoints = npoints;
    in.numberofcorners = 3;
    in.segmentlist = const_cast<int *>(segments);
    in.segmentmarkerlist = const_cast<int *>(segflags);
    in.numberofsegments = nsegments;
    in.numberofregions = nregions;
    in.regionlist = nregions > 0 ? const_cast<double *>(regattr) : NULL;
    std::vector<char> opt(options.begin(), options.end());
    opt.push_back(0);
    triangulate(opt.data(), &in, &out, NULL);

    const int nn = out.numberofpoints, ne = out.numberoftriangles, ns = out.numberofsegments;
    if (ne <= 0) { std::fprintf(stderr, "Error: triangulation failed\n"); return 40; }
    std::fprintf(stderr, "nnode = %d, nelem = %d, nseg = %d\n", nn, ne, ns);
    std::FILE *fp = std::fopen(out_path, "wb");
    if (!fp) { std::fprintf(stderr, "cannot open %s\n", out_path); return 20; }
    const char magic[8] = {'D','E','S','M','S','H','2','0'};
    const int hdr[3] = {nn, ne, ns};
    std::fwrite(magic, 1, 8, fp);
    std::fwrite(hdr, sizeof(int), 3, fp);
    std::vector<double> coord((size_t)2 * nn);
    for (int i = 0; i < nn; ++i) for (int d = 0; d < 2; ++d) coord[(size_t)d * nn + i] = out.pointlist[2 * i + d];
    std::fwrite(coord.data(), sizeof(double), coord.size(), fp);
    std::vector<int> conn((size_t)3 * ne);
    for (int e = 0; e < ne; ++e) for (int k = 0; k < 3; ++k) conn[(size_t)k * ne + e] = out.trianglelist[3 * e + k];
    std::fwrite(conn.data(), sizeof(int), conn.size(), fp);
    std::vector<int> seg((size_t)2 * ns);
    for (int s = 0; s < ns; ++s) for (int k = 0; k < 2; ++k) seg[(size_t)k * ns + s] = out.segmentlist[2 * s + k];
    std::fwrite(seg.data(), sizeof(int), seg.size(), fp);
    std::fwrite(out.segmentmarkerlist, sizeof(int), (size_t)ns, fp);
    std::vector<double> attr((size_t)ne, 0.0);
    if (out.triangleattributelist) for (int e = 0; e < ne; ++e) attr[e] = out.triangleattributelist[e];
    std::fwrite(attr.data(), sizeof(double), attr.size(), fp);
    std::fclose(fp);
    return 0;
}

static int from_poly(char **a)
{
    const char *path = a[0];
    const int meshing_option = std::atoi(a[1]);
    const double resolution = std::atof(a[2]), min_angle = std::atof(a[3]);
    const int nmat = std::atoi(a[4]);
    const double std_elem_size = 1.5 * resolution * resolution;               // mesh.cxx:1890
    std::FILE *fp = std::fopen(path, "r");
    if (!fp) { std::fprintf(stderr, "Error: Cannot open poly_filename '%s'\n", path); return 20; }
    char buffer[2550];
    int npoints, dim, nattr, nbdrym;
    if (!next_line(buffer, 2550, fp) || std::sscanf(buffer, "%d %d %d %d", &npoints, &dim, &nattr, &nbdrym) != 4 ||
        dim != 2 || nattr != 0 || nbdrym != 0) { std::fprintf(stderr, "Error: bad node header\n"); return 13; }
    std::vector<double> points((size_t)npoints * 2);
    for (int i = 0; i < npoints; i++) {
        int k;
        if (!next_line(buffer, 2550, fp) || std::sscanf(buffer, "%d %lf %lf", &k, &points[2*i], &points[2*i+1]) != 3 || k != i) {
            std::fprintf(stderr, "Error: bad node line %d\n", i); return 13;
        }
    }
    int nseg, has_bdryflag;
    if (!next_line(buffer, 2550, fp) || std::sscanf(buffer, "%d %d", &nseg, &has_bdryflag) != 2 || has_bdryflag != 1) {
        std::fprintf(stderr, "Error: bad segment header\n"); return 13;
    }
    std::vector<int> segments((size_t)nseg * 2), segflags((size_t)nseg);
    for (int i = 0; i < nseg; i++) {
        int junk, flag;
        if (!next_line(buffer, 255, fp) || std::sscanf(buffer, "%d %d %d %d", &junk, &segments[2*i], &segments[2*i+1], &flag) != 4) {
            std::fprintf(stderr, "Error: bad segment line %d\n", i); return 13;
        }
        bool ok = flag == 0;
        for (int j = 0; j < 10; j++) if (flag == 1 << j) ok = true;
        if (!ok) { std::fprintf(stderr, "Error: bdry_flag has multiple bits set\n"); return 13; }
        if (segments[2*i] < 0 || segments[2*i] >= npoints || segments[2*i+1] < 0 || segments[2*i+1] >= npoints) {
            std::fprintf(stderr, "Error: segment contains out-of-range node\n"); return 13;
        }
        segflags[i] = flag;
    }
    int nholes;
    if (!next_line(buffer, 255, fp) || std::sscanf(buffer, "%d", &nholes) != 1 || nholes != 0) { std::fprintf(stderr, "Error: holes\n"); return 13; }
    int nregions;
    if (!next_line(buffer, 255, fp) || std::sscanf(buffer, "%d", &nregions) != 1 || nregions <= 0) { std::fprintf(stderr, "Error: regions\n"); return 13; }
    std::vector<double> regattr((size_t)nregions * 4);
    bool has_max_size = false;
    for (int i = 0; i < nregions; i++) {
        int junk;
        double *x = &regattr[(size_t)i * 4];
        if (!next_line(buffer, 255, fp) || std::sscanf(buffer, "%d %lf %lf %lf %lf", &junk, x, x+1, x+2, x+3) != 5) {
            std::fprintf(stderr, "Error: bad region line %d\n", i); return 13;
        }
        if (x[2] < 0 || x[2] >= nmat) { std::fprintf(stderr, "Error: region mattype out of range\n"); return 13; }
        if (x[3] > 0) {
            has_max_size = true;
            if (meshing_option == 91) x[3] *= std_elem_size;
        }
    }
    std::fclose(fp);
    double max_elem_size = std_elem_size;
    if (has_max_size) max_elem_size = 0;
    return run(min_angle, max_elem_size, npoints, nseg, points.data(), segments.data(), segflags.data(),
               nregions, regattr.data(), a[5]);
}

static int uniform(char **a)
{
    const double Lx = std::atof(a[0]), Lz = std::atof(a[1]), res = std::atof(a[2]), min_angle = std::atof(a[3]);
    const int nregions = std::atoi(a[4]);
    const double points[8] = {0, 0, 0, -Lz, Lx, -Lz, Lx, 0};
    const int segments[8] = {0, 1, 1, 2, 2, 3, 3, 0};
    const int segflags[4] = {BOUNDX0, BOUNDZ0, BOUNDX1, BOUNDZ1};
    std::vector<double> regattr((size_t)nregions * 4);
    for (int i = 0; i < nregions; i++) {
        regattr[i*4] = 0.5 * Lx; regattr[i*4 + 1] = -0.5 * Lz; regattr[i*4 + 2] = 0; regattr[i*4 + 3] = -1;
    }
    return run(min_angle, 1.5 * res * res, 4, 4, points, segments, segflags, nregions, regattr.data(), a[5]);
}

int main(int argc, char **argv)
{
    if (argc == 8 && !std::strcmp(argv[1], "--poly")) return from_poly(argv + 2);
    if (argc == 8 && !std::strcmp(argv[1], "--uniform")) return uniform(argv + 2);
    std::fprintf(stderr, "usage: see the header of trimesh_driver.cpp\n");
    return 2;
}
