// tetmesh_driver.cpp -- TEST INFRASTRUCTURE (dev-time): produce the mesh the reference makes for
// `mesh.meshing_option = 2` (box with a refined zone) by calling the reference's vendored TetGen
// (compiled from /root/reference/tetgen where it lies, oracle/Makefile target `ref`) with the
// same input polyhedron, region list and switches as the reference builds them:
//   polyhedron + regions : new_mesh_refined_zone   mesh.cxx:1642-1845
//   switches             : tetrahedralize_polyhedron mesh.cxx:1222-1328, set_*_str 71-108, 771-783
// Output: raw TetGen result in the host library's mesh-file format (magic DESMESH0); the host
// library then applies discard_internal_segments + renumbering_mesh exactly as create_new_mesh
// does (mesh.cxx:3499-3502).
//
// usage: tetmesh xlength ylength zlength resolution largest_size x0 x1 y0 y1 z0 z1 out.desmesh
//        [max_ratio=2 min_tet_angle=22 optlevel=3]
//        tetmesh --uniform xlength ylength zlength resolution out.desmesh      (meshing_option = 1:
//        new_mesh_uniform_resolution, mesh.cxx:1461-1636: the box alone, one region, a<0.7 d^3>)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tetgen.h"

static const int BOUNDX0 = 1, BOUNDX1 = 2, BOUNDY0 = 4, BOUNDY1 = 8, BOUNDZ0 = 16, BOUNDZ1 = 32;

static int write_mesh(const tetgenio &out, const char *out_path);

static int uniform(int argc, char **argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: see header\n"); return 2; }
    const double Lx = atof(argv[2]), Ly = atof(argv[3]), Lz = atof(argv[4]), d = atof(argv[5]);
    const int cx[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cy[8] = {0, 0, 0, 0, 1, 1, 1, 1}, cz[8] = {0, 1, 1, 0, 0, 1, 1, 0};
    double points[8 * 3];
    for (int c = 0; c < 8; ++c) { points[c*3] = cx[c] * Lx; points[c*3 + 1] = cy[c] * Ly; points[c*3 + 2] = -cz[c] * Lz; }
    const int face[6][4] = {{0,1,5,4}, {0,3,2,1}, {1,2,6,5}, {3,7,6,2}, {7,4,5,6}, {0,4,7,3}};
    int segflags[6] = {BOUNDX0, BOUNDY0, BOUNDZ0, BOUNDX1, BOUNDY1, BOUNDZ1};
    int segments[6 * 4];
    for (int f = 0; f < 6; ++f) for (int k = 0; k < 4; ++k) segments[f*4 + k] = face[f][k];
    double regattr[5] = { 0.5*Lx, 0.5*Ly, -0.5*Lz, 0, -1 };
    const double elem_size = 0.7 * d * d * d;
    std::string options = "Q";
    options += 'q'; options += std::to_string((long double)2.0);
    options += "qq"; options += std::to_string((long double)22.0);
    options += "qqq"; options += std::to_string((long double)(180 - 3 * 22.0));
    options += 'a'; options += std::to_string((long double)elem_size);
    options += "pzs3A";
    std::fprintf(stderr, "tetgen switches: %s\n", options.c_str());
    tetgenio in, out;
    in.pointlist = points; in.numberofpoints = 8;
    tetgenio::polygon polys[6]; tetgenio::facet fl[6];
    for (int i = 0; i < 6; ++i) {
        polys[i].vertexlist = &segments[i*4]; polys[i].numberofvertices = 4;
        fl[i].polygonlist = &polys[i]; fl[i].numberofpolygons = 1; fl[i].holelist = NULL; fl[i].numberofholes = 0;
    }
    in.facetlist = fl; in.facetmarkerlist = segflags; in.numberoffacets = 6;
    in.holelist = NULL; in.numberofholes = 0;
    in.numberofregions = 1; in.regionlist = regattr;
    std::vector<char> opt(options.begin(), options.end()); opt.push_back(0);
    tetrahedralize(opt.data(), &in, &out, NULL, NULL);
    in.pointlist = NULL; in.facetmarkerlist = NULL; in.facetlist = NULL; in.regionlist = NULL;
    for (int i = 0; i < 6; ++i) { polys[i].vertexlist = NULL; fl[i].polygonlist = NULL; }
    return write_mesh(out, argv[6]);
}

int main(int argc, char **argv)
{
    if (argc > 1 && !std::strcmp(argv[1], "--uniform")) return uniform(argc, argv);
    if (argc < 13) { std::fprintf(stderr, "usage: see header\n"); return 2; }
    const double Lx = atof(argv[1]), Ly = atof(argv[2]), Lz = atof(argv[3]), d = atof(argv[4]);
    const double largest_size = atof(argv[5]);
    const double rzx0 = atof(argv[6]), rzx1 = atof(argv[7]), rzy0 = atof(argv[8]), rzy1 = atof(argv[9]);
    const double rzz0 = atof(argv[10]), rzz1 = atof(argv[11]);
    const char *out_path = argv[12];
    const double max_ratio = argc > 13 ? atof(argv[13]) : 2.0;
    const double min_tet_angle = argc > 14 ? atof(argv[14]) : 22.0;
    const int optlevel = argc > 15 ? atoi(argv[15]) : 3;

    // bounds of the refined zone kept one resolution away from the walls (mesh.cxx:1657-1664)
    const double x0 = std::max(rzx0, d / Lx), x1 = std::min(rzx1, 1 - d / Lx);
    const double y0 = std::max(rzy0, d / Ly), y1 = std::min(rzy1, 1 - d / Ly);
    const double z0 = std::max(rzz0, d / Lz), z1 = std::min(rzz1, 1 - d / Lz);

    // 8 corners of the outer box then 8 of the inner one, corner order of mesh.cxx:1726-1776:
    //   0:(0,0,0) 1:(0,0,-Lz) 2:(Lx,0,-Lz) 3:(Lx,0,0) 4:(0,Ly,0) 5:(0,Ly,-Lz) 6:(Lx,Ly,-Lz) 7:(Lx,Ly,0)
    const double bx[2][2] = {{0.0, Lx}, {x0*Lx, x1*Lx}};
    const double by[2][2] = {{0.0, Ly}, {y0*Ly, y1*Ly}};
    const double bz[2][2] = {{0.0, -Lz}, {-z0*Lz, -z1*Lz}};          // [box][top, bottom]
    const int cx[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cy[8] = {0, 0, 0, 0, 1, 1, 1, 1}, cz[8] = {0, 1, 1, 0, 0, 1, 1, 0};
    double points[16 * 3];
    for (int b = 0; b < 2; ++b)
        for (int c = 0; c < 8; ++c) {
            points[(b*8 + c)*3 + 0] = bx[b][cx[c]];
            points[(b*8 + c)*3 + 1] = by[b][cy[c]];
            points[(b*8 + c)*3 + 2] = bz[b][cz[c]];
        }
    // six quadrilateral faces per box, vertex order and flags of mesh.cxx:1778-1813
    const int face[6][4] = {{0,1,5,4}, {0,3,2,1}, {1,2,6,5}, {3,7,6,2}, {7,4,5,6}, {0,4,7,3}};
    const int flag[6] = {BOUNDX0, BOUNDY0, BOUNDZ0, BOUNDX1, BOUNDY1, BOUNDZ1};
    int segments[12 * 4], segflags[12];
    for (int b = 0; b < 2; ++b)
        for (int f = 0; f < 6; ++f) {
            for (int k = 0; k < 4; ++k) segments[(b*6 + f)*4 + k] = b*8 + face[f][k];
            segflags[b*6 + f] = b == 0 ? flag[f] : 0;
        }
    // two regions {x, y, z, attribute, max volume} (mesh.cxx:1815-1833)
    const double vol1 = 0.7 * (d*d*d), vol0 = vol1 * largest_size;
    double regattr[2 * 5] = { d/2, d/2, -d/2, 0.0, vol0,
                              x0*Lx + d/2, y0*Ly + d/2, -z0*Lz - d/2, 0.0, vol1 };

    // switches: Q q<ratio>qq<min>qqq<max> a pzs<opt>A  (mesh.cxx:1237-1248)
    const double max_dihedral_angle = 180 - 3 * min_tet_angle;
    std::string options = "Q";
    options += 'q'; options += std::to_string((long double)max_ratio);
    options += "qq"; options += std::to_string((long double)min_tet_angle);
    options += "qqq"; options += std::to_string((long double)max_dihedral_angle);
    options += 'a';                               // max volume comes from the region attributes
    options += "pzs"; options += std::to_string(optlevel); options += 'A';
    std::fprintf(stderr, "tetgen switches: %s\n", options.c_str());

    tetgenio in, out;
    in.pointlist = points; in.numberofpoints = 16;
    tetgenio::polygon polys[12]; tetgenio::facet fl[12];
    for (int i = 0; i < 12; ++i) {
        polys[i].vertexlist = &segments[i*4]; polys[i].numberofvertices = 4;
        fl[i].polygonlist = &polys[i]; fl[i].numberofpolygons = 1; fl[i].holelist = NULL; fl[i].numberofholes = 0;
    }
    in.facetlist = fl; in.facetmarkerlist = segflags; in.numberoffacets = 12;
    in.holelist = NULL; in.numberofholes = 0;
    in.numberofregions = 2; in.regionlist = regattr;
    std::vector<char> opt(options.begin(), options.end()); opt.push_back(0);
    tetrahedralize(opt.data(), &in, &out, NULL, NULL);
    in.pointlist = NULL; in.facetmarkerlist = NULL; in.facetlist = NULL; in.regionlist = NULL;
    for (int i = 0; i < 12; ++i) { polys[i].vertexlist = NULL; fl[i].polygonlist = NULL; }
    return write_mesh(out, out_path);
}

static int write_mesh(const tetgenio &out, const char *out_path)
{
    const int nn = out.numberofpoints, ne = out.numberoftetrahedra, ns = out.numberoftrifaces;
    std::fprintf(stderr, "tetgen: %d nodes, %d tets, %d boundary/internal faces\n", nn, ne, ns);
    if (ne <= 0) return 40;

    FILE *fp = std::fopen(out_path, "wb");
    if (!fp) return 20;
    const char magic[8] = {'D','E','S','M','E','S','H','0'};
    int hdr[3] = {nn, ne, ns};
    std::fwrite(magic, 1, 8, fp); std::fwrite(hdr, sizeof(int), 3, fp);
    std::vector<double> soa((size_t)3*nn);
    for (int n = 0; n < nn; ++n) for (int k = 0; k < 3; ++k) soa[(size_t)k*nn + n] = out.pointlist[n*3 + k];
    std::fwrite(soa.data(), sizeof(double), soa.size(), fp);
    std::vector<int> conn((size_t)4*ne), seg((size_t)3*ns), sf((size_t)ns);
    for (int e = 0; e < ne; ++e) for (int k = 0; k < 4; ++k) conn[(size_t)k*ne + e] = out.tetrahedronlist[e*4 + k];
    for (int q = 0; q < ns; ++q) { for (int k = 0; k < 3; ++k) seg[(size_t)k*ns + q] = out.trifacelist[q*3 + k]; sf[q] = out.trifacemarkerlist[q]; }
    std::fwrite(conn.data(), sizeof(int), conn.size(), fp);
    std::fwrite(seg.data(), sizeof(int), seg.size(), fp);
    std::fwrite(sf.data(), sizeof(int), sf.size(), fp);
    std::vector<double> reg((size_t)ne);
    for (int e = 0; e < ne; ++e) reg[e] = out.tetrahedronattributelist ? out.tetrahedronattributelist[e] : 0.0;
    std::fwrite(reg.data(), sizeof(double), reg.size(), fp);
    std::fclose(fp);
    return 0;
}
