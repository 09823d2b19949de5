// des_host.hpp -- host side of the MI355X time-stepper: the part of DynEarthSol that stays
// on the CPU (".cfg" input, mesh topology, initial conditions, the driver loop, output).
// It mirrors the reference's host surface so the same .cfg gives the same model:
//   Config      <-> input.cxx (boost::program_options front-end)
//   HostMesh    <-> mesh.cxx topology builders + bc.cxx:create_boundary_normals
//   HostFields  <-> the Variables arrays allocate_variables() owns (fields.cxx:56-122)
//   ic_*        <-> ic.cxx
// All 2-D arrays use the reference's SoA layout a[d*n+i] (array2d.hpp:410-425).
#ifndef DES_HOST_HPP
#define DES_HOST_HPP

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "des_params.h"

namespace des {

// Error carrying one of the reference's ExitCode numbers (utils.hpp:20-55).
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};

enum CfgType { CFG_STR, CFG_INT, CFG_DBL, CFG_BOOL, CFG_UINT };

// ".cfg" front-end: same keys, defaults, syntax and error behaviour as
// get_input_parameters() (input.cxx:1503-1519) without boost.
class Config {
public:
    // Parse `filename`, then apply `overrides` ("key = value" lines, same syntax) on top.
    void load(const std::string &filename, const std::string &overrides = "");
    void load_string(const std::string &text, const std::string &overrides = "");

    bool has(const std::string &key) const;      // explicitly given OR defaulted
    bool given(const std::string &key) const;    // explicitly given (vm.count() semantics for no-default keys)
    std::string s(const std::string &key) const;
    int i(const std::string &key) const;
    double d(const std::string &key) const;
    bool b(const std::string &key) const;
    // "[a, b, c]" list of exactly `len` numbers; optional_size as input.cxx:970-996
    std::vector<double> list(const std::string &key, int len, int optional_size = 0) const;

private:
    void parse(const std::string &text, const std::string &origin, bool allow_dup);
    std::map<std::string, std::string> values_;
    std::map<std::string, bool> explicit_;
};

// validate_parameters() (input.cxx:999-1500) for the options the host uses; fills the
// POD handed to the device.  Throws Error with the reference's exit code.  `ndims`: which build
// of the reference this host stands for (3: -DTHREED, tets; 2: triangles) -- a compile-time
// switch there (constants.hpp:12-25), a run-time one here.
void build_params(const Config &cfg, des_params &out, int ndims = 3);

struct HostMesh {
    int nd = 3;                     // NDIMS of the build (constants.hpp:12-16); NODES_PER_ELEM = nd + 1
    int nnode = 0, nelem = 0, nseg = 0;
    std::vector<double> coord;      // [nd][nnode]
    std::vector<int> conn;          // [nd+1][nelem]
    std::vector<int> segment;       // [nd][nseg]
    std::vector<int> segflag;       // [nseg]
    std::vector<double> regattr;    // [nelem]

    std::vector<unsigned> bcflag;
    std::vector<int> bnodes[DES_NBDRY];
    std::vector<int> bfacet_elem[DES_NBDRY], bfacet_facet[DES_NBDRY];
    std::vector<int> sup_idx, sup_arr, sup_lidx;
    std::vector<int> conn_surf;     // [nd+1][etop]
    std::vector<int> top_nodes, elem_and_nodes, ssup_idx, ssup_arr, top_elems;
    std::vector<double> bnormals;   // [nd][10]
    std::vector<double> edge_vec;
    int edge_slot[DES_NBDRY * DES_NBDRY];

    des_mesh view() const;
};

// create_new_mesh (mesh.cxx:3460-3506): meshing_option 1 with meshing_elem_shape 1
// (regular grid split into 5 tets per cell, mesh.cxx:147-346, 1431-1459) is built here;
// tetgen-based options need a mesh file written by tools/ (see DESIGN.md).
void create_new_mesh(const Config &cfg, HostMesh &m, const std::string &mesh_file);
void load_mesh_file(const std::string &path, HostMesh &m);
void save_mesh_file(const std::string &path, const HostMesh &m);
// renumbering_mesh (mesh.cxx:2696-2821)
void renumbering_mesh(const Config &cfg, HostMesh &m);
// create_boundary_flags/nodes/facets, create_support, create_top_elems,
// create_surface_info (mesh.cxx:2837-3329) and create_boundary_normals (bc.cxx:94-224)
void build_topology(HostMesh &m, const int vbc_types[DES_NBDRY]);

// MarkerSet "markerset" (markerset.hpp:14-60).  Markers stay with the host: between remeshings
// they only ride with their element, so the time step needs their per-element counts alone;
// the set is kept for the output frames / checkpoints (write_save_file, write_chkpt_file).
struct HostMarkers {
    int nmarkers = 0, last_id = 0, reserved_space = 0;
    std::vector<double> eta;        // shapefn, SoA [nd+1][nmarkers]
    std::vector<int> elem, mattype, id, genesis;
    std::vector<double> time, z, distance, slope;
};

// What restart() (dynearthsol.cxx:231-435) restores besides the fields a fresh init() builds.
struct RestartState {
    bool active = false;
    int frame = 0, steps = 0, info_display_next_step = 0;
    double time = 0, dt = 0, max_global_vel_mag = 0, reference_frame_time = 0, last_remesh_time = 0;
    std::vector<double> coord0, volume_old, edvacc_surf, dhacc, strain_rate, force, delta_plstrain;
};

struct HostFields {
    RestartState restart;
    std::vector<double> vel, temperature, radiogenic, stress, strain, plstrain, viscosity;
    std::vector<double> stressyy;   // [nelem], 2-D builds only (fields.cxx:75)
    std::vector<int> elemmarkers;   // [nelem][nmat]
    HostMarkers markers;
    double compensation_pressure = 0;
    double bottom_temperature = 0;
};

// init() after the mesh exists (dynearthsol.cxx:172-221): markers counts, temperature,
// lithostatic stress, weak zone, initial viscosity.
void initial_conditions(const Config &cfg, des_params &p, const HostMesh &m, HostFields &f);

// restart() (dynearthsol.cxx:231-435): mesh, marker set and fields from <model>.save.N /
// <model>.chkpt.N / <model>.info instead of create_new_mesh() + initial conditions.  Leaves the
// mesh ready for build_topology().
void restart_from_files(const Config &cfg, des_params &p, HostMesh &m, HostFields &f);

// ic.is_restarting_weakzone (dynearthsol.cxx:403-406)
void restart_weak_zone(const Config &cfg, const des_params &p, const HostMesh &m, HostFields &f);

// markers.init_marker_option = 2 (markerset.cxx:556-663) and phase_changes (phasechanges.cxx:109-152): markers.cpp
void regularly_spaced_markers(const Config &cfg, const des_params &p, const HostMesh &m, HostFields &f);
int phase_changes(const Config &cfg, const des_params &p, const HostMesh &m, HostFields &f,
                  const double *coord, const double *temperature);

// ref_pressure (matprops.cxx:153-174)
double ref_pressure(const des_params &p, double z);

// MatProps::rho (matprops.cxx:642-664) of one element for the "density" output field
double elem_density(const des_params &p, const int *conn, int nelem, const double *temperature,
                    const int *elemmarkers, int e);   // NODES_PER_ELEM = p.ndims + 1

} // namespace des

// the handle behind include/des_host.h
struct des_host {
    des::Config cfg;
    des_params params;
    des::HostMesh mesh;
    des::HostFields fields;
    des_mesh view;
};

#endif
