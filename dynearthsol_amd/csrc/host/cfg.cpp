// cfg.cpp -- self-contained ".cfg" front-end.
//
// Accepts exactly the reference's input language (input.cxx:16-1519, which sits on
// boost::program_options::parse_config_file): "[section]" headers, "key = value" lines,
// '#' comments, unknown or repeated keys are errors, booleans are yes/no/on/off/true/false/1/0,
// per-material lists are "[a, b, c]" and broadcast from one entry (input.cxx:983-989).
#include "des_host.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace des {

namespace {

struct OptionDef { const char *name; CfgType type; const char *def; bool required; };

const OptionDef kOptions[] = {
#include "cfg_options.inc"
};
const int kNumOptions = sizeof(kOptions) / sizeof(kOptions[0]);

const OptionDef *find_option(const std::string &key)
{
    for (int i = 0; i < kNumOptions; ++i)
        if (key == kOptions[i].name) return &kOptions[i];
    return nullptr;
}

std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b-1])) --b;
    return s.substr(a, b - a);
}

bool parse_bool(const std::string &v, bool &out)
{
    std::string s;
    for (char c : v) s += (char)std::tolower((unsigned char)c);
    if (s.empty() || s == "on" || s == "yes" || s == "1" || s == "true") { out = true; return true; }
    if (s == "off" || s == "no" || s == "0" || s == "false") { out = false; return true; }
    return false;
}

bool parse_int(const std::string &v, long long &out)
{
    if (v.empty()) return false;
    char *end = nullptr;
    out = std::strtoll(v.c_str(), &end, 10);
    return end && *end == '\0';
}

bool parse_double(const std::string &v, double &out)
{
    if (v.empty()) return false;
    char *end = nullptr;
    out = std::strtod(v.c_str(), &end);
    return end && *end == '\0';
}

void check_value(const OptionDef &o, const std::string &v, const std::string &origin)
{
    bool ok = true;
    long long iv; double dv; bool bv;
    switch (o.type) {
    case CFG_INT:  ok = parse_int(v, iv); break;
    case CFG_UINT: ok = parse_int(v, iv) && iv >= 0; break;
    case CFG_DBL:  ok = parse_double(v, dv); break;
    case CFG_BOOL: ok = parse_bool(v, bv); break;
    case CFG_STR:  break;
    }
    if (!ok)
        throw Error(10, "Error reading config_file '" + origin + "'\nthe argument ('" + v +
                        "') for option '" + o.name + "' is invalid");
}

// read_numbers (input.cxx:932-967)
int read_numbers(const std::string &input, std::vector<double> &vec, int len)
{
    std::istringstream stream(input);
    vec.resize(len);
    char sentinel = 0;
    stream >> sentinel;
    if (sentinel != '[') return 1;
    for (int i = 0; i < len; ++i) {
        stream >> vec[i];
        if (i == len-1) break;
        char sep = 0;
        stream >> sep;
        if (sep != ',') return 1;
    }
    stream >> sentinel;
    if (sentinel == ',') stream >> sentinel;
    if (sentinel != ']') return 1;
    if (!stream.good()) return 1;
    return 0;
}

} // namespace

void Config::parse(const std::string &text, const std::string &origin, bool is_override)
{
    std::istringstream in(text);
    std::string line, section;
    std::map<std::string, bool> seen;
    while (std::getline(in, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        line = trim(line);
        if (line.empty()) continue;
        if (line[0] == '[') {
            if (line[line.size()-1] != ']')
                throw Error(10, "Error reading config_file '" + origin + "'\nthe options configuration file contains an invalid line '" + line + "'");
            section = trim(line.substr(1, line.size() - 2));
            continue;
        }
        size_t eq = line.find('=');
        if (eq == std::string::npos)
            throw Error(10, "Error reading config_file '" + origin + "'\nthe options configuration file contains an invalid line '" + line + "'");
        std::string key = trim(line.substr(0, eq));
        std::string val = trim(line.substr(eq + 1));
        if (!section.empty()) key = section + "." + key;
        (void)is_override;
        const OptionDef *o = find_option(key);
        if (!o)
            throw Error(10, "Error reading config_file '" + origin + "'\nunrecognised option '" + key + "'");
        if (seen.count(key))
            throw Error(11, "option '" + key + "' cannot be specified more than once from option: " + key);
        seen[key] = true;
        check_value(*o, val, origin);
        values_[key] = val;
        explicit_[key] = true;
    }
}

void Config::load_string(const std::string &text, const std::string &overrides)
{
    values_.clear();
    explicit_.clear();
    for (int i = 0; i < kNumOptions; ++i)
        if (kOptions[i].def) values_[kOptions[i].name] = kOptions[i].def;
    parse(text, "<string>", false);
    if (!overrides.empty()) parse(overrides, "<overrides>", true);
    for (int i = 0; i < kNumOptions; ++i)
        if (kOptions[i].required && !explicit_.count(kOptions[i].name))
            throw Error(10, std::string("Error reading config_file\nthe option '") + kOptions[i].name +
                            "' is required but missing");
}

void Config::load(const std::string &filename, const std::string &overrides)
{
    std::ifstream f(filename.c_str());
    if (!f)
        throw Error(10, "Error reading config_file '" + filename + "'\ncan not read options configuration file '" + filename + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    load_string(ss.str(), overrides);
}

bool Config::has(const std::string &key) const { return values_.count(key) != 0; }
bool Config::given(const std::string &key) const { return explicit_.count(key) != 0; }

std::string Config::s(const std::string &key) const
{
    std::map<std::string, std::string>::const_iterator it = values_.find(key);
    if (it == values_.end()) throw Error(60, "config key not set: " + key);
    return it->second;
}

int Config::i(const std::string &key) const
{
    long long v = 0;
    if (!parse_int(s(key), v)) throw Error(11, "bad integer for " + key);
    return (int)v;
}

double Config::d(const std::string &key) const
{
    double v = 0;
    if (!parse_double(s(key), v)) throw Error(11, "bad number for " + key);
    return v;
}

bool Config::b(const std::string &key) const
{
    bool v = false;
    if (!parse_bool(s(key), v)) throw Error(11, "bad boolean for " + key);
    return v;
}

// get_numbers (input.cxx:970-996)
std::vector<double> Config::list(const std::string &key, int len, int optional_size) const
{
    if (!has(key)) throw Error(11, "Error: " + key + " is not provided.");
    std::string str = s(key);
    std::vector<double> values;
    int err = read_numbers(str, values, len);
    if (err && optional_size > 0) {
        err = read_numbers(str, values, optional_size);
    } else if (err && optional_size == -1) {
        err = read_numbers(str, values, 1);
        if (!err) {
            values.resize(len);
            std::fill(values.begin(), values.end(), values[0]);
        }
    }
    if (err)
        throw Error(11, "Error: incorrect format for " + key + ",\n       must be '[d0, d1, d2, ...]'");
    return values;
}

// find_max_vbc (bc.cxx:66-91)
static double find_max_vbc(const Config &c)
{
    double m = 1e-12;
    const char *side[6] = {"x0", "x1", "y0", "y1", "z0", "z1"};
    for (int k = 0; k < 6; ++k) {
        int t = c.i(std::string("bc.vbc_") + side[k]);
        if (t % 2 == 1 || t == 4)
            m = std::max(m, std::fabs(c.d(std::string("bc.vbc_val_") + side[k])));
    }
    const char *slant[4] = {"n0", "n1", "n2", "n3"};
    for (int k = 0; k < 4; ++k) {
        int t = c.i(std::string("bc.vbc_") + slant[k]);
        if (t % 2 == 1)
            m = std::max(m, std::fabs(c.d(std::string("bc.vbc_val_") + slant[k])));
    }
    return m;
}

void build_params(const Config &c, des_params &p, int ndims)
{
    std::memset(&p, 0, sizeof(p));
    if (ndims != 2 && ndims != 3) throw Error(30, "ndims must be 2 or 3");
    p.ndims = ndims;

    // stopping / output conditions, input.cxx:1006-1020
    if (!(c.given("sim.max_steps") || c.given("sim.max_time_in_yr")))
        throw Error(10, "Must provide either sim.max_steps or sim.max_time_in_yr");
    if (!(c.given("sim.output_step_interval") || c.given("sim.output_time_interval_in_yr")))
        throw Error(10, "Must provide either sim.output_step_interval or sim.output_time_interval_in_yr");
    p.quality_check_step_interval = c.i("mesh.quality_check_step_interval");
    if (p.quality_check_step_interval < 1)
        throw Error(11, "mesh.quality_check_step_interval must be positive.");
    if (c.i("sim.checkpoint_frame_interval") < 1)
        throw Error(11, "sim.checkpoint_frame_interval must be positive.");
    // input.cxx:1042-1047
    p.is_outputting_averaged_fields = c.b("sim.is_outputting_averaged_fields");
    if (p.is_outputting_averaged_fields && c.given("sim.output_step_interval") &&
        c.i("sim.output_step_interval") % p.quality_check_step_interval != 0)
        throw Error(11, "sim.output_step_interval must be a multiple of mesh.quality_check_step_interval!.");

    // control
    p.gravity = c.d("control.gravity");
    p.inertial_scaling = c.d("control.inertial_scaling");
    p.damping_factor = c.d("control.damping_factor");
    p.dt_fraction = c.d("control.dt_fraction");
    p.fixed_dt = c.d("control.fixed_dt");
    p.characteristic_speed = c.d("control.characteristic_speed");
    p.surface_diffusivity = c.d("control.surface_diffusivity");
    p.surf_base_level = c.d("control.surf_base_level");
    p.damping_option = c.i("control.damping_option");
    p.ref_pressure_option = c.i("control.ref_pressure_option");
    p.surface_process_option = c.i("control.surface_process_option");
    p.is_quasi_static = c.b("control.is_quasi_static");
    p.has_thermal_diffusion = c.b("control.has_thermal_diffusion");
    p.is_using_mixed_stress = c.b("control.is_using_mixed_stress");
    p.has_moving_mesh = c.b("control.has_moving_mesh");
    if (p.dt_fraction < 0 || p.dt_fraction > 1)
        throw Error(11, "control.dt_fraction must be between 0 and 1.");
    if (p.damping_factor < 0 || p.damping_factor > 1)
        throw Error(11, "control.damping_factor must be between 0 and 1.");
    if (p.ref_pressure_option < 0 || p.ref_pressure_option > 2)
        throw Error(11, "Error: control.ref_pressure_option must be 0, 1, or 2");
    if (p.damping_option < 0 || p.damping_option > 4)
        throw Error(11, "Error: unknown damping_option");          // fields.cxx:572-574
    if (p.surface_process_option != 0 && p.surface_process_option != 1)
        throw Error(31, "surface_process_option other than 0/1 is host-coupled in the reference and not offloaded");
    if (c.b("control.has_hydraulic_diffusion") ||
        c.b("control.use_global_velocity_scaling") || c.b("control.has_hydration_processes"))
        throw Error(31, "hydraulic diffusion / global velocity scaling / hydration are outside the offloaded hot path");
    // (ic.has_body_force_adjustment: the loop calls the engine's initial_body_force_adjustment before the first step,
    //  host/run.cpp)
    // the pseudo-transient loop of a step (dynearthsol.cxx:803-864) runs inside des_dev_step
    p.has_PT = c.b("control.has_PT");
    p.PT_max_iter = c.i("control.PT_max_iter");
    p.PT_relative_tolerance = c.d("control.PT_relative_tolerance");

    if (c.b("monitor.enabled"))
        throw Error(31, "monitor.enabled: monitor-point time series (monitor.cxx) are not written by this driver");

    // bc, with the normalisations of input.cxx:1247-1292
    p.surface_temperature = c.d("bc.surface_temperature");
    p.winkler_delta_rho = c.d("bc.winkler_delta_rho");
    p.elastic_foundation_constant = c.d("bc.elastic_foundation_constant");
    p.sea_water_density = c.d("bc.sea_water_density");
    p.vbc_val_z1_loading_period = c.d("bc.vbc_val_z1_loading_period");
    p.has_winkler_foundation = c.b("bc.has_winkler_foundation");
    p.has_elastic_foundation = c.b("bc.has_elastic_foundation");
    p.has_water_loading = c.b("bc.has_water_loading");
    const char *bname[DES_NBDRY] = {"x0", "x1", "y0", "y1", "z0", "z1", "n0", "n1", "n2", "n3"};
    for (int k = 0; k < DES_NBDRY; ++k) {
        p.vbc_types[k] = c.i(std::string("bc.vbc_") + bname[k]);
        p.vbc_values[k] = c.d(std::string("bc.vbc_val_") + bname[k]);
    }
    for (int k = 0; k < 4; ++k)
        p.vbc_val_l[k] = c.d(std::string("bc.vbc_val_") + bname[k] + "_l");
    for (int k = 0; k < 6; ++k) {
        p.stress_bc_types[k] = c.i(std::string("bc.stress_bc_") + bname[k]);
        p.stress_bc_values[k] = c.d(std::string("bc.stress_val_") + bname[k]);
    }
    if (p.has_winkler_foundation && p.gravity == 0) p.has_winkler_foundation = 0;
    if (p.has_winkler_foundation && p.vbc_types[4] != 0) p.vbc_types[4] = 0;
    if (p.has_water_loading && p.gravity == 0) p.has_water_loading = 0;
    if (p.has_water_loading && p.vbc_types[5] != 0) p.vbc_types[5] = 0;
    if (ndims == 3) {
        if (p.vbc_types[4] > 3) throw Error(11, "bc.vbc_z0 is not 0, 1, 2, or 3.");
        if (p.vbc_types[5] > 3) throw Error(11, "bc.vbc_z1 is not 0, 1, 2, or 3.");
    } else {
        if (p.vbc_types[4] > 4) throw Error(11, "bc.vbc_z0 is not 0, 1, 2, 3, or 4.");
        if (p.vbc_types[5] > 4) throw Error(11, "bc.vbc_z1 is not 0, 1, 2, 3, or 4.");
    }
    for (int k = 6; k < 10; ++k) {
        int t = p.vbc_types[k];
        if (t != 1 && t != 3 && t != 11 && t != 13)
            throw Error(11, std::string("bc.vbc_") + bname[k] + " is not 1, 3, 11, or 13.");
    }

    // mesh
    p.xlength = c.d("mesh.xlength");
    p.ylength = c.d("mesh.ylength");
    p.zlength = c.d("mesh.zlength");
    if (c.d("mesh.smallest_size") > c.d("mesh.largest_size"))
        throw Error(11, "mesh.smallest_size is greater than mesh.largest_size.");

    // mat, input.cxx:1364-1498
    std::string rh = c.s("mat.rheology_type");
    if (rh == "elastic") p.rheol_type = DES_RH_ELASTIC;
    else if (rh == "viscous") p.rheol_type = DES_RH_VISCOUS;
    else if (rh == "maxwell") p.rheol_type = DES_RH_MAXWELL;
    else if (rh == "elasto-plastic") p.rheol_type = DES_RH_EP;
    else if (rh == "elasto-visco-plastic") p.rheol_type = DES_RH_EVP;
    else if (rh.find("rate-state-friction") != std::string::npos || rh.find("rsf") != std::string::npos)
        throw Error(31, "rate-and-state friction rheologies are outside the offloaded hot path");
    else
        throw Error(11, "Error: unknown rheology: '" + rh + "'");

    p.nmat = c.i("mat.num_materials");
    if (p.nmat < 1) throw Error(11, "mat.num_materials must be greater than 0.");
    if (p.nmat > DES_MAX_MAT) throw Error(52, "mat.num_materials exceeds DES_MAX_MAT");
    {
        // phase changes run on the host's marker set every 10 steps (host/markers.cpp, run.cpp);
        // checks of input.cxx:1399-1404
        const int pco = c.i("mat.phase_change_option");
        if (pco != 0 && p.nmat == 1) throw Error(11, "mat.phase_change_option is chosen, but mat.num_materials is 1.");
        if (pco == 1 && p.nmat < 8) throw Error(11, "mat.phase_change_option is 1, but mat.num_materials is less than 8.");
        if (pco != 0 && pco != 1 && pco != 101) throw Error(11, "Error: unknown phase_change_option: " + std::to_string(pco));
    }
    p.mattype_ref = c.i("mat.mattype_ref");
    if (p.mattype_ref < 0 || p.mattype_ref >= p.nmat)
        throw Error(11, "Error: mat.mattype_ref must be within [0, mat.num_materials-1]");
    if (p.nmat == 1 && p.ref_pressure_option != 0) p.ref_pressure_option = 0;

    p.visc_min = c.d("mat.min_viscosity");
    p.visc_max = c.d("mat.max_viscosity");
    p.tension_max = c.d("mat.max_tension");
    p.therm_diff_max = c.d("mat.max_thermal_diffusivity");

    struct { const char *key; double *dst; } lists[] = {
        {"mat.rho0", p.rho0}, {"mat.alpha", p.alpha},
        {"mat.bulk_modulus", p.bulk_modulus}, {"mat.shear_modulus", p.shear_modulus},
        {"mat.visc_exponent", p.visc_exponent}, {"mat.visc_coefficient", p.visc_coefficient},
        {"mat.visc_activation_energy", p.visc_activation_energy},
        {"mat.visc_activation_volume", p.visc_activation_volume},
        {"mat.heat_capacity", p.heat_capacity}, {"mat.therm_cond", p.therm_cond},
        {"mat.pls0", p.pls0}, {"mat.pls1", p.pls1},
        {"mat.cohesion0", p.cohesion0}, {"mat.cohesion1", p.cohesion1},
        {"mat.friction_angle0", p.friction_angle0}, {"mat.friction_angle1", p.friction_angle1},
        {"mat.dilation_angle0", p.dilation_angle0}, {"mat.dilation_angle1", p.dilation_angle1},
        {"mat.porosity", p.porosity},
    };
    for (size_t k = 0; k < sizeof(lists)/sizeof(lists[0]); ++k) {
        std::vector<double> v = c.list(lists[k].key, p.nmat, -1);
        for (int m = 0; m < p.nmat; ++m) lists[k].dst[m] = v[m];
    }
    // the remaining per-material lists are still syntax-checked, as the reference does
    const char *unused_lists[] = {"mat.radiogenic_heat_prod", "mat.hydraulic_perm", "mat.fluid_rho0",
        "mat.fluid_alpha", "mat.fluid_bulk_modulus", "mat.fluid_visc", "mat.biot_coeff",
        "mat.bulk_modulus_s", "mat.direct_a", "mat.evolution_b", "mat.characteristic_velocity",
        "mat.characteristic_distance"};
    for (size_t k = 0; k < sizeof(unused_lists)/sizeof(unused_lists[0]); ++k)
        (void)c.list(unused_lists[k], p.nmat, -1);

    // dynearthsol.cxx:55-59
    p.max_vbc_val = (p.characteristic_speed == 0) ? find_max_vbc(c) : p.characteristic_speed;
    p.compensation_pressure = 0;

    // what only the 2-D build reads
    p.is_plane_strain = (ndims == 2) ? c.b("mat.is_plane_strain") : 0;       // input.cxx:1392-1397
    p.mattype_oceanic_crust = c.i("mat.mattype_oceanic_crust");
    p.num_vbc_period_x0 = c.i("bc.num_vbc_period_x0");
    p.num_vbc_period_x1 = c.i("bc.num_vbc_period_x1");
    if (p.num_vbc_period_x0 < 1 || p.num_vbc_period_x0 > DES_MAX_PERIOD || p.num_vbc_period_x1 < 1 || p.num_vbc_period_x1 > DES_MAX_PERIOD)
        throw Error(52, "bc.num_vbc_period_x? must be within [1, DES_MAX_PERIOD]");
    {
        // input.cxx:1428-1431
        std::vector<double> t0 = c.list("bc.vbc_period_x0_time_in_yr", p.num_vbc_period_x0, 1);
        std::vector<double> t1 = c.list("bc.vbc_period_x1_time_in_yr", p.num_vbc_period_x1, 1);
        std::vector<double> r0 = c.list("bc.vbc_period_x0_ratio", p.num_vbc_period_x0, 1);
        std::vector<double> r1 = c.list("bc.vbc_period_x1_ratio", p.num_vbc_period_x1, 1);
        // a list given with a single entry stays a single entry (get_numbers' optional size): interp1
        // sees the vectors as they were read
        p.num_vbc_period_x0 = (int)std::min(t0.size(), r0.size());
        p.num_vbc_period_x1 = (int)std::min(t1.size(), r1.size());
        for (int k = 0; k < p.num_vbc_period_x0; ++k) { p.vbc_period_x0_time_in_yr[k] = t0[k]; p.vbc_period_x0_ratio[k] = r0[k]; }
        for (int k = 0; k < p.num_vbc_period_x1; ++k) { p.vbc_period_x1_time_in_yr[k] = t1[k]; p.vbc_period_x1_ratio[k] = r1[k]; }
    }
    // dynearthsol.cxx:85-101
    p.vbc_vertical_div_x0[0] = 0.; p.vbc_vertical_div_x0[1] = c.d("bc.vbc_val_division_x0_min");
    p.vbc_vertical_div_x0[2] = c.d("bc.vbc_val_division_x0_max"); p.vbc_vertical_div_x0[3] = 1.;
    p.vbc_vertical_div_x1[0] = 0.; p.vbc_vertical_div_x1[1] = c.d("bc.vbc_val_division_x1_min");
    p.vbc_vertical_div_x1[2] = c.d("bc.vbc_val_division_x1_max"); p.vbc_vertical_div_x1[3] = 1.;
    for (int k = 0; k < 4; ++k) {
        p.vbc_vertical_ratio_x0[k] = c.d("bc.vbc_val_x0_ratio" + std::to_string(k));
        p.vbc_vertical_ratio_x1[k] = c.d("bc.vbc_val_x1_ratio" + std::to_string(k));
    }
    p.bottom_shear_zone_thickness = c.d("bc.bottom_shear_zone_thickness");
    p.surf_diff_ratio_terrig = c.d("control.surf_diff_ratio_terrig");
    p.surf_diff_ratio_marine = c.d("control.surf_diff_ratio_marine");
}

} // namespace des
