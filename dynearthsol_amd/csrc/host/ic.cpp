// ic.cpp -- initial conditions, computed once on the host before the state is uploaded
// (the reference does the same in init(), dynearthsol.cxx:159-228).
#include "des_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace des {

namespace {

const double DEG2RAD = M_PI / 180;                 // constants.hpp:77
const double YEAR2SEC = 365.2422 * 86400;          // constants.hpp:76

const double prem_depth[46] = {
    0e3, 3e3, 15e3, 24.4e3, 40e3, 60e3, 80e3, 115e3, 150e3, 185e3, 220e3, 265e3, 310e3,
    355e3, 400e3, 450e3, 500e3, 550e3, 600e3, 635e3, 670e3, 721e3, 771e3, 871e3, 971e3,
    1071e3, 1171e3, 1271e3, 1371e3, 1471e3, 1571e3, 1671e3, 1771e3, 1871e3, 1971e3,
    2071e3, 2171e3, 2271e3, 2371e3, 2471e3, 2571e3, 2671e3, 2741e3, 2771e3, 2871e3, 2891e3 };
const double prem_kbar[2][46] = {
  { 0, 0.3, 3.3, 6.0, 11.2, 17.8, 24.5, 36.1, 47.8, 59.4, 71.1, 86.4, 102.0, 117.7, 133.5,
    152.2, 171.3, 190.7, 210.4, 224.3, 238.3, 260.7, 282.9, 327.6, 372.8, 418.6, 464.8, 511.6,
    558.9, 606.8, 655.2, 704.1, 753.5, 803.6, 854.3, 905.6, 957.6, 1010.3, 1063.8, 1118.2,
    1173.4, 1229.7, 1269.7, 1287.0, 1345.6, 1357.5 },
  { 0, 0.82, 4.1, 6.7, 11.2, 17.8, 24.5, 36.1, 47.8, 59.4, 71.1, 86.4, 102.0, 117.7, 133.5,
    152.2, 171.3, 190.7, 210.4, 224.3, 238.3, 260.7, 282.9, 327.6, 372.8, 418.6, 464.8, 511.6,
    558.9, 606.8, 655.2, 704.1, 753.5, 803.6, 854.3, 905.6, 957.6, 1010.3, 1063.8, 1118.2,
    1173.4, 1229.7, 1269.7, 1287.0, 1345.6, 1357.5 } };

// matprops.cxx:12-101: piecewise-linear PREM pressure, table in kilobar -> 1e8 Pa
double prem_pressure(double depth, int which)
{
    if (depth <= 0) return 0;
    int n;
    for (n = 1; n < 46; n++)
        if (depth <= prem_depth[n]) break;
    double p0 = prem_kbar[which][n-1] * 1e8, p1 = prem_kbar[which][n] * 1e8;
    return p0 + (p1 - p0) * (depth - prem_depth[n-1]) / (prem_depth[n] - prem_depth[n-1]);
}

// Marker-count weighted means (matprops.cxx:116-149)
double harmonic_mean(const double *s, const int *n, int nmat)
{
    if (nmat == 1) return s[0];
    double result = 0; int m = 0;
    for (int i = 0; i < nmat; i++) { if (n[i] == 0) continue; result += n[i] / s[i]; m += n[i]; }
    return m / result;
}
double arithmetic_mean(const double *s, const int *n, int nmat)
{
    if (nmat == 1) return s[0];
    double result = 0; int m = 0;
    for (int i = 0; i < nmat; i++) { if (n[i] == 0) continue; result += n[i] * s[i]; m += n[i]; }
    return result / m;
}

struct ElemView {
    const des_params &p; const HostMesh &m; const HostFields &f;
    double elemT(int e) const {
        const int npe = m.nd + 1;                      // NODES_PER_ELEM
        double T = 0;
        for (int i = 0; i < npe; ++i) T += f.temperature[m.conn[(size_t)i*m.nelem + e]];
        T /= npe;
        return T;
    }
    // MatProps::rho (matprops.cxx:642-664)
    double rho(int e) const {
        double TinCelsius = elemT(e) - 273;
        double result = 0; int n = 0;
        const int *mk = &f.elemmarkers[(size_t)e*p.nmat];
        for (int k = 0; k < p.nmat; k++) {
            result += p.rho0[k] * (1 - p.alpha[k] * TinCelsius) * mk[k];
            n += mk[k];
        }
        return result / n;
    }
    double bulkm(int e) const { return harmonic_mean(p.bulk_modulus, &f.elemmarkers[(size_t)e*p.nmat], p.nmat); }
    double cp(int e) const { return arithmetic_mean(p.heat_capacity, &f.elemmarkers[(size_t)e*p.nmat], p.nmat); }
    double k(int e) const { return arithmetic_mean(p.therm_cond, &f.elemmarkers[(size_t)e*p.nmat], p.nmat); }
    // MatProps::visc (matprops.cxx:333-377) with zero strain rate, as init() calls it
    double visc(int e) const {
        const double min_strain_rate = 1e-30;
        double T = elemT(e);
        const int ne = m.nelem;
        double s0 = (m.nd == 3) ? (f.stress[e] + f.stress[(size_t)ne + e] + f.stress[(size_t)2*ne + e]) / 3
                                : (f.stress[e] + f.stress[(size_t)ne + e]) / 2;       // trace(s) / NDIMS
        double edot = std::max(0.0, min_strain_rate);       // strain_rate is all zero at init
        double result = 0; int n = 0;
        const int *mk = &f.elemmarkers[(size_t)e*p.nmat];
        for (int k = 0; k < p.nmat; k++) {
            if (mk[k] == 0) continue;
            double pow_edot = 1 / p.visc_exponent[k] - 1;
            double coef = std::pow(0.75 * p.visc_coefficient[k], -1 / p.visc_exponent[k]);
            double nR = p.visc_exponent[k] * 8.3144;
            double visc0 = 0.25 * std::pow(edot, pow_edot) * coef
                * std::exp((p.visc_activation_energy[k] + p.visc_activation_volume[k] * s0) / (nR * T)) * 1e6;
            result += mk[k] / visc0;
            n += mk[k];
        }
        double v = n / result;
        return std::min(std::max(v, p.visc_min), p.visc_max);
    }
};

} // namespace

double elem_density(const des_params &p, const int *conn, int nelem, const double *temperature,
                    const int *elemmarkers, int e)
{
    const int npe = p.ndims + 1;
    double T = 0;
    for (int i = 0; i < npe; ++i) T += temperature[conn[(size_t)i*nelem + e]];
    T /= npe;
    double TinCelsius = T - 273;
    double result = 0; int n = 0;
    const int *mk = &elemmarkers[(size_t)e*p.nmat];
    for (int k = 0; k < p.nmat; k++) {
        result += p.rho0[k] * (1 - p.alpha[k] * TinCelsius) * mk[k];
        n += mk[k];
    }
    return result / n;
}

namespace {

// MarkerSet::random_eta (markerset.cxx:116-133)
void random_eta(double eta[4], int nd)
{
    while (1) {
        double sum = 0;
        for (int n = 0; n < nd; n++) {
            eta[n] = (rand() / (double)RAND_MAX);
            sum += eta[n];
        }
        if (sum < 1) { eta[nd] = 1 - sum; break; }
    }
}

// MarkerSet::random_markers + initial_mattype (markerset.cxx:524-553, 666-700).  Only the
// per-element material counts reach the time-stepper; the markers themselves stay with the host
// (same libc rand() sequence as the reference, so the set is the one the reference would build).
void create_elemmarkers(const Config &cfg, const des_params &p, const HostMesh &m, HostFields &f)
{
    const int ne = m.nelem, nmat = p.nmat;
    const int nd = m.nd, npe = nd + 1;
    const int mpe = cfg.i("markers.markers_per_element");
    f.elemmarkers.assign((size_t)ne*nmat, 0);
    const int mattype_option = cfg.i("ic.mattype_option");
    const int imo = cfg.i("markers.init_marker_option");
    if (imo == 2) { regularly_spaced_markers(cfg, p, m, f); return; }
    if (imo != 1)
        throw Error(11, "Error: unknown init_marker_option: " + std::to_string(imo));           // markerset.cxx:50-53
    if (mattype_option != 0 && mattype_option != 1)
        throw Error(11, "Error: unknown ic.mattype_option");
    std::vector<double> layer_mt, depths;
    if (mattype_option == 1) {
        const int nlayers = cfg.i("ic.num_mattype_layers");
        layer_mt = cfg.list("ic.layer_mattypes", nlayers);
        depths = cfg.list("ic.mattype_layer_depths", nlayers - 1);
        if (!std::is_sorted(depths.begin(), depths.end()))
            throw Error(11, "Error: the content of ic.mattype_layer_depths is not ordered from small to big values.");
    }
    HostMarkers &mk = f.markers;
    const size_t nm = (size_t)ne * mpe;
    mk.nmarkers = mk.last_id = (int)nm;
    mk.reserved_space = (int)(nm * 2.0);                       // over_alloc_ratio, markerset.cxx:25
    mk.eta.assign(npe * nm, 0.0);
    mk.elem.assign(nm, 0); mk.mattype.assign(nm, 0); mk.id.assign(nm, 0); mk.genesis.assign(nm, 0);
    mk.time.assign(nm, 0.0); mk.z.assign(nm, 0.0); mk.distance.assign(nm, 0.0); mk.slope.assign(nm, 0.0);

    unsigned seed = (unsigned)cfg.i("markers.random_seed");
    srand(seed ? seed : 1u);
    size_t im = 0;
    for (int e = 0; e < ne; e++)
        for (int k = 0; k < mpe; k++, im++) {
            double eta[4];
            random_eta(eta, nd);
            int mt;
            if (mattype_option == 0) {
                mt = (int)m.regattr[e];
                if (mt < 0 || mt >= nmat) throw Error(11, "region attribute is not a valid material");
            } else {
                double z = 0;
                for (int j = 0; j < npe; j++)
                    z += m.coord[(size_t)(nd-1)*m.nnode + m.conn[(size_t)j*ne + e]] * eta[j];
                mt = (int)layer_mt[layer_mt.size() - 1];
                for (size_t i = 0; i < depths.size(); ++i)
                    if (z >= -p.zlength * depths[i]) { mt = (int)layer_mt[i]; break; }
            }
            for (int j = 0; j < npe; j++) mk.eta[(size_t)j*nm + im] = eta[j];
            mk.elem[im] = e; mk.mattype[im] = mt; mk.id[im] = (int)im;
            ++f.elemmarkers[(size_t)e*nmat + mt];
        }
}

// ic.cxx:834-1024, options 0-2
void initial_temperature(const Config &cfg, const des_params &p, const HostMesh &m, HostFields &f)
{
    const int nn = m.nnode, ne = m.nelem;
    const int nd = m.nd, npe = nd + 1;
    const double *z = &m.coord[(size_t)(nd-1)*nn];
    const double t_top = p.surface_temperature, t_bot = cfg.d("bc.mantle_temperature");
    ElemView ev = {p, m, f};
    f.radiogenic.assign((size_t)ne, 0.0);
    switch (cfg.i("ic.temperature_option")) {
    case 0: {
        const double age = cfg.d("ic.oceanic_plate_age_in_yr") * YEAR2SEC;
        const double diffusivity = ev.k(0) / ev.rho(0) / ev.cp(0);
        for (int i = 0; i < nn; ++i) {
            double w = -z[i] / std::sqrt(4 * diffusivity * age);
            f.temperature[i] = t_top + (t_bot - t_top) * std::erf(w);
        }
        break;
    }
    case 1: {
        const double pi = 3.14159265358979323846;
        const int mc = cfg.i("mat.mattype_crust"), mm = cfg.i("mat.mattype_mantle");
        const int dens_c = (int)p.rho0[mc];          // ic.cxx:858: truncated to int
        const int dens_m = (int)p.rho0[mm];
        const double cond_c = p.therm_cond[std::min(p.nmat-1, mc)];
        const double cond_m = p.therm_cond[std::min(p.nmat-1, mm)];
        const double diff_m = cond_m/1000./dens_m;
        const double age = cfg.d("ic.continental_plate_age_in_yr") * YEAR2SEC;
        const double hs = cfg.d("ic.radiogenic_heating_of_crust");
        const double hr = cfg.d("ic.radiogenic_folding_depth");
        const double hc = cfg.d("ic.radiogenic_crustal_thickness");
        const double hl = cfg.d("ic.lithospheric_thickness");
        const double tr = dens_c * hs * hr*hr / cond_c * exp(1.-exp(-hc/hr));
        const double q_m = (t_bot - t_top - tr) / (hc / cond_c+(hl-hc) / cond_m);
        const double tm  = t_top + (q_m/cond_c) * hc + tr;
        const double tau_d = hl*hl / (pi*pi*diff_m);
        for (int i = 0; i < nn; ++i) {
            double y = -z[i];
            double tss;
            if (y <= hc) tss = t_top + (q_m/cond_c)*y + (dens_c*hs*hr*hr/cond_c) * exp(1.-exp(-y/hr));
            else         tss = tm + (q_m/cond_m) * (y - hc);
            double tt = 0., pp = -1., an;
            for (int k = 1; k < 101; k++) {
                an = 1.*k;
                pp = -pp;
                tt = tt + pp/(an)*exp(-an*an*age/tau_d)*sin(pi*k*(hl-y)/hl);
            }
            f.temperature[i] = tss + 2./pi*(t_bot-t_top)*tt;
            if (f.temperature[i] > t_bot || y >= hl) f.temperature[i] = t_bot;
            if (y == 0.) f.temperature[i] = t_top;
        }
        break;
    }
    case 2: {
        const int nlayer = cfg.i("ic.num_radiogenic_heat_layer");
        const double hr = cfg.d("ic.radiogenic_folding_depth");
        std::vector<double> layer_bdy = cfg.list("ic.radiogenic_heat_boundry", nlayer+1, 1);
        if (layer_bdy[0] == -1) layer_bdy[0] = 0;                         // input.cxx:1434-1437
        if (layer_bdy[nlayer] == -1) layer_bdy[nlayer] = p.zlength;
        std::vector<double> layer_mat = cfg.list("ic.radiogenic_heat_mat_in_layer", nlayer, 1);
        std::vector<double> mat_hp = cfg.list("mat.radiogenic_heat_prod", p.nmat, -1);
        std::vector<double> dT_layer_init(nlayer), thickness(nlayer), cond(nlayer), hp(nlayer), rhohp(nlayer);
        double total_thickness = layer_bdy[nlayer] - layer_bdy[0];
        double avg_cond = 0., dTh_sum = 0., dTc = 0.;
        for (int i = 0; i < nlayer; i++) {
            int mat = (int)layer_mat[i];
            cond[i] = p.therm_cond[mat];
            double rho = p.rho0[mat];
            hp[i] = mat_hp[mat];
            rhohp[i] = hp[i] * rho;
            thickness[i] = layer_bdy[i+1] - layer_bdy[i];
            dT_layer_init[i] = dTh_sum;
            dTh_sum += hp[i]*rho*hr*hr*(1-exp(-thickness[i]/hr)) / cond[i];
            avg_cond += thickness[i]/cond[i];
        }
        avg_cond = total_thickness/avg_cond;
        double qm = (t_bot-t_top-dTh_sum) / total_thickness * avg_cond;
        for (int i = 0; i < nlayer; i++) {
            dT_layer_init[i] += dTc;
            dTc += thickness[i] * qm / cond[i];
        }
        for (int i = 0; i < nn; ++i) {
            double y = -z[i];
            bool is_layer = false;
            for (int j = 0; j < nlayer; j++) {
                if (y >= layer_bdy[j] && y < layer_bdy[j+1]) {
                    double dTr = rhohp[j] * hr*hr *(1-exp(-(y-layer_bdy[j])/hr)) / cond[j];
                    f.temperature[i] = t_top + dT_layer_init[j] + qm * (y-layer_bdy[j]) / cond[j] + dTr;
                    is_layer = true;
                    break;
                }
            }
            if (!is_layer) {
                if (y >= layer_bdy[nlayer]) f.temperature[i] = t_bot;
                else if (y <= layer_bdy[0]) f.temperature[i] = t_top;
            }
        }
        for (int e = 0; e < ne; ++e) {
            double zcenter = 0;
            for (int j = 0; j < npe; ++j) zcenter += z[m.conn[(size_t)j*ne + e]];
            zcenter /= npe;
            double y = -zcenter;
            bool is_layer = false;
            for (int k = 0; k < nlayer; k++) {
                if (y >= layer_bdy[k] && y < layer_bdy[k+1]) {
                    f.radiogenic[e] = hp[k] * exp(-(y-layer_bdy[k])/hr);
                    is_layer = true;
                    break;
                }
            }
            if (!is_layer) {
                if (y >= layer_bdy[nlayer]) f.radiogenic[e] = 0.;
                else if (y <= layer_bdy[0]) f.radiogenic[e] = hp[0];
            }
        }
        break;
    }
    case 3: {
        // radiogenic_heat_and_adiabat (ic.cxx:724-830): continental geotherm after Hasterok &
        // Chapman 2011 with an optional heat-flux dome, capped by the mantle adiabat; markers whose
        // element lies in the adiabatic part become asthenosphere
        const double F = 0.74;                                              // partition coefficient
        const double DEG2RAD = 3.14159265358979323846 / 180;
        const int nlayer = cfg.i("ic.num_radiogenic_heat_layer");
        std::vector<double> layer_bdy = cfg.list("ic.radiogenic_heat_boundry", nlayer+1, 1);
        if (layer_bdy[0] == -1) layer_bdy[0] = 0;                         // input.cxx:1434-1437
        if (layer_bdy[nlayer] == -1) layer_bdy[nlayer] = p.zlength;
        std::vector<double> layer_mat = cfg.list("ic.radiogenic_heat_mat_in_layer", nlayer, 1);
        std::vector<double> mat_hp = cfg.list("mat.radiogenic_heat_prod", p.nmat, -1);
        std::vector<double> thickness(nlayer), rho(nlayer), cond(nlayer), hp(nlayer);
        for (int i = 0; i < nlayer; i++) {
            const int mat = (int)layer_mat[i];
            cond[i] = p.therm_cond[mat];
            rho[i] = p.rho0[mat];
            hp[i] = mat_hp[mat];
            thickness[i] = layer_bdy[i+1] - layer_bdy[i];
        }
        const double wx_r = 1. / cfg.d("ic.radiogenic_heat_dome_width");
        const double az = cfg.d("ic.radiogenic_heat_dome_azimuth") * DEG2RAD;
        const double wy = cfg.d("ic.radiogenic_heat_dome_width_y");
        const double wy_r = (wy == 0) ? wx_r : (wy < 0.0 ? 0. : 1. / wy);
        const double cx = cfg.d("ic.radiogenic_heat_dome_center_x") * p.xlength;
        const double cy = cfg.d("ic.radiogenic_heat_dome_center_y") * p.ylength;
        const double heat_flux = cfg.d("ic.surface_heat_flux"), amplitude = cfg.d("ic.radiogenic_heat_dome_amplitude");
        const int mt_asth = cfg.i("mat.mattype_asthenosphere");
        std::vector<int> in_asth((size_t)nn, 0);
        for (int n = 0; n < nn; n++) {
            const double zz = -z[n];
            const double zPotT = t_bot * std::exp(p.gravity * zz * 4e-8);
            const double dx = m.coord[n] - cx;
            double radius_sq;
            if (nd == 3) {
                const double dy = m.coord[(size_t)nn + n] - cy;
                const double dx_rot = dx * std::cos(az) - dy * std::sin(az);
                const double dy_rot = dx * std::sin(az) + dy * std::cos(az);
                radius_sq = std::pow(dx_rot * wx_r, 2) + std::pow(dy_rot * wy_r, 2);
            } else {
                radius_sq = std::pow(dx * wx_r, 2);                        // ic.cxx:775-777
            }
            const double xsfh = heat_flux + amplitude / 1e6 * std::exp(-radius_sq);
            hp[0] = (1. - F) * xsfh / rho[0] / layer_bdy[1];
            double t = t_top, q = xsfh;
            for (int i = 0; i < nlayer; i++) {
                if (zz >= layer_bdy[i]) {
                    double dd = std::min(zz - layer_bdy[i], thickness[i]);
                    t += q * dd / cond[i] - (rho[i] * hp[i]) / (2. * cond[i]) * dd * dd;
                    q -= rho[i] * hp[i] * dd;
                }
                if (t > zPotT) { in_asth[n] = 1; break; }
            }
            if (in_asth[n]) {
                t = zPotT;
            } else {
                double rs = 0.;
                for (int i = 0; i < nlayer; i++)
                    if (zz >= layer_bdy[i]) rs = hp[i];
                for (int k = m.sup_idx[n]; k < m.sup_idx[n+1]; ++k)
                    f.radiogenic[m.sup_arr[k]] += rs / npe;
            }
            f.temperature[n] = t;
        }
        HostMarkers &mk = f.markers;
        const size_t nm = (size_t)mk.nmarkers;
        for (size_t mi = 0; mi < nm; ++mi) {
            const int e = mk.elem[mi], current_mt = mk.mattype[mi];
            double t = 0;
            for (int i = 0; i < npe; ++i) t += in_asth[m.conn[(size_t)i*ne + e]] * mk.eta[(size_t)i*nm + mi];
            if (t >= 0.5 && current_mt != mt_asth) {
                mk.mattype[mi] = mt_asth;
                --f.elemmarkers[(size_t)e*p.nmat + current_mt];
                ++f.elemmarkers[(size_t)e*p.nmat + mt_asth];
            }
        }
        break;
    }
    case 90:
        throw Error(31, "ic.temperature_option 90 (external temperature file) is not offloaded");
    default:
        throw Error(11, "Error: unknown ic.temperature option");
    }
    double max_temp = 0.0;
    for (int i = 0; i < nn; ++i)
        if (f.temperature[i] > max_temp) max_temp = f.temperature[i];
    f.bottom_temperature = max_temp;
}

// ic.cxx:322-362
void initial_stress_state(des_params &p, const HostMesh &m, HostFields &f)
{
    const int ne = m.nelem, nn = m.nnode;
    const int nd = m.nd, npe = nd + 1, nstr = nd * (nd + 1) / 2;
    f.stress.assign((size_t)nstr*ne, 0.0);
    f.strain.assign((size_t)nstr*ne, 0.0);
    if (nd == 2) f.stressyy.assign((size_t)ne, 0.0);                        // fields.cxx:75
    if (p.gravity == 0) { f.compensation_pressure = 0; p.compensation_pressure = 0; return; }
    ElemView ev = {p, m, f};
    double ks = ev.bulkm(0);
    for (int e = 0; e < ne; ++e) {
        double zcenter = 0;
        for (int i = 0; i < npe; ++i) zcenter += m.coord[(size_t)(nd-1)*nn + m.conn[(size_t)i*ne + e]];
        zcenter /= npe;
        double pr = ref_pressure(p, zcenter);
        if (p.ref_pressure_option == 1 || p.ref_pressure_option == 2) ks = ev.bulkm(e);
        for (int i = 0; i < nd; ++i) {
            f.stress[(size_t)i*ne + e] = -pr;
            f.strain[(size_t)i*ne + e] = -pr / ks / nd;
        }
        if (p.is_plane_strain) f.stressyy[e] = -pr;                         // ic.cxx:357-358
    }
    f.compensation_pressure = ref_pressure(p, -p.zlength);
    p.compensation_pressure = f.compensation_pressure;
}

// ic.cxx:497-654, zone shapes ic.cxx:13-300
void initial_weak_zone(const Config &c, const des_params &p, const HostMesh &m, HostFields &f, bool fresh = true)
{
    const int ne = m.nelem, nn = m.nnode;
    const int nd = m.nd, npe = nd + 1;
    if (fresh) f.plstrain.assign((size_t)ne, 0.0);       // init(): plstrain starts at zero (fields.cxx:96)
    const int option = c.i("ic.weakzone_option");
    if (option == 0) return;
    const double res = c.d("mesh.resolution");
    const double x0[3] = { c.d("ic.weakzone_xcenter") * p.xlength,
                           c.d("ic.weakzone_ycenter") * p.ylength,
                          -c.d("ic.weakzone_zcenter") * p.zlength };
    const double plstrain = c.d("ic.weakzone_plstrain");
    // planar zone
    const double az = std::tan(c.d("ic.weakzone_azimuth") * DEG2RAD);
    const double incl = 1/std::tan(c.d("ic.weakzone_inclination") * DEG2RAD);
    const double halfwidth = c.d("ic.weakzone_halfwidth") * res;
    const double ymin = c.d("ic.weakzone_y_min") * p.ylength, ymax = c.d("ic.weakzone_y_max") * p.ylength;
    const double zmin = -c.d("ic.weakzone_depth_max") * p.zlength, zmax = -c.d("ic.weakzone_depth_min") * p.zlength;
    // ellipsoid
    const double semi[3] = {c.d("ic.weakzone_xsemi_axis"), c.d("ic.weakzone_ysemi_axis"), c.d("ic.weakzone_zsemi_axis")};
    const double sd = c.d("ic.weakzone_standard_deviation");
    const double gauss_amp = c.d("ic.weakzone_gaussian_amplitude");
    if (option < 1 || option > 5)
        throw Error(11, "Error: unknown weakzone_option");
    // Multi_planar_zone of General_planar_zone segments (ic.cxx:72-179, 577-622): unit-normal planes
    // bounded in x, y and z; a point is in the zone if any segment contains it
    struct Segment { double nx, ny, nz, xmin, xmax, ymin, ymax, zmin, zmax, halfwidth, c[3]; };
    std::vector<Segment> segs;
    if (option == 5) {
        const int n = c.i("ic.weakzone_num_segments");
        auto L = [&](const char *key) { return c.list(std::string("ic.weakzone_segments_") + key, n, -1); };
        const std::vector<double> xc = L("xcenter"), yc = L("ycenter"), zc = L("zcenter"), azs = L("azimuth"),
            incs = L("inclination"), hws = L("halfwidth"), xmn = L("x_min"), xmx = L("x_max"), ymn = L("y_min"),
            ymx = L("y_max"), dmn = L("depth_min"), dmx = L("depth_max");
        for (int i = 0; i < n; ++i) {
            Segment g;
            g.nx = -std::cos(azs[i] * DEG2RAD) * std::sin(incs[i] * DEG2RAD);
            g.nz = -std::cos(incs[i] * DEG2RAD);
            g.ny = std::sin(azs[i] * DEG2RAD) * std::sin(incs[i] * DEG2RAD);
            g.xmin = xmn[i] * p.xlength; g.xmax = xmx[i] * p.xlength;
            g.ymin = ymn[i] * p.ylength; g.ymax = ymx[i] * p.ylength;
            g.zmin = -dmx[i] * p.zlength; g.zmax = -dmn[i] * p.zlength;
            g.halfwidth = hws[i] * res;
            g.c[0] = xc[i] * p.xlength; g.c[1] = yc[i] * p.ylength; g.c[2] = -zc[i] * p.zlength;
            segs.push_back(g);
        }
    }

    for (int e = 0; e < ne; ++e) {
        double center[3] = {0, 0, 0};
        for (int i = 0; i < npe; ++i)
            for (int d = 0; d < nd; ++d) center[d] += m.coord[(size_t)d*nn + m.conn[(size_t)i*ne + e]];
        for (int d = 0; d < nd; ++d) center[d] /= npe;
        const double *x = center;
        bool inside = false;
        double value = 1;
        if (nd == 2) {
            // the !THREED forms of the zones (ic.cxx:28-300): x = {x, z}, no y terms
            const double xz[2] = {x0[0], x0[2]};
            if (option == 1 || option == 4) {
                // Planar_zone; Gaussian_planar_zone degenerates to it without the strict-inequality
                // difference of its z test (ic.cxx:241, 254-256)
                const bool zin = (option == 1) ? (x[1] > zmin && x[1] < zmax) : !(x[1] <= zmin || x[1] >= zmax);
                inside = zin && std::fabs((x[0] - xz[0]) + incl * (x[1] - xz[1])) < halfwidth;
            } else if (option == 2) {
                inside = (x[0]-xz[0])*(x[0]-xz[0]) / (semi[0]*semi[0]) + (x[1]-xz[1])*(x[1]-xz[1]) / (semi[2]*semi[2]) < 1;
            } else if (option == 3) {
                double r2 = (x[0]-xz[0])*(x[0]-xz[0]) + (x[1]-xz[1])*(x[1]-xz[1]);
                inside = r2 < (sd * sd * 16.);
                value = exp(-(r2 / (2.*sd*sd)));
            } else {
                for (const Segment &g : segs) {
                    if (x[0] <= g.xmin || x[0] >= g.xmax) continue;
                    if (x[1] <= g.zmin || x[1] >= g.zmax) continue;
                    double dist = g.nx * (x[0] - g.c[0]) + g.nz * (x[1] - g.c[2]);
                    if (std::fabs(dist) < g.halfwidth) { inside = true; break; }
                }
            }
        } else if (option == 1) {
            inside = (x[2] > zmin && x[2] < zmax && x[1] > ymin && x[1] < ymax &&
                      std::fabs((x[0] - x0[0]) - az * (x[1] - x0[1]) + incl * (x[2] - x0[2])) < halfwidth);
        } else if (option == 2) {
            // Ellipsoidal_zone, ic.cxx (sum of squared normalised offsets < 1)
            double r = 0;
            for (int d = 0; d < 3; ++d) r += (x[d]-x0[d])*(x[d]-x0[d]) / (semi[d]*semi[d]);
            inside = r < 1;
        } else if (option == 3) {
            // Gaussian point zone: within 4 sigma, value = exp(-r^2 / 2 sigma^2) (ic.cxx:261-316)
            double r2 = (x[0]-x0[0])*(x[0]-x0[0]) + (x[1]-x0[1])*(x[1]-x0[1]) + (x[2]-x0[2])*(x[2]-x0[2]);
            inside = r2 < (sd * sd * 16.);
            value = exp(-(r2 / (2.*sd*sd)));
        } else if (option == 5) {
            for (const Segment &g : segs) {
                if (x[0] <= g.xmin || x[0] >= g.xmax) continue;
                if (x[2] <= g.zmin || x[2] >= g.zmax) continue;
                if (x[1] <= g.ymin || x[1] >= g.ymax) continue;
                double dist = g.nx * (x[0] - g.c[0]) + g.nz * (x[2] - g.c[2]);
                dist += g.ny * (x[1] - g.c[1]);
                if (std::fabs(dist) < g.halfwidth) { inside = true; break; }
            }
        } else {
            // Gaussian_planar_zone (ic.cxx:207-259): planar zone whose x position bulges along strike
            if (!(x[2] <= zmin || x[2] >= zmax) && !(x[1] <= ymin || x[1] >= ymax)) {
                const double dy = x[1] - x0[1];
                const double x_shift = gauss_amp * std::exp(-dy * dy * (1.0 / (2.0 * sd * sd)));
                inside = std::fabs((x[0] - x0[0] - x_shift) - az * dy + incl * (x[2] - x0[2])) < halfwidth;
            }
        }
        if (inside) f.plstrain[e] = plstrain * value;
    }
}

} // namespace

// restart() with ic.is_restarting_weakzone (dynearthsol.cxx:403-406): a new weak zone on top of
// the restored plastic strain (elements outside the zone keep theirs)
void restart_weak_zone(const Config &c, const des_params &p, const HostMesh &m, HostFields &f)
{
    initial_weak_zone(c, p, m, f, false);                // only the elements inside the zone are overwritten
}

// matprops.cxx:153-174 (has_hydraulic_diffusion == false)
double ref_pressure(const des_params &p, double z)
{
    double depth = -z;
    double pr = 0;
    if (p.ref_pressure_option == 0)
        pr = p.rho0[p.mattype_ref] * p.gravity * depth;
    else if (p.ref_pressure_option == 1)
        pr = prem_pressure(depth, 0);
    else if (p.ref_pressure_option == 2)
        pr = prem_pressure(depth, 1);
    return pr;
}

void initial_conditions(const Config &cfg, des_params &p, const HostMesh &m, HostFields &f)
{
    const int nn = m.nnode, ne = m.nelem;
    f.vel.assign((size_t)m.nd*nn, 0.0);
    f.temperature.assign((size_t)nn, 0.0);      // init() calls mat->rho(0) while T is still 0
    create_elemmarkers(cfg, p, m, f);
    initial_temperature(cfg, p, m, f);
    initial_stress_state(p, m, f);
    initial_weak_zone(cfg, p, m, f);
    f.viscosity.resize((size_t)ne);
    ElemView ev = {p, m, f};
    for (int e = 0; e < ne; ++e) f.viscosity[e] = ev.visc(e);     // dynearthsol.cxx:218-221
}

} // namespace des
