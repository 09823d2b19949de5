// output.cpp -- frame / checkpoint / .info writers in the reference's binary format.
//   FrameFile        <-> BinaryOutput (binaryio.cxx:64-204)
//   des_output_write <-> Output::_write / write_info (output.cxx:41-274)
//   des_output_write_checkpoint <-> Output::write_checkpoint (output.cxx:372-409)
// Arrays arrive SoA (array2d.hpp -DSOA) and are written AoS, as Array2D::pack_to does, so the
// files are what Dynearthsol.py / 2vtk.py / compare.py expect.
#include "des_run.h"
#include "des_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

const std::size_t headerlen = 4096;                   // binaryio.cxx:39
const double YEAR2SEC = 365.2422 * 86400;             // constants.hpp:77

long long now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(
        std::chrono::steady_clock::now().time_since_epoch()).count();
}

// binaryio.cxx:44-60: keep a previous file of that name as .old, .old2, ...
void rename_to_old_backup(const std::string &fpath)
{
    int max_n = 0;
    for (int n = 1; n <= max_n + 200; ++n) {
        std::string candidate = fpath + ".old" + (n == 1 ? "" : std::to_string(n));
        if (std::FILE *f = std::fopen(candidate.c_str(), "r")) { std::fclose(f); max_n = n; }
    }
    const int next_n = max_n + 1;
    const std::string backup = fpath + ".old" + (next_n == 1 ? "" : std::to_string(next_n));
    if (std::rename(fpath.c_str(), backup.c_str()) == 0)
        std::fprintf(stderr, "[Runtime][IO] Renamed '%s' -> '%s' (preserving previous output)\n", fpath.c_str(), backup.c_str());
}

// binaryio.cxx:64-204
class FrameFile {
public:
    explicit FrameFile(const std::string &filename, bool rename_if_exists = false, int ndims = 3)
        : header_(headerlen, '\0'), eof_pos_(headerlen)
    {
        if (rename_if_exists) rename_to_old_backup(filename);
        f_ = std::fopen(filename.c_str(), "wb");
        if (!f_) throw des::Error(20, "Error: cannot open file: " + filename);             // EXIT_IO_OPEN
        const std::string rev = "# DynEarthSol ndims=" + std::to_string(ndims) + " revision=4\n";       // binaryio.cxx:39-40
        std::memcpy(&header_[0], rev.data(), rev.size());
        hd_pos_ = rev.size();
        std::fseek(f_, (long)eof_pos_, SEEK_SET);
    }
    ~FrameFile() { close(); }
    void close()
    {
        if (!f_) return;
        std::fseek(f_, 0, SEEK_SET);
        std::fwrite(header_.data(), 1, headerlen, f_);
        std::fclose(f_);
        f_ = nullptr;
    }
    template <typename T> void scalar(const T &a, const char *name)
    {
        entry(name);
        eof_pos_ += std::fwrite(&a, sizeof(T), 1, f_) * sizeof(T);
    }
    template <typename T> void array(const T *a, const char *name, std::size_t n)
    {
        entry(name);
        eof_pos_ += std::fwrite(a, sizeof(T), n, f_) * sizeof(T);
    }
    // Array2D<T,N>::pack_to: SoA a[d*n + i] -> AoS buf[i*N + d]
    template <typename T> void array2d(const T *a, int ncomp, const char *name, std::size_t n)
    {
        buf_.resize(n * ncomp * sizeof(T));
        T *b = reinterpret_cast<T *>(buf_.data());
        for (int d = 0; d < ncomp; ++d)
            for (std::size_t i = 0; i < n; ++i) b[i * ncomp + d] = a[(std::size_t)d * n + i];
        array(b, name, n * ncomp);
    }

private:
    void entry(const char *name)
    {
        char line[256];
        int len = std::snprintf(line, sizeof(line), "%s\t%ld\n", name, (long)eof_pos_);
        if (len >= (int)sizeof(line) || hd_pos_ + len >= headerlen)
            throw des::Error(60, std::string("Error: exceeding header length at Output::write_array, name=") + name);
        std::memcpy(&header_[hd_pos_], line, len);
        hd_pos_ += len;
    }
    std::FILE *f_;
    std::string header_;
    std::size_t hd_pos_, eof_pos_;
    std::vector<char> buf_;
};

// utils.hpp:304-311
std::string fmt_time_ns(long long duration)
{
    int hours = (int)(duration / 3600000000000LL);
    int minutes = (int)((duration % 3600000000000LL) / 60000000000LL);
    double seconds = (duration % 60000000000LL) / 1e9;
    char b[64];
    std::snprintf(b, sizeof(b), "%03d:%02d:%09.6f", hours, minutes, seconds);
    return b;
}

// geometry.cxx:77-105, 1873-1909
double tri_area(const double *a, const double *b, const double *c)
{
    double ab0 = b[0] - a[0], ab1 = b[1] - a[1], ab2 = b[2] - a[2];
    double ac0 = c[0] - a[0], ac1 = c[1] - a[1], ac2 = c[2] - a[2];
    double d0 = ab1*ac2 - ab2*ac1, d1 = ab2*ac0 - ab0*ac2, d2 = ab0*ac1 - ab1*ac0;
    return std::sqrt(d0*d0 + d1*d1 + d2*d2) / 2;
}

thread_local std::string g_err;

} // namespace

struct des_output {
    const des_host *host;
    std::string modelname;
    long long start_time;
    bool is_averaged;
    int average_interval;
    int frame, start_frame;
    bool quiet;
    bool has_marker_output;
    bool may_overwrite;         // same-name restart (output.cxx:30-31)
};

extern "C" {

des_output *des_output_create(const des_host *host, int start_frame)
{
    des_output *o = new des_output();
    o->host = host;
    o->modelname = host->cfg.s("sim.modelname");
    o->start_time = now_ns();
    o->is_averaged = host->cfg.b("sim.is_outputting_averaged_fields");
    o->average_interval = host->cfg.i("mesh.quality_check_step_interval");
    o->frame = o->start_frame = start_frame;
    o->may_overwrite = host->cfg.b("sim.is_restarting") &&
                       o->modelname == host->cfg.s("sim.restarting_from_modelname");
    o->quiet = false;
    o->has_marker_output = host->cfg.b("sim.has_marker_output");
    return o;
}

void des_output_destroy(des_output *o) { delete o; }
int des_output_frame(const des_output *o) { return o->frame; }

static void write_info(des_output *o, const des_frame *f, double dt, long long run_time_ns)
{
    const des::HostMesh &m = o->host->mesh;
    char buffer[256];
    std::snprintf(buffer, 255, "%6d\t%10d\t%12.6e\t%12.4e\t%12.6e\t%8d\t%8d\t%8d\n",
                  o->frame, (int)f->steps, f->time, dt, run_time_ns * 1e-9, m.nnode, m.nelem, m.nseg);
    const std::string filename = o->modelname + ".info";
    // first output of a same-name restart: keep only the rows of earlier frames (output.cxx:51-72)
    if (o->may_overwrite && o->frame == o->start_frame) {
        std::vector<std::string> kept_lines;
        if (std::FILE *r = std::fopen(filename.c_str(), "r")) {
            char line[256];
            while (std::fgets(line, sizeof(line), r)) {
                int f_col;
                if (std::sscanf(line, "%d", &f_col) == 1 && f_col < o->start_frame) kept_lines.push_back(line);
            }
            std::fclose(r);
        }
        rename_to_old_backup(filename);
        if (std::FILE *w = std::fopen(filename.c_str(), "w")) {
            for (const std::string &line : kept_lines) std::fputs(line.c_str(), w);
            std::fclose(w);
        }
    }
    std::FILE *fp = std::fopen(filename.c_str(), o->frame == 0 ? "w" : "a");
    if (!fp) throw des::Error(20, "Error: cannot open file '" + filename + "' for writing");
    if (std::fputs(buffer, fp) == EOF) { std::fclose(fp); throw des::Error(21, "Error: failed writing to file '" + filename + "'"); }
    std::fclose(fp);
}

int des_output_write(des_output *o, const des_frame *f, int exact)
{
    try {
        const des_host *h = o->host;
        const des::HostMesh &m = h->mesh;
        const std::size_t nn = m.nnode, ne = m.nelem;
        const long long run_time_ns = now_ns() - o->start_time;
        const bool averaged = !exact && o->is_averaged;
        if (averaged && !(f->coord_avg0 && f->strain0 && f->stress_avg && f->dplstrain_avg))
            throw des::Error(60, "averaged output requested without the average_fields state");

        double dt = f->dt, inv_dt = 0;
        if (averaged) {
            dt = (f->time - f->avg_time0) / o->average_interval;
            inv_dt = 1.0 / (f->time - f->avg_time0);
        }

        char filename[256];
        std::snprintf(filename, 255, "%s.save.%06d", o->modelname.c_str(), o->frame);
        const int nd = m.nd, npe = nd + 1, nstr = nd * (nd + 1) / 2;
        FrameFile bin(filename, o->may_overwrite && o->frame == o->start_frame, nd);

        bin.array2d(f->coord, nd, "coordinate", nn);
        bin.array2d(m.conn.data(), npe, "connectivity", ne);
        bin.scalar((int)f->steps, "steps");
        bin.scalar(double(run_time_ns) * 1e-9, "walltime_sec");
        bin.scalar((int)m.nnode, "nnode");
        bin.scalar((int)m.nelem, "nelem");
        bin.scalar(f->time, "time_sec");
        bin.scalar(dt, "dt_sec");
        bin.scalar((int)m.nseg, "nseg");

        bin.array2d(f->vel, nd, "velocity", nn);
        std::vector<double> tmp;
        if (averaged) {
            // average_velocity = displacement / delta_t
            tmp.resize(nd * nn);
            for (std::size_t i = 0; i < nd * nn; ++i) tmp[i] = (f->coord[i] - f->coord_avg0[i]) * inv_dt;
            bin.array2d(tmp.data(), nd, "velocity averaged", nn);
        }
        bin.array(f->temperature, "temperature", nn);
        tmp.assign(nn, 0.0);                       // no hydraulic diffusion on the device path
        bin.array(tmp.data(), "pore pressure", nn);
        bin.array(f->radiogenic, "radiogenic source", ne);
        bin.array(f->plstrain, "plastic strain", ne);

        if (averaged) {
            tmp.resize(ne);
            for (std::size_t i = 0; i < ne; ++i) tmp[i] = f->dplstrain_avg[i] * inv_dt;
            bin.array(tmp.data(), "plastic strain-rate", ne);
        } else {
            bin.array(f->delta_plstrain, "plastic strain-rate", ne);
        }
        if (averaged) {
            // average_strain_rate = delta_strain / delta_t
            tmp.resize(nstr * ne);
            for (std::size_t i = 0; i < nstr * ne; ++i) tmp[i] = (f->strain[i] - f->strain0[i]) * inv_dt;
            bin.array2d(tmp.data(), nstr, "strain-rate", ne);
        } else {
            bin.array2d(f->strain_rate, nstr, "strain-rate", ne);
        }
        bin.array2d(f->strain, nstr, "strain", ne);
        bin.array2d(f->stress, nstr, "stress", ne);
        bin.array(f->viscosity, "viscosity", ne);
        if (averaged) {
            const double w = 1.0 / (o->average_interval + 1);
            tmp.resize(nstr * ne);
            for (std::size_t i = 0; i < nstr * ne; ++i) tmp[i] = f->stress_avg[i] * w;
            bin.array2d(tmp.data(), nstr, "stress averaged", ne);
        }

        tmp.resize(ne);
        for (std::size_t e = 0; e < ne; ++e)
            tmp[e] = des::elem_density(h->params, m.conn.data(), m.nelem, f->temperature, f->elemmarkers, (int)e);
        bin.array(tmp.data(), "density", ne);

        for (std::size_t e = 0; e < ne; ++e) {
            double c[4][3];
            for (int i = 0; i < npe; ++i) {
                const std::size_t n = m.conn[(std::size_t)i * ne + e];
                for (int d = 0; d < nd; ++d) c[i][d] = f->coord[(std::size_t)d * nn + n];
            }
            if (nd == 2) {
                // elem_quality, geometry.cxx:1901-1906
                auto dist2 = [](const double *a, const double *b) {
                    double sum = 0;
                    for (int i = 0; i < 2; ++i) { double d = b[i] - a[i]; sum += d * d; }
                    return sum;
                };
                const double normalization_factor = 4 * std::sqrt(3);
                const double dist2_sum = dist2(c[0], c[1]) + dist2(c[1], c[2]) + dist2(c[0], c[2]);
                tmp[e] = normalization_factor * f->volume[e] / dist2_sum;
                continue;
            }
            const double normalization_factor = 216 * std::sqrt(3);
            const double area_sum = (tri_area(c[0], c[1], c[2]) + tri_area(c[0], c[1], c[3]) +
                                     tri_area(c[2], c[3], c[0]) + tri_area(c[2], c[3], c[1]));
            const double vol = f->volume[e];
            tmp[e] = normalization_factor * vol * vol / (area_sum * area_sum * area_sum);
        }
        bin.array(tmp.data(), "mesh quality", ne);

        const int nmat = h->params.nmat;
        for (std::size_t e = 0; e < ne; ++e) {
            // the most abundant marker mattype in this element
            const int *a = f->elemmarkers + e * nmat;
            tmp[e] = (double)(std::max_element(a, a + nmat) - a);
        }
        bin.array(tmp.data(), "material", ne);

        bin.array2d(f->force, nd, "force", nn);
        bin.array2d(f->coord0, nd, "coord0", nn);
        bin.array(m.bcflag.data(), "bcflag", nn);

        if (o->has_marker_output) {
            // MarkerSet::write_save_file (markerset.cxx:939-970)
            const des::HostMarkers &mk = h->fields.markers;
            const std::size_t nm = (std::size_t)mk.nmarkers;
            const int itmp[1] = { mk.nmarkers };
            bin.array(itmp, "markerset size", 1);
            tmp.assign(nd * nm, 0.0);                              // calculate_marker_coord (:989-1007)
            for (std::size_t n = 0; n < nm; ++n)
                for (int d = 0; d < nd; ++d) {
                    double sum = 0;
                    for (int k = 0; k < npe; ++k)
                        sum += f->coord[(std::size_t)d * nn + m.conn[(std::size_t)k * ne + mk.elem[n]]] * mk.eta[(std::size_t)k * nm + n];
                    tmp[n * nd + d] = sum;
                }
            bin.array(tmp.data(), "markerset.coord", nd * nm);
            bin.array2d(mk.eta.data(), npe, "markerset.eta", nm);
            bin.array(mk.elem.data(), "markerset.elem", nm);
            bin.array(mk.mattype.data(), "markerset.mattype", nm);
            bin.array(mk.id.data(), "markerset.id", nm);
            bin.array(mk.time.data(), "markerset.time", nm);
            bin.array(mk.z.data(), "markerset.z", nm);
            bin.array(mk.distance.data(), "markerset.distance", nm);
            bin.array(mk.slope.data(), "markerset.slope", nm);
            bin.array(mk.genesis.data(), "markerset.genesis", nm);
        }
        bin.close();

        write_info(o, f, dt, run_time_ns);

        if (!o->quiet) {
            if (dt / YEAR2SEC > 0.001)
                std::printf("  Output # %d, step = %lld, time = %.5e yr, vmax = %.5e m/s, dt = %.5e yr, wt = %s\n",
                            o->frame, f->steps, f->time / YEAR2SEC, f->max_global_vel_mag, dt / YEAR2SEC,
                            fmt_time_ns(run_time_ns).c_str());
            else
                std::printf("  Output # %d, step = %lld, time = %.5e sec, vmax = %.5e m/s, dt = %.5e sec, wt = %s\n",
                            o->frame, f->steps, f->time, f->max_global_vel_mag, dt, fmt_time_ns(run_time_ns).c_str());
            std::fflush(stdout);
        }
        o->frame++;
        return DES_OK;
    } catch (const des::Error &e) {
        g_err = e.what();
        return e.code;
    }
}

// a frame some other rank writes: keep the numbering in step
int des_output_skip(des_output *o) { o->frame++; return DES_OK; }

int des_output_write_checkpoint(des_output *o, const des_frame *f)
{
    try {
        const des_host *h = o->host;
        const des::HostMesh &m = h->mesh;
        char filename[256];
        std::snprintf(filename, 255, "%s.chkpt.%06d", o->modelname.c_str(), o->frame);
        FrameFile bin(filename, o->may_overwrite && o->frame == o->start_frame, m.nd);
        bin.scalar(f->time, "time");
        bin.scalar(f->info_display_next_step, "info_display_next_step");
        bin.scalar(h->fields.compensation_pressure, "compensation_pressure");
        bin.scalar(h->fields.bottom_temperature, "bottom_temperature");
        bin.scalar(f->dt, "dt");
        bin.scalar(f->max_global_vel_mag, "max_global_vel_mag");
        bin.scalar(f->reference_frame_time, "reference_frame_time");
        bin.scalar(f->last_remesh_time, "last_remesh_time");
        bin.array2d(m.segment.data(), m.nd, "segment", (std::size_t)m.nseg);
        bin.array(m.segflag.data(), "segflag", (std::size_t)m.nseg);
        bin.array(f->edvacc_surf, "dv surface acc", m.conn_surf.size() / (m.nd + 1));
        bin.array(f->dhacc, "dhacc", (std::size_t)m.nnode);
        bin.array(f->volume_old, "volume_old", (std::size_t)m.nelem);
        if (h->params.is_plane_strain && f->stressyy)                      // output.cxx:394-395
            bin.array(f->stressyy, "stressyy", (std::size_t)m.nelem);
        {   // MarkerSet::write_chkpt_file (markerset.cxx:877-891)
            const des::HostMarkers &mk = h->fields.markers;
            const int itmp[3] = { mk.nmarkers, mk.last_id, mk.reserved_space };
            bin.array(itmp, "markerset size", 3);
            bin.array(mk.genesis.data(), "markerset.genesis", (std::size_t)mk.nmarkers);
        }
        // not in the reference's file (it rebuilds the counts from its marker sets)
        bin.array(f->elemmarkers, "elemmarkers", (std::size_t)m.nelem * h->params.nmat);
        return DES_OK;
    } catch (const des::Error &e) {
        g_err = e.what();
        return e.code;
    }
}

} // extern "C"

namespace des {
const std::string &output_last_error() { return g_err; }
void output_set_quiet(des_output *o, bool q) { o->quiet = q; }
}
