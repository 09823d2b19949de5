// restart.cpp -- restart() of the reference (dynearthsol.cxx:231-435): rebuild the host model
// from a frame + checkpoint pair in the reference's binary format (binaryio.cxx:206-330),
// written either by this library (output.cpp) or by the reference itself.
#include "des_host.hpp"

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace des {

namespace {

const std::size_t headerlen = 4096;

// BinaryInput (binaryio.cxx:206-330)
class FrameReader {
public:
    explicit FrameReader(const std::string &filename, int ndims = 3) : name_(filename)
    {
        f_ = std::fopen(filename.c_str(), "rb");
        if (!f_) throw Error(20, "Error: cannot open file: " + filename);                 // EXIT_IO_OPEN
        std::vector<char> header(headerlen + 1, '\0');
        if (std::fread(header.data(), 1, headerlen, f_) != headerlen)
            throw Error(21, "error reading file header");                                   // EXIT_IO_RW
        const std::string revs = "# DynEarthSol ndims=" + std::to_string(ndims) + " revision=4";   // binaryio.cxx:39-40, 230-238
        const char *rev = revs.c_str();
        char *line = std::strtok(header.data(), "\n");
        if (!line || std::strncmp(line, rev, std::strlen(rev)) != 0)
            throw Error(22, std::string("Error: mismatching revision string in header\n  Expect: ") + rev +
                            "\n  Got: " + (line ? line : ""));                              // EXIT_IO_RESTART
        while ((line = std::strtok(nullptr, "\n")) != nullptr) {
            char *tab = std::strchr(line, '\t');
            if (!tab) throw Error(21, std::string("Error: error parsing file header\n Line is:") + line);
            std::size_t loc = 0;
            std::sscanf(tab, "%zu", &loc);
            offset_[std::string(line, tab - line)] = loc;
        }
    }
    ~FrameReader() { if (f_) std::fclose(f_); }
    bool has(const std::string &name) const { return offset_.count(name) != 0; }
    // bytes between this entry and the next one (entries are written back to back)
    std::size_t bytes(const std::string &name)
    {
        std::map<std::string, std::size_t>::const_iterator it = offset_.find(name);
        if (it == offset_.end()) throw Error(22, "Error: no array with a name: " + name + " in " + name_);
        std::fseek(f_, 0, SEEK_END);
        std::size_t next = (std::size_t)std::ftell(f_);
        for (std::map<std::string, std::size_t>::const_iterator o = offset_.begin(); o != offset_.end(); ++o)
            if (o->second > it->second && o->second < next) next = o->second;
        return next - it->second;
    }
    template <typename T> void scalar(T &a, const std::string &name) { read(&a, name, 1); }
    template <typename T> void array(std::vector<T> &a, const std::string &name, std::size_t n)
    {
        a.resize(n);
        if (n) read(a.data(), name, n);
    }
    // AoS file [n][ncomp] -> SoA a[d*n + i]
    template <typename T> void array2d(std::vector<T> &a, int ncomp, const std::string &name, std::size_t n)
    {
        std::vector<T> buf(n * ncomp);
        if (n) read(buf.data(), name, n * ncomp);
        a.resize(n * ncomp);
        for (std::size_t i = 0; i < n; ++i)
            for (int d = 0; d < ncomp; ++d) a[(std::size_t)d * n + i] = buf[i * ncomp + d];
    }

private:
    template <typename T> void read(T *dst, const std::string &name, std::size_t n)
    {
        std::map<std::string, std::size_t>::const_iterator it = offset_.find(name);
        if (it == offset_.end()) throw Error(22, "Error: no array with a name: " + name + " in " + name_);
        std::fseek(f_, (long)it->second, SEEK_SET);
        if (std::fread(dst, sizeof(T), n, f_) != n) throw Error(21, "Error: cannot read array: " + name);
    }
    std::FILE *f_;
    std::string name_;
    std::map<std::string, std::size_t> offset_;
};

std::string frame_name(const std::string &model, const char *kind, int frame)
{
    char b[32];
    std::snprintf(b, sizeof(b), ".%s.%06d", kind, frame);
    return model + b;
}

} // namespace

void restart_from_files(const Config &cfg, des_params &p, HostMesh &m, HostFields &f)
{
    const std::string model = cfg.s("sim.restarting_from_modelname");
    const int frame = cfg.i("sim.restarting_from_frame");
    RestartState &rs = f.restart;
    rs.active = true; rs.frame = frame;

    const int nd = m.nd, npe = nd + 1, nstr = nd * (nd + 1) / 2;
    FrameReader save(frame_name(model, "save", frame), nd);

    // frame metadata: the .info row, else the scalars embedded in the frame (:246-288)
    bool got_meta = false;
    int nnode = 0, nelem = 0, nseg = 0;
    if (std::FILE *fp = std::fopen((model + ".info").c_str(), "r")) {
        int fr, steps, nn, ne, ns;
        while (std::fscanf(fp, "%d %d %*f %*f %*f %d %d %d\n", &fr, &steps, &nn, &ne, &ns) == 5)
            if (fr == frame) { rs.steps = steps; nnode = nn; nelem = ne; nseg = ns; got_meta = true; break; }
        std::fclose(fp);
        if (!got_meta) throw Error(22, "Error: frame " + std::to_string(frame) + " not found in " + model + ".info.");
    }
    if (!got_meta) {
        if (!(save.has("steps") && save.has("nseg")))
            throw Error(22, "Error: cannot read frame metadata from " + model + ".info and the frame has none embedded.");
        save.scalar(rs.steps, "steps"); save.scalar(nnode, "nnode"); save.scalar(nelem, "nelem"); save.scalar(nseg, "nseg");
    }

    FrameReader chk(frame_name(model, "chkpt", frame), nd);

    // mesh (replacing create_new_mesh, :304-317)
    m.nnode = nnode; m.nelem = nelem; m.nseg = nseg;
    save.array2d(m.coord, nd, "coordinate", (std::size_t)nnode);
    save.array2d(m.conn, npe, "connectivity", (std::size_t)nelem);
    chk.array2d(m.segment, nd, "segment", (std::size_t)nseg);
    chk.array(m.segflag, "segflag", (std::size_t)nseg);
    m.regattr.assign((std::size_t)nelem, 0.0);

    // marker set (MarkerSet::read_chkpt_file, markerset.cxx:901-930) and the counts it implies (:76-83)
    HostMarkers &mk = f.markers;
    std::vector<int> itmp;
    chk.array(itmp, "markerset size", 3);
    mk.nmarkers = itmp[0]; mk.last_id = itmp[1]; mk.reserved_space = itmp[2];
    const std::size_t nm = (std::size_t)mk.nmarkers;
    save.array2d(mk.eta, npe, "markerset.eta", nm);
    save.array(mk.elem, "markerset.elem", nm);
    save.array(mk.mattype, "markerset.mattype", nm);
    save.array(mk.id, "markerset.id", nm);
    save.array(mk.time, "markerset.time", nm);
    save.array(mk.z, "markerset.z", nm);
    save.array(mk.distance, "markerset.distance", nm);
    save.array(mk.slope, "markerset.slope", nm);
    chk.array(mk.genesis, "markerset.genesis", nm);
    f.elemmarkers.assign((std::size_t)nelem * p.nmat, 0);
    for (std::size_t i = 0; i < nm; ++i) {
        if (mk.elem[i] < 0 || mk.elem[i] >= nelem || mk.mattype[i] < 0 || mk.mattype[i] >= p.nmat)
            throw Error(22, "Error: marker outside the mesh / material table in the restart files");
        ++f.elemmarkers[(std::size_t)mk.elem[i] * p.nmat + mk.mattype[i]];
    }

    save.array2d(rs.coord0, nd, "coord0", (std::size_t)nnode);

    // misc. items (:343-352)
    chk.scalar(rs.time, "time");
    chk.scalar(rs.info_display_next_step, "info_display_next_step");
    chk.scalar(f.compensation_pressure, "compensation_pressure");
    chk.scalar(f.bottom_temperature, "bottom_temperature");
    chk.scalar(rs.dt, "dt");
    chk.scalar(rs.max_global_vel_mag, "max_global_vel_mag");
    chk.scalar(rs.reference_frame_time, "reference_frame_time");
    chk.scalar(rs.last_remesh_time, "last_remesh_time");
    p.compensation_pressure = f.compensation_pressure;

    // fields (:359-392)
    save.array2d(f.vel, nd, "velocity", (std::size_t)nnode);
    save.array(f.temperature, "temperature", (std::size_t)nnode);
    save.array2d(f.strain, nstr, "strain", (std::size_t)nelem);
    save.array2d(f.stress, nstr, "stress", (std::size_t)nelem);
    if (nd == 2) {
        f.stressyy.assign((std::size_t)nelem, 0.0);
        if (p.is_plane_strain) chk.array(f.stressyy, "stressyy", (std::size_t)nelem);        // dynearthsol.cxx:380-381
    }
    save.array(f.plstrain, "plastic strain", (std::size_t)nelem);
    save.array(f.radiogenic, "radiogenic source", (std::size_t)nelem);
    chk.array(rs.volume_old, "volume_old", (std::size_t)nelem);
    chk.array(rs.edvacc_surf, "dv surface acc", chk.bytes("dv surface acc") / sizeof(double));   // one per top facet
    chk.array(rs.dhacc, "dhacc", (std::size_t)nnode);
    save.array2d(rs.strain_rate, nstr, "strain-rate", (std::size_t)nelem);
    save.array(f.viscosity, "viscosity", (std::size_t)nelem);
    save.array2d(rs.force, nd, "force", (std::size_t)nnode);
    save.array(rs.delta_plstrain, "plastic strain-rate", (std::size_t)nelem);

    if (cfg.b("ic.is_restarting_weakzone"))
        restart_weak_zone(cfg, p, m, f);
}

} // namespace des
