// dynearthsol3d-hip / dynearthsol2d-hip -- the reference's `dynearthsol3d config_file` (built
// -DTHREED) and `dynearthsol2d config_file` (2-D build, Makefile `ndims = 2`) with the explicit
// time step on an MI355X: same .cfg in, same <modelname>.save.NNNNNN / .chkpt.NNNNNN / .info out.
// One program: the dimension comes from the name it is called by ("2d" in it: the 2-D build) or
// from --ndims.
// The loop is des_run (include/des_run.h, mirroring dynearthsol.cxx:593-982); this file only
// binds it to the HIP engine of include/des_dev.h.  There is no CPU fallback: without a GPU
// the run stops with the reference's exit category 31 (EXIT_UNSUPPORTED_LIB).
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "des_dev.h"
#include "des_run.h"

namespace {
void *eng_create(int device, const des_params *p, const des_mesh *m, int *err) { return des_dev_create(device, p, m, err); }
void eng_destroy(void *h) { des_dev_destroy((des_dev *)h); }
int eng_upload(void *h, int f, const void *a, long long n) { return des_dev_upload((des_dev *)h, f, a, n); }
int eng_download(void *h, int f, void *a, long long n) { return des_dev_download((des_dev *)h, f, a, n); }
long long eng_field_count(const void *h, int f) { return des_dev_field_count((const des_dev *)h, f); }
int eng_set_clock(void *h, double dt, double t, long long s) { return des_dev_set_clock((des_dev *)h, dt, t, s); }
int eng_init_geometry(void *h) { return des_dev_init_geometry((des_dev *)h); }
int eng_compute_dt(void *h, double *dt) { return des_dev_compute_dt((des_dev *)h, dt); }
int eng_step(void *h, int n, des_scalars *s) { return des_dev_step((des_dev *)h, n, s); }
int eng_check_nan(void *h, long long *n) { return des_dev_check_nan((des_dev *)h, n); }
int eng_set_isostasy(void *h, int on) { return des_dev_set_isostasy((des_dev *)h, on); }
int eng_bfa(void *h, des_scalars *s) { return des_dev_body_force_adjustment((des_dev *)h, s); }
int eng_quality(void *h, double sv, double b, double bd, des_quality *q) { return des_dev_mesh_quality((des_dev *)h, sv, b, bd, q); }
}

int main(int argc, const char *argv[])
{
    std::string cfg, mesh, remesher = std::getenv("DES_REMESH_CMD") ? std::getenv("DES_REMESH_CMD") : "";
    int device = 0, quiet = 0;
    const char *base = std::strrchr(argv[0], '/');
    int ndims = std::strstr(base ? base + 1 : argv[0], "2d") ? 2 : 3;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--ndims") && i + 1 < argc) ndims = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--mesh") && i + 1 < argc) mesh = argv[++i];
        else if (!std::strcmp(argv[i], "--remesher") && i + 1 < argc) remesher = argv[++i];
        else if (!std::strcmp(argv[i], "--quiet")) quiet = 1;
        else if (!std::strcmp(argv[i], "-h") || !std::strcmp(argv[i], "--help")) { cfg.clear(); break; }
        else cfg = argv[i];
    }
    if (cfg.empty()) {
        std::fprintf(stderr, "Usage: %s config_file [--ndims 2|3] [--device N] [--mesh file.desmesh] [--remesher command] [--quiet]\n", argv[0]);
        return 1;                                                   // EXIT_USAGE
    }
    if (des_dev_device_count() <= device) {
        std::fprintf(stderr, "Error: no HIP device %d visible; the time step has no CPU fallback\n", device);
        return DES_ERR_UNSUPPORTED;
    }
    const des_engine_api api = { eng_create, eng_destroy, eng_upload, eng_download, eng_field_count,
                                 eng_set_clock, eng_init_geometry, eng_compute_dt, eng_step, eng_check_nan, eng_quality,
                                 des_dev_last_error, eng_set_isostasy, 0, eng_bfa };
    // One des_run per mesh: where the reference would call remesh() the loop returns with the state
    // saved; with a remesher command (include/des_run.h) the run goes on from the remeshed pair.
    std::string overrides;
    int rc = 0;
    const int max_rounds = 100;                    // (dynearthsol_amd/driver.py: run_with_remesher has the same cap)
    for (int round = 0; ; ++round) {
        if (round >= max_rounds) {
            std::fprintf(stderr, "Error: the mesh needed remeshing more than %d times\n", max_rounds);
            return DES_ERR_UNSUPPORTED;
        }
        int err = 0;
        des_host *host = des_host_create_nd(ndims, cfg.c_str(), nullptr, overrides.empty() ? nullptr : overrides.c_str(),
                                            (round == 0 && !mesh.empty()) ? mesh.c_str() : nullptr, &err);
        if (!host) {
            std::fprintf(stderr, "%s\n", des_host_last_error());
            return err ? err : DES_ERR_INTERNAL;
        }
        char model[512] = "";
        des_host_cfg_string(host, "sim.modelname", model, sizeof(model));
        des_run_stats st;
        rc = des_run(host, &api, device, quiet, &st);
        des_host_destroy(host);
        if (!st.remesh_needed || remesher.empty()) break;
        // the model name goes into a shell command line: letters, digits and . _ - / + only
        for (const char *q = model; *q; ++q)
            if (!std::isalnum((unsigned char)*q) && !std::strchr("._-/+", *q)) {
                std::fprintf(stderr, "Error: sim.modelname '%s' cannot be handed to a remesher command (character '%c')\n", model, *q);
                return DES_ERR_UNSUPPORTED;
            }
        const std::string cmd = remesher + " '" + model + "' " + std::to_string(st.last_frame);
        if (!quiet) { std::printf("  Remeshing: %s\n", cmd.c_str()); std::fflush(stdout); }
        if (std::system(cmd.c_str()) != 0) {
            std::fprintf(stderr, "Error: the remesher command failed: %s\n", cmd.c_str());
            return DES_ERR_UNSUPPORTED;
        }
        overrides = "sim.is_restarting = yes\nsim.restarting_from_modelname = " + std::string(model) +
                    "\nsim.restarting_from_frame = " + std::to_string(st.last_frame + 1) + "\n";
    }
    return rc;
}
