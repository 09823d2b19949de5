// dynearthsol3d-hip -- the reference's `dynearthsol3d config_file` with the explicit time step
// on an MI355X: same .cfg in, same <modelname>.save.NNNNNN / .chkpt.NNNNNN / .info out.
// The loop is des_run (include/des_run.h, mirroring dynearthsol.cxx:593-982); this file only
// binds it to the HIP engine of include/des_dev.h.  There is no CPU fallback: without a GPU
// the run stops with the reference's exit category 31 (EXIT_UNSUPPORTED_LIB).
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
int eng_quality(void *h, double sv, double b, double bd, des_quality *q) { return des_dev_mesh_quality((des_dev *)h, sv, b, bd, q); }
}

int main(int argc, const char *argv[])
{
    std::string cfg, mesh;
    int device = 0, quiet = 0;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--mesh") && i + 1 < argc) mesh = argv[++i];
        else if (!std::strcmp(argv[i], "--quiet")) quiet = 1;
        else if (!std::strcmp(argv[i], "-h") || !std::strcmp(argv[i], "--help")) { cfg.clear(); break; }
        else cfg = argv[i];
    }
    if (cfg.empty()) {
        std::fprintf(stderr, "Usage: %s config_file [--device N] [--mesh file.desmesh] [--quiet]\n", argv[0]);
        return 1;                                                   // EXIT_USAGE
    }
    if (des_dev_device_count() <= device) {
        std::fprintf(stderr, "Error: no HIP device %d visible; the time step has no CPU fallback\n", device);
        return DES_ERR_UNSUPPORTED;
    }
    int err = 0;
    des_host *host = des_host_create(cfg.c_str(), nullptr, mesh.empty() ? nullptr : mesh.c_str(), &err);
    if (!host) {
        std::fprintf(stderr, "%s\n", des_host_last_error());
        return err ? err : DES_ERR_INTERNAL;
    }
    const des_engine_api api = { eng_create, eng_destroy, eng_upload, eng_download, eng_field_count,
                                 eng_set_clock, eng_init_geometry, eng_compute_dt, eng_step, eng_check_nan, eng_quality,
                                 des_dev_last_error, eng_set_isostasy, 0 };
    des_run_stats st;
    int rc = des_run(host, &api, device, quiet, &st);
    des_host_destroy(host);
    return rc;
}
