// capi.cpp -- extern "C" face of the host library (include/des_host.h)
#include "des_host.h"
#include "des_host.hpp"

#include <cstring>
#include <string>

namespace {
thread_local std::string g_last_error;

des_host *create_impl(int ndims, const char *path, const char *text, const char *overrides,
                      const char *mesh_file, int *err)
{
    des_host *h = new des_host();
    try {
        std::string ov = overrides ? overrides : "";
        if (path) h->cfg.load(path, ov);
        else      h->cfg.load_string(text ? text : "", ov);
        des::build_params(h->cfg, h->params, ndims);
        h->mesh.nd = ndims;
        if (h->cfg.b("sim.is_restarting")) {
            des::restart_from_files(h->cfg, h->params, h->mesh, h->fields);
            des::build_topology(h->mesh, h->params.vbc_types);
        } else {
            des::create_new_mesh(h->cfg, h->mesh, mesh_file ? mesh_file : "");
            des::build_topology(h->mesh, h->params.vbc_types);
            des::initial_conditions(h->cfg, h->params, h->mesh, h->fields);
        }
        h->view = h->mesh.view();
        if (err) *err = DES_OK;
        return h;
    } catch (const des::Error &e) {
        g_last_error = e.what();
        if (err) *err = e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        if (err) *err = DES_ERR_INTERNAL;
    }
    delete h;
    return nullptr;
}
} // namespace

extern "C" {

des_host *des_host_create(const char *cfg_path, const char *overrides, const char *mesh_file, int *err)
{
    return create_impl(3, cfg_path, nullptr, overrides, mesh_file, err);
}

des_host *des_host_create_from_string(const char *cfg_text, const char *overrides, const char *mesh_file, int *err)
{
    return create_impl(3, nullptr, cfg_text, overrides, mesh_file, err);
}

des_host *des_host_create_nd(int ndims, const char *cfg_path, const char *cfg_text, const char *overrides,
                             const char *mesh_file, int *err)
{
    return create_impl(ndims, cfg_path, cfg_path ? nullptr : cfg_text, overrides, mesh_file, err);
}

void des_host_destroy(des_host *h) { delete h; }

const des_params *des_host_params(const des_host *h) { return &h->params; }
const des_mesh *des_host_mesh(const des_host *h) { return &h->view; }

const void *des_host_array(const des_host *h, const char *name, long long *count)
{
    struct { const char *n; const void *p; long long c; } t[] = {
        {"coord", h->mesh.coord.data(), (long long)h->mesh.coord.size()},
        {"connectivity", h->mesh.conn.data(), (long long)h->mesh.conn.size()},
        {"segment", h->mesh.segment.data(), (long long)h->mesh.segment.size()},
        {"segflag", h->mesh.segflag.data(), (long long)h->mesh.segflag.size()},
        {"vel", h->fields.vel.data(), (long long)h->fields.vel.size()},
        {"temperature", h->fields.temperature.data(), (long long)h->fields.temperature.size()},
        {"radiogenic", h->fields.radiogenic.data(), (long long)h->fields.radiogenic.size()},
        {"stress", h->fields.stress.data(), (long long)h->fields.stress.size()},
        {"strain", h->fields.strain.data(), (long long)h->fields.strain.size()},
        {"plstrain", h->fields.plstrain.data(), (long long)h->fields.plstrain.size()},
        {"viscosity", h->fields.viscosity.data(), (long long)h->fields.viscosity.size()},
        {"stressyy", h->fields.stressyy.data(), (long long)h->fields.stressyy.size()},
        {"elemmarkers", h->fields.elemmarkers.data(), (long long)h->fields.elemmarkers.size()},
        {"markerset.eta", h->fields.markers.eta.data(), (long long)h->fields.markers.eta.size()},
        {"markerset.elem", h->fields.markers.elem.data(), (long long)h->fields.markers.elem.size()},
        {"markerset.mattype", h->fields.markers.mattype.data(), (long long)h->fields.markers.mattype.size()},
        {"markerset.id", h->fields.markers.id.data(), (long long)h->fields.markers.id.size()},
    };
    for (size_t i = 0; i < sizeof(t)/sizeof(t[0]); ++i)
        if (std::strcmp(name, t[i].n) == 0) { if (count) *count = t[i].c; return t[i].p; }
    if (count) *count = 0;
    return nullptr;
}

int des_host_cfg_int(const des_host *h, const char *key, int *out)
{
    try { if (!h->cfg.has(key)) return DES_ERR_CONFIG_VALUE; *out = h->cfg.i(key); return DES_OK; }
    catch (const des::Error &e) { g_last_error = e.what(); return e.code; }
}

int des_host_cfg_double(const des_host *h, const char *key, double *out)
{
    try { if (!h->cfg.has(key)) return DES_ERR_CONFIG_VALUE; *out = h->cfg.d(key); return DES_OK; }
    catch (const des::Error &e) { g_last_error = e.what(); return e.code; }
}

int des_host_cfg_string(const des_host *h, const char *key, char *out, int cap)
{
    try {
        if (!h->cfg.has(key) || cap < 1) return DES_ERR_CONFIG_VALUE;
        const std::string v = h->cfg.s(key);
        if ((int)v.size() >= cap) return DES_ERR_CONFIG_VALUE;
        std::memcpy(out, v.c_str(), v.size() + 1);
        return DES_OK;
    }
    catch (const des::Error &e) { g_last_error = e.what(); return e.code; }
}

int des_host_save_mesh(const des_host *h, const char *path)
{
    try { des::save_mesh_file(path, h->mesh); return DES_OK; }
    catch (const des::Error &e) { g_last_error = e.what(); return e.code; }
}

const char *des_host_last_error(void) { return g_last_error.c_str(); }

// used by partition.cpp
const des::HostMesh *des_host_mesh_internal(const des_host *h) { return &h->mesh; }
void des_host_set_error(const char *msg) { g_last_error = msg; }

} // extern "C"
