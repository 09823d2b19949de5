"""ctypes face of include/des_run.h: the reference's main loop + output writers over an engine.

``run(host)`` is what ``bin/dynearthsol3d-hip config.cfg`` does: it binds des_run to the HIP
engine (no CPU fallback).  Tests pass ``api=`` to drive the same loop with the CPU oracle."""
import ctypes as C

from . import DesError, Host, load_hip_lib, load_host_lib
from ._structs import DesMesh, DesParams, DesScalars


class DesQuality(C.Structure):
    _fields_ = [("small_elem", C.c_int), ("bottom_node", C.c_int), ("worst_elem", C.c_int),
                ("pad_", C.c_int), ("worst_quality", C.c_double)]


_vp, _i, _ll, _d = C.c_void_p, C.c_int, C.c_longlong, C.c_double
CREATE_T = C.CFUNCTYPE(_vp, _i, C.POINTER(DesParams), C.POINTER(DesMesh), C.POINTER(_i))
DESTROY_T = C.CFUNCTYPE(None, _vp)
UPLOAD_T = C.CFUNCTYPE(_i, _vp, _i, _vp, _ll)
DOWNLOAD_T = C.CFUNCTYPE(_i, _vp, _i, _vp, _ll)
COUNT_T = C.CFUNCTYPE(_ll, _vp, _i)
CLOCK_T = C.CFUNCTYPE(_i, _vp, _d, _d, _ll)
INITGEOM_T = C.CFUNCTYPE(_i, _vp)
DT_T = C.CFUNCTYPE(_i, _vp, C.POINTER(_d))
STEP_T = C.CFUNCTYPE(_i, _vp, _i, C.POINTER(DesScalars))
NAN_T = C.CFUNCTYPE(_i, _vp, C.POINTER(_ll))
QUALITY_T = C.CFUNCTYPE(_i, _vp, _d, _d, _d, C.POINTER(DesQuality))
ERR_T = C.CFUNCTYPE(C.c_char_p)
ISO_T = C.CFUNCTYPE(_i, _vp, _i)
BFA_T = C.CFUNCTYPE(_i, _vp, C.POINTER(DesScalars))


class EngineApi(C.Structure):
    """des_engine_api"""
    _fields_ = [("create", CREATE_T), ("destroy", DESTROY_T), ("upload", UPLOAD_T), ("download", DOWNLOAD_T),
                ("field_count", COUNT_T), ("set_clock", CLOCK_T), ("init_geometry", INITGEOM_T), ("compute_dt", DT_T), ("step", STEP_T),
                ("check_nan", NAN_T), ("mesh_quality", QUALITY_T), ("last_error", ERR_T), ("set_isostasy", ISO_T), ("no_files", _i),
                ("body_force_adjustment", BFA_T)]


class RunStats(C.Structure):
    """des_run_stats"""
    _fields_ = [("steps", _ll), ("time", _d), ("dt", _d), ("frames", _i), ("checkpoints", _i),
                ("exit_code", _i), ("remesh_needed", _i), ("compute_seconds", _d), ("phase_changed_markers", _ll),
                ("last_frame", _i), ("pad_", _i)]


def api_from_lib(lib, prefix, create=None):
    """Engine table from a library exporting `<prefix>_*` with the des_dev.h signatures."""
    g = lambda name, T: C.cast(getattr(lib, prefix + "_" + name), T)
    api = EngineApi()
    api.create = create if create is not None else g("create", CREATE_T)
    api.destroy = g("destroy", DESTROY_T)
    api.upload = g("upload", UPLOAD_T)
    api.download = g("download", DOWNLOAD_T)
    api.field_count = g("field_count", COUNT_T)
    api.set_clock = g("set_clock", CLOCK_T)
    api.init_geometry = g("init_geometry", INITGEOM_T)
    api.compute_dt = g("compute_dt", DT_T)
    api.step = g("step", STEP_T)
    api.check_nan = g("check_nan", NAN_T)
    api.mesh_quality = g("mesh_quality", QUALITY_T)
    api.set_isostasy = g("set_isostasy", ISO_T)
    if hasattr(lib, prefix + "_body_force_adjustment"):
        api.body_force_adjustment = g("body_force_adjustment", BFA_T)
    if hasattr(lib, prefix + "_last_error"):
        api.last_error = g("last_error", ERR_T)
    return api


def hip_api():
    lib = load_hip_lib()
    if lib.des_dev_device_count() < 1:
        raise DesError(31, "no HIP device visible; the device path has no CPU fallback")
    return api_from_lib(lib, "des_dev")


def run(host, device=0, quiet=True, api=None):
    """dynearthsol.cxx main(): init tail + time loop + output frames, on `host` (a Host).
    Returns RunStats; raises DesError with the reference's exit code if the run stopped."""
    assert isinstance(host, Host)
    lib = load_host_lib()
    lib.des_run.argtypes = [C.c_void_p, C.POINTER(EngineApi), C.c_int, C.c_int, C.POINTER(RunStats)]
    if api is None:
        api = hip_api()
    st = RunStats()
    rc = lib.des_run(host._h, C.byref(api), device, 1 if quiet else 0, C.byref(st))
    if rc != 0 and not st.remesh_needed:
        raise DesError(rc, "des_run stopped")
    return st


def run_with_remesher(make_host, remesher, device=0, quiet=True, api=None, max_rounds=100):
    """The whole-program round trip of include/des_run.h: `make_host(overrides)` builds the Host
    (overrides = None for the first round, the restart keys afterwards); where the loop stops for a
    remesh, `remesher(modelname, frame)` must leave the remeshed model as frame + 1 (a callable, or a
    command string run as `<command> <modelname> <frame>`); the run then restarts from that pair on
    a new engine.  Returns the list of RunStats, one per mesh."""
    import subprocess
    stats = []
    overrides = None
    for _ in range(max_rounds):
        host = make_host(overrides)
        model = host.cfg_string("sim.modelname")
        st = run(host, device=device, quiet=quiet, api=api)
        host.close()
        stats.append(st)
        if not st.remesh_needed:
            return stats
        if callable(remesher):
            remesher(model, st.last_frame)
        else:
            subprocess.check_call("%s %s %d" % (remesher, model, st.last_frame), shell=True)
        overrides = ("sim.is_restarting = yes\nsim.restarting_from_modelname = %s\nsim.restarting_from_frame = %d\n"
                     % (model, st.last_frame + 1))
    raise DesError(31, "the mesh needed remeshing more than %d times" % max_rounds)
