"""dynearthsol_amd -- MI355X-native explicit time-stepper behind DynEarthSol's host surface.

Python is plumbing only: the package loads two in-tree shared libraries through ctypes,

* ``libdes_host.so`` -- C++ host side (``.cfg`` front-end, mesh topology, initial
  conditions, driver loop; mirrors input.cxx / mesh.cxx / ic.cxx / dynearthsol.cxx), and
* ``libdes_hip.so``  -- the hand-written HIP kernels for gfx950 behind the C-ABI of
  ``include/des_dev.h``.

There is no CPU fallback: constructing a :class:`DeviceEngine` without the HIP library or
without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

from ._structs import DesMesh, DesParams, DesScalars, F, FIELDS, INT_FIELDS

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HOST_LIB_PATH = os.path.join(_HERE, "libdes_host.so")
HIP_LIB_PATH = os.environ.get("DES_HIP_LIB", os.path.join(_HERE, "libdes_hip.so"))   # override: A/B of kernel builds

_host_lib = None
_hip_lib = None

DATA_DIR = os.path.join(REPO_ROOT, "data")


def reference_mesh(name="test-3d-big-460"):
    """Path of a mesh made by the reference's own TetGen and kept under data/ as `<name>.desmesh.xz`
    (recipe: `make -C oracle refmesh`, which needs /root/reference; the file itself is data and
    travels).  `test-3d-big-460` is test-3d-big.cfg's box at mesh.resolution = 460 m: 1,001,310 tets /
    185,637 nodes, the headline mesh of SURVEY.md 8(d).  Unpacked once, next to the archive."""
    path = os.path.join(DATA_DIR, name + ".desmesh")
    if not os.path.exists(path):
        import lzma
        import shutil
        if not os.path.exists(path + ".xz"):
            return None
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with lzma.open(path + ".xz", "rb") as src, open(tmp, "wb") as dst:
            shutil.copyfileobj(src, dst, 1 << 24)
        os.replace(tmp, path)                 # atomic: several ranks may unpack at once
    return path


class DesError(RuntimeError):
    """Carries the reference's ExitCode number (utils.hpp:20-55)."""

    def __init__(self, code, msg):
        super().__init__("[DES exit %d] %s" % (code, msg))
        self.code = code


def load_host_lib():
    global _host_lib
    if _host_lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise DesError(31, "libdes_host.so is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(HOST_LIB_PATH)
        lib.des_host_create.restype = C.c_void_p
        lib.des_host_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        lib.des_host_create_from_string.restype = C.c_void_p
        lib.des_host_create_from_string.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        lib.des_host_destroy.argtypes = [C.c_void_p]
        lib.des_host_params.restype = C.POINTER(DesParams)
        lib.des_host_params.argtypes = [C.c_void_p]
        lib.des_host_mesh.restype = C.POINTER(DesMesh)
        lib.des_host_mesh.argtypes = [C.c_void_p]
        lib.des_host_array.restype = C.c_void_p
        lib.des_host_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
        lib.des_host_cfg_int.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        lib.des_host_cfg_double.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
        lib.des_host_save_mesh.argtypes = [C.c_void_p, C.c_char_p]
        lib.des_host_last_error.restype = C.c_char_p
        _host_lib = lib
    return _host_lib


def bind_engine_api(lib, prefix):
    """Declare the argument types of an engine library (`des_dev_*` or, in tests, the
    oracle's `des_oracle_*`, which has the same call shape)."""
    g = lambda name: getattr(lib, prefix + "_" + name)
    g("destroy").argtypes = [C.c_void_p]
    g("upload").argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
    g("download").argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
    g("field_count").restype = C.c_longlong
    g("field_count").argtypes = [C.c_void_p, C.c_int]
    g("set_clock").argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_longlong]
    g("set_isostasy").argtypes = [C.c_void_p, C.c_int]
    g("init_geometry").argtypes = [C.c_void_p]
    g("compute_dt").argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    g("step").argtypes = [C.c_void_p, C.c_int, C.POINTER(DesScalars)]
    g("check_nan").argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    return lib


def load_hip_lib():
    """Load the HIP extension; fails loudly if it is missing (no CPU fallback)."""
    global _hip_lib
    if _hip_lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise DesError(31, "libdes_hip.so (HIP kernels for gfx950) is not built; "
                               "there is no CPU fallback for the device path")
        lib = C.CDLL(HIP_LIB_PATH)
        bind_engine_api(lib, "des_dev")
        lib.des_dev_device_count.restype = C.c_int
        lib.des_dev_create.restype = C.c_void_p
        lib.des_dev_create.argtypes = [C.c_int, C.POINTER(DesParams), C.POINTER(DesMesh), C.POINTER(C.c_int)]
        lib.des_dev_sync.argtypes = [C.c_void_p]
        lib.des_dev_timer_start.argtypes = [C.c_void_p]
        lib.des_dev_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.des_dev_profile_enable.argtypes = [C.c_void_p, C.c_int]
        lib.des_dev_profile_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double),
                                             C.POINTER(C.c_longlong)]
        lib.des_dev_algorithmic_bytes_per_step.restype = C.c_double
        lib.des_dev_algorithmic_bytes_per_step.argtypes = [C.c_void_p]
        lib.des_dev_last_error.restype = C.c_char_p
        dp = C.POINTER(C.c_double)
        lib.des_dev_libm_eval.argtypes = [C.c_int, C.c_int, C.c_longlong, dp, dp, dp]
        ip = C.POINTER(C.c_int)
        lib.des_dev_eigen_eval.argtypes = [C.c_int, C.c_int, C.c_int, C.c_longlong, dp, dp, dp, ip]
        lib.des_dev_elasto_plastic_eval.argtypes = [C.c_int, C.c_int, C.c_longlong, dp, dp, dp, dp, ip]
        _hip_lib = lib
    return _hip_lib


LIBM_FN = {"pow": 0, "exp": 1, "sin": 2, "cos": 3, "tan": 4, "atan2": 5, "sincos_s": 6, "sincos_c": 7}       # DES_LIBM_* (des_dev.h)


def copy_ceiling(nbytes=1 << 30, reps=20, device=0):
    """Measured streaming-copy bandwidth of the device in GB/s (read + written bytes over time)."""
    lib = load_hip_lib()
    lib.des_dev_copy_ceiling.argtypes = [C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_double)]
    g = C.c_double(0)
    rc = lib.des_dev_copy_ceiling(device, nbytes, reps, C.byref(g))
    if rc:
        raise DesError(rc, lib.des_dev_last_error().decode())
    return g.value


def plane_ceiling(nr=18, nw=15, nelem=8800000, reps=10, device=0):
    """GB/s of a bytes-only kernel in the stress update's memory shape (nr planes read, nw written, 8 B per lane and plane)."""
    lib = load_hip_lib()
    lib.des_dev_plane_ceiling.argtypes = [C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_double)]
    g = C.c_double(0)
    rc = lib.des_dev_plane_ceiling(device, nr, nw, nelem, reps, C.byref(g))
    if rc:
        raise DesError(rc, lib.des_dev_last_error().decode())
    return g.value


def libm_eval(fn, x, y=None, device=0):
    """One function of the portable libm (csrc/des_libm.hpp) evaluated on the GPU."""
    import numpy as np
    lib = load_hip_lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert y.shape == x.shape
        yp = y.ctypes.data_as(dp)
    rc = lib.des_dev_libm_eval(device, LIBM_FN[fn], x.size, x.ctypes.data_as(dp), yp, out.ctypes.data_as(dp))
    if rc:
        raise DesError(rc, lib.des_dev_last_error().decode())
    return out


def eigen_eval(fn, a, libm="ocml", device=0):
    """The device build of the 3x3 solvers (des_dev_eigen_eval): fn in 'c', 'h', 'q'; a[n][6] =
    {A00, A11, A22, A01, A02, A12}.  Returns (w[n][3], q[n][3][3] or None, branch[n])."""
    lib = load_hip_lib()
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 6)
    n = a.shape[0]
    w, q, br = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros(n, dtype=np.int32)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rc = lib.des_dev_eigen_eval(device, "chq".index(fn), {"ocml": 0, "portable": 1}[libm], n, a.ctypes.data_as(dp),
                                w.ctypes.data_as(dp), q.ctypes.data_as(dp), br.ctypes.data_as(ip))
    if rc:
        raise DesError(rc, lib.des_dev_last_error().decode())
    return w, (None if fn == "c" else q), br


def elasto_plastic_eval(props, de, s, libm="ocml", device=0):
    """n calls of the device's elasto_plastic (des_dev_elasto_plastic_eval): props[n][7] = {bulkm,
    shearm, amc, anphi, anpsi, hardn, ten_max}, de[n][6], s[n][6].  Returns (s_new, depls, mode)."""
    lib = load_hip_lib()
    props = np.ascontiguousarray(props, dtype=np.float64).reshape(-1, 7)
    n = props.shape[0]
    de = np.ascontiguousarray(de, dtype=np.float64).reshape(n, 6)
    s = np.array(s, dtype=np.float64).reshape(n, 6).copy()
    depls, mode = np.zeros(n), np.zeros(n, dtype=np.int32)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rc = lib.des_dev_elasto_plastic_eval(device, {"ocml": 0, "portable": 1}[libm], n, props.ctypes.data_as(dp), de.ctypes.data_as(dp),
                                         s.ctypes.data_as(dp), depls.ctypes.data_as(dp), mode.ctypes.data_as(ip))
    if rc:
        raise DesError(rc, lib.des_dev_last_error().decode())
    return s, depls, mode


class Host:
    """Host-side model: parsed ``.cfg``, mesh topology and initial fields
    (get_input_parameters + init(), input.cxx:1503 / dynearthsol.cxx:159-228)."""

    def __init__(self, cfg_path=None, cfg_text=None, overrides=None, mesh_file=None, ndims=3):
        """ndims: which build of the reference this host stands for (3: tets, 2: triangles)."""
        lib = load_host_lib()
        err = C.c_int(0)
        ov = overrides.encode() if overrides else None
        mf = mesh_file.encode() if mesh_file else None
        self.ndims = ndims
        if ndims != 3:
            lib.des_host_create_nd.restype = C.c_void_p
            lib.des_host_create_nd.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
            h = lib.des_host_create_nd(ndims, cfg_path.encode() if cfg_path is not None else None,
                                       None if cfg_path is not None else (cfg_text or "").encode(), ov, mf, C.byref(err))
        elif cfg_path is not None:
            h = lib.des_host_create(cfg_path.encode(), ov, mf, C.byref(err))
        else:
            h = lib.des_host_create_from_string((cfg_text or "").encode(), ov, mf, C.byref(err))
        if not h:
            raise DesError(err.value, lib.des_host_last_error().decode())
        self._lib, self._h = lib, C.c_void_p(h)
        self.params = lib.des_host_params(self._h).contents
        self.mesh = lib.des_host_mesh(self._h).contents
        self.nnode, self.nelem = self.mesh.nnode, self.mesh.nelem

    def array(self, name):
        n = C.c_longlong(0)
        p = self._lib.des_host_array(self._h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        ints = ("elemmarkers", "connectivity", "segment", "segflag", "markerset.elem", "markerset.mattype", "markerset.id")
        ctype = C.c_int if name in ints else C.c_double
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(n.value,)).copy()

    def cfg_int(self, key):
        v = C.c_int(0)
        if self._lib.des_host_cfg_int(self._h, key.encode(), C.byref(v)):
            raise KeyError(key)
        return v.value

    def cfg_string(self, key):
        buf = C.create_string_buffer(512)
        self._lib.des_host_cfg_string.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        if self._lib.des_host_cfg_string(self._h, key.encode(), buf, 512):
            raise KeyError(key)
        return buf.value.decode()

    def cfg_double(self, key):
        v = C.c_double(0)
        if self._lib.des_host_cfg_double(self._h, key.encode(), C.byref(v)):
            raise KeyError(key)
        return v.value

    def save_mesh(self, path):
        rc = self._lib.des_host_save_mesh(self._h, path.encode())
        if rc:
            raise DesError(rc, self._lib.des_host_last_error().decode())

    def close(self):
        if self._h:
            self._lib.des_host_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EngineBase:
    """Shared ctypes plumbing for an engine with the des_dev call shape."""

    prefix = None

    def __init__(self, lib, handle):
        self._lib, self._h = lib, C.c_void_p(handle)

    def _f(self, name):
        return getattr(self._lib, self.prefix + "_" + name)

    def _check(self, rc, what):
        if rc:
            raise DesError(rc, "%s_%s failed" % (self.prefix, what))

    def field_count(self, field):
        return self._f("field_count")(self._h, F[field])

    def upload(self, field, arr):
        dt = np.int32 if field in INT_FIELDS else np.float64
        a = np.ascontiguousarray(arr, dtype=dt).ravel()
        self._check(self._f("upload")(self._h, F[field], a.ctypes.data_as(C.c_void_p), a.size), "upload(%s)" % field)

    def download(self, field):
        n = self.field_count(field)
        a = np.empty(n, dtype=np.int32 if field in INT_FIELDS else np.float64)
        self._check(self._f("download")(self._h, F[field], a.ctypes.data_as(C.c_void_p), n), "download(%s)" % field)
        return a

    def set_isostasy(self, on):
        """isostasy_adjustment mode: step() runs that loop's body instead of a time step."""
        self._check(self._f("set_isostasy")(self._h, int(bool(on))), "set_isostasy")

    def body_force_adjustment(self):
        """initial_body_force_adjustment (dynearthsol.cxx:546-591); returns the scalars afterwards."""
        sc = DesScalars()
        f = self._f("body_force_adjustment")
        f.argtypes = [C.c_void_p, C.c_void_p]
        self._check(f(self._h, C.byref(sc)), "body_force_adjustment")
        return sc

    def set_clock(self, dt, time=0.0, steps=0):
        self._check(self._f("set_clock")(self._h, dt, time, steps), "set_clock")

    def init_geometry(self):
        self._check(self._f("init_geometry")(self._h), "init_geometry")

    def compute_dt(self):
        dt = C.c_double(0)
        self._check(self._f("compute_dt")(self._h, C.byref(dt)), "compute_dt")
        return dt.value

    def step(self, nsteps, want_scalars=True):
        sc = DesScalars()
        self._check(self._f("step")(self._h, nsteps, C.byref(sc) if want_scalars else None), "step")
        return sc if want_scalars else None

    def check_nan(self):
        n = C.c_longlong(0)
        self._f("check_nan")(self._h, C.byref(n))
        return n.value

    # ---- domain decomposition hooks (same call shape on the device engine and the oracle)
    def set_halo(self, part):
        f = self._f("set_halo")
        if self.prefix == "des_oracle":
            f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
            self._check(f(self._h, part.owned[0], part.owned[1], part.host.nnode), "set_halo")
            g = self._f("set_owned_global")
            g.argtypes = [C.c_void_p, C.c_int]
            self._check(g(self._h, int(part.halo.owned_global_begin)), "set_owned_global")
        else:
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            self._check(f(self._h, C.byref(part.halo), part.host.nnode), "set_halo")
        self._part = part

    def phase(self, ph):
        f = self._f("phase")
        f.argtypes = [C.c_void_p, C.c_int]
        return f(self._h, ph)

    def residual_blocks(self):
        """(first global block, partials) of the partition-independent residual's blocks this rank owns (des_params.h:
        DES_RES_BLOCK) -- the pseudo-transient loop's driver puts all ranks' together and hands them to residual_set"""
        f = self._f("residual_blocks")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        first, count = C.c_int(0), C.c_int(0)
        f(self._h, None, 0, C.byref(first), C.byref(count))
        out = np.zeros(max(count.value, 1))
        self._check(f(self._h, out.ctypes.data_as(C.c_void_p), count.value, C.byref(first), C.byref(count)), "residual_blocks")
        return first.value, out[:count.value]

    def residual_set(self, blocks):
        """the fixed-shape sum over the GLOBAL block array: sets and returns l2_residual"""
        f = self._f("residual_set")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        blocks = np.ascontiguousarray(blocks, dtype=np.float64)
        l2 = C.c_double(0)
        self._check(f(self._h, blocks.ctypes.data_as(C.c_void_p), len(blocks), C.byref(l2)), "residual_set")
        return l2.value

    def halo_pack(self, kind, idx, width):
        f = self._f("halo_pack")
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        buf = np.empty(len(idx) * width)
        self._check(f(self._h, kind, idx.ctypes.data_as(C.c_void_p), len(idx), buf.ctypes.data_as(C.c_void_p)), "halo_pack")
        return buf

    def halo_unpack(self, kind, idx, buf):
        f = self._f("halo_unpack")
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        self._check(f(self._h, kind, idx.ctypes.data_as(C.c_void_p), len(idx), buf.ctypes.data_as(C.c_void_p)), "halo_unpack")

    def wall_get(self):
        f = self._f("wall_get")
        f.argtypes = [C.c_void_p, C.c_void_p]
        out = np.empty(3)
        self._check(f(self._h, out.ctypes.data_as(C.c_void_p)), "wall_get")
        return out

    def wall_set(self, red):
        f = self._f("wall_set")
        f.argtypes = [C.c_void_p, C.c_void_p]
        red = np.ascontiguousarray(red, dtype=np.float64)
        self._check(f(self._h, red.ctypes.data_as(C.c_void_p)), "wall_set")

    def dt_partials(self, recompute):
        f = self._f("dt_partials")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        out = np.empty(6)
        self._check(f(self._h, out.ctypes.data_as(C.c_void_p), int(recompute)), "dt_partials")
        return out

    def dt_finalize(self, red):
        f = self._f("dt_finalize")
        f.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        red = np.ascontiguousarray(red, dtype=np.float64)
        dt = C.c_double(0)
        self._check(f(self._h, red.ctypes.data_as(C.c_void_p), C.byref(dt)), "dt_finalize")
        return dt.value

    def close(self):
        if self._h:
            self._f("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_from_host(self, host):
        """Replays init() + the first compute_dt of main() (dynearthsol.cxx:175-221, 643)."""
        coord = host.array("coord")
        self.upload("COORD", coord)
        self.upload("COORD0", coord)
        self.upload("ELEMMARKERS", host.array("elemmarkers"))
        self.upload("VEL", host.array("vel"))
        # compute_volume, volume_old = volume, apply_vbcs, compute_mass -- with T still 0,
        # exactly as init() orders them (compute_mass precedes initial_temperature)
        self.init_geometry()
        self.upload("TEMPERATURE", host.array("temperature"))
        self.upload("RADIOGENIC", host.array("radiogenic"))
        self.upload("STRESS", host.array("stress"))
        self.upload("STRAIN", host.array("strain"))
        self.upload("PLSTRAIN", host.array("plstrain"))
        self.upload("VISCOSITY", host.array("viscosity"))
        if getattr(host, "ndims", 3) == 2:
            self.upload("STRESSYY", host.array("stressyy"))
        return self.compute_dt()


class DeviceEngine(EngineBase):
    """The MI355X engine behind include/des_dev.h."""

    prefix = "des_dev"

    def __init__(self, host, device=0):
        """`host` is a Host (whole mesh) or a decomp.Partition (one rank's local mesh)."""
        lib = load_hip_lib()
        if lib.des_dev_device_count() <= device:
            raise DesError(31, "no HIP device %d visible; the device path has no CPU fallback" % device)
        err = C.c_int(0)
        h = lib.des_dev_create(device, C.byref(host.params), C.byref(host.mesh), C.byref(err))
        if not h:
            raise DesError(err.value, lib.des_dev_last_error().decode())
        super().__init__(lib, h)
        self._host = host      # keeps the mesh arrays alive

    def sync(self):
        self._check(self._lib.des_dev_sync(self._h), "sync")

    def timer_start(self):
        self._check(self._lib.des_dev_timer_start(self._h), "timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        self._check(self._lib.des_dev_timer_stop(self._h, C.byref(ms)), "timer_stop")
        return ms.value

    def profile_enable(self, on=True):
        self._check(self._lib.des_dev_profile_enable(self._h, int(on)), "profile_enable")

    def profile_read(self, cap=64):
        names = ((C.c_char * 64) * cap)()
        ms = (C.c_double * cap)()
        calls = (C.c_longlong * cap)()
        n = self._lib.des_dev_profile_read(self._h, cap, names, ms, calls)
        return [(names[i].value.decode(), ms[i], calls[i]) for i in range(n)]

    def algorithmic_bytes_per_step(self):
        return self._lib.des_dev_algorithmic_bytes_per_step(self._h)

    def exchange(self):
        self._lib.des_dev_exchange.argtypes = [C.c_void_p]
        self._check(self._lib.des_dev_exchange(self._h), "exchange")

    def comm_info(self):
        """{'rccl_ranks': ncclCommCount of the attached communicator (0: none), 'rank', 'overlapped'}"""
        n, r, o = C.c_int(0), C.c_int(0), C.c_int(0)
        self._lib.des_dev_comm_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        self._check(self._lib.des_dev_comm_info(self._h, C.byref(n), C.byref(r), C.byref(o)), "comm_info")
        return {"rccl_ranks": n.value, "rank": r.value, "overlapped": bool(o.value)}

    def set_overlap(self, on):
        """in-order / overlapped multi-GPU schedule from the next step() call on (the same on every rank)"""
        self._lib.des_dev_set_overlap.argtypes = [C.c_void_p, C.c_int]
        self._check(self._lib.des_dev_set_overlap(self._h, int(bool(on))), "set_overlap")

    def comm_init(self, dist, rank, world):
        """Attach an RCCL communicator: rank 0 creates the ncclUniqueId, torch.distributed only
        carries those 128 bytes; all halo traffic then stays inside the engine."""
        import torch
        idbuf = (C.c_ubyte * 128)()
        if rank == 0:
            self._lib.des_dev_comm_unique_id.argtypes = [C.c_void_p]
            self._check(self._lib.des_dev_comm_unique_id(idbuf), "comm_unique_id")
        t = torch.tensor(list(idbuf), dtype=torch.uint8)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0)
        idbuf = (C.c_ubyte * 128)(*t.cpu().tolist())
        self._lib.des_dev_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        rc = self._lib.des_dev_comm_init(self._h, world, rank, idbuf)
        if rc:
            raise DesError(rc, self._lib.des_dev_last_error().decode())

    def comm_selfcheck(self, world):
        """des_dev_comm_selfcheck: rank count, the exchange's own messages filled with a verifiable pattern, the three
        reductions -- collective, before the first step; raises DesError naming what failed"""
        self._lib.des_dev_comm_selfcheck.argtypes = [C.c_void_p, C.c_int]
        rc = self._lib.des_dev_comm_selfcheck(self._h, int(world))
        if rc:
            raise DesError(rc, self._lib.des_dev_last_error().decode())


def config_string():
    """the engine's environment switches that are set in this process and that the library has read so far
    (des_dev_config_string): 'NAME=value NAME=value', '' when every switch is at its default"""
    lib = load_hip_lib()
    lib.des_dev_config_string.argtypes = [C.c_char_p, C.c_int]
    lib.des_dev_config_string.restype = C.c_int
    n = lib.des_dev_config_string(None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.des_dev_config_string(buf, n + 1)
    return buf.value.decode()


__all__ = ["Host", "DeviceEngine", "reference_mesh", "EngineBase", "DesError", "DesParams", "DesMesh", "DesScalars",
           "F", "FIELDS", "load_host_lib", "load_hip_lib", "bind_engine_api", "config_string"]
