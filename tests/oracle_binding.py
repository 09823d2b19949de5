"""ctypes binding of the CPU oracle (oracle/libdes_oracle.so) -- tests only."""
import ctypes as C
import os
import subprocess

import numpy as np

from dynearthsol_amd import EngineBase, bind_engine_api, DesParams, DesMesh, REPO_ROOT

ORACLE_DIR = os.path.join(REPO_ROOT, "oracle")
_libs = {}


def cpu_budget():
    """CPUs this process may really use: the cgroup's quota when there is one (a box can show 256 CPUs and grant 16 --
    an OpenMP team of 256 on 16 is orders of magnitude slower than one of 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max": n = min(n, max(1, -(-int(parts[0]) // int(parts[1]))))
            else:
                q = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh: per = int(fh.read())
                if q > 0: n = min(n, max(1, -(-q // per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def load_oracle(omp=False, ndims=3):
    name = ("libdes_oracle2d_omp.so" if omp else "libdes_oracle2d.so") if ndims == 2 else "libdes_oracle_omp.so" if omp else "libdes_oracle.so"
    if name not in _libs:
        path = os.path.join(ORACLE_DIR, name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
        lib = C.CDLL(path)
        bind_engine_api(lib, "des_oracle")
        lib.des_oracle_create.restype = C.c_void_p
        lib.des_oracle_create.argtypes = [C.POINTER(DesParams), C.POINTER(DesMesh)]
        lib.des_oracle_threads.restype = C.c_int
        lib.des_oracle_set_threads.restype = C.c_int
        lib.des_oracle_set_threads.argtypes = [C.c_int]
        lib.des_oracle_set_libm.restype = C.c_int
        lib.des_oracle_set_libm.argtypes = [C.c_int]
        lib.des_oracle_libm_eval.restype = None
        lib.des_oracle_libm_eval.argtypes = [C.c_int, C.c_longlong] + [C.POINTER(C.c_double)] * 3
        lib.des_oracle_clib_eval.restype = None
        lib.des_oracle_clib_eval.argtypes = [C.c_int, C.c_longlong] + [C.POINTER(C.c_double)] * 3
        d6 = C.POINTER(C.c_double)
        lib.des_oracle_principal_values3.argtypes = [d6, d6]
        lib.des_oracle_principal_stresses3.argtypes = [d6, d6, d6]
        for f in ("dsyevh3", "dsyevq3"):
            getattr(lib, "des_oracle_" + f).argtypes = [d6, d6, d6]
        lib.des_oracle_dsyevc3.argtypes = [d6, d6]
        lib.des_oracle_elasto_plastic.restype = C.c_double
        lib.des_oracle_elasto_plastic.argtypes = [C.c_double] * 7 + [d6, d6, C.POINTER(C.c_int)]
        lib.des_oracle_maxwell.argtypes = [C.c_double] * 5 + [d6, d6]
        if omp: lib.des_oracle_set_threads(min(16, cpu_budget()))      # (callers may set another count afterwards)
        _libs[name] = lib
    return _libs[name]


class OracleEngine(EngineBase):
    prefix = "des_oracle"

    def __init__(self, host, omp=False):
        lib = load_oracle(omp, getattr(host, "ndims", 3))
        h = lib.des_oracle_create(C.byref(host.params), C.byref(host.mesh))
        if not h:
            raise RuntimeError("des_oracle_create failed")
        super().__init__(lib, h)
        self._host = host


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def oracle_libm_eval(fn, x, y=None, clib=False, omp=False):
    """CPU build of the portable libm (csrc/des_libm.hpp) through the oracle library; with
    clib=True the C library's own function (std::pow ...), i.e. what the oracle calls by default."""
    from dynearthsol_amd import LIBM_FN
    lib = load_oracle(omp)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
    f = lib.des_oracle_clib_eval if clib else lib.des_oracle_libm_eval
    f(LIBM_FN[fn], x.size, dptr(x), dptr(y) if y is not None else None, dptr(out))
    return out


class portable_libm:
    """with portable_libm(): the oracle (both builds) and every device engine CREATED inside use
    the portable libm instead of glibc / ocml."""

    def __enter__(self):
        self._old = [(lib, lib.des_oracle_set_libm(1)) for lib in (load_oracle(False), load_oracle(True), load_oracle(ndims=2))]
        self._env = os.environ.get("DES_LIBM")
        os.environ["DES_LIBM"] = "portable"
        return self

    def __exit__(self, *exc):
        for lib, old in self._old:
            lib.des_oracle_set_libm(old)
        if self._env is None:
            os.environ.pop("DES_LIBM", None)
        else:
            os.environ["DES_LIBM"] = self._env
        return False
