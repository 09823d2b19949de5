#!/usr/bin/env python3
"""Where does a last-bit perturbation of the creep law go on the headline model?  (CPU only.)

Two engines step the reference's 1,001,310-tet TetGen mesh of test-3d-big.cfg (evp) side by
side: A = the OpenMP oracle with glibc's libm; B = a second instance of it with des_libm.hpp
(--perturb portable), or after a 1-ulp nudge of one stress component (--perturb ulp), or the HIP
engine with ocml (--perturb device; needs the GPU) / with des_libm.hpp (--perturb device-portable).  Every --every steps the script prints, per field, the
relative difference and WHERE it is largest, plus the counters of the step's discontinuities:

  * damping (fields.cxx:497-505, option 1): f -= 0.8 copysign(f, v) if |v| > 1e-13 -- the number
    of velocity components whose `|v| > 1e-13` test or whose sign differs between A and B;
  * the viscosity clamp (matprops.cxx:370-375): elements sitting on min/max viscosity in one run
    and not in the other;
  * the evp switch (rheology.cxx:908-918): not observable from the state, but it is continuous
    (both candidates agree where it flips).

Results: profiles/r02_divergence_tetgen1M.txt, DESIGN.md section 2.  Not part of the suite.

  python tests/soak_diverge.py [--steps 800] [--every 20] [--perturb portable|ulp] [--mesh-file F]
"""
import argparse
import ctypes as C
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
import dynearthsol_amd as des                            # noqa: E402
import oracle_binding as ob                              # noqa: E402


def second_instance():
    """The OpenMP oracle loaded a second time from a copy, so that its libm switch is its own."""
    src = os.path.join(ob.ORACLE_DIR, "libdes_oracle_omp.so")
    tmp = os.path.join(tempfile.mkdtemp(prefix="des_oracle2_"), "libdes_oracle_omp_b.so")
    shutil.copy(src, tmp)
    lib = C.CDLL(tmp)
    des.bind_engine_api(lib, "des_oracle")
    lib.des_oracle_create.restype = C.c_void_p
    lib.des_oracle_create.argtypes = [C.POINTER(des.DesParams), C.POINTER(des.DesMesh)]
    lib.des_oracle_set_threads.argtypes = [C.c_int]
    lib.des_oracle_set_libm.argtypes = [C.c_int]
    return lib


class EngineB(ob.OracleEngine):
    def __init__(self, host, lib):
        h = lib.des_oracle_create(C.byref(host.params), C.byref(host.mesh))
        des.EngineBase.__init__(self, lib, h)
        self._host = host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=800)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--perturb", choices=("portable", "ulp", "device", "device-portable"), default="portable")
    ap.add_argument("--fine-from", type=int, default=-1, help="from this step on, report every --fine-every steps")
    ap.add_argument("--fine-every", type=int, default=5)
    ap.add_argument("--mesh-file", default=des.reference_mesh("test-3d-big-460"))
    ap.add_argument("--resolution", type=float, default=460.0)
    a = ap.parse_args()
    libA = ob.load_oracle(omp=True)
    libA.des_oracle_set_threads(a.threads)
    on_device = a.perturb.startswith("device")
    if not on_device:
        libB = second_instance()
        libB.des_oracle_set_threads(a.threads)
    if a.mesh_file and os.path.exists(a.mesh_file):
        ov = "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n"
        host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(a.resolution), xlen="400e3"), overrides=ov, mesh_file=a.mesh_file)
    else:
        host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(a.resolution), xlen="400e3"))
    A = ob.OracleEngine(host, omp=True)
    if on_device:
        os.environ["DES_LIBM"] = "portable" if a.perturb == "device-portable" else "ocml"
        B = des.DeviceEngine(host)
    else:
        B = EngineB(host, libB)
    assert A.init_from_host(host) == B.init_from_host(host)
    if on_device:
        print("# A: oracle with glibc's libm; B: HIP engine with %s" % ("des_libm.hpp (glibc's pow/exp bits)" if a.perturb == "device-portable" else "ocml"))
    elif a.perturb == "portable":
        libB.des_oracle_set_libm(1)
        print("# A: oracle with glibc pow/exp; B: oracle with des_libm.hpp (differs in the last bit of ~0.1 % of the calls)")
    else:
        s = B.download("STRESS")
        s[len(s) // 2] = np.nextafter(s[len(s) // 2], np.inf)
        B.upload("STRESS", s)
        print("# A: oracle; B: oracle with ONE stress component moved by 1 ulp")
    nn, ne = host.nnode, host.nelem
    print("# nnode %d nelem %d; visc_min %.3g visc_max %.3g" % (nn, ne, host.params.visc_min, host.params.visc_max))
    coord0 = A.download("COORD").reshape(3, nn)
    t = time.time()
    small = 1e-13
    done = 0
    while done < a.steps:
        every = a.fine_every if 0 <= a.fine_from <= done else a.every
        sa, sb = A.step(every), B.step(every)
        done += every
        out = ["step %5d dt %s" % (sa.steps, "equal" if sa.dt == sb.dt else "%.1e" % abs(sb.dt / sa.dt - 1))]
        va, vb = A.download("VEL"), B.download("VEL")
        for f in ("VEL", "STRESS", "COORD", "VISCOSITY"):
            x, y = (va, vb) if f == "VEL" else (A.download(f), B.download(f))
            d = np.abs(x - y)
            k = int(d.argmax())
            n = nn if f in ("VEL", "COORD") else ne
            out.append("%s %.1e @%d(c%d)" % (f.lower()[:5], d[k] / max(np.abs(x).max(), 1e-300), k % n, k // n))
            if f == "VEL":
                kn = k % nn
                out.append("x=(%.0f,%.0f,%.0f)" % tuple(coord0[:, kn]))
                out.append("v=%.2e" % x[k])
            if f == "VISCOSITY":
                lo, hi = host.params.visc_min, host.params.visc_max
                out.append("clamp_lo %d/%d clamp_hi %d/%d clamp_differs %d" % (
                    (x <= lo).sum(), (y <= lo).sum(), (x >= hi).sum(), (y >= hi).sum(),
                    ((x <= lo) != (y <= lo)).sum() + ((x >= hi) != (y >= hi)).sum()))
        thr = (np.abs(va) > small) != (np.abs(vb) > small)
        sgn = (np.signbit(va) != np.signbit(vb)) & (np.abs(va) > small) & (np.abs(vb) > small)
        near = (np.abs(va) > small / 2) & (np.abs(va) < small * 2)
        out.append("damping: thr_differs %d sign_differs %d near_thr %d" % (thr.sum(), sgn.sum(), near.sum()))
        out.append("(%.0f s)" % (time.time() - t))
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
