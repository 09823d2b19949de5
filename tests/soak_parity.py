#!/usr/bin/env python3
"""Long parity soak on the reference's 1M-tet TetGen mesh (test-3d-big.cfg at 460 m, evp):
device engine vs CPU oracle every 100 steps up to --steps, and (with --sensitivity, CPU only)
the oracle against itself after a 1-ulp change of its initial stress -- the yardstick for what
"equal" can mean on this model.  Not part of the test suite (minutes of CPU); results are
kept in profiles/.

  python tests/soak_parity.py [--steps 1000] [--ndims 2] [--per-call 100] [--sensitivity | --portable-libm] [--threads 16]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)          # (lives under tests/ because it drives the oracle: test infrastructure)
import bench                                             # noqa: E402
import dynearthsol_amd as des                            # noqa: E402
from oracle_binding import OracleEngine, load_oracle     # noqa: E402

FIELDS = ("COORD", "VEL", "STRESS", "TEMPERATURE", "PLSTRAIN")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--sensitivity", action="store_true")
    ap.add_argument("--portable-libm", action="store_true",
                    help="both sides use csrc/des_libm.hpp instead of ocml / glibc: expect zeros")
    ap.add_argument("--mesh-file", default=des.reference_mesh("test-3d-big-460"))
    ap.add_argument("--ndims", type=int, default=3, help="2: the bench's 2-D box (1.28M triangles, evp) instead of the 1M-tet mesh")
    ap.add_argument("--per-call", type=int, default=100, help="steps per des_dev_step call (and per comparison)")
    a = ap.parse_args()
    load_oracle(omp=True).des_oracle_set_threads(a.threads)
    if a.portable_libm:
        load_oracle(omp=True).des_oracle_set_libm(1)
        os.environ["DES_LIBM"] = "portable"
    ov = "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n"
    if a.ndims == 2:
        import cfgs
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, lx=400e3, lz=100e3, res=250.0)), ndims=2)
    else:
        host = des.Host(cfg_text=bench.BENCH_CFG.format(res="460.0", xlen="400e3"), overrides=ov, mesh_file=a.mesh_file)
    ref = OracleEngine(host, omp=True)
    if a.sensitivity:
        other = OracleEngine(host, omp=True)
        ref.init_from_host(host)
        other.init_from_host(host)
        other.upload("STRESS", np.nextafter(other.download("STRESS"), np.inf))
        print("# oracle vs oracle with initial stress moved by 1 ulp")
    else:
        other = des.DeviceEngine(host)
        assert other.init_from_host(host) == ref.init_from_host(host)
        print("# device engine (%s) vs oracle (%s)" % (os.environ.get("DES_LIBM", "des_libm.hpp: pow/exp = glibc's bits"),
                                                       "des_libm.hpp" if a.portable_libm else "the C library's libm"))
    print("# nnode %d nelem %d" % (host.mesh.nnode, host.mesh.nelem))
    t = time.time()
    for _ in range(a.steps // a.per_call):
        so, sr = other.step(a.per_call), ref.step(a.per_call)
        cols = []
        for f in FIELDS:
            x, y = other.download(f), ref.download(f)
            cols.append("%s %.1e" % (f.lower(), np.abs(x - y).max() / max(np.abs(y).max(), 1e-300)))
        print("step %5d  dt %s  %s  (%.0f s)" % (sr.steps, "equal" if so.dt == sr.dt else "%.1e" % abs(so.dt / sr.dt - 1),
                                                " ".join(cols), time.time() - t), flush=True)


if __name__ == "__main__":
    main()
