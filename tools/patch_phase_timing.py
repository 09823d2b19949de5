#!/usr/bin/env python3
"""Where a workgroup of the patch passes EN1 / EN3 spends its life: 100-MHz wall-clock stamps taken at the phase boundaries
by every workgroup of the LAST launch of each pass (instrumented build: tools/build_variant.sh stamps -DDES_STAMPS,
selected with DES_HIP_LIB; the shipped library holds no stamps).

    DES_HIP_LIB=build/variants/stamps.so python tools/patch_phase_timing.py [--ranks 8] [--steps 60]     (MI355X box)

--ranks N: the middle rank's shard of the headline mesh cut N ways (as tools/time_shard.py) instead of the whole mesh.
Prints per pass: the phases' mean (p10 / p50 / p90) in us per workgroup, the launch's span, and the mean number of
workgroups in flight (sum of lifetimes / span)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
import dynearthsol_amd as des                          # noqa: E402
from dynearthsol_amd.decomp import Partition           # noqa: E402

SLOTS, WG = 8, 8192
PHASES = {
    0: ("EN1", [("staging (records -> LDS, to the barrier)", 0, 1), ("element phase (own rounds)", 1, 2), ("barrier wait", 2, 3),
                ("node phase, wavefront 0 {volume_n, dvoldt}", 3, 4), ("node phase, wavefront 1 {mass}", 3, 5),
                ("node phase, wavefront 2 {tmass, T}", 3, 6)]),
    1: ("EN3", [("staging + first element's loads, to the barrier", 0, 1), ("element phase (own elements)", 1, 2), ("barrier wait", 2, 3),
                ("force sums (to the second barrier when split)", 3, 4), ("rest of the nodal update", 4, 5), ("residual partial", 5, 6)]),
}


def main():
    argv = sys.argv[1:]
    ranks, steps = 1, 60
    if "--ranks" in argv:
        ranks = int(argv[argv.index("--ranks") + 1])
    if "--steps" in argv:
        steps = int(argv[argv.index("--steps") + 1])
    mesh = des.reference_mesh("test-3d-big-460")
    ov = "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n"
    host = des.Host(cfg_text=bench.BENCH_CFG.format(res="460.0", xlen="400e3"), overrides=ov, mesh_file=mesh)
    if ranks > 1:
        part = Partition(host, ranks, ranks // 2)
        dev = des.DeviceEngine(part)
        coord = part.local("coord")
        dev.upload("COORD", coord); dev.upload("COORD0", coord)
        dev.upload("ELEMMARKERS", part.local("elemmarkers")); dev.upload("VEL", part.local("vel"))
        dev.init_geometry()
        for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"),
                        ("STRAIN", "strain"), ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity")):
            dev.upload(f, part.local(name))
        dev.compute_dt()
        src = part
    else:
        dev = des.DeviceEngine(host)
        dev.init_from_host(host)
        src = host
    # the call ends on a fused step's launches: steps chosen so that the last EN1 / EN3 launches are interior ones
    dev.step(steps + 3, want_scalars=False)
    dev.sync()
    lib = des.load_hip_lib()
    if not hasattr(lib, "des_dev_debug_stamps"):
        sys.exit("this library has no stamps: build one with tools/build_variant.sh stamps -DDES_STAMPS and set DES_HIP_LIB")
    lib.des_dev_debug_stamps.argtypes = [C.c_int, C.c_void_p, C.c_int]
    print("# %s: %d tets / %d nodes; library %s" % ("shard of %d" % ranks if ranks > 1 else "whole mesh", src.nelem, src.nnode,
                                                     os.environ.get("DES_HIP_LIB", "default")))
    for p, (name, phases) in PHASES.items():
        buf = np.zeros(SLOTS * WG, dtype=np.uint64)
        n = lib.des_dev_debug_stamps(p, buf.ctypes.data, buf.size)
        assert n == buf.size, n
        t = buf.reshape(SLOTS, WG).astype(np.float64) * 0.01       # 100 MHz -> us
        live = t[0] > 0
        if not live.any():
            print("%s: no stamps" % name); continue
        t = t[:, live]
        ends = np.max(t[1:7], axis=0)
        span = ends.max() - t[0].min()
        life = ends - t[0]
        print("%s: %d workgroups, span %.1f us, mean life %.2f us (p10 %.2f / p50 %.2f / p90 %.2f), %.0f in flight on average"
              % (name, t.shape[1], span, life.mean(), *np.percentile(life, [10, 50, 90]), life.sum() / span))
        for label, a, b in phases:
            ok = (t[a] > 0) & (t[b] > 0)
            if not ok.any():
                continue
            d = (t[b] - t[a])[ok]
            print("    %-52s %6.2f (%5.2f / %5.2f / %5.2f)" % (label, d.mean(), *np.percentile(d, [10, 50, 90])))
    # the pipelined stress update: per wavefront, sums over its tiles
    buf = np.zeros(SLOTS * WG, dtype=np.uint64)
    if lib.des_dev_debug_stamps(2, buf.ctypes.data, buf.size) == buf.size:
        t = buf.reshape(SLOTS, WG).astype(np.float64)
        live = t[6] > 0
        if live.any():
            t = t[:, live]
            ntile = t[6]
            life = (t[1] - t[0]) * 0.01
            print("E2<GEO> pipelined: %d wavefronts, %.1f tiles each (%d..%d), life %.1f us (p10 %.1f / p90 %.1f), span %.1f us"
                  % (t.shape[1], ntile.mean(), ntile.min(), ntile.max(), life.mean(), *np.percentile(life, [10, 90]),
                     (t[1].max() - t[0].min()) * 0.01))
            for label, k in (("wait for this tile's DMA", 2), ("LDS reads, gathers issued, next DMA issued", 3), ("wait for the gathers", 4),
                             ("arithmetic + stores issued", 5)):
                d = t[k] * 0.01 / ntile
                print("    %-52s %6.2f us per tile (%5.2f / %5.2f / %5.2f)" % (label, d.mean(), *np.percentile(d, [10, 50, 90])))
    buf = np.zeros(SLOTS * WG, dtype=np.uint64)
    if lib.des_dev_debug_stamps(3, buf.ctypes.data, buf.size) == buf.size:
        t = buf.reshape(SLOTS, WG).astype(np.float64)
        live = t[6] > 0
        if live.any():
            t = t[:, live]
            print("  inside the element code, us per tile:")
            for label, k in (("geometry, strain rate, spin (gathers waited for here)", 0), ("LDS reads of stress / strain + next DMA issued", 1),
                             ("rotation, strain update, early stores", 2), ("creep viscosity (pow, exp)", 3), ("maxwell + plastic_props + Mohr-Coulomb pre-filter", 4),
                             ("stores", 5)):
                d = t[k] * 0.01 / t[6]
                print("    %-64s %6.2f (%5.2f / %5.2f / %5.2f)" % (label, d.mean(), *np.percentile(d, [10, 50, 90])))
    dev.close()


if __name__ == "__main__":
    main()
