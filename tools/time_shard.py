#!/usr/bin/env python3
"""Step time of ONE rank's shard of the headline mesh cut N ways (the middle rank of host/partition.cpp's slabs: owned
nodes + four-layer ghost region), as a stand-alone engine on the one GPU of this box -- no exchange, every local node
treated as owned (the physics at the cut faces is then wrong, the work per step is the rank's).  What a rank of the
strong-scaling run computes per step, for DESIGN.md section 6's projection.

    python tools/time_shard.py [steps] [--ranks 4,8] [--kernels]        (on the MI355X box)

--kernels adds the HIP-event time of each pass (engine knobs such as DES_PATCH=<nodes per block> are read from the
environment when the engine is created, so one run per setting compares them).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
import dynearthsol_amd as des                          # noqa: E402
from dynearthsol_amd.decomp import Partition           # noqa: E402


def main():
    argv = sys.argv[1:]
    kernels = "--kernels" in argv
    if kernels: argv.remove("--kernels")
    ranks = (1, 2, 4, 8)
    if "--ranks" in argv:
        i = argv.index("--ranks")
        ranks = tuple(int(x) for x in argv[i + 1].split(","))
        del argv[i:i + 2]
    steps = int(argv[0]) if argv else 400
    mesh = des.reference_mesh("test-3d-big-460")
    ov = "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\n"
    host = des.Host(cfg_text=bench.BENCH_CFG.format(res="460.0", xlen="400e3"), overrides=ov, mesh_file=mesh)
    print("# headline mesh %d tets / %d nodes; %d-step calls after 40 warm-up steps" % (host.nelem, host.nnode, steps))
    print("%6s %6s %10s %10s %12s %12s" % ("ranks", "rank", "nelem", "nnode", "us per step", "x vs 1 rank"))
    base = None
    for nranks in ranks:
        rank = nranks // 2
        part = Partition(host, nranks, rank) if nranks > 1 else None
        src = part if part is not None else host
        dev = des.DeviceEngine(src)
        if part is None:
            dev.init_from_host(host)
        else:
            coord = part.local("coord")
            dev.upload("COORD", coord); dev.upload("COORD0", coord)
            dev.upload("ELEMMARKERS", part.local("elemmarkers")); dev.upload("VEL", part.local("vel"))
            dev.init_geometry()
            for f, name in (("TEMPERATURE", "temperature"), ("RADIOGENIC", "radiogenic"), ("STRESS", "stress"),
                            ("STRAIN", "strain"), ("PLSTRAIN", "plstrain"), ("VISCOSITY", "viscosity")):
                dev.upload(f, part.local(name))
            dev.compute_dt()
        dev.step(40, want_scalars=False)
        dev.sync()
        best = 1e9
        for _ in range(3):
            dev.timer_start()
            dev.step(steps, want_scalars=False)
            best = min(best, dev.timer_stop() / steps)
        us = 1e3 * best
        base = base or us
        line = "%6d %6d %10d %10d %12.1f %12.2f" % (nranks, rank, src.nelem, src.nnode, us, base / us)
        if kernels:
            dev.profile_enable(True)
            dev.step(40, want_scalars=False)
            k = {n: 1e3 * t / c for n, t, c in dev.profile_read() if c}
            line += "   " + "  ".join("%s %.1f" % (n.split("_")[0], k[n]) for n in sorted(k) if n[:2] in ("EN", "E2"))
        print(line)
        dev.close()


if __name__ == "__main__":
    main()
