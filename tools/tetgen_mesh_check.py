#!/usr/bin/env python3
"""Parity of the HIP path with the oracle on the reference's own 1M-tet TetGen mesh
(oracle/_ref/test-3d-big-460.desmesh: test-3d-big.cfg box, resolution 460 m -> 1,001,310 tets /
185,637 nodes, the counts SURVEY.md 8d records).  Dev-time tool: the 21-MB mesh is not committed."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, dynearthsol_amd as des
from oracle_binding import OracleEngine
mesh = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "oracle", "_ref", "test-3d-big-460.desmesh")
for rheol, nsteps, tol in (("elasto-plastic", 20, 0.0), ("elasto-visco-plastic", 20, 1e-10)):
    ov = "mesh.meshing_option = 2\nmesh.meshing_elem_shape = 0\nmat.rheology_type = %s\n" % rheol
    host = des.Host(cfg_text=bench.BENCH_CFG.format(res="460.0", xlen="400e3"), overrides=ov, mesh_file=mesh)
    assert (host.nnode, host.nelem) == (185637, 1001310)
    dev, ora = des.DeviceEngine(host), OracleEngine(host, omp=True)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    sd, so = dev.step(nsteps), ora.step(nsteps)
    worst = 0.0
    for f in ("COORD", "VEL", "STRESS", "STRAIN", "TEMPERATURE", "PLSTRAIN", "VOLUME", "MASS", "FORCE"):
        a, b = dev.download(f), ora.download(f)
        r = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
        worst = max(worst, r)
        assert r <= tol, (rheol, f, r)
    assert (sd.dt, sd.steps) == (so.dt, so.steps) if tol == 0 else abs(sd.dt - so.dt) <= 1e-10 * so.dt
    print("%s: %d steps on %d tets, max rel diff %.2e (bar %.0e), dt %.6e" % (rheol, nsteps, host.nelem, worst, tol, sd.dt), flush=True)
