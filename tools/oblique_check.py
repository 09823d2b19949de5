#!/usr/bin/env python3
"""examples/oblique-rift-3d.cfg (BASELINE configs[4]) on its TetGen mesh: device vs oracle over 10k steps."""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs, dynearthsol_amd as des
from oracle_binding import OracleEngine
h = des.Host(cfg_text=cfgs.OBLIQUE, mesh_file=os.path.join(ROOT, "tests", "golden", "oblique-rift-3d.desmesh"))
d, o = des.DeviceEngine(h), OracleEngine(h)
assert d.init_from_host(h) == o.init_from_host(h)
for k in range(10):
    sd, so = d.step(1000), o.step(1000)
    out = []
    for f in ("COORD", "VEL", "STRESS", "TEMPERATURE", "PLSTRAIN", "STRAIN"):
        a, b = d.download(f), o.download(f)
        out.append("%s %.1e" % (f, np.abs(a - b).max() / np.abs(b).max()))
    print(so.steps, "dt", sd.dt == so.dt, " ".join(out), "yielding", int((o.download("DELTA_PLSTRAIN") > 0).sum()), flush=True)
