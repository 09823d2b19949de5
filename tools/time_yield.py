#!/usr/bin/env python3
"""Cost of the two-pass stress update when MANY elements yield: the fast-loading elasto-plastic
model of tests/cfgs.py (YIELD) on a 1.6M-tet regular mesh, marched until a large share of the
mesh yields, then E2 timed with and without the deferral (DES_E2_DEFER).  One process per
setting (the switch is read at engine creation).

  python tools/time_yield.py [--res 400] [--march 80] [--steps 40]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import cfgs
    import dynearthsol_amd as des
    kw = dict(cfgs.YIELD, lx=160e3, ly=32e3, lz=32e3, res=a.res, rheol=a.rheology)
    host = des.Host(cfg_text=cfgs.make(**kw))
    dev = des.DeviceEngine(host)
    dev.init_from_host(host)
    out = {"nelem": host.nelem, "defer": os.environ.get("DES_E2_DEFER", "1"), "points": []}
    done = 0
    for march in a.march:
        dev.step(march - done)
        done = march
        frac = float((dev.download("DELTA_PLSTRAIN") > 0).mean())
        dev.profile_enable(True)
        dev.step(a.steps)
        dev.sync()
        prof = {n: ms / max(c, 1) for n, ms, c in dev.profile_read()}
        dev.profile_enable(False)
        done += a.steps
        sc = dev.step(0)
        out["points"].append({"step": done, "yielding_fraction": frac, "n_defer_fraction": sc.n_return_mapping / host.nelem,
                              # (inside a multi-step call the first pass is E2<GEO>, profile name E2G_...; the
                              #  plain E2 only runs in the first step of the call)
                              "E2_us": 1e3 * prof.get("E2G_geom_rotate_update_stress", prof.get("E2_update_stress", 0)),
                              "E2R_us": 1e3 * prof.get("E2_return_mapping", 0),
                              "checksum": float(np.abs(dev.download("STRESS")).sum())})
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=float, default=800.0)
    ap.add_argument("--march", type=int, nargs="+", default=[5, 12, 20, 35, 60, 120])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rheology", default="elasto-plastic")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    res = {}
    for d in ("1", "0"):
        env = dict(os.environ, DES_E2_DEFER=d)
        o = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", "--res", str(a.res), "--steps", str(a.steps),
                                     "--rheology", a.rheology, "--march"] + [str(m) for m in a.march], env=env)
        res[d] = json.loads(o.decode().strip().splitlines()[-1])
    print("nelem", res["1"]["nelem"])
    for p1, p0 in zip(res["1"]["points"], res["0"]["points"]):
        same = p1["checksum"] == p0["checksum"]
        print("step %4d  yielding %5.1f %%  through the return mapping %5.1f %%   two-pass E2 %.1f + %.1f us   one-pass E2 %.1f us   same stress: %s"
              % (p1["step"], 100 * p1["yielding_fraction"], 100 * p1["n_defer_fraction"], p1["E2_us"], p1["E2R_us"], p0["E2_us"], same))


if __name__ == "__main__":
    main()
