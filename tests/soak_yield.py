#!/usr/bin/env python3
"""Yield-heavy parity at scale (not part of the suite: a minute of 16 CPU threads): the fast-loading
elasto-plastic model of tests/cfgs.py (YIELD) on a 1.6M-tet regular mesh, device against oracle with the
portable libm on both sides (sin / cos / tan / atan2 are in play once elements yield), every field compared
every few steps while the yielding fraction grows to ~10 % (the fused step runs its stress update in one pass, the
Mohr-Coulomb return and the QL fall-back inlined; DES_E2_DEFER=1 pins the two-pass variant).

  python tests/soak_yield.py [--res 800] [--steps 126] [--threads 16]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import cfgs                                              # noqa: E402
import dynearthsol_amd as des                            # noqa: E402
from oracle_binding import OracleEngine, load_oracle, portable_libm   # noqa: E402

FIELDS = ("COORD", "VEL", "STRESS", "STRAIN", "PLSTRAIN", "DELTA_PLSTRAIN", "TEMPERATURE", "VOLUME", "MASS")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=float, default=800.0)
    ap.add_argument("--steps", type=int, default=126)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    os.environ["DES_LIBM"] = "portable"
    load_oracle(omp=True).des_oracle_set_threads(min(a.threads, len(os.sched_getaffinity(0))))
    kw = dict(cfgs.YIELD, lx=160e3, ly=32e3, lz=32e3, res=a.res)
    host = des.Host(cfg_text=cfgs.make(**kw))
    print("# device vs oracle, portable libm on both sides; nnode %d nelem %d" % (host.nnode, host.nelem), flush=True)
    with portable_libm():
        dev, ora = des.DeviceEngine(host), OracleEngine(host, omp=True)
        dev.init_from_host(host)
        ora.init_from_host(host)
        t0, done = time.time(), 0
        for n in (5, 7, 8, 15, 25, 66):
            if done + n > a.steps:
                break
            sd, so = dev.step(n), ora.step(n)
            done += n
            same_clock = (sd.dt, sd.time, sd.steps) == (so.dt, so.time, so.steps)
            worst = 0.0
            for f in FIELDS:
                x, y = dev.download(f), ora.download(f)
                m = np.abs(y).max()
                worst = max(worst, np.abs(x - y).max() / (m if m else 1.0))
            frac = float((dev.download("DELTA_PLSTRAIN") > 0).mean())
            print("step %4d  clock equal %s  yielding %.1f %%  through the return mapping %d = %d  worst rel diff over %d fields %.1e  (%.0f s)"
                  % (done, same_clock, 100 * frac, sd.n_return_mapping, so.n_return_mapping, len(FIELDS), worst, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
