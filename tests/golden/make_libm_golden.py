#!/usr/bin/env python3
"""Writes tests/golden/libm_bits.json: arguments and result BITS of the portable libm
(dynearthsol_amd/csrc/des_libm.hpp) as its CPU build returns them (g++ -O2 -ffp-contract=off).
The CPU and the gfx950 builds must both reproduce these bits (tests/test_libm.py): that is what
keeps "the same source gives the same bits" true across compilers and compiler versions.

  python tests/golden/make_libm_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle_binding import oracle_libm_eval      # noqa: E402


def main():
    rng = np.random.default_rng(20240607)
    n = 400
    lu = lambda a, b: np.exp(rng.uniform(np.log(a), np.log(b), n))
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-310, 1e308, 709.9, -745.0, 1e-300])
    sx, sy = [a.ravel() for a in np.meshgrid(special, special)]
    cases = {
        "pow": (np.concatenate([lu(1e-25, 1e-8), lu(1e-300, 1e300), rng.uniform(0.99, 1.01, n), sx]),
                np.concatenate([rng.uniform(-1, 0, n), rng.uniform(-1.02, 1.02, n), rng.uniform(-6e4, 6e4, n), sy])),
        "exp": (np.concatenate([rng.uniform(0, 200, n), rng.uniform(-745, 709.7, n), special]), None),
        "sin": (np.concatenate([rng.uniform(-3.2, 3.2, n), rng.uniform(-1e5, 1e5, n), special]), None),
        "cos": (np.concatenate([rng.uniform(-3.2, 3.2, n), rng.uniform(-1e5, 1e5, n), special]), None),
        "tan": (np.concatenate([rng.uniform(0, 1.5533, n), special]), None),
        "atan2": (np.concatenate([rng.uniform(-10, 10, n), lu(1e-30, 1e30), sx]),
                  np.concatenate([rng.uniform(-10, 10, n), lu(1e-30, 1e30) * rng.choice([-1, 1], n), sy])),
    }
    hx = lambda a: ["%016x" % v for v in np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)]
    out = {}
    for fn, (x, y) in cases.items():
        r = oracle_libm_eval(fn, x, y)
        out[fn] = {"x": hx(x), "y": hx(y) if y is not None else None, "result": hx(r)}
    with open(os.path.join(HERE, "libm_bits.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print({k: len(v["x"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
