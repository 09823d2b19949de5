#!/usr/bin/env python3
"""Generate tests/golden/eigen_kat.json from the reference's own 3x3-C solver.

The solver is built by `make -C oracle ref` from /root/reference/3x3-C/*.c where they lie
(both with the reference's flags, -O3 -ffast-math, and as plain IEEE).  Inputs: the matrices
of the reference's tests.cxx (83-96, 132-142) plus seeded random symmetric tensors
(indefinite, near-degenerate, degenerate, stress-like).  Outputs: eigenvalues from dsyevc3
and eigenvalues/eigenvectors from dsyevh3 and dsyevq3, as produced by the reference code.
Dev-time tool: needs /root/reference; the fixture it writes travels, the reference does not.
"""
import ctypes as C
import json
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load(name):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", name))
    A33 = C.POINTER(C.c_double * 3)
    lib._Z7dsyevc3PA3_dPd.argtypes = [A33, C.POINTER(C.c_double)]
    lib._Z7dsyevh3PA3_dS0_Pd.argtypes = [A33, A33, C.POINTER(C.c_double)]
    lib._Z7dsyevq3PA3_dS0_Pd.argtypes = [A33, A33, C.POINTER(C.c_double)]
    return lib


def call(lib, fn, a):
    A = ((C.c_double * 3) * 3)(*[(C.c_double * 3)(*row) for row in a])
    Q = ((C.c_double * 3) * 3)()
    w = (C.c_double * 3)()
    if fn == "c":
        lib._Z7dsyevc3PA3_dPd(A, w)
        return list(w), None
    f = lib._Z7dsyevh3PA3_dS0_Pd if fn == "h" else lib._Z7dsyevq3PA3_dS0_Pd
    f(A, Q, w)
    return list(w), [list(r) for r in Q]


def cases():
    out = []
    # tests.cxx:83-96: diag(3e4, -1, 3) with off-diagonals (2, 4, 2)
    out.append([[3e4, 2, 4], [2, -1, 2], [4, 2, 3]])
    # tests.cxx:132-142: s = {3e4, -1e-5, 3, 2, 4, 8} -> XX YY ZZ XY XZ YZ
    out.append([[3e4, 2, 4], [2, -1e-5, 8], [4, 8, 3]])
    rng = np.random.RandomState(12345)
    for k in range(40):
        m = rng.standard_normal((3, 3)) * 10.0 ** rng.randint(-3, 9)
        out.append(((m + m.T) / 2).tolist())
    for k in range(20):       # lithostatic-like: large negative mean, small deviator
        p = -10.0 ** rng.uniform(6, 9)
        d = rng.standard_normal((3, 3)) * abs(p) * 10.0 ** rng.uniform(-8, -1)
        out.append((np.eye(3) * p + (d + d.T) / 2).tolist())
    for k in range(10):       # two equal eigenvalues (QL fallback territory)
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        lam = np.diag([1.0 + k, 1.0 + k, -2.0 * k - 1])
        m = q @ lam @ q.T
        out.append(((m + m.T) / 2).tolist())
    out.append(np.diag([5.0, 5.0, 5.0]).tolist())
    out.append(np.zeros((3, 3)).tolist())
    out.append(np.diag([-3.0, 2.0, 7.0]).tolist())
    return out


def main():
    fast, ieee = load("libkopp3x3.so"), load("libkopp3x3_ieee.so")
    rec = []
    for a in cases():
        r = {"A": a}
        for tag, lib in (("fastmath", fast), ("ieee", ieee)):
            r[tag] = {"c": call(lib, "c", a)[0]}
            w, q = call(lib, "h", a); r[tag]["h_w"], r[tag]["h_q"] = w, q
            w, q = call(lib, "q", a); r[tag]["q_w"], r[tag]["q_q"] = w, q
        rec.append(r)
    json.dump({"source": "3x3-C/dsyev{c,h,q}3.c compiled from /root/reference (oracle/Makefile: ref)",
               "cases": rec}, open(os.path.join(HERE, "eigen_kat.json"), "w"), indent=0)
    print("wrote %d cases" % len(rec))


if __name__ == "__main__":
    main()
