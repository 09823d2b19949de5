"""The DEVICE build of the 3x3 eigen-solvers and of elasto_plastic, called one tensor at a time
through the C-ABI (des_dev_eigen_eval, des_dev_elasto_plastic_eval):

* tests/golden/eigen_kat.json -- 75 tensors run through the reference's own compiled 3x3-C (the
  two matrices of the reference's tests.cxx:83-96, 132-142 + random / lithostatic-like / degenerate
  ones; made by tests/golden/make_eigen_golden.py) -- put to desk::dsyevc3 / dsyevh3 / dsyevq3:
  dsyevq3 (no libm: Householder + QL, sqrt only) must equal the reference's IEEE build bit for
  bit; dsyevc3 / dsyevh3 call atan2 / cos / sin and are held to the bits of the CPU restatement
  with the same libm (portable) and to Cardano's documented error against the reference vectors
  (rheology.cxx:14-18: 6.6e-4 max|lambda|); and dsyevh3 must take the dsyevq3 branch exactly
  where the reference's build did (3x3-C/dsyevh3.c:152, 177).
* forced shear / tensile / apex / degenerate inputs to one elasto_plastic call (SURVEY.md 7 step
  1-ii: whole runs of the benchmark configs never yield early, so they never get here)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import dynearthsol_amd as des
from oracle_binding import load_oracle, dptr, portable_libm

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "eigen_kat.json")))["cases"]


def six(A):
    A = np.array(A, dtype=np.float64)
    return [A[0, 0], A[1, 1], A[2, 2], A[0, 1], A[0, 2], A[1, 2]]


A6 = np.array([six(c["A"]) for c in KAT])
SCALE = np.array([max(np.abs(np.array(c["A"])).max(), 1e-300) for c in KAT])


def oracle_eig(fn, A):
    lib = load_oracle()
    A = np.ascontiguousarray(np.array(A, dtype=np.float64).ravel())
    Q, w = np.zeros(9), np.zeros(3)
    if fn == "c":
        lib.des_oracle_dsyevc3(dptr(A), dptr(w))
    else:
        getattr(lib, "des_oracle_dsyev%s3" % fn)(dptr(A), dptr(Q), dptr(w))
    return w, Q.reshape(3, 3)


def test_device_dsyevq3_equals_the_reference_ieee_build_bit_for_bit():
    for libm in ("ocml", "portable"):                       # no libm call inside: the policy cannot matter
        w, q, rc = des.eigen_eval("q", A6, libm)
        assert (rc == 0).all()
        for i, c in enumerate(KAT):
            assert np.array_equal(w[i], np.array(c["ieee"]["q_w"])), c["A"]
            assert np.array_equal(q[i], np.array(c["ieee"]["q_q"])), c["A"]


def test_device_dsyevc3_against_the_reference_vectors():
    for libm in ("ocml", "portable"):
        w, _, _ = des.eigen_eval("c", A6, libm)
        for i, c in enumerate(KAT):
            for build in ("ieee", "fastmath"):              # both builds of the reference's source
                err = np.abs(w[i] - np.array(c[build]["c"])).max()
                assert err <= 6.6e-4 * SCALE[i], (libm, build, c["A"], err)
    # with the same libm on both sides: the bits of the CPU restatement (which equals the reference's
    # IEEE build to the bit under glibc: tests/test_oracle_eigen.py)
    w, _, _ = des.eigen_eval("c", A6, "portable")
    with portable_libm():
        for i, c in enumerate(KAT):
            assert np.array_equal(w[i], oracle_eig("c", c["A"])[0]), c["A"]


def test_device_dsyevh3_takes_the_ql_branch_where_the_reference_does():
    ref_fell_back = np.array([c["ieee"]["h_w"] == c["ieee"]["q_w"] and c["ieee"]["h_q"] == c["ieee"]["q_q"] for c in KAT])
    assert ref_fell_back.sum() >= 3 and (~ref_fell_back).sum() >= 30, "the vector set must exercise both branches"
    for libm in ("ocml", "portable"):
        w, q, br = des.eigen_eval("h", A6, libm)
        assert np.array_equal(br.astype(bool), ref_fell_back), (libm, np.flatnonzero(br.astype(bool) != ref_fell_back))
        for i, c in enumerate(KAT):
            if ref_fell_back[i]:                            # the QL result: bit-equal to the reference
                assert np.array_equal(w[i], np.array(c["ieee"]["h_w"])) and np.array_equal(q[i], np.array(c["ieee"]["h_q"])), c["A"]
            else:                                           # Cardano eigenvalues + cross-product vectors
                assert np.abs(w[i] - np.array(c["ieee"]["h_w"])).max() <= 6.6e-4 * SCALE[i]
                A = np.array(c["A"], dtype=np.float64)
                A = np.triu(A) + np.triu(A, 1).T
                # V diag(w) V^T == A as far as Cardano's eigenvalues allow: no worse than the reference's own build
                # (3.8e-3 of max|A| on the near-degenerate lithostatic tensor of the set, 1e-6 and below elsewhere)
                rw, rq = np.array(c["ieee"]["h_w"]), np.array(c["ieee"]["h_q"])
                ref_err = np.abs(rq @ np.diag(rw) @ rq.T - A).max()
                assert np.abs(q[i] @ np.diag(w[i]) @ q[i].T - A).max() <= 2 * ref_err + 1e-9 * SCALE[i], c["A"]
                # same column, same sign as the reference's build
                assert np.abs(q[i] - np.array(c["ieee"]["h_q"])).max() <= 1e-5, c["A"]
    w, q, _ = des.eigen_eval("h", A6, "portable")
    with portable_libm():
        for i, c in enumerate(KAT):
            wo, qo = oracle_eig("h", c["A"])
            assert np.array_equal(w[i], wo) and np.array_equal(q[i], qo), c["A"]


# ---- one elasto_plastic call with forced inputs --------------------------------------------
K, G = 50e9, 30e9


def mc_params(coh, phi, psi, tension_max=1e9):
    sphi, spsi = np.sin(np.radians(phi)), np.sin(np.radians(psi))
    anphi, anpsi = (1 + sphi) / (1 - sphi), (1 + spsi) / (1 - spsi)
    return 2 * coh * np.sqrt(anphi), anphi, anpsi, min(tension_max, coh / np.tan(np.radians(phi)))


def forced_cases():
    rng = np.random.RandomState(5)
    props, de, s, want = [], [], [], []
    def add(s0, d, coh=4.4e7, phi=30.0, psi=0.0, hardn=0.0, mode=None):
        amc, anphi, anpsi, ten_max = mc_params(coh, phi, psi)
        props.append([K, G, amc, anphi, anpsi, hardn, ten_max]); de.append(d); s.append(s0); want.append(mode)
    z = [0.0] * 6
    add([-2e8, -2e8, -2e8, 0, 0, 0], [1e-6, -2e-6, 3e-6, 1e-7, 0, -2e-7], mode=0)            # far below yield
    add([-1.0e9, -3e8, -1e8, 2e7, -1e7, 3e7], z, mode=10)                                       # shear
    add([-1.0e9, -3e8, -1e8, 2e7, -1e7, 3e7], z, hardn=5e9, psi=10.0, mode=10)                  # shear, hardening + dilation
    add([6e7, 2e8, 6e7, 0, 0, 0], z, mode=1)                                                    # tensile
    add([9e7, 9e7, 9e7, 0, 0, 0], z, mode=1)                                                    # isotropic tension: apex, triple eigenvalue
    add([-1.0e9, -1e8, -1e8, 0, 0, 0], z, mode=10)                                              # axisymmetric: double eigenvalue
    add([-3.0e8, -3.0e8, 1.2e8, 0, 0, 0], z, mode=None)                                         # both surfaces violated: h decides
    for _ in range(300):                                                                         # random states around the surface
        p = -10.0 ** rng.uniform(7, 9.3) if _ % 5 else 10.0 ** rng.uniform(7, 8.6)     # every fifth one in tension
        dev = rng.standard_normal(6) * abs(p) * rng.uniform(0, 1.2)
        s0 = [p + dev[0], p + dev[1], p - dev[0] - dev[1], dev[3], dev[4], dev[5]]
        add(s0, list(rng.standard_normal(6) * 1e-5), coh=10.0 ** rng.uniform(6, 8), phi=rng.uniform(5, 40), psi=rng.uniform(0, 10),
            hardn=rng.choice([0.0, 1e9]))
    return np.array(props), np.array(de), np.array(s), want


def oracle_ep(props, de, s):
    lib = load_oracle()
    out, depls, fm = s.copy(), np.zeros(len(s)), np.zeros(len(s), dtype=np.int32)
    for i in range(len(s)):
        m = C.c_int(0)
        si = np.ascontiguousarray(out[i])
        depls[i] = lib.des_oracle_elasto_plastic(*[float(v) for v in props[i]], dptr(np.ascontiguousarray(de[i])), dptr(si), C.byref(m))
        out[i], fm[i] = si, m.value
    return out, depls, fm


def test_device_elasto_plastic_forced_shear_tensile_apex():
    props, de, s, want = forced_cases()
    # the same libm on both sides: every branch, bit for bit
    sd, dd, md = des.elasto_plastic_eval(props, de, s, "portable")
    with portable_libm():
        so, do, mo = oracle_ep(props, de, s)
    assert np.array_equal(md % 100, mo), np.flatnonzero(md % 100 != mo)
    assert np.array_equal(sd, so) and np.array_equal(dd, do)
    for i, w in enumerate(want):
        if w is not None:
            assert md[i] % 100 == w, (i, md[i])
    fm = md % 100
    assert (fm == 0).sum() > 20 and (fm == 1).sum() > 5 and (fm == 10).sum() > 20, np.bincount(fm)
    assert ((md // 100) % 10 == 1).sum() >= 2, "no input reached the dsyevq3 fallback inside the return mapping"
    assert (md[fm != 0] >= 1000).all()                      # every return went through the pre-filter's 'maybe' exit
    # default libm (ocml) against the oracle on the C library: 1-2 ulp in the trigonometric calls only
    sd, dd, md2 = des.elasto_plastic_eval(props, de, s, "ocml")
    so, do, mo = oracle_ep(props, de, s)
    assert np.array_equal(md2 % 100, mo)
    scale = np.abs(so).max(axis=1, keepdims=True)
    assert (np.abs(sd - so) <= 1e-11 * scale).all(), (np.abs(sd - so) / scale).max()
    assert (np.abs(dd - do) <= 1e-11 * np.maximum(np.abs(do), 1e-300) + 1e-25).all()
