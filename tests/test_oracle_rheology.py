"""Known-answer checks of the oracle's constitutive updates (rheology.cxx:248-484)."""
import ctypes as C

import numpy as np

from oracle_binding import load_oracle, dptr

K, G = 50e9, 30e9


def mc_params(coh=4.4e7, phi=30.0, psi=0.0, tension_max=1e9):
    sphi, spsi = np.sin(np.radians(phi)), np.sin(np.radians(psi))
    anphi = (1 + sphi) / (1 - sphi)
    anpsi = (1 + spsi) / (1 - spsi)
    amc = 2 * coh * np.sqrt(anphi)
    ten_max = min(tension_max, coh / np.tan(np.radians(phi)))
    return amc, anphi, anpsi, ten_max


def ep(s, de, hardn=0.0, **kw):
    lib = load_oracle()
    amc, anphi, anpsi, ten_max = mc_params(**kw)
    s = np.array(s, dtype=np.float64)
    de = np.array(de, dtype=np.float64)
    fm = C.c_int(0)
    depls = lib.des_oracle_elasto_plastic(K, G, amc, anphi, anpsi, hardn, ten_max, dptr(de), dptr(s), C.byref(fm))
    return s, depls, fm.value, (amc, anphi, anpsi, ten_max)


def principal(s):
    A = np.array([[s[0], s[3], s[4]], [s[3], s[1], s[5]], [s[4], s[5], s[2]]])
    return np.linalg.eigvalsh(A)


def test_below_yield_is_pure_hooke():
    s0 = np.array([-2e8, -2e8, -2e8, 0, 0, 0.0])
    de = np.array([1e-6, -2e-6, 3e-6, 1e-7, 0, -2e-7])
    s, depls, fm, _ = ep(s0, de)
    lam = K - 2.0 / 3 * G
    ref = s0.copy()
    ref[:3] += 2 * G * de[:3] + lam * de[:3].sum()
    ref[3:] += 2 * G * de[3:]
    assert depls == 0 and fm == 0
    assert np.array_equal(s, ref)


def test_shear_failure_returns_to_the_yield_surface():
    # strong differential stress under confinement -> shear return (failure_mode 10)
    s0 = np.array([-1.0e9, -3e8, -1e8, 2e7, -1e7, 3e7])
    s, depls, fm, (amc, anphi, anpsi, ten_max) = ep(s0, np.zeros(6))
    assert fm == 10 and depls > 0
    p = principal(s)
    fs = p[0] - p[2] * anphi + amc
    assert abs(fs) <= 1e-9 * abs(p).max()
    # the intermediate principal direction is unchanged by the non-associated flow rule
    assert np.trace(np.diag(s[:3])) < 0


def test_tensile_failure_caps_the_largest_principal_stress():
    # one principal stress above the tension cut-off, the others well below it (the return
    # is single-surface: rheology.cxx:425-435 lowers p[2] to ten_max and the others by alam*a2)
    s0 = np.array([6e7, 2e8, 6e7, 0, 0, 0.0])
    s, depls, fm, (amc, anphi, anpsi, ten_max) = ep(s0, np.zeros(6))
    assert fm == 1 and depls > 0
    p = principal(s)
    assert abs(p[2] - ten_max) <= 1e-9 * ten_max


def test_hardening_reduces_the_plastic_multiplier():
    s0 = np.array([-1.0e9, -3e8, -1e8, 0, 0, 0.0])
    _, d0, _, _ = ep(s0, np.zeros(6), hardn=0.0)
    _, d1, _, _ = ep(s0, np.zeros(6), hardn=5e9)
    assert 0 < d1 < d0


def test_maxwell_relaxes_deviatoric_stress_at_the_analytic_rate():
    # zero strain rate: deviator decays as ((1 - a)/(1 + a))^n with a = dt*G/(2*eta)
    lib = load_oracle()
    eta, dt = 1e21, 1e9
    s = np.array([1e8, -5e7, -5e7, 2e7, 0, 0.0])
    s_init = s.copy()
    de = np.zeros(6)
    n = 50
    for _ in range(n):
        lib.des_oracle_maxwell(K, G, eta, dt, 0.0, dptr(de), dptr(s))
    a = 0.5 * dt * G / eta
    fac = ((1 - a) / (1 + a)) ** n
    mean0 = s_init[:3].mean()
    assert np.allclose(s[:3] - s[:3].mean(), (s_init[:3] - mean0) * fac, rtol=1e-12, atol=1e-3)
    assert np.allclose(s[3:], s_init[3:] * fac, rtol=1e-12, atol=1e-3)
    assert abs(s[:3].mean() - mean0) < 1e-3          # pressure untouched
    # and matches exp(-G t / eta) to first order in a
    assert abs(fac - np.exp(-G * n * dt / eta)) < 1e-4
