"""The portable libm (dynearthsol_amd/csrc/des_libm.hpp): accuracy against mpmath and glibc on
the CPU build, special values, and -- on the GPU -- that the device build returns the same
bits as the CPU build.  It exists so that device-vs-oracle parity of the creep / yield
rheologies can be checked to the bit (tests/test_gpu_parity.py, portable-libm cases)."""
import math

import numpy as np
import pytest

from oracle_binding import oracle_libm_eval

mp = pytest.importorskip("mpmath")


def _ulp_err(got, want_mp):
    want = float(want_mp)
    if math.isinf(want) or want == 0.0:
        return 0.0 if got == want else float("inf")
    e = max(math.frexp(want)[1], -1021)
    return float(abs(mp.mpf(got) - want_mp) / mp.mpf(2) ** (e - 53))


def _samples(rng, n):
    lu = lambda a, b: np.exp(rng.uniform(np.log(a), np.log(b), n))
    return {
        # (x, y, mpmath function, bound in ulp)
        "pow creep law": (lu(1e-25, 1e-8), rng.uniform(-1, 0, n), "pow", 0.52),
        "pow wide": (lu(1e-300, 1e300), rng.uniform(-1.02, 1.02, n), "pow", 0.52),
        "pow near 1": (rng.uniform(0.99, 1.01, n), rng.uniform(-6e4, 6e4, n), "pow", 0.52),
        "pow subnormal": (lu(5e-324, 2.2e-308), rng.uniform(-1, 1, n), "pow", 0.8),
        "exp": (rng.uniform(-700, 700, n), None, "exp", 0.52),
        "exp small": (lu(1e-20, 1) * rng.choice([-1, 1], n), None, "exp", 0.52),
        "sin": (rng.uniform(-3.2, 3.2, n), None, "sin", 0.9),
        "cos": (rng.uniform(-3.2, 3.2, n), None, "cos", 0.9),
        "sin far": (rng.uniform(-1e5, 1e5, n), None, "sin", 0.9),
        "cos far": (rng.uniform(-1e5, 1e5, n), None, "cos", 0.9),
        "tan": (rng.uniform(0, 1.5533, n), None, "tan", 2.5),
        "atan2": (rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), "atan2", 1.7),
        "atan2 wide": (lu(1e-30, 1e30), lu(1e-30, 1e30) * rng.choice([-1, 1], n), "atan2", 1.7),
    }


def test_accuracy_against_mpmath():
    mp.mp.dps = 50
    rng = np.random.default_rng(7)
    fns = {"pow": mp.power, "exp": mp.exp, "sin": mp.sin, "cos": mp.cos, "tan": mp.tan, "atan2": mp.atan2}
    for name, (x, y, fn, bound) in _samples(rng, 1500).items():
        got = oracle_libm_eval(fn, x, y)
        worst = 0.0
        for i in range(x.size):
            want = fns[fn](mp.mpf(x[i]), mp.mpf(y[i])) if y is not None else fns[fn](mp.mpf(x[i]))
            worst = max(worst, _ulp_err(got[i], want))
        assert worst <= bound, (name, worst)


def test_close_to_glibc_on_many_points():
    """2e5 points per case: never more than a few ulp from numpy (C library / its SIMD loops)."""
    rng = np.random.default_rng(11)
    ref = {"pow": np.power, "exp": np.exp, "sin": np.sin, "cos": np.cos, "tan": np.tan, "atan2": np.arctan2}
    for name, (x, y, fn, bound) in _samples(rng, 200000).items():
        got = oracle_libm_eval(fn, x, y)
        with np.errstate(all="ignore"):
            want = ref[fn](x, y) if y is not None else ref[fn](x)
            ulps = np.abs(got - want) / np.maximum(np.spacing(np.abs(want)), 5e-324)
        ulps = np.where(got == want, 0.0, ulps)
        assert np.nanmax(ulps) <= math.ceil(bound) + 1, (name, np.nanmax(ulps))


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
def test_special_values():
    inf, nan = np.inf, np.nan
    v = np.array([0.0, 1.0, 0.5, 2.0, 3.0, inf, nan, 1e-310, 1e308, 1e-300])
    X, Y = [a.ravel() for a in np.meshgrid(v, np.concatenate([v, -v]))]
    with np.errstate(all="ignore"):
        want = np.power(X, Y)
    got = oracle_libm_eval("pow", X, Y)
    same = (got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= np.spacing(np.abs(want)))
    assert same.all(), list(zip(X[~same], Y[~same], got[~same], want[~same]))
    assert np.isnan(oracle_libm_eval("pow", np.array([-2.0]), np.array([0.5]))).all()   # x < 0: NaN by contract
    e = np.array([0.0, -0.0, inf, -inf, nan, 710.0, -746.0, 709.0, -745.0, 1e-320])
    with np.errstate(all="ignore"):
        want = np.exp(e)
    got = oracle_libm_eval("exp", e)
    assert ((got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= np.spacing(np.abs(want)))).all()
    a = np.array([0.0, -0.0, 1.0, -1.0, inf, -inf, 1e-320])
    Y2, X2 = [q.ravel() for q in np.meshgrid(a, a)]
    got, want = oracle_libm_eval("atan2", Y2, X2), np.arctan2(Y2, X2)
    assert (np.abs(got - want) <= 2 * np.spacing(np.abs(want))).all()
    assert (np.signbit(got) == np.signbit(want)).all()
    assert oracle_libm_eval("sin", np.array([0.0]))[0] == 0.0 and oracle_libm_eval("cos", np.array([0.0]))[0] == 1.0
    assert np.isnan(oracle_libm_eval("sin", np.array([inf, nan]))).all()


@pytest.mark.gpu
def test_device_build_gives_the_same_bits():
    import dynearthsol_amd as des
    rng = np.random.default_rng(3)
    n = 400000
    for name, (x, y, fn, _) in _samples(rng, n).items():
        cpu = oracle_libm_eval(fn, x, y)
        gpu = des.libm_eval(fn, x, y)
        assert np.array_equal(cpu.view(np.uint64), gpu.view(np.uint64)), (name, int((cpu.view(np.uint64) != gpu.view(np.uint64)).sum()))
    # special values, over/underflow and subnormal results included
    inf, nan = np.inf, np.nan
    v = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 3.0, inf, -inf, nan, 1e-310, 1e308, 1e-300, 745.0, -745.0, 709.9, 1e6])
    X, Y = [a.ravel() for a in np.meshgrid(v, v)]
    for fn, args in (("pow", (X, Y)), ("atan2", (X, Y)), ("exp", (X, None)), ("sin", (X, None)), ("cos", (X, None)), ("tan", (X, None))):
        cpu, gpu = oracle_libm_eval(fn, *args), des.libm_eval(fn, *args)
        assert np.array_equal(cpu.view(np.uint64), gpu.view(np.uint64)) or \
            (np.isnan(cpu) == np.isnan(gpu)).all() and np.array_equal(cpu[~np.isnan(cpu)].view(np.uint64), gpu[~np.isnan(gpu)].view(np.uint64)), fn


def _golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "libm_bits.json")) as f:
        g = json.load(f)
    un = lambda h: np.array([int(v, 16) for v in h], dtype=np.uint64).view(np.float64)
    return {fn: (un(c["x"]), un(c["y"]) if c["y"] is not None else None, np.array([int(v, 16) for v in c["result"]], dtype=np.uint64))
            for fn, c in g.items()}


def _same_bits(got, want_bits):
    g = got.view(np.uint64)
    nan = np.isnan(got) & np.isnan(want_bits.view(np.float64))       # any NaN payload / sign
    return bool(((g == want_bits) | nan).all())


def test_cpu_build_reproduces_the_committed_bits():
    """tests/golden/libm_bits.json (make_libm_golden.py): pins the functions across compilers."""
    for fn, (x, y, want) in _golden().items():
        assert _same_bits(oracle_libm_eval(fn, x, y), want), fn


@pytest.mark.gpu
def test_device_build_reproduces_the_committed_bits():
    import dynearthsol_amd as des
    for fn, (x, y, want) in _golden().items():
        assert _same_bits(des.libm_eval(fn, x, y), want), fn
