"""The portable libm (dynearthsol_amd/csrc/des_libm.hpp): accuracy against mpmath and glibc on
the CPU build, special values, and -- on the GPU -- that the device build returns the same
bits as the CPU build.  It exists so that device-vs-oracle parity of the creep / yield
rheologies can be checked to the bit (tests/test_gpu_parity.py, portable-libm cases).

pow and exp go further: they must return the bits of the C library the CPU reference runs on
(glibc >= 2.28 on an x86-64 host with FMA), because the creep law is the one libm use that
reaches the state of a model that does not yield -- test_pow_exp_return_the_c_librarys_bits
(CPU build) and test_device_pow_exp_return_the_c_librarys_bits (gfx950 build), 2e7 arguments
each, zero mismatches allowed."""
import math
import os
import platform
import subprocess

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
        "sin": (rng.uniform(-3.2, 3.2, n), None, "sin", 0.56),
        "cos": (rng.uniform(-3.2, 3.2, n), None, "cos", 0.56),
        "sin far": (rng.uniform(-1e5, 1e5, n), None, "sin", 0.56),
        "cos far": (rng.uniform(-1e5, 1e5, n), None, "cos", 0.56),
        "tan": (rng.uniform(0, 1.5533, n), None, "tan", 0.65),
        "atan2": (rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), "atan2", 0.6),
        "atan2 wide": (lu(1e-30, 1e30), lu(1e-30, 1e30) * rng.choice([-1, 1], n), "atan2", 0.6),
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
    assert np.isnan(oracle_libm_eval("pow", np.array([-2.0]), np.array([0.5]))).all()   # x < 0, y not an integer
    assert oracle_libm_eval("pow", np.array([-2.0]), np.array([3.0]))[0] == -8.0
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


def _host_runs_the_fma_variant_of_glibc():
    """glibc picks __ieee754_pow_fma / __ieee754_exp_fma when the CPU has FMA and AVX2
    (sysdeps/x86_64/fpu/multiarch/ifunc-fma4.h); des_libm.hpp restates that variant."""
    kind, ver = platform.libc_ver()
    if kind != "glibc" or tuple(int(v) for v in ver.split(".")[:2]) < (2, 28) or platform.machine() != "x86_64":
        return False
    with open("/proc/cpuinfo") as f:
        flags = next((ln for ln in f if ln.startswith("flags")), "").split()
    return "fma" in flags and "avx2" in flags


needs_glibc_fma = pytest.mark.skipif(not _host_runs_the_fma_variant_of_glibc(),
                                     reason="host C library is not glibc >= 2.28 on x86-64 with FMA + AVX2: its pow/exp "
                                            "take another code path (last-bit differences in a few calls per thousand)")


def _c_library_cases(rng, n):
    """Arguments of the creep law (matprops.cxx:359-366: pow(edot, 1/n - 1), exp((E + V p)/(n R T)))
    over and beyond the range any model reaches, plus everything off the main path."""
    lu = lambda a, b: np.exp(rng.uniform(np.log(a), np.log(b), n))
    sgn = lambda: rng.choice([-1.0, 1.0], n)
    return {
        "pow creep law": ("pow", lu(1e-30, 1e-8), 1 / rng.uniform(1, 6, n) - 1),
        "pow wide": ("pow", lu(1e-320, 1e308), rng.uniform(-3, 3, n)),
        "pow near 1": ("pow", rng.uniform(0.99, 1.01, n), rng.uniform(-6e4, 6e4, n)),
        "pow tiny / huge exponents": ("pow", rng.uniform(0.5, 2, n), rng.standard_normal(n) * 10.0 ** rng.integers(-70, 66, n)),
        "pow negative base": ("pow", -lu(1e-5, 1e5), np.where(rng.random(n) < 0.85, np.round(rng.uniform(-20, 20, n)), rng.uniform(-20, 20, n))),
        "exp creep law": ("exp", rng.uniform(0, 400, n), None),
        "exp wide": ("exp", rng.uniform(-760, 720, n), None),
        "exp tiny": ("exp", lu(1e-320, 1) * sgn(), None),
    }


def _c_library_trig_cases(rng, n):
    """Arguments of the trigonometric calls on the path -- plastic_props (matprops.cxx:598-605: sin, tan of angles in
    degrees x pi/180) and the Kopp solver (3x3-C/dsyevc3.c:64-67: atan2(sqrt|.|, q), then cos / sin of a third of it,
    which a compiler turns into one sincos) -- and far beyond: every branch of s_sin.c / s_tan.c / e_atan2.c the
    restatement covers (des_libm_trig.hpp)."""
    lu = lambda a, b: np.exp(rng.uniform(np.log(a), np.log(b), n))
    sgn = lambda: rng.choice([-1.0, 1.0], n)
    cases = {}
    for fn in ("sin", "cos", "sincos_s", "sincos_c"):
        cases[fn + " angles of the path"] = (fn, rng.uniform(0, 1.6, n), None)
        cases[fn + " |x| < 3.2"] = (fn, rng.uniform(-3.2, 3.2, n), None)
        cases[fn + " |x| < 1e5"] = (fn, rng.uniform(-1e5, 1e5, n), None)
        cases[fn + " |x| < 1.05e8"] = (fn, rng.uniform(-1.05e8, 1.05e8, n), None)
        cases[fn + " tiny"] = (fn, lu(1e-320, 1) * sgn(), None)
    cases["tan angles of the path"] = ("tan", rng.uniform(0, 1.5707, n), None)
    cases["tan |x| < 25"] = ("tan", rng.uniform(-25, 25, n), None)
    cases["tan |x| < 1e8"] = ("tan", rng.uniform(-1e8, 1e8, n), None)
    cases["tan tiny"] = ("tan", lu(1e-320, 1) * sgn(), None)
    cases["tan near pi/2"] = ("tan", (np.pi / 2 + rng.integers(-3, 4, n) * np.pi) + lu(1e-12, 1e-2) * sgn(), None)
    q = lu(1e-40, 1e80)
    cases["atan2 Cardano"] = ("atan2", q * lu(1e-9, 1e9), q * sgn())
    cases["atan2 |.| < 10"] = ("atan2", rng.uniform(-10, 10, n), rng.uniform(-10, 10, n))
    cases["atan2 wide"] = ("atan2", lu(1e-300, 1e300) * sgn(), lu(1e-300, 1e300) * sgn())
    cases["atan2 extreme ratios"] = ("atan2", lu(1e-30, 1e30) * sgn(), lu(1e-30, 1e30) * lu(1e-20, 1e20) * sgn())
    cases["atan2 near the axes and the diagonals"] = ("atan2", 1.0 + lu(1e-16, 1e-1) * sgn(), sgn() * (1.0 + lu(1e-16, 1e-1) * sgn()))
    return cases


def _mismatches(a, b):
    return int(((a.view(np.uint64) != b.view(np.uint64)) & ~(np.isnan(a) & np.isnan(b))).sum())


@needs_glibc_fma
def test_pow_exp_return_the_c_librarys_bits():
    """CPU build of deslibm::pow / exp against std::pow / std::exp of this host: 2e7 arguments
    (5e6 of them in the creep law's own range), special values included -- no mismatch."""
    rng = np.random.default_rng(2028)
    total = 0
    for name, (fn, x, y) in _c_library_cases(rng, 2_500_000).items():
        ours, libc = oracle_libm_eval(fn, x, y, omp=True), oracle_libm_eval(fn, x, y, clib=True, omp=True)
        assert _mismatches(ours, libc) == 0, (name, _mismatches(ours, libc))
        total += x.size
    assert total >= 20_000_000
    inf, nan = np.inf, np.nan
    v = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, -2.0, 3.0, -3.0, inf, -inf, nan, 5e-324, -5e-324, 1e-310, 1e308, -1e308,
                  2.2250738585072014e-308, 709.78, -745.13, -708.4, 1024.0, 1e-20])
    X, Y = [a.ravel().copy() for a in np.meshgrid(v, v)]
    assert _mismatches(oracle_libm_eval("pow", X, Y), oracle_libm_eval("pow", X, Y, clib=True)) == 0
    assert _mismatches(oracle_libm_eval("exp", v), oracle_libm_eval("exp", v, clib=True)) == 0
    # results in the subnormal range and at the overflow threshold (e_exp.c specialcase)
    e = np.concatenate([rng.uniform(-745.2, -707.5, 200000), rng.uniform(709.0, 709.8, 200000)])
    assert _mismatches(oracle_libm_eval("exp", e), oracle_libm_eval("exp", e, clib=True)) == 0
    xs, ys = rng.uniform(1e-3, 1e-1, 200000), rng.uniform(100, 160, 200000)
    assert _mismatches(oracle_libm_eval("pow", xs, ys), oracle_libm_eval("pow", xs, ys, clib=True)) == 0


@needs_glibc_fma
def test_trig_return_the_c_librarys_bits():
    """CPU build of deslibm::sin / cos / sincos / tan / atan2 against the host's C library: 3.4e7 arguments, special
    values included -- no mismatch.  (sincos is compared with ::sincos, which in glibc 2.35 has no FMA variant and is
    not bit for bit sin and cos.)"""
    rng = np.random.default_rng(2031)
    total = 0
    for name, (fn, x, y) in _c_library_trig_cases(rng, 1_000_000).items():
        ours, libc = oracle_libm_eval(fn, x, y, omp=True), oracle_libm_eval(fn, x, y, clib=True, omp=True)
        assert _mismatches(ours, libc) == 0, (name, _mismatches(ours, libc))
        total += x.size
    assert total >= 30_000_000
    inf, nan = np.inf, np.nan
    v = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, -2.0, 3.0, inf, -inf, nan, 5e-324, -5e-324, 1e-310, 1e308, -1e308,
                  2.2250738585072014e-308, 0.126, 0.855469, 2.426265, 0.0608, 0.787, 25.0, 1e8, 105414350.0, 0.0625, 1e300, 1e-300])
    for fn in ("sin", "cos", "sincos_s", "sincos_c", "tan"):
        w = v[~(np.abs(v) > 1e8) | ~np.isfinite(v)]          # (beyond its own reduction glibc calls __branred: not restated)
        assert _mismatches(oracle_libm_eval(fn, w), oracle_libm_eval(fn, w, clib=True)) == 0, fn
    Y, X = [a.ravel().copy() for a in np.meshgrid(v, v)]
    assert _mismatches(oracle_libm_eval("atan2", Y, X), oracle_libm_eval("atan2", Y, X, clib=True)) == 0
    # the neighbourhoods of the branch points of the three routines
    for fn, pts in (("sin", (0.126, 0.855469, 2.426265)), ("cos", (0.855469, 2.426265)), ("tan", (1.259e-8, 0.0608, 0.787, 25.0))):
        for p0 in pts:
            x = np.nextafter(p0, 0) + np.arange(-2000, 2000) * np.spacing(p0)
            x = np.concatenate([x, -x])
            assert _mismatches(oracle_libm_eval(fn, x), oracle_libm_eval(fn, x, clib=True)) == 0, (fn, p0)
    u = np.concatenate([0.0625 + np.arange(-2000, 2000) * np.spacing(0.0625), 1.0 + np.arange(-2000, 2000) * np.spacing(0.5)])
    for sx in (1.0, -1.0):
        one = np.full(u.size, sx)
        assert _mismatches(oracle_libm_eval("atan2", u, one), oracle_libm_eval("atan2", u, one, clib=True)) == 0
        assert _mismatches(oracle_libm_eval("atan2", np.abs(one), sx * u), oracle_libm_eval("atan2", np.abs(one), sx * u, clib=True)) == 0


def test_trig_tables_equal_the_c_librarys():
    """des_libm_trig_tables.hpp against the tables of this image's static libm (tools/gen_libm_trig_tables.py reads
    them out of it: they cannot be recomputed, the library's low words are not the correctly rounded remainders)."""
    import glob
    import re
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if not glob.glob("/usr/lib/x86_64-linux-gnu/libm-2.35.a"):
        pytest.skip("no static libm of glibc 2.35 to read the tables from")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_libm_trig_tables", os.path.join(sys_path, "gen_libm_trig_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "dynearthsol_amd", "csrc", "des_libm_trig_tables.hpp")) as f:
        src = f.read()
    for name, member, sym in (("des_sincostab", "sincostab.o", "__sincostab"), ("des_tan_xfg", "s_tan-fma.o", "xfg.12"),
                              ("des_atan_cij", "e_atan2-fma.o", "cij")):
        m = re.search(r"double %s\[\d+\] = \{(.*?)\};" % name, src, re.S)
        ours = np.array([float(t) for t in m.group(1).replace("\n", " ").split(",") if t.strip()])
        assert np.array_equal(ours.view(np.uint64), np.array(gen.table(member, sym)).view(np.uint64)), name


def test_tables_equal_the_c_librarys():
    """The pow / exp tables of des_libm_tables.hpp (recomputed by tools/gen_libm_tables.py from the
    recipe glibc's sources document) and the quoted coefficients against the data objects
    __exp_data / __pow_log_data of this image's static libm."""
    import glob
    import re
    import struct
    import tempfile
    arch = glob.glob("/lib/x86_64-linux-gnu/libm-2.*.a") + glob.glob("/usr/lib/x86_64-linux-gnu/libm-2.*.a")
    if not arch:
        pytest.skip("no static libm to read the C library's tables from")

    def rodata(member):
        with tempfile.TemporaryDirectory() as d:
            subprocess.check_call(["ar", "x", arch[0], member], cwd=d)
            path = os.path.join(d, member)
            for ln in subprocess.check_output(["readelf", "-S", "-W", path]).decode().splitlines():
                m = re.search(r"\]\s+\.rodata\s+PROGBITS\s+\S+\s+(\S+)\s+(\S+)", ln)
                if m:
                    off, size = int(m.group(1), 16), int(m.group(2), 16)
                    with open(path, "rb") as f:
                        return f.read()[off:off + size]
        raise AssertionError("no .rodata in " + member)

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "dynearthsol_amd", "csrc", "des_libm_tables.hpp")) as f:
        src = f.read()

    def arr(name):
        m = re.search(r"double %s\[\d+\] = \{(.*?)\};" % name, src, re.S)
        return np.array([float(t) for t in m.group(1).replace("\n", " ").split(",") if t.strip()])

    def scalar(name):
        return float(re.search(r"%s = ([-+0-9.e]+)" % name, src).group(1))

    e = rodata("e_exp_data.o")           # struct exp_data (sysdeps/ieee754/dbl-64/math_config.h)
    invln2N, shift, nhi, nlo = struct.unpack("<4d", e[:32])
    assert (invln2N, nhi, nlo) == (scalar("des_exp_invln2N"), scalar("des_exp_negln2hiN"), scalar("des_exp_negln2loN"))
    assert shift == 6755399441055744.0
    assert np.array_equal(np.frombuffer(e[32:64]), arr("des_exp_C"))
    tab = np.frombuffer(e[112:112 + 2048], dtype=np.uint64)
    assert np.array_equal(tab[0::2], arr("des_exp_tail").view(np.uint64))
    k = np.arange(128, dtype=np.uint64)
    assert np.array_equal(tab[1::2], arr("des_exp_hi").view(np.uint64) - (k << np.uint64(45)))
    p = rodata("e_pow_log_data.o")       # struct pow_log_data
    assert struct.unpack("<2d", p[:16]) == (scalar("des_ln2hi"), scalar("des_ln2lo"))
    assert np.array_equal(np.frombuffer(p[16:72]), arr("des_log_A"))
    t = np.frombuffer(p[72:72 + 128 * 32]).reshape(128, 4)
    assert np.array_equal(t[:, 0], arr("des_log_invc")) and np.array_equal(t[:, 2], arr("des_log_chi"))
    assert np.array_equal(t[:, 3], arr("des_log_clo"))


@pytest.mark.gpu
@needs_glibc_fma
def test_device_pow_exp_return_the_c_librarys_bits():
    """The gfx950 build against the C library of the GPU box's host: 2e7 arguments, no mismatch."""
    import dynearthsol_amd as des
    rng = np.random.default_rng(2029)
    total = 0
    for name, (fn, x, y) in _c_library_cases(rng, 2_500_000).items():
        dev, libc = des.libm_eval(fn, x, y), oracle_libm_eval(fn, x, y, clib=True, omp=True)
        assert _mismatches(dev, libc) == 0, (name, _mismatches(dev, libc))
        total += x.size
    assert total >= 20_000_000


@pytest.mark.gpu
@needs_glibc_fma
def test_device_trig_return_the_c_librarys_bits():
    """The gfx950 build of sin / cos / sincos / tan / atan2 against the C library of the GPU box's host: 3.4e7
    arguments, no mismatch."""
    import dynearthsol_amd as des
    rng = np.random.default_rng(2032)
    total = 0
    for name, (fn, x, y) in _c_library_trig_cases(rng, 1_000_000).items():
        dev, libc = des.libm_eval(fn, x, y), oracle_libm_eval(fn, x, y, clib=True, omp=True)
        assert _mismatches(dev, libc) == 0, (name, _mismatches(dev, libc))
        total += x.size
    assert total >= 30_000_000


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
