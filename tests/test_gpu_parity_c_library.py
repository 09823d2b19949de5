"""Device == the CPU oracle ON THE C LIBRARY'S libm, to the bit, for models that yield.

Until round 3 this held only where pow / exp were the libm calls that reach the state (creep without yielding:
tests/test_gpu_headline.py); sin / cos / tan / atan2 of the device were its own 1-ulp routines, so every model that
yields could only be compared with both sides switched to that portable libm (test_gpu_parity_portable_libm.py) or
within the oracle's own response to a 1-ulp perturbation.  Now the device's sin, tan (plastic_props), atan2 and
sincos (the Kopp solver) restate glibc 2.35's routines (csrc/des_libm_trig.hpp; tests/test_libm.py: 3.4e7 arguments,
no mismatch, CPU and gfx950 builds), so the SAME cases are run here with the oracle left on std::sin / std::tan /
std::atan2 / sincos / std::pow / std::exp of the host -- what the reference's CPU build calls -- and the device on its
default libm.  Needs a host whose glibc runs the FMA variants (x86-64 with FMA + AVX2; the GPU boxes are)."""
import numpy as np
import pytest

import test_gpu_parity_portable_libm as P
from oracle_binding import load_oracle
from test_libm import needs_glibc_fma

pytestmark = [pytest.mark.gpu, needs_glibc_fma]


class c_library_libm:
    """stands in for oracle_binding.portable_libm: the oracle stays on the C library, the device on its default"""

    def __enter__(self):
        import os
        for lib in (load_oracle(False), load_oracle(True)):
            assert lib.des_oracle_set_libm(-1) == 0, "the oracle must be on the C library's libm here"
        assert os.environ.get("DES_LIBM") in (None, "portable")
        return self

    def __exit__(self, *exc):
        return False


CASES = {
    "evp 300 steps": lambda: P.test_creep_rheologies_bit_exact("elasto-visco-plastic", 300),
    "maxwell 300 steps": lambda: P.test_creep_rheologies_bit_exact("maxwell", 300),
    "two materials evp": P.test_two_material_evp_bit_exact,
    "Mohr-Coulomb return on half of the mesh": P.test_mohr_coulomb_return_bit_exact,
    "yield-heavy chaotic run, 300 steps": P.test_yield_heavy_chaotic_run_bit_exact,
    "oblique-rift-3d.cfg 3000 steps": P.test_oblique_rift_3000_steps_bit_exact,
    "conjugate-faults-3d.cfg 1000 steps": P.test_conjugate_faults_1000_steps_bit_exact,
    "test-3d-equ-tiny.cfg 400 steps": P.test_equ_benchmark_bit_exact,
    "three decomposed engines": P.test_decomposed_engines_bit_exact,
}
CASES.update({"random options with creep and yield %d" % s: (lambda s=s: P.test_random_option_combinations_bit_exact_with_creep_and_yield(s))
              for s in range(12)})


@pytest.mark.parametrize("name", list(CASES))
def test_device_equals_the_oracle_on_the_c_library(name, monkeypatch):
    monkeypatch.setattr(P, "portable_libm", c_library_libm)
    CASES[name]()


def test_oblique_rift_10k_steps_bit_identical_to_the_c_library_oracle():
    """BASELINE configs[4] at its own length: examples/oblique-rift-3d.cfg, 10,000 steps, Mohr-Coulomb weak zone
    yielding -- every compared field and dt identical to the oracle on the C library (round 2: 1e-10 for 1000 steps,
    then 'inside the oracle's own 1-ulp response')."""
    import dynearthsol_amd as des
    from oracle_binding import OracleEngine
    from test_oblique_rift import host, FIELDS
    with c_library_libm():
        h = host()
        dev, ora = des.DeviceEngine(h), OracleEngine(h)
        assert dev.init_from_host(h) == ora.init_from_host(h)
        for k in range(10):
            sd, so = dev.step(1000), ora.step(1000)
            assert (sd.dt, sd.time, sd.steps, sd.n_return_mapping) == (so.dt, so.time, so.steps, so.n_return_mapping)
            for f in FIELDS + ("DELTA_PLSTRAIN", "VISCOSITY", "STRAIN_RATE"):
                assert np.array_equal(dev.download(f), ora.download(f)), (f, 1000 * (k + 1))
        assert (ora.download("PLSTRAIN") > 0).sum() > 100 and dev.check_nan() == 0
