"""Device == oracle TO THE BIT for every rheology once both sides use the same libm.

The default-mode tests (test_gpu_parity.py, test_oblique_rift.py, test_equ_benchmarks.py) have to
allow 1e-10 -- or, on models that yield, the oracle's own response to a 1-ulp change -- wherever
pow/exp/sin/tan/atan2/cos enter, because glibc and ROCm's ocml round those differently.  Here the
engine runs with DES_LIBM=portable and the oracle with des_oracle_set_libm(1): both call
dynearthsol_amd/csrc/des_libm.hpp (pure IEEE arithmetic; tests/test_libm.py checks the device and
CPU builds return the same bits and that it is a <= 1-ulp libm).  Everything else on the path
was already bit-identical, so now ALL of it is: creep viscosity, Mohr-Coulomb return, the
Cardano / QL eigen-solvers, chaotic yield-heavy runs, 1000+ steps, 1M elements.  That turns
"the differences are libm rounding" from an argument into a measurement."""
import os

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd import decomp
from oracle_binding import OracleEngine, load_oracle, portable_libm
from test_gpu_parity import STATE, _random_overrides, transplant
from test_oblique_rift import host as oblique_host, conjugate_host
from test_gpu_decomp import _NoReduce

pytestmark = pytest.mark.gpu


def bit_exact(dev, ora, fields=STATE, note=""):
    for f in fields:
        a, b = dev.download(f), ora.download(f)
        assert np.array_equal(a, b), "%s differs (%d entries) %s" % (f, int((a != b).sum()), note)


def pair(host, omp=False):
    if omp:      # libgomp defaults to every core of the machine; the box gives this job 16
        load_oracle(omp=True).des_oracle_set_threads(min(16, os.cpu_count() or 1))
    dev, ora = des.DeviceEngine(host), OracleEngine(host, omp=omp)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    return dev, ora


def test_the_switch_reaches_both_sides(monkeypatch):
    """DES_LIBM / des_oracle_set_libm select the libm; anything else is refused.  (The portable
    pow/exp return glibc's bits, sin/cos/tan/atan2 are 1-2 ulp from it, so the two ORACLE modes stay
    within rounding of each other on this yielding model; the device's ocml differs in every function.)"""
    monkeypatch.setenv("DES_LIBM", "ocml")                # d0 / o0: ROCm's libm against the C library's
    host = des.Host(cfg_text=cfgs.make(**cfgs.YIELD))
    d0, o0 = pair(host)
    d0.step(40); o0.step(40)              # the oracle's switch is process-wide and acts at step time
    with portable_libm():
        d1, o1 = pair(host)
        d1.step(40); o1.step(40)
    bit_exact(d1, o1)
    assert not np.array_equal(d0.download("STRESS"), d1.download("STRESS"))       # ocml vs portable
    for f in ("STRESS", "VEL", "COORD"):
        ref = o0.download(f)
        assert np.abs(o1.download(f) - ref).max() <= 1e-4 * np.abs(ref).max(), f   # same physics
    os.environ["DES_LIBM"] = "fast"
    try:
        with pytest.raises(des.DesError):
            des.DeviceEngine(host)
    finally:
        del os.environ["DES_LIBM"]


@pytest.mark.parametrize("rheol,nsteps", [("elasto-visco-plastic", 300), ("maxwell", 300), ("viscous", 1)])
def test_creep_rheologies_bit_exact(rheol, nsteps):
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, rheol=rheol)))
        dev, ora = pair(host)
        done = 0
        while done < nsteps:
            n = min(100, nsteps - done)
            sd, so = dev.step(n), ora.step(n)
            done += n
            assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel)
            bit_exact(dev, ora, note="after %d steps" % done)


def test_two_material_evp_bit_exact():
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, nmat=2)))
        dev, ora = pair(host)
        dev.step(120); ora.step(120)
        bit_exact(dev, ora)


def test_mohr_coulomb_return_bit_exact():
    """~50 % of the elements through dsyevh3 + the return mapping in one step (the set-up of
    test_gpu_parity.py::test_mohr_coulomb_return_single_step_from_identical_state)."""
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**cfgs.YIELD))
        march = OracleEngine(host)
        march.init_from_host(host)
        sc = march.step(60)
        dev, ora = des.DeviceEngine(host), OracleEngine(host)
        last = []
        for eng in (dev, ora):
            eng.init_from_host(host)
            transplant(march, eng)
            eng.init_geometry()
            eng.set_clock(sc.dt, sc.time, 0)
            last.append(eng.step(1))
        assert (ora.download("DELTA_PLSTRAIN") > 0).sum() > host.nelem // 4
        bit_exact(dev, ora)
        # the same elements went through dsyevh3 + the return mapping (second pass on the device)
        assert last[0].n_return_mapping == last[1].n_return_mapping >= (ora.download("DELTA_PLSTRAIN") > 0).sum()


@pytest.mark.parametrize("mode", ["0", "1"])
def test_one_pass_and_two_pass_stress_update_give_the_same_bits(mode, monkeypatch):
    """DES_E2_DEFER pins the stress update to one pass (return mapping inline) or two (yield
    candidates set aside for E2_return_mapping); by default the engine picks per call."""
    monkeypatch.setenv("DES_E2_DEFER", mode)
    with portable_libm():
        for kw in (cfgs.YIELD, dict(cfgs.YIELD, rheol="elasto-visco-plastic", tmantle=1573, alpha=3e-5, vmin="1e19")):
            host = des.Host(cfg_text=cfgs.make(**kw))
            dev, ora = pair(host)
            for k in range(4):
                sd, so = dev.step(25), ora.step(25)
                assert sd.n_return_mapping == so.n_return_mapping
                bit_exact(dev, ora, note="mode %s after %d steps" % (mode, 25 * (k + 1)))
            assert so.n_return_mapping > 0


@pytest.mark.parametrize("mode", ["0", "1"])
def test_averaged_fields_ride_in_the_stress_update_while_elements_yield(mode, monkeypatch):
    """Output::average_fields folded into E2<GEO> (the end-of-step stress / strain of the step before, this step's
    delta_plstrain): with half of the mesh yielding, in one pass and in two (the elements set aside add their
    delta_plstrain in the return-mapping pass, everything else exactly once in the first), across interval
    boundaries and call boundaries, the running sums and snapshots equal the oracle's to the bit."""
    monkeypatch.setenv("DES_E2_DEFER", mode)
    ov = "sim.is_outputting_averaged_fields = yes\nsim.output_step_interval = 100\n"
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**dict(cfgs.YIELD, qcsi=10)), overrides=ov)
        dev, ora = pair(host)
        for n in (1, 4, 5, 7, 10, 3, 30):
            sd, so = dev.step(n), ora.step(n)
            assert (sd.dt, sd.time, sd.steps, sd.avg_time0) == (so.dt, so.time, so.steps, so.avg_time0)
            for f in ("STRESS_AVG", "DPLSTRAIN_AVG", "STRAIN0", "COORD_AVG0", "STRESS", "STRAIN", "PLSTRAIN", "DELTA_PLSTRAIN"):
                assert np.array_equal(dev.download(f), ora.download(f)), "%s, mode %s, after %d steps" % (f, mode, sd.steps)
        assert so.n_return_mapping > host.nelem // 10 and np.abs(ora.download("DPLSTRAIN_AVG")).max() > 0


def test_yield_heavy_chaotic_run_bit_exact():
    """Half of the mesh yielding every step: in default mode only statistics can be compared
    (test_yield_heavy_run_stays_statistically_identical); with one libm the 300-step trajectories
    are the same bits, yield decisions and dsyevq3 fall-backs included."""
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**cfgs.YIELD))
        dev, ora = pair(host)
        for k in range(3):
            dev.step(100); ora.step(100)
            assert (ora.download("DELTA_PLSTRAIN") > 0).sum() > host.nelem // 4
            bit_exact(dev, ora, note="after %d steps" % (100 * (k + 1)))
        assert dev.check_nan() == 0


@pytest.mark.parametrize("seed", range(12))
def test_random_option_combinations_bit_exact_with_creep_and_yield(seed):
    rng = np.random.default_rng(5000 + seed)
    ov = _random_overrides(rng)
    base = [cfgs.EVP, cfgs.YIELD, dict(cfgs.EVP, rheol="maxwell")][seed % 3]
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**dict(base, nmat=1 + seed % 2)), overrides=ov)
        dev, ora = pair(host)
        dev.step(25); ora.step(25)
        if ora.check_nan():
            pytest.skip("this combination blows up on the CPU too")
        bit_exact(dev, ora, note="with\n" + ov)


def test_oblique_rift_3000_steps_bit_exact():
    """examples/oblique-rift-3d.cfg on the reference's own mesh: two materials, PREM reference
    pressure, yielding weak zone (default mode: 1e-10 for 1000 steps, then 'inside the noise')."""
    with portable_libm():
        h = oblique_host()
        dev, ora = pair(h)
        for k in range(3):
            sd, so = dev.step(1000), ora.step(1000)
            assert (sd.dt, sd.time, sd.steps) == (so.dt, so.time, so.steps)
            bit_exact(dev, ora, note="after %d steps" % (1000 * (k + 1)))
        assert (ora.download("PLSTRAIN") > 0).sum() > 50


def test_conjugate_faults_1000_steps_bit_exact():
    with portable_libm():
        h = conjugate_host()
        dev, ora = pair(h, omp=True)
        for k in range(2):
            dev.step(500); ora.step(500)
            bit_exact(dev, ora, note="after %d steps" % (500 * (k + 1)))


def test_equ_benchmark_bit_exact():
    """benchmarks-cores/test-3d-equ-tiny.cfg (7 materials, evp, Winkler + surface processes):
    the file's 400 steps."""
    with portable_libm():
        h = des.Host(cfg_text=cfgs.make_equ())
        dev, ora = pair(h)
        for k in range(4):
            dev.step(100); ora.step(100)
            bit_exact(dev, ora, STATE + ("MASS",), note="after %d steps" % (100 * (k + 1)))


def test_decomposed_engines_bit_exact():
    """3 device engines exchanging ghost state == ONE oracle, yield-heavy, to the bit."""
    with portable_libm():
        host = des.Host(cfg_text=cfgs.make(**cfgs.YIELD))
        ora = OracleEngine(host)
        ora.init_from_host(host)
        parts = [decomp.Partition(host, 3, r) for r in range(3)]
        engs = [des.DeviceEngine(p) for p in parts]
        steppers = [decomp.PhasedStepper(e, p, None) for e, p in zip(engs, parts)]
        comm = decomp.LoopbackComm(steppers)
        for e, p in zip(engs, parts):
            decomp.init_rank(e, p, _NoReduce())
        comm.reduce_dt_all(recompute=True)
        decomp.run_loopback(steppers, 60)
        ora.step(60)
        for f, nc, kind, n in (("COORD", 3, "node", host.nnode), ("VEL", 3, "node", host.nnode),
                               ("STRESS", 6, "elem", host.nelem), ("PLSTRAIN", 1, "elem", host.nelem),
                               ("VISCOSITY", 1, "elem", host.nelem)):
            got = decomp.assemble(parts, [e.download(f) for e in engs], nc, n, kind)
            assert np.array_equal(got, ora.download(f)), f


def test_full_size_mesh_evp_bit_exact():
    """The bench workload (1,097,600 tets, evp): 60 steps against the OpenMP oracle."""
    import bench
    with portable_libm():
        host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(400e3 / 560), xlen=repr(400e3)))
        dev, ora = pair(host, omp=True)
        for k in range(2):
            dev.step(30); ora.step(30)
            bit_exact(dev, ora, ("COORD", "VEL", "STRESS", "STRAIN", "PLSTRAIN", "TEMPERATURE", "VISCOSITY", "MASS", "FORCE"),
                      note="after %d steps" % (30 * (k + 1)))
