"""Whole-step behaviour of the CPU oracle (the checker the GPU path is compared with)."""
import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from oracle_binding import OracleEngine

FIELDS = ("COORD", "VEL", "FORCE", "TEMPERATURE", "STRESS", "STRAIN", "STRAIN_RATE", "PLSTRAIN",
          "VISCOSITY", "VOLUME", "VOLUME_N", "MASS", "TMASS")


def run(kw, nsteps, omp=False, overrides=None):
    h = des.Host(cfg_text=cfgs.make(**kw), overrides=overrides)
    o = OracleEngine(h, omp=omp)
    dt = o.init_from_host(h)
    sc = o.step(nsteps)
    return h, o, dt, sc


def test_initial_dt_is_the_elastic_limit():
    # regular 2 km grid: the smallest tet height is 2000/sqrt(3) (corner tets); dt_elastic =
    # 0.5*minl/(max_vbc*inertial_scaling) (geometry.cxx:1623-1626)
    h, o, dt, _ = run(cfgs.EP, 0)
    assert dt == pytest.approx(0.5 * (2e3 / np.sqrt(3)) / (1e-9 * 1e4), rel=1e-12)


def test_threaded_and_serial_oracle_agree_to_the_bit():
    # the reference's own race check: 1 thread vs N threads (omp-gcc-matrix.yml:61-105)
    for kw in (cfgs.EP, cfgs.EVP, cfgs.YIELD):
        _, a, _, sa = run(kw, 30)
        _, b, _, sb = run(kw, 30, omp=True)
        for f in FIELDS:
            assert np.array_equal(a.download(f), b.download(f)), f
        assert sa.dt == sb.dt


def test_step_splitting_is_invisible():
    _, a, _, _ = run(cfgs.EVP, 24)
    h, b, _, _ = run(cfgs.EVP, 7)
    b.step(10); b.step(7)
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f)), f


def test_lithostatic_column_stays_in_equilibrium():
    # no boundary motion, no weak zone: the Winkler bottom balances the overburden and
    # nothing moves faster than round-off of the 2.6e8 Pa stresses allows
    kw = dict(cfgs.EP, vx=0.0, control="characteristic_speed = 1e-9\n")
    h, o, dt, sc = run(kw, 50, overrides="ic.weakzone_option = 0\n")
    v = o.download("VEL")
    assert np.abs(v).max() < 1e-15
    s = o.download("STRESS").reshape(6, -1)
    s0 = h.array("stress").reshape(6, -1)
    assert np.abs(s - s0).max() < 1e-3


def test_extension_unloads_horizontal_stress_elastically():
    h, o, dt, sc = run(cfgs.EP, 60)
    v = o.download("VEL").reshape(3, -1)
    x = o.download("COORD").reshape(3, -1)
    assert np.all(v[0][x[0] < 1] == -1e-9) and np.all(v[0][x[0] > 40e3 - 1] == 1e-9)   # vbc honoured
    assert np.all(v[1][(x[1] == 0) | (x[1] == 8e3)] == 0)
    sxx = o.download("STRESS").reshape(6, -1)[0]
    sxx0 = h.array("stress").reshape(6, -1)[0]
    assert (sxx - sxx0).mean() > 0                             # less compressive after stretching
    assert o.check_nan() == 0 and sc.status == 0 and sc.steps == 60


def test_yielding_run_accumulates_plastic_strain():
    h, o, dt, sc = run(cfgs.YIELD, 100)
    pls = o.download("PLSTRAIN")
    assert (o.download("DELTA_PLSTRAIN") > 0).sum() > 100
    assert pls.max() > 0.5 and np.all(pls >= 0)


def test_creep_limits_viscosity_between_bounds():
    h, o, dt, sc = run(cfgs.EVP, 20)
    visc = o.download("VISCOSITY")
    assert visc.min() == 1e19 and visc.max() == 1e24 and np.unique(visc).size > 100


def test_average_fields_follow_the_reference_schedule():
    """Output::average_fields (output.cxx:327-370): at steps % interval == 1 the snapshots are
    taken and the sums restart, otherwise stress / delta_plstrain are accumulated."""
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EP, qcsi=5)),
                    overrides="sim.is_outputting_averaged_fields = yes\n")
    ora = OracleEngine(host)
    ora.init_from_host(host)
    acc = None
    for step in range(1, 13):
        sc = ora.step(1)
        s = ora.download("STRESS")
        if step % 5 == 1:
            acc, t0, c0, e0 = s.copy(), sc.time, ora.download("COORD"), ora.download("STRAIN")
        else:
            acc += s
        assert np.array_equal(ora.download("STRESS_AVG"), acc)
        assert sc.avg_time0 == t0
        assert np.array_equal(ora.download("COORD_AVG0"), c0) and np.array_equal(ora.download("STRAIN0"), e0)


def test_pseudo_transient_loop_of_the_restatement():
    """control.has_PT (dynearthsol.cxx:803-864) in the oracle: with no iteration allowed the step is
    the plain one; a huge tolerance stops after exactly one iteration per step; a tight one iterates
    until the residual's relative change is below it (never past PT_max_iter); the clock and the
    temperature do not move inside the loop."""
    import cfgs
    import dynearthsol_amd as des
    from oracle_binding import OracleEngine
    base = "control.has_PT = yes\ncontrol.PT_max_iter = %d\ncontrol.PT_relative_tolerance = %s\n"

    def run(ov, n=6):
        host = des.Host(cfg_text=cfgs.make(**cfgs.EVP), overrides=ov)
        o = OracleEngine(host)
        o.init_from_host(host)
        sc = o.step(n)
        return sc, {f: o.download(f) for f in ("COORD", "VEL", "STRESS", "TEMPERATURE")}

    plain, fp = run(None)
    none, f0 = run(base % (0, "1e-6"))
    assert none.n_pt_iterations == 0 and (none.time, none.steps) == (plain.time, plain.steps)
    assert all(np.array_equal(fp[k], f0[k]) for k in fp)
    one, f1 = run(base % (50, "1e30"))
    assert one.n_pt_iterations == 6 and (one.time, one.steps) == (plain.time, plain.steps)
    assert not np.array_equal(f1["STRESS"], fp["STRESS"])
    # one step: update_temperature precedes the loop and is not repeated inside it
    (p1, g1), (q1, h1) = run(None, 1), run(base % (50, "1e30"), 1)
    assert q1.n_pt_iterations == 1 and np.array_equal(g1["TEMPERATURE"], h1["TEMPERATURE"])
    assert not np.array_equal(g1["COORD"], h1["COORD"])
    many, _ = run(base % (7, "1e-12"))
    assert many.n_pt_iterations == 6 * 7                   # the cap
    some, _ = run(base % (500, "1e-3"))
    assert 6 <= some.n_pt_iterations < 6 * 500
