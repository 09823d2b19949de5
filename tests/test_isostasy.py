"""isostasy_adjustment (dynearthsol.cxx:496-544): the loop main() runs before the first frame when
ic.isostasy_adjustment_time_in_yr > 0 -- the time step without clock, temperature update, NMD,
velocity bcs, rotate_stress and compute_dt, with vertical motion only.  It is a caller of the
offloaded path, so it runs on the engine: set_isostasy(1); step(n); set_isostasy(0)."""
import os

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from dynearthsol_amd import decomp, driver
from oracle_binding import OracleEngine, portable_libm
from test_driver_output import oracle_api, read_frame, as_f64, in_tmp, YEAR2SEC      # noqa: F401

STATE = ("COORD", "VEL", "FORCE", "TEMPERATURE", "STRESS", "STRAIN", "STRAIN_RATE", "PLSTRAIN", "VISCOSITY",
         "VOLUME", "VOLUME_OLD", "VOLUME_N", "MASS", "TMASS", "DPRESSURE", "DHACC", "DH", "EDVACC_SURF")
NO_WINKLER = "bc.has_winkler_foundation = no\nbc.vbc_z0 = 1\n"


def same(a, b, fields=STATE):
    for f in fields:
        x, y = a.download(f), b.download(f)
        assert np.array_equal(x, y), "%s: %d of %d entries differ" % (f, int((x != y).sum()), x.size)


def iso_run(eng, host, n):
    dt = eng.init_from_host(host)
    eng.set_isostasy(True)
    sc = eng.step(n)
    eng.set_isostasy(False)
    return dt, sc


@pytest.mark.parametrize("overrides", [None, NO_WINKLER])
def test_what_the_loop_does_and_does_not_do(overrides):
    host = des.Host(cfg_text=cfgs.make(**cfgs.EVP), overrides=overrides)
    o = OracleEngine(host)
    dt, sc = iso_run(o, host, 30)
    assert (sc.time, sc.steps, sc.dt) == (0.0, 0, dt)                          # the clock does not move
    c0, c = host.array("coord").reshape(3, -1), o.download("COORD").reshape(3, -1)
    assert np.array_equal(c[:2], c0[:2])                                       # vertical motion only
    assert np.abs(c[2] - c0[2]).max() > 0
    v = o.download("VEL").reshape(3, -1)
    assert not v[:2].any() and np.abs(v[2]).max() > 0
    assert np.array_equal(o.download("TEMPERATURE"), host.array("temperature"))  # no update_temperature
    zmin = c0[2].min()
    bottom = c0[2] == zmin
    if overrides:                                                              # bottom held without Winkler
        assert np.array_equal(c[2][bottom], c0[2][bottom])
    else:
        assert np.abs(c[2][bottom] - zmin).max() > 0
    # differs from ordinary time steps
    t = OracleEngine(host)
    t.init_from_host(host)
    t.step(30)
    assert not np.array_equal(t.download("STRESS"), o.download("STRESS"))
    # and the engine goes on with time steps afterwards
    sc = o.step(10)
    assert sc.steps == 10 and sc.time > 0 and o.check_nan() == 0


def test_driver_runs_it_before_the_first_frame(in_tmp, capfd):
    years = 50.0
    ov = ("sim.modelname = iso\nsim.max_steps = 20\nsim.output_step_interval = 20\n"
          "ic.isostasy_adjustment_time_in_yr = %r\n" % years)
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides=ov)
    st = driver.run(host, api=oracle_api(), quiet=False)
    out = capfd.readouterr().out
    assert (st.steps, st.frames, st.exit_code) == (20, 2, 0)
    # the same by hand: dt, steps = years*YEAR2SEC/dt, loop, compute_dt again (dynearthsol.cxx:503-504, 643)
    o = OracleEngine(host)
    dt0 = o.init_from_host(host)
    n = int(years * YEAR2SEC / dt0)
    assert n >= 5
    assert "Adjusting isostasy for 50 yrs..." in out and "Adjusted isostasy for %d steps." % n in out
    o.set_isostasy(True); o.step(n); o.set_isostasy(False)
    dt1 = o.compute_dt()
    f0 = read_frame("iso.save.000000")
    nn, ne = host.nnode, host.nelem
    assert np.array_equal(as_f64(f0["coordinate"], nn, 3).T.ravel(), o.download("COORD"))
    assert np.array_equal(as_f64(f0["stress"], ne, 6).T.ravel(), o.download("STRESS"))
    info = open("iso.info").read().split("\n")[0].split()
    assert int(info[1]) == 0 and float(info[2]) == 0.0 and float(info[3]) == pytest.approx(dt1, rel=1e-5)
    o.step(20)
    f1 = read_frame("iso.save.000001")
    assert np.array_equal(as_f64(f1["coordinate"], nn, 3).T.ravel(), o.download("COORD"))
    assert np.array_equal(as_f64(f1["velocity"], nn, 3).T.ravel(), o.download("VEL"))


def test_engine_table_without_the_entry_is_refused(in_tmp):
    ov = "sim.modelname = iso2\nsim.max_steps = 2\nic.isostasy_adjustment_time_in_yr = 50\n"
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides=ov)
    api = oracle_api()
    api.set_isostasy = driver.ISO_T()
    with pytest.raises(des.DesError) as e:
        driver.run(host, api=api)
    assert e.value.code == 31


def _loopback(host, nranks, engine_cls, niso, nsteps):
    parts = [decomp.Partition(host, nranks, r) for r in range(nranks)]
    engs = [engine_cls(p) for p in parts]
    steppers = [decomp.PhasedStepper(e, p, None) for e, p in zip(engs, parts)]
    comm = decomp.LoopbackComm(steppers)

    class NoReduce:
        def reduce_dt(self, engine, recompute):
            return None
    for e, p in zip(engs, parts):
        decomp.init_rank(e, p, NoReduce())
    comm.reduce_dt_all(recompute=True)
    for e in engs: e.set_isostasy(True)
    decomp.run_loopback(steppers, niso)
    for e in engs: e.set_isostasy(False)
    comm.reduce_dt_all(recompute=True)
    decomp.run_loopback(steppers, nsteps)
    return parts, engs


def _assert_assembled(host, parts, engs, ref):
    for f, nc, kind, n in (("COORD", 3, "node", host.nnode), ("VEL", 3, "node", host.nnode), ("TEMPERATURE", 1, "node", host.nnode),
                           ("STRESS", 6, "elem", host.nelem), ("STRAIN", 6, "elem", host.nelem), ("PLSTRAIN", 1, "elem", host.nelem)):
        got = decomp.assemble(parts, [e.download(f) for e in engs], nc, n, kind)
        assert np.array_equal(got, ref.download(f)), f


def test_decomposed_oracle_equals_one_oracle():
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP))
    ref = OracleEngine(host)
    iso_run(ref, host, 12)
    ref.compute_dt()
    ref.step(12)
    parts, engs = _loopback(host, 3, OracleEngine, 12, 12)
    _assert_assembled(host, parts, engs, ref)


# ---- on the MI355X -----------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,overrides", [
    ("ep", cfgs.EP, None),
    ("ep_no_winkler", cfgs.EP, NO_WINKLER),
    ("ep_no_surface_process_2mat", dict(cfgs.EP, nmat=2), "control.surface_process_option = 0\n"),
    ("elastic_damping3", dict(cfgs.EP, rheol="elastic"), "control.damping_option = 3\n"),
])
def test_device_isostasy_loop_bit_exact(name, kw, overrides):
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=overrides)
    d, o = des.DeviceEngine(host), OracleEngine(host)
    (dtd, sd), (dto, so) = iso_run(d, host, 40), iso_run(o, host, 40)
    assert dtd == dto and (sd.time, sd.steps, sd.dt) == (so.time, so.steps, so.dt) == (0.0, 0, dto)
    same(d, o)
    assert d.compute_dt() == o.compute_dt()
    sd, so = d.step(25), o.step(25)                          # ... and the time steps that follow
    assert (sd.time, sd.steps, sd.dt) == (so.time, so.steps, so.dt)
    same(d, o)


@pytest.mark.gpu
def test_device_isostasy_loop_with_creep_and_yield_bit_exact_with_one_libm():
    with portable_libm():
        for kw in (cfgs.EVP, cfgs.YIELD):
            host = des.Host(cfg_text=cfgs.make(**kw))
            d, o = des.DeviceEngine(host), OracleEngine(host)
            iso_run(d, host, 40); iso_run(o, host, 40)
            assert d.compute_dt() == o.compute_dt()
            d.step(20); o.step(20)
            same(d, o)


@pytest.mark.gpu
def test_device_isostasy_within_1e10_default_libm():
    host = des.Host(cfg_text=cfgs.make(**cfgs.EVP))
    d, o = des.DeviceEngine(host), OracleEngine(host)
    iso_run(d, host, 40); iso_run(o, host, 40)
    for f in ("COORD", "VEL", "STRESS", "STRAIN", "VISCOSITY"):
        a, b = d.download(f), o.download(f)
        assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), f


@pytest.mark.gpu
def test_decomposed_device_engines_equal_one_oracle():
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP))
    ref = OracleEngine(host)
    iso_run(ref, host, 12)
    ref.compute_dt()
    ref.step(12)
    parts, engs = _loopback(host, 3, des.DeviceEngine, 12, 12)
    _assert_assembled(host, parts, engs, ref)


@pytest.mark.gpu
def test_executable_runs_the_isostasy_adjustment(tmp_path):
    import subprocess
    cfg = tmp_path / "iso.cfg"
    cfg.write_text(cfgs.apply_overrides(cfgs.make(**cfgs.EP), "sim.modelname = isoexe\nsim.max_steps = 20\n"
                                        "sim.output_step_interval = 20\nic.isostasy_adjustment_time_in_yr = 50\n"))
    exe = os.path.join(des.REPO_ROOT, "dynearthsol_amd", "bin", "dynearthsol3d-hip")
    r = subprocess.run([exe, str(cfg)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Adjusting isostasy for 50 yrs..." in r.stdout and "Adjusted isostasy for" in r.stdout
    host = des.Host(cfg_path=str(cfg))
    o = OracleEngine(host)
    dt0 = o.init_from_host(host)
    n = int(50 * YEAR2SEC / dt0)
    o.set_isostasy(True); o.step(n); o.set_isostasy(False)
    o.compute_dt()
    o.step(20)
    f1 = read_frame(str(tmp_path / "isoexe.save.000001"))
    assert np.array_equal(as_f64(f1["coordinate"], host.nnode, 3).T.ravel(), o.download("COORD"))
    assert np.array_equal(as_f64(f1["stress"], host.nelem, 6).T.ravel(), o.download("STRESS"))
