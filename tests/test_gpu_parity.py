"""Parity of the HIP path (through the C-ABI of include/des_dev.h) with the CPU oracle.

Bars (north_star): integer data bit-exact; fp64 fields within 1e-10 relative -- and in fact
BIT-EXACT wherever no transcendental function is involved (elastic / elasto-plastic below
yield), because the device keeps the reference's operation and summation order and is built
with -ffp-contract=off.  Where libm enters (creep law pow/exp, Mohr-Coulomb sin/tan, the
Cardano eigen-solver atan2/cos/sin) glibc and ROCm's ocml differ by <= 2 ulp per call; those
cases use the compare.py metric (max|d|/max|ref|, benchmarks-cores/compare.py:102-109) with the
tolerance written next to each test.
"""
import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from oracle_binding import OracleEngine

pytestmark = pytest.mark.gpu

STATE = ("COORD", "VEL", "FORCE", "TEMPERATURE", "STRESS", "STRAIN", "STRAIN_RATE", "PLSTRAIN",
         "DELTA_PLSTRAIN", "VISCOSITY", "VOLUME", "VOLUME_OLD", "VOLUME_N", "MASS", "TMASS", "DPRESSURE",
         "DH", "DHACC", "EDVACC_SURF", "FORCE_RESIDUAL")


def reldiff(ref, new):
    m = np.abs(ref).max()
    d = np.abs(new - ref).max()
    return d if m == 0 else d / m


def pair(kw, overrides=None):
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=overrides)
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)       # first compute_dt
    return host, dev, ora


def assert_bit_exact(dev, ora, fields=STATE):
    for f in fields:
        a, b = dev.download(f), ora.download(f)
        assert np.array_equal(a, b), "%s: max rel diff %.3e" % (f, reldiff(b, a))


def assert_close(dev, ora, tol, fields=STATE, skip=()):
    for f in fields:
        if f in skip:
            continue
        r = reldiff(ora.download(f), dev.download(f))
        assert r <= tol, "%s: rel diff %.3e > %.1e" % (f, r, tol)


def test_init_geometry_and_first_dt_bit_exact():
    host, dev, ora = pair(cfgs.EP)
    assert_bit_exact(dev, ora, ("VOLUME", "VOLUME_OLD", "VOLUME_N", "MASS", "TMASS", "VEL", "COORD"))


def test_elasto_plastic_100_steps_bit_exact():
    # the reference's test-3d.cfg physics (no element yields)
    host, dev, ora = pair(cfgs.EP)
    for _ in range(4):
        sd, so = dev.step(25), ora.step(25)
        assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel)
        assert sd.l2_residual == pytest.approx(so.l2_residual, rel=1e-12)   # tree vs serial sum
        assert_bit_exact(dev, ora)
    assert dev.check_nan() == 0


def test_hipgraph_replay_of_interior_steps_gives_the_same_bits(monkeypatch):
    """DES_GRAPH=1 replays the launches of interior steps from two captured hipGraphs (with /
    without the compute_dt variant of E1); steps at call boundaries and quality-check steps are
    launched directly.  Off by default: the stream is not launch-bound."""
    ov = "mesh.quality_check_step_interval = 15\n"
    host, dev, ora = pair(cfgs.EVP, overrides=ov)
    monkeypatch.setenv("DES_GRAPH", "1")
    host2 = des.Host(cfg_text=cfgs.make(**cfgs.EVP), overrides=ov)
    devg = des.DeviceEngine(host2)
    devg.init_from_host(host2)
    for n in (47, 3, 1, 26):
        sd, sg = dev.step(n), devg.step(n)
        assert (sd.dt, sd.time, sd.steps) == (sg.dt, sg.time, sg.steps)
        assert_bit_exact(dev, devg)


def test_interior_steps_leave_out_output_only_stores_same_bits(monkeypatch):
    """Inside a multi-step call E2<GEO> does not store strain_rate / viscosity / delta_plstrain / volume_old
    (nothing reads them before the next launch overwrites them); the last step of every call does.  Every
    downloadable field must equal the oracle's and the DES_E2_ELIDE=0 engine's after calls of any length,
    a compute_dt step (10, 20, ...) and a quality-check step (15, 30, ...) being the last or not."""
    ov = "mesh.quality_check_step_interval = 15\n"
    host, dev, ora = pair(cfgs.EVP, overrides=ov)
    monkeypatch.setenv("DES_E2_ELIDE", "0")
    host2 = des.Host(cfg_text=cfgs.make(**cfgs.EVP), overrides=ov)
    dev0 = des.DeviceEngine(host2)
    dev0.init_from_host(host2)
    for n in (2, 8, 1, 4, 15, 11, 9):
        sd, s0, so = dev.step(n), dev0.step(n), ora.step(n)
        assert (sd.dt, sd.time, sd.steps) == (s0.dt, s0.time, s0.steps) == (so.dt, so.time, so.steps)
        assert_bit_exact(dev, dev0)
        assert_bit_exact(dev, ora)                 # STATE holds those four fields


@pytest.mark.parametrize("ov", ["mesh.quality_check_step_interval = 25\n",
                                "mesh.quality_check_step_interval = 4\nsim.is_outputting_averaged_fields = yes\n",
                                "control.surface_process_option = 0\n"])
def test_surface_step_left_to_the_next_steps_passes_same_bits(monkeypatch, ov):
    """Inside a multi-step call a plain step launches no S2 / S3: the next step's EN1 redoes the surface diffusion
    for the surface nodes of each patch and commits the heights with its own records, the next stress update adds
    the edvacc_surf terms (engine/launch.hpp: s2_defer_ok).  Every downloadable field -- DH, DHACC, EDVACC_SURF and
    the averaged fields among them -- must equal the oracle's and the DES_S2_DEFER=0 engine's after calls of any
    length, compute_dt (10, 20, ...) and quality-check steps falling first, last or in between; and the launches
    must really be gone."""
    host, dev, ora = pair(cfgs.EVP, overrides=ov)
    monkeypatch.setenv("DES_S2_DEFER", "0")
    host2 = des.Host(cfg_text=cfgs.make(**cfgs.EVP), overrides=ov)
    dev0 = des.DeviceEngine(host2)
    dev0.init_from_host(host2)
    monkeypatch.delenv("DES_S2_DEFER")
    dev.profile_enable(True); dev0.profile_enable(True)
    fields = STATE + (("STRESS_AVG", "DPLSTRAIN_AVG", "STRAIN0", "COORD_AVG0") if "averaged" in ov else ())
    total = 0
    for n in (2, 8, 1, 4, 15, 11, 9, 3):
        sd, s0, so = dev.step(n), dev0.step(n), ora.step(n)
        total += n
        assert (sd.dt, sd.time, sd.steps, sd.l2_residual, sd.max_surf_vel) == (s0.dt, s0.time, s0.steps, s0.l2_residual, s0.max_surf_vel)
        assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel)
        assert_bit_exact(dev, dev0, fields)
        assert_bit_exact(dev, ora, fields)
    k, k0 = (dict((name, calls) for name, _, calls in d.profile_read()) for d in (dev, dev0))
    assert k0["S3_edvacc_step_finalize"] == total and k["S3_edvacc_step_finalize"] < total - 15, (k, k0)
    assert k.get("S2_surface_diffusion", 0) < k0.get("S2_surface_diffusion", 0) or "surface_process_option = 0" in ov


@pytest.mark.parametrize("kw", ["EVP", "YIELD", "EVP2"])
def test_first_step_of_a_call_on_a_finished_state_same_bits(monkeypatch, kw):
    """A call that follows a finished call with nothing uploaded in between starts like an interior step -- EN1, then
    E2<GEO> with nothing pending (engine/launch.hpp: fresh_ok) -- instead of E1<A> + N1 + E2: same bits as the DES_FRESH=0
    engine and as the oracle after calls of any length; an upload or a clock change in between brings the classic first
    step back; and the E1 launch is really gone."""
    cfg = {"EVP": cfgs.EVP, "YIELD": cfgs.YIELD, "EVP2": dict(cfgs.EVP, nmat=2)}[kw]
    host, dev, ora = pair(cfg)
    monkeypatch.setenv("DES_FRESH", "0")
    host2 = des.Host(cfg_text=cfgs.make(**cfg))
    dev0 = des.DeviceEngine(host2)
    dev0.init_from_host(host2)
    monkeypatch.delenv("DES_FRESH")
    dev.profile_enable(True); dev0.profile_enable(True)
    ncalls = 0
    for n in (3, 1, 1, 8, 12, 2, 5):
        sd, s0, so = dev.step(n), dev0.step(n), ora.step(n)
        ncalls += 1
        assert (sd.dt, sd.time, sd.steps, sd.l2_residual, sd.max_surf_vel) == (s0.dt, s0.time, s0.steps, s0.l2_residual, s0.max_surf_vel)
        assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel)
        assert_bit_exact(dev, dev0)
        if kw != "YIELD":                            # (the yielding model against glibc: tests/test_gpu_parity_c_library.py)
            assert_bit_exact(dev, ora)
    k, k0 = (dict((name, calls) for name, _, calls in d.profile_read()) for d in (dev, dev0))
    e1, e10 = k["E1_geom_rotate_strainrate"], k0["E1_geom_rotate_strainrate"]
    assert e10 - e1 == ncalls - 1, (k, k0)           # one E1<A> per call but the first
    # an upload between two calls: the next call must take the classic first step again (the uploaded field counts)
    T = dev.download("TEMPERATURE") + 1.0
    for eng in (dev, dev0, ora):
        eng.upload("TEMPERATURE", T)
    dev.step(4), dev0.step(4), ora.step(4)
    assert_bit_exact(dev, dev0)
    if kw != "YIELD":
        assert_bit_exact(dev, ora)
    k = dict((name, calls) for name, _, calls in dev.profile_read())
    k0 = dict((name, calls) for name, _, calls in dev0.profile_read())
    assert k0["E1_geom_rotate_strainrate"] - k["E1_geom_rotate_strainrate"] == ncalls - 1


def test_step_splitting_is_invisible_on_the_device():
    # E1 fuses the end of step t with the start of step t+1; the API boundary must not show
    host, dev, ora = pair(cfgs.EP)
    host2 = des.Host(cfg_text=cfgs.make(**cfgs.EP))
    dev2 = des.DeviceEngine(host2)
    dev2.init_from_host(host2)
    dev.step(23)
    for n in (1, 9, 10, 3):
        dev2.step(n)
    assert_bit_exact(dev, dev2)


@pytest.mark.parametrize("name,overrides", [
    ("elastic", "mat.rheology_type = elastic\n"),
    ("no_surface_process", "control.surface_process_option = 0\n"),
    ("damping2", "control.damping_option = 2\n"),
    ("damping3", "control.damping_option = 3\n"),
    ("damping4", "control.damping_option = 4\n"),
    ("no_damping", "control.damping_option = 0\n"),
    ("no_thermal", "control.has_thermal_diffusion = no\n"),
    ("no_nmd", "control.is_using_mixed_stress = no\n"),
    ("no_winkler_fixed_bottom", "bc.has_winkler_foundation = no\nbc.vbc_z0 = 1\n"),
    ("elastic_foundation", "bc.has_elastic_foundation = yes\nbc.elastic_foundation_constant = 1e6\n"),
    ("water_loading", "bc.has_water_loading = yes\ncontrol.surf_base_level = 1e3\n"),
    ("side_walls_free", "bc.vbc_y0 = 0\nbc.vbc_y1 = 2\n"),
    ("vbc_types", "bc.vbc_x0 = 3\nbc.vbc_x1 = 6\nbc.vbc_val_x1_l = 2e-10\nbc.vbc_y0 = 5\nbc.vbc_val_y0 = 1e-10\nbc.vbc_y1 = 7\n"),
    ("neumann", "bc.stress_bc_z1 = 3\nbc.stress_val_z1 = 1e6\nbc.stress_bc_x0 = 1\nbc.stress_val_x0 = -2e6\n"),
    ("fixed_dt", "control.fixed_dt = 1e7\n"),
    ("dynamic", "control.is_quasi_static = no\ncontrol.fixed_dt = 1e-2\n"),
    ("no_gravity", "control.gravity = 0\n"),
])
def test_option_matrix_bit_exact(name, overrides):
    host, dev, ora = pair(cfgs.EP, overrides=overrides)
    dev.step(12); ora.step(12)
    assert_bit_exact(dev, ora)


@pytest.mark.parametrize("name,kw,overrides", [
    ("ep_quasi_static", cfgs.EP, "control.has_moving_mesh = no\n"),
    ("elastic_geotherm_dynamic", dict(cfgs.EVP, rheol="elastic"), "control.has_moving_mesh = no\ncontrol.is_quasi_static = no\ncontrol.fixed_dt = 1e-2\n"),
    ("ep_two_materials_no_nmd", dict(cfgs.EP, nmat=2), "control.has_moving_mesh = no\ncontrol.is_using_mixed_stress = no\n"),
])
def test_fixed_mesh_runs_bit_exact(name, kw, overrides):
    """control.has_moving_mesh = no: main() skips update_mesh (dynearthsol.cxx:870-873) -- no
    coordinate update, no surface processes, volumes and masses keep their initial values (also the
    thermal mass, although rho(T) changes) -- while rotate_stress and compute_dt still run."""
    host, dev, ora = pair(kw, overrides=overrides)
    sd, so = dev.step(35), ora.step(35)
    assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel)
    assert np.array_equal(dev.download("COORD"), host.array("coord"))
    assert_bit_exact(dev, ora)


def test_thermal_diffusion_with_a_geotherm_bit_exact():
    # elastic rheology keeps libm out; the geotherm makes update_temperature and rho(T) matter
    host, dev, ora = pair(dict(cfgs.EVP, rheol="elastic"))
    dev.step(40); ora.step(40)
    T0 = host.array("temperature")
    assert np.abs(dev.download("TEMPERATURE") - T0).max() > 1e-3       # conduction really acted
    assert_bit_exact(dev, ora)


def test_two_materials_prem_reference_pressure_bit_exact():
    kw = dict(cfgs.EVP, rheol="elasto-plastic", nmat=2, control="ref_pressure_option = 1\n")
    host, dev, ora = pair(kw, overrides="bc.vbc_y0 = 0\nbc.vbc_y1 = 0\n")
    assert host.params.ref_pressure_option == 1
    dev.step(30); ora.step(30)
    assert_bit_exact(dev, ora)


@pytest.mark.parametrize("rheol", ["elasto-visco-plastic", "maxwell", "viscous"])
def test_viscous_rheologies_within_1e10(rheol):
    # pow()/exp() in the creep law differ by <= 2 ulp between glibc and ocml; after 100 steps
    # the fields agree to ~1e-13.  Bar: 1e-10 (north_star).  dpressure is a difference of two
    # traces of ~2.6e8 Pa stresses (cancellation), so it is compared against the stress scale.
    # (the purely viscous rheology is "experimental" in the reference and blows up after a few
    # steps with these parameters on the CPU too: 1 step)
    nsteps = 100 if rheol != "viscous" else 1
    host, dev, ora = pair(dict(cfgs.EVP, rheol=rheol))
    dev.step(nsteps); ora.step(nsteps)
    assert_close(dev, ora, 1e-10, skip=("DPRESSURE", "DELTA_PLSTRAIN", "FORCE_RESIDUAL"))
    smax = np.abs(ora.download("STRESS")).max()
    assert np.abs(dev.download("DPRESSURE") - ora.download("DPRESSURE")).max() <= 1e-10 * smax
    assert np.array_equal(dev.download("PLSTRAIN"), ora.download("PLSTRAIN"))


def test_two_material_evp_within_1e10():
    host, dev, ora = pair(dict(cfgs.EVP, nmat=2))
    dev.step(60); ora.step(60)
    assert_close(dev, ora, 1e-10, skip=("DPRESSURE", "DELTA_PLSTRAIN", "FORCE_RESIDUAL"))


def transplant(src, dst, fields=("COORD", "VEL", "TEMPERATURE", "STRESS", "STRAIN", "PLSTRAIN")):
    for f in fields:
        dst.upload(f, src.download(f))


def test_mohr_coulomb_return_single_step_from_identical_state():
    """The rare branch gets its own test (0.17 % of elements in production runs): march the
    oracle into heavy yielding, copy that state to both engines, take ONE step on each.
    ~50 % of the elements go through dsyevh3 + return mapping; sin/tan/atan2/cos differ by
    <= 2 ulp, and the eigen-decomposition of a nearly isotropic 2.6e8 Pa tensor amplifies that
    by |s|/|dev s| ~ 1e2-1e3: bar 1e-11 on stress, 1e-9 on the plastic-strain increment."""
    host = des.Host(cfg_text=cfgs.make(**cfgs.YIELD))
    march = OracleEngine(host)
    march.init_from_host(host)
    sc = march.step(60)
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    for eng in (dev, ora):
        eng.init_from_host(host)
        transplant(march, eng)
        eng.init_geometry()
        eng.set_clock(sc.dt, sc.time, 0)
        eng.step(1)
    dpl = ora.download("DELTA_PLSTRAIN")
    assert (dpl > 0).sum() > host.nelem // 4
    assert reldiff(ora.download("STRESS"), dev.download("STRESS")) <= 1e-11
    assert reldiff(ora.download("STRAIN"), dev.download("STRAIN")) <= 1e-13   # via the spin of rotate_stress
    assert reldiff(dpl, dev.download("DELTA_PLSTRAIN")) <= 1e-9
    assert np.array_equal(dev.download("DELTA_PLSTRAIN") > 0, dpl > 0)      # same elements yield
    assert reldiff(ora.download("VEL"), dev.download("VEL")) <= 1e-11


def test_yield_heavy_run_stays_statistically_identical():
    """Half of the mesh yielding every step is a chaotic map: the CPU oracle itself turns a
    1-ulp change of ONE stress component into 2e-8 after 20 steps and 2e-5 after 100 (yield
    decisions and the dsyevh3 -> dsyevq3 fallback are discontinuities; see the next test, and
    rheology.cxx:727 for the reference's own CPU-vs-OpenACC remark).  The device differs from
    glibc by <= 2 ulp in EVERY element, so the bar is: after 30 steps fields within 1e-4 and the
    same set of yielding elements to 1 %; after 100 steps the same statistics (number of
    yielding elements, mean plastic strain) to 1 %."""
    host, dev, ora = pair(cfgs.YIELD)
    dev.step(30); ora.step(30)
    yd, yo = dev.download("DELTA_PLSTRAIN") > 0, ora.download("DELTA_PLSTRAIN") > 0
    assert yo.sum() > host.nelem // 4
    assert (yd != yo).sum() <= 0.01 * host.nelem
    for f in ("COORD", "STRESS", "STRAIN", "PLSTRAIN", "VEL"):
        assert reldiff(ora.download(f), dev.download(f)) <= 1e-4, f
    dev.step(70); ora.step(70)
    yd, yo = dev.download("DELTA_PLSTRAIN") > 0, ora.download("DELTA_PLSTRAIN") > 0
    assert abs(int(yd.sum()) - int(yo.sum())) <= 0.01 * host.nelem
    assert dev.download("PLSTRAIN").mean() == pytest.approx(ora.download("PLSTRAIN").mean(), rel=1e-2)
    assert reldiff(ora.download("COORD"), dev.download("COORD")) <= 1e-4
    assert dev.check_nan() == 0


def test_oracle_is_equally_sensitive_to_one_ulp():
    """Evidence for the bar above, CPU only vs CPU: nudging ONE stress component of ONE element
    by one ulp changes the oracle's own 100-step yield-heavy result by about as much as
    the GPU differs from it."""
    host = des.Host(cfg_text=cfgs.make(**cfgs.YIELD))
    a, b = OracleEngine(host), OracleEngine(host)
    a.init_from_host(host); b.init_from_host(host)
    s = host.array("stress")
    s[host.nelem // 2] = np.nextafter(s[host.nelem // 2], 0)
    b.upload("STRESS", s)
    a.step(100); b.step(100)
    assert reldiff(a.download("STRESS"), b.download("STRESS")) > 1e-9


def test_full_size_mesh_against_the_oracle():
    """BASELINE's full size (1.1M tets, the bench workload): 3 steps against the oracle
    (evp: 1e-12 after 3 steps), then size-independent properties on the device alone:
    run-to-run reproducibility to the bit and invisibility of the step split."""
    import bench
    host = des.Host(cfg_text=bench.BENCH_CFG.format(res=repr(400e3 / 560), xlen=repr(400e3)))
    assert host.nelem == 1097600 and host.nnode == 244035
    dev, ora = des.DeviceEngine(host), OracleEngine(host, omp=True)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    dev.step(3); ora.step(3)
    assert_close(dev, ora, 1e-12, fields=("COORD", "VEL", "STRESS", "STRAIN", "TEMPERATURE", "VOLUME", "MASS"))
    ora.close()
    dev2 = des.DeviceEngine(host)
    dev2.init_from_host(host)
    dev.step(27)
    dev2.step(3); dev2.step(20); dev2.step(7)
    assert_bit_exact(dev, dev2, ("COORD", "VEL", "STRESS", "STRAIN", "PLSTRAIN", "TEMPERATURE", "VISCOSITY"))
    assert dev.check_nan() == 0


def test_errors_are_reference_exit_codes():
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP))
    dev = des.DeviceEngine(host)
    with pytest.raises(des.DesError):
        dev.upload("STRESS", np.zeros(5))                 # wrong size
    with pytest.raises(des.DesError) as e:
        des.DeviceEngine(host, device=99)
    assert e.value.code == 31


AVG_FIELDS = ("STRESS_AVG", "DPLSTRAIN_AVG", "STRAIN0", "COORD_AVG0")


@pytest.mark.parametrize("splits", [(27,), (1, 4, 5, 7, 10)])
def test_averaged_output_fields_bit_exact(splits):
    """Output::average_fields (output.cxx:327-370) runs inside the device step when
    sim.is_outputting_averaged_fields is on (the reference's default): running sums of stress /
    delta_plstrain and the interval-start snapshots agree with the oracle to the bit, across
    interval boundaries (interval 10) and however the steps are batched."""
    kw = dict(cfgs.YIELD, qcsi=10)
    ov = "sim.is_outputting_averaged_fields = yes\nsim.output_step_interval = 100\nmat.rheology_type = elasto-plastic\n"
    host, dev, ora = pair(kw, overrides=ov)
    assert host.params.is_outputting_averaged_fields == 1
    for n in splits:
        sd, so = dev.step(n), ora.step(n)
        # (with half of the elements yielding dt itself differs in the last digits, see
        # test_yield_heavy_run_stays_statistically_identical)
        assert sd.avg_time0 == pytest.approx(so.avg_time0, rel=1e-9) and sd.steps == so.steps
    assert so.avg_time0 > 0
    assert np.abs(ora.download("DPLSTRAIN_AVG")).max() > 0
    assert_close(dev, ora, 1e-4, fields=AVG_FIELDS)
    host, dev, ora = pair(cfgs.EP, overrides="sim.is_outputting_averaged_fields = yes\nmesh.quality_check_step_interval = 10\n")
    for n in splits:
        sd, so = dev.step(n), ora.step(n)
        assert sd.avg_time0 == so.avg_time0
    assert_bit_exact(dev, ora, STATE + AVG_FIELDS)


def _random_overrides(rng):
    """A random but valid combination of the options the step depends on."""
    pick = lambda *a: a[rng.integers(len(a))]
    ov = []
    ov.append("control.damping_option = %d\n" % pick(0, 1, 2, 3, 4))
    ov.append("control.has_thermal_diffusion = %s\n" % pick("yes", "no"))
    ov.append("control.is_using_mixed_stress = %s\n" % pick("yes", "no"))
    ov.append("control.surface_process_option = %d\n" % pick(0, 1))
    ov.append("control.gravity = %s\n" % pick("10", "9.81", "0"))
    quasi = pick(True, True, False)
    ov.append("control.is_quasi_static = %s\n" % ("yes" if quasi else "no"))
    ov.append("control.fixed_dt = %s\n" % ("0" if quasi and pick(True, False) else ("1e7" if quasi else "1e-2")))
    bottom = pick("winkler", "fixed", "elastic")
    if bottom == "fixed":
        ov.append("bc.has_winkler_foundation = no\nbc.vbc_z0 = 1\n")
    elif bottom == "elastic":
        ov.append("bc.has_elastic_foundation = yes\nbc.elastic_foundation_constant = 1e6\n")
    if pick(True, False):
        ov.append("bc.has_water_loading = yes\ncontrol.surf_base_level = 1e3\n")
    ov.append("bc.vbc_x0 = %d\nbc.vbc_x1 = %d\n" % (pick(1, 3), pick(1, 3)))
    ov.append("bc.vbc_y0 = %d\nbc.vbc_y1 = %d\n" % (pick(0, 1, 2, 5), pick(0, 1, 2, 7)))
    if pick(True, False, False):
        ov.append("bc.stress_bc_z1 = 3\nbc.stress_val_z1 = 1e6\n")
    return "".join(ov)


@pytest.mark.parametrize("seed", range(40))
def test_random_option_combinations_bit_exact(seed):
    """Options interact (e.g. damping x quasi-static mass, thermal x surface bc, NMD x foundation):
    seeded random combinations on top of the one-at-a-time matrix, elasto-plastic below yield and
    elastic, one or two materials -- no transcendental on the path, so bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    ov = _random_overrides(rng)
    kw = dict(cfgs.EVP, rheol=["elasto-plastic", "elastic"][seed % 2], nmat=1 + seed % 3 % 2)
    host, dev, ora = pair(kw, overrides=ov)
    dev.step(11); ora.step(11)
    assert ora.check_nan() == 0, ov
    for f in STATE:
        a, b = dev.download(f), ora.download(f)
        assert np.array_equal(a, b), "%s with\\n%s" % (f, ov)


@pytest.mark.parametrize("seed", range(12))
def test_classic_passes_and_patch_passes_give_the_same_bits(seed, monkeypatch):
    """The engine's default since round 2 are the node-block patch passes (EN1 / EN2 / EN3,
    csrc/passes/en*.hpp); the classic pairs (N1, N2, E3 + N3) stay behind DES_PATCH=0 and must keep
    giving the same bits -- over random option combinations, a multi-step call (EN1 only runs inside
    one), single steps, and a compute_dt step."""
    rng = np.random.default_rng(4000 + seed)
    ov = _random_overrides(rng)
    kw = dict(cfgs.EVP if seed % 3 else cfgs.EP, nmat=1 + seed % 2, lx=60e3)
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=ov)
    engines = []
    for patch in ("0", "64"):
        monkeypatch.setenv("DES_PATCH", patch)
        e = des.DeviceEngine(host)
        e.init_from_host(host)
        engines.append(e)
    for n in (1, 7, 1, 13):                                 # steps 1, 8, 9, 22: crosses 10 and 20
        sa, sb = engines[0].step(n), engines[1].step(n)
        assert (sa.dt, sa.steps, sa.n_return_mapping) == (sb.dt, sb.steps, sb.n_return_mapping)
        assert abs(sa.l2_residual - sb.l2_residual) <= 1e-12 * abs(sa.l2_residual)     # summed over other blocks
        for f in STATE:
            a, b = engines[0].download(f), engines[1].download(f)
            assert np.array_equal(a, b), (ov, n, f)


@pytest.mark.parametrize("tol,moving", [("1e-2", "yes"), ("1e-4", "yes"), ("1e-4", "no")])
def test_pseudo_transient_loop_bit_exact(tol, moving):
    """control.has_PT (dynearthsol.cxx:803-864): inside every step the quasi-static part is iterated
    with the boundaries at rest until the residual settles.  Same number of iterations, same bits
    (elasto-plastic: no libm on the path), residual to the rounding of its tree sum."""
    ov = ("control.has_PT = yes\ncontrol.PT_max_iter = 40\ncontrol.PT_relative_tolerance = %s\n"
          "control.has_moving_mesh = %s\n" % (tol, moving))
    host, dev, ora = pair(cfgs.EP, overrides=ov)
    total = 0
    for n in (1, 4, 8):                                     # crosses step 10 (compute_dt)
        sd, so = dev.step(n), ora.step(n)
        assert sd.n_pt_iterations == so.n_pt_iterations > 0 and (sd.dt, sd.steps) == (so.dt, so.steps)
        assert abs(sd.l2_residual - so.l2_residual) <= 1e-12 * so.l2_residual
        total += so.n_pt_iterations
        assert_bit_exact(dev, ora)
    assert total >= (13 if tol == "1e-2" else 30)


@pytest.mark.parametrize("moving", ["yes", "no"])
def test_initial_body_force_adjustment_bit_exact(moving):
    """ic.has_body_force_adjustment (dynearthsol.cxx:546-591, 753-761): before the first step the pseudo-transient loop
    runs on the initial state with apply_stress_bcs_neumann held back (fields.cxx:690).  Same iteration count and bits
    as the oracle, the steps that follow included -- and the tractions were really held back."""
    ov = ("control.has_PT = yes\ncontrol.PT_max_iter = 25\ncontrol.PT_relative_tolerance = 1e-4\n"
          "control.has_moving_mesh = %s\nbc.stress_bc_z1 = 3\nbc.stress_val_z1 = 2e6\nbc.stress_bc_x0 = 1\nbc.stress_val_x0 = -1e6\n" % moving)
    host, dev, ora = pair(cfgs.EP, overrides=ov)
    ref = OracleEngine(host)
    ref.init_from_host(host)
    # (the lithostatic start is in equilibrium: the loop would stop after two idle iterations -- push it out of balance)
    pushed = ora.download("STRESS") * 1.03
    for eng in (dev, ora, ref):
        eng.upload("STRESS", pushed)
    sd, so = dev.body_force_adjustment(), ora.body_force_adjustment()
    assert sd.n_pt_iterations == so.n_pt_iterations > 3 and (sd.dt, sd.steps, sd.time) == (so.dt, so.steps, so.time) == (so.dt, 0, 0.0)
    assert abs(sd.l2_residual - so.l2_residual) <= 1e-12 * so.l2_residual
    assert_bit_exact(dev, ora)
    for n in (1, 3, 8):
        sd, so = dev.step(n), ora.step(n)
        assert sd.n_pt_iterations == so.n_pt_iterations > 0 and (sd.dt, sd.steps) == (so.dt, so.steps)
        assert_bit_exact(dev, ora)
    # the loop on its own, tractions applied, is another model: the step's PT loop of an oracle that skipped the
    # adjustment leaves other forces on the traction boundaries
    ref.step(12)
    assert not np.array_equal(ref.download("VEL"), ora.download("VEL"))


@pytest.mark.parametrize("name,kw,overrides", [
    ("evp_nmd", dict(cfgs.EVP, nmat=2), None),
    ("evp_averaged_fields", cfgs.EVP, "sim.is_outputting_averaged_fields = yes\nmesh.quality_check_step_interval = 5\n"),
    ("evp_no_nmd", cfgs.EVP, "control.is_using_mixed_stress = no\n"),
    ("ep_yielding_water_load", dict(cfgs.EP, nmat=2), "bc.has_water_loading = yes\ncontrol.surf_base_level = 1e3\n"),
    ("maxwell", dict(cfgs.EVP, rheol="maxwell"), None),
    ("elastic_no_thermal", dict(cfgs.EVP, rheol="elastic"), "control.has_thermal_diffusion = no\n"),
])
def test_pipelined_stress_update_and_three_wave_shape_give_the_same_bits(monkeypatch, capfd, name, kw, overrides):
    """E2<GEO> has three launch forms chosen by the size of the launch (engine/launch.hpp): the plain one-pass kernel, the
    same held to three waves per SIMD (shards), and the pipelined kernel whose wavefronts take their tiles' planes out of
    LDS, where they arrive by DMA a tile ahead (1M-tet meshes).  DES_E2_PIPE / DES_E2_W3 (read when an engine is created)
    pin them: on a small mesh all three must leave the bits of the default engine in every field, across the rheologies, with
    Output::average_fields riding in the pass, with and without NMD_stress, across call boundaries (the last step of a call
    stores every field, the first one may start fresh) and compute_dt steps."""
    host = des.Host(cfg_text=cfgs.make(**dict(kw, res=1e3)), overrides=overrides)
    assert host.nelem % 2 == 0 and host.nelem > 4000                   # (the pipelined kernel wants an even plane stride)
    monkeypatch.setenv("DES_PATCH_VERBOSE", "1")                       # the engine says which launch form it took
    engines = [des.DeviceEngine(host)]
    for var, val in (("DES_E2_PIPE", "1"), ("DES_E2_W3", "1")):
        monkeypatch.setenv(var, val)
        engines.append(des.DeviceEngine(host))
        assert var + "=" + val in des.config_string()              # (the library has read it: des_dev_config_string)
        monkeypatch.delenv(var)
    dts = [e.init_from_host(host) for e in engines]
    assert dts[0] == dts[1] == dts[2]
    for n in (7, 1, 16, 9):
        s = [e.step(n) for e in engines]
        for k in (1, 2):
            assert (s[k].dt, s[k].time, s[k].steps, s[k].n_return_mapping) == (s[0].dt, s[0].time, s[0].steps, s[0].n_return_mapping)
            assert_bit_exact(engines[k], engines[0])
    for e in engines:
        e.close()
    assert capfd.readouterr().err.count("E2<GEO>: pipelined launch") == 1         # the second engine only
