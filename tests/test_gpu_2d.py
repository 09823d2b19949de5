"""The 2-D (triangle) build -- BASELINE configs[0], SURVEY.md 8 rows a1 / a8 / a9 / a16 / a17 / a19 in
their !THREED form -- on the GPU, through the same C-ABI (des_dev_create with des_params::ndims = 2),
against the oracle compiled -DDES_NDIMS=2 (oracle/libdes_oracle2d.so).

Bar: every field equal to the CPU build's bit for bit.  The device keeps the reference's operation
and summation order (-ffp-contract=off) and its pow / exp return glibc's bits (tests/test_libm.py), so
models that do not yield compare against the oracle as it is (glibc).  Where the Mohr-Coulomb law's
sin / tan decide (yielding models) the oracle is switched to the same portable libm as the device
(`portable_libm()`, as tests/test_gpu_parity_portable_libm.py does for 3-D) and the bits are equal
again; against glibc those runs agree to the compare.py metric written in the test.  l2_residual is a
tree sum on the device and a serial one on the CPU: 1e-12 relative.
"""
import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from oracle_binding import OracleEngine, portable_libm

pytestmark = pytest.mark.gpu

STATE = ("COORD", "VEL", "FORCE", "TEMPERATURE", "STRESS", "STRAIN", "STRAIN_RATE", "PLSTRAIN",
         "DELTA_PLSTRAIN", "VISCOSITY", "VOLUME", "VOLUME_OLD", "VOLUME_N", "MASS", "TMASS", "DPRESSURE",
         "DH", "DHACC", "EDVACC_SURF", "FORCE_RESIDUAL", "STRESSYY", "EDVOLDT", "NTMP")


def pair(kw, overrides=None):
    host = des.Host(cfg_text=cfgs.make(**kw), overrides=overrides, ndims=2)
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)       # first compute_dt
    return host, dev, ora


def assert_bit_exact(dev, ora, fields=STATE):
    for f in fields:
        a, b = dev.download(f), ora.download(f)
        assert a.shape == b.shape, f
        if not np.array_equal(a, b):
            m = np.abs(b).max()
            raise AssertionError("%s: max rel diff %.3e, %d of %d entries differ"
                                 % (f, np.abs(a - b).max() / (m if m else 1), int((a != b).sum()), a.size))


def run(dev, ora, calls, per_call):
    for _ in range(calls):
        sd, so = dev.step(per_call), ora.step(per_call)
        assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel, sd.status) == (so.dt, so.time, so.steps, so.max_surf_vel, so.status)
        assert (sd.max_global_vel_mag, sd.global_dt_min, sd.n_return_mapping) == (so.max_global_vel_mag, so.global_dt_min, so.n_return_mapping)
        assert sd.l2_residual == pytest.approx(so.l2_residual, rel=1e-12)
        assert_bit_exact(dev, ora)
    assert dev.check_nan() == 0


def test_a_2d_model_has_the_references_2d_shapes():
    host, dev, ora = pair(cfgs.EP)
    nn, ne = host.nnode, host.nelem
    assert (nn, ne) == (21 * 5, 2 * 20 * 4)                              # nx * nz nodes, two triangles per cell
    assert dev.field_count("COORD") == 2 * nn and dev.field_count("STRESS") == 3 * ne and dev.field_count("STRESSYY") == ne
    assert host.mesh.etop == host.mesh.ntop - 1 == 20
    assert_bit_exact(dev, ora, ("VOLUME", "VOLUME_OLD", "VOLUME_N", "MASS", "TMASS", "VEL", "COORD"))
    vol = dev.download("VOLUME")
    assert vol.sum() == pytest.approx(40e3 * 8e3, rel=1e-12) and vol.min() > 0        # triangle areas tile the box


@pytest.mark.parametrize("rheol", ["elastic", "viscous", "maxwell", "elasto-plastic", "elasto-visco-plastic"])
def test_every_rheology_is_bit_exact(rheol):
    kw = dict(cfgs.EVP, rheol=rheol, res=1e3)
    host, dev, ora = pair(kw)
    run(dev, ora, 4, 25)


def test_yielding_elasto_plastic_is_bit_exact():
    # fast loading: shear and tensile returns of the 2-D Mohr-Coulomb law (rheology.cxx:371-483, !THREED)
    with portable_libm():
        host, dev, ora = pair(dict(cfgs.YIELD, res=1e3))
        run(dev, ora, 6, 50)
    dpl = dev.download("PLSTRAIN") - np.asarray(host.array("plstrain"))
    assert (dpl > 0).sum() > 20, "the model was meant to yield"


def test_yielding_model_against_glibc_within_1e9():
    # device (portable libm: pow / exp = glibc's bits, sin / tan within 1 ulp) vs the oracle on glibc
    host, dev, ora = pair(dict(cfgs.YIELD, res=1e3))
    dev.step(100), ora.step(100)
    for f in ("COORD", "VEL", "STRESS", "STRAIN", "PLSTRAIN", "TEMPERATURE"):
        a, b = dev.download(f), ora.download(f)
        assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), f           # compare.py metric (benchmarks-cores/compare.py:102-109)


@pytest.mark.parametrize("rheol", ["elasto-plastic", "elasto-visco-plastic"])
def test_plane_strain_elasto_plastic2d_is_bit_exact(rheol):
    # mat.is_plane_strain: elasto_plastic2d with the out-of-plane stress (rheology.cxx:486-701)
    kw = dict(cfgs.YIELD, rheol=rheol, res=1e3, mat_extra="is_plane_strain = yes\n")
    if rheol != "elasto-plastic":
        kw.update(tmantle=1573, alpha=3e-5, vmin="1e19", ic="oceanic_plate_age_in_yr = 2e5\n")
    with portable_libm():
        host, dev, ora = pair(kw)
        assert host.params.is_plane_strain == 1 and np.abs(host.array("stressyy")).max() > 0
        run(dev, ora, 6, 50)
    syy = dev.download("STRESSYY")
    assert not np.array_equal(syy, host.array("stressyy"))
    if rheol == "elasto-plastic":
        assert (dev.download("PLSTRAIN") - np.asarray(host.array("plstrain")) > 0).sum() > 20


def test_two_materials_water_loading_and_quality_interval_are_bit_exact():
    # layered materials (marker-count means), water load on the top, dhacc reset + plastic-strain decay of
    # the top elements every quality_check_step_interval steps (bc.cxx:1837-1850)
    kw = dict(cfgs.EVP, nmat=2, res=1e3, qcsi=7, water="yes", control="surf_base_level = -100\nsurf_diff_ratio_marine = 0.5\n")
    host, dev, ora = pair(kw)
    run(dev, ora, 5, 20)
    assert np.abs(dev.download("DHACC")).max() > 0


BCS = [
    "vbc_x0 = 3\nvbc_x1 = 2\nbottom_shear_zone_thickness = 3e3\n",
    "vbc_x0 = 4\nvbc_x1 = 6\nvbc_val_x1_l = 2e-10\nvbc_z0 = 1\nvbc_val_z0 = 1e-10\nhas_winkler_foundation = no\n",
    "vbc_x0 = 0\nvbc_x1 = 1\nvbc_z0 = 4\nvbc_val_z0 = 3e-10\nhas_winkler_foundation = no\nvbc_z1 = 2\n",
    "num_vbc_period_x0 = 3\nvbc_period_x0_time_in_yr = [0, 1, 2]\nvbc_period_x0_ratio = [1, 0.5, 2]\n"
    "vbc_val_division_x0_min = 0.3\nvbc_val_division_x0_max = 0.6\nvbc_val_x0_ratio0 = 1\nvbc_val_x0_ratio1 = 0.8\n"
    "vbc_val_x0_ratio2 = 0.2\nvbc_val_x0_ratio3 = 0\nhas_elastic_foundation = yes\nelastic_foundation_constant = 1e8\n",
    "stress_bc_x1 = 1\nstress_val_x1 = 1e6\nstress_bc_z1 = 3\nstress_val_z1 = -2e6\nvbc_x1 = 0\n",
]


def bc_overrides(bc):
    return "".join("bc." + line + "\n" for line in bc.strip().splitlines())


@pytest.mark.parametrize("bc", BCS)
def test_2d_boundary_conditions_are_bit_exact(bc):
    # the 2-D apply_vbcs (bc.cxx:247-300, 425-481, 587-650): time-dependent and depth-dependent side
    # velocities, the sheared bottom zone, tangential loading, z types up to 4; Neumann tractions
    with portable_libm():                    # two of these load the model past yield
        host, dev, ora = pair(dict(cfgs.EP, res=1e3), overrides=bc_overrides(bc))
        run(dev, ora, 3, 20)


@pytest.mark.parametrize("opt", [1, 2, 3, 4])
def test_damping_options_are_bit_exact(opt):
    host, dev, ora = pair(dict(cfgs.EP, res=1e3, control="damping_option = %d\n" % opt))
    run(dev, ora, 2, 20)


def test_averaged_fields_and_isostasy_are_bit_exact():
    kw = dict(cfgs.EVP, res=1e3, qcsi=5)
    host, dev, ora = pair(kw, overrides="sim.is_outputting_averaged_fields = yes\nsim.output_step_interval = 50\n")
    for e in (dev, ora):
        e.set_isostasy(1)
        e.step(5)
        e.set_isostasy(0)
    assert dev.compute_dt() == ora.compute_dt()
    assert_bit_exact(dev, ora)
    run(dev, ora, 2, 13)
    assert_bit_exact(dev, ora, ("STRESS_AVG", "DPLSTRAIN_AVG", "STRAIN0", "COORD_AVG0"))


def test_mesh_quality_reductions_match():
    import ctypes as C
    host, dev, ora = pair(cfgs.EP)
    dev.step(30), ora.step(30)
    thr = float(np.median(ora.download("VOLUME")))          # half of the elements are "too small": the first one is reported

    class Q(C.Structure):
        _fields_ = [("small_elem", C.c_int), ("bottom_node", C.c_int), ("worst_elem", C.c_int), ("pad_", C.c_int),
                    ("worst_quality", C.c_double)]
    out = []
    for e in (dev, ora):
        q = Q()
        f = e._f("mesh_quality")
        f.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        assert f(e._h, thr, -8e3, 1e-6, C.byref(q)) == 0
        out.append((q.small_elem, q.bottom_node, q.worst_elem, q.worst_quality))
    assert out[0] == out[1] and out[0][0] >= 0 and 0 < out[0][3] < 1


def test_what_a_2d_model_cannot_have_is_refused_with_the_dimension_code():
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), ndims=2)
    dev = des.DeviceEngine(host)
    import ctypes as C
    assert dev._lib.des_dev_phase(dev._h, 0) == 30 and dev._lib.des_dev_exchange(dev._h) == 30
    with pytest.raises(des.DesError) as ei:
        des.DeviceEngine(des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides="control.has_PT = yes\n", ndims=2))
    assert ei.value.code == 30
    # a 3-D handle still refuses a 2-D model's arrays by size
    with pytest.raises(des.DesError):
        dev.upload("COORD", np.zeros(3 * host.nnode))
