"""The 2-D (triangle) build -- BASELINE configs[0], SURVEY.md 8 rows a1 / a8 / a9 / a16 / a17 / a19 in
their !THREED form -- on the GPU, through the same C-ABI (des_dev_create with des_params::ndims = 2),
against the oracle compiled -DDES_NDIMS=2 (oracle/libdes_oracle2d.so).

Bar: every field equal to the CPU build's bit for bit.  The device keeps the reference's operation
and summation order (-ffp-contract=off) and its pow / exp / sin / cos / tan return glibc's bits
(tests/test_libm.py), so models compare against the oracle as it is (glibc), yielding ones included
(round 3: test_2d_yielding_models_equal_the_oracle_on_the_c_library and the two tests after it).  The
older yielding tests switch the oracle to the same portable libm as the device (`portable_libm()`, as
tests/test_gpu_parity_portable_libm.py does for 3-D): kept, they pin the restated routines on the CPU
side too.  l2_residual is a tree sum on the device and a serial one on the CPU: 1e-12 relative.
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


def run(dev, ora, calls, per_call, l2_rel=1e-12):
    for _ in range(calls):
        sd, so = dev.step(per_call), ora.step(per_call)
        assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel, sd.status) == (so.dt, so.time, so.steps, so.max_surf_vel, so.status)
        assert (sd.max_global_vel_mag, sd.global_dt_min, sd.n_return_mapping) == (so.max_global_vel_mag, so.global_dt_min, so.n_return_mapping)
        assert sd.l2_residual == pytest.approx(so.l2_residual, rel=l2_rel)
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


C_LIBRARY_CASES = {
    "shear_and_tensile_returns": (dict(cfgs.YIELD, res=1e3), None, (6, 50)),
    "plane_strain_ep": (dict(cfgs.YIELD, rheol="elasto-plastic", res=1e3, mat_extra="is_plane_strain = yes\n"), None, (6, 50)),
    "plane_strain_evp": (dict(cfgs.YIELD, rheol="elasto-visco-plastic", res=1e3, mat_extra="is_plane_strain = yes\n", tmantle=1573,
                              alpha=3e-5, vmin="1e19", ic="oceanic_plate_age_in_yr = 2e5\n"), None, (6, 50)),
    "two_materials_water_load": (dict(cfgs.EVP, nmat=2, res=1e3, qcsi=7, water="yes",
                                      control="surf_base_level = -100\nsurf_diff_ratio_marine = 0.5\n"), None, (5, 40)),
}


@pytest.mark.parametrize("case", sorted(C_LIBRARY_CASES))
def test_2d_yielding_models_equal_the_oracle_on_the_c_library(case):
    """Round 3: the device's sin / cos / tan return glibc's bits as its pow / exp do (tests/test_libm.py), so 2-D models
    that YIELD equal the oracle running on the host's C library bit for bit -- no common libm compiled into both sides."""
    kw, ov, (calls, per_call) = C_LIBRARY_CASES[case]
    host, dev, ora = pair(kw, overrides=ov)
    run(dev, ora, calls, per_call)
    if "plane_strain" in case:
        assert not np.array_equal(dev.download("STRESSYY"), host.array("stressyy"))
    if case in ("shear_and_tensile_returns", "plane_strain_ep"):
        assert (dev.download("PLSTRAIN") - np.asarray(host.array("plstrain")) > 0).sum() > 20, "the model was meant to yield"


def test_reference_2d_benchmarks_equal_the_oracle_on_the_c_library():
    """benchmarks-cores/test-tiny.cfg for 404 steps (its weak zone yields) and test-topo.cfg for its 2000, against the
    oracle on glibc."""
    import os
    host = des.Host(cfg_text=cfgs.TEST_TINY, mesh_file=os.path.join(des.REPO_ROOT, "tests", "golden", "test-tiny.desmesh"), ndims=2)
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    run(dev, ora, 4, 101)
    host = des.Host(cfg_text=cfgs.TEST_TINY, overrides=cfgs.TEST_TOPO_OVERRIDES, ndims=2,
                    mesh_file=os.path.join(des.REPO_ROOT, "tests", "golden", "test-topo.desmesh"))
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    run(dev, ora, 4, 500)


def test_320k_triangles_500_steps_against_the_oracle_on_the_c_library():
    """400 km x 100 km at 500 m, two materials, evp with a water load and surface diffusion: calls of 97 + 103 + 1 + 199 + 100
    steps (patch passes, the end-of-step pass inside the next stress update, store elision across every kind of call
    boundary), a few hundred elements yielding -- every field equal to the OpenMP oracle's on glibc."""
    kw = dict(cfgs.EVP, nmat=2, lx=400e3, lz=100e3, res=500.0, qcsi=50, water="yes", control="surf_base_level = -100\nsurf_diff_ratio_marine = 0.5\n")
    host = des.Host(cfg_text=cfgs.make(**kw), ndims=2)
    dev, ora = des.DeviceEngine(host), OracleEngine(host, omp=True)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    for n in (97, 103, 1, 199, 100):
        sd, so = dev.step(n), ora.step(n)
        assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel)
        assert_bit_exact(dev, ora)
    assert (dev.download("PLSTRAIN") > 0).sum() > 100


def test_late_surface_step_across_every_kind_of_call_boundary():
    """Round 5: a plain step inside a multi-step call leaves its surface step (simple_diffusion, edvacc_surf,
    correct_surface_element) to the next step's k2p_temp_dvoldt<1> -- except on compute_dt steps, on the steps of the
    quality-check interval and on the last step of a call; a compute_dt step leaves its end-of-step element pass and
    compute_mass behind like any other step (its rotation keeps the dt it ran with).  Call lengths that put every one of those next to every other,
    on a mesh with enough blocks for the balanced launch order (DES2D_TOP_BALANCE): all fields, dh / dhacc / edvacc_surf
    among them, equal to the oracle's after every call -- and to an engine with both switched off."""
    import os
    kw = dict(cfgs.EVP, nmat=2, lx=100e3, lz=30e3, res=500.0, qcsi=7, water="yes", control="surf_base_level = -100\nsurf_diff_ratio_marine = 0.5\n")
    host = des.Host(cfg_text=cfgs.make(**kw), ndims=2)
    dev, ora = des.DeviceEngine(host), OracleEngine(host, omp=True)
    os.environ["DES2D_SURF_DEFER"] = "0"; os.environ["DES2D_TOP_BALANCE"] = "0"; os.environ["DES2D_DT_DEFER"] = "0"
    try:
        plain = des.DeviceEngine(host)
    finally:
        del os.environ["DES2D_SURF_DEFER"], os.environ["DES2D_TOP_BALANCE"], os.environ["DES2D_DT_DEFER"]
    assert dev.init_from_host(host) == ora.init_from_host(host) == plain.init_from_host(host)
    for n in (1, 2, 3, 9, 10, 11, 19, 7, 14, 41):
        sd, so, sp = dev.step(n), ora.step(n), plain.step(n)
        assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel) == (sp.dt, sp.time, sp.steps, sp.max_surf_vel)
        assert sd.l2_residual == sp.l2_residual
        assert_bit_exact(dev, ora)
        assert_bit_exact(dev, plain)
    assert np.abs(dev.download("DH")).max() > 0 and np.abs(dev.download("EDVACC_SURF")).max() > 0


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


OPTIONS = [
    "control.has_moving_mesh = no\n",
    "control.gravity = 0\n",
    "control.is_using_mixed_stress = no\n",
    "control.has_thermal_diffusion = no\n",
    "control.fixed_dt = 1e7\n",
    "control.is_quasi_static = no\ncontrol.dt_fraction = 0.5\n",
    "control.surface_process_option = 0\n",
    "control.ref_pressure_option = 2\nbc.has_winkler_foundation = no\nbc.vbc_z0 = 1\n",
    "control.characteristic_speed = 2e-9\nbc.winkler_delta_rho = 100\n",
    "mesh.meshing_elem_shape = 2\n",
]


@pytest.mark.parametrize("ov", OPTIONS)
def test_2d_option_matrix_is_bit_exact(ov):
    # the switches of the step one at a time, on two materials with a geotherm (evp)
    with portable_libm():
        host, dev, ora = pair(dict(cfgs.EVP, nmat=2, res=1e3), overrides=ov)
        run(dev, ora, 3, 15)


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


@pytest.mark.parametrize("tol,moving", [("1e-2", "yes"), ("1e-4", "yes"), ("1e-4", "no")])
def test_pseudo_transient_loop_bit_exact(tol, moving):
    """control.has_PT (dynearthsol.cxx:803-864) in the 2-D build: the quasi-static part of every step iterated
    with the boundaries at rest (bc.cxx:330-343) until the residual settles.  Same iteration counts, same bits."""
    ov = ("control.has_PT = yes\ncontrol.PT_max_iter = 40\ncontrol.PT_relative_tolerance = %s\n"
          "control.has_moving_mesh = %s\n" % (tol, moving))
    host, dev, ora = pair(dict(cfgs.EP, res=1e3), overrides=ov)
    total = 0
    for n in (1, 4, 8):                                     # crosses step 10 (compute_dt)
        sd, so = dev.step(n), ora.step(n)
        assert sd.n_pt_iterations == so.n_pt_iterations > 0 and (sd.dt, sd.steps) == (so.dt, so.steps)
        assert abs(sd.l2_residual - so.l2_residual) <= 1e-12 * so.l2_residual
        total += so.n_pt_iterations
        assert_bit_exact(dev, ora)
    assert total >= 13


@pytest.mark.parametrize("knob", ["DES2D_PATCH=0", "DES2D_PATCH=64", "DES2D_PATCH=40", "DES2D_CLUSTER=0", "DES2D_GEO=0", "DES2D_ELIDE=0",
                                  "DES2D_CLUSTER_ASPECT=1", "DES2D_MASS_FUSE=0", "DES2D_FOLD=0", "DES2D_SURF_DEFER=0", "DES2D_TOP_BALANCE=0", "DES2D_DT_DEFER=0", "DES2D_PATCH_IT=3", "DES2D_SR_FUSE=0"])
def test_patch_passes_and_plain_kernels_give_the_same_bits(monkeypatch, knob):
    """The node-block patch passes (des_dev2d_patch.hpp: temperature + dvoldt, NMD + force, mass; the default) against the
    one-kernel-per-loop path (DES2D_PATCH=0), other block sizes and groupings, and with the end-of-step pass / the store
    elision switched off: every field equal, and equal to the oracle's."""
    kw = dict(cfgs.EVP, nmat=2, res=1e3, qcsi=7, water="yes", control="surf_base_level = -100\nsurf_diff_ratio_marine = 0.5\n")
    with portable_libm():
        host, dev, ora = pair(kw)
        name, value = knob.split("=")
        monkeypatch.setenv(name, value)
        other = des.DeviceEngine(host)
        monkeypatch.delenv(name)
        other.init_from_host(host)
        for _ in range(3):
            sd, so = dev.step(23), other.step(23)
            ora.step(23)
            assert (sd.dt, sd.time, sd.steps, sd.max_surf_vel) == (so.dt, so.time, so.steps, so.max_surf_vel)
            # (the per-step residual is the by-product of the force pass: summed per patch block since round 5 -- k2p_force<1> --,
            #  per 256 node ids by the plain kernels; the terms are the same bits, the association is the grouping's)
            assert abs(sd.l2_residual - so.l2_residual) <= 1e-12 * so.l2_residual
            assert_bit_exact(dev, other)
            assert_bit_exact(dev, ora)


@pytest.mark.parametrize("moving", ["yes", "no"])
def test_initial_body_force_adjustment_2d_bit_exact(moving):
    """ic.has_body_force_adjustment in the 2-D build (dynearthsol.cxx:546-591): the pseudo-transient loop on the initial
    state, Neumann tractions held back (fields.cxx:690); same iteration count and bits as the 2-D oracle, the steps that
    follow included."""
    ov = ("control.has_PT = yes\ncontrol.PT_max_iter = 25\ncontrol.PT_relative_tolerance = 1e-4\n"
          "control.has_moving_mesh = %s\nbc.stress_bc_z1 = 3\nbc.stress_val_z1 = 2e6\nbc.stress_bc_x1 = 1\nbc.stress_val_x1 = -1e6\nbc.vbc_x1 = 0\n" % moving)
    host, dev, ora = pair(dict(cfgs.EP, res=1e3), overrides=ov)
    pushed = ora.download("STRESS") * 1.03           # (the lithostatic start is in equilibrium: push it out of balance)
    for eng in (dev, ora):
        eng.upload("STRESS", pushed)
    sd, so = dev.body_force_adjustment(), ora.body_force_adjustment()
    assert sd.n_pt_iterations == so.n_pt_iterations > 3 and (sd.dt, sd.steps, sd.time) == (so.dt, 0, 0.0)
    assert abs(sd.l2_residual - so.l2_residual) <= 1e-12 * so.l2_residual
    assert_bit_exact(dev, ora)
    for n in (1, 3, 8):
        sd, so = dev.step(n), ora.step(n)
        assert sd.n_pt_iterations == so.n_pt_iterations > 0 and (sd.dt, sd.steps) == (so.dt, so.steps)
        assert_bit_exact(dev, ora)


def test_what_a_2d_model_cannot_have_is_refused_with_the_dimension_code():
    host = des.Host(cfg_text=cfgs.make(**cfgs.EP), ndims=2)
    dev = des.DeviceEngine(host)
    assert dev._lib.des_dev_exchange(dev._h) == 30                              # (RCCL inside des_dev_step)
    # (des_dev_set_overlap used to be refused too; since round 4 the 2-D engine has the overlapped schedule -- on a mesh
    #  without neighbours it selects nothing)
    dev.set_overlap(True)
    assert not dev.comm_info()["overlapped"]
    dev.set_overlap(False)
    # an upload with the 3-D size of the field is refused
    with pytest.raises(des.DesError):
        dev.upload("COORD", np.zeros(3 * host.nnode))


# ---- the reference's own 2-D case: benchmarks-cores/test-tiny.cfg (BASELINE configs[0]) -------------
import os
import subprocess

from dynearthsol_amd import driver
from test_driver_output import oracle_api, read_frame, in_tmp  # noqa: E402,F401

TINY_MESH = os.path.join(des.REPO_ROOT, "tests", "golden", "test-tiny.desmesh")
EXE2D = os.path.join(des.REPO_ROOT, "dynearthsol_amd", "bin", "dynearthsol2d-hip")


def test_reference_test_tiny_is_bit_exact_on_its_triangle_mesh():
    """Eight materials, evp, Gaussian weak zone, continental geotherm, surface diffusion, PREM reference
    pressure, on the 97-node mesh the reference's Triangle makes from cube.poly: the four steps of the
    benchmark, then 404 with yielding."""
    host = des.Host(cfg_text=cfgs.TEST_TINY, mesh_file=TINY_MESH, ndims=2)
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    run(dev, ora, 4, 1)                          # the benchmark itself, against the oracle on glibc
    with portable_libm():                        # later elements of the weak zone yield now and then: one libm on both sides
        dev, ora = des.DeviceEngine(host), OracleEngine(host)
        assert dev.init_from_host(host) == ora.init_from_host(host)
        run(dev, ora, 4, 101)


def test_reference_test_rect_tiny_is_bit_exact_on_the_equilateral_mesh():
    """benchmarks-cores/test-rect-tiny.cfg: the same model on the 2-D build's own equilateral mesh
    (meshing_elem_shape = 2, built by the host library)."""
    host = des.Host(cfg_text=cfgs.TEST_TINY, overrides=cfgs.TEST_RECT_TINY_OVERRIDES, ndims=2)
    assert (host.nnode, host.nelem) == (99, 155)
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    run(dev, ora, 4, 1)
    with portable_libm():
        dev, ora = des.DeviceEngine(host), OracleEngine(host)
        assert dev.init_from_host(host) == ora.init_from_host(host)
        run(dev, ora, 4, 101)


def test_reference_test_topo_2000_steps_bit_exact():
    """benchmarks-cores/test-topo.cfg: 10 km of relief on the top boundary (topo.poly through the reference's
    Triangle), surface diffusivity 1e-2, the 2000 steps of the benchmark: the 1-D surface diffusion with its
    terrigenous / marine branches, dhacc resets and the top elements' rescaling every second step."""
    host = des.Host(cfg_text=cfgs.TEST_TINY, overrides=cfgs.TEST_TOPO_OVERRIDES, ndims=2,
                    mesh_file=os.path.join(des.REPO_ROOT, "tests", "golden", "test-topo.desmesh"))
    assert (host.nnode, host.nelem, host.mesh.etop) == (361, 653, 24)
    z0 = host.array("coord").reshape(2, -1)[1].max()
    with portable_libm():
        dev, ora = des.DeviceEngine(host), OracleEngine(host)
        assert dev.init_from_host(host) == ora.init_from_host(host)
        run(dev, ora, 4, 500)
    z1 = dev.download("COORD").reshape(2, -1)[1].max()
    assert z0 == 10e3 and 9.4e3 < z1 < 9.7e3                     # the ridge has been worn down


def test_a_million_triangles_against_the_oracle():
    """The 2-D engine at size: 400 km x 100 km at 250 m = 1.28M triangles (regular 2-D mesher), evp, 20 steps."""
    kw = dict(cfgs.EVP, lx=400e3, lz=100e3, res=250.0)
    host = des.Host(cfg_text=cfgs.make(**kw), ndims=2)
    assert host.nelem == 2 * 1600 * 400
    dev, ora = des.DeviceEngine(host), OracleEngine(host)
    assert dev.init_from_host(host) == ora.init_from_host(host)
    run(dev, ora, 2, 10, l2_rel=1e-10)           # a serial sum over 642k nodes against a tree


def test_dynearthsol2d_executable_writes_the_frames_the_oracle_loop_writes(in_tmp):
    with open("tiny.cfg", "w") as f:
        f.write(cfgs.apply_overrides(cfgs.TEST_TINY, "sim.modelname = gpu\nsim.has_initial_checkpoint = yes\n"))
    out = subprocess.run([EXE2D, "tiny.cfg", "--mesh", TINY_MESH], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "Output # 4" in out.stdout and "Ending simulation." in out.stdout
    host = des.Host(cfg_path="tiny.cfg", mesh_file=TINY_MESH, ndims=2, overrides="sim.modelname = cpu\n")
    st = driver.run(host, api=oracle_api(2))
    assert st.frames == 5
    for frame in range(5):
        a, b = read_frame("gpu.save.%06d" % frame, ndims=2), read_frame("cpu.save.%06d" % frame, ndims=2)
        assert sorted(a) == sorted(b)
        for name in a:
            if name != "walltime_sec":
                assert np.array_equal(a[name], b[name]), (frame, name)
    # the same program under its 3-D name refuses the 2-D mesh file
    out3 = subprocess.run([EXE2D.replace("2d", "3d"), "tiny.cfg", "--mesh", TINY_MESH], capture_output=True, text=True, timeout=300)
    assert out3.returncode == 30
    # ... and takes the dimension as an option
    out2 = subprocess.run([EXE2D.replace("2d", "3d"), "tiny.cfg", "--ndims", "2", "--mesh", TINY_MESH, "--quiet"],
                          capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr


def test_python_driver_runs_a_2d_model_on_the_device(in_tmp):
    host = des.Host(cfg_text=cfgs.make(**dict(cfgs.EVP, nmat=2, res=1e3)), ndims=2,
                    overrides="sim.max_steps = 60\nsim.output_step_interval = 30\nsim.modelname = py2d\nmesh.quality_check_step_interval = 10\n")
    st = driver.run(host)
    assert (st.steps, st.frames, st.exit_code) == (60, 3, 0)
    fr = read_frame("py2d.save.000002", ndims=2)
    assert fr["steps"].view(np.int32)[0] == 60 and fr["stress"].size == 3 * host.nelem * 8
