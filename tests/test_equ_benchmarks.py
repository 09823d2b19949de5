"""The reference's regular-mesh 3-D benchmarks, benchmarks-cores/test-3d-equ-tiny.cfg and
test-3d-equ-long.cfg (values restated in cfgs.EQU): the host library meshes them itself
(meshing_option = 1, meshing_elem_shape = 1), so they run from the .cfg alone."""
import os

import numpy as np
import pytest

import cfgs
import dynearthsol_amd as des
from oracle_binding import OracleEngine

REF_CFG = "/root/reference/benchmarks-cores/test-3d-equ-%s.cfg"


def test_tiny_model_as_the_reference_builds_it():
    h = des.Host(cfg_text=cfgs.make_equ())
    assert (h.nnode, h.nelem) == (3978, 12500)                       # 50 x 2 x 25 cells, 5 tets each
    p = h.params
    assert p.nmat == 7 and p.has_water_loading == 1 and p.rheol_type == 7
    mk = h.array("elemmarkers").reshape(-1, 7)
    assert (mk.sum(axis=1) == 8).all()
    # two layers (crust over mantle) and the adiabatic part turned into asthenosphere by the
    # geotherm (radiogenic_heat_and_adiabat, ic.cxx:815-829)
    used = np.nonzero(mk.sum(axis=0))[0].tolist()
    assert used == [1, 2, 3]
    T = h.array("temperature")
    z = h.array("coord").reshape(3, -1)[2]
    assert T.min() == 273 and T.max() == pytest.approx(1723 * np.exp(9.81 * 125e3 * 4e-8))
    assert np.all(np.diff(T[np.argsort(-z)][::200]) >= -1e-9)        # hotter with depth
    assert h.array("radiogenic").max() > 0
    ora = OracleEngine(h)
    ora.init_from_host(h)
    sc = ora.step(100)
    assert ora.check_nan() == 0 and sc.steps == 100


@pytest.mark.skipif(not os.path.exists(REF_CFG % "tiny"), reason="reference tree only in the build container")
@pytest.mark.parametrize("which", ["tiny", "long"])
def test_restated_values_equal_the_reference_files(which):
    """The .cfg files of the reference repository, unchanged, give the same model."""
    if which == "long":
        ov = "mesh.resolution = 10e3\n"                              # same file, coarser: keeps the test quick
        a = des.Host(cfg_path=REF_CFG % which, overrides=ov)
        b = des.Host(cfg_text=cfgs.make_equ(long=True), overrides=ov)
    else:
        a, b = des.Host(cfg_path=REF_CFG % which), des.Host(cfg_text=cfgs.make_equ())
    assert bytes(a.params) == bytes(b.params)
    for name in ("coord", "connectivity", "temperature", "stress", "elemmarkers", "radiogenic", "viscosity"):
        assert np.array_equal(a.array(name), b.array(name)), name


@pytest.mark.gpu
def test_tiny_benchmark_device_against_oracle():
    """The file's 400 steps.  This model answers a 1-ulp change of the initial stress with a 1e-5
    relative change of the velocities after ONE step (they are small differences of large forces),
    and yielding sets in around step 50.  Measured: device == oracle to the bit for 20 steps, 1e-16
    at 50, 2e-9 at 400 -- six orders below the oracle's own 1-ulp response at every stage."""
    h = des.Host(cfg_text=cfgs.make_equ())
    dev, ora, pert = des.DeviceEngine(h), OracleEngine(h), OracleEngine(h)
    assert dev.init_from_host(h) == ora.init_from_host(h) == pert.init_from_host(h)
    s0 = pert.download("STRESS")
    pert.upload("STRESS", np.nextafter(s0, 2 * s0))
    fields = ("COORD", "VEL", "STRESS", "STRAIN", "TEMPERATURE", "PLSTRAIN", "VISCOSITY", "MASS", "DHACC")
    rel = lambda ref, new: np.abs(new - ref).max() / max(np.abs(ref).max(), 1e-300)
    sd, so = dev.step(50), ora.step(50)
    pert.step(50)
    assert (sd.dt, sd.steps) == (so.dt, so.steps)
    for f in fields:
        assert rel(ora.download(f), dev.download(f)) <= 1e-12, f
    sd, so = dev.step(350), ora.step(350)                            # the file's max_steps
    pert.step(350)
    assert sd.steps == so.steps == 400 and dev.check_nan() == 0
    for f in fields:
        ref = ora.download(f)
        r = rel(ref, dev.download(f))
        assert r <= 1e-7 and r <= max(rel(ref, pert.download(f)), 1e-10), (f, r)
